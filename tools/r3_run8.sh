set -uo pipefail
mkdir -p gpurun_out/r3g
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3g/gpu_tests_run.log 2>&1; tail -4 gpurun_out/r3g/gpu_tests_run.log
rm -f gpurun_out/accuracy_all_bodies.jsonl
timeout -k 10 600 python tests/measure/all_bodies_survey.py uniform16 uniform20b uniform20_2d uniform22 > gpurun_out/r3g/survey_more.log 2>&1; tail -2 gpurun_out/r3g/survey_more.log | cut -c1-300
cp gpurun_out/accuracy_all_bodies.jsonl gpurun_out/r3g/accuracy_more_inputs.jsonl
