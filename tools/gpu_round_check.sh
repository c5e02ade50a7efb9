# What a round is checked with on the GPU box (through gpurun): the whole -m gpu suite, smoke(), the bench line, the mixed-mode and
# leaf-pair kernel traces.  Outputs under gpurun_out/check/; the summaries worth keeping are copied into profiles/<round>/ by hand.
# A step that fails or runs into its time limit ends the script: no further GPU step is started after it.
set -euo pipefail
O=gpurun_out/check
R="${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p $O
rm -f gpurun_out/accuracy_all_bodies.jsonl
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gpu_tests_run.log 2>&1 || { tail -30 $O/gpu_tests_run.log; exit 1; }
tail -4 $O/gpu_tests_run.log
cp gpurun_out/accuracy_all_bodies.jsonl $O/accuracy_all_bodies.jsonl
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench_full.json 2> $O/bench_full.err || { tail -20 $O/bench_full.err; exit 1; }
cat $O/bench_full.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_refine -o bench -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --refine 1e-5 > $R/$O/bench_refine_prof.json 2> $R/$O/prof_refine.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_leaf -o leaf -- python3 $R/tools/time_leaf_pairs.py > $R/$O/time_leaf_pairs.txt 2> $R/$O/prof_leaf.err
tail -3 $R/$O/time_leaf_pairs.txt
cd $R
timeout -k 10 300 python tests/measure/all_bodies_survey.py uniform20_2d > $O/survey_2d.log 2>&1 || { tail -20 $O/survey_2d.log; exit 1; }
cp gpurun_out/accuracy_all_bodies.jsonl $O/accuracy_after_survey.jsonl
NBX_LEAF_TIMING_REPS=300 ./nbody_sim -N 1048576 -d 3 -m p --seed 5 2>&1 | grep "Time taken" | tee $O/nbody_sim_near_field.txt
./nbody_sim -N 1048576 -d 3 -m p --seed 5 2>&1 | grep "Time taken" | tee -a $O/nbody_sim_near_field.txt
