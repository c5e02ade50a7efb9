set -uo pipefail
mkdir -p gpurun_out/r3h
timeout -k 10 300 python -m pytest tests/test_gpu_strict.py -x -q -k "not config5" > gpurun_out/r3h/strict_tests.log 2>&1; tail -3 gpurun_out/r3h/strict_tests.log
rm -f gpurun_out/accuracy_all_bodies.jsonl
timeout -k 10 600 python tests/measure/all_bodies_survey.py uniform20 uniform20_2d > gpurun_out/r3h/survey.log 2>&1
cp gpurun_out/accuracy_all_bodies.jsonl gpurun_out/r3h/accuracy.jsonl
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3h/prof_refine -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --refine 1e-5 > $GRAFT_REPO_ROOT/gpurun_out/r3h/bench_refine_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3h/prof_refine.err )
python - <<'PY'
import json,csv,glob
for l in open('gpurun_out/r3h/accuracy.jsonl'):
    d=json.loads(l); print(d['what'],'mixed',{k:d['mixed'][k] for k in ('max_rel','n_over_tol','selected','refined','cost_ms')},'default_ms',d['default_ms'])
for r in csv.DictReader(open(glob.glob('gpurun_out/r3h/prof_refine/*kernel_stats.csv')[0])):
    if 'accel' in r['Name'] or 'refine' in r['Name']: print(r['Name'][-60:], r['Calls'], r['AverageNs'])
PY
