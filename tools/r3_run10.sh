set -uo pipefail
mkdir -p gpurun_out/r3i
for V in 0 1 2 3 4; do
( cd /tmp && export TMPDIR=/tmp NBX_LIST_VARIANT=$V && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3i/prof_v$V -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --refine 1e-5 > $GRAFT_REPO_ROOT/gpurun_out/r3i/bench_v$V.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3i/prof_v$V.err )
python - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob('gpurun_out/r3i/prof_v$V/*kernel_stats.csv')[0])):
    if 'accel_f64' in r['Name']: print('variant $V', r['Name'][-45:], r['Calls'], r['AverageNs'])
PY
done
