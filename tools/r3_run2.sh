set -uo pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests/test_gpu_strict.py -x -q -k "not every_body" > gpurun_out/r3/strict_tests.log 2>&1; tail -3 gpurun_out/r3/strict_tests.log
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3/bench_default_nocheck.json 2> gpurun_out/r3/bench_default_nocheck.err
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --refine 1e-5 > gpurun_out/r3/bench_refine_nocheck.json 2> gpurun_out/r3/bench_refine_nocheck.err
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3/bench_default_nocheck2.json 2>> gpurun_out/r3/bench_default_nocheck.err
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --refine 1e-5 > gpurun_out/r3/bench_refine_nocheck2.json 2>> gpurun_out/r3/bench_refine_nocheck.err
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/prof_refine -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --refine 1e-5 > $GRAFT_REPO_ROOT/gpurun_out/r3/bench_refine_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3/prof_refine.err )
timeout -k 10 400 python bench.py > gpurun_out/r3/bench_full.json 2> gpurun_out/r3/bench_full.err; tail -c 600 gpurun_out/r3/bench_full.err
for R in 2 4; do
NBODY_BENCH_BACKEND=gloo NBODY_BENCH_DEVICE=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port 2951$R bench.py --gpus $R --steps 3 --warmup 1 > gpurun_out/r3/bench_${R}rank_rehearsal.json 2> gpurun_out/r3/bench_${R}rank_rehearsal.err; tail -c 400 gpurun_out/r3/bench_${R}rank_rehearsal.err
done
ls -la gpurun_out/r3
