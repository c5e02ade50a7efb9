set -uo pipefail
mkdir -p gpurun_out/r3f
rm -f gpurun_out/accuracy_all_bodies.jsonl
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > gpurun_out/r3f/gpu_tests_run.log 2>&1; tail -5 gpurun_out/r3f/gpu_tests_run.log
cp gpurun_out/accuracy_all_bodies.jsonl gpurun_out/r3f/ 2>/dev/null
python tools/time_leaf_pairs.py > gpurun_out/r3f/time_leaf_pairs.txt 2>&1; tail -3 gpurun_out/r3f/time_leaf_pairs.txt
bash tools/profile_bench.sh r3 > gpurun_out/r3f/profile_bench.log 2>&1; tail -3 gpurun_out/r3f/profile_bench.log
timeout -k 10 400 python bench.py > gpurun_out/r3f/bench_full.json 2> gpurun_out/r3f/bench_full.err; tail -c 300 gpurun_out/r3f/bench_full.err
for R in 2 4; do
NBODY_BENCH_BACKEND=gloo NBODY_BENCH_DEVICE=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port 2961$R bench.py --gpus $R --steps 3 --warmup 1 > gpurun_out/r3f/bench_${R}rank_rehearsal.json 2> gpurun_out/r3f/bench_${R}rank_rehearsal.err; tail -c 200 gpurun_out/r3f/bench_${R}rank_rehearsal.err
done
ls gpurun_out/r3f
