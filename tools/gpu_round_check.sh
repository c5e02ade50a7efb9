# What a round is checked with on the GPU box (through gpurun): the whole -m gpu suite, smoke(), the bench line, the mixed-mode and
# leaf-pair kernel traces.  Outputs under gpurun_out/r3z/; the summaries worth keeping are copied into profiles/<round>/ by hand.
set -uo pipefail
mkdir -p gpurun_out/r3z
rm -f gpurun_out/accuracy_all_bodies.jsonl
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3z/gpu_tests_run.log 2>&1; tail -4 gpurun_out/r3z/gpu_tests_run.log
cp gpurun_out/accuracy_all_bodies.jsonl gpurun_out/r3z/accuracy_all_bodies.jsonl
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3z/smoke.log 2>&1; tail -2 gpurun_out/r3z/smoke.log
timeout -k 10 400 python bench.py > gpurun_out/r3z/bench_full.json 2> gpurun_out/r3z/bench_full.err; tail -c 200 gpurun_out/r3z/bench_full.err
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3z/prof_refine -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --refine 1e-5 > $GRAFT_REPO_ROOT/gpurun_out/r3z/bench_refine_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3z/prof_refine.err )
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3z/prof_leaf -o leaf -- python3 $GRAFT_REPO_ROOT/tools/time_leaf_pairs.py > $GRAFT_REPO_ROOT/gpurun_out/r3z/time_leaf_pairs.txt 2> $GRAFT_REPO_ROOT/gpurun_out/r3z/prof_leaf.err )
tail -3 gpurun_out/r3z/time_leaf_pairs.txt
