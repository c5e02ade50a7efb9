# What a round is checked with on the GPU box (through gpurun): the whole -m gpu suite, smoke(), the bench line, the rocprofv3 passes of
# the bench command, the leaf-pair path at the leaf sizes of interest.  Outputs under gpurun_out/check/; the summaries worth keeping are
# copied into profiles/<round>/ by hand.  A step that fails or runs into its time limit ends the script: no further GPU step is started.
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
timeout -k 10 500 bash tools/profile_bench.sh check > $O/profile_bench.log 2>&1 || { tail -20 $O/profile_bench.log; exit 1; }
for shape in 5 bvh16; do
    timeout -k 10 300 python tools/time_leaf_pairs.py 1048576 $shape > $O/time_leaf_pairs_$shape.txt 2>&1 || { tail -20 $O/time_leaf_pairs_$shape.txt; exit 1; }
    grep "tree_leaf" $O/time_leaf_pairs_$shape.txt
done
./nbody_sim -N 1048576 -d 3 -m p --seed 5 --steps 5 2>&1 | grep -E "Time taken|Resident plan|pair kernel|Near-field stepping" | tee $O/nbody_sim_near_field.txt
