set -uo pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests/test_gpu_leaf_pairs.py -x -q -s > gpurun_out/r3/leaf_tests2.log 2>&1; tail -4 gpurun_out/r3/leaf_tests2.log
python tools/time_leaf_pairs.py > gpurun_out/r3/time_leaf_pairs2.txt 2>&1; cat gpurun_out/r3/time_leaf_pairs2.txt
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/pmc_leaf1b -o leaf -- python3 $GRAFT_REPO_ROOT/tools/time_leaf_pairs.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3/pmc_leaf1b.err )
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/pmc_leaf2b -o leaf -- python3 $GRAFT_REPO_ROOT/tools/time_leaf_pairs.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3/pmc_leaf2b.err )
python tools/time_shards.py 1048576 8:16,24,32,48,64 4:16,24,32 2:16 > gpurun_out/r3/shard_slices_sweep.txt 2>&1; cat gpurun_out/r3/shard_slices_sweep.txt
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_leaf_pairs.py > gpurun_out/r3/gpu_tests_run4.log 2>&1; tail -15 gpurun_out/r3/gpu_tests_run4.log
