set -uo pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests/test_gpu_leaf_pairs.py -x -q > gpurun_out/r3/leaf_tests3.log 2>&1; tail -3 gpurun_out/r3/leaf_tests3.log
python tools/time_leaf_pairs.py > gpurun_out/r3/time_leaf_pairs3.txt 2>&1; cat gpurun_out/r3/time_leaf_pairs3.txt
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/pmc_leaf1c -o leaf -- python3 $GRAFT_REPO_ROOT/tools/time_leaf_pairs.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3/pmc_leaf1c.err )
python - <<'PY'
import csv,glob
from collections import defaultdict
f=glob.glob('gpurun_out/r3/pmc_leaf1c/*counter_collection.csv')[0]
per=defaultdict(float); n=defaultdict(int)
for r in csv.DictReader(open(f)):
    if 'leaf_pair_kernel<3, 1>' in r['Kernel_Name']:
        per[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print({k:'%.4g'%(v/n[k]*8 if False else v/n[k]) for k,v in per.items()})
PY
