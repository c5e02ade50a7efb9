#!/usr/bin/env bash
# rocprofv3 --pmc passes (separate) over tools/time_leaf_pairs.py for one leaf shape; per-kernel per-launch means of every counter
# for the pair kernels are printed by the awk at the end.   tools/profile_leaf.sh <shape> <tag>      (through gpurun, from the repo root)
set -uo pipefail
shape="${1:-bvh8}"; tag="${2:-leaf}"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/pmc_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --output-format csv -d "$out/p$i" -o leaf -- python3 "$root/tools/time_leaf_pairs.py" 1048576 "$shape" > "$out/run$i.txt" 2> "$out/err$i.txt" || { echo "pass $i failed"; tail -5 "$out/err$i.txt"; }
done
for f in "$out"/p*/leaf_counter_collection.csv; do
    python3 - "$f" <<'PY'
import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "leaf_pack_kernel" in k or "leaf_pair_kernel" in k:
        per[(k.split("(")[0][-60:], r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
for (k, c), d in sorted(per.items()):
    print(f"{k:62s} {c:32s} launches={len(d):4d} mean={sum(d.values()) / len(d):.5g}")
PY
done
