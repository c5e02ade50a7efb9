#!/usr/bin/env bash
# Kernel + memory-copy trace of the single-process node layer with 4 virtual ranks on one GPU: shows the
# per-step peer copies of the exchange running on the comm streams beside the LOCAL force passes.
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/prof_node"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out" -o node -- "$root/nbody_sim" -N 1048576 -m g --seed 1 --devices 0,0,0,0 --steps 3 --dt 1 > "$out/run.txt" 2> "$out/run.err" || { tail -5 "$out/run.err"; exit 1; }
ls "$out"
