#!/usr/bin/env bash
# A/B build of the library WITH the measured-and-lost variants of the force kernel compiled in (this round's candidates under NBX_AB in
# csrc/force_kernel.hip; earlier rounds' lost variants live in the history): nbody-simulation-parallel_amd/libnbody_hip_ab.so.  Select it for a Python tool with NBODY_HIP_LIBRARY=<path>,
# e.g.  NBODY_HIP_LIBRARY=$PWD/nbody-simulation-parallel_amd/libnbody_hip_ab.so python tools/time_variants.py 1048576 4 fastpk
# Measurement aid; not part of `make`.
set -euo pipefail
cd "$(dirname "$0")/.."
P=nbody-simulation-parallel_amd
DEFS="${1:--DNBX_AB}"   # no candidate is compiled in at present: add table entries under #ifdef NBX_AB in force_kernel.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Iinclude $DEFS -fno-slp-vectorize \
    -c $P/csrc/force_kernel.hip -o $P/csrc/force_kernel_ab.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $P/libnbody_hip_ab.so $P/csrc/force_kernel_ab.o $P/csrc/force_launch.o $P/csrc/state_kernels.o \
    $P/csrc/nbx_api.o $P/csrc/nbx_node.o $P/csrc/leaf_pair_kernel.o $P/csrc/close_hash.o $P/csrc/measure_kernels.o -ldl -Wl,--version-script=$P/csrc/libnbody_hip.map
echo built $P/libnbody_hip_ab.so
