#!/usr/bin/env bash
# A/B build of the library with the leaf-pair path compiled differently: nbody-simulation-parallel_amd/libnbody_hip_leafab.so
# (OUT=<name> in the environment for another file name).
#   tools/build_ab_leaf.sh -DNBX_LEAF_PACK=0        without packed small leaves (csrc/leaf_plan.h PackBlock)
# Select it for a Python tool with NBODY_HIP_LIBRARY=<path> (capi.py).  Measurement aid; not part of `make`.
set -euo pipefail
cd "$(dirname "$0")/.."
P=nbody-simulation-parallel_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Iinclude "$@" \
    -c $P/csrc/leaf_pair_kernel.hip -o $P/csrc/leaf_pair_kernel_ab.o
OUT="${OUT:-libnbody_hip_leafab.so}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $P/$OUT $P/csrc/force_kernel.o $P/csrc/force_launch.o $P/csrc/state_kernels.o \
    $P/csrc/nbx_api.o $P/csrc/nbx_node.o $P/csrc/leaf_pair_kernel_ab.o $P/csrc/close_hash.o $P/csrc/measure_kernels.o -ldl -Wl,--version-script=$P/csrc/libnbody_hip.map
echo built $P/$OUT "($*)"
