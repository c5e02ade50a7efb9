#!/usr/bin/env bash
# Build the REFERENCE's own hot-path object code into oracle/_ref/libnbody_ref.so.
# Runs only where /root/reference exists (the development container); the GPU box receives the
# prebuilt .so.  Nothing is copied from the reference: its sources are compiled where they lie.
#
# Recipe: the reference's Makefile flags (nbody-sim-new/Makefile:2), plus
#  * -DMultipoleExpansion='Expansion<D,10>'  -- fmm_omp.cpp:228 names an undeclared type; the FMM
#    code is never reached from the brute-force path, the macro only lets the TU parse;
#  * --unresolved-symbols=ignore-all -- FMM members that the reference declares but never defines
#    (FMM_OMP<D>::{m2l,l2l,l2p,p2p}_phase, FMMNode<D>::{translate_local_to_children,
#    compute_direct_forces}) stay unresolved; brute force never calls them.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ref="${NBODY_REFERENCE_DIR:-/root/reference}"
out="$here/_ref"
if [ ! -d "$ref/nbody-sim-new" ]; then
  echo "build_ref.sh: $ref/nbody-sim-new not present -- keeping any prebuilt $out" >&2
  exit 0
fi
mkdir -p "$out"
flags=(-std=c++17 -O3 -fopenmp -fPIC -I "$ref/parlaylib/include" -I "$ref/nbody-sim-new" '-DMultipoleExpansion=Expansion<D,10>' -w)
if [ ! -f "$out/methods_ref.o" ] || [ "$ref/nbody-sim-new/methods.cpp" -nt "$out/methods_ref.o" ]; then
  g++ "${flags[@]}" -c "$ref/nbody-sim-new/methods.cpp" -o "$out/methods_ref.o"
fi
g++ "${flags[@]}" -c "$here/ref_driver.cpp" -o "$out/ref_driver.o"
g++ -shared -fopenmp -o "$out/libnbody_ref.so" "$out/ref_driver.o" "$out/methods_ref.o" \
    -Wl,--unresolved-symbols=ignore-all -lpthread
echo "built $out/libnbody_ref.so"
