#!/usr/bin/env bash
# Sweep of the harness over the reference's problem sizes (nbody-sim-new/run_simulations.sh:26-60), written
# for this build: HIP rows at every size and both dimensions (`-m g`, never gated by the 1e6-body CPU limit),
# CPU brute-force rows + accuracy column at the four smallest sizes (`-m a -a 1`).  A failing run is logged and
# the sweep moves on, like the reference's.  Results land in results/ in the reference's file/CSV format;
# tools/aggregate_results.py then writes analysis-compatible averages.
#   tools/run_sweep.sh [extra nbody_sim args]
set -uo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
exe="$root/nbody_sim"
[ -x "$exe" ] || make -C "$root" nbody_sim || { echo "Build failed. Exiting."; exit 1; }
sizes=(1000 10000 100000 200000 500000 1000000 2000000 5000000)
run() {  # N dim accuracy methods
  echo "Running simulation for N=$1, dimension=$2, accuracy=$3, methods=$4"
  "$exe" -N "$1" -d "$2" -a "$3" -m "$4" "${@:5}" > /dev/null || echo "Simulation failed for N=$1, dimension=$2 -- moving on"
  echo "------------------------------------------"
}
for dim in 2 3; do
  for n in "${sizes[@]}"; do run "$n" "$dim" 0 g "$@"; done
done
for dim in 2 3; do
  for n in "${sizes[@]:0:4}"; do run "$n" "$dim" 1 a "$@"; done
done
python3 "$root/tools/aggregate_results.py" results
echo "Results are available in the results directory"
