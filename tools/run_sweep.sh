#!/usr/bin/env bash
# Sweep of the harness over the reference's problem sizes (nbody-sim-new/run_simulations.sh:26-60), written
# for this build: HIP rows at every size, both dimensions and 1/2/4/8 GPUs (`-m g`, never gated by the 1e6-body
# CPU limit), CPU brute-force rows + accuracy column at the four smallest sizes (`-m a -a 1`).
# A GPU count above the node's device count runs as virtual ranks (ranks round-robin over the devices: the same
# sharded code path, peer-copy exchange) and is marked as such in the sidecar's DistinctDevices column.
# A failing run is logged and the sweep moves on, like the reference's.  Results land in results/ in the
# reference's file/CSV format; tools/aggregate_results.py then writes analysis-compatible averages
# (aggregated_results.csv, the reference's four columns; aggregated_hip.csv, with the GPU count).
#   tools/run_sweep.sh [extra nbody_sim args]
#   SIZES="1000 10000" GPU_COUNTS="1 2" DIMS="3" ACC_SIZES="1000" tools/run_sweep.sh
set -uo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
# OpenMP rows: without OMP_NUM_THREADS the runtime starts one thread per visible hardware thread, and a container that
# shows 256 of them but grants 16 CPUs of bandwidth (cgroup cpu.max) then spends its time throttled -- a 0.2 ms row
# reads 200 ms.  Default to the CPUs the box really grants.
if [ -z "${OMP_NUM_THREADS:-}" ]; then
  cpus="$(nproc)"
  if [ -r /sys/fs/cgroup/cpu.max ] && read -r quota period < /sys/fs/cgroup/cpu.max && [ "$quota" != "max" ] && [ "${period:-0}" -gt 0 ]; then
    granted=$(( (quota + period - 1) / period ))
    [ "$granted" -ge 1 ] && [ "$granted" -lt "$cpus" ] && cpus="$granted"
  fi
  export OMP_NUM_THREADS="$cpus"
fi
echo "OpenMP rows use OMP_NUM_THREADS=$OMP_NUM_THREADS"
exe="$root/nbody_sim"
[ -x "$exe" ] || make -C "$root" nbody_sim || { echo "Build failed. Exiting."; exit 1; }
read -r -a sizes <<< "${SIZES:-1000 10000 100000 200000 500000 1000000 2000000 5000000}"
read -r -a gpu_counts <<< "${GPU_COUNTS:-1 2 4 8}"
read -r -a dims <<< "${DIMS:-2 3}"
read -r -a acc_sizes <<< "${ACC_SIZES-${sizes[*]:0:4}}"
ndev="$("$exe" --device-count 2>/dev/null || echo 0)"
[ "$ndev" -ge 1 ] 2>/dev/null || { echo "No HIP device: the HIP rows will be reported as failed"; ndev=1; }
device_list() {  # ranks -> "0,1,..." round-robin over the node's devices
  local g="$1" list="" i
  for ((i = 0; i < g; ++i)); do list+="${list:+,}$((i % ndev))"; done
  echo "$list"
}
run() {  # N dim accuracy methods [extra...]
  echo "Running simulation for N=$1, dimension=$2, accuracy=$3, methods=$4 ${*:5}"
  "$exe" -N "$1" -d "$2" -a "$3" -m "$4" "${@:5}" > /dev/null || echo "Simulation failed for N=$1, dimension=$2 -- moving on"
  sleep 1.1   # run ids (file names) have one-second resolution, like the reference's
  echo "------------------------------------------"
}
for dim in "${dims[@]}"; do
  for g in "${gpu_counts[@]}"; do
    for n in "${sizes[@]}"; do
      if [ "$g" -eq 1 ]; then run "$n" "$dim" 0 g "$@"; else run "$n" "$dim" 0 g --devices "$(device_list "$g")" "$@"; fi
    done
  done
done
for dim in "${dims[@]}"; do
  for n in "${acc_sizes[@]}"; do run "$n" "$dim" 1 a "$@"; done
done
python3 "$root/tools/aggregate_results.py" results
echo "Results are available in the results directory"
