#!/usr/bin/env bash
# BASELINE config 5 in full: N = 4,194,304 Plummer sphere, 1000 kick/drift steps, energy logged every 50 steps,
# sharded over the GPUs of one node (one process, RCCL all-gather of positions per step on a second stream).
# Parameters as stated in tests/test_gpu_config5.py: a = 1e5, M = 1e12, seed 5, G = 0.05, dt = 0.5
# (the unsoftened law makes the closest pair, not the sphere, set the admissible G dt^2 -- see that file).
#   tools/run_config5.sh            # 8 GPUs, 1000 steps of the reference's unsoftened law (~8 min at 8 GPUs)
#   GPUS=1 STEPS=20 tools/run_config5.sh
#   SOFTENING=600 G=1e4 tools/run_config5.sh   # physically resolved variant: softened at the inter-particle scale
#                                              # (median nearest neighbour 1230), t_dyn = 100, 1000 steps = 5 t_dyn
#   LAW=newton SOFTENING=600 G=0.1 tools/run_config5.sh   # the same sphere under the attractive softened Newtonian law:
#                                              # a stationary solution (t_dyn = sqrt(a^3/(G M)) = 100), energy and virial ratio hold
# A run can be cut into segments where a job may not last 1000 steps (one GPU: 3.5 s per step):
#   STEPS=250 DUMP=seg1 tools/run_config5.sh
#   STEPS=250 LOAD=seg1_Leapfrog_HIP.f64 STEP_OFFSET=250 E0=<E of step 0> DUMP=seg2 tools/run_config5.sh   ... and so on;
# k + k steps through a dump/load equal 2k steps bit for bit (tests/test_gpu_harness.py).  The state file is 56 B per body
# (235 MB at this N), so the segments need storage that survives between jobs.
set -euo pipefail
cd "$(dirname "$0")/.."
GPUS="${GPUS:-8}"
N="${N:-4194304}"                 # BASELINE config 5's size; smaller values are for rehearsals
DEVICES="${DEVICES:-}"            # explicit device list instead of 0..GPUS-1, e.g. 0,0 = two virtual ranks on one GPU
STEPS="${STEPS:-1000}"
EVERY="${EVERY:-50}"
G="${G:-0.05}"
SOFTENING="${SOFTENING:-0}"
LAW="${LAW:-reference}"
INTEGRATOR="${INTEGRATOR:-kd}"       # kd = the reference helpers' order (kick, drift); kdk = synchronised leapfrog (extension)
LOAD="${LOAD:-}"; STEP_OFFSET="${STEP_OFFSET:-0}"; E0="${E0:-}"; DUMP="${DUMP:-}"
OUT="${OUT:-gpurun_out/config5_${GPUS}gpu_${STEPS}steps.log}"
mkdir -p "$(dirname "$OUT")"
[ -x ./nbody_sim ] || make nbody_sim
if [ -n "$DEVICES" ]; then where=(--devices "$DEVICES"); else where=(--gpus "$GPUS"); fi
[ -n "$LOAD" ] && where+=(--load "$LOAD" --step-offset "$STEP_OFFSET")
[ -n "$E0" ] && where+=(--e0 "$E0")
[ -n "$DUMP" ] && where+=(--dump "$DUMP")
./nbody_sim -N "$N" -d 3 -m g --init plummer --seed 5 --G "$G" --law "$LAW" --integrator "$INTEGRATOR" --softening "$SOFTENING" --dt 0.5 --steps "$STEPS" --energy-every "$EVERY" "${where[@]}" | tee "$OUT"
grep -E "^step |Time taken|Kernel time" "$OUT" > "${OUT%.log}.summary.txt"
echo "summary: ${OUT%.log}.summary.txt"
