#!/usr/bin/env bash
# rocprofv3 passes over bench.py on the GPU box; summaries are copied into profiles/ afterwards.
#   tools/profile_bench.sh <tag>      (run through gpurun from the repo root)
set -uo pipefail
tag="${1:-r2}"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# pass 1: kernel trace + stats (per-kernel durations)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_trace.json" 2> "$out/trace.err" || { echo "trace pass failed"; tail -5 "$out/trace.err"; exit 1; }
# pass 2/3: HBM traffic counters, one pass each (FETCH_SIZE and WRITE_SIZE do not fit together)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o bench -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_fetch.json" 2> "$out/fetch.err" || { echo "fetch pass failed"; tail -5 "$out/fetch.err"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o bench -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_write.json" 2> "$out/write.err" || { echo "write pass failed"; tail -5 "$out/write.err"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out/pmc_sq" -o bench -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_sq.json" 2> "$out/sq.err" || { echo "sq pass failed"; tail -5 "$out/sq.err"; }
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_sq2" -o bench -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_sq2.json" 2> "$out/sq2.err" || { echo "sq2 pass failed"; tail -5 "$out/sq2.err"; }
find "$out" -name "*.csv" | head -40
python3 "$root/tools/summarize_prof.py" "$tag" || true   # summaries -> profiles/$tag (copied back by hand: profiles/ on the box is scratch)
mkdir -p "$root/gpurun_out/profiles_$tag" && cp -r "$root/profiles/$tag/." "$root/gpurun_out/profiles_$tag/"
