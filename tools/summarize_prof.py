#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_bench.sh (gpurun_out/prof_<tag>/) into the small summaries that are
committed under profiles/<tag>/:
  bench_kernel_stats.csv     the --kernel-trace --stats table (per-kernel calls / average / min / max)
  bench_under_rocprof.json   bench.py's own JSON line from the trace pass (its HIP-event kernel time must agree)
  pmc_force_kernel.json      per-launch means of every collected counter for the dominant force kernel, per pass
    python tools/summarize_prof.py <tag>"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    stats = os.path.join(src, "trace", "bench_kernel_stats.csv")
    dominant = None
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(dst, "bench_kernel_stats.csv"))
        rows = list(csv.DictReader(open(stats)))
        force = [r for r in rows if "accel_" in r["Name"]]
        if force:
            dominant = max(force, key=lambda r: float(r["TotalDurationNs"]))["Name"]
    bj = os.path.join(src, "bench_trace.json")
    if os.path.exists(bj):
        for line in open(bj):
            if line.startswith("{"):
                with open(os.path.join(dst, "bench_under_rocprof.json"), "w") as f:
                    f.write(line)
    out = {}
    for sub in sorted(os.listdir(src)):
        path = os.path.join(src, sub, "bench_counter_collection.csv")
        if not sub.startswith("pmc_") or not os.path.exists(path):
            continue
        sums, counts, meta = defaultdict(float), defaultdict(int), None
        per_dispatch = defaultdict(dict)
        for r in csv.DictReader(open(path)):
            if "accel_" not in r["Kernel_Name"] or (dominant and r["Kernel_Name"] != dominant):
                continue
            per_dispatch[r["Dispatch_Id"]][r["Counter_Name"]] = per_dispatch[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size") if k in r}
        for d in per_dispatch.values():
            for k, v in d.items():
                sums[k] += v
                counts[k] += 1
        if meta:
            out[sub] = {"kernel": meta, "per_launch_mean": {k: sums[k] / counts[k] for k in sums}, "launches": dict(counts)}
    # Calibration of FETCH_SIZE / WRITE_SIZE in THIS path's access pattern (guide: "calibrate on a known byte count in your own
    # access pattern"): the helper kernels of the same run read and write exactly known byte counts with the same coalesced
    # 4-/8-byte-per-lane streams -- classify_*_kernel reads 12 B per body, kick_drift_kernel reads (12 S + 56) B and writes
    # 60 B per body (S = source slices).
    cfg = {}
    try:
        cfg = json.loads(open(os.path.join(dst, "bench_under_rocprof.json")).read())["config"]
    except Exception:
        pass
    n, S = cfg.get("n_bodies"), cfg.get("acc_planes", cfg.get("source_slices"))   # planes of partial sums the consumers read
    if n and S:
        known = {"classify_close_kernel": (12.0 * n, None), "classify_sources_kernel": (12.0 * n, None),
                 "kick_drift_kernel": ((12.0 * S + 56.0) * n, 60.0 * n)}
        calib = {}
        for sub, counter, col in (("pmc_fetch", "FETCH_SIZE", 0), ("pmc_write", "WRITE_SIZE", 1)):
            path = os.path.join(src, sub, "bench_counter_collection.csv")
            if not os.path.exists(path):
                continue
            per = defaultdict(lambda: defaultdict(float))
            for r in csv.DictReader(open(path)):
                for k in known:
                    if k in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        per[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
            for k, d in per.items():
                true_bytes = known[k][col]
                if true_bytes and d:
                    mean_kb = sum(d.values()) / len(d)
                    calib.setdefault(k, {})[counter] = {"reported_KB": mean_kb, "true_KB": true_bytes / 1024.0,
                                                        "true_over_reported": true_bytes / 1024.0 / mean_kb, "launches": len(d)}
        if calib:
            out["calibration"] = calib
    if out:
        with open(os.path.join(dst, "pmc_force_kernel.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
    print("dominant force kernel:", dominant)
    print("wrote", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
