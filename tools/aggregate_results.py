#!/usr/bin/env python3
"""Average the per-run CSVs written by nbody_sim (results/run_*_N_*_*D.csv) into one table with the columns
of the reference's analysis/aggregated_results.csv -- `Bodies,Method,Dimension,Average Runtime (s)` -- so the
BruteForce_HIP rows no longer have to be typed in by hand (the reference notebook's cell 4 did that for its
CUDA numbers; rows sharded over G GPUs carry the label BruteForce_HIP_x<G>).  A second file, aggregated_hip.csv,
carries the GPU count, kernel time, pair-interactions/s and the speed-up over the one-GPU row from the *_hip.csv
sidecars (per-N rows for 1/2/4/8 GPUs).
    python tools/aggregate_results.py [results_dir]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "results"
    runs = defaultdict(list)
    for path in sorted(glob.glob(os.path.join(d, "run_*_N_*_*D.csv"))):
        if path.endswith("_hip.csv"):
            continue
        with open(path) as f:
            for row in csv.DictReader(f):
                try:
                    runs[(int(row["Bodies"]), row["Method"], int(row["Dimension"]))].append(float(row["Time(s)"]))
                except (KeyError, ValueError):
                    continue
    out = os.path.join(d, "aggregated_results.csv")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Bodies", "Method", "Dimension", "Average Runtime (s)"])
        for (n, m, dim), ts in sorted(runs.items()):
            w.writerow([n, m, dim, sum(ts) / len(ts)])
    hip = defaultdict(list)
    for path in sorted(glob.glob(os.path.join(d, "run_*_hip.csv"))):
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (int(row["Bodies"]), int(row["Dimension"]), int(row.get("GPUs", 1) or 1), int(row.get("DistinctDevices", 1) or 1))
                hip[key].append((float(row["Time(s)"]), float(row["KernelTime(s)"]), float(row["PairInteractionsPerSec"])))
    one_gpu = {(n, dim): sum(r[1] for r in rows) / len(rows) for (n, dim, g, dd), rows in hip.items() if g == 1}
    out2 = os.path.join(d, "aggregated_hip.csv")
    with open(out2, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Bodies", "Dimension", "GPUs", "Distinct Devices", "Virtual Ranks", "Runs", "Average Runtime (s)",
                    "Average Kernel Time (s)", "Pair Interactions/s (kernel)", "Fraction of fp32 peak of the GPUs used (20 flop/pair)",
                    "Kernel Speed-up vs 1 GPU"])
        for (n, dim, g, dd), rows in sorted(hip.items()):
            k = len(rows)
            rate = sum(r[2] for r in rows) / k
            kern = sum(r[1] for r in rows) / k
            base = one_gpu.get((n, dim))
            # Ranks that share a device ("virtual ranks": the sharded code path rehearsed on fewer GPUs) time-slice that device:
            # their per-rank kernel time is no per-GPU figure, so rate, fraction of peak and speed-up are left to the
            # whole-call runtime column for those rows.
            virtual = dd < g
            w.writerow([n, dim, g, dd, "yes" if virtual else "no", k, sum(r[0] for r in rows) / k, "" if virtual else kern,
                        "" if virtual else rate, "" if virtual else rate * 20 / (157.3e12 * dd),
                        "" if virtual or not (base and kern > 0) else base / kern])
    print(f"wrote {out} ({len(runs)} rows) and {out2} ({len(hip)} rows)")


if __name__ == "__main__":
    main()
