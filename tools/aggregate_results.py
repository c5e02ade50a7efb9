#!/usr/bin/env python3
"""Average the per-run CSVs written by nbody_sim (results/run_*_N_*_*D.csv) into one table with the columns
of the reference's analysis/aggregated_results.csv -- `Bodies,Method,Dimension,Average Runtime (s)` -- so the
BruteForce_HIP rows no longer have to be typed in by hand (the reference notebook's cell 4 did that for its
CUDA numbers).  A second file carries kernel time and pair-interactions/s from the *_hip.csv sidecars.
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
                hip[(int(row["Bodies"]), int(row["Dimension"]))].append(
                    (float(row["Time(s)"]), float(row["KernelTime(s)"]), float(row["PairInteractionsPerSec"])))
    out2 = os.path.join(d, "aggregated_hip.csv")
    with open(out2, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Bodies", "Dimension", "Runs", "Average Runtime (s)", "Average Kernel Time (s)", "Pair Interactions/s (kernel)",
                    "Fraction of MI355X fp32 peak (20 flop/pair)"])
        for (n, dim), rows in sorted(hip.items()):
            k = len(rows)
            rate = sum(r[2] for r in rows) / k
            w.writerow([n, dim, k, sum(r[0] for r in rows) / k, sum(r[1] for r in rows) / k, rate, rate * 20 / 157.3e12])
    print(f"wrote {out} ({len(runs)} rows) and {out2} ({len(hip)} rows)")


if __name__ == "__main__":
    main()
