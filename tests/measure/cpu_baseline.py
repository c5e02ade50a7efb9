"""CPU baseline of record: the reference's own brute-force object code
(oracle/_ref/libnbody_ref.so = nbody-sim-new/methods.cpp built with the reference Makefile's flags) timed on the
GPU box's host cores, same seeded uniform bodies as the GPU runs.

Two thread counts: the box's CPU share for one GPU (16) and all physical cores (sockets x cores, capped by the
affinity mask).  NOTE the header line's `cgroup cpu.max`: the GPU boxes of this pool grant 16 CPUs of bandwidth
(1600000/100000), so a 128-thread run there is time-sliced onto 16 CPUs' worth and is NOT an all-cores figure.  OpenMP and ParlayLib fix their pools at start-up, so each
configuration runs in its own child process (this script re-invoked with --child).  One JSON line per
(threads, solver, N) is appended to the output file; the header line carries lscpu / affinity / cgroup facts.

usage: python tests/measure/cpu_baseline.py [--out profiles/r2/cpu_baseline.jsonl] [--sizes 65536,262144] [--threads 16,all]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SOLVERS = {0: "brute_force_seq_n_body", 1: "brute_force_omp_n_body_1", 2: "brute_force_omp_n_body_2",
           3: "brute_force_parlay_n_body_1", 4: "brute_force_parlay_n_body_2"}


def host_facts():
    facts = {}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True).stdout
        for line in out.splitlines():
            k, _, v = line.partition(":")
            if k.strip() in ("Model name", "CPU(s)", "Thread(s) per core", "Core(s) per socket", "Socket(s)", "NUMA node(s)"):
                facts[k.strip()] = v.strip()
    except OSError:
        pass
    facts["sched_affinity"] = len(os.sched_getaffinity(0))
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            facts["cgroup " + os.path.basename(p)] = open(p).read().strip()
        except OSError:
            pass
    return facts


def physical_cores(facts):
    try:
        phys = int(facts["Core(s) per socket"]) * int(facts["Socket(s)"])
    except (KeyError, ValueError):
        phys = facts["sched_affinity"]
    return max(1, min(phys, facts["sched_affinity"]))


def child(threads, sizes, dim, seed):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import nbody_amd as nbx
    from oracle_lib import Oracle, Reference
    ref, o = Reference(), Oracle()
    for n in sizes:
        b = nbx.uniform_bodies(n, dim, seed)
        for v in (0, 1, 2, 3, 4):
            if v == 0 and (n > 65536 or threads != 16):
                continue  # sequential path: once, at N = 65,536
            if v >= 3 and n > 65536:
                continue  # the ParlayLib twins run 5-10x slower than the OpenMP ones here: N = 65,536 only
            dt = ref.time_brute_force(v, b)
            pairs = n * (n - 1) / (2 if v in (0, 1, 3) else 1)   # symmetric variants evaluate each pair once
            print(json.dumps({"solver": SOLVERS[v] + f"<{dim}>", "n": n, "threads": 1 if v == 0 else threads,
                              "omp_threads_seen": o.num_threads(), "parlay_workers": ref.parlay_num_workers(),
                              "seconds": dt, "ordered_pair_interactions_per_s": n * (n - 1) / dt,
                              "pair_evaluations_per_s": pairs / dt,
                              "n_2pow20_extrapolated_s": dt * (1048576 / n) ** 2}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cpu_baseline.jsonl"))
    ap.add_argument("--sizes", default="65536,262144")
    ap.add_argument("--threads", default="16,all")
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--child", type=int, default=0)
    args = ap.parse_args()
    sizes = [int(s) for s in args.sizes.split(",")]
    if args.child:
        child(args.child, sizes, args.dim, args.seed)
        return
    facts = host_facts()
    phys = physical_cores(facts)
    counts = []
    for t in args.threads.split(","):
        c = phys if t == "all" else int(t)
        if c not in counts:
            counts.append(c)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write(json.dumps({"host": facts, "physical_cores_usable": phys, "thread_counts": counts,
                            "code": "reference methods.cpp object code, g++ -std=c++17 -O3 -fopenmp (oracle/build_ref.sh)"}) + "\n")
        f.flush()
        for c in counts:
            env = dict(os.environ, OMP_NUM_THREADS=str(c), PARLAY_NUM_THREADS=str(c), OMP_PROC_BIND="spread", OMP_PLACES="cores")
            t0 = time.perf_counter()
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(c), "--sizes", args.sizes,
                                  "--dim", str(args.dim), "--seed", str(args.seed)], env=env, stdout=subprocess.PIPE,
                                 stderr=subprocess.PIPE, text=True)
            for line in p.stdout:            # one line per finished solver: keep the file (and the terminal) moving
                f.write(line)
                f.flush()
                print(f"[cpu_baseline] {c} threads: {line[:110].rstrip()} ...", flush=True)
            err = p.stderr.read()
            if p.wait():
                f.write(json.dumps({"threads": c, "error": err[-400:]}) + "\n")
            f.flush()
            print(f"[cpu_baseline] {c} threads done in {time.perf_counter() - t0:.1f} s", flush=True)
    print(open(args.out).read())


if __name__ == "__main__":
    main()
