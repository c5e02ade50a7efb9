"""CPU baseline of record: the reference's own brute-force object code
(oracle/_ref/libnbody_ref.so = nbody-sim-new/methods.cpp built with the reference Makefile's flags) timed on the GPU
box's host cores, same seeded uniform bodies as the GPU runs.  Thread count = the box's CPU share for one GPU."""
import os, sys, time, subprocess
os.environ.setdefault("OMP_NUM_THREADS", "16")
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import nbody_amd as nbx
from oracle_lib import Reference, Oracle
ref, o = Reference(), Oracle()
print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Core|Socket'", shell=True, capture_output=True, text=True).stdout)
print("OMP_NUM_THREADS", os.environ["OMP_NUM_THREADS"], "oracle sees", o.num_threads(), "threads", flush=True)
for n, variants in ((65536, ((0, "brute_force_seq_n_body"), (1, "brute_force_omp_n_body_1"), (2, "brute_force_omp_n_body_2"))),
                    (262144, ((1, "brute_force_omp_n_body_1"), (2, "brute_force_omp_n_body_2")))):
    b = nbx.uniform_bodies(n, 3, 1)
    for v, name in variants:
        t0 = time.perf_counter(); ref.brute_force(v, b); dt = time.perf_counter() - t0
        print(f"{name}<3>  N={n:7d}  {dt:9.3f} s  {n*(n-1)/dt:.3e} pair-interactions/s" +
              (f"  (N=2^20 extrapolated: {dt*(1048576/n)**2:8.1f} s)" if n != 65536 else ""), flush=True)
