"""Where the wall time of the drop-in call goes: nbx_brute_force_forces (what BruteForce_HIP's CSV row times, SURVEY 8b
"timing semantics": the whole call, pack + H2D + kernel + D2H) against its phases through the context API."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nbx

sizes = [int(a) for a in sys.argv[1:]] or [1000, 10000, 100000, 1 << 20]
nbx.brute_force_hip_n_body(nbx.uniform_bodies(64, 3, 1))          # runtime + code object load, not what is timed


def clock(f, reps):
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3


for n in sizes:
    b = nbx.uniform_bodies(n, 3, 1)
    reps = 5 if n <= 200000 else 3
    one = clock(lambda: nbx.brute_force_hip_n_body(b), reps)
    ph = {}
    t = time.perf_counter()
    c = nbx.Context(n, 3); ph["create"] = time.perf_counter() - t
    t = time.perf_counter(); c.upload(b); ph["upload"] = time.perf_counter() - t
    t = time.perf_counter(); c.compute_accel(nbx.SRC_ALL); c.synchronize(); ph["accel(first)"] = time.perf_counter() - t
    t = time.perf_counter(); c.compute_accel(nbx.SRC_ALL); c.synchronize(); ph["accel(again)"] = time.perf_counter() - t
    t = time.perf_counter(); f = c.forces(); ph["forces"] = time.perf_counter() - t
    ms, _ = c.kernel_time()
    t = time.perf_counter(); c.close(); ph["destroy"] = time.perf_counter() - t
    print(f"N={n}: one-shot {one:9.3f} ms (best of {reps});  kernel {ms:8.3f} ms;  " +
          "  ".join(f"{k} {v*1e3:.3f}" for k, v in ph.items()), flush=True)
