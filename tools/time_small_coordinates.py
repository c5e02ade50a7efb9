"""Step time of a unit-scale system (every body a close-set candidate) at N = 2^20: the fast kernel with the sorted-cell
close-set refinement (default) against the guarded kernel that round 1 fell back to for such inputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nbx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rng = np.random.default_rng(1)
b = np.zeros((n, 7))
b[:, :3] = rng.normal(size=(n, 3))               # Gaussian blob of unit size around the origin
b[:, 6] = 1.0 / n
names = nbx.variants()
exact = [i for i, v in enumerate(names) if "exact" in v][0]
for label, variant in (("default", -1), ("guarded kernel", exact)):
    with nbx.Context(n, 3) as c:
        c.upload(b)
        c.set_tuning(0, variant)
        c.step(1e-9, 1, 1e-30)                   # warm-up
        c.synchronize(); c.kernel_time()
        t0 = time.perf_counter()
        c.step(1e-9, 5, 1e-30)
        c.synchronize()
        wall = (time.perf_counter() - t0) / 5
        ms, _ = c.kernel_time()
        print(f"{label:15s} variant {c.effective_tuning()[0]:22s} {wall*1e3:8.2f} ms per step (force kernel / graph step {ms:8.2f} ms)  "
              f"{n*n/wall:.3e} pair-interactions/s", flush=True)
