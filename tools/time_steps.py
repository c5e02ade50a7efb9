"""Wall time per kick/drift step at small N (launch-bound regime), BASELINE config 2: N=65,536, 100 steps;
hipGraph replay (nbx_ctx_step default) against eager launches (NBODY_HIP_NO_GRAPHS=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nbx
for n in (4096, 16384, 65536, 262144):
    b = nbx.uniform_bodies(n, 3, 1)
    row = []
    for mode in ("graph", "eager"):
        os.environ["NBODY_HIP_NO_GRAPHS"] = "0" if mode == "graph" else "1"
        with nbx.Context(n, 3) as c:
            c.upload(b)
            c.step(1.0, 5); c.synchronize()
            t0 = time.perf_counter()
            c.step(1.0, 100); c.synchronize()
            dt = (time.perf_counter() - t0) / 100
            row.append(f"{mode} {dt*1e3:8.4f} ms/step = {n*n/dt/1e12:6.3f} T-pairs/s")
            tuning = c.effective_tuning()
    print(f"N={n:7d}: " + " | ".join(row) + f"  tuning {tuning}", flush=True)
