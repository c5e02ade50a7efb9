"""Interleaved A/B timing of force-kernel variants in one process (guide rule 24).
   python tools/time_variants.py [N] [rounds] [name-substring ...]
NBX_TIME_PLAIN=1 in the environment times the plain fp32 builds (nbx_ctx_set_refine(0)); the default is the library's default
precision (mixed mode: the builds that also write the spread sums; the time is the force kernel's own, without the fp64 pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nbx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pats = sys.argv[3:]
names = nbx.variants()
sel = [i for i, nm in enumerate(names) if not pats or any(p in nm for p in pats)]
b = nbx.uniform_bodies(n, 3, 1)
with nbx.Context(n, 3) as c:
    c.upload(b)
    if os.environ.get("NBX_TIME_PLAIN"):
        c.set_refine(0.0)
    print("precision:", "plain fp32" if os.environ.get("NBX_TIME_PLAIN") else "mixed mode (spread-sum builds)", flush=True)
    res = {v: [] for v in sel}
    forces = {}
    for r in range(rounds + 1):
        for v in sel:
            c.set_tuning(0, v)
            c.compute_accel()
            ms, _ = c.kernel_time()
            if r: res[v].append(ms)
            elif os.environ.get("NBX_TIME_COMPARE"):
                forces[v] = c.forces()
    if forces:   # NBX_TIME_COMPARE=1: every variant's forces against the first selected one (builds of one summation must agree bit for bit)
        first = sel[0]
        scale = np.abs(forces[first]).max()
        for v in sel[1:]:
            d = np.abs(forces[v] - forces[first]).max() / scale
            print(f"forces of {names[v]} vs {names[first]}: max |dF| / max |F| = {d:.3e}" + ("  (identical)" if d == 0 else ""), flush=True)
    print(f"N={n}  interactions {n*n:.3e}  rounds {rounds}", flush=True)
    for v in sorted(sel, key=lambda v: min(res[v])):
        best, med = min(res[v]), sorted(res[v])[len(res[v]) // 2]
        print(f"{names[v]:30s} best {best:9.3f} ms  median {med:9.3f} ms  {n*n/best*1e3/1e12:6.3f} T-pairs/s  {n*n/best*1e3*20/157.3e12*100:5.1f}% of fp32 peak", flush=True)
