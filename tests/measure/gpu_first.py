"""First GPU pass: parity of every force-kernel variant against the CPU oracle at small N and an
interleaved A/B timing of all variants (one process) at larger N."""
import os, sys, time, json
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import nbody_amd as nbx
from oracle_lib import Oracle, accel_errors

o = Oracle()
names = nbx.variants()
print("devices:", nbx.device_count(), "variants:", len(names), flush=True)

def parity(n, dim, variant, splits=0):
    b = o.round_inputs_to_f32(o.generate(12345, n, dim))
    ref = o.brute_force_seq(b)
    with nbx.Context(n, dim) as c:
        c.set_tuning(splits, variant)
        c.upload(b)
        c.compute_accel()
        f = c.forces(o.G)
    rel, ab = accel_errors(f, ref, b[:, -1], o.G)
    return rel, ab

bad = 0
for dim in (3, 2):
    for n in (2, 257, 1024, 4096):
        for v, nm in enumerate(names):
            rel, ab = parity(n, dim, v)
            flag = "" if rel < 1e-5 else "   <-- FAIL"
            if flag: bad += 1
            print(f"parity D={dim} N={n:5d} {nm:32s} max|da|/|a| = {rel:.3e}  max-abs = {ab:.3e}{flag}", flush=True)
# split / accumulate paths
for splits in (1, 3, 8):
    rel, ab = parity(8192, 3, -1, splits)
    print(f"parity D=3 N=8192 default variant splits={splits}: {rel:.3e}", flush=True)
print("parity failures:", bad, flush=True)

# timing: interleaved rounds
sel = [int(x) for x in os.environ.get("NBX_VARIANTS", "").split(",") if x] or list(range(len(names)))
for n in (131072, 524288):
    b = o.generate(1, n, 3)
    with nbx.Context(n, 3) as c:
        c.upload(b)
        res = {v: [] for v in sel}
        rounds = 3 if n <= 131072 else 2
        for r in range(rounds + 1):
            for v in sel:
                c.set_tuning(1, v)
                c.compute_accel()
                ms, cnt = c.kernel_time()
                if r > 0: res[v].append(ms)
        print(f"--- N={n} (interactions {n*n:.3e}) ---", flush=True)
        for v in sorted(sel, key=lambda v: min(res[v])):
            best = min(res[v]); med = sorted(res[v])[len(res[v])//2]
            print(f"{names[v]:32s} best {best:9.3f} ms  median {med:9.3f} ms  -> {n*n/best*1e3/1e12:6.3f} T-interactions/s  ({n*n/best*1e3*20/157.3e12*100:5.1f}% of fp32 peak)", flush=True)
