"""N = 2^22 (BASELINE config 5's size) on one GPU: sampled-row parity + timing; and one shard of 8."""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import nbody_amd as nbx
from oracle_lib import Oracle, assert_force_parity
o = Oracle()
n = 1 << 22
b = o.round_inputs_to_f32(nbx.uniform_bodies(n, 3, 2))
rows = np.unique(np.random.default_rng(0).integers(0, n, 48))
ref = o.force_rows_omp_2(b, rows); S = o.force_magnitude_sums(b, rows)
with nbx.Context(n, 3) as c:
    c.upload(b); c.compute_accel(); f = c.forces(o.G); ms, _ = c.kernel_time()
    print("N=2^22 1 GPU:", c.effective_tuning(), f"{ms:.1f} ms  {n*n/ms*1e3:.3e} pairs/s", assert_force_parity(f[rows], ref, S, "N=2^22", n_sources=n), flush=True)
with nbx.Context(n, 3, n_shards=8, shard=3) as c:
    c.upload(b); c.compute_accel(nbx.SRC_LOCAL); c.compute_accel(nbx.SRC_REMOTE); fs = c.forces(o.G); ms, cnt = c.kernel_time()
    lo = 3 * c.shard_len
    sel = rows[(rows >= lo) & (rows < lo + c.count)]
    print("N=2^22 shard 3/8:", c.effective_tuning(), f"{ms*cnt:.1f} ms", np.abs(fs[sel - lo] - f[sel]).max() / np.abs(f[sel]).max() if sel.size else "no sampled rows in shard", flush=True)
