"""Force-kernel time against the number of source slices: is the automatic choice (nbx_api.hip auto_splits) near the best,
and does a workgroup count that fills the chip's 768 workgroup slots (256 CUs x 3) a whole number of times matter?
    python tools/probe_splits.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nbx

sizes = [int(a) for a in sys.argv[1:]] or [1 << 20, 65536, 100000, 262144]
for n in sizes:
    b = nbx.uniform_bodies(n, 3, 1)
    with nbx.Context(n, 3) as c:
        c.upload(b)
        pad = c.shard_pad
        tgt_blocks = pad // 2048
        tiles = pad // 256
        lo = max(1, (tiles + 255) // 256)
        cand = sorted(set([0] + [s for s in range(lo, min(256, tiles // 2) + 1)
                                 if s <= 40 or s % 4 == 0]))
        reps = 4 if n >= 1 << 19 else 20
        for S in cand:
            c.set_tuning(S, -1)
            c.compute_accel(); c.synchronize(); c.kernel_time()
            for _ in range(reps): c.compute_accel()
            ms, cnt = c.kernel_time()
            name, s_eff = c.effective_tuning()
            wgs = tgt_blocks * s_eff
            print(f"N={n} S={S:3d} (effective {s_eff:3d}) workgroups {wgs:6d} = {wgs / 768:6.2f} x 768, {tiles / s_eff:7.2f} tiles/slice: "
                  f"{ms:9.4f} ms  {n * n / ms * 1e3:.4e} pairs/s", flush=True)
