"""Force-kernel time against the number of source slices at small N (is the automatic choice of nbx_api.hip auto_splits near the best?)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, nbody_amd as nbx
for n in (65536, 131072, 262144):
    b = nbx.uniform_bodies(n, 3, 1)
    with nbx.Context(n, 3) as c:
        c.upload(b)
        for S in (0, 8, 16, 24, 32, 48, 64, 96):
            c.set_tuning(S, -1)
            for _ in range(3): c.compute_accel()
            c.synchronize(); c.kernel_time()
            for _ in range(20): c.compute_accel()
            ms, cnt = c.kernel_time()
            print(n, 'S', S, c.effective_tuning(), f'{ms:.4f} ms  {n*n/ms*1e3:.3e}', flush=True)
