"""Bodies with sub-threshold / close / duplicate pairs that STRADDLE shard boundaries (test input only).

For every boundary k*L (L = ceil(n/G)) of a G-shard split three pairs are planted, one member in shard k-1
and one in shard k:
  (a) indices (kL-1, kL)   : 0 < r^2 < 1e-10      -> the reference skips the pair (methods.cpp:24)
  (b) indices (kL-2, kL+1) : 1e-10 <= r^2 < 2.4e-7 -> counted; the kTiny bias of the unguarded kernel would be visible
  (c) indices (kL-3, kL+2) : exact duplicates       -> contribute exactly 0
alternating between the x- and the y-coordinate being the small one.  Built from the product's seeded
generator so that worker processes can rebuild the same array without the oracle."""
import numpy as np


def shard_len(n, n_shards):
    return -(-n // n_shards)


def make_bodies(nbx, n, dim, n_shards, seed=123):
    b = nbx.uniform_bodies(n, dim, seed)
    L = shard_len(n, n_shards)
    special = []
    for k in range(1, n_shards):
        e = k * L
        if e + 3 > n or e - 3 < 0:
            continue
        ax = k % 2 if dim >= 2 else 0          # which coordinate carries the small values
        far = [2.0e6 + 1.0e5 * k, 3.0e6 + 1.0e5 * k, 4.0e6][:dim]

        def at(small):
            p = list(far)
            p[ax] = small
            return p
        b[e - 1, :dim] = at(3.0);        b[e, :dim] = at(3.0 + 4.8e-7)          # (a) 2 ulp at 3.0: r^2 = 2.3e-13
        b[e - 2, :dim] = at(100.0);      b[e + 1, :dim] = at(100.0 + 1.6e-5)    # (b) 2 ulp at 100: r^2 = 2.3e-10
        b[e - 3, :dim] = at(7.0e6);      b[e + 2, :dim] = at(7.0e6)             # (c) duplicates outside the candidate region
        special += [e - 1, e, e - 2, e + 1, e - 3, e + 2]
    b[:, :dim] = b[:, :dim].astype(np.float32)
    b[:, -1] = b[:, -1].astype(np.float32)
    return np.ascontiguousarray(b), np.array(special, dtype=np.int64)
