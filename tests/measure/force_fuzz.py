"""Measurement / robustness run (GPU box, not part of the suite): random systems through the brute-force path against the oracle's
sequential reference -- N from 1 to 7,000, both dimensions, coordinate boxes from 1 to 1e7 (inside and outside the close set, and
across its boundary), planted identical / sub-threshold / just-above-threshold pairs, one GPU or 2-5 shards (context passes ALL and
LOCAL + REMOTE, and the node layer's virtual ranks), every kernel variant of the library, mixed mode on and off.
    python tests/measure/force_fuzz.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import nbody_amd as nbx  # noqa: E402
from oracle_lib import Oracle, assert_force_parity  # noqa: E402


def one_case(rng, oracle, k, variants):
    dim = int(rng.choice([2, 3]))
    n = int(rng.choice([1, 2, 3, 17, 255, 256, 257])) if rng.random() < 0.15 else int(rng.integers(4, 7000))
    b = oracle.generate(int(rng.integers(1, 1 << 30)), n, dim)
    box = float(rng.choice([1.0, 100.0, 3.0e4, 1.0e7]))
    b[:, :dim] *= box / 1.0e7
    if rng.random() < 0.25:
        b[:, :dim] = np.abs(b[:, :dim]) + 20000.0
    if n >= 8 and rng.random() < 0.6:                                   # planted pairs, wherever they fall among the shards
        for (i, j, gap) in ((0, n - 1, 0.0), (1, n // 2, 4.0e-6), (2, n // 3 + 3, 2.0e-5), (3, n - 2, 4.0e-4)):
            b[j, :dim] = b[i, :dim]
            b[j, 0] += gap
    b = oracle.round_inputs_to_f32(b)
    ref = oracle.brute_force_seq(b)
    S = oracle.force_magnitude_sums(b)
    live = S > 0
    how = rng.choice(["one_shot", "context", "shards", "node"])
    what = f"case {k}: D={dim} n={n} box={box:g} {how}"

    def check(f, lo=0, hi=n, tag=""):
        assert np.isfinite(f).all(), (what, tag)
        m = live[lo:hi]
        assert not f[~m].any(), (what, tag, "a body without any counted pair must get exactly zero")
        if m.any():
            assert_force_parity(f[m], ref[lo:hi][m], S[lo:hi][m], what + tag)

    if how == "one_shot":
        check(nbx.brute_force_hip_n_body(b, oracle.G))
    elif how == "context":
        with nbx.Context(n, dim) as c:
            c.upload(b)
            v = int(rng.integers(-1, len(variants)))
            c.set_tuning(0, v)
            if rng.random() < 0.4:
                c.set_refine(1.0e-5)
            c.compute_accel()
            check(c.forces(oracle.G), tag=f" variant {variants[v] if v >= 0 else 'default'}")
    elif how == "shards":
        g = int(rng.integers(2, 6))
        for r in range(g):
            with nbx.Context(n, dim, n_shards=g, shard=r) as c:
                if c.count == 0:
                    continue
                c.upload(b)
                if rng.random() < 0.3:
                    c.set_refine(1.0e-5)
                lo, hi = r * c.shard_len, r * c.shard_len + c.count
                if rng.random() < 0.5:
                    c.compute_accel(nbx.SRC_ALL)
                else:
                    c.compute_accel(nbx.SRC_LOCAL)
                    c.compute_accel(nbx.SRC_REMOTE)
                check(c.forces(oracle.G), lo, hi, f" shard {r}/{g}")
    else:
        g = int(rng.integers(2, 5))
        with nbx.Node(n, dim, [0] * g) as node:
            node.upload(b)
            check(node.forces(oracle.G), tag=f" node x{g}")
    return n * n


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    rng = np.random.default_rng(seed)
    oracle = Oracle()
    variants = nbx.variants()
    pairs = 0
    for k in range(cases):
        pairs += one_case(rng, oracle, k, variants)
        if (k + 1) % 50 == 0:
            print(f"{k + 1} cases, {pairs:.3e} ordered pairs so far: all within tolerance", flush=True)
    print(f"force fuzz: {cases} systems (seed {seed}), {pairs:.3e} ordered pairs, every body within the stated tolerance of the sequential reference")


if __name__ == "__main__":
    main()
