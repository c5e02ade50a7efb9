"""Measurement / robustness run (GPU box, not part of the suite): random leaf structures through nbx_leaf_pair_forces against the
oracle -- leaf sizes from empty to several workgroups, lists with repeats, empty and consecutive leaves, streams of several tiles,
bodies in no leaf, structures of tiny leaves (packed several to a wave), launches of tens of thousands of leaves, small and large coordinate boxes (guarded and unguarded waves, and both in one launch), masses above and below the
bound of the unguarded loop, planted identical positions; both dimensions, all three laws.
    python tests/measure/leaf_fuzz.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import nbody_amd as nbx  # noqa: E402
from oracle_lib import Oracle, assert_force_parity  # noqa: E402


def one_case(rng, oracle, k):
    dim = int(rng.choice([2, 3]))
    regime = rng.choice(["tiny", "small", "medium", "big", "mixed", "many"], p=[0.19, 0.19, 0.19, 0.19, 0.19, 0.05])
    # tiny: a few bodies per leaf and many leaves -- the structures the planner PACKS, several leaves to a wave (mean leaf <= 8)
    # many: launches of more than 4,096 workgroups (dealt to the XCDs per duration class) whose lists have more than 200,000 entries
    #       (laid out on 8 host threads), packed or not
    n_leaves = int(rng.integers(1, 300)) if regime == "tiny" else int(rng.integers(9000, 30000)) if regime == "many" else int(rng.integers(1, 40))
    hi = {"tiny": int(rng.integers(2, 11)), "small": 20, "medium": 80, "big": 300, "mixed": 200, "many": int(rng.choice([8, 12, 40]))}[regime]
    sizes = rng.integers(0, hi, n_leaves)
    if regime == "tiny" and rng.random() < 0.3:
        sizes[int(rng.integers(0, n_leaves))] = int(rng.integers(11, 40))   # one larger leaf among them (its own workgroup, or the 9-16 class)
    if regime == "mixed":
        sizes[rng.random(n_leaves) < 0.6] //= 16
    sizes[rng.random(n_leaves) < 0.1] = 0
    n = int(sizes.sum()) + int(rng.integers(0, 9))
    if n == 0:
        return None
    b = oracle.generate(int(rng.integers(1, 1 << 30)), n, dim)
    box = rng.choice([1.0, 3.0e4, 1.0e7])
    b[:, :dim] *= box / 1.0e7
    if rng.random() < 0.3:
        b[:, :dim] = np.abs(b[:, :dim]) + 20000.0                      # every body outside the close set
    if rng.random() < 0.2:
        b[int(rng.integers(0, n)), -1] = 5.0e10                        # a mass above the unguarded loop's bound
    lo = np.concatenate([[0], np.cumsum(sizes)])
    lb = rng.permutation(n)[:lo[-1]]
    if lo[-1] >= 4 and rng.random() < 0.5:                              # identical positions, same leaf or not
        i, j = lb[0], lb[int(rng.integers(1, lo[-1]))]
        b[j, :dim] = b[i, :dim]
    b = oracle.round_inputs_to_f32(b)
    lists = []
    for t in range(n_leaves):
        kind = rng.random()
        length = int(rng.integers(0, 4)) if kind < 0.2 else int(rng.integers(1, 30)) if kind < 0.8 or regime == "many" else int(rng.integers(60, 200))
        l = rng.integers(0, n_leaves, length)
        if length >= 4 and rng.random() < 0.6:                          # a run of consecutive leaves (merged by the planner)
            s = int(rng.integers(0, n_leaves))
            run = np.arange(s, min(n_leaves, s + int(rng.integers(2, 6))))
            l[:run.size] = run[:length]
        if length and rng.random() < 0.5:
            l[int(rng.integers(0, length))] = t                        # the leaf itself
        lists.append(l)
    so = np.concatenate([[0], np.cumsum([len(l) for l in lists])])
    ss = np.concatenate(lists) if so[-1] else np.zeros(0, dtype=np.int64)
    leaves = (lo, lb, so, ss)
    law = int(rng.integers(0, 3))
    f = nbx.leaf_pair_forces_hip(b, *leaves, law=law, G=oracle.G)
    ref = oracle.leaf_pair_forces(b, leaves, law)
    S = oracle.leaf_pair_magnitude_sums(b, leaves, law)
    assert np.isfinite(f).all(), (k, "non-finite")
    live = S > 0
    assert not f[~live].any(), (k, "a body without any counted pair must get exactly zero")
    if live.any():
        assert_force_parity(f[live], ref[live], S[live], f"case {k}: D={dim} {regime} box={box:g} law={law} n={n} leaves={n_leaves} list entries={so[-1]}")
    return int((sizes[:, None] * 0).size), int(sum(sizes[t] * sizes[lists[t]].sum() for t in range(n_leaves)))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    rng = np.random.default_rng(seed)
    oracle = Oracle()
    done, pairs = 0, 0
    for k in range(cases):
        r = one_case(rng, oracle, k)
        if r:
            done += 1
            pairs += r[1]
        if (k + 1) % 50 == 0:
            print(f"{k + 1} cases, {pairs:.3e} pair terms so far: all within tolerance", flush=True)
    print(f"leaf fuzz: {done} structures (seed {seed}), {pairs:.3e} pair terms, every body within the stated tolerance of the oracle")


if __name__ == "__main__":
    main()
