"""CPU: SURVEY 8(f-4), the leaf-pair direct-sum oracle (oracle_leaf_pair_forces) and its host-side plumbing.
Law 1 (tree leaf) is pinned twice: to the reference's own Barnes-Hut octree walked with theta = 0, and to the reference's own
BVH -- the leaves of BVH<D>(bodies, 16) read through its public root, and per body the sum of BVH::calculate_force(body, leaf)
over all of them (bvh.cpp:143-176) -- live where oracle/_ref exists, and through committed golden outputs everywhere;
law 0 is the brute-force oracle again; law 2 (FMM P2P) is a restatement only (PARITY UNPINNED: executing FMM_Parlay is
undefined behaviour -- its constructor leaves the tree pointing into a destroyed vector, fmm_parlay.cpp:16-22 -- so the
reference cannot produce a vector to pin it to; see the oracle)."""
import numpy as np
import pytest

from conftest import golden


def _one_leaf(n):
    return (np.array([0, n]), np.arange(n), np.array([0, 1]), np.array([0]))


@pytest.mark.parametrize("dim", (2, 3))
def test_tree_leaf_law_matches_reference_octree_golden(oracle, dim):
    g = golden(f"octree_direct_D{dim}_N512.npz")
    b = np.ascontiguousarray(g["bodies_f32"])
    f = oracle.leaf_pair_forces(b, _one_leaf(512), 1)
    ref = g["forces_octree_theta0"]
    assert np.abs(f - ref).max() <= 1e-12 * np.abs(ref).max()          # the tree sums in its own order
    # the pair at r^2 = 3.6e-10 separates the laws: skipped by the leaf law (< 1e-9), counted by the brute-force law
    assert np.abs(f[10]).max() < 1e-20 < 1e6 < np.abs(g["forces_brute_seq"][10]).max()
    assert np.array_equal(oracle.leaf_pair_forces(b, _one_leaf(512), 0), oracle.brute_force_omp_2(b))
    # FMM P2P law: attractive brute force except below 1e-10 (smoothing) -- this pair is above it, so f2 = -f0 there
    f0, f2 = oracle.leaf_pair_forces(b, _one_leaf(512), 0), oracle.leaf_pair_forces(b, _one_leaf(512), 2)
    assert np.allclose(f2, -f0, rtol=1e-13, atol=0)


@pytest.mark.parametrize("dim", (2, 3))
def test_tree_leaf_law_matches_live_reference_octree(oracle, reference, dim):
    b = oracle.round_inputs_to_f32(oracle.generate(31 + dim, 900, dim))
    f = oracle.leaf_pair_forces(b, _one_leaf(900), 1)
    ref = reference.octree_direct_forces(b)
    assert np.abs(f - ref).max() <= 1e-12 * np.abs(ref).max()


def every_leaf_lists(n_leaves):
    """Every leaf on every target leaf's list, own leaf included: list_offsets, list_sources."""
    return (np.arange(n_leaves + 1, dtype=np.uint32) * n_leaves, np.tile(np.arange(n_leaves, dtype=np.uint32), n_leaves))


@pytest.mark.parametrize("dim,n", ((2, 4096), (3, 4096), (3, 3000)))
def test_tree_leaf_law_on_the_reference_bvh_golden(oracle, dim, n):
    """Leaves of a reference-BUILT tree (multi-body, ragged at N = 3000) and the reference BVH's own leaf sums."""
    g = golden(f"bvh_leaves_D{dim}_N{n}.npz")
    b = np.ascontiguousarray(g["bodies_f32"])
    lo, lb = g["leaf_offsets"], g["leaf_bodies"]
    sizes = np.diff(lo)
    assert lo[-1] == n and sizes.max() <= 16 and sizes.min() >= 8 and np.array_equal(np.sort(lb), np.arange(n))
    ref = g["forces_bvh_all_leaves"]
    f = oracle.leaf_pair_forces(b, (lo, lb) + every_leaf_lists(lo.size - 1), 1)
    assert np.abs(f - ref).max() <= 1e-12 * np.abs(ref).max()          # per-leaf partial sums vs one running sum
    # with every leaf on every list it is the all-pairs sum under the leaf law: equal to the octree-pinned single-leaf form
    one = oracle.leaf_pair_forces(b, _one_leaf(n), 1)
    assert np.abs(f - one).max() <= 1e-12 * np.abs(ref).max()
    # the planted pairs: r^2 = 3.6e-10 and an exact duplicate are skipped by the leaf law (bvh.cpp:151-164)
    d2 = ((b[20, :dim] - b[21, :dim]) ** 2).sum()
    assert 1e-10 < d2 < 1e-9 and np.array_equal(b[30, :dim], b[31, :dim]) and np.isfinite(ref).all()


@pytest.mark.parametrize("dim", (2, 3))
def test_tree_leaf_law_on_the_live_reference_bvh(oracle, reference, dim):
    n = 2500
    b = oracle.round_inputs_to_f32(oracle.generate(41 + dim, n, dim))
    lo, lb = reference.bvh_leaves(b, 16)
    ref = reference.bvh_leaf_forces(b, 16)
    f = oracle.leaf_pair_forces(b, (lo, lb) + every_leaf_lists(lo.size - 1), 1)
    assert np.abs(f - ref).max() <= 1e-12 * np.abs(ref).max()
    lo8, lb8 = reference.bvh_leaves(b, 5)                               # another leaf capacity: other tree, same sums
    f8 = oracle.leaf_pair_forces(b, (lo8, lb8) + every_leaf_lists(lo8.size - 1), 1)
    assert np.diff(lo8).max() <= 5 and np.abs(f8 - reference.bvh_leaf_forces(b, 5)).max() <= 1e-12 * np.abs(ref).max()


def test_fmm_p2p_smoothing_known_answers(oracle):
    """fmm_parlay.cpp:992-1020 by hand: identical positions skipped; r^2 < 1e-10 smoothed, not skipped."""
    G = oracle.G
    def body(p, m):
        return list(p) + [0, 0, 0] + [m]
    d = 4.0e-6                                                      # r^2 = 1.6e-11 < 1e-10
    b = np.array([body((1, 1, 1), 2.0), body((1 + d, 1, 1), 3.0), body((1, 1, 1), 5.0)], dtype=np.float64)
    f = oracle.leaf_pair_forces(b, _one_leaf(3), 2)
    dd = b[1, 0] - b[0, 0]
    r2s = dd * dd + 1e-5 * 1e-5
    want01 = G * 2.0 * 3.0 / (r2s * np.sqrt(r2s))                    # body 0 pulled towards body 1 (+x); body 2 coincides: skipped
    assert f[0, 0] == pytest.approx(want01, rel=1e-14) and f[0, 1] == 0 and f[0, 2] == 0
    assert f[1, 0] == pytest.approx(-(G * 3.0 * 2.0 + G * 3.0 * 5.0) / (r2s * np.sqrt(r2s)), rel=1e-14)
    # the brute-force and tree-leaf laws skip all three pairs
    assert not oracle.leaf_pair_forces(b, _one_leaf(3), 0).any() and not oracle.leaf_pair_forces(b, _one_leaf(3), 1).any()


def test_leaf_lists_partition_the_all_pairs_sum(oracle):
    """Host logic of leaves.py: grid leaves partition the bodies; with every leaf on every list the leaf-pair sum is
    the all-pairs sum; with the 3^D neighbour lists it is the sum over bodies of adjacent cells only."""
    import nbody_amd as nbx
    n, dim = 3000, 3
    b = oracle.round_inputs_to_f32(oracle.generate(8, n, dim))
    lo, lb, so, ss = nbx.leaves.uniform_grid_leaves(b, dim, 2)
    assert lo[0] == 0 and lo[-1] == n and np.array_equal(np.sort(lb), np.arange(n)) and (np.diff(lo) > 0).all()
    assert so.size == lo.size and (ss[so[:-1]] == np.arange(lo.size - 1)).all()           # own leaf first
    A = nbx.leaves.all_pairs_leaves(n, 128)
    assert np.array_equal(oracle.leaf_pair_forces(b, A, 0), oracle.brute_force_omp_2(b))
    near = oracle.leaf_pair_forces(b, (lo, lb, so, ss), 1)
    # brute-force check of the neighbour sum for a few bodies
    cell = np.empty(n, dtype=np.int64)
    for l in range(lo.size - 1):
        cell[lb[lo[l]:lo[l + 1]]] = l
    for i in (0, 17, n - 1):
        srcs = np.concatenate([lb[lo[s]:lo[s + 1]] for s in ss[so[cell[i]]:so[cell[i] + 1]]])
        dvec = b[srcs, :3] - b[i, :3]
        r2 = (dvec ** 2).sum(1)
        ok = r2 >= 1e-9
        want = (oracle.G * b[i, -1] * b[srcs[ok], -1] / r2[ok] ** 2)[:, None] * dvec[ok]
        assert np.allclose(near[i], want.sum(0), rtol=1e-10, atol=0)
    e = nbx.leaves.uniform_grid_leaves(np.zeros((0, 7)), 3, 2)
    assert e[0].tolist() == [0] and e[1].size == 0
