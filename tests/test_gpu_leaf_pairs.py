"""GPU: SURVEY 8(f-4) -- nbx_leaf_pair_forces (csrc/leaf_pair_kernel.hip, through the C ABI) against the oracle's
restatement of the reference's leaf direct sums (fmm_parlay.cpp:992-1020, bvh.cpp:150-176, octree.cpp:105-125) and
against the committed outputs of the reference's own octree walked with theta = 0.  Same stated fp32 tolerance as the
brute-force path (oracle_lib.TOL_*)."""
import numpy as np
import pytest

from conftest import golden
from oracle_lib import assert_force_parity

pytestmark = pytest.mark.gpu
LAWS = ((0, "brute"), (1, "tree_leaf"), (2, "fmm_p2p"))


import contextlib
import os


@contextlib.contextmanager
def planner(which):
    """NBODY_HIP_LEAF_PLANNER=host|device: which planner lays the structure out (csrc/leaf_plan.h on the host, csrc/leaf_plan_device.h
    on the device); the library reads it on every call.  Unset, the size of the structure decides."""
    before = os.environ.get("NBODY_HIP_LEAF_PLANNER")
    os.environ["NBODY_HIP_LEAF_PLANNER"] = which
    try:
        yield
    finally:
        if before is None:
            del os.environ["NBODY_HIP_LEAF_PLANNER"]
        else:
            os.environ["NBODY_HIP_LEAF_PLANNER"] = before


def _all_paths(nbx, b, leaves, law, G, what=""):
    """The three ways into the pair kernel give the same bits: the one-shot call (nbx_leaf_pair_forces: validates, lays out and
    uploads per call), the resident PLAN fed host bodies (nbx_leaf_plan_forces), and the plan fed bodies that live in a context
    (nbx_leaf_plan_forces_ctx) -- twice, the second time with the sums left on the device (nbx_leaf_plan_get_forces).  All of it
    through BOTH planners: the structure laid out on the host and on the device must give the same bits too."""
    n, dim = b.shape[0], (b.shape[1] - 1) // 2
    with planner("host"):
        f = nbx.leaf_pair_forces_hip(b, *leaves, law=law, G=G)
    for which in ("host", "device"):
        with planner(which):
            if which == "device":
                assert np.array_equal(nbx.leaf_pair_forces_hip(b, *leaves, law=law, G=G), f), f"{what}: one-shot call, device planner, differs from the host planner's"
            with nbx.LeafPlan(n, dim, *leaves) as plan:
                assert np.array_equal(plan.forces(b, law, G), f), f"{what}: plan (host bodies, {which} planner) differs from the one-shot call"
                if n:
                    with nbx.Context(n, dim) as c:
                        c.upload(b)
                        assert np.array_equal(plan.forces_ctx(c, law, G), f), f"{what}: plan (resident bodies, {which} planner) differs from the one-shot call"
                        plan.forces_ctx(c, law, G, fetch=False)
                        assert np.array_equal(plan.get_forces(), f), f"{what}: second evaluation of the unchanged structure ({which} planner)"
    return f


def _check(nbx, oracle, b, leaves, law, what):
    f = _all_paths(nbx, b, leaves, law, oracle.G, what)
    ref = oracle.leaf_pair_forces(b, leaves, law)
    S = oracle.leaf_pair_magnitude_sums(b, leaves, law)
    assert f.shape == ref.shape and np.isfinite(f).all()
    live = S > 0
    assert not f[~live].any(), f"{what}: bodies without any counted pair must get exactly zero"
    if live.any():
        assert_force_parity(f[live], ref[live], S[live], what)
    return f, ref


@pytest.mark.parametrize("dim", (3, 2))
@pytest.mark.parametrize("law,name", LAWS)
def test_grid_leaves_against_oracle(nbx, oracle, dim, law, name):
    n = 20000
    b = oracle.round_inputs_to_f32(oracle.generate(50 + dim, n, dim))
    leaves = nbx.leaves.uniform_grid_leaves(b, dim, 3 if dim == 3 else 4)
    f, ref = _check(nbx, oracle, b, leaves, law, f"grid leaves, law {name}, D={dim}")
    if law != 0:                                # the tree codes' laws are the brute-force law with the sign flipped
        f0 = nbx.leaf_pair_forces_hip(b, *leaves, law=nbx.LAW_BRUTE, G=oracle.G)
        assert np.abs(f + f0).max() <= 1e-4 * np.abs(f0).max()


@pytest.mark.parametrize("dim", (3, 2))
def test_reference_octree_golden(nbx, oracle, dim):
    """Committed output of the reference's octree (theta = 0 walk): every pair through octree.cpp:105-125."""
    g = golden(f"octree_direct_D{dim}_N512.npz")
    b = np.ascontiguousarray(g["bodies_f32"])
    one = (np.array([0, 512]), np.arange(512), np.array([0, 1]), np.array([0]))
    f = _all_paths(nbx, b, one, nbx.LAW_TREE_LEAF, float(g["G"]), "octree golden")
    assert_force_parity(f, g["forces_octree_theta0"], oracle.leaf_pair_magnitude_sums(b, one, 1), f"octree golden D={dim}")
    assert np.allclose(f[10], g["forces_octree_theta0"][10], rtol=1e-4, atol=0)      # partner at r^2 = 3.6e-10: skipped
    fb = _all_paths(nbx, b, one, nbx.LAW_BRUTE, float(g["G"]), "octree golden, brute law")
    assert_force_parity(fb, g["forces_brute_seq"], oracle.force_magnitude_sums(b), f"brute law over one leaf D={dim}")
    assert np.allclose(fb[10], g["forces_brute_seq"][10], rtol=1e-4, atol=0)         # ... and counted by the brute-force law


@pytest.mark.parametrize("dim,n", ((2, 4096), (3, 4096), (3, 3000)))
def test_reference_bvh_golden(nbx, oracle, dim, n):
    """Leaves of the REFERENCE's own BVH<D>(bodies, 16) (read through its public root, bvh.h:91-103) as the CSR arrays of the
    call, every leaf on every list, the tree codes' leaf law: the device reproduces the reference BVH's own near-field sums
    (per body, BVH::calculate_force(body, leaf) over all leaves, bvh.cpp:143-176; committed by tests/golden/make_golden.py)
    within the stated fp32 tolerance, and the oracle restatement matches them to 1e-12.
    NBX_LAW_FMM_P2P stays PARITY-UNPINNED: FMM_Parlay<D>'s constructor leaves its tree pointing into a destroyed local
    vector (fmm_parlay.cpp:16-22 with fmm.cpp:389-395), so the reference's p2p_phase cannot be executed to produce a vector;
    that law is checked against its restatement (fmm_parlay.cpp:992-1020) and hand-computed known answers only."""
    g = golden(f"bvh_leaves_D{dim}_N{n}.npz")
    b = np.ascontiguousarray(g["bodies_f32"])
    lo, lb = g["leaf_offsets"], g["leaf_bodies"]
    nl = lo.size - 1
    leaves = (lo, lb, np.arange(nl + 1, dtype=np.uint32) * nl, np.tile(np.arange(nl, dtype=np.uint32), nl))
    ref = g["forces_bvh_all_leaves"]
    f = _all_paths(nbx, b, leaves, nbx.LAW_TREE_LEAF, float(g["G"]), "reference BVH leaves")
    S = oracle.leaf_pair_magnitude_sums(b, leaves, 1)
    e = assert_force_parity(f, ref, S, f"reference BVH leaves D={dim} N={n}")
    orc = oracle.leaf_pair_forces(b, leaves, 1)
    assert np.abs(orc - ref).max() <= 1e-12 * np.abs(ref).max()
    # the planted sub-threshold pair (r^2 = 3.6e-10) and the exact duplicate are skipped, as the reference skips them
    for i in (20, 21, 30, 31):
        assert np.allclose(f[i], ref[i], rtol=1e-4, atol=0), i
    # a near-field list (each leaf + the leaves whose boxes come within one leaf diagonal): against the pinned restatement
    cen = np.array([b[lb[lo[l]:lo[l + 1]], :dim].mean(axis=0) for l in range(nl)])
    diag = np.array([np.linalg.norm(np.ptp(b[lb[lo[l]:lo[l + 1]], :dim], axis=0)) for l in range(nl)])
    near = [np.r_[l, np.setdiff1d(np.nonzero(np.linalg.norm(cen - cen[l], axis=1) < 1.5 * (diag[l] + diag.mean()))[0], [l])] for l in range(nl)]
    so = np.r_[0, np.cumsum([len(x) for x in near])].astype(np.uint32)
    _check(nbx, oracle, b, (lo, lb, so, np.concatenate(near).astype(np.uint32)), 1, f"near-field lists on reference BVH leaves D={dim}")
    print(f"\nreference BVH leaves D={dim} N={n}: {nl} leaves of {np.diff(lo).min()}-{np.diff(lo).max()} bodies, errors {e}")


def test_all_pairs_lists_equal_the_brute_force_path(nbx, oracle):
    """Every leaf on every list: the leaf kernel under the brute-force law must agree with the all-pairs kernel."""
    n, dim = 6000, 3
    b = oracle.round_inputs_to_f32(oracle.generate(61, n, dim))
    A = nbx.leaves.all_pairs_leaves(n, 100)                        # ragged last leaf, leaves below one tile
    f, ref = _check(nbx, oracle, b, A, 0, "all-pairs lists")
    assert_force_parity(f, oracle.brute_force_seq(b), oracle.force_magnitude_sums(b), "leaf kernel vs sequential reference")
    assert_force_parity(nbx.brute_force_hip_n_body(b, oracle.G), ref, oracle.force_magnitude_sums(b), "brute-force kernel vs leaf oracle")


def test_ragged_structure_and_close_pairs(nbx, oracle):
    """Empty leaves, a leaf larger than one 128-body tile, leaves off every list, an empty list, repeated list entries,
    bodies in no leaf; identical positions, sub-threshold and just-above-threshold pairs for every law."""
    dim, n = 3, 1000
    b = oracle.generate(70, n, dim)
    b[:, :3] = b[:, :3] / 1.0e5                                     # box of 100: close pairs are representable in fp32
    b[1, :3] = b[0, :3]                                            # identical positions
    b[3, :3] = b[2, :3]; b[3, 0] += 4.0e-6                          # r^2 = 1.6e-11: < 1e-10 (brute: skip, tree: skip, fmm: smoothed)
    b[5, :3] = b[4, :3]; b[5, 1] += 2.0e-5                          # r^2 = 4e-10: brute counts, tree skips, fmm counts unsmoothed
    b[7, :3] = b[6, :3]; b[7, 2] += 4.0e-5                          # r^2 = 1.6e-9: everyone counts
    b = oracle.round_inputs_to_f32(b)
    sizes = [300, 0, 1, 129, 0, 64, 200, 256]                       # 950 of 1000 bodies; 50 in no leaf
    lo = np.concatenate([[0], np.cumsum(sizes)])
    lb = np.random.default_rng(1).permutation(n)[:lo[-1]]
    lb[:8] = np.arange(8)                                           # the planted pairs share leaf 0 ...
    rest = np.setdiff1d(np.arange(8, n), [])
    lb[8:] = np.random.default_rng(2).permutation(rest)[:lo[-1] - 8]
    lists = [[0, 3, 0], [2], [], [0, 1, 2, 3, 4, 5, 6, 7], [0], [5], [7, 6], [3]]   # repeated entry, empty list, empty sources
    so = np.concatenate([[0], np.cumsum([len(l) for l in lists])])
    ss = np.array([s for l in lists for s in l])
    leaves = (lo, lb, so, ss)
    for law, name in LAWS:
        f, ref = _check(nbx, oracle, b, leaves, law, f"ragged, law {name}")
        out = np.setdiff1d(np.arange(n), lb)
        assert not f[out].any()
        for i in range(8):                                          # the planted bodies, dominated by their partner where counted
            assert np.linalg.norm(f[i] - ref[i]) <= 2e-5 * max(np.linalg.norm(ref[i]), 1e-300), (name, i, f[i], ref[i])
    f0 = oracle.leaf_pair_forces(b, leaves, 0); f1 = oracle.leaf_pair_forces(b, leaves, 1); f2 = oracle.leaf_pair_forces(b, leaves, 2)
    assert np.abs(f2[2]).max() > 1e3 * np.abs(f0[2]).max()          # only the FMM law keeps the 4e-6 pair (smoothed)
    assert np.abs(f0[4]).max() > 1e3 * np.abs(f1[4]).max()          # the tree-leaf law drops the 2e-5 pair


@pytest.mark.parametrize("dim", (3, 2))
@pytest.mark.parametrize("big", (False, True, "small"))
def test_every_block_shape_and_long_lists(nbx, oracle, dim, big):
    """Target leaves of every size from 1 to 70, or 1 to 258 in steps of 3 (big: up to three workgroups per leaf), or 1 to 30 with a
    mean of 12 (small: the library launches one wave per workgroup, the leaf uncut), so that every lanes-per-target count, every
    cut of a leaf into the two waves' pieces and every partly filled last tile occurs; source lists of 1 to 300 entries in random
    order, with repeats and empty leaves in them (more copy runs than the kernel holds in LDS at a time: chunked; streams of
    several tiles), odd and even leaf sizes (padded source pairs); all three laws."""
    sizes = list(range(1, 71)) if big is False else list(range(1, 260, 3)) if big is True else list(range(1, 25)) * 3 + [30, 27]
    sizes += [0, 0]                                                  # two empty leaves, also as sources
    rng = np.random.default_rng(90 + dim + 2 * (big is True) + 5 * (big == "small"))
    order = rng.permutation(len(sizes))
    sizes = [sizes[i] for i in order]
    n = sum(sizes) + 17                                              # 17 bodies in no leaf
    b = oracle.generate(91 + dim, n, dim)
    b[:, :dim] /= 1.0e3                                              # a box of 1e4: denser, larger pair terms
    b = oracle.round_inputs_to_f32(b)
    lo = np.concatenate([[0], np.cumsum(sizes)])
    lb = rng.permutation(n)[:lo[-1]]
    n_leaves = len(sizes)
    lists = []
    for t in range(n_leaves):
        k = [1, 2, 3, 27, 64, 65, 129, 300][t % 8] if sizes[t] else 5
        lists.append(rng.integers(0, n_leaves, k))
        if t % 3 == 0:
            lists[-1][0] = t                                         # own leaf on the list: every body meets itself
    so = np.concatenate([[0], np.cumsum([len(l) for l in lists])])
    ss = np.concatenate(lists)
    leaves = (lo, lb, so, ss)
    for law, name in LAWS:
        _check(nbx, oracle, b, leaves, law, f"every block shape, law {name}, D={dim}, big={big}")


@pytest.mark.parametrize("dim", (3, 2))
def test_unguarded_and_guarded_waves(nbx, oracle, dim):
    """The pair loop without any compare runs for a wave whose targets all lie outside the close set (every |coordinate| >= 2^14,
    csrc/nbx_internal.h) while no mass exceeds 1e10; every other wave takes the guarded loop.  Both must give the law's result:
    (a) the reference generator's box (coordinates up to 1e7: nearly every wave unguarded) with identical positions planted --
    inside one leaf and across two leaves -- and every body meeting itself in its own leaf;
    (b) the same bodies with one mass above the bound: every wave guarded, same forces up to that body's own contribution;
    (c) a grid cell that straddles the close set's boundary: guarded and unguarded waves side by side."""
    n = 6000
    b = oracle.generate(120 + dim, n, dim)
    b[:, :dim] = np.abs(b[:, :dim]) + 20000.0                         # every body outside the close set
    leaves = nbx.leaves.uniform_grid_leaves(oracle.round_inputs_to_f32(b), dim, 2 if dim == 3 else 3)
    lo, lb, so, ss = leaves
    first, second = lb[lo[0]], lb[lo[0] + 1]                         # two bodies of leaf 0 ...
    other = lb[lo[1]]                                                # ... and one of its neighbour
    b[second, :dim] = b[first, :dim]
    b[other, :dim] = b[first, :dim]                                  # (stays in leaf 1's list: a pair across leaves at r^2 = 0)
    b = oracle.round_inputs_to_f32(b)
    for law, name in LAWS:
        f, _ = _check(nbx, oracle, b, leaves, law, f"unguarded waves, law {name}, D={dim}")
        heavy = b.copy()
        heavy[lb[lo[2]], -1] = 3.0e10                                # above kFastMaxMass: the guarded loop everywhere
        fh, _ = _check(nbx, oracle, heavy, leaves, law, f"guarded by a heavy mass, law {name}, D={dim}")
        blind = [t for t in range(lo.size - 1) if 2 not in ss[so[t]:so[t + 1]]]     # leaves whose lists do not name leaf 2
        far = np.concatenate([lb[lo[t]:lo[t + 1]] for t in blind])
        S = oracle.leaf_pair_magnitude_sums(b, leaves, law)[far]
        gap = np.linalg.norm(f[far] - fh[far], axis=1)
        assert far.size and (gap <= 1.0e-6 * S).all(), f"the two loops must agree to fp32 rounding where the heavy body is not a source: {(gap / S).max():.2e}"
    c = oracle.generate(130 + dim, n, dim)
    c[:, :dim] = c[:, :dim] / 1.0e7 * 60000.0                        # a box of 6e4: about half of the bodies inside the close set
    c = oracle.round_inputs_to_f32(c)
    cl = nbx.leaves.uniform_grid_leaves(c, dim, 2 if dim == 3 else 3)
    for law, name in LAWS:
        _check(nbx, oracle, c, cl, law, f"guarded and unguarded waves mixed, law {name}, D={dim}")


@pytest.mark.parametrize("dim", (3, 2))
@pytest.mark.parametrize("box", (1.0e7, 6.0e4, 1.0))
def test_packed_small_leaves(nbx, oracle, dim, box):
    """Leaves of a few bodies are PACKED several to a wave (csrc/leaf_plan.h PackBlock, leaf_pack_kernel): grid cells of ~5 bodies
    (0 to ~15, Poisson) with 3^D-cell lists, and the reference BVH's own shape -- median-split leaves of at most 16 and at most 8
    bodies (bvh.cpp:34-73, methods.h:57) with box-distance near-field lists.  Three coordinate regimes -- the reference generator's
    box (nearly every wave takes the pair loop without compares), a box that straddles the close set's boundary (guarded and
    unguarded waves side by side; within a wave, leaves on both sides), a unit box (every wave guarded) -- with identical positions
    and sub-threshold pairs planted inside a leaf and across two leaves; all three laws; one-shot call, plan and resident plan."""
    n = 20000
    b = oracle.generate(160 + dim, n, dim)
    b[:, :dim] = np.abs(b[:, :dim]) / 1.0e7 * box + (20000.0 if box == 1.0e7 else 0.0)
    grid = nbx.leaves.uniform_grid_leaves(oracle.round_inputs_to_f32(b), dim, 4 if dim == 3 else 6)
    lo, lb, so, ss = grid
    sizes = np.diff(lo)
    big = int(np.argmax(sizes))
    assert 3 <= sizes[big] <= 16 and np.diff(so).max() <= 3 ** dim
    i0, i1, i2 = lb[lo[big]], lb[lo[big] + 1], lb[lo[big] + 2]
    nb = int(ss[so[big] + 1])                                       # a neighbour leaf of it
    j0 = lb[lo[nb]]
    b[i1, :dim] = b[i0, :dim]                                       # identical positions inside a leaf
    b[j0, :dim] = b[i0, :dim]                                       # ... and across two leaves (stays on the neighbour's list)
    if box <= 6.0e4:
        b[i2, :dim] = b[i0, :dim]; b[i2, 0] += 2.0e-5 * max(box, 1.0) / 1.0     # r^2 = 4e-10 (unit box): brute counts, tree skips, fmm counts
    b = oracle.round_inputs_to_f32(b)
    for law, name in LAWS:
        _check(nbx, oracle, b, grid, law, f"packed grid leaves, law {name}, D={dim}, box {box:g}")
    for max_leaf in (16, 8):
        tree = nbx.leaves.median_split_leaves(b, dim, max_leaf, reach=0.5)
        tsz = np.diff(tree[0])
        assert tsz.max() <= max_leaf and tsz.min() >= max_leaf // 2
        _check(nbx, oracle, b, tree, nbx.LAW_TREE_LEAF, f"median-split leaves of <= {max_leaf}, D={dim}, box {box:g}")
    with nbx.LeafPlan(n, dim, *grid) as plan:
        slots, runs, groups, waves = plan.info()
        assert groups < 0.5 * (sizes > 0).sum(), "small leaves must share waves"
    heavy = b.copy()
    heavy[int(lb[lo[0]]), -1] = 3.0e10                              # above kFastMaxMass: every wave takes the guarded loop
    _check(nbx, oracle, heavy, grid, nbx.LAW_FMM_P2P, f"packed grid leaves, guarded by a heavy mass, D={dim}, box {box:g}")


@pytest.mark.parametrize("shape", ("packed grid cells", "median-split 16", "median-split 8"))
def test_large_launches_xcd_order_and_threaded_layout(nbx, oracle, shape):
    """Launches of more than 4,096 workgroups (csrc/leaf_plan.h order_launch deals the blocks of a duration class to the XCDs) and
    lists of more than 200,000 entries (the layout's copy runs, padded slots and packed waves are built on 8 host threads): every
    body of the structure against the oracle, all three ways in.  N = 2^18: 32^3 grid cells of ~8 bodies (packed, > 4,096 waves,
    880,000 list entries), median-split leaves of 16 bodies (16,384 one-leaf workgroups) and of 8 bodies (32,768 leaves, packed)."""
    n = 1 << 18
    b = oracle.round_inputs_to_f32(oracle.generate(177, n, 3))
    if shape == "packed grid cells":
        leaves = nbx.leaves.uniform_grid_leaves(b, 3, 5)
    else:
        leaves = nbx.leaves.median_split_leaves(b, 3, 16 if shape.endswith("16") else 8, reach=0.5)
    assert leaves[3].size > 200000
    with nbx.LeafPlan(n, 3, *leaves) as plan:
        slots, runs, groups, waves = plan.info()
        assert groups > 4096, (shape, groups)
    _check(nbx, oracle, b, leaves, nbx.LAW_TREE_LEAF, f"large launch, {shape}")


def test_invalid_structures_are_rejected_before_any_launch(nbx, oracle):
    b = oracle.generate(1, 10, 3)
    ok = (np.array([0, 5, 10]), np.arange(10), np.array([0, 1, 2]), np.array([0, 1]))
    nbx.leaf_pair_forces_hip(b, *ok)
    bad = [
        (np.array([0, 5, 10]), np.arange(10), np.array([0, 1, 2]), np.array([0, 2])),      # source leaf out of range
        (np.array([0, 5, 10]), np.r_[np.arange(9), 10], np.array([0, 1, 2]), np.array([0, 1])),   # body out of range
        (np.array([0, 5, 10]), np.r_[np.arange(9), 0], np.array([0, 1, 2]), np.array([0, 1])),    # body in two leaves
        (np.array([0, 6, 5]), np.arange(6), np.array([0, 1, 2]), np.array([0, 1])),        # decreasing offsets
        (np.array([1, 5, 10]), np.arange(10), np.array([0, 1, 2]), np.array([0, 1])),      # offsets not starting at 0
    ]
    for leaves in bad:
        for which in ("host", "device"):     # the device planner checks the indices as it follows them; nothing out of range is ever dereferenced
            with planner(which):
                with pytest.raises(nbx.NbxError):
                    nbx.leaf_pair_forces_hip(b, *leaves)
                with pytest.raises(nbx.NbxError):
                    nbx.LeafPlan(10, 3, *leaves)
                nbx.leaf_pair_forces_hip(b, *ok)        # and the next good call is served
    with pytest.raises(nbx.NbxError):
        nbx.leaf_pair_forces_hip(b, *ok, law=7)
    with nbx.LeafPlan(10, 3, *ok) as plan:
        with pytest.raises(nbx.NbxError):
            plan.forces(b, law=7)
        with pytest.raises(nbx.NbxError):
            plan.get_forces()                                       # nothing evaluated yet
        with pytest.raises(ValueError):
            plan.forces(oracle.generate(1, 11, 3))                  # another body count
        with nbx.Context(11, 3) as c, pytest.raises(nbx.NbxError):
            c.upload(oracle.generate(1, 11, 3))
            plan.forces_ctx(c)                                      # a context of another size
        with nbx.Context(10, 3, n_shards=2, shard=0) as c, pytest.raises(nbx.NbxError):
            c.upload(b)
            plan.forces_ctx(c)                                      # a sharded context
    assert nbx.leaf_pair_forces_hip(np.zeros((0, 7)), np.array([0]), np.zeros(0), np.array([0]), np.zeros(0)).shape == (0, 3)


def test_a_plan_names_its_context_by_identity_not_by_address(nbx, oracle):
    """the arena of a destroyed context is parked and handed to the next one of the same size, so an address cannot say
    which context an evaluation read.  The plan remembers the context's id: forces after the context is gone are refused, and a
    new context (same address or not) is not mistaken for the old one."""
    n = 4096
    b = oracle.round_inputs_to_f32(oracle.generate(31, n, 3))
    leaves = nbx.leaves.uniform_grid_leaves(b, 3, 2)
    with nbx.LeafPlan(n, 3, *leaves) as plan:
        c = nbx.Context(n, 3)
        c.upload(b)
        want = plan.forces_ctx(c, nbx.LAW_TREE_LEAF, oracle.G)
        plan.forces_ctx(c, nbx.LAW_TREE_LEAF, oracle.G, fetch=False)
        c.close()
        with pytest.raises(nbx.NbxError):
            plan.get_forces()                       # the masses lived in the context
        with nbx.Context(n, 3) as c2:               # takes the parked arena: very likely the same addresses
            c2.upload(b * np.r_[np.ones(6), 2.0])   # other masses
            with pytest.raises(nbx.NbxError):
                plan.kick_drift(c2, 1.0)            # never evaluated from THIS context
            got = plan.forces_ctx(c2, nbx.LAW_TREE_LEAF, oracle.G)
            assert np.allclose(got, 4.0 * want, rtol=1e-12)   # m_i m_j both doubled; the plan read c2's masses


def test_leaf_kernel_timing_at_fmm_like_sizes(nbx, oracle):
    """N = 2^20, leaves of <= ~100 bodies (FMM_MAX_BODIES_PER_LEAF, methods.h:26), 27-cell lists: EVERY body against the
    oracle (8.7e8 pair terms, a few seconds of its OpenMP loop), and the kernel's own time."""
    n, dim = 1 << 20, 3
    b = oracle.round_inputs_to_f32(oracle.generate(5, n, dim))
    leaves = nbx.leaves.uniform_grid_leaves(b, dim, 5)              # 32^3 cells, 32 bodies per leaf on average
    f, ms = nbx.leaf_pair_forces_hip(b, *leaves, law=nbx.LAW_FMM_P2P, G=oracle.G, return_kernel_ms=True)
    ref = oracle.leaf_pair_forces(b, leaves, 2)
    S = oracle.leaf_pair_magnitude_sums(b, leaves, 2)
    assert_force_parity(f, ref, S, "FMM-like leaves, every body")
    # the resident plan, the pinned TREE_LEAF law, every body; and what a force evaluation of an unchanged structure costs
    import time
    with nbx.LeafPlan(n, dim, *leaves) as plan, nbx.Context(n, dim) as c:
        c.upload(b)
        ft = plan.forces_ctx(c, nbx.LAW_TREE_LEAF, oracle.G)
        assert_force_parity(ft, oracle.leaf_pair_forces(b, leaves, 1), oracle.leaf_pair_magnitude_sums(b, leaves, 1), "plan, tree-leaf law, every body")
        walls = []
        for _ in range(20):
            c.synchronize()
            t0 = time.perf_counter()
            plan.forces_ctx(c, nbx.LAW_TREE_LEAF, oracle.G, fetch=False)
            c.synchronize()
            walls.append((time.perf_counter() - t0) * 1e3)
        assert np.array_equal(plan.get_forces(), ft)
        one = plan.forces_ctx(c, nbx.LAW_TREE_LEAF, oracle.G, fetch=False, timed=True)
        warm = plan.time_kernel(nbx.LAW_TREE_LEAF, 300)
        print(f"\nplan: evaluation of the unchanged structure from resident bodies: wall median {np.median(walls):.3f} ms (min {min(walls):.3f}), "
              f"pair kernel {one:.3f} ms single launch, {warm:.3f} ms mean of launches 151-300; slots/runs/workgroups/waves {plan.info()}")
        # measured 0.38-0.47 ms over the round's boxes (profiles/r4); the bound leaves room for a loaded box, the one-shot call takes 3.2 ms
        assert np.median(walls) <= 0.8, f"an evaluation of an unchanged structure took {np.median(walls):.3f} ms"
    lo, _, so, ss = leaves
    sizes = np.diff(lo).astype(np.int64)
    pairs = int(sum(sizes[t] * sizes[ss[so[t]:so[t + 1]]].sum() for t in range(sizes.size)))
    print(f"\nleaf-pair kernel: N={n}, {sizes.size} leaves (max {sizes.max()}), {pairs:.3e} pair terms in {ms:.2f} ms = {pairs / ms * 1e3:.3e} pairs/s")


def test_plan_stepping_matches_the_reference_helpers(nbx, oracle):
    """nbx_leaf_plan_kick_drift: update_body_velocities + update_body_positions (methods.cpp:425-450) fed the leaf sums of a
    standing structure, k steps on the device, against the same loop on the host built from the oracle's leaf sums on the
    fp32-representable positions the device sees.  Strong coupling (G x 1e26) so that the forces bend the paths; bodies in no
    leaf only drift.  Per-body bound from the stated force tolerance, as in test_config2_trajectory_with_coupling."""
    n, dim, steps, dt, scale = 8000, 3, 4, 1.5, 1e26
    b0 = oracle.round_inputs_to_f32(oracle.generate(150, n, dim))
    lo, lb, so, ss = nbx.leaves.uniform_grid_leaves(b0, dim, 2)
    keep = lo[-1] - 37                                             # the last 37 slots' bodies end up in no leaf
    lo = np.minimum(lo, keep)
    leaves = (lo, lb[:keep], so, ss)
    G = oracle.G * scale
    ref = b0.copy()
    S0 = oracle.leaf_pair_magnitude_sums(b0, leaves, 1) * scale
    for _ in range(steps):
        f = oracle.leaf_pair_forces(oracle.round_inputs_to_f32(ref), leaves, 1) * scale
        oracle.update_body_velocities(ref, np.ascontiguousarray(f), dt)
        oracle.update_body_positions(ref, dt)
    got = b0.copy()
    with nbx.LeafPlan(n, dim, *leaves) as plan, nbx.Context(n, dim) as c:
        c.upload(b0)
        for _ in range(steps):
            plan.forces_ctx(c, nbx.LAW_TREE_LEAF, G, fetch=False)
            plan.kick_drift(c, dt)
        c.download(got)
        # the context's own fp32 source copy followed the drift: a brute-force evaluation sees the new positions
        c.compute_accel()
        fb = c.forces(G)
    dv = np.linalg.norm(ref[:, dim:2 * dim] - b0[:, dim:2 * dim], axis=1)
    assert np.median(dv) > 1e-4 and dv.max() > 1.0, "coupling too weak to test anything"
    err = np.linalg.norm(got[:, dim:2 * dim] - ref[:, dim:2 * dim], axis=1)
    bound = 1.25 * steps * 4.0e-6 * S0 / b0[:, -1] * dt
    out = np.setdiff1d(np.arange(n), lb[:keep])
    assert out.size == 37 and np.array_equal(got[out, dim:2 * dim], b0[out, dim:2 * dim]), "bodies in no leaf keep their velocity"
    live = S0 > 0
    assert (err[live] <= bound[live]).all(), f"velocity error {float((err[live] / bound[live]).max()):.2f} x the per-body bound"
    assert np.allclose(got[:, :dim], ref[:, :dim], rtol=1e-9, atol=steps * dt * float(bound.max()))
    assert np.array_equal(got[:, -1], b0[:, -1])
    cur = oracle.round_inputs_to_f32(got)
    assert_force_parity(fb, oracle.brute_force_seq(cur) * scale, oracle.force_magnitude_sums(cur) * scale, "brute force after the plan's drifts")


@pytest.mark.parametrize("shape", ("grid cells of ~40 bodies", "packed grid cells"))
def test_plan_step_equals_the_two_calls(nbx, oracle, shape):
    """nbx_leaf_plan_step(plan, ctx, law, G, dt, k) leaves the bodies and the plan's sums exactly as k x {nbx_leaf_plan_forces_ctx;
    nbx_leaf_plan_kick_drift} do: bit for bit, call after call, after dt changed, with another context, and for k = 0."""
    n, dim, G = 20000, 3, oracle.G * 1e26
    b0 = oracle.round_inputs_to_f32(oracle.generate(151, n, dim))
    leaves = nbx.leaves.uniform_grid_leaves(b0, dim, 3 if shape.startswith("grid") else 4)
    law = nbx.LAW_TREE_LEAF

    def by_calls(c, plan, dt, k):
        for _ in range(k):
            plan.forces_ctx(c, law, G, fetch=False)
            plan.kick_drift(c, dt)

    with nbx.LeafPlan(n, dim, *leaves) as pa, nbx.LeafPlan(n, dim, *leaves) as pb, nbx.Context(n, dim) as ca, nbx.Context(n, dim) as cb, nbx.Context(n, dim) as cc:
        for c in (ca, cb, cc):
            c.upload(b0)
        ga, gb = b0.copy(), b0.copy()
        for dt, k in ((1.5, 3), (1.5, 2), (0.75, 2), (0.75, 0), (0.75, 1)):
            by_calls(ca, pa, dt, k)
            pb.step(cb, law, G, dt, k)
            ca.download(ga); cb.download(gb)
            assert np.array_equal(ga, gb), f"bodies after {k} steps of dt = {dt}"
            if k:
                assert np.array_equal(pa.get_forces(), pb.get_forces()), "the plan's sums after the last step"
        assert not np.array_equal(ga[:, dim:2 * dim], b0[:, dim:2 * dim]), "coupling too weak to test anything"
        pb.step(cc, law, G, 1.5, 3)                     # another context: its arrays, not the first one's
        by_calls(ca, pa, 1.5, 0)
        gc = b0.copy()
        cc.download(gc)
        with nbx.Context(n, dim) as cd:
            cd.upload(b0)
            by_calls(cd, pa, 1.5, 3)
            gd = b0.copy()
            cd.download(gd)
        assert np.array_equal(gc, gd)
        cb.download(gb)
        assert np.array_equal(ga, gb), "stepping another context must leave the first one alone"


def test_calls_reuse_the_parked_device_allocation(nbx, oracle):
    """A finished call parks its device allocation for the next one on that device (a tree code calls once per step); results must
    not depend on what the allocation held before: a large call, then smaller ones of another shape, a release in between."""
    lib = nbx.load_library()
    big = oracle.round_inputs_to_f32(oracle.generate(140, 30000, 3))
    big_leaves = nbx.leaves.uniform_grid_leaves(big, 3, 3)
    f_big = nbx.leaf_pair_forces_hip(big, *big_leaves, law=nbx.LAW_FMM_P2P, G=oracle.G)
    small = oracle.round_inputs_to_f32(oracle.generate(141, 3000, 2))
    small_leaves = nbx.leaves.uniform_grid_leaves(small, 2, 3)
    f1 = nbx.leaf_pair_forces_hip(small, *small_leaves, law=nbx.LAW_TREE_LEAF, G=oracle.G)      # inside the big call's allocation
    assert lib.nbx_release_cached() == 0
    f2 = nbx.leaf_pair_forces_hip(small, *small_leaves, law=nbx.LAW_TREE_LEAF, G=oracle.G)      # a fresh one
    assert np.array_equal(f1, f2)
    _check(nbx, oracle, small, small_leaves, nbx.LAW_TREE_LEAF, "small call inside a parked allocation")
    assert np.array_equal(f_big, nbx.leaf_pair_forces_hip(big, *big_leaves, law=nbx.LAW_FMM_P2P, G=oracle.G))   # grows again
    assert lib.nbx_release_cached() == 0
