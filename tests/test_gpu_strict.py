"""GPU: the strict fp64 kernel variant and the mixed mode (fp32 for all targets, fp64 for the suspects).

The reference's arithmetic is fp64 throughout (nbody-sim-new/vector.h:9-12, methods.cpp:21-37); the strict kernel is that
arithmetic type on the device's fp32-representable inputs, so against `brute_force_seq_n_body` on the same inputs it is held to
    |dF_i| <= 1e-9 |F_i|   and   |dF_i| <= 2e-12 S_i      (S_i = sum_j |f_ij|; two fp64 summation orders)
on small and edge-case inputs here and on >= 1,024 sampled rows at N = 2^20 / 2^22 in tests/all_bodies.py -- where, so pinned,
it then checks EVERY body of the default fp32 path and of the mixed mode on the device."""
import numpy as np
import pytest

import all_bodies
from all_bodies import TOL_STRICT_BACKWARD, TOL_STRICT_REL
from conftest import golden
from oracle_lib import KAPPA_WELL, TOL_BACKWARD, TOL_REL

pytestmark = pytest.mark.gpu
STRICT = "strict_f64_t4"


def _inputs(oracle, seed, n, dim):
    return oracle.round_inputs_to_f32(oracle.generate(seed, n, dim))


def _strict_forces(nbx, b, G, splits=0, variant=STRICT):
    n, dim = b.shape[0], (b.shape[1] - 1) // 2
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.set_tuning(splits, nbx.variants().index(variant))
        assert c.effective_tuning()[0] == variant
        c.compute_accel()
        return c.forces(G)


def _assert_strict(f, ref, S, what):
    d = np.sqrt(((f - ref) ** 2).sum(axis=1))
    nrm = np.sqrt((ref ** 2).sum(axis=1))
    assert np.isfinite(f).all(), what
    assert (d <= TOL_STRICT_REL * nrm).all(), f"{what}: relative {np.max(d / np.where(nrm > 0, nrm, 1)):.3e}"
    assert (d <= TOL_STRICT_BACKWARD * S).all(), f"{what}: backward {np.max(d / np.where(S > 0, S, 1)):.3e}"


@pytest.mark.parametrize("dim", (3, 2))
@pytest.mark.parametrize("n", (1, 2, 3, 255, 257, 1024, 1025, 4096))
def test_strict_matches_sequential_reference(nbx, oracle, dim, n):
    b = _inputs(oracle, 100 + n, n, dim)
    _assert_strict(_strict_forces(nbx, b, oracle.G), oracle.brute_force_seq(b), oracle.force_magnitude_sums(b), f"D={dim} N={n}")


@pytest.mark.parametrize("dim,n", [(d, n) for d in (2, 3) for n in (64, 1024)])
def test_strict_on_the_golden_fixtures(nbx, oracle, dim, n):
    """Outputs of the reference's own object code (tests/golden/make_golden.py)."""
    g = golden(f"bf_D{dim}_N{n}.npz")
    b = np.ascontiguousarray(g["bodies_f32"])
    _assert_strict(_strict_forces(nbx, b, float(g["G"])), g["forces_seq_f32"], oracle.force_magnitude_sums(b), f"golden D={dim} N={n}")


def test_strict_skip_rule_and_duplicates(nbx, oracle):
    """methods.cpp:24 compared in fp64 on the device: r^2 = 9.8e-11 skipped, 1.21e-10 counted; coincident bodies contribute 0."""
    k = golden("kat.npz")
    G = oracle.G
    f = _strict_forces(nbx, np.ascontiguousarray(k["two_bodies"]), G)
    assert abs(f[0, 0] + G / 8) <= 1e-15 * G and abs(f[1, 0] - G / 8) <= 1e-15 * G and not f[:, 1:].any()
    for name in ("near_skip", "near_keep"):
        b = oracle.round_inputs_to_f32(np.ascontiguousarray(k[name + "_bodies"]))
        ref = oracle.brute_force_seq(b)
        got = _strict_forces(nbx, b, G)
        if name == "near_skip":
            assert not ref.any() and not got.any()
        else:
            assert ref[0, 0] < 0 and np.allclose(got, ref, rtol=1e-12, atol=0)
    b = _inputs(oracle, 9, 600, 3)
    b[100:110, :3] = b[5, :3]
    b[300, :3] = (1.0, 1.0, 1.0)
    b[301, :3] = (1.0 + 2.4e-7, 1.0, 1.0)
    b = oracle.round_inputs_to_f32(b)
    _assert_strict(_strict_forces(nbx, b, G), oracle.brute_force_seq(b), oracle.force_magnitude_sums(b), "duplicates")


def test_strict_slices_and_shard_passes(nbx, oracle):
    """hi/lo planes per source slice; LOCAL + REMOTE (accumulating into the planes) equal ALL to fp64 re-association."""
    n, dim = 5000, 3
    b = _inputs(oracle, 11, n, dim)
    ref, S = oracle.brute_force_seq(b), oracle.force_magnitude_sums(b)
    v = nbx.variants().index(STRICT)
    for splits in (1, 2, 7):
        _assert_strict(_strict_forces(nbx, b, oracle.G, splits), ref, S, f"splits={splits}")
    for ranks in (2, 3):
        for r in range(ranks):
            with nbx.Context(n, dim, n_shards=ranks, shard=r) as c:
                c.upload(b)
                c.set_tuning(0, v)
                c.compute_accel(nbx.SRC_LOCAL)
                c.compute_accel(nbx.SRC_REMOTE)
                lo = r * c.shard_len
                _assert_strict(c.forces(oracle.G), ref[lo:lo + c.count], S[lo:lo + c.count], f"rank {r} of {ranks}")


def test_strict_leapfrog_matches_the_oracle_trajectory(nbx, oracle):
    """kick/drift fed by the strict kernel, with a coupling strong enough to bend the paths, against the oracle's leaves
    (methods.cpp:425-450) fed the oracle's forces on the same fp32-representable positions: with fp64 forces the velocity
    changes agree to 1e-9 of their size (the fp32 path's bound in test_leapfrog_strong_coupling is 2e-5)."""
    n, dim, steps, dt = 512, 3, 8, 2.0
    G = oracle.G * 1e24
    b = _inputs(oracle, 33, n, dim)
    ref = b.copy()
    for _ in range(steps):
        f = oracle.brute_force_seq(oracle.round_inputs_to_f32(ref)) * 1e24
        oracle.update_body_velocities(ref, np.ascontiguousarray(f), dt)
        oracle.update_body_positions(ref, dt)
    got = b.copy()
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.set_tuning(0, nbx.variants().index(STRICT))
        c.step(dt, steps, G)
        c.download(got)
    dv = np.abs(got[:, dim:2 * dim] - b[:, dim:2 * dim]).max()
    assert dv > 1e-3, "coupling too weak to test anything"
    assert np.allclose(got[:, dim:2 * dim], ref[:, dim:2 * dim], rtol=0, atol=1e-9 * dv)
    assert np.allclose(got[:, :dim], ref[:, :dim], rtol=1e-13, atol=0)


def test_mixed_mode_small_systems(nbx, oracle):
    """Mixed mode on small inputs: never worse than the plain fp32 path, the suspects come back at fp64 quality, and a
    tolerance so tight that every target is a suspect reproduces the strict kernel."""
    for dim, n in ((3, 3000), (2, 2500), (3, 100)):
        b = _inputs(oracle, 33 + n, n, dim)
        ref, S = oracle.brute_force_seq(b), oracle.force_magnitude_sums(b)
        with nbx.Context(n, dim) as c:
            c.upload(b)
            c.set_refine(0.0)
            c.compute_accel()
            plain = c.forces(oracle.G)
            c.set_refine(1e-5)
            c.compute_accel()
            mixed = c.forces(oracle.G)
            sel, done = c.refine_stats()
            Q = c.aux()
            assert sel == done and (Q >= 0).all()
            e_plain = np.sqrt(((plain - ref) ** 2).sum(axis=1)) / np.sqrt((ref ** 2).sum(axis=1))
            e_mixed = np.sqrt(((mixed - ref) ** 2).sum(axis=1)) / np.sqrt((ref ** 2).sum(axis=1))
            changed = (mixed != plain).any(axis=1)
            assert changed.sum() <= sel, "only listed targets may change"
            assert (e_mixed[changed] <= 1e-7).all(), "re-evaluated targets come back in fp64 (rounded to the hi/lo planes)"
            assert e_mixed.max() <= max(e_plain.max(), 1e-7)
            c.set_refine(1e-7, 1e6)   # every target is a suspect
            c.compute_accel()
            sel, done = c.refine_stats()
            assert sel == n and done == n
            _assert_strict(c.forces(oracle.G), ref, S, f"all targets refined D={dim} N={n}")
            c.set_refine(0.0)
            c.compute_accel()
            assert np.array_equal(c.forces(oracle.G), plain), "switching the mode off restores the plain path bit for bit"


def test_mixed_mode_sharded_and_stepping(nbx, oracle):
    """The refinement runs after the REMOTE pass over ALL chunks, and inside graph-replayed steps."""
    n, dim = 6000, 3
    b = _inputs(oracle, 77, n, dim)
    ref, S = oracle.brute_force_seq(b), oracle.force_magnitude_sums(b)
    for r in range(3):
        with nbx.Context(n, dim, n_shards=3, shard=r) as c:
            c.upload(b)
            c.set_refine(1e-7, 1e6)
            c.compute_accel(nbx.SRC_LOCAL)
            c.compute_accel(nbx.SRC_REMOTE)
            lo = r * c.shard_len
            assert c.refine_stats() == (c.count, c.count)
            _assert_strict(c.forces(oracle.G), ref[lo:lo + c.count], S[lo:lo + c.count], f"mixed, rank {r} of 3")
    G, dt = oracle.G * 1e24, 0.25
    a, bb = b.copy(), b.copy()
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.set_refine(1e-5)
        c.step(dt, 6, G)        # graph replay
        c.download(a)
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.set_refine(1e-5)
        for _ in range(6):      # eager
            c.compute_accel()
            c.kick_drift(dt, G)
        c.download(bb)
    assert np.array_equal(a, bb), "graph-replayed mixed-mode steps equal eager ones bit for bit"


def test_every_body_at_n1048576(nbx, oracle):
    """BASELINE config 3's input: all 1,048,576 bodies of the default fp32 path and of the mixed mode against the strict kernel,
    itself pinned to the oracle on >= 1,024 rows in the same test.  Frozen bounds (oracle_lib.py): T1 for every body, T2 for
    kappa <= KAPPA_WELL, and the north star's plain 1e-5 for EVERY body in mixed mode."""
    n = 1 << 20
    b = _inputs(oracle, 3, n, 3)
    rec = all_bodies.survey(nbx, oracle, b, "uniform 3D N=2^20 (BASELINE config 3 input, seed 3)")
    all_bodies.write_record(rec)
    print("\n" + "\n".join(f"  {k}: {v}" for k, v in rec.items()))
    assert rec["default"]["max_backward"] <= TOL_BACKWARD, rec["default"]
    assert rec["default"]["max_rel_kappa_le_4"] <= TOL_REL and KAPPA_WELL == 4.0, rec["default"]
    assert rec["mixed"]["max_rel"] <= TOL_REL and rec["mixed"]["n_over_tol"] == 0, rec["mixed"]
    assert rec["mixed"]["refined"] == rec["mixed"]["selected"] <= n // 50, rec["mixed"]
    assert rec["mixed"]["cost_ms"] <= 0.04 * rec["default_ms"], "mixed mode must stay within 4 % of the plain path"


def test_every_body_of_config5_n4194304(nbx, oracle):
    """BASELINE config 5's input (Plummer sphere, N = 4,194,304): the same check of every body."""
    n = 1 << 22
    b = oracle.round_inputs_to_f32(nbx.plummer_bodies(n, 3, seed=5, a=1.0e5, total_mass=1.0e12))
    rec = all_bodies.survey(nbx, oracle, b, "Plummer N=2^22 (BASELINE config 5 input, seed 5)")
    all_bodies.write_record(rec)
    print("\n" + "\n".join(f"  {k}: {v}" for k, v in rec.items()))
    assert rec["default"]["max_backward"] <= TOL_BACKWARD, rec["default"]
    assert rec["default"]["max_rel_kappa_le_4"] <= TOL_REL, rec["default"]
    assert rec["mixed"]["max_rel"] <= TOL_REL and rec["mixed"]["n_over_tol"] == 0, rec["mixed"]
    assert rec["mixed"]["refined"] == rec["mixed"]["selected"] <= n // 50, rec["mixed"]


def test_mixed_mode_long_lists_and_node_layer(nbx, oracle):
    """There is no capacity to overflow (round 3 re-evaluated at most max(16,384, shard/16) targets and said so only in
    nbx_ctx_refine_stats): the list has room for every target of the shard and the fp64 pass picks its source slices on the device
    from the list's length.  With EVERY target of a 40,000-body system listed -- 157 list blocks, the buffer's budget then allows
    fewer slices than a short list gets -- every body comes back at fp64 quality, nothing lost or written twice.  And the
    single-process node layer (nbx_node_set_refine / nbx_node_refine_stats: every rank refines its own shard after its REMOTE pass)."""
    n, dim = 40000, 3
    b = _inputs(oracle, 55, n, dim)
    rows = np.arange(0, n, 13)
    ref = oracle.force_rows_omp_2(b, rows)
    S = oracle.force_magnitude_sums(b, rows)
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.set_refine(0.0)
        c.compute_accel()
        plain = c.forces(oracle.G)
        c.set_refine(1e-7, 1e6)                     # every target is a suspect
        c.compute_accel()
        sel, done = c.refine_stats()
        mixed = c.forces(oracle.G)
        c.set_tuning(0, nbx.variants().index(STRICT))
        c.compute_accel()
        strict = c.forces(oracle.G)
    assert sel == n and done == n, (sel, done)
    changed = (mixed != plain).any(axis=1)
    assert changed.sum() >= n - 400                 # (a re-evaluated target may happen to round to the same fp32-plane pair)
    _assert_strict(mixed[rows], ref, S, "every target listed")
    # against the whole-chunk strict kernel on ALL bodies: same arithmetic, one Newton step less and another slice order
    d = np.sqrt(((mixed - strict) ** 2).sum(axis=1)) / np.sqrt((strict ** 2).sum(axis=1))
    assert d.max() <= 1e-9, d.max()
    for dim2, n2 in ((2, 9000), (3, 300000)):       # 2D; and a list of 1,172 blocks: each of the 32 workgroup rows walks ~37 of them
        bb = _inputs(oracle, 57 + dim2, n2, dim2)
        with nbx.Context(n2, dim2) as c:
            c.upload(bb)
            c.set_refine(1e-7, 1e6)
            c.compute_accel()
            assert c.refine_stats() == (n2, n2)
            mixed = c.forces(oracle.G)
            c.set_tuning(0, nbx.variants().index(STRICT))
            c.compute_accel()
            strict = c.forces(oracle.G)
        d = np.sqrt(((mixed - strict) ** 2).sum(axis=1)) / np.sqrt((strict ** 2).sum(axis=1))
        assert d.max() <= 1e-9, (dim2, n2, d.max())
    m = 6000
    bb = _inputs(oracle, 56, m, dim)
    with nbx.Node(m, dim, [0, 0, 0]) as node:
        node.upload(bb)
        node.set_refine(1e-7, 1e6)
        f = node.forces(oracle.G)
        assert node.refine_stats() == (m, m)
    _assert_strict(f, oracle.brute_force_seq(bb), oracle.force_magnitude_sums(bb), "node layer, all targets refined")


def test_mixed_mode_is_the_default_precision(nbx, oracle):
    """ABI 4: every entry point a maintainer binds runs the mixed mode unless told otherwise -- a new context, the one-shot
    call (nbx_brute_force_forces = nbx_brute_force_forces_ex with rel_tolerance < 0), the node layer; nbx_set_default_refine
    changes it process-wide for what is created afterwards.  On a 2D input (plane sums cancel hard: dozens of suspects at this
    size) the default result equals the explicit 1e-5 one bit for bit and differs from plain fp32 on listed bodies only."""
    assert nbx.get_default_refine() == (1e-5, 0.0)
    n, dim = 30000, 2
    b = _inputs(oracle, 91, n, dim)
    f_default, info = nbx.brute_force_hip_n_body(b, oracle.G, return_info=True)
    assert info.refine_tolerance == 1e-5 and info.refine_selected == info.refine_refined > 0 and info.refine_ms > 0
    f_explicit = nbx.brute_force_hip_n_body(b, oracle.G, rel_tolerance=1e-5)
    f_plain, info0 = nbx.brute_force_hip_n_body(b, oracle.G, rel_tolerance=0.0, return_info=True)
    assert info0.refine_tolerance == 0.0 and info0.refine_selected == 0
    assert np.array_equal(f_default, f_explicit)
    changed = (f_default != f_plain).any(axis=1)
    assert 0 < changed.sum() <= info.refine_selected
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.compute_accel()
        assert c.refine_stats() == (info.refine_selected, info.refine_selected)
        assert np.array_equal(c.forces(oracle.G), f_default)
    with nbx.Node(n, dim, [0, 0]) as node:
        node.upload(b)
        node.forces(oracle.G)
        assert node.refine_stats()[0] > 0
    try:
        nbx.set_default_refine(0.0)
        assert nbx.get_default_refine() == (0.0, 0.0)
        assert np.array_equal(nbx.brute_force_hip_n_body(b, oracle.G), f_plain)
        with nbx.Context(n, dim) as c:
            c.upload(b)
            c.compute_accel()
            with pytest.raises(nbx.NbxError):
                c.refine_stats()
        with pytest.raises(nbx.NbxError):
            nbx.set_default_refine(0.5)
    finally:
        nbx.set_default_refine(1e-5)
    # every body of this input within the tolerance in the default precision (oracle: the sequential reference on the same inputs)
    ref = oracle.brute_force_seq(b)
    rel = np.sqrt(((f_default - ref) ** 2).sum(axis=1)) / np.sqrt((ref ** 2).sum(axis=1))
    assert rel.max() <= TOL_REL, rel.max()
