"""GPU: the HIP path (through the C ABI) against the CPU oracle = the reference's sequential
brute-force path on the same fp32-representable inputs.  Tolerance: oracle_lib.TOL_* (stated in
DESIGN.md): (T1) |dF| <= 4e-6 * sum_j|f_ij| for every body and (T2) |dF| <= 1e-5*|F| for every body whose
pair forces do not cancel below 1/4 of their magnitude sum; the unconditional 1e-5 bound (T3) is asserted on
BASELINE's uniform 3D configs in test_gpu_fullsize.py."""
import numpy as np
import pytest

from conftest import golden
from oracle_lib import TOL_BACKWARD, assert_force_parity

pytestmark = pytest.mark.gpu


def _oracle_inputs(oracle, seed, n, dim):
    return oracle.round_inputs_to_f32(oracle.generate(seed, n, dim))


@pytest.mark.parametrize("dim", (3, 2))
@pytest.mark.parametrize("n", (1, 2, 3, 255, 257, 1024, 1025, 4096))
def test_forces_match_sequential_reference(nbx, oracle, dim, n):
    b = _oracle_inputs(oracle, 100 + n, n, dim)
    ref = oracle.brute_force_seq(b)
    f = nbx.brute_force_hip_n_body(b, oracle.G)
    assert f.shape == (n, dim)
    assert_force_parity(f, ref, oracle.force_magnitude_sums(b), f"D={dim} N={n}")


@pytest.mark.parametrize("dim,n", [(d, n) for d in (2, 3) for n in (2, 3, 64, 1024)])
def test_golden_fixtures(nbx, oracle, dim, n):
    """Committed outputs of the reference's own object code for the fp32-representable inputs."""
    g = golden(f"bf_D{dim}_N{n}.npz")
    b = np.ascontiguousarray(g["bodies_f32"])
    f = nbx.brute_force_hip_n_body(b, float(g["G"]))
    assert_force_parity(f, g["forces_seq_f32"], oracle.force_magnitude_sums(b), f"golden D={dim} N={n}")
    # the reference's own 1 % accuracy metric (utils.h:170-219) must read 100 %
    assert oracle.compute_accuracy(f, g["forces_seq_f32"]) == 100.0
    # unrounded fp64 inputs: input quantisation alone costs ~5e-5 relative; still 100 % by the 1 % metric
    f2 = nbx.brute_force_hip_n_body(np.ascontiguousarray(g["bodies"]), float(g["G"]))
    assert oracle.compute_accuracy(f2, g["forces_seq"]) == 100.0


def test_every_variant(nbx, oracle):
    names = nbx.variants()
    for dim, n in ((3, 1500), (2, 700)):
        b = _oracle_inputs(oracle, 5, n, dim)
        ref = oracle.brute_force_seq(b)
        S = oracle.force_magnitude_sums(b)
        with nbx.Context(n, dim) as c:
            c.upload(b)
            for v, name in enumerate(names):
                c.set_tuning(0, v)
                assert c.effective_tuning()[0] == name
                c.compute_accel()
                assert_force_parity(c.forces(oracle.G), ref, S, f"variant {name} D={dim}")


def test_known_answers_and_guard(nbx, oracle):
    k = golden("kat.npz")
    G = oracle.G
    f = nbx.brute_force_hip_n_body(np.ascontiguousarray(k["two_bodies"]), G)
    assert abs(f[0, 0] + G / 8) <= 1e-6 * G / 8 and abs(f[1, 0] - G / 8) <= 1e-6 * G / 8 and not f[:, 1:].any()
    c = nbx.brute_force_hip_n_body(np.ascontiguousarray(k["coincident_bodies"]), G)
    assert np.allclose(c, k["coincident_forces_seq"], rtol=1e-6, atol=0)
    # r^2 = 9.8e-11 < 1e-10: skipped exactly; r^2 = 1.21e-10: counted (methods.cpp:24).  Positions
    # near 1.0 have an fp32 spacing of 1.2e-7, so both separations survive the rounding.
    for name in ("near_skip", "near_keep"):
        b = oracle.round_inputs_to_f32(np.ascontiguousarray(k[name + "_bodies"]))
        ref = oracle.brute_force_seq(b)
        got = nbx.brute_force_hip_n_body(b, G)
        if name == "near_skip":
            assert not ref.any() and not got.any()
        else:
            assert ref[0, 0] < 0 and np.allclose(got, ref, rtol=1e-5, atol=0)
    # many coincident / sub-threshold pairs inside one tile and across tiles
    b = _oracle_inputs(oracle, 9, 600, 3)
    b[100:110, :3] = b[5, :3]                       # exact duplicates of body 5
    b[300, :3] = (1.0, 1.0, 1.0)
    b[301, :3] = (1.0 + 2.4e-7, 1.0, 1.0)            # 0 < r^2 < 1e-10 apart (2 ulp at 1.0)
    b = oracle.round_inputs_to_f32(b)
    ref = oracle.brute_force_seq(b)
    assert_force_parity(nbx.brute_force_hip_n_body(b, G), ref, oracle.force_magnitude_sums(b), "duplicates")


def test_empty_input(nbx):
    for dim in (2, 3):
        f = nbx.brute_force_hip_n_body(np.zeros((0, 2 * dim + 1)))
        assert f.shape == (0, dim)


def test_source_slices_and_shard_passes(nbx, oracle):
    """acc partial slices summed in fixed order; LOCAL + REMOTE source passes equal the ALL pass;
    shards (one context per virtual rank on one device) reproduce the single-context result."""
    n, dim = 5000, 3
    b = _oracle_inputs(oracle, 11, n, dim)
    ref = oracle.brute_force_seq(b)
    S = oracle.force_magnitude_sums(b)
    base = None
    with nbx.Context(n, dim) as c:
        c.upload(b)
        for splits in (1, 2, 7, 16):
            c.set_tuning(splits, -1)
            c.compute_accel()
            f = c.forces(oracle.G)
            assert_force_parity(f, ref, S, f"splits={splits}")
            c.compute_accel()
            assert np.array_equal(f, c.forces(oracle.G)), "same launch twice must be bit-identical"
            base = f if base is None else base
    for G_ in (2, 3, 8):
        parts = []
        for r in range(G_):
            with nbx.Context(n, dim, n_shards=G_, shard=r) as c:
                c.upload(b)
                c.compute_accel(nbx.SRC_ALL)
                fa = c.forces(oracle.G)
                c.compute_accel(nbx.SRC_LOCAL)
                c.compute_accel(nbx.SRC_REMOTE)
                fb = c.forces(oracle.G)
                assert fa.shape == (c.count, dim)
                lo = r * c.shard_len
                assert_force_parity(fb, ref[lo:lo + c.count], S[lo:lo + c.count], f"local+remote G={G_} r={r}")
                assert np.abs(fa - fb).max() <= 4 * TOL_BACKWARD * S[lo:lo + c.count].max()
                parts.append(fa)
        full = np.concatenate(parts)
        assert full.shape == ref.shape
        assert_force_parity(full, ref, S, f"sharded G={G_}")


@pytest.mark.parametrize("dim", (3, 2))
def test_leapfrog_matches_reference_helpers(nbx, oracle, dim):
    """nsteps x {forces; update_body_velocities; update_body_positions} (methods.cpp:425-450).
    The integrator runs in fp64 on the device with the reference's arithmetic; only the pair sum is
    fp32, and with G = 4.471e-21 it moves velocities by ~1e-25 per step, so positions and velocities
    agree with the fp64 oracle to the last bits of a double; a larger G exercises the coupling."""
    n, steps, dt = 777, 6, 250.0
    b0 = _oracle_inputs(oracle, 21, n, dim)
    ref = b0.copy()
    oracle.leapfrog(ref, dt, steps, variant=0)
    got = b0.copy()
    nbx.leapfrog_hip_n_body(got, dt, steps, oracle.G)
    assert np.array_equal(got[:, -1], b0[:, -1])
    assert np.allclose(got[:, :dim], ref[:, :dim], rtol=1e-14, atol=0)
    assert np.allclose(got[:, dim:2 * dim], ref[:, dim:2 * dim], rtol=1e-14, atol=0)
    # golden trajectory from the reference's own helpers (fp64 inputs: positions differ by the
    # fp32 rounding of the *sources* only through forces ~1e-20 => still ~1e-15 relative)
    g = golden(f"traj_D{dim}_N64.npz")
    cur = np.ascontiguousarray(g["states"][0]).copy()
    nbx.leapfrog_hip_n_body(cur, float(g["dt"]), int(g["steps"]), oracle.G)
    assert np.allclose(cur, g["states"][-1], rtol=1e-13, atol=0)


def test_leapfrog_strong_coupling(nbx, oracle):
    """G scaled up by 1e24 so that forces bend the trajectories: device state vs a host loop built
    from the oracle's leaves with the same G (forces = oracle forces * G'/G)."""
    n, dim, steps, dt = 512, 3, 8, 2.0
    Gs = oracle.G * 1e24
    b0 = _oracle_inputs(oracle, 33, n, dim)
    ref = b0.copy()
    for _ in range(steps):
        f = oracle.brute_force_seq(oracle.round_inputs_to_f32(ref)) * 1e24
        oracle.update_body_velocities(ref, np.ascontiguousarray(f), dt)
        oracle.update_body_positions(ref, dt)
    got = b0.copy()
    nbx.leapfrog_hip_n_body(got, dt, steps, Gs)
    dv = np.abs(got[:, dim:2 * dim] - b0[:, dim:2 * dim]).max()
    assert dv > 1e-3, "coupling too weak to test anything"
    assert np.allclose(got[:, dim:2 * dim], ref[:, dim:2 * dim], rtol=0, atol=2e-5 * dv)
    assert np.allclose(got[:, :dim], ref[:, :dim], rtol=1e-9, atol=0)


def test_context_step_by_step_equals_one_shot(nbx, oracle):
    n, dim = 900, 3
    b = _oracle_inputs(oracle, 44, n, dim)
    one = b.copy()
    nbx.leapfrog_hip_n_body(one, 10.0, 3, oracle.G)
    two = b.copy()
    with nbx.Context(n, dim) as c:
        c.upload(b)
        for _ in range(3):
            c.compute_accel()
            c.kick_drift(10.0, oracle.G)
        c.download(two)
    assert np.array_equal(one, two)


def test_graph_replayed_steps_equal_eager_steps(nbx, oracle):
    """nbx_ctx_step captures one step into a hipGraph from 4 steps on; the replay must be bit-identical to
    launching the same kernels one by one."""
    for n, dim in ((5000, 3), (700, 2)):
        b = _oracle_inputs(oracle, 46, n, dim)
        b[:, :dim] = b[:, :dim] / 50.0           # pull part of the system into the candidate region
        b = oracle.round_inputs_to_f32(b)
        eager, graph = b.copy(), b.copy()
        Gs = oracle.G * 1e22
        with nbx.Context(n, dim) as c:
            c.upload(b)
            for _ in range(12):
                c.compute_accel()
                c.kick_drift(1.5, Gs)
            c.download(eager)
        with nbx.Context(n, dim) as c:
            c.upload(b)
            c.step(1.5, 7, Gs)
            c.step(1.5, 5, Gs)                   # second call reuses the instantiated graph
            c.download(graph)
        assert np.array_equal(eager, graph)
        assert np.abs(eager[:, dim:2 * dim] - b[:, dim:2 * dim]).max() > 0


def test_wrong_order_is_an_error(nbx, oracle):
    with nbx.Context(16, 3) as c:
        with pytest.raises(nbx.NbxError):
            c.compute_accel()
        c.upload(oracle.generate(1, 16, 3))
        with pytest.raises(nbx.NbxError):
            c.kick_drift(1.0)
        with pytest.raises(nbx.NbxError):
            c.compute_accel(nbx.SRC_REMOTE)
    with nbx.Context(16, 3, n_shards=2, shard=1) as c:
        c.upload(oracle.generate(1, 16, 3))
        with pytest.raises(nbx.NbxError):
            c.step(1.0, 1)


def _fast_variants(nbx):
    return [(i, n) for i, n in enumerate(nbx.variants()) if n.startswith("fast")]


def test_fast_path_close_set_semantics(nbx, oracle):
    """The unguarded fast kernels + close-set pipeline must reproduce the reference's skip rule for
    every pair: sub-threshold pairs (0 < r^2 < 1e-10), exact duplicates and ordinary close pairs, both
    for candidate targets (a coordinate below 16384) and for all others."""
    n, dim = 20000, 3
    b = oracle.generate(91, n, dim)
    b[:1500, 0] = 1.0 + 8000.0 * (b[:1500, 0] / 1e7)          # 7.5 % of the bodies near the x = 0 plane
    b[200, :3] = (3.0, 5.0e6, 5.0e6)
    b[201, :3] = (3.0 + 4.8e-7, 5.0e6, 5.0e6)                 # 2 ulp apart: r^2 = 2.3e-13 -> skipped
    b[300, :3] = (100.0, 2.0e6, 2.0e6)
    b[301, :3] = (100.0 + 7.7e-6, 2.0e6, 2.0e6)               # r^2 = 5.9e-11 -> skipped
    b[302, :3] = (100.0 + 1.6e-5, 2.0e6, 2.0e6)               # r^2 = 2.6e-10 from body 300 -> counted
    b[5000:5004, :3] = b[4999, :3]                            # exact duplicates far from every plane
    b[6000, :3] = (9000.0, 9000.0, 9000.0)
    b[6001, :3] = (9000.0 + 9.765625e-4, 9000.0, 9000.0)      # closest possible pair outside the close set
    b = oracle.round_inputs_to_f32(b)
    ref = oracle.brute_force_seq(b)
    S = oracle.force_magnitude_sums(b)
    with nbx.Context(n, dim) as c:
        c.upload(b)
        for v, name in _fast_variants(nbx):
            c.set_tuning(0, v)
            assert c.effective_tuning()[0] == name, "preconditions hold: the fast variant itself must run"
            c.compute_accel()
            f = c.forces(oracle.G)
            assert_force_parity(f, ref, S, f"fast variant {name}")
            # the skipped pair exerts nothing: bodies 200/201 feel the same field up to their 5e-7 offset
            c.compute_accel()
            assert np.array_equal(f, c.forces(oracle.G)), "bit-reproducible despite the atomically built list"
    # sharded: close-set targets go through LOCAL (store) and REMOTE (accumulate) scatter paths
    v0 = _fast_variants(nbx)[0][0]
    for r in range(3):
        with nbx.Context(n, dim, n_shards=3, shard=r) as c:
            c.upload(b)
            c.set_tuning(0, v0)
            c.compute_accel(nbx.SRC_LOCAL)
            c.compute_accel(nbx.SRC_REMOTE)
            lo = r * c.shard_len
            assert_force_parity(c.forces(oracle.G), ref[lo:lo + c.count], S[lo:lo + c.count], f"fast sharded r={r}")


def test_fast_path_preconditions_fall_back_to_guarded_kernel(nbx, oracle):
    n, dim = 3000, 3
    fast = _fast_variants(nbx)[0]
    # (1) a mass above 1e10 would overflow m/kTiny^2 for a coincident source
    b = oracle.round_inputs_to_f32(oracle.generate(5, n, dim))
    b[7, -1] = float(np.float32(3.0e12))
    b[8, :3] = b[7, :3]
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.set_tuning(0, fast[0])
        assert "exact" in c.effective_tuning()[0]
        c.compute_accel()
        assert_force_parity(c.forces(oracle.G), oracle.brute_force_seq(b), oracle.force_magnitude_sums(b), "huge mass")
    # (1b) the scan runs on the device (pack pass) over every shard's bodies: a negative huge mass in another shard, or a NaN
    #      mass anywhere but last, must be seen as well
    for bad in (-3.0e12, float("nan")):
        b = oracle.round_inputs_to_f32(oracle.generate(5, n, dim))
        b[n // 2 + 11, -1] = bad
        with nbx.Context(n, dim, n_shards=3, shard=0) as c:
            c.upload(b)
            c.set_tuning(0, fast[0])
            assert "exact" in c.effective_tuning()[0], bad
    # (2) a whole system inside the candidate region (small box) no longer forces the guarded kernel: see
    #     test_small_coordinate_systems_keep_the_fast_path (sorted-cell refinement); a system in which MOST targets really
    #     own a sub-threshold pair still does
    b = oracle.generate(6, n, dim)
    b[:, :3] = b[:, :3] / 1e9                                   # box of 0.01: typical separations 7e-4, r^2 ~ 5e-7 and below
    b = oracle.round_inputs_to_f32(b)
    with nbx.Context(n, dim) as c:
        c.upload(b)
        assert "exact" in c.effective_tuning()[0]
        c.compute_accel()
        assert_force_parity(c.forces(oracle.G), oracle.brute_force_seq(b), oracle.force_magnitude_sums(b), "tiny box")
    # (3) bodies that DRIFT into the candidate region after upload stay correct (lists are rebuilt per position update)
    b = oracle.round_inputs_to_f32(oracle.generate(7, n, dim))
    b[:, 3:6] = 0.0
    b[:300, 0] = 17000.0 + np.arange(300) * 3.0
    b[:300, 3] = -100.0                                        # 10 steps of dt=1 carry 128 of them across x = 16384
    b = oracle.round_inputs_to_f32(b)
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.set_tuning(0, fast[0])
        assert c.effective_tuning()[0] == fast[1]
        c.step(1.0, 10, oracle.G)
        cur = b.copy()
        c.download(cur)
        assert (cur[:300, 0] < 16384).sum() >= 100
        c.compute_accel()
        cr = oracle.round_inputs_to_f32(cur)
        assert_force_parity(c.forces(oracle.G), oracle.brute_force_seq(cr), oracle.force_magnitude_sums(cr), "after drift")


def test_one_reciprocal_variant_extent_precondition(nbx, oracle):
    """fastpk1r (one v_rcp_f32 per two pairs: W = 1/(r2a*r2b)) is only launched while every |coordinate| <= 1e8, so the
    product cannot overflow; beyond that the library substitutes the two-reciprocal fast kernel.  Both sides of the
    switch against the oracle, including separations of ~2e9 where the product WOULD overflow."""
    one = [i for i, n in enumerate(nbx.variants()) if n.startswith("fastpk1r")][0]
    n, dim = 4096, 3
    b = oracle.generate(77, n, dim)
    b[:, :3] = (b[:, :3] - 5.0e6) * 18.0                         # |coordinates| up to 9e7 < 1e8: separations up to 3e8
    b[5, :3] = b[4, :3]                                          # coincident pair: r^2 = kTiny on one side of the product
    b = oracle.round_inputs_to_f32(b)
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.set_tuning(0, one)
        assert c.effective_tuning()[0].startswith("fastpk1r")
        c.compute_accel()
        assert_force_parity(c.forces(oracle.G), oracle.brute_force_seq(b), oracle.force_magnitude_sums(b), "one-rcp inside the extent")
    for scale in (30.0, 250.0):                                  # |coordinates| up to 1.5e8 / 1.2e9 (r2a*r2b up to 3e38: would overflow)
        big = oracle.generate(78, n, dim)
        big[:, :3] = (big[:, :3] - 5.0e6) * scale
        big = oracle.round_inputs_to_f32(big)
        with nbx.Context(n, dim) as c:
            c.upload(big)
            c.set_tuning(0, one)
            assert c.effective_tuning()[0].startswith("fastpk_"), "extent precondition must demote the one-reciprocal kernel"
            c.compute_accel()
            f = c.forces(oracle.G)
        assert np.isfinite(f).all()
        assert_force_parity(f, oracle.brute_force_seq(big), oracle.force_magnitude_sums(big), f"extent x{scale:g}")
    # a NaN in the FIRST coordinate of a body (finite y, z after it): the pack pass's running maximum must keep the NaN, so
    # the extent precondition fails and the two-reciprocal kernel is the one that runs ("a NaN sorts above everything")
    nanb = oracle.round_inputs_to_f32(oracle.generate(79, n, dim))
    nanb[100, 0] = float("nan")
    with nbx.Context(n, dim) as c:
        c.upload(nanb)
        c.set_tuning(0, one)
        assert c.effective_tuning()[0].startswith("fastpk_"), "a NaN coordinate must fail the extent precondition"


@pytest.mark.parametrize("dim", (3, 2))
def test_small_coordinate_systems_keep_the_fast_path(nbx, oracle, dim):
    """Unit-scale systems (every coordinate below 16384, so EVERY body is a close-set candidate): the fast kernel stays in
    use, with the close pairs found through sorted cells (csrc/close_hash.hip) instead of candidates x candidates --
    single context, graph-replayed steps, and shards with planted pairs straddling the shard boundary."""
    import cross_shard_case as csc
    n = 40000
    rng = np.random.default_rng(3)
    b = oracle.generate(60 + dim, n, dim)
    b[:, :dim] = rng.normal(scale=1.0, size=(n, dim))          # Gaussian blob around the origin, both signs
    L = csc.shard_len(n, 2)
    b[100, :dim] = b[101, :dim];            b[100, 0] += 4.0e-6      # r^2 = 1.6e-11: skipped by the reference
    b[200, :dim] = b[201, :dim];            b[200, 1] += 2.0e-4      # r^2 = 4e-8: counted, below the kTiny-bias level
    b[300, :dim] = b[301, :dim]                                      # exact duplicates
    b[L - 1, :dim] = b[L, :dim];            b[L - 1, 0] += 4.0e-6    # the same three kinds straddling the 2-shard boundary
    b[L - 2, :dim] = b[L + 1, :dim];        b[L - 2, 1] += 2.0e-4
    b[L - 3, :dim] = b[L + 2, :dim]
    b = oracle.round_inputs_to_f32(b)
    ref = oracle.brute_force_seq(b)
    S = oracle.force_magnitude_sums(b)
    special = [100, 101, 200, 201, 300, 301, L - 1, L, L - 2, L + 1, L - 3, L + 2]
    with nbx.Context(n, dim) as c:
        c.upload(b)
        assert c.effective_tuning()[0].startswith("fast"), c.effective_tuning()
        c.compute_accel()
        f = c.forces(oracle.G)
        assert_force_parity(f, ref, S, f"small coordinates D={dim}")
        for i in special:
            assert np.linalg.norm(f[i] - ref[i]) <= 5e-5 * np.linalg.norm(ref[i]), (i, f[i], ref[i])
        c.compute_accel()
        assert np.array_equal(f, c.forces(oracle.G))
        # stepping (graph replay from 4 steps on) rebuilds the sorted cells every step
        Gs = oracle.G * 1e-6
        eager, graph = b.copy(), b.copy()
        for _ in range(6):
            c.compute_accel()
            c.kick_drift(1e-3, Gs)
        c.download(eager)
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.step(1e-3, 6, Gs)
        c.download(graph)
    assert np.array_equal(eager, graph)
    for r in range(2):
        with nbx.Context(n, dim, n_shards=2, shard=r) as c:
            c.upload(b)
            assert c.effective_tuning()[0].startswith("fast")
            lo = r * c.shard_len
            c.compute_accel(nbx.SRC_LOCAL)
            c.compute_accel(nbx.SRC_REMOTE)
            fs = c.forces(oracle.G)
            assert_force_parity(fs, ref[lo:lo + c.count], S[lo:lo + c.count], f"small coordinates, shard {r} D={dim}")
            for i in special:
                if lo <= i < lo + c.count:
                    assert np.linalg.norm(fs[i - lo] - ref[i]) <= 5e-5 * np.linalg.norm(ref[i]), (r, i)


def test_signed_and_extreme_coordinates(nbx, oracle):
    """Bodies on both sides of the coordinate planes, at the origin, and spread over 12 decades of
    separation: the candidate test works on |coordinate| and the fp32 path must neither overflow nor
    lose the skip rule."""
    n, dim = 4096, 3
    rng = np.random.default_rng(5)
    b = oracle.generate(8, n, dim)
    b[:, :3] -= 5.0e6                                          # box centred on the origin: all sign combinations
    b[:64, :3] = rng.normal(scale=30.0, size=(64, 3))          # a tight cluster straddling the planes
    b[0, :3] = 0.0
    b[1, :3] = (1e-4, -1e-4, 0.0)                              # 1.4e-4 from the origin: counted
    b[2, :3] = (3e-6, 0.0, 0.0)                                # 3e-6 from the origin: r^2 = 9e-12, skipped
    b[100, :3] = (-7.0e6, 7.0e6, -7.0e6)
    b[101:104, -1] = (1.0, 1.0e-3, 9.9e7)
    b = oracle.round_inputs_to_f32(b)
    ref = oracle.brute_force_seq(b)
    S = oracle.force_magnitude_sums(b)
    for v, name in enumerate(nbx.variants()):
        with nbx.Context(n, dim) as c:
            c.upload(b)
            c.set_tuning(0, v)
            c.compute_accel()
            assert_force_parity(c.forces(oracle.G), ref, S, f"signed coordinates, {c.effective_tuning()[0]}")


@pytest.mark.parametrize("kind", ("blobs", "lattice", "plane", "masses", "tiny_box_offset"))
def test_structured_distributions(nbx, oracle, kind):
    """Inputs that are not uniform-random: clustered bodies, a lattice (many equal distances and exact
    duplicates), a system confined to a plane (dz = 0 everywhere), masses spread over 12 decades, and a small
    box far from the origin (large coordinates, small separations: the regime where fp32 input rounding bites,
    hence the oracle on the rounded inputs)."""
    rng = np.random.default_rng(17)
    n, dim = 3000, 3
    b = oracle.generate(40, n, dim)
    if kind == "blobs":
        centres = rng.uniform(1e6, 9e6, size=(6, 3))
        b[:, :3] = centres[rng.integers(0, 6, n)] + rng.normal(scale=2.0e3, size=(n, 3))
    elif kind == "lattice":
        g = np.stack(np.meshgrid(*[np.arange(15)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n].astype(float)
        b[:, :3] = 1.0e6 + 512.0 * g
        b[7, :3] = b[8, :3]
    elif kind == "plane":
        b[:, 2] = 4.0e6
    elif kind == "masses":
        b[:, -1] = 10.0 ** rng.uniform(-4, 8, n)
    elif kind == "tiny_box_offset":
        b[:, :3] = 8.0e6 + rng.uniform(0, 5.0e3, size=(n, 3))
    b = oracle.round_inputs_to_f32(b)
    ref = oracle.brute_force_seq(b)
    S = oracle.force_magnitude_sums(b)
    f = nbx.brute_force_hip_n_body(b, oracle.G)
    assert_force_parity(f, ref, S, kind)
    assert oracle.compute_accuracy(f, ref) == 100.0 or kind in ("lattice",)   # symmetric lattices have exactly cancelling components


def test_fast_path_boundary_pairs(nbx, oracle):
    """Pairs sitting exactly on the thresholds of the close-set argument: one fp32 step apart across 8192 and
    across 16384 (candidate / non-candidate split), and sub-threshold neighbours just inside the candidate
    region -- every one must come out like the reference's guarded sum."""
    n, dim = 4096, 3
    b = oracle.generate(55, n, dim)
    f32 = np.float32
    def below(x):
        return float(np.nextafter(f32(x), f32(0)))
    b[10, :3] = (below(8192.0), 5.0e6, 5.0e6);  b[11, :3] = (8192.0, 5.0e6, 5.0e6)         # d = 4.88e-4
    b[20, :3] = (below(16384.0), 6.0e6, 6.0e6); b[21, :3] = (16384.0, 6.0e6, 6.0e6)       # d = 9.77e-4: 21 is no candidate
    b[30, :3] = (7.0e6, below(16384.0), 7.0e6); b[31, :3] = (7.0e6, 16384.0, 7.0e6)       # same in y
    b[40, :3] = (100.0, 100.0, 100.0);          b[41, :3] = (100.0 + 7.63e-6, 100.0, 100.0)  # r^2 = 5.8e-11 -> skipped
    b[50, :3] = (-3000.0, 2.0e6, 2.0e6);        b[51, :3] = (-3000.0 - 2.4414e-4, 2.0e6, 2.0e6)  # negative side, counted
    b = oracle.round_inputs_to_f32(b)
    ref = oracle.brute_force_seq(b)
    S = oracle.force_magnitude_sums(b)
    for v, name in _fast_variants(nbx) + [(-1, "default")]:
        with nbx.Context(n, dim) as c:
            c.upload(b)
            c.set_tuning(0, v)
            c.compute_accel()
            f = c.forces(oracle.G)
        assert_force_parity(f, ref, S, f"boundary pairs, {name}")
        for i in (10, 11, 20, 21, 30, 31, 50, 51):       # dominated by the partner: compare directly
            assert np.allclose(f[i], ref[i], rtol=2e-5, atol=0), (name, i)
        assert np.allclose(f[40], ref[40], rtol=1e-4) and np.allclose(f[41], ref[41], rtol=1e-4)


def test_device_side_accuracy_metric(nbx, oracle):
    """nbx_ctx_accuracy = the reference's compute_accuracy (utils.h:170-219) without copying the forces back.
    With the reference's G the metric is close to vacuous -- force components are ~1e-25..1e-21, below its
    ACCURACY_FORCE_THRESHOLD of 1e-20, so they are only held to |f| <= 1e-9 -- hence the second half with a
    coupling scaled by 1e10, where the 1 % rule actually bites."""
    n, dim = 5000, 3
    b = _oracle_inputs(oracle, 61, n, dim)
    ref = oracle.brute_force_seq(b)
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.compute_accel()
        f = c.forces(oracle.G)
        assert c.accuracy(ref, oracle.G) == 100.0 == oracle.compute_accuracy(f, ref)
        off = ref.copy()
        off[::7, 1] *= 1.02
        assert oracle.compute_accuracy(f, off) > 99.0, "vacuous at the reference's force magnitudes"
        assert c.accuracy(off, oracle.G) == oracle.compute_accuracy(f, off)
        scale = 1e10
        noisy = ref * scale
        noisy[::7, 1] *= 1.02                      # 2 % off in one component of every 7th body
        noisy[3, 0] = 1e-25                        # tiny reference component: absolute rule
        want = oracle.compute_accuracy(np.ascontiguousarray(f * scale), noisy)
        assert 80.0 < want < 90.0
        assert c.accuracy(noisy, oracle.G * scale) == want
    for r in range(2):                             # shards compare their own slice
        with nbx.Context(n, dim, n_shards=2, shard=r) as c:
            c.upload(b)
            c.compute_accel()
            lo = r * c.shard_len
            assert c.accuracy(ref[lo:lo + c.count], oracle.G) == 100.0


def test_close_set_mode_follows_the_bodies_during_a_run(nbx, oracle):
    """The refinement mode is chosen at upload and re-evaluated during a run from an asynchronous read-back of the device
    counters every 16 steps: a cloud that flies from outside the candidate region (all |coordinates| > 16384) through a
    compact state around the origin and out again goes candidate_pairs -> sorted_cells -> candidate_pairs, with forces
    matching the oracle in every state."""
    n, dim, dt = 20000, 3, 1.0
    rng = np.random.default_rng(9)
    b = oracle.generate(70, n, dim)
    p0 = rng.uniform(20000.0, 30000.0, size=(n, dim))
    target = rng.normal(scale=100.0, size=(n, dim))
    b[:, :dim] = p0
    b[:, dim:2 * dim] = (target - p0) / 20.0                    # ballistic: at the target after 20 steps, gone after ~40
    b = oracle.round_inputs_to_f32(b)
    G = oracle.G * 1e-12                                         # forces far too weak to bend the flight
    seen = []
    with nbx.Context(n, dim) as c:
        c.upload(b)
        assert c.close_set_mode()[0] == "candidate_pairs"
        for step in range(1, 57):
            c.compute_accel()
            if step in (1, 20, 56):
                cur = b.copy()
                c.download(cur)
                cr = oracle.round_inputs_to_f32(cur)
                assert_force_parity(c.forces(oracle.G), oracle.brute_force_seq(cr), oracle.force_magnitude_sums(cr), f"step {step}")
                c.compute_accel()
            c.kick_drift(dt, G)
            c.synchronize()                                      # lets the pending counter copy land before the next poll
            seen.append(c.close_set_mode())
    modes = [m for m, _, _ in seen]
    assert modes[10] == "candidate_pairs"                        # polled at step 16 at the earliest
    assert "sorted_cells" in modes[17:34], modes
    assert modes[-1] == "candidate_pairs", seen[-5:]
    assert max(cand for _, cand, _ in seen) == n                 # the poll at step 16 saw every body in the candidate set


def test_graph_replayed_run_through_a_mode_flip_equals_the_eager_loop(nbx, oracle):
    """The same fly-through (candidate_pairs -> sorted_cells -> candidate_pairs) driven through nbx_ctx_step, whose steps are
    graph replays in blocks of 16 with the close-set poll in between: ONE call of 56 steps, and seven calls of 8 steps with a
    synchronisation in between (so that every poll has landed and the next call re-captures the step for the new mode), both
    bit for bit equal to the eager loop of compute_accel + kick_drift.  The in-call re-capture drains the stream before it
    destroys the executable graph or reallocates (round 3's ADVICE); whether it triggers inside the one long call depends on
    the device keeping up with the host -- the results must not."""
    n, dim, dt, steps = 20000, 3, 1.0, 56
    rng = np.random.default_rng(9)
    b = oracle.generate(70, n, dim)
    p0 = rng.uniform(20000.0, 30000.0, size=(n, dim))
    target = rng.normal(scale=100.0, size=(n, dim))
    b[:, :dim] = p0
    b[:, dim:2 * dim] = (target - p0) / 20.0
    b = oracle.round_inputs_to_f32(b)
    G = oracle.G * 1e-12
    eager, one_call, short_calls = b.copy(), b.copy(), b.copy()
    with nbx.Context(n, dim) as c:
        c.upload(b)
        for _ in range(steps):
            c.compute_accel()
            c.kick_drift(dt, G)
            c.synchronize()
        c.download(eager)
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.step(dt, steps, G)
        c.download(one_call)
        after_one_call = c.close_set_mode()[0]
    modes = []
    with nbx.Context(n, dim) as c:
        c.upload(b)
        for _ in range(steps // 8):
            c.step(dt, 8, G)
            c.synchronize()
            modes.append(c.close_set_mode()[0])
        c.download(short_calls)
    assert np.array_equal(one_call, eager), "one long nbx_ctx_step call differs from the eager loop"
    assert np.array_equal(short_calls, eager), "short nbx_ctx_step calls differ from the eager loop"
    assert "sorted_cells" in modes and modes[-1] == "candidate_pairs", modes     # the re-capture between calls really happened
    assert after_one_call in ("candidate_pairs", "sorted_cells")


def test_fuzz_fast_path_against_guarded_kernel(nbx, oracle):
    """Randomised inputs over twelve decades of coordinate scale, offsets, clusters, duplicates, sub-threshold pairs, 1-4
    shards, D = 2 and 3: the default path (fast kernel + whichever close-set mode the library picks) against the oracle,
    and against the guarded kernel on the same device (which never takes the close-set pipeline)."""
    rng = np.random.default_rng(2026)
    exact = [i for i, v in enumerate(nbx.variants()) if "exact" in v][0]
    modes = set()
    import os
    for case in range(int(os.environ.get("NBX_FUZZ_CASES", "36"))):   # NBX_FUZZ_CASES=400 for a long soak
        dim = 3 if case % 3 else 2
        n = int(rng.integers(1500, 6000))
        scale = 10.0 ** rng.uniform(-2, 7)
        offset = rng.choice([0.0, 0.0, scale * 3.0, -scale * 10.0])
        b = oracle.generate(1000 + case, n, dim)
        kind = case % 4
        if kind == 0:
            b[:, :dim] = rng.uniform(-scale, scale, size=(n, dim)) + offset
        elif kind == 1:
            b[:, :dim] = rng.normal(scale=scale, size=(n, dim)) + offset
        elif kind == 2:
            centres = rng.uniform(-scale, scale, size=(5, dim))
            b[:, :dim] = centres[rng.integers(0, 5, n)] + rng.normal(scale=scale * 1e-3, size=(n, dim)) + offset
        else:
            b[:, :dim] = np.round(rng.uniform(-scale, scale, size=(n, dim)) / (scale / 8.0)) * (scale / 8.0) + offset   # lattice: many duplicates
        k = int(rng.integers(0, 6))                              # plant a few close pairs of every kind
        for j in range(k):
            i0, i1 = rng.integers(0, n, 2)
            if i0 == i1:
                continue
            b[i1, :dim] = b[i0, :dim]
            b[i1, int(rng.integers(0, dim))] += float(rng.choice([0.0, 3e-6, 2e-5, 3e-4, 2e-3]))
        b = oracle.round_inputs_to_f32(b)
        ref = oracle.brute_force_seq(b)
        S = oracle.force_magnitude_sums(b)
        G_ = int(rng.integers(1, 5))
        parts, parts_exact = [], []
        for r in range(G_):
            with nbx.Context(n, dim, n_shards=G_, shard=r) as c:
                c.upload(b)
                modes.add(c.close_set_mode()[0])
                if G_ == 1:
                    c.compute_accel()
                else:
                    c.compute_accel(nbx.SRC_LOCAL)
                    c.compute_accel(nbx.SRC_REMOTE)
                parts.append(c.forces(oracle.G))
                c.set_tuning(0, exact)
                c.compute_accel()
                parts_exact.append(c.forces(oracle.G))
        f, fe = np.concatenate(parts), np.concatenate(parts_exact)
        what = f"case {case}: D={dim} n={n} scale={scale:.1e} offset={offset:.1e} kind={kind} shards={G_}"
        assert_force_parity(fe, ref, S, what + " [guarded kernel]")
        assert_force_parity(f, ref, S, what + " [default path]")
    assert {"candidate_pairs", "sorted_cells"} <= modes, modes    # the fuzz really visited both refinement modes
