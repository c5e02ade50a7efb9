"""GPU: the softened pair law (nbx_ctx_set_softening) -- an EXTENSION: the reference's brute force is unsoftened
(SURVEY F4), so the checker is the oracle's own restatement of  a_i = sum_j m_j d/(r^2+eps^2)^2  in fp64 ("parity
unpinned": nothing in the reference to pin it to).  What it buys is in the last test: under the unsoftened law a
fixed-step kick/drift must resolve the closest pair (tests/test_gpu_config5.py); softened at the inter-particle scale
the Plummer sphere of BASELINE config 5 evolves for dynamical times with a small energy error."""
import numpy as np
import pytest

from oracle_lib import assert_force_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim", (3, 2))
@pytest.mark.parametrize("eps", (1.0e-3, 50.0, 2.0e5))
def test_softened_forces_match_the_checker(nbx, oracle, dim, eps):
    n = 6000
    b = oracle.generate(90 + dim, n, dim)
    b[11, :dim] = b[10, :dim]                                   # coincident pair: zero force between them, finite weights
    b[21, :dim] = b[20, :dim]; b[21, 0] += 3.0                   # a close pair, softened or not depending on eps
    b = oracle.round_inputs_to_f32(b)
    ref, S = oracle.force_rows_softened(b, eps)
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.set_softening(eps)
        assert c.effective_tuning()[0].startswith("fast")
        c.compute_accel()
        f = c.forces(oracle.G)
        assert_force_parity(f, ref, S, f"softened eps={eps} D={dim}")
        c.compute_accel()
        assert np.array_equal(f, c.forces(oracle.G))
        c.set_softening(0.0)                                    # back to the reference law
        c.compute_accel()
        assert_force_parity(c.forces(oracle.G), oracle.brute_force_seq(b), oracle.force_magnitude_sums(b), "eps=0 again")
    for r in range(3):                                          # sharded passes
        with nbx.Context(n, dim, n_shards=3, shard=r) as c:
            c.upload(b)
            c.set_softening(eps)
            c.compute_accel(nbx.SRC_LOCAL)
            c.compute_accel(nbx.SRC_REMOTE)
            lo = r * c.shard_len
            assert_force_parity(c.forces(oracle.G), ref[lo:lo + c.count], S[lo:lo + c.count], f"softened shard {r}")


def test_softened_energy_and_argument_checks(nbx, oracle):
    n, dim = 3000, 3
    b = oracle.round_inputs_to_f32(oracle.generate(12, n, dim))
    b[11, :3] = b[10, :3]
    for eps in (1.0e-2, 1.0e4):
        ke_ref, pe_ref = oracle.energy_softened(b, eps)
        with nbx.Context(n, dim) as c:
            c.upload(b)
            c.set_softening(eps)
            ke, pe = c.energy(oracle.G)
        assert abs(ke - ke_ref) <= 1e-13 * ke_ref and abs(pe - pe_ref) <= 2e-6 * pe_ref, (eps, pe, pe_ref)
        tot = np.zeros(2)
        for r in range(2):
            with nbx.Context(n, dim, n_shards=2, shard=r) as c:
                c.upload(b)
                c.set_softening(eps)
                tot += c.energy(oracle.G)
        assert abs(tot[1] - pe_ref) <= 2e-6 * pe_ref
    with nbx.Context(n, dim) as c:
        c.upload(b)
        for bad in (-1.0, 1e-9, float("nan"), 1e20):
            with pytest.raises(nbx.NbxError):
                c.set_softening(bad)
        c.set_softening(1e-6)                                   # m / eps^4 = 1e8 / 1e-24 = 1e32: representable
        c.compute_accel()
    heavy = b.copy()
    heavy[5, -1] = float(np.float32(3.0e15))                    # 3e15 / 1e-24 overflows fp32
    with nbx.Context(n, dim) as c:
        c.upload(heavy)
        c.set_softening(1e-6)
        with pytest.raises(nbx.NbxError):
            c.compute_accel()
        c.set_softening(1.0)
        c.compute_accel()
    # ... and too LARGE a softening length is refused as well: every fp32 pair weight m / eps^4 would underflow and the call
    # would return an all-zero field with status OK.  Masses <= 1e8: eps = 1e9 leaves 1e8 / 1e36 = 1e-28 (a normal fp32, just
    # inside the bound); eps = 1e10 -> 1e-32: refused.  Under the Newtonian law the weight is m / eps^3.
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.set_softening(1.0e9)
        c.compute_accel()
        f = c.forces(oracle.G)
        assert np.isfinite(f).all() and (np.abs(f).max(axis=1) > 0).all(), "forces at the accepted upper bound must not vanish"
        c.set_softening(1.0e10)
        with pytest.raises(nbx.NbxError) as e:
            c.compute_accel()
        assert e.value.status == 1 and "underflow" in str(e.value)
        c.set_law(nbx.FORCE_LAW_NEWTON)                         # m / eps^3 = 1e8 / 1e30: representable again
        c.compute_accel()
        assert (np.abs(c.forces(oracle.G)).max(axis=1) > 0).all()


def test_every_variant_with_softening_on(nbx, oracle):
    """A caller's variant choice plus a softening length never ends in an opaque HIP error: variants without a softened
    build (the exact and strict kernels, the A/B table entries) are replaced by the default fast kernel, for both laws."""
    n, dim, eps = 2500, 3, 40.0
    b = oracle.round_inputs_to_f32(oracle.generate(14, n, dim))
    ref, S = oracle.force_rows_softened(b, eps)
    refn, Sn = oracle.force_rows_softened(b, eps, newton=True)
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.set_softening(eps)
        for v, name in enumerate(nbx.variants()):
            c.set_tuning(0, v)
            assert c.effective_tuning()[0].startswith("fast"), (name, c.effective_tuning())
            c.set_law(nbx.FORCE_LAW_REFERENCE)
            c.compute_accel()
            assert_force_parity(c.forces(oracle.G), ref, S, f"softened, asked for {name}")
            c.set_law(nbx.FORCE_LAW_NEWTON)
            c.compute_accel()
            assert_force_parity(c.forces(oracle.G), refn, Sn, f"softened Newtonian, asked for {name}")


def test_softened_plummer_sphere_conserves_energy_over_dynamical_times(nbx):
    """BASELINE config 5's system at N = 262,144 with the law softened at the inter-particle scale: eps = 1500 (median
    nearest-neighbour distance of this sample ~3,100), G = 1e4 so that t_dyn = sqrt(a^4/(G M)) = 100, dt = 0.5,
    400 steps = 2 t_dyn.  Unsoftened, the same G and dt give |dE/E0| ~ 1e2 from the first kick of the closest pairs."""
    n = 1 << 18
    b = nbx.plummer_bodies(n, 3, seed=5, a=1.0e5, total_mass=1.0e12)
    G, dt, eps = 1.0e4, 0.5, 1500.0
    with nbx.Context(n, 3) as c:
        c.upload(b)
        c.set_softening(eps)
        e0 = sum(c.energy(G))
        worst = 0.0
        for _ in range(8):
            c.step(dt, 50, G)
            ke, pe = c.energy(G)
            worst = max(worst, abs(ke + pe - e0) / e0)
        ms, launches = c.kernel_time()
    print(f"\nsoftened Plummer N={n}: max |dE/E0| = {worst:.3e} over 400 steps (2 t_dyn), KE/E0 at the end {ke / e0:.3f}, {ms:.1f} ms per step")
    assert ke > 0.2 * e0, "two dynamical times must convert a sizeable part of the potential energy"
    assert worst < 2e-2


@pytest.mark.parametrize("dim", (3, 2))
def test_newtonian_law_matches_the_checker(nbx, oracle, dim):
    """nbx_ctx_set_law(NEWTON): attractive, softened  F_i = +G m_i sum_j m_j d/(r^2+eps^2)^(3/2)  (extension; checker
    restated in the oracle file, parity unpinned by construction)."""
    n, eps = 6000, 2.0e4
    b = oracle.generate(95 + dim, n, dim)
    b[11, :dim] = b[10, :dim]
    b = oracle.round_inputs_to_f32(b)
    ref, S = oracle.force_rows_softened(b, eps, newton=True)
    rep, _ = oracle.force_rows_softened(b, eps)
    with nbx.Context(n, dim) as c:
        c.upload(b)
        c.set_law(nbx.FORCE_LAW_NEWTON)
        with pytest.raises(nbx.NbxError):                      # no softening length yet
            c.compute_accel()
        c.set_softening(eps)
        c.compute_accel()
        f = c.forces(oracle.G)
        assert_force_parity(f, ref, S, f"newton D={dim}")
        assert (np.einsum("ij,ij->i", f, rep) < 0).mean() > 0.8     # attractive: mostly opposite to the repulsive law's force
        ke_ref, pe_ref = oracle.energy_softened(b, eps, newton=True)
        ke, pe = c.energy(oracle.G)
        assert pe < 0 and abs(pe - pe_ref) <= 2e-6 * abs(pe_ref) and abs(ke - ke_ref) <= 1e-13 * ke_ref
        # the device-side accuracy metric follows the law's sign as well (scaled so that the 1 % rule bites)
        scale = 1e12
        good = c.accuracy(np.ascontiguousarray(ref * scale), oracle.G * scale)     # per-COMPONENT 1 % rule: a few cancelling
        assert good >= 99.9 and good == oracle.compute_accuracy(np.ascontiguousarray(f * scale), np.ascontiguousarray(ref * scale))
        assert c.accuracy(np.ascontiguousarray(-ref * scale), oracle.G * scale) < 5.0  # components miss it in fp32
        c.set_law(nbx.FORCE_LAW_REFERENCE)
        c.compute_accel()
        assert_force_parity(c.forces(oracle.G), rep, oracle.force_rows_softened(b, eps)[1], "back to the reference law, softened")
    with nbx.Node(n, dim, [0, 0, 0]) as node:                   # three virtual ranks: LOCAL + REMOTE passes, kick/drift sign
        node.upload(b)
        node.set_softening(eps)
        node.set_law(nbx.FORCE_LAW_NEWTON)
        assert_force_parity(node.forces(oracle.G), ref, S, f"newton, node D={dim}")
        Gs = oracle.G * 1e22
        node.step(2.0, 1, Gs)
        got = b.copy()
        node.download(got)
    cur = b.copy()
    oracle.update_body_velocities(cur, np.ascontiguousarray(ref * 1e22), 2.0)
    oracle.update_body_positions(cur, 2.0)
    dv, dv_ref = got[:, dim:2 * dim] - b[:, dim:2 * dim], cur[:, dim:2 * dim] - b[:, dim:2 * dim]
    assert np.abs(dv_ref).max() > 1e-6 and np.allclose(dv, dv_ref, rtol=1e-4, atol=4e-6 * np.abs(dv_ref).max())


def test_newtonian_plummer_sphere_stays_in_equilibrium(nbx):
    """The physically meaningful form of BASELINE config 5 (`--law newton` "with Plummer softening"): a Plummer
    sphere with velocities from its own distribution function, under the attractive softened Newtonian law, is a
    stationary solution.  N = 262,144, a = 1e5, M = 1e12, G = 0.1 (t_dyn = sqrt(a^3/(G M)) = 100), eps = 1500, dt = 0.5,
    400 steps = 2 t_dyn: energy conserved, virial ratio 2K/|U| stays near 1, the half-mass radius stays put."""
    n = 1 << 18
    G, dt, eps, a = 0.1, 0.5, 1500.0, 1.0e5
    b = nbx.plummer_bodies(n, 3, seed=5, a=a, total_mass=1.0e12, G=G)
    r_half0 = np.median(np.linalg.norm(b[:, :3] - 5.0e6, axis=1))
    with nbx.Context(n, 3) as c:
        c.upload(b)
        c.set_softening(eps)
        c.set_law(nbx.FORCE_LAW_NEWTON)
        ke0, pe0 = c.energy(G)
        e0 = ke0 + pe0
        worst, virial = 0.0, [2 * ke0 / abs(pe0)]
        for _ in range(8):
            c.step(dt, 50, G)
            ke, pe = c.energy(G)
            worst = max(worst, abs(ke + pe - e0) / abs(e0))
            virial.append(2 * ke / abs(pe))
        cur = b.copy()
        c.download(cur)
    r_half1 = np.median(np.linalg.norm(cur[:, :3] - 5.0e6, axis=1))
    print(f"\nNewtonian Plummer N={n}: E0 = {e0:.4e} (bound), max |dE/E0| = {worst:.3e} over 2 t_dyn, "
          f"virial 2K/|U| {min(virial):.3f}..{max(virial):.3f}, half-mass radius {r_half0:.0f} -> {r_half1:.0f}")
    assert e0 < 0 and worst < 2e-3
    assert 0.9 < min(virial) and max(virial) < 1.1
    assert abs(r_half1 - r_half0) < 0.05 * r_half0


@pytest.mark.parametrize("dim", (3, 2))
def test_kick_drift_kick_matches_the_helpers_composed_that_way(nbx, oracle, dim):
    """nbx_ctx_step_kdk / nbx_node_step_kdk: the reference's two helpers (methods.cpp:425-450) as a synchronised
    kick-drift-kick leapfrog -- against the oracle's helpers composed the same way, with a coupling strong enough to bend
    the orbits; the pure-kick call leaves accelerations valid."""
    n, steps, dt = 600, 5, 2.0
    gs = 1e24
    b0 = oracle.round_inputs_to_f32(oracle.generate(35 + dim, n, dim))
    ref = b0.copy()
    for _ in range(steps):
        f = oracle.brute_force_seq(oracle.round_inputs_to_f32(ref)) * gs
        oracle.update_body_velocities(ref, np.ascontiguousarray(f), dt / 2)
        oracle.update_body_positions(ref, dt)
        f = oracle.brute_force_seq(oracle.round_inputs_to_f32(ref)) * gs
        oracle.update_body_velocities(ref, np.ascontiguousarray(f), dt / 2)
    dv = np.abs(ref[:, dim:2 * dim] - b0[:, dim:2 * dim]).max()
    assert dv > 1e-3
    got = b0.copy()
    with nbx.Context(n, dim) as c:
        c.upload(b0)
        c.step_kdk(dt, steps, oracle.G * gs)
        f_end = c.forces(oracle.G)                              # still valid: the last call was a pure kick
        c.download(got)
    assert np.allclose(got[:, dim:2 * dim], ref[:, dim:2 * dim], rtol=0, atol=2e-5 * dv)
    assert np.allclose(got[:, :dim], ref[:, :dim], rtol=1e-9, atol=0)
    cr = oracle.round_inputs_to_f32(got)
    assert_force_parity(f_end, oracle.brute_force_seq(cr), oracle.force_magnitude_sums(cr), "forces after the closing half-kick")
    many = b0.copy()
    with nbx.Node(n, dim, [0, 0, 0]) as node:
        node.upload(b0)
        node.step_kdk(dt, steps, oracle.G * gs)
        node.download(many)
    assert np.allclose(many[:, dim:2 * dim], got[:, dim:2 * dim], rtol=0, atol=2e-5 * dv)
    assert np.allclose(many[:, :dim], got[:, :dim], rtol=1e-9, atol=0)


def test_kick_drift_kick_is_second_order(nbx):
    """Softened Plummer sphere (repulsive reference law, eps at the inter-particle scale), one dynamical time: halving dt
    cuts the kick-drift-kick energy error ~4x and leaves it far below the kick-drift (first-order) error."""
    n = 32768
    b = nbx.plummer_bodies(n, 3, seed=7, a=1.0e5, total_mass=1.0e12)
    G, eps = 1.0e4, 3000.0

    def drift(scheme, dt, steps):
        with nbx.Context(n, 3) as c:
            c.upload(b)
            c.set_softening(eps)
            e0 = sum(c.energy(G))
            getattr(c, scheme)(dt, steps, G)
            return abs(sum(c.energy(G)) - e0) / e0

    kd = drift("step", 2.0, 50)
    k1, k2 = drift("step_kdk", 2.0, 50), drift("step_kdk", 1.0, 100)
    print(f"\n|dE/E0| after 1 t_dyn: kick-drift dt=2: {kd:.3e}; kick-drift-kick dt=2: {k1:.3e}, dt=1: {k2:.3e} (ratio {k1 / k2:.2f})")
    assert k1 < 0.2 * kd and 2.5 < k1 / k2 < 6.0
