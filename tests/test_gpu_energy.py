"""GPU: the energy diagnostic (nbx_ctx_energy) against the oracle's energy, and a scaled-down version of
BASELINE config 5 (Plummer-sphere initial condition, many kick/drift steps, energy-drift check).
The reference has no time loop and no energy check: what is pinned here is (a) the
device energy equals the fp64 oracle energy of the same state, (b) kick->drift with the reference's
helpers is the symplectic Euler map, whose energy error is bounded and first order in dt."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim", (3, 2))
def test_energy_matches_oracle(nbx, oracle, dim):
    n = 3000
    b = oracle.round_inputs_to_f32(oracle.generate(12, n, dim))
    b[10, :dim] = b[11, :dim]                       # a coincident pair: excluded from the potential, like the force
    ke_ref, pe_ref = oracle.energy(b)
    with nbx.Context(n, dim) as c:
        c.upload(b)
        ke, pe = c.energy(oracle.G)
    assert abs(ke - ke_ref) <= 1e-13 * ke_ref
    assert abs(pe - pe_ref) <= 2e-6 * pe_ref
    # shards: the shares add up to the same totals
    tot = np.zeros(2)
    for r in range(3):
        with nbx.Context(n, dim, n_shards=3, shard=r) as c:
            c.upload(b)
            tot += c.energy(oracle.G)
    assert abs(tot[0] - ke_ref) <= 1e-13 * ke_ref and abs(tot[1] - pe_ref) <= 2e-6 * pe_ref


def _drift(nbx, bodies, G, dt, steps):
    n = bodies.shape[0]
    with nbx.Context(n, 3) as c:
        c.upload(bodies)
        e0 = sum(c.energy(G))
        worst = 0.0
        for _ in range(steps // 10):
            c.step(dt, 10, G)
            worst = max(worst, abs(sum(c.energy(G)) - e0) / abs(e0))
        ke, pe = c.energy(G)
    return worst, ke, pe, e0


def test_plummer_energy_drift(nbx):
    """Cold Plummer sphere under the reference's (repulsive, 1/r^3) law with G chosen so that the
    dynamical time is ~100 steps: the sphere expands, potential energy turns into kinetic energy, and the
    total must stay put up to the integrator's O(dt) error."""
    n = 16384
    b = nbx.plummer_bodies(n, 3, seed=3, a=1.0e5, total_mass=1.0)
    b[:, 3:6] = 0.0
    G = 1.0e16                                       # t_dyn = sqrt(a^4 / (G M)) = 100
    d1, ke, pe, e0 = _drift(nbx, b, G, 1.0, 200)
    assert e0 > 0 and ke > 0.2 * e0, "the run must convert a sizeable part of the potential energy"
    assert d1 < 8e-2, f"energy drift {d1:.3e} over 2 dynamical times (first-order map, dt = t_dyn/100)"
    d2, _, _, _ = _drift(nbx, b, G, 0.5, 400)
    print(f"\nPlummer N={n}: max |dE/E0| = {d1:.3e} (dt=1), {d2:.3e} (dt=0.5); KE/E0 at the end {ke / e0:.3f}")
    assert d2 < 0.7 * d1, f"first-order integrator: halving dt must shrink the energy error ({d1:.3e} -> {d2:.3e})"
