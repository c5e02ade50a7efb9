"""GPU: BASELINE config 5 at its own size on one MI355X: N = 4,194,304 bodies, Plummer-sphere initial condition,
device-resident kick/drift steps, energy-drift check -- through the C ABI context and through the C++ harness.
(The full config -- 1000 steps on 8 GPUs -- is tools/run_config5.sh; a step costs ~3.7 s on one GPU, so the
test takes 10.)  The reference has no time loop, no Plummer generator and no energy check:
the deliverable is the build's own, stated here in full.

  law        the reference's brute-force law exactly as written (methods.cpp:21-37): repulsive,
             F_i = -G m_i sum_j m_j (p_j - p_i)/r^4, pairs with r^2 < 1e-10 skipped; conserved energy
             E = sum_i m_i v_i^2/2 + sum_{i<j} G m_i m_j/(2 r_ij^2)   (nbx_ctx_energy)
  bodies     Plummer sphere, scale radius a = 1e5, total mass M = 1e12 in N equal masses, radii cut at 10 a,
             centred at (5e6, 5e6, 5e6) (middle of the reference's box), seed 5; velocities drawn from the
             Plummer distribution function for the reference's G = 4.471e-21 (~3e-7: effectively a cold start)
  coupling   G = 0.05, dt = 0.5.  The law is singular and UNSOFTENED, so a fixed-step kick/drift has to resolve the
             closest pair of the sample, not the sphere: the first kick gives a pair at separation r the velocity
             G m dt / r^3 (exact answer: ~sqrt(G m)/r), i.e. an energy error  dE = sum_i m (G m dt / r_nn,i^3)^2 / 2
             over nearest-neighbour distances.  This sample's closest pair sits at r = 4.97 (mean spacing 1230); with
             G = 1e4 (t_dyn = sqrt(a^4/(G M)) = 100, dt = t_dyn/200) the measured energy error after 10 steps was
             150 E0 and equals that closed form to four digits (2.580e19 vs 2.578e19).  dE/E0 scales as G dt^2:
             G = 0.05 puts the first-kick error at 7.5e-4 E0.  t_dyn = 4.5e4, 1000 steps = 0.011 t_dyn.
             The test checks both: the drift bound, and that the measured error IS the first-kick closed form
             (nearest neighbours from a k-d tree on the host).  A physically resolved run of several t_dyn needs the
             softened law (nbx_ctx_set_softening; an extension, the reference's brute force has none -- SURVEY F4).
"""
import os
import re
import subprocess
import time

import numpy as np
import pytest

from oracle_lib import assert_force_parity, assert_plain_relative

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N = 1 << 22
A, M, SEED, G5, DT, STEPS = 1.0e5, 1.0e12, 5, 0.05, 0.5, 10


def test_config5_context_n4194304(nbx, oracle):
    b = nbx.plummer_bodies(N, 3, seed=SEED, a=A, total_mass=M)
    b = oracle.round_inputs_to_f32(b)
    rows = np.unique(np.random.default_rng(SEED).integers(0, N, 1100))
    assert rows.size >= 1024
    with nbx.Context(N, 3) as c:
        c.upload(b)
        assert c.effective_tuning()[0].startswith("fast"), c.effective_tuning()
        # sampled-row parity on the initial state (>= 1,024 rows) -- scale-free, so the oracle's G serves
        c.compute_accel()
        f = c.forces(oracle.G)[rows]
        ms, _ = c.kernel_time()
        ref = oracle.force_rows_omp_2(b, rows)
        e = assert_force_parity(f, ref, oracle.force_magnitude_sums(b, rows), "config 5 initial state", n_sources=N)
        worst = float((np.linalg.norm(f - ref, axis=1) / np.linalg.norm(ref, axis=1)).max())
        # energy before / after STEPS device-resident steps
        ke0, pe0 = c.energy(G5)
        t0 = time.perf_counter()
        c.step(DT, STEPS, G5)
        c.synchronize()
        wall = time.perf_counter() - t0
        ke1, pe1 = c.energy(G5)
        step_ms, _ = c.kernel_time()
        cur = b.copy()
        c.download(cur)
    e0, e1 = ke0 + pe0, ke1 + pe1
    drift = abs(e1 - e0) / abs(e0)
    print(f"\nconfig 5, N={N}: force kernel {ms:.0f} ms ({N * N / ms * 1e3:.3e} pair-interactions/s), rows {rows.size}: "
          f"max |da|/|a| {worst:.2e}, {e}\n  {STEPS} steps in {wall:.1f} s ({wall / STEPS:.2f} s/step, whole-step device time {step_ms:.0f} ms)\n"
          f"  E0 = {e0:.9e} (KE {ke0:.3e}, PE {pe0:.6e})  E{STEPS} = {e1:.9e} (KE {ke1:.6e}, PE {pe1:.6e})  |dE/E0| = {drift:.3e}")
    assert pe0 > 0 and ke0 < 1e-6 * pe0, "cold start: the energy is potential"
    assert ke1 > 1e-9 * e0 and pe1 < pe0, "the sphere must have started to expand (potential -> kinetic)"
    assert drift < 2e-3, f"energy drift {drift:.3e} after {STEPS} steps"
    # the error is the first kick of the closest pairs, in closed form (see the module docstring)
    from scipy.spatial import cKDTree
    p32 = b[:, :3]
    nn = cKDTree(p32).query(p32, k=2, workers=-1)[0][:, 1]
    m = b[0, -1]
    predicted = float((0.5 * m * (G5 * m * DT / nn ** 3) ** 2).sum())
    print(f"  closest pair r = {nn.min():.3f}, median nearest neighbour {np.median(nn):.0f}; first-kick closed form dE = {predicted:.4e}, "
          f"measured E{STEPS} - E0 = {e1 - e0:.4e}")
    # (at G = 0.05 the closest pair's kick, 48, is only ~2x its exact escape speed, so the potential it legitimately
    #  releases is ~10 % of the closed form, and later steps add a little: a band, not four digits, at this coupling)
    assert 0.5 * predicted - 2e-6 * e0 <= e1 - e0 <= 1.5 * predicted + 2e-6 * e0   # 2e-6 E0: fp32 noise floor of the potential sum
    assert np.isfinite(cur).all() and np.array_equal(cur[:, -1], b[:, -1])
    r0 = np.linalg.norm(b[:, :3] - 5.0e6, axis=1).mean()
    r1 = np.linalg.norm(cur[:, :3] - 5.0e6, axis=1).mean()
    assert r1 > r0, "repulsive law: the mean radius grows"


def test_config5_through_the_harness(tmp_path, nbx):
    """The same configuration through `nbody_sim --init plummer --energy-every` (C++ host side, HipSimulation<3>)."""
    exe = os.path.join(ROOT, "nbody_sim")
    assert os.path.exists(exe)
    p = subprocess.run([exe, "-N", str(N), "-m", "g", "--init", "plummer", "--seed", str(SEED), "--G", str(G5), "--dt", str(DT),
                        "--steps", str(STEPS), "--energy-every", "5"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "Error executing" not in p.stderr, p.stderr[-2000:]
    drifts = [float(x) for x in re.findall(r"\|dE/E0\| = ([0-9.eE+-]+)", p.stdout)]
    assert len(drifts) == 2 and max(drifts) < 2e-3, p.stdout[-2000:]
    m = re.search(r"Kernel time: ([0-9.eE+-]+) s  \(([0-9.eE+-]+) pair-interactions/s", p.stdout)
    assert m and float(m.group(2)) > 3.0e12, p.stdout[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("step ") or "Kernel time" in l or "Time taken" in l]
    print("\n" + "\n".join(lines))
