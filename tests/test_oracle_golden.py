"""CPU: the oracle (oracle/nbody_oracle.c) against the committed golden vectors, which are outputs of
the reference's own object code (tests/golden/make_golden.py).  Integer-exact comparisons: the
restatement follows the reference operation for operation, so fp64 results must be bit-identical
for seq, omp_2, kick, drift and the generator; omp_1 depends on the OpenMP partition and is held
to re-association noise."""
import numpy as np
import pytest

from conftest import golden

CASES = [(d, n) for d in (2, 3) for n in (2, 3, 64, 1024)]


@pytest.mark.parametrize("dim,n", CASES)
def test_generator_matches_reference(oracle, dim, n):
    g = golden(f"bf_D{dim}_N{n}.npz")
    b = oracle.generate(int(g["seed"]), n, dim)
    assert np.array_equal(b, g["bodies"])  # mt19937 + libstdc++ uniform_real_distribution, bit for bit
    pos, vel, m = b[:, :dim], b[:, dim:2 * dim], b[:, -1]
    assert pos.min() >= 1 and pos.max() < 1e7 and vel.min() >= -10 and vel.max() < 10 and m.min() >= 1 and m.max() < 1e8


@pytest.mark.parametrize("dim,n", CASES)
def test_seq_and_omp2_bit_exact(oracle, dim, n):
    g = golden(f"bf_D{dim}_N{n}.npz")
    assert float(g["G"]) == oracle.G == 4.471e-21
    for key_b, suffix in (("bodies", ""), ("bodies_f32", "_f32")):
        b = np.ascontiguousarray(g[key_b])
        assert np.array_equal(oracle.brute_force_seq(b), g["forces_seq" + suffix])
        assert np.array_equal(oracle.brute_force_omp_2(b), g["forces_omp_2" + suffix])
        rows = np.arange(n, dtype=np.int64)[::-1]
        assert np.array_equal(oracle.force_rows_omp_2(b, rows), g["forces_omp_2" + suffix][::-1])


@pytest.mark.parametrize("dim,n", CASES)
def test_omp1_matches_to_reassociation_noise(oracle, dim, n):
    g = golden(f"bf_D{dim}_N{n}.npz")
    f = oracle.brute_force_omp_1(np.ascontiguousarray(g["bodies"]))
    ref = g["forces_omp_1"]
    scale = oracle.force_magnitude_sums(np.ascontiguousarray(g["bodies"]))
    err = np.abs(f - ref).max(axis=1)
    assert (err <= 1e-13 * np.maximum(scale, 1e-300)).all()


@pytest.mark.parametrize("dim", (2, 3))
def test_trajectory_bit_exact(oracle, dim):
    g = golden(f"traj_D{dim}_N64.npz")
    dt, steps = float(g["dt"]), int(g["steps"])
    cur = np.ascontiguousarray(g["states"][0]).copy()
    for s in range(steps):
        f = oracle.brute_force_seq(cur)
        assert np.array_equal(f, g["forces"][s])
        oracle.update_body_velocities(cur, f, dt)
        oracle.update_body_positions(cur, dt)
        assert np.array_equal(cur, g["states"][s + 1])
    # the composed loop helper does the same thing
    again = np.ascontiguousarray(g["states"][0]).copy()
    oracle.leapfrog(again, dt, steps, variant=0)
    assert np.array_equal(again, g["states"][steps])


def test_known_answers(oracle):
    k = golden("kat.npz")
    G = oracle.G
    for name in ("two", "coincident", "near_skip", "near_keep"):
        b = np.ascontiguousarray(k[name + "_bodies"])
        assert np.array_equal(oracle.brute_force_seq(b), k[name + "_forces_seq"])
        assert np.array_equal(oracle.brute_force_omp_2(b), k[name + "_forces_omp_2"])
    # two unit masses 2 apart: F0 = (-G/8, 0, 0), F1 = (+G/8, 0, 0)  (repulsive, 1/r^3)
    f = k["two_forces_seq"]
    assert f[0, 0] == -G / 8 and f[1, 0] == G / 8 and not f[:, 1:].any()
    assert abs(f[0, 0] - (-5.58875e-22)) < 1e-27
    # coincident bodies exert nothing on each other; both feel only the third body
    c = k["coincident_forces_seq"]
    assert c[0, 0] == -G * 3.0 * 2.0 and c[1, 0] == -G * 7.0 * 2.0 and abs(c[2, 0] - G * 2.0 * 10.0) < 1e-33
    # r^2 < 1e-10 is skipped, r^2 >= 1e-10 is counted (methods.cpp:24)
    assert not k["near_skip_forces_seq"].any()
    assert k["near_keep_forces_seq"][0, 0] < 0 < k["near_keep_forces_seq"][1, 0]
    # kick then drift with dt = 1: v0.x = x0.x = -G/8
    s = np.ascontiguousarray(k["two_bodies"]).copy()
    ff = oracle.brute_force_seq(s)
    oracle.update_body_velocities(s, ff, 1.0)
    oracle.update_body_positions(s, 1.0)
    assert np.array_equal(s, k["two_after_step_dt1"])
    assert s[0, 3] == -G / 8 and s[0, 0] == -G / 8


def test_empty_and_single(oracle):
    for dim in (2, 3):
        e = np.zeros((0, 2 * dim + 1))
        assert oracle.brute_force_seq(e).shape == (0, dim)
        one = oracle.generate(7, 1, dim)
        assert not oracle.brute_force_seq(one).any() and not oracle.brute_force_omp_2(one).any()


def test_accuracy_metric(oracle):
    # utils.h:170-219: % of bodies with every component within 1 %; tiny reference components use 1e-9 absolute
    ref = np.array([[1.0, 2.0, 3.0], [1e-30, 1.0, 1.0], [1.0, 1.0, 1.0], [2.0, 2.0, 2.0]])
    f = ref.copy()
    f[0, 0] *= 1.009   # inside 1 %
    f[2, 1] *= 1.011   # outside 1 %
    f[1, 0] = 1e-10    # |ref| < 1e-20 and |f| <= 1e-9: fine
    assert oracle.compute_accuracy(f, ref) == 75.0
    f[1, 0] = 1e-8     # |f| > 1e-9: inaccurate
    assert oracle.compute_accuracy(f, ref) == 50.0


def test_energy_matches_force_law(oracle):
    # F = -grad U for U = sum G mi mj / (2 r^2): central difference on one coordinate
    b = oracle.generate(3, 16, 3)
    b[:, :3] = b[:, :3] / 1e6  # shrink the box so the potential is not negligible against rounding
    f = oracle.brute_force_seq(b)
    h = 1e-6
    for i, k in ((0, 0), (5, 2)):
        bp, bm = b.copy(), b.copy()
        bp[i, k] += h
        bm[i, k] -= h
        dU = (oracle.energy(bp)[1] - oracle.energy(bm)[1]) / (2 * h)
        assert abs(-dU - f[i, k]) <= 1e-6 * abs(f[i, k])


def test_oracle_under_sanitizers():
    """make -C oracle asan-check: every oracle entry point on small/degenerate inputs under ASan + UBSan."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "asan-check"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "oracle selftest ok" in p.stdout, p.stdout + p.stderr
