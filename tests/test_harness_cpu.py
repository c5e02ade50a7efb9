"""CPU: the C++ host side (nbody-simulation-parallel_amd/host): the harness keeps the reference's
CLI / file naming / CSV schema (nbody-sim-new/main.cpp:41-43, 59-63, 159-170, 885-928), its CPU rows
reproduce the oracle bit for bit, and the HIP row fails loudly (logged, row skipped -- the reference's
safely_execute contract, utils.h:95-103) when no GPU is present."""
import glob
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "nbody_sim")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "nbody_sim"], cwd=ROOT, stdout=subprocess.DEVNULL)
    return EXE


def _run(exe, tmp_path, *args):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    p = subprocess.run([exe, *args], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    return p


@pytest.mark.parametrize("dim", (2, 3))
def test_cli_files_csv_and_cpu_rows(exe, tmp_path, oracle, dim):
    n, seed = 300, 3
    p = _run(exe, tmp_path, "-N", str(n), "-d", str(dim), "-a", "1", "--seed", str(seed), "--dump", "d")
    assert p.returncode == 0, p.stderr
    csvs = glob.glob(os.path.join(tmp_path, "results", "run_*_N_%d_%dD.csv" % (n, dim)))
    assert len(csvs) == 1 and re.search(r"run_\d{8}_\d{6}_N_%d_%dD\.csv$" % (n, dim), csvs[0])
    assert os.path.exists(csvs[0][:-4] + ".out")
    lines = open(csvs[0]).read().strip().splitlines()
    assert lines[0] == "Method,Bodies,Dimension,Time(s),Accuracy(%)"
    rows = [l.split(",") for l in lines[1:]]
    names = [r[0] for r in rows]
    assert names[:3] == ["BruteForce_Sequential", "BruteForce_OpenMP1", "BruteForce_OpenMP2"]
    for r in rows:
        assert r[1] == str(n) and r[2] == str(dim) and re.fullmatch(r"\d+\.\d{6}", r[3]) and r[4] == "100.00"
    # bodies follow the reference generator's stream (seeded), forces equal the oracle bit for bit
    w = 2 * dim + 1
    bodies = np.fromfile(os.path.join(tmp_path, "d_bodies.f64")).reshape(n, w)
    assert np.array_equal(bodies, oracle.generate(seed, n, dim))
    f_seq = np.fromfile(os.path.join(tmp_path, "d_BruteForce_Sequential.f64")).reshape(n, dim)
    f_o2 = np.fromfile(os.path.join(tmp_path, "d_BruteForce_OpenMP2.f64")).reshape(n, dim)
    f_o1 = np.fromfile(os.path.join(tmp_path, "d_BruteForce_OpenMP1.f64")).reshape(n, dim)
    assert np.array_equal(f_seq, oracle.brute_force_seq(bodies))
    assert np.array_equal(f_o2, oracle.brute_force_omp_2(bodies))
    assert np.allclose(f_o1, f_seq, rtol=1e-9, atol=0)
    if "BruteForce_HIP" not in names:  # no GPU here: the failure must be loud and logged
        assert "Error executing BruteForce_HIP" in p.stderr and "no CPU fallback" in p.stderr
        assert "Error executing BruteForce_HIP" in open(csvs[0][:-4] + ".out").read()


def test_cli_validation_and_gates(exe, tmp_path):
    assert _run(exe, tmp_path, "-d", "4").returncode == 1
    assert _run(exe, tmp_path, "-N", "0").returncode == 1
    assert _run(exe, tmp_path, "-m", "x").returncode == 1
    h = _run(exe, tmp_path, "-h")
    assert h.returncode == 0 and "-N, --bodies" in h.stdout and "-m, --methods" in h.stdout
    # -m g: HIP only -- no CPU brute-force rows at all
    p = _run(exe, tmp_path, "-N", "64", "-m", "g", "--seed", "1")
    assert p.returncode == 0
    csv = sorted(glob.glob(os.path.join(tmp_path, "results", "run_*_N_64_3D.csv")))[-1]
    body = open(csv).read()
    assert body.startswith("Method,Bodies,Dimension,Time(s)\n") and "BruteForce_Sequential" not in body
    # tiny inputs do not trip the reference's n/3 modulo (utils.h:141 is undefined for n < 3)
    assert _run(exe, tmp_path, "-N", "2", "-m", "a", "--seed", "1").returncode == 0


def test_aggregate_results_has_the_analysis_columns(exe, tmp_path):
    """tools/aggregate_results.py: per-run CSVs -> the reference's aggregated_results.csv schema
    (nbody-sim-new/analysis/aggregated_results.csv:1)."""
    for seed in ("1", "2"):
        assert _run(exe, tmp_path, "-N", "200", "-d", "3", "-m", "a", "--seed", seed).returncode == 0
        import time
        time.sleep(1.1)  # run ids have one-second resolution
    p = subprocess.run(["python3", os.path.join(ROOT, "tools", "aggregate_results.py"), "results"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = open(os.path.join(tmp_path, "results", "aggregated_results.csv")).read().strip().splitlines()
    assert lines[0] == "Bodies,Method,Dimension,Average Runtime (s)"
    assert any(l.startswith("200,BruteForce_Sequential,3,") for l in lines[1:])


@pytest.mark.parametrize("dim", (2, 3))
def test_cpp_leaf_builder_equals_python_leaf_builder(exe, tmp_path, dim):
    """`-m p`: host/leaf_pairs_hip.cpp build_uniform_leaves<D> and leaves.py uniform_grid_leaves must produce the same
    CSR arrays (the GPU row itself fails loudly here: no device, no fallback)."""
    import nbody_amd as nbx
    n = 3000
    p = _run(exe, tmp_path, "-N", str(n), "-d", str(dim), "-m", "p", "--seed", "3", "--dump", "d")
    assert p.returncode == 0
    assert "Error executing NearField_HIP" in p.stderr and "no CPU fallback" in p.stderr
    bodies = np.fromfile(os.path.join(tmp_path, "d_bodies.f64")).reshape(n, 2 * dim + 1)
    depth = 1
    while depth < 10 and n / 2.0 ** (depth * dim) > 64.0:
        depth += 1
    want = nbx.leaves.uniform_grid_leaves(bodies, dim, depth)
    for name, w in zip(("leaf_offsets", "leaf_bodies", "list_offsets", "list_sources"), want):
        got = np.fromfile(os.path.join(tmp_path, f"d_{name}.u32"), dtype=np.uint32)
        assert np.array_equal(got, w), name
