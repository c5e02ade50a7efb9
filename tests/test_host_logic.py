"""CPU: host-side logic of the package that needs no GPU: the numpy body generator reproduces the
reference generator's stream (= the oracle's = libstdc++'s), Plummer sampling is sane, the ctypes
layer validates shapes."""
import os

import numpy as np
import pytest

from conftest import ROOT, golden


@pytest.mark.parametrize("dim,n", [(2, 64), (3, 1024), (3, 3)])
def test_uniform_bodies_match_reference_stream(nbx, oracle, dim, n):
    g = golden(f"bf_D{dim}_N{n}.npz")
    b = nbx.uniform_bodies(n, dim, int(g["seed"]))
    assert np.array_equal(b, g["bodies"])
    assert np.array_equal(nbx.uniform_bodies(5000, dim, 99), oracle.generate(99, 5000, dim))


def test_plummer_bodies(nbx):
    b = nbx.plummer_bodies(20000, 3, seed=4, a=1e5)
    r = np.sqrt(((b[:, :3] - 5e6) ** 2).sum(axis=1))
    assert r.max() < 1e6 and 0.6e5 < np.median(r) < 2.0e5      # half-mass radius of a Plummer sphere = 1.305 a
    assert np.allclose(b[:, 6], 1e12 / 20000)
    assert abs(b[:, 3:6].mean()) < 0.05 * np.abs(b[:, 3:6]).max()
    assert np.abs(b[:, :3]).min() > 8192, "stays clear of the close-set region"


def test_body_stride_and_shapes(nbx):
    assert nbx.body_stride(3) == 7 and nbx.body_stride(2) == 5
    with pytest.raises(ValueError):
        nbx.uniform_bodies(4, 4)


def test_aggregate_keeps_reference_columns_and_adds_gpu_counts(tmp_path):
    """tools/aggregate_results.py on synthetic sweep output (no GPU needed): the main table keeps the reference's four
    columns (analysis/aggregated_results.csv:1) with BruteForce_HIP_x<G> rows; the sidecar table carries GPUs, distinct
    devices and the kernel speed-up over the one-GPU row."""
    import csv
    import subprocess
    res = tmp_path / "results"
    res.mkdir()
    runs = [("01012026_000001", 1000000, "BruteForce_HIP", 1, 1, 0.30, 0.21), ("01012026_000002", 1000000, "BruteForce_HIP", 1, 1, 0.32, 0.23),
            ("01012026_000003", 1000000, "BruteForce_HIP_x2", 2, 2, 0.20, 0.11), ("01012026_000004", 1000000, "BruteForce_HIP_x8", 8, 1, 0.31, 0.22)]
    for rid, n, label, gpus, distinct, t, k in runs:
        (res / f"run_{rid}_N_{n}_3D.csv").write_text(f"Method,Bodies,Dimension,Time(s)\n{label},{n},3,{t:.6f}\n")
        (res / f"run_{rid}_N_{n}_3D_hip.csv").write_text(
            "Method,Bodies,Dimension,Time(s),KernelTime(s),PairInteractionsPerSec,GPUs,DistinctDevices\n"
            f"{label},{n},3,{t:.6f},{k:.6f},{n * n / k:.6e},{gpus},{distinct}\n")
    p = subprocess.run(["python3", os.path.join(ROOT, "tools", "aggregate_results.py"), str(res)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    agg = list(csv.DictReader(open(res / "aggregated_results.csv")))
    assert list(agg[0].keys()) == ["Bodies", "Method", "Dimension", "Average Runtime (s)"]
    by = {r["Method"]: float(r["Average Runtime (s)"]) for r in agg}
    assert by["BruteForce_HIP"] == pytest.approx(0.31) and set(by) == {"BruteForce_HIP", "BruteForce_HIP_x2", "BruteForce_HIP_x8"}
    hip = {(int(r["GPUs"]), int(r["Distinct Devices"])): r for r in csv.DictReader(open(res / "aggregated_hip.csv"))}
    assert hip[(1, 1)]["Runs"] == "2" and float(hip[(2, 2)]["Kernel Speed-up vs 1 GPU"]) == pytest.approx(0.22 / 0.11)
    # eight virtual ranks on ONE device time-slice it: no per-GPU kernel figures for that row, only the whole-call runtime
    v = hip[(8, 1)]
    assert v["Virtual Ranks"] == "yes" and v["Kernel Speed-up vs 1 GPU"] == "" and v["Pair Interactions/s (kernel)"] == ""
    assert float(v["Average Runtime (s)"]) == pytest.approx(0.31) and hip[(2, 2)]["Virtual Ranks"] == "no"


def test_shell_tools_parse_and_the_sweep_sizes_its_openmp_teams(tmp_path):
    """tools/*.sh are syntactically valid, and run_sweep.sh does not leave the OpenMP team size to the number of visible
    hardware threads (a container may grant far fewer CPUs than it shows: 0.2 ms rows then read 100-200 ms of throttling)."""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scripts = sorted(glob.glob(os.path.join(root, "tools", "*.sh")))
    assert scripts
    for sh in scripts:
        assert subprocess.run(["bash", "-n", sh]).returncode == 0, sh
    # run the script's prologue only: a fake harness that reports no device and succeeds, no sizes to sweep
    fake = tmp_path / "nbody_sim"
    fake.write_text("#!/bin/sh\nif [ \"$1\" = --device-count ]; then echo 0; fi\nexit 0\n")
    fake.chmod(0o755)
    text = open(os.path.join(root, "tools", "run_sweep.sh")).read().replace('exe="$root/nbody_sim"', f'exe="{fake}"')
    text = text.replace('python3 "$root/tools/aggregate_results.py" results', "true")
    script = tmp_path / "run_sweep.sh"
    script.write_text(text)
    env = {k: v for k, v in os.environ.items() if k != "OMP_NUM_THREADS"}
    env.update(SIZES="1000", GPU_COUNTS="1", DIMS="3", ACC_SIZES="")
    p = subprocess.run(["bash", str(script)], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    line = [l for l in p.stdout.splitlines() if l.startswith("OpenMP rows use OMP_NUM_THREADS=")]
    assert line and 1 <= int(line[0].split("=")[1]) <= (os.cpu_count() or 1)
    p = subprocess.run(["bash", str(script)], cwd=tmp_path, env=dict(env, OMP_NUM_THREADS="3"), capture_output=True, text=True, timeout=60)
    assert "OMP_NUM_THREADS=3" in p.stdout                       # a caller's choice is kept
