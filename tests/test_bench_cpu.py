"""bench.py's record-keeping, checked without a GPU: the committed PMC profile resolves for the kernel the default run launches
(a hand-kept name table went stale and the driver's line carried traffic: null), and a plain
`python bench.py --gpus N` starts its ranks as a child process instead of exiting with a usage line (weak #6)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import nbody_amd as nbx  # noqa: E402


def _default_variant_name():
    lib = nbx.load_library()
    return lib.nbx_variant_name(lib.nbx_default_variant()).decode()


def test_library_reports_the_default_kernels_symbol_without_a_device():
    sym = bench.launched_kernel_symbol(_default_variant_name(), 3, 1.0e-5)
    assert sym.startswith("void nbx::(anonymous namespace)::accel_fast3l_kernel<3, ") and sym.endswith(">(nbx::KArgs)"), sym
    plain = bench.launched_kernel_symbol(_default_variant_name(), 3, 0.0)
    assert plain != sym and "accel_fast3l_kernel<3, " in plain
    assert "accel_fast3l_kernel<2, " in bench.launched_kernel_symbol(_default_variant_name(), 2, 1.0e-5)
    # a variant without a mixed-mode build runs its plain kernel whatever the tolerance
    assert "accel_f64_kernel<3, " in bench.launched_kernel_symbol("strict_f64_t4", 3, 1.0e-5)


def test_committed_pmc_profile_resolves_for_the_default_run():
    launched = bench.launched_kernel_symbol(_default_variant_name(), 3, 1.0e-5)
    found = None
    for rnd in bench.PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", rnd, "pmc_force_kernel.json")
        if not os.path.exists(path):
            continue
        t = bench.pmc_traffic(json.load(open(path)), launched)
        if t is not None:
            found = (rnd, t)
            break
    assert found, f"no profiles/<round>/pmc_force_kernel.json of '{launched}' in {bench.PROFILE_ROUNDS}"
    rnd, t = found
    algorithmic = 28.0 * (1 << 20)
    # the {hi, lo} planes of the fp64 slice sums: ~19x the compulsory 29.4 MB, 0.07 ms at 8 TB/s
    assert 5.0 * algorithmic < t["bytes"] < 40.0 * algorithmic, (rnd, t)
    assert 1.8 < t["f_fetch"] < 2.1 and 0.95 < t["f_write"] < 1.05, t


def test_a_profile_of_another_kernel_is_refused():
    path = os.path.join(ROOT, "profiles", "r4", "pmc_force_kernel.json")
    pmc = json.load(open(path))
    assert bench.pmc_traffic(pmc, "void nbx::(anonymous namespace)::accel_fast3l_kernel<3, 4, 3, 4, 64, 1, 0>(nbx::KArgs)") is None


def test_plain_gpus_n_starts_the_ranks_as_a_child_process():
    """No GPU here: each rank must get as far as bench.py's own 'needs a GPU' exit -- proof that `--gpus 2` without a launcher
    started two ranks under torch.distributed.run (it used to exit 1 with a usage line before importing anything)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--bodies", "4096"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "must be launched with" not in p.stderr
    assert p.stderr.count("bench.py needs a GPU") >= 2, p.stderr[-2000:]


def test_cpu_baseline_leg_runs_its_solvers_in_child_processes(reference):
    """bench.py's cpu_baseline on a small sample: the reference's object code, one fresh process per thread count (OpenMP sizes its
    pool at start-up), both solvers reported in ordered pairs per second, the --cpu-baseline-n override honoured."""
    out = bench.cpu_baseline(8192, 3, 1, budget_s=0.05, n_sample=3000)
    assert out["kind"] == "reference" and out["cores"] >= 1 and out["value"] > 0
    assert [r["n"] for r in out["all_runs"]][:2] == [3000, 3000]
    assert {r["solver"] for r in out["all_runs"]} == {"brute_force_omp_n_body_2<3>", "brute_force_omp_n_body_1<3>"}
    assert "first 3000 bodies" in out["sample"]
