"""Accuracy evidence of SURVEY 8(d): device accelerations vs the oracle (reference arithmetic, fp64, fp32-rounded
inputs) on BASELINE's uniform configs -- FULL N at N = 65,536 (oracle brute_force_omp_2) and >= 1,024 sampled
target rows at N = 2^20 -- with the distribution of per-body relative errors, the condition numbers of the
offenders and the max-abs acceleration error.  Test infrastructure (uses oracle/); prints JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "16")
import numpy as np
import nbody_amd as nbx
from oracle_lib import Oracle, force_errors

o = Oracle()

# the oracle calls below are single blocking C calls of several minutes at N = 2^20: keep a heartbeat on stderr so that a
# supervised run is not taken for hung (ctypes releases the GIL during the call)
import threading
def _heartbeat():
    t0 = time.time()
    while True:
        time.sleep(60)
        sys.stderr.write(f"[accuracy_survey] still working, {time.time() - t0:.0f} s\n")
        sys.stderr.flush()
threading.Thread(target=_heartbeat, daemon=True).start()
cases = [(65536, 3, None, 2), (65536, 2, None, 2), (1 << 20, 3, 2048, 3), (1 << 20, 2, 1024, 3)]
if "full20" in sys.argv[1:]:        # every one of the 1,048,576 bodies against the oracle: ~5 minutes of host time at 16 threads
    cases = [(1 << 20, 3, None, 3)]
elif "plummer22" in sys.argv[1:]:   # BASELINE config 5's own input (N = 4,194,304 Plummer sphere, tests/test_gpu_config5.py): 32,768 rows
    cases = [(1 << 22, 3, 32768, 5)]
elif len(sys.argv) > 1:
    cases = [c for c in cases if str(c[0]) in sys.argv[1:]]
for n, dim, nrows, seed in cases:
    if "plummer22" in sys.argv[1:]:
        b = o.round_inputs_to_f32(nbx.plummer_bodies(n, 3, seed=seed, a=1.0e5, total_mass=1.0e12))
    else:
        b = o.round_inputs_to_f32(o.generate(seed, n, dim))
    with nbx.Context(n, dim) as c:
        c.upload(b)
        if os.environ.get("NBX_SURVEY_VARIANT"):      # e.g. lds_t1_w8_exact_u8: fp64 second-level sums, for comparison
            c.set_tuning(0, nbx.variants().index(os.environ["NBX_SURVEY_VARIANT"]))
        c.compute_accel()
        f = c.forces(o.G)
        name = c.effective_tuning()[0]
    t0 = time.perf_counter()
    if nrows is None:
        rows = np.arange(n)
        ref = o.brute_force_omp_2(b)
        S = o.force_magnitude_sums(b)
    else:
        rows = np.unique(np.random.default_rng(seed).integers(0, n, nrows))
        ref = o.force_rows_omp_2(b, rows)
        S = o.force_magnitude_sums(b, rows)
    t_or = time.perf_counter() - t0
    fr = f[rows]
    e = force_errors(fr, ref, S)
    m = b[rows, -1][:, None]
    dF = np.sqrt(((fr - ref) ** 2).sum(1)); nF = np.sqrt((ref ** 2).sum(1))
    rel = dF / nF
    kappa = S / nF
    over = rel > 1e-5
    e.pop("n", None)
    out = dict(n=n, dim=dim, rows=int(rows.size), variant=name, oracle_s=round(t_or, 2), **e,
               max_abs_accel_err=float(np.abs((fr - ref) / m).max()), max_abs_accel=float(np.abs(ref / m).max()),
               n_over_1e5=int(over.sum()), kappa_of_offenders_min=float(kappa[over].min()) if over.any() else None,
               rel_percentiles={str(p): float(np.percentile(rel, p)) for p in (50, 90, 99, 99.9, 100)},
               kappa_percentiles={str(p): float(np.percentile(kappa, p)) for p in (50, 90, 99, 99.9, 100)},
               backward_percentiles={str(p): float(np.percentile(dF / S, p)) for p in (50, 99, 100)})
    print(json.dumps(out), flush=True)
