"""Every body of a large system checked ON THE DEVICE (test infrastructure; used by tests/test_gpu_strict.py and
tests/measure/all_bodies_survey.py).

The CPU oracle cannot visit all N^2 pairs at N = 2^20 in test time (11 minutes on 16 threads, round 2's host survey), so the
strict fp64 kernel -- the reference's arithmetic type on the same fp32-representable inputs -- is first pinned to the oracle on
>= 1,024 sampled rows (every pair of those rows, `oracle_force_rows_omp_2`), and then serves as the yardstick for ALL bodies:
relative error |dF_i|/|F_i| and backward error |dF_i|/S_i (S_i = sum_j |f_ij|, from the same strict launch's magnitude-sum
build) of the plain fp32 path (nbx_ctx_set_refine(0); record key "default" for continuity with round 3's records) and of the
mixed mode (the library's default precision since ABI 4)."""
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_STRICT_REL = 1.0e-9        # strict kernel vs oracle: relative, every compared row
TOL_STRICT_BACKWARD = 2.0e-12  # strict kernel vs oracle: against the magnitude sum (two fp64 summation orders over <= 2^22 terms)
PCTS = (50.0, 90.0, 99.0, 99.9, 99.99)


def _norm(a):
    return np.sqrt((a * a).sum(axis=1))


def _variant(nbx, name):
    return nbx.variants().index(name)


def _timed(c, reps=2):
    best = None
    for _ in range(reps):
        c.synchronize()
        t0 = time.perf_counter()
        c.compute_accel()
        c.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best * 1e3


def error_stats(f, ref, S, tol=1.0e-5):
    d, n = _norm(f - ref), _norm(ref)
    assert (n > 0).all() and (S > 0).all()
    rel, back, kappa = d / n, d / S, S / n
    w = int(np.argmax(rel))
    out = dict(max_rel=float(rel[w]), kappa_at_max_rel=float(kappa[w]), max_backward=float(back.max()),
               n_over_tol=int((rel > tol).sum()), min_kappa_over_tol=float(kappa[rel > tol].min()) if (rel > tol).any() else None,
               max_rel_kappa_le_4=float(rel[kappa <= 4.0].max()) if (kappa <= 4.0).any() else 0.0,
               n_kappa_gt_4=int((kappa > 4.0).sum()), max_kappa=float(kappa.max()))
    for p in PCTS:
        out[f"rel_p{p:g}"] = float(np.percentile(rel, p))
        out[f"backward_p{p:g}"] = float(np.percentile(back, p))
    return out, rel, back, kappa


def survey(nbx, oracle, bodies, label, G=None, refine_tol=1.0e-5, sigma_factor=0.0, rows=1100, seed=7, dump=None, variant=None,
           sigmas=(12.0, 24.0, 32.0, 40.0, 48.0, 64.0, 96.0)):
    """Returns a dict of everything measured; raises AssertionError only for the strict-vs-oracle pin.
    variant: name of the fast kernel variant to survey (None: the library default)."""
    G = oracle.G if G is None else G
    n, dim = bodies.shape[0], (bodies.shape[1] - 1) // 2
    m = bodies[:, -1]
    sample = np.unique(np.random.default_rng(seed).integers(0, n, rows))
    rec = dict(what=label, n=n, dim=dim, sampled_rows=int(sample.size))
    with nbx.Context(n, dim) as c:
        c.upload(bodies)
        c.set_refine(0.0)   # a new context starts in the library's default precision (mixed mode): "default fp32" below means plain
        # 1. strict kernel (+ magnitude sums), pinned to the oracle on the sampled rows
        c.set_tuning(0, _variant(nbx, "strict_f64_t4_mag"))
        rec["strict_mag_ms"] = _timed(c, 1)
        fs = c.forces(G)
        S = c.aux() * (abs(G) * m)
        ref = oracle.force_rows_omp_2(bodies, sample) * (G / oracle.G)
        Sref = oracle.force_magnitude_sums(bodies, sample) * (abs(G) / oracle.G)
        d = _norm(fs[sample] - ref)
        rec["strict_vs_oracle_max_rel"] = float((d / _norm(ref)).max())
        rec["strict_vs_oracle_max_backward"] = float((d / Sref).max())
        rec["magnitude_sums_vs_oracle_max_rel"] = float(np.abs(S[sample] / Sref - 1.0).max())
        assert rec["strict_vs_oracle_max_rel"] <= TOL_STRICT_REL, rec
        assert rec["strict_vs_oracle_max_backward"] <= TOL_STRICT_BACKWARD, rec
        assert rec["magnitude_sums_vs_oracle_max_rel"] <= 1.0e-5, rec
        c.set_tuning(0, _variant(nbx, "strict_f64_t4"))
        rec["strict_ms"] = _timed(c, 1)
        rec["strict_kernel_ms"] = c.kernel_time()[0]
        assert np.array_equal(c.forces(G), fs), "the magnitude-sum build must not change the forces"
        # 2. the default fp32 path, ALL bodies against the strict result
        c.set_tuning(0, -1 if variant is None else _variant(nbx, variant))
        rec["default_variant"] = c.effective_tuning()[0]
        rec["default_ms"] = _timed(c, 3)
        rec["default_kernel_ms"] = c.kernel_time()[0]
        ff = c.forces(G)
        rec["default"], rel_f, back_f, kappa = error_stats(ff, fs, S, refine_tol)
        # 3. mixed mode, ALL bodies against the strict result
        c.set_refine(refine_tol, sigma_factor)
        assert c.effective_tuning()[0] == rec["default_variant"]
        rec["mixed_ms"] = _timed(c, 3)
        rec["mixed_kernel_ms"] = c.kernel_time()[0]
        rec["mixed_refine_ms"] = c.refine_time() / 3
        fm = c.forces(G)
        Q = c.aux()
        sel, done = c.refine_stats()
        rec["mixed"], rel_m, _, _ = error_stats(fm, fs, S, refine_tol)
        rec["mixed"].update(selected=sel, refined=done, tolerance=refine_tol, sigma_factor=sigma_factor,
                            cost_ms=rec["mixed_ms"] - rec["default_ms"])
        c.set_refine(0.0)
    # the selection statistic against the error it is meant to predict: sigma needed per body = rel * tol_unit / (u spread)
    a = _norm(ff) / (abs(G) * m)
    spread = np.sqrt(Q) / a
    u = 2.0 ** -24
    need = rel_f / (u * spread)          # rel_i = need_i * u * spread_i: the sigma factor that would just flag body i at tol = rel_i
    rec["sigma_needed"] = {f"p{p:g}": float(np.percentile(need, p)) for p in PCTS}
    rec["sigma_needed"]["max"] = float(need.max())
    over = rel_f > 0.5 * refine_tol
    rec["sigma_needed"]["max_among_rel_gt_half_tol"] = float(need[over].max()) if over.any() else None
    rec["sigma_needed"]["n_rel_gt_half_tol"] = int(over.sum())
    for sf in sigmas:
        flagged = spread * u * sf > refine_tol
        rec[f"rule_sigma_{sf:g}"] = dict(flagged=int(flagged.sum()), missed_over_tol=int((~flagged & (rel_f > refine_tol)).sum()),
                                         worst_unflagged_rel=float(rel_f[~flagged].max()) if (~flagged).any() else 0.0)
    if dump:
        keep = np.argsort(rel_f)[-200000:]
        np.savez_compressed(dump, idx=keep.astype(np.uint32), rel=rel_f[keep].astype(np.float32), spread=spread[keep].astype(np.float32),
                            kappa=kappa[keep].astype(np.float32), rel_mixed=rel_m[keep].astype(np.float32))
    return rec


def write_record(rec, name="accuracy_all_bodies.jsonl"):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name), "a") as fh:
        fh.write(json.dumps(rec) + "\n")
