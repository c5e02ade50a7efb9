#!/usr/bin/env python3
"""bench.py -- pair-interactions/s of the MI355X all-pairs force + kick/drift step.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 it is launched as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
one rank per GPU over RCCL.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2]/[3]): N = 1,048,576 bodies, 3D, uniform-random with the
reference's generator ranges (nbody-sim-new/utils.h:113-115), fp32 device arithmetic.  A "step" is one
pass of the hot path over all bodies: all-pairs force evaluation (N^2 ordered pairs, split over the
ranks' target shards) + fused kick/drift (+ the per-step RCCL all-gather of positions when N > 1).
value = N^2 * K / t, inputs resident in HBM before the timed region.  Total work is fixed as ranks are
added => "scaling": "strong".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# multi-process GPU work on this pool's hosts needs dmabuf IPC (RCCL / cross-process device memory fail with
# hipIpcGetMemHandle: invalid argument otherwise); the launcher's environment normally carries it already
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# host threads for the CPU baseline: the GPU box's CPU share for one GPU is 16 hardware threads
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

FLOP_PER_INTERACTION = 20          # SURVEY 8d convention (3 sub, 5 r^2, rcp, 3 mul, 6 fma-acc, 2 guard)
PEAK_FP32_TFLOPS = 157.3           # MI355X_MICROARCH.md: peak FP32 vector = FP32 matrix
HBM_PEAK_GBPS = 8000.0
PROFILE_ROUNDS = ("r5", "r4")      # profiles/<round>/pmc_force_kernel.json, newest first: the committed PMC passes of this command
SHADER_PEAK_MHZ = 2400.0           # MI355X_MICROARCH.md: the clock the 157.3 TFLOP/s figure assumes


def host_facts():
    """CPU model / sockets / cores of this box (lscpu) and how many hardware threads this process may use."""
    import subprocess
    facts = {}
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True).stdout.splitlines():
            k, _, v = line.partition(":")
            if k.strip() in ("Model name", "CPU(s)", "Thread(s) per core", "Core(s) per socket", "Socket(s)"):
                facts[k.strip()] = v.strip()
    except OSError:
        pass
    facts["usable_hw_threads"] = len(os.sched_getaffinity(0))
    try:
        phys = int(facts["Core(s) per socket"]) * int(facts["Socket(s)"])
    except (KeyError, ValueError):
        phys = facts["usable_hw_threads"]
    quota = None   # cgroup CPU bandwidth limit: "max" or "<quota> <period>" -- more threads than that are only time-sliced
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    facts["cgroup_cpu_quota"] = quota
    usable = min(phys, facts["usable_hw_threads"])
    if quota:
        usable = min(usable, int(quota + 0.5))
    facts["physical_cores_usable"] = max(1, usable)
    return facts


def cpu_baseline_child(threads, n_s, dim, seed):
    """Child process of cpu_baseline(): OpenMP / ParlayLib size their pools at start-up, so every thread count
    runs in a fresh process (env set by the parent).  Prints one JSON line per reference solver."""
    import numpy as np
    import nbody_amd as nbx
    from oracle_lib import Reference
    ref = Reference()
    sub = np.ascontiguousarray(nbx.uniform_bodies(n_s, dim, seed))
    for v, name in ((2, "brute_force_omp_n_body_2"), (1, "brute_force_omp_n_body_1")):
        dt = ref.time_brute_force(v, sub)
        print(json.dumps({"solver": f"{name}<{dim}>", "threads": threads, "n": n_s, "seconds": dt,
                          "value": n_s * (n_s - 1) / dt}), flush=True)


def cpu_baseline(n_bodies, dim, seed, budget_s=10.0, n_sample=0):
    """Reported baseline (not the target): the reference's own brute_force_omp_n_body_2 object code
    (oracle/_ref, kind "reference") or the oracle port of it (kind "port") on a bounded sample of
    the same workload, timed on this host's cores: at 16 threads (the box's CPU share for one GPU, the primary
    figure) and, when the process may use more (sockets x cores, capped by the affinity mask AND the cgroup CPU quota --
    the GPU boxes of this pool grant exactly 16 CPUs, so there the two coincide), on all usable physical cores (also
    omp_1, the reference's faster symmetric variant).  value = ordered pair interactions/s (N(N-1) per evaluation for every solver, so the
    figures compare with the GPU's; the symmetric omp_1 evaluates half as many pairs)."""
    import subprocess
    import numpy as np
    from oracle_lib import Oracle, have_reference
    o = Oracle()
    threads = o.num_threads()
    host = host_facts()
    bodies = o.generate(seed, n_bodies, dim)
    # calibrate with the port on a few rows, then size the sample for ~budget_s
    rows = np.arange(0, 2048, dtype=np.int64)
    t0 = time.perf_counter()
    o.force_rows_omp_2(bodies, rows)
    rate = rows.size * n_bodies / (time.perf_counter() - t0)
    if have_reference():
        try:
            if n_sample:    # --cpu-baseline-n: SURVEY 8(d)'s sizes (262,144 or the full N); minutes of host time, never the driver's default
                n_s = int(min(n_bodies, n_sample))
            else:
                n_s = int(min(n_bodies, max(4096, (rate * budget_s) ** 0.5)))
                n_s = 1 << (n_s.bit_length() - 1)
            runs = []
            counts = [threads] + ([host["physical_cores_usable"]] if host["physical_cores_usable"] > threads else [])
            for c in counts:
                env = dict(os.environ, OMP_NUM_THREADS=str(c), PARLAY_NUM_THREADS=str(c), OMP_PROC_BIND="spread", OMP_PLACES="cores")
                p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", str(c), "--bodies", str(n_s),
                                      "--dim", str(dim), "--seed", str(seed)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                t_child = time.perf_counter()
                while True:     # a sign of life per minute on stderr: a full-N baseline is silent for five minutes per solver
                    try:
                        out, err = p.communicate(timeout=60)
                        break
                    except subprocess.TimeoutExpired:
                        sys.stderr.write(f"[bench] cpu baseline ({c} threads, {n_s} bodies): {time.perf_counter() - t_child:.0f} s so far\n")
                        sys.stderr.flush()
                        if time.perf_counter() - t_child > (3000 if n_sample else 600):
                            p.kill()
                            p.communicate()
                            raise RuntimeError("cpu baseline child timed out")
                if p.returncode:
                    raise RuntimeError(err[-300:])
                runs += [json.loads(l) for l in out.splitlines() if l.startswith("{")]
            first = runs[0]
            return {"value": first["value"], "unit": "pair-interactions/s", "cores": first["threads"], "kind": "reference",
                    "sample": f"reference {first['solver']} object code (oracle/_ref) on the first {n_s} bodies of the workload, "
                              f"{n_s*(n_s-1):.3e} ordered pairs in {first['seconds']:.2f} s, OMP_NUM_THREADS={first['threads']} "
                              "(one GPU's share of the host)",
                    "host": host, "all_runs": runs}
        except Exception as e:  # fall through to the port
            sys.stderr.write(f"[bench] reference baseline unavailable ({e}); using the oracle port\n")
    nrows = int(min(n_bodies, max(2048, rate * budget_s / n_bodies)))
    rows = np.linspace(0, n_bodies - 1, nrows).astype(np.int64)
    t0 = time.perf_counter()
    o.force_rows_omp_2(bodies, rows)
    dt = time.perf_counter() - t0
    return {"value": nrows * (n_bodies - 1) / dt, "unit": "pair-interactions/s", "cores": threads, "kind": "port",
            "sample": f"oracle port of brute_force_omp_n_body_2<{dim}>: {nrows} target rows x {n_bodies} sources "
                      f"({nrows*(n_bodies-1):.3e} pair evaluations) in {dt:.2f} s, OMP_NUM_THREADS={threads}", "host": host}


def leaf_pair_roofline(device):
    """The repository's second hand-written kernel family (near-field sums of the tree codes, csrc/leaf_pair_kernel.hip)
    against the same fp32 roofline, on tools/time_leaf_pairs.py's workload: N = 2^20 bodies in 32^3 grid leaves, 27-cell lists,
    the TREE_LEAF law (the one pinned on the reference's own BVH leaves), through the resident plan (nbx_leaf_plan_*): bodies in a
    context, structure laid out once.  Not part of the headline metric; a few seconds on rank 0 of a 1-GPU run."""
    import numpy as np
    import nbody_amd as nbx
    n = 1 << 20
    b = nbx.uniform_bodies(n, 3, 5)
    leaves = nbx.leaves.uniform_grid_leaves(b, 3, 5)
    lo, _, so, ss = leaves
    sizes = np.diff(lo).astype(np.int64)
    pairs = int((sizes * np.add.reduceat(sizes[ss], so[:-1])).sum())
    law = nbx.LAW_TREE_LEAF
    with nbx.LeafPlan(n, 3, *leaves, device=device) as plan, nbx.Context(n, 3, device=device) as ctx:
        ctx.upload(b)
        ctx.synchronize()
        time.sleep(0.5)                                   # idle clocks: what the first evaluation after host-side work meets
        cold = plan.forces_ctx(ctx, law, fetch=False, timed=True)
        walls = []
        for _ in range(20):
            ctx.synchronize()
            t0 = time.perf_counter()
            plan.forces_ctx(ctx, law, fetch=False)
            ctx.synchronize()
            walls.append((time.perf_counter() - t0) * 1e3)
        plan.time_kernel(law, 100)                        # 30 ms of launches: the clocks are up (the host built the leaves for seconds just before)
        single = min(plan.forces_ctx(ctx, law, fetch=False, timed=True) for _ in range(3))
        warm = plan.time_kernel(law, 300)
    tflops = lambda ms: pairs * 20.0 / (ms * 1e-3) / 1e12
    # the same measurement at the leaf sizes the reference's trees make: its BVH's 16-body leaves, what a 16-body
    # cap makes of smaller nodes (8), and 4-body cells; structures of small leaves go through the packed kernel (leaf_pack_kernel)
    by_size = []
    for label, build in (("median-split leaves of 16 bodies (the reference BVH's cap, methods.h:57), box-distance near-field lists", lambda: nbx.leaves.median_split_leaves(b, 3, 16, reach=0.5)),
                         ("median-split leaves of 8 bodies", lambda: nbx.leaves.median_split_leaves(b, 3, 8, reach=0.5)),
                         ("64^3 grid cells of ~4 bodies, 27-cell lists", lambda: nbx.leaves.uniform_grid_leaves(b, 3, 6))):
        sl = build()
        ssz = np.diff(sl[0]).astype(np.int64)
        sp = int((ssz * np.add.reduceat(ssz[sl[3]], sl[2][:-1])).sum())
        # what a caller that rebuilds its tree per evaluation pays (the reference does: methods.cpp:377-401): a NEW plan for the
        # structure (validation + layout on the device + buffers; the previous plan destroyed first), and the whole one-shot call
        remake, oneshot = 1e30, 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            with nbx.LeafPlan(n, 3, *sl, device=device):
                remake = min(remake, (time.perf_counter() - t0) * 1e3)
        for _ in range(3):
            t0 = time.perf_counter()
            nbx.leaf_pair_forces_hip(b, *sl, law=law, device=device)
            oneshot = min(oneshot, (time.perf_counter() - t0) * 1e3)
        with nbx.LeafPlan(n, 3, *sl, device=device) as plan, nbx.Context(n, 3, device=device) as ctx:
            ctx.upload(b)
            ctx.synchronize()
            for _ in range(20):                           # as for the 32-body row above: 20 evaluations, then the best of 3 timed ones
                plan.forces_ctx(ctx, law, fetch=False)
                ctx.synchronize()
            plan.time_kernel(law, 100)
            one = min(plan.forces_ctx(ctx, law, fetch=False, timed=True) for _ in range(3))
            many = plan.time_kernel(law, 300)
            slots, runs, groups, waves = plan.info()
        by_size.append({"workload": f"N={n}, {ssz.size} {label} (mean {ssz.mean():.1f})", "pair_terms_per_launch": sp,
                        "kernel": "leaf_pack_kernel / leaf_fused_kernel<3, NBX_LAW_TREE_LEAF> (several leaves to a wave)" if ssz.mean() <= 8.0 else "leaf_pair_kernel<3, NBX_LAW_TREE_LEAF, 1>",
                        "workgroups": int(groups), "kernel_ms": one, "frac": sp * 20.0 / (one * 1e-3) / 1e12 / 157.3,
                        "new_plan_ms": remake, "one_shot_call_ms": oneshot,
                        "new_plan_means": "nbx_leaf_plan_create wall time, best of 3 (CSR arrays from host memory, layout on the device: csrc/leaf_plan_device.h); "
                                          "one_shot_call_ms = nbx_leaf_pair_forces wall time, best of 3 (58 MB of bodies in, 25 MB of forces out)",
                        "back_to_back": {"kernel_ms": many, "frac": sp * 20.0 / (many * 1e-3) / 1e12 / 157.3}})
    return {"kernel": "leaf_pair_kernel<3, NBX_LAW_TREE_LEAF, 2>", "workload": f"N={n}, {sizes.size} grid leaves (mean {sizes.mean():.1f} bodies), 27-cell lists",
            "entry": "nbx_leaf_plan_forces_ctx (resident plan + resident bodies)", "law_pin": "TREE_LEAF: pinned on the reference's own BVH leaves (tests/golden/bvh_leaves_*.npz)",
            "pair_terms_per_launch": pairs, "flop_per_pair_term": 20, "bound": "mfma",
            "bound_detail": "fp32 VALU issue, as for the force kernel (no MFMA instructions)",
            "achieved": tflops(single), "peak": 157.3, "unit": "TFLOP/s", "frac": tflops(single) / 157.3,
            "kernel_ms": single, "kernel_ms_means": "one launch per evaluation (best of 3 evaluations in a row, device warm: 100 launches just before; the launch from idle clocks is first_launch_from_idle_clocks)",
            "evaluation_wall_ms_median": float(np.median(walls)),
            "first_launch_from_idle_clocks": {"kernel_ms": cold, "achieved": tflops(cold), "frac": tflops(cold) / 157.3},
            "back_to_back": {"kernel_ms": warm, "achieved": tflops(warm), "frac": tflops(warm) / 157.3,
                             "means": "mean of launches 151-300 of 300 back to back (nbx_leaf_plan_time_kernel): clocks up"},
            "at_the_reference_trees_leaf_sizes": by_size}


def launched_kernel_symbol(variant_name, dim, refine_tol):
    """Demangled symbol of the force kernel a run with this variant / precision mode launches (from the library; no device)."""
    import nbody_amd as nbx
    capi = nbx.package.capi
    try:
        return capi.variant_kernel_symbol(variant_name, dim, bool(refine_tol))
    except nbx.NbxError:       # a variant without a mixed-mode build runs its plain kernel whatever the tolerance
        return capi.variant_kernel_symbol(variant_name, dim, False)


def pmc_traffic(pmc, launched):
    """HBM bytes per launch of kernel `launched` from a profiles/<round>/pmc_force_kernel.json (tools/summarize_prof.py), or None
    when the profile is of another kernel.  Units and gfx950 corrections as the guide's HBM section prescribes: KB -> bytes;
    FETCH_SIZE under-reports coalesced streaming reads by 2, confirmed for THIS path's streams by the helper kernels of the same
    profile whose byte counts are known exactly ("calibration")."""
    if not (pmc["pmc_fetch"]["kernel"]["Kernel_Name"] == launched == pmc["pmc_write"]["kernel"]["Kernel_Name"]):
        return None
    fetch_kb = pmc["pmc_fetch"]["per_launch_mean"]["FETCH_SIZE"]
    write_kb = pmc["pmc_write"]["per_launch_mean"]["WRITE_SIZE"]
    cal = pmc.get("calibration", {})
    ff = [v["FETCH_SIZE"]["true_over_reported"] for v in cal.values() if "FETCH_SIZE" in v]
    wf = [v["WRITE_SIZE"]["true_over_reported"] for v in cal.values() if "WRITE_SIZE" in v]
    f_fetch = sum(ff) / len(ff) if ff else 2.0      # guide: x2 on gfx950 for coalesced streaming reads
    f_write = sum(wf) / len(wf) if wf else 1.0
    return {"bytes": (f_fetch * fetch_kb + f_write * write_kb) * 1024.0, "fetch_kb": fetch_kb, "write_kb": write_kb,
            "f_fetch": f_fetch, "f_write": f_write}


def _norm(a):
    import numpy as np
    return np.sqrt((a * a).sum(axis=1))


def parity_block(system, bodies, G, args, world, rank, dist, refine_tol=1.0e-5):
    """BASELINE metric, second half, for ANY number of ranks -- run by every rank after the timed region.
    (1) SURVEY 8d protocol: >= 1,024 sampled target rows (all rows at N <= 65,536) of one sharded force evaluation on the
        initial positions, gathered to rank 0 and compared with the reference's arithmetic (oracle rows in fp64 on the
        fp32-representable inputs): max-abs / max-relative acceleration error, backward error.
    (2) EVERY body, on the device: the strict fp64 kernel variant (the reference's arithmetic type; pinned to the oracle on the
        same sampled rows, here, first) is the yardstick for all N targets of the default fp32 path and of the mixed mode
        (nbx_ctx_set_refine): each rank checks its own shard, rank 0 combines.
    Returns the `accuracy` object on rank 0, None elsewhere."""
    import numpy as np
    import nbody_amd as nbx
    be, ctx, lay = system.be, system.be.ctx, system.layout
    n, dim = bodies.shape[0], args.dim
    lo, hi = lay.bounds()
    names = nbx.variants()
    rows = np.arange(n) if n <= 65536 else np.unique(np.linspace(0, n - 1, 1024).astype(np.int64))
    mine = rows[(rows >= lo) & (rows < hi)]
    # the device rounds positions and masses to fp32 at the boundary; the checker's inputs ARE those values, so
    # they are uploaded as such -- the fp64 mass that scales a force is then the mass the oracle uses
    initial = bodies.copy()
    initial[:, :dim] = initial[:, :dim].astype(np.float32).astype(np.float64)
    initial[:, -1] = initial[:, -1].astype(np.float32).astype(np.float64)
    rounded_m = initial[lo:hi, -1]

    def evaluate():
        be.synchronize()
        be.upload(initial, system.group)   # back to the initial state (the timed steps moved the bodies): own shard up, the rest gathered
        system.compute_forces()
        return system.forces(G)

    def refine_counts():
        try:
            return ctx.refine_stats()
        except nbx.NbxError:      # the mixed mode does not apply to the kernel variant that ran (an exact / strict variant was asked for)
            return (0, 0)

    headline_tol = float(args.refine)            # precision mode of the TIMED steps (default: mixed mode, 1e-5)
    other_tol = 0.0 if headline_tol else refine_tol
    ctx.set_refine(headline_tol)
    f_head = evaluate()
    head_sel = refine_counts() if headline_tol else (0, 0)
    out = {"rank": rank, "lo": int(lo), "rows": mine, "f_rows": f_head[mine - lo]}
    if not args.no_all_bodies:
        # every rank runs the SAME sequence of evaluations (each holds an exchange = a collective), whatever its shard holds;
        # only the statistics below depend on the shard being non-empty
        ctx.set_tuning(args.splits, names.index("strict_f64_t4_mag"))
        t0 = time.perf_counter()
        f_str = evaluate()
        be.synchronize()
        out["strict_s"] = time.perf_counter() - t0
        S = ctx.aux() * (abs(G) * rounded_m)
        ctx.set_tuning(args.splits, args.variant)
        ctx.set_refine(other_tol)
        f_other = evaluate()
        other_sel = refine_counts() if other_tol else (0, 0)
        be.synchronize()
        t0 = time.perf_counter()
        system.step(args.dt, G, 3)      # what a step costs in the OTHER precision mode (not part of `value`)
        be.synchronize()
        out["other_step_s"] = (time.perf_counter() - t0) / 3
        ctx.set_refine(headline_tol)
        if hi > lo:
            nrm = _norm(f_str)

            def stats(f):
                d = _norm(f - f_str)
                rel, back = d / nrm, d / S
                return {"max_rel": float(rel.max()), "n_over_1e-5": int((rel > 1e-5).sum()), "max_backward": float(back.max()),
                        "rel_p99.9": float(np.percentile(rel, 99.9)), "rel_p50": float(np.percentile(rel, 50)),
                        "kappa_at_max_rel": float((S / nrm)[int(np.argmax(rel))])}
            f_mix, f_def = (f_head, f_other) if headline_tol else (f_other, f_head)
            sel, done = head_sel if headline_tol else other_sel
            out.update(strict_rows=f_str[mine - lo], S_rows=S[mine - lo], default=stats(f_def), mixed=stats(f_mix),
                       selected=int(sel), refined=int(done), count=int(hi - lo))
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, out)
    else:
        gathered = [out]
    if rank != 0:
        return None
    from oracle_lib import KAPPA_WELL, Oracle, force_errors
    o = Oracle()
    rounded = o.round_inputs_to_f32(bodies)
    assert np.array_equal(rounded[:, :dim], initial[:, :dim]) and np.array_equal(rounded[:, -1], initial[:, -1])
    ref = o.force_rows_omp_2(rounded, rows) * (G / o.G)
    S_ref = o.force_magnitude_sums(rounded, rows) * (abs(G) / o.G)
    got_rows = np.concatenate([g["rows"] for g in gathered])
    assert np.array_equal(got_rows, rows), "the ranks' sampled rows must tile the sample"
    f = np.concatenate([g["f_rows"] for g in gathered])
    m = rounded[rows, -1][:, None]
    da = (f - ref) / m
    rel = _norm(da) / _norm(ref / m)
    e = force_errors(f, ref, S_ref)
    acc = {"rows": int(rows.size), "ranks": world, "max_abs_accel_err": float(np.abs(da).max()), "max_rel_accel_err": float(rel.max()),
           "sampled_rows_within_1e-5": bool(rel.max() <= 1e-5), "sampled_rows_over_1e-5": int((rel > 1e-5).sum()),
           "sampled_rows_means": "a statement about these rows only; every body is covered by all_bodies below",
           "n_ill": e["n_ill"], "ill_means": f"kappa = sum_j|f_ij| / |F_i| > {KAPPA_WELL:g}",
           "max_backward_err": e["max_backward"], "max_abs_accel": float(np.abs(ref / m).max()),
           "reference": "oracle rows of brute_force_omp_n_body_2 in fp64 on the fp32-rounded inputs (= sequential path up to fp64 re-association)"}
    parts = [g for g in gathered if "default" in g]
    headline_tol = float(args.refine)
    if parts:
        fs = np.concatenate([g["strict_rows"] for g in gathered if "strict_rows" in g])
        Ss = np.concatenate([g["S_rows"] for g in gathered if "S_rows" in g])
        d = _norm(fs - ref)

        def combine(key):
            w = max(parts, key=lambda g: g[key]["max_rel"])[key]
            return {"max_rel": w["max_rel"], "kappa_at_max_rel": w["kappa_at_max_rel"], "n_over_1e-5": sum(g[key]["n_over_1e-5"] for g in parts),
                    "max_backward": max(g[key]["max_backward"] for g in parts), "rel_p99.9_worst_rank": max(g[key]["rel_p99.9"] for g in parts),
                    "rel_p50_worst_rank": max(g[key]["rel_p50"] for g in parts)}
        mixed = combine("mixed")
        mixed.update(tolerance=headline_tol or refine_tol, selected=sum(g["selected"] for g in parts), refined=sum(g["refined"] for g in parts))
        plain = combine("default")
        other_ms = max(g["other_step_s"] for g in gathered if "other_step_s" in g) * 1e3
        (plain if headline_tol else mixed)["ms_per_step"] = other_ms       # the mode that was NOT timed as `value`
        acc["all_bodies"] = {
            "headline_mode": "mixed_mode" if headline_tol else "default_fp32",
            "checked": sum(g["count"] for g in parts),
            "yardstick": "strict fp64 kernel variant (strict_f64_t4_mag) on the device, every target x every source",
            "yardstick_vs_oracle_rows": {"max_rel": float((d / _norm(ref)).max()), "max_backward": float((d / S_ref).max()),
                                         "magnitude_sums_max_rel": float(np.abs(Ss / S_ref - 1.0).max())},
            "strict_evaluation_s": max(g["strict_s"] for g in parts),
            "default_fp32": plain, "mixed_mode": mixed}
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bodies", type=int, default=1 << 20, help="N (default 2^20 = BASELINE metric config)")
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--dt", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--variant", type=int, default=-1, help="force-kernel variant id (-1: library default)")
    ap.add_argument("--splits", type=int, default=0, help="source slices (0: automatic)")
    ap.add_argument("--refine", type=float, default=1.0e-5,
                    help="per-body relative tolerance of the TIMED steps: the library's mixed mode (fp32 for all bodies + fp64 for the "
                         "suspects; default 1e-5 = the north star's tolerance, met for every body); 0 = plain fp32")
    ap.add_argument("--no-all-bodies", action="store_true", help="skip the device-side check of every body against the strict fp64 kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-n", type=int, default=0,
                    help="bodies of the CPU baseline's sample (0: automatic, ~10 s of host time = 131,072 on a GPU box's 16 threads; "
                         "SURVEY 8(d) names 262,144 or the full N -- minutes, for profiles/, not for the driver's default run)")
    ap.add_argument("--cpu-baseline-child", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        cpu_baseline_child(args.cpu_baseline_child, args.bodies, args.dim, args.seed)
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Plain `python bench.py --gpus N`: this process has made no GPU call (and makes none): it starts the N ranks as a CHILD
        # `python -m torch.distributed.run`, which inherits stdout (rank 0's one JSON line goes straight through), waits, and
        # exits with the child's status.  Never an exec: a process that has touched the GPU must not be replaced on this pool.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=dict(os.environ, NBODY_BENCH_LAUNCHER="bench.py started torch.distributed.run as a child process")).returncode)

    import numpy as np
    import torch
    import torch.distributed as dist
    import nbody_amd as nbx
    if not os.path.exists(nbx.LIB_PATH):  # checkout without build products: local rank 0 builds in-tree, the others wait
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            for _ in range(600):
                if os.path.exists(nbx.LIB_PATH) and os.path.exists(os.path.join(ROOT, "nbody_sim")):
                    break
                time.sleep(1.0)
            time.sleep(2.0)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # under a launcher the launcher's world size is the truth
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.bodies < world:
        sys.exit(f"bench.py: {args.bodies} bodies cannot be sharded over {world} ranks")
    # Rehearsal knobs (not used by the driver): NBODY_BENCH_BACKEND=gloo stages the exchange through host memory
    # and NBODY_BENCH_DEVICE=<i> puts every rank on one GPU, so the N > 1 code path can run on a single-GPU box.
    backend = os.environ.get("NBODY_BENCH_BACKEND", "nccl")
    if "NBODY_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["NBODY_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    check_store = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        try:   # the exchange self-check's bookkeeping rides the rendezvous store: a transport of its own, nothing printed
            check_store = dist.distributed_c10d._get_default_store()
        except Exception:
            check_store = None

    # synthetic bodies, identical on every rank (the reference generator's stream, seeded)
    bodies = nbx.uniform_bodies(args.bodies, args.dim, args.seed)
    N = args.bodies

    system = nbx.package.dist.make_hip_system(bodies, args.dim, rank=rank, world_size=world, device_index=local_rank,
                                             variant=args.variant, source_splits=args.splits, refine_tol=args.refine,
                                             check_store=check_store)
    be = system.be
    G = nbx.REFERENCE_G

    def sync_all():
        be.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # First contact with a multi-GPU node: before anything is timed, prove that the collective really moves the
    # chunks -- every rank poisons the chunks it does not own, one exchange runs, and each remote chunk is compared
    # on the host with the generated bodies (identical on every rank).  A mismatch ends the run non-zero.
    exchange_check = None
    if world > 1:
        try:
            bad = system.verify_exchange(bodies)
        except nbx.package.dist.ExchangeError as e:
            sys.stderr.write(f"[bench] rank {rank}: {e}\n")
            os._exit(3)   # the communicator may be wedged: no collective teardown
        exchange_check = {"mismatching_values": bad, "checked_values_per_rank": int(args.dim * (N - system.layout.count)),
                          "transport": backend,
                          "all_gather_form": "in place" if getattr(be, "inplace_gather", True) else "separate send buffer (the in-place form failed its self-check)"}
        if bad:
            sys.stderr.write(f"[bench] rank {rank}: position exchange FAILED its self-check: {bad} fp32 values differ from the "
                             f"generated bodies after one {backend} all-gather\n")
            dist.destroy_process_group()
            sys.exit(3)

    # the force kernel's workgroups stamp the shader clock they held (nbx_ctx_enable_clock_stamps: two scalar clock reads and
    # one 16-byte store per workgroup of ~40 ms); on before the warm-up so that the captured step carries it
    clock_stamps = True
    try:
        be.ctx.enable_clock_stamps(True)
    except nbx.NbxError:   # a variant without stamps was asked for
        clock_stamps = False
    for _ in range(args.warmup):
        system.step(args.dt, G, 1)
    sync_all()
    be.kernel_time()  # reset the force-kernel event log
    if world > 1:
        be.enable_timing(True)   # events around LOCAL / REMOTE (compute stream) and the exchange (comm stream)
    sync_all()
    t0 = time.perf_counter()
    system.step(args.dt, G, args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        mine = dict(be.pass_times(), rank=rank, device=local_rank, targets=int(system.layout.count), wall_s=elapsed)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
        be.enable_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms, launches = be.kernel_time()  # HIP events on the stream the kernels ran on
    refine_ms_total = be.ctx.refine_time()  # the mixed mode's kernels behind those launches (third event of each evaluation)
    held = None
    if clock_stamps:
        try:
            held = be.ctx.shader_clock()     # the LAST timed force launch's workgroups (this rank)
        except nbx.NbxError:
            held = None
        be.ctx.enable_clock_stamps(False)
    ceiling = None
    if rank == 0:   # the same chip, straight after the timed steps (still warm): a pure v_pk_fma_f32 stream for ~50 ms
        try:
            tf, mhz = nbx.package.capi.measure_valu_ceiling(local_rank, 50.0)
            ceiling = {"tflops": tf, "shader_mhz": mhz}
        except nbx.NbxError:
            ceiling = None

    result = None
    if rank == 0:
        interactions = float(N) * float(N) * args.steps
        value = interactions / elapsed
        my_pairs_per_launch = float(system.layout.count) * float(N) / (1 if world == 1 else 1)  # per step on this rank
        launches_per_step = 1 if world == 1 else 2
        kern_s_per_step = kern_ms * 1e-3 * launches_per_step
        achieved_tflops = my_pairs_per_launch * FLOP_PER_INTERACTION / kern_s_per_step / 1e12
        variant_name, source_splits = be.ctx.effective_tuning()
        result = {
            "metric": "body-pair interactions/sec at N=2^20 (all-pairs force + kick/drift step)" if N == 1 << 20
                      else f"body-pair interactions/sec at N={N}",
            "value": value, "unit": "pair-interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"N={N} {args.dim}D uniform-random bodies (reference generator ranges, seed {args.seed}), "
                                   "one all-pairs force evaluation + fused kick/drift per step",
                       "n_bodies": N, "dim": args.dim, "lds_tile": 256, "kernel_variant": variant_name, "source_slices": source_splits,
                       "acc_planes": source_splits * (2 if variant_name.startswith(("fastpk3l", "strict")) else 1),   # fp32 planes of partial sums: {hi, lo} per slice for the fp64-sum kernels
                       "parallelism": "1 GPU" if world == 1 else f"{world} target shards, RCCL all-gather of positions per step"},
            # bound: the compute roofline of the two the contract names ("hbm" | "mfma").  The kernel issues no MFMA
            # (north star); its limit is fp32 VALU issue, and on MI355X the fp32 matrix peak equals the fp32 vector
            # peak (157.3 TFLOP/s), which is the figure used.
            "roofline": {"bound": "mfma", "bound_detail": "compute roofline = fp32 VALU issue (no MFMA instructions; fp32 matrix peak = fp32 vector peak = 157.3 TFLOP/s)",
                         "achieved": achieved_tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tflops / PEAK_FP32_TFLOPS, "traffic": None,
                         "kernel": "nbx force kernel, variant " + variant_name, "kernel_ms_mean": kern_ms,
                         "kernel_launches_timed": launches, "flop_per_interaction": FLOP_PER_INTERACTION,
                         "interactions_per_launch": my_pairs_per_launch / launches_per_step,
                         "whole_step_frac": value / world * FLOP_PER_INTERACTION / 1e12 / PEAK_FP32_TFLOPS if world == 1 else None,
                         "mixed_mode_kernels_ms_per_step": refine_ms_total / max(args.steps, 1),
                         # what THIS box held during THIS run: frac above is against 157.3 TFLOP/s = 2.4 GHz x one packed FMA per
                         # lane pair per cycle; the chip clocks lower under a dense VALU load, and by how much differs from box to box
                         "shader_mhz": held["median_mhz"] if held else None,
                         "shader_mhz_detail": dict(held, source="in-kernel stamps (s_memtime / s_memrealtime) of the last timed force launch, "
                                                   "one pair per workgroup, nbx_ctx_shader_clock") if held else "not measured (kernel variant without stamps)",
                         "frac_at_held_clock": achieved_tflops / (PEAK_FP32_TFLOPS * held["median_mhz"] / SHADER_PEAK_MHZ) if held else None,
                         "this_box_ceiling": dict(ceiling, kernel="pure v_pk_fma_f32 stream, 16 chains per lane, every CU full, ~50 ms right after the "
                                                  "timed steps (nbx_measure_valu_ceiling)") if ceiling else None,
                         "frac_of_this_box_ceiling": achieved_tflops / ceiling["tflops"] if ceiling else None,
                         "note": "arithmetic intensity ~7.5e5 flop/B: HBM-light; compare runs across boxes by frac_at_held_clock / frac_of_this_box_ceiling, not by frac",
                         "hbm": {"algorithmic_bytes_per_launch": 28.0 * N / world if world == 1 else (16.0 * N + 12.0 * system.layout.count),
                                 "achieved_GBps": (28.0 * N if world == 1 else (16.0 * N + 12.0 * system.layout.count)) / kern_s_per_step / 1e9,
                                 "peak_GBps": HBM_PEAK_GBPS}},
        }
    if rank == 0 and world == 1 and N == 1 << 20 and args.dim == 3:
        # HBM traffic of the force kernel per launch: PMC counters cannot be read from inside the process; they come from
        # the committed rocprofv3 --pmc passes of this same command (tools/profile_bench.sh -> tools/summarize_prof.py ->
        # profiles/<round>/pmc_force_kernel.json), and only if that profile is of the kernel that just ran -- identified by the
        # symbol the LIBRARY reports for it (nbx_variant_kernel_symbol: the registered name, demangled as rocprofv3 prints it),
        # no hand-kept table to go stale when a template parameter is added -- otherwise the field stays null and says why.
        launched = launched_kernel_symbol(variant_name, args.dim, args.refine)
        result["roofline"]["kernel_symbol"] = launched
        seen = []
        for rnd in PROFILE_ROUNDS:
            prof = os.path.join("profiles", rnd, "pmc_force_kernel.json")
            try:
                with open(os.path.join(ROOT, prof)) as f:
                    pmc = json.load(f)
                t = pmc_traffic(pmc, launched)
            except Exception as e:
                seen.append(f"{prof}: {e.__class__.__name__}")
                continue
            if t is None:
                seen.append(f"{prof} is a profile of '{pmc['pmc_fetch']['kernel']['Kernel_Name']}'")
                continue
            result["roofline"]["traffic"] = t["bytes"]
            result["roofline"]["traffic_over_algorithmic"] = t["bytes"] / (28.0 * N)
            result["roofline"]["traffic_source"] = (f"{prof}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate) of this command and this "
                                                    f"kernel, per launch: {t['fetch_kb'] / 1024:.0f} MiB x {t['f_fetch']:.2f} fetched + {t['write_kb'] / 1024:.0f} MiB "
                                                    f"x {t['f_write']:.2f} written; factors calibrated on the helper kernels of the same profile "
                                                    "(known byte counts), = the guide's gfx950 FETCH_SIZE correction")
            break
        else:
            result["roofline"]["traffic_source"] = f"no committed PMC profile of the kernel this run launched ('{launched}'): " + "; ".join(seen)
    if rank == 0:
        result["launcher"] = os.environ.get("NBODY_BENCH_LAUNCHER", "torch.distributed.run started by the caller" if "WORLD_SIZE" in os.environ else "single process")
    if rank == 0 and world > 1:
        # self-description of the N > 1 path (never executed on hardware before the driver's scaling run)
        result["rccl_ranks"] = dist.get_world_size() if backend == "nccl" else 0
        result["exchange_transport"] = backend
        result["exchange_check"] = exchange_check
        result["per_rank"] = per_rank
        result["upload"] = {"bytes_over_this_ranks_host_link": be.upload_bytes, "whole_array_bytes": int(bodies.size * 8),
                            "how": "own shard host-to-device; other shards' masses by one all-gather, positions by the step's exchange"}
        result["exchange_hidden_behind_local_pass"] = all(bool(r["exchange_hidden"]) for r in per_rank)
    if world > 1:
        dist.barrier()
    acc = None
    if not args.no_cpu_baseline:   # parity of THIS run, whatever the rank count (every rank takes part; rank 0 holds the result)
        acc = parity_block(system, bodies, G, args, world, rank, dist)
    if world > 1:
        dist.barrier()             # the last collective: rank 0's CPU-side work below holds nobody in a barrier
    if rank == 0:
        result["precision"] = {"mode": "mixed" if args.refine else "plain fp32", "per_body_relative_tolerance": args.refine or None,
                               "means": "fp32 pair terms and sums for every body; bodies whose fp32 sum cannot be trusted to the tolerance "
                                        "(selection rule of nbx_ctx_set_refine) re-evaluated in fp64 inside every timed step" if args.refine
                                        else "fp32 pair terms and sums only"}
        result["tolerance_met_for_all_bodies"] = None
        if acc is not None:
            result["accuracy"] = acc
            ab = acc.get("all_bodies")
            if ab:
                result["tolerance_met_for_all_bodies"] = bool(ab[ab["headline_mode"]]["n_over_1e-5"] == 0)
            if world > 1:
                result["accuracy"]["nccl_parity_path"] = "first contact: this N > 1 parity path had only been rehearsed over gloo on one GPU before this run"
            other = None
            if world == 1 and args.dim == 3:   # before the CPU baseline: 20 s of host-only work send the GPU's clocks down, and a 0.3-ms kernel does not bring them back
                try:
                    other = {"leaf_pair": leaf_pair_roofline(local_rank)}
                except Exception as e:   # never at the expense of the headline line
                    other = {"leaf_pair": f"not measured: {e.__class__.__name__}: {e}"}
            result["cpu_baseline"] = cpu_baseline(N, args.dim, args.seed, n_sample=args.cpu_baseline_n)   # rank 0's host cores, after the last collective
            if other is not None:
                result["other_kernels"] = other
        print(json.dumps(result), flush=True)
    be.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
