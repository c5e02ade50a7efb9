"""Near-field (leaf-pair) direct sums at FMM-like sizes: N bodies in a uniform grid of leaves (2^(3*level) cells), every
leaf against its 27-cell neighbourhood, through nbx_leaf_pair_forces (one-shot) and nbx_leaf_plan_* (resident); prints the kernel's own time per law.
    python tools/time_leaf_pairs.py [N] [level | bvh<K>] [box] [--dump DIR]
--dump DIR: write the structure's CSR arrays as raw uint32 files (for tools/time_leaf_layout.cpp) and stop.
level: uniform grid of 2^level cells per axis with 27-cell lists (5: 32 bodies per leaf at N = 2^20, 6: 4 bodies per leaf);
bvh<K> (e.g. bvh16): median-split leaves of at most K bodies, the reference BVH's own (bvh.cpp:34-73, methods.h:57), with
box-distance near-field lists (leaves.median_split_leaves).
box (default: the reference generator's 1e7) rescales the positions: in a box below 2^14 = 16,384 every target lies inside the
close set (csrc/nbx_internal.h) and every wave takes the kernel's guarded pair loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nbx

dump_dir = None
if "--dump" in sys.argv:
    k = sys.argv.index("--dump")
    dump_dir = sys.argv[k + 1]
    del sys.argv[k:k + 2]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
shape = sys.argv[2] if len(sys.argv) > 2 else "5"
b = nbx.uniform_bodies(n, 3, 5)
if len(sys.argv) > 3:
    b[:, :3] *= float(sys.argv[3]) / 1.0e7
t_build = time.perf_counter()
if shape.startswith("bvh"):
    leaves = nbx.leaves.median_split_leaves(b, 3, int(shape[3:] or 16), reach=0.5)
else:
    leaves = nbx.leaves.uniform_grid_leaves(b, 3, int(shape))
print(f"leaves '{shape}' built on the host in {time.perf_counter() - t_build:.1f} s (not part of any timing below)", flush=True)
if dump_dir:
    os.makedirs(dump_dir, exist_ok=True)
    for name, a in zip(("leaf_offsets", "leaf_bodies", "list_offsets", "list_sources"), leaves):
        np.asarray(a, dtype=np.uint32).tofile(os.path.join(dump_dir, name + ".u32"))
    print(f"arrays written to {dump_dir}")
    sys.exit(0)
lo, _, so, ss = leaves
sizes = np.diff(lo).astype(np.int64)
src = np.add.reduceat(sizes[ss], so[:-1])          # bodies on each leaf's list (every list here is non-empty)
pairs = int((sizes * src).sum())
print(f"N={n}, {sizes.size} leaves (mean {sizes.mean():.1f}, max {sizes.max()}), {pairs:.3e} pair terms", flush=True)
def timed(law):
    """Best kernel time and best whole-call time of three ONE-SHOT calls (nbx_leaf_pair_forces: validation, layout, H2D, gather,
    kernel, scatter, D2H per call)."""
    best, wall = 1e30, 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        _, ms = nbx.leaf_pair_forces_hip(b, *leaves, law=law, return_kernel_ms=True)
        wall = min(wall, (time.perf_counter() - t0) * 1e3)
        best = min(best, ms)
    return best, wall


def rate(ms):
    tflops = pairs * 20.0 / (ms * 1e-3) / 1e12      # 20 flop per pair term, the brute-force path's convention
    return f"{ms:.3f} ms = {pairs / ms * 1e3:.3e} pairs/s = {tflops:.1f} TFLOP/s = {tflops / 157.3:.3f} of the MI355X fp32 vector peak"


LAWS = ((nbx.LAW_BRUTE, "brute"), (nbx.LAW_TREE_LEAF, "tree_leaf"), (nbx.LAW_FMM_P2P, "fmm_p2p"))
cold = {name: timed(law) for law, name in LAWS}              # all of these before any sustained load
for law, name in LAWS:
    print(f"law {name:9s}: one-shot call: kernel, one launch after the call's host work and copies (clocks at ~2.05 GHz) {rate(cold[name][0])};  "
          f"whole call (validation, layout, H2D, gather, kernel, scatter, D2H) {cold[name][1]:.2f} ms", flush=True)
# the resident plan (nbx_leaf_plan_*): the structure validated, laid out and uploaded once, bodies resident in a context
# what a tree code that rebuilds its tree every step pays per step: a new plan for the new structure (validation, layout, uploads)
remake = 1e30
for _ in range(3):
    t0 = time.perf_counter()
    with nbx.LeafPlan(n, 3, *leaves):
        remake = min(remake, (time.perf_counter() - t0) * 1e3)
print(f"plan alone (nbx_leaf_plan_create for a structure of this shape, best of 3; the previous plan destroyed first): {remake:.2f} ms", flush=True)
t0 = time.perf_counter()
plan = nbx.LeafPlan(n, 3, *leaves)
ctx = nbx.Context(n, 3)
ctx.upload(b)
ctx.synchronize()
print(f"plan: created in {(time.perf_counter() - t0) * 1e3:.2f} ms (validation, layout, uploads, context); slots/runs/workgroups/waves {plan.info()}", flush=True)
for law, name in LAWS:
    walls = []
    for _ in range(30):
        ctx.synchronize()
        t0 = time.perf_counter()
        plan.forces_ctx(ctx, law, fetch=False)
        ctx.synchronize()
        walls.append((time.perf_counter() - t0) * 1e3)
    one = min(plan.forces_ctx(ctx, law, fetch=False, timed=True) for _ in range(3))
    print(f"law {name:9s}: plan, evaluation of the unchanged structure from resident bodies: wall median {np.median(walls):.3f} ms "
          f"(min {min(walls):.3f}); pair kernel, single launch {rate(one)}", flush=True)
for law, name in LAWS:
    warm = plan.time_kernel(law, 300)
    print(f"law {name:9s}: pair kernel, mean of launches 151-300 back to back (nbx_leaf_plan_time_kernel; clocks up, ~2.3 GHz) {rate(warm)}", flush=True)
