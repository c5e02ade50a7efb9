"""Near-field stepping through the resident plan at N = 2^20 (32^3 grid leaves): k x {nbx_leaf_plan_forces_ctx; nbx_leaf_plan_kick_drift}
against nbx_leaf_plan_step (the same steps in one call).
    python tools/time_leaf_steps.py [N] [grid depth]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nbody_amd as nbx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 5
b = nbx.uniform_bodies(n, 3, 5)
leaves = nbx.leaves.uniform_grid_leaves(b, 3, depth)
G = 4.471e-21
with nbx.LeafPlan(n, 3, *leaves) as plan, nbx.Context(n, 3) as c:
    c.upload(b); c.synchronize()
    def t(f):
        c.synchronize(); t0 = time.perf_counter(); f(); c.synchronize(); return (time.perf_counter() - t0) * 1e3
    def calls(k):
        for _ in range(k):
            plan.forces_ctx(c, 1, G, fetch=False); plan.kick_drift(c, 1.0)
    print("two calls x 1: %.3f ms" % t(lambda: calls(1)))
    print("two calls x 20: %.3f ms" % t(lambda: calls(20)))
    print("two calls x 20: %.3f ms" % t(lambda: calls(20)))
    print("step(1): %.3f ms" % t(lambda: plan.step(c, 1, G, 1.0, 1)))
    print("step(1) again: %.3f ms" % t(lambda: plan.step(c, 1, G, 1.0, 1)))
    print("step(20): %.3f ms" % t(lambda: plan.step(c, 1, G, 1.0, 20)))
    print("step(20): %.3f ms" % t(lambda: plan.step(c, 1, G, 1.0, 20)))
    print("step(200): %.3f ms" % t(lambda: plan.step(c, 1, G, 1.0, 200)))
    print("two calls x 200: %.3f ms" % t(lambda: calls(200)))
