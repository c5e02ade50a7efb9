"""Per-rank cost of the sharded step on ONE GPU: a context with n_shards=G, shard=r runs its LOCAL and
REMOTE force passes + kick/drift exactly as one rank of a G-GPU run would (without the all-gather)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nbx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
# optional: source-slice counts to try per G, e.g. "8:16,24,32,48,64 4:16,32" (0 = the library's automatic choice)
sweeps = {int(a.split(":")[0]): [int(x) for x in a.split(":")[1].split(",")] for a in sys.argv[2:]}
b = nbx.uniform_bodies(n, 3, 1)
base = None
for G, S in [(G, S) for G in (1, 2, 4, 8) for S in sweeps.get(G, [0])]:
    with nbx.Context(n, 3, n_shards=G, shard=G // 2) as c:
        c.set_tuning(S, -1)
        c.upload(b)
        def step():
            if G == 1:
                c.compute_accel(nbx.SRC_ALL)
            else:
                c.compute_accel(nbx.SRC_LOCAL); c.compute_accel(nbx.SRC_REMOTE)
            c.kick_drift(1.0)
        step(); c.synchronize(); c.kernel_time()
        t0 = time.perf_counter()
        K = 5
        for _ in range(K): step()
        c.synchronize()
        dt = (time.perf_counter() - t0) / K
        ms, cnt = c.kernel_time()
        base = base or dt
        planes = c.effective_tuning()[1]
        print(f"G={G} S={planes:3d} (partial sums written per step: {12 * planes * c.shard_pad / 1e6:6.1f} MB): wall {dt*1e3:8.3f} ms/step (ideal {base/G*1e3:8.3f}, efficiency {base/G/dt*100:5.1f}%)  main-kernel sum {ms*cnt/K:8.3f} ms/step  tuning {c.effective_tuning()}", flush=True)
