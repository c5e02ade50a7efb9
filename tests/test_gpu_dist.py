"""GPU: the one-process-per-rank path end to end on the single-GPU box: two ranks share cuda:0, each
owns one shard through HipShardBackend (HIP kernels, torch streams, the library launching on the torch
compute stream, torch-owned exchange buffers).  The position exchange is staged over gloo here because a
single device cannot host two RCCL ranks; on a multi-GPU node the same code takes the
all_gather_into_tensor(nccl) branch of HipShardBackend.start_exchange."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, dim, steps, dt, gscale, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nbody_amd as nbx
        bodies = nbx.uniform_bodies(n, dim, 77)
        bodies[:, :dim] = bodies[:, :dim].astype(np.float32)
        bodies[:, -1] = bodies[:, -1].astype(np.float32)
        G = nbx.REFERENCE_G * gscale
        system = nbx.package.dist.make_hip_system(bodies, dim, rank=rank, world_size=world, device_index=0)
        system.compute_forces()
        system.be.synchronize()
        f0 = system.forces(G)
        system.step(dt, G, steps)
        system.be.synchronize()
        final = system.gather_bodies(bodies)
        lo, hi = system.layout.bounds()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), f0=f0, final=final, lo=lo, hi=hi,
                 tuning=np.array(system.be.ctx.effective_tuning()[0]))
        system.be.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,dim", [(2, 6000, 3), (3, 5001, 2)])
def test_two_ranks_one_gpu(tmp_path, oracle, world, n, dim):
    from oracle_lib import assert_force_parity
    steps, dt, gscale = 3, 2.0, 1e24
    mp.spawn(_worker, args=(world, _free_port(), n, dim, steps, dt, gscale, str(tmp_path)), nprocs=world, join=True)
    bodies = oracle.round_inputs_to_f32(oracle.generate(77, n, dim))
    ref_f0 = oracle.brute_force_seq(bodies) * gscale
    S = oracle.force_magnitude_sums(bodies) * gscale
    ref = bodies.copy()
    for _ in range(steps):
        f = oracle.brute_force_seq(oracle.round_inputs_to_f32(ref)) * gscale
        oracle.update_body_velocities(ref, np.ascontiguousarray(f), dt)
        oracle.update_body_positions(ref, dt)
    finals = []
    for r in range(world):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        lo, hi = int(z["lo"]), int(z["hi"])
        assert_force_parity(z["f0"], ref_f0[lo:hi], S[lo:hi], f"rank {r} forces")
        finals.append(z["final"])
    for fin in finals[1:]:
        assert np.array_equal(fin, finals[0])
    d = dim
    moved = np.abs(ref[:, d:2 * d] - bodies[:, d:2 * d]).max()
    assert moved > 1e-6
    assert np.allclose(finals[0][:, d:2 * d], ref[:, d:2 * d], rtol=0, atol=3e-5 * moved)
    assert np.allclose(finals[0][:, :d], ref[:, :d], rtol=1e-9, atol=3e-5 * moved * dt * steps)


def test_bench_line_for_two_ranks_carries_parity(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per process), rehearsed on the one
    GPU with the gloo transport: the JSON line of an N > 1 run must prove correctness, not only delivery -- the exchange
    self-check, sampled rows of the SHARDED evaluation against the oracle (here all rows: N <= 65,536), every body of every
    shard against the strict fp64 kernel in default and mixed mode, and the CPU baseline beside it."""
    import json
    import subprocess
    env = dict(os.environ, NBODY_BENCH_BACKEND="gloo", NBODY_BENCH_DEVICE="0", OMP_NUM_THREADS="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--bodies", "32768"]
    p = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["exchange_check"]["mismatching_values"] == 0
    a = r["accuracy"]
    assert a["ranks"] == 2 and a["rows"] == 32768 and a["max_backward_err"] <= 1e-5
    ab = a["all_bodies"]
    assert ab["checked"] == 32768 and ab["yardstick_vs_oracle_rows"]["max_rel"] <= 1e-9
    assert ab["default_fp32"]["max_backward"] <= 1e-5
    assert ab["mixed_mode"]["n_over_1e-5"] == 0 and ab["mixed_mode"]["max_rel"] <= 1e-5
    assert r["cpu_baseline"]["value"] > 0 and len(r["per_rank"]) == 2


_RCCL_ONE_RANK = r"""
import os, sys, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))   # bench.py's call
assert dist.get_backend() == "nccl"
compute, comm = torch.cuda.Stream(), torch.cuda.Stream()
pos_all = torch.full((1, 3, 4096), float("nan"), device="cuda")                         # [n_shards][dim][pad], poisoned
with torch.cuda.stream(compute):
    pos_all[0].copy_(torch.arange(3 * 4096, device="cuda", dtype=torch.float32).view(3, 4096))
comm.wait_stream(compute)                                                               # HipShardBackend.start_exchange
with torch.cuda.stream(comm):
    work = dist.all_gather_into_tensor(pos_all.view(-1), pos_all[0].view(-1), async_op=True)   # in place: src = own chunk
with torch.cuda.stream(comm):                                                           # finish_exchange
    work.wait()
compute.wait_stream(comm)
torch.cuda.synchronize()
assert torch.equal(pos_all.view(-1).cpu(), torch.arange(3 * 4096, dtype=torch.float32))
flag = torch.ones(1, device="cuda")
dist.all_reduce(flag)                                                                   # the self-check's device collective
assert float(flag) == 1.0
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")                             # bench.py: max of the ranks' elapsed time
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
gathered = [None]
dist.all_gather_object(gathered, {"rank": 0, "ms": [1.0, 2.0]})                         # bench.py: per-rank pass times, parity rows
assert gathered[0]["ms"] == [1.0, 2.0]
dist.barrier()
dist.destroy_process_group()
print("rccl one rank ok")
"""


def test_rccl_call_pattern_with_one_rank():
    """What one GPU can show of the RCCL branch before a multi-GPU node exists: torch's nccl backend initialises with bench.py's
    arguments, and the exchange's call pattern -- wait on the compute stream, in-place all_gather_into_tensor of the own chunk with
    async_op on the comm stream, wait, hand back to the compute stream -- runs and leaves the buffer intact.  One rank moves no
    data between devices: the wire stays unexercised."""
    import subprocess
    p = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, str(_free_port())], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "rccl one rank ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
