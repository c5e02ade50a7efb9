"""GPU: pairs below the close-set thresholds whose two members live in DIFFERENT shards must come out like the
reference's guarded sequential sum (methods.cpp:24: `if (dist_sq < 1e-10) continue;`) through every sharded
driver: Context passes (ALL and LOCAL+REMOTE), the single-process node with virtual ranks, and the
one-process-per-rank path (two ranks on the one GPU, exchange staged over gloo).  Input: cross_shard_case.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import cross_shard_case as csc
from oracle_lib import assert_force_parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference(oracle, b):
    return oracle.brute_force_seq(b), oracle.force_magnitude_sums(b)


def _check_special(f, ref, special, lo, hi, what):
    """The planted bodies individually: skipped / duplicate partners leave an ordinary far-field force (a few 1e-6
    relative in fp32), the counted close partner dominates its sum."""
    for i in special:
        if lo <= i < hi:
            err = np.linalg.norm(f[i - lo] - ref[i])
            assert err <= 5e-5 * np.linalg.norm(ref[i]), (what, int(i), f[i - lo], ref[i])


@pytest.mark.parametrize("dim", (3, 2))
@pytest.mark.parametrize("n_shards,n", [(2, 6000), (3, 10000), (8, 33001)])
def test_context_passes(nbx, oracle, dim, n_shards, n):
    b, special = csc.make_bodies(nbx, n, dim, n_shards)
    assert special.size >= 6
    ref, S = _reference(oracle, b)
    # the planted pairs are what they claim to be
    e = csc.shard_len(n, n_shards)
    r2 = lambda i, j: float(((b[i, :dim] - b[j, :dim]) ** 2).sum())
    assert 0 < r2(e - 1, e) < 1e-10 and 1e-10 <= r2(e - 2, e + 1) < 2.4e-7 and r2(e - 3, e + 2) == 0.0
    for r in range(n_shards):
        with nbx.Context(n, dim, n_shards=n_shards, shard=r) as c:
            c.upload(b)
            assert c.effective_tuning()[0].startswith("fast"), "the unguarded fast path itself must be what runs"
            lo, hi = r * c.shard_len, r * c.shard_len + c.count
            c.compute_accel(nbx.SRC_ALL)
            fa = c.forces(oracle.G)
            assert_force_parity(fa, ref[lo:hi], S[lo:hi], f"ALL pass, shard {r}/{n_shards} D={dim}")
            _check_special(fa, ref, special, lo, hi, "ALL")
            c.compute_accel(nbx.SRC_LOCAL)
            c.compute_accel(nbx.SRC_REMOTE)
            fb = c.forces(oracle.G)
            assert_force_parity(fb, ref[lo:hi], S[lo:hi], f"LOCAL+REMOTE passes, shard {r}/{n_shards} D={dim}")
            _check_special(fb, ref, special, lo, hi, "LOCAL+REMOTE")
            # same passes again on unchanged positions: the rebuilt lists give bit-identical results
            c.compute_accel(nbx.SRC_LOCAL)
            c.compute_accel(nbx.SRC_REMOTE)
            assert np.array_equal(fb, c.forces(oracle.G))


@pytest.mark.parametrize("dim", (3, 2))
@pytest.mark.parametrize("ranks,n", [(2, 6000), (4, 9001)])
def test_node_virtual_ranks(nbx, oracle, dim, ranks, n):
    b, special = csc.make_bodies(nbx, n, dim, ranks)
    ref, S = _reference(oracle, b)
    with nbx.Node(n, dim, [0] * ranks) as node:
        node.upload(b)
        f = node.forces(oracle.G)
        assert_force_parity(f, ref, S, f"node, {ranks} virtual ranks D={dim}")
        _check_special(f, ref, special, 0, n, "node")
        # one step with a strong coupling: the skipped partner must not kick its neighbour (m/r^4 ~ 1e25 if it did)
        G = oracle.G * 1e20
        node.step(1.0, 1, G)
        got = b.copy()
        node.download(got)
    cur = b.copy()
    oracle.update_body_velocities(cur, np.ascontiguousarray(ref * 1e20), 1.0)
    oracle.update_body_positions(cur, 1.0)
    d = dim
    dv = np.abs(cur[:, d:2 * d] - b[:, d:2 * d])
    assert np.isfinite(got).all()
    assert np.allclose(got[:, d:2 * d] - b[:, d:2 * d], cur[:, d:2 * d] - b[:, d:2 * d], rtol=1e-4, atol=4e-6 * dv.max())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, dim, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nbody_amd as nbx
        import cross_shard_case
        bodies, _ = cross_shard_case.make_bodies(nbx, n, dim, world)
        system = nbx.package.dist.make_hip_system(bodies, dim, rank=rank, world_size=world, device_index=0)
        system.compute_forces()
        system.be.synchronize()
        f0 = system.forces(nbx.REFERENCE_G)
        lo, hi = system.layout.bounds()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), f0=f0, lo=lo, hi=hi, tuning=np.array(system.be.ctx.effective_tuning()[0]))
        system.be.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dim", (3, 2))
def test_two_ranks_one_gpu(tmp_path, nbx, oracle, dim):
    world, n = 2, 7001
    mp.spawn(_worker, args=(world, _free_port(), n, dim, str(tmp_path)), nprocs=world, join=True)
    b, special = csc.make_bodies(nbx, n, dim, world)
    ref, S = _reference(oracle, b)
    for r in range(world):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        lo, hi = int(z["lo"]), int(z["hi"])
        assert str(z["tuning"]).startswith("fast")
        assert_force_parity(z["f0"], ref[lo:hi], S[lo:hi], f"rank {r}/{world} D={dim}")
        _check_special(z["f0"], ref, special, lo, hi, f"rank {r}")
