"""GPU: the single-process multi-GPU node (nbx_node_*, csrc/nbx_node.hip) rehearsed on ONE device with
virtual ranks (the same device listed several times): peer-copy exchange between the ranks' buffers,
LOCAL || exchange then REMOTE ordering, kick/drift, energy -- against the oracle and against a single
context.  The RCCL exchange needs distinct devices; with one rank it is exercised for loading, communicator
creation and teardown (ncclCommInitAll on one device)."""
import numpy as np
import pytest

from oracle_lib import assert_force_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ranks,n,dim", [(2, 5000, 3), (3, 4099, 2), (8, 9000, 3)])
def test_virtual_ranks_forces_and_steps(nbx, oracle, ranks, n, dim):
    b = oracle.round_inputs_to_f32(oracle.generate(31, n, dim))
    ref = oracle.brute_force_seq(b)
    S = oracle.force_magnitude_sums(b)
    gscale = 1e24
    G = oracle.G * gscale
    with nbx.Node(n, dim, [0] * ranks) as node:
        assert node.exchange == nbx.EXCHANGE_PEER_COPY        # AUTO falls back: the ranks share a device
        node.upload(b)
        assert node.verify_exchange() == 0                    # poisoned-buffer self-check of the exchange (peer copies here)
        f = node.forces(oracle.G)
        assert f.shape == (n, dim)
        assert_force_parity(f, ref, S, f"node, {ranks} virtual ranks")
        steps, dt = 4, 2.0
        node.step(dt, steps, G)
        node.synchronize()
        got = b.copy()
        node.download(got)
        ke, pe = node.energy(G)
    cur = b.copy()
    for _ in range(steps):
        ff = oracle.brute_force_seq(oracle.round_inputs_to_f32(cur)) * gscale
        oracle.update_body_velocities(cur, np.ascontiguousarray(ff), dt)
        oracle.update_body_positions(cur, dt)
    d = dim
    moved = np.abs(cur[:, d:2 * d] - b[:, d:2 * d]).max()
    assert moved > 1e-6, "coupling too weak to detect a stale exchange"
    assert np.allclose(got[:, d:2 * d], cur[:, d:2 * d], rtol=0, atol=3e-5 * moved)
    assert np.allclose(got[:, :d], cur[:, :d], rtol=1e-9, atol=3e-5 * moved * dt * steps)
    r32 = oracle.round_inputs_to_f32(got)
    r32[:, d:2 * d] = got[:, d:2 * d]
    ke_ref, pe_ref = oracle.energy(r32)
    assert abs(ke - ke_ref) <= 1e-12 * ke_ref and abs(pe - pe_ref * gscale) <= 3e-6 * pe_ref * gscale


def test_node_matches_single_context_trajectory(nbx, oracle):
    n, dim = 6000, 3
    b = oracle.round_inputs_to_f32(oracle.generate(32, n, dim))
    G = oracle.G * 1e24
    one = b.copy()
    nbx.leapfrog_hip_n_body(one, 1.5, 6, G)
    with nbx.Node(n, dim, [0, 0, 0, 0], nbx.EXCHANGE_PEER_COPY) as node:
        node.upload(b)
        node.step(1.5, 6, G)
        many = b.copy()
        node.download(many)
    dv = np.abs(one[:, 3:6] - b[:, 3:6]).max()
    assert np.allclose(many[:, 3:6], one[:, 3:6], rtol=0, atol=2e-5 * dv)   # LOCAL+REMOTE sums vs one ALL sum
    assert np.allclose(many[:, :3], one[:, :3], rtol=1e-9, atol=2e-5 * dv * 1.5 * 6)


def test_rccl_single_rank_and_argument_checks(nbx, oracle):
    n, dim = 3000, 3
    b = oracle.round_inputs_to_f32(oracle.generate(33, n, dim))
    with nbx.Node(n, dim, [0], nbx.EXCHANGE_RCCL) as node:     # dlopen(librccl), ncclCommInitAll({0}), teardown
        assert node.exchange == nbx.EXCHANGE_RCCL
        node.upload(b)
        f = node.forces(oracle.G)
        node.step(1.0, 2)
        node.synchronize()
    assert_force_parity(f, oracle.brute_force_seq(b), oracle.force_magnitude_sums(b), "node, one rank")
    with pytest.raises(nbx.NbxError):
        nbx.Node(n, dim, [0, 0], nbx.EXCHANGE_RCCL)              # RCCL needs distinct devices
    with pytest.raises(nbx.NbxError):
        nbx.Node(n, dim, [0, 99])                                # no such device
    lib = nbx.load_library()
    assert lib.nbx_node_step(None, 1.0, 1.0, 1) == 1 and lib.nbx_node_destroy(None) == 0


def test_parked_streams_and_communicators_are_reused_and_released(nbx, oracle):
    """Destroyed contexts / nodes park their streams and RCCL communicators for the next ones (nbx_release_cached gives
    them back): many one-shot calls, a node that takes the previous node's communicator, a release in between -- results
    stay the same throughout."""
    import time
    n, dim = 2000, 3
    b = oracle.round_inputs_to_f32(oracle.generate(35, n, dim))
    lib = nbx.load_library()
    first = nbx.brute_force_hip_n_body(b)
    t0 = time.perf_counter()
    for _ in range(40):
        assert np.array_equal(nbx.brute_force_hip_n_body(b), first)
    per_call = (time.perf_counter() - t0) / 40
    print(f"one-shot call at N={n}: {per_call * 1e3:.3f} ms")
    # (3.2 ms per call when every call made and destroyed a stream; no assertion on a shared box's clock)
    for round_ in range(3):
        with nbx.Node(n, dim, [0], nbx.EXCHANGE_RCCL) as node:      # rounds 1, 2 take the communicator parked by the one before
            node.upload(b)
            assert np.array_equal(node.forces(oracle.G), first)
        if round_ == 1:
            assert lib.nbx_release_cached() == 0                   # round 2 has to build a new one
    with nbx.Node(n, dim, [0, 0, 0], nbx.EXCHANGE_PEER_COPY) as node:   # comm streams come from the pool as well
        node.upload(b)
        f3 = node.forces(oracle.G)
    assert_force_parity(f3, oracle.brute_force_seq(b), oracle.force_magnitude_sums(b), "three virtual ranks after a release")
    assert lib.nbx_release_cached() == 0
    assert np.array_equal(nbx.brute_force_hip_n_body(b), first)


def test_pass_times_of_a_sharded_evaluation(nbx, oracle):
    """nbx_node_enable_timing / nbx_node_pass_times: events around every rank's LOCAL and REMOTE pass and its part of the
    exchange -- what `nbody_sim --gpus G` prints as per_rank lines (the facts bench.py --gpus N carries in its JSON line).
    Timing must not change a single bit of the forces."""
    n, dim, ranks = 60000, 3, 4
    b = oracle.round_inputs_to_f32(oracle.generate(35, n, dim))
    with nbx.Node(n, dim, [0] * ranks) as node:
        node.upload(b)
        f0 = node.forces(oracle.G)
        with pytest.raises(nbx.NbxError):
            node.pass_times(0)                      # nothing timed yet
        node.enable_timing(True)
        f1 = node.forces(oracle.G)
        assert np.array_equal(f0, f1)
        times = [node.pass_times(r) for r in range(ranks)]
        with pytest.raises(nbx.NbxError):
            node.pass_times(ranks)
        node.step(1.0, 2, oracle.G)                 # steps are timed too: the last evaluation's figures replace the first's
        again = node.pass_times(1)
    assert [t["rank"] for t in times] == list(range(ranks)) and sum(t["targets"] for t in times) == n
    for t in times + [again]:
        assert t["device"] == 0 and t["local_ms"] > 0 and t["remote_ms"] > 0 and t["exchange_ms"] >= 0   # (virtual ranks share a GPU: no ordering of the two)
        assert isinstance(t["exchange_hidden"], bool)
    with nbx.Node(5000, dim, [0]) as one:           # one rank: no passes to tell apart, zeros and no error
        one.upload(b[:5000])
        one.enable_timing(True)
        one.forces(oracle.G)
        t = one.pass_times(0)
        assert t["targets"] == 5000 and t["local_ms"] == 0 and t["exchange_ms"] == 0
