"""GPU: the drop-in entry points make and destroy a context per call (one device allocation, a stream from the library's
pool): many calls must not leak device memory, and calls from several host threads at once -- distinct contexts are
independent (SURVEY 8b "threading") -- must give the single-threaded answer."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_bytes():
    """hipMemGetInfo through the HIP runtime the library itself is linked to (already loaded: same SONAME)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    if hip.hipDeviceSynchronize() != 0 or hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) != 0:
        pytest.skip("hipMemGetInfo unavailable through this process's HIP runtime")
    return free.value


def test_repeated_calls_do_not_leak_device_memory(nbx, oracle):
    b = oracle.round_inputs_to_f32(oracle.generate(3, 20000, 3))
    leaves = nbx.leaves.uniform_grid_leaves(b, 3, 2)
    first = nbx.brute_force_hip_n_body(b)
    nbx.leaf_pair_forces_hip(b, *leaves)
    with nbx.Node(b.shape[0], 3, [0, 0], nbx.EXCHANGE_PEER_COPY) as node:
        node.upload(b)
        node.forces(oracle.G)
    before = _free_bytes()
    for i in range(150):
        assert np.array_equal(nbx.brute_force_hip_n_body(b), first)
        if i % 10 == 0:
            nbx.leaf_pair_forces_hip(b, *leaves)
            bb = b.copy()
            nbx.leapfrog_hip_n_body(bb, 1.0, 2)
            with nbx.Node(b.shape[0], 3, [0, 0, 0], nbx.EXCHANGE_PEER_COPY) as node:
                node.upload(b)
                node.step(1.0, 1)
                node.synchronize()
    after = _free_bytes()
    assert before - after < 64 << 20, f"device memory shrank by {(before - after) >> 20} MiB over 150 calls"
    assert nbx.load_library().nbx_release_cached() == 0
    assert np.array_equal(nbx.brute_force_hip_n_body(b), first)


def test_concurrent_calls_from_host_threads(nbx, oracle):
    sizes = (3000, 5000, 7000, 9000)
    bodies = [oracle.round_inputs_to_f32(oracle.generate(10 + i, n, 3 if i % 2 else 2)) for i, n in enumerate(sizes)]
    want = [nbx.brute_force_hip_n_body(b) for b in bodies]
    errors = []

    def worker(k):
        try:
            for _ in range(25):
                got = nbx.brute_force_hip_n_body(bodies[k])
                if not np.array_equal(got, want[k]):
                    errors.append(f"thread {k}: result differs from the single-threaded one")
                    return
        except Exception as e:                                   # noqa: BLE001 -- reported below
            errors.append(f"thread {k}: {e!r}")

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(sizes))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
