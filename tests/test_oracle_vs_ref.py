"""CPU, development container only (skipped where oracle/_ref is absent): the oracle against the
reference's own object code on fresh seeds and sizes beyond the committed fixtures."""
import numpy as np
import pytest


@pytest.mark.parametrize("dim", (2, 3))
@pytest.mark.parametrize("n,seed", [(1, 1), (5, 2), (257, 3), (2048, 4)])
def test_live_reference(oracle, reference, dim, n, seed):
    assert reference.G() == oracle.G
    b = oracle.generate(seed, n, dim)
    assert np.array_equal(b, reference.generate(seed, n, dim))
    assert np.array_equal(oracle.brute_force_seq(b), reference.brute_force(0, b))
    assert np.array_equal(oracle.brute_force_omp_2(b), reference.brute_force(2, b))
    f1, r1 = oracle.brute_force_omp_1(b), reference.brute_force(1, b)
    assert np.array_equal(f1, r1)  # same thread count in one process => same partition => same bits
    f = oracle.brute_force_seq(b)
    a, c = b.copy(), b.copy()
    oracle.update_body_velocities(a, f, 123.5)
    reference.update_body_velocities(c, f, 123.5)
    assert np.array_equal(a, c)
    oracle.update_body_positions(a, 123.5)
    reference.update_body_positions(c, 123.5)
    assert np.array_equal(a, c)
    noisy = f * (1 + 0.02 * np.sin(np.arange(f.size).reshape(f.shape)))
    assert oracle.compute_accuracy(noisy, f) == reference.compute_accuracy(noisy, f)


def test_layout(reference):
    assert reference.sizeof_body(3) == 56 and reference.sizeof_body(2) == 40  # body.h:8-11
