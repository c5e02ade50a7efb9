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
    # the ParlayLib twins (methods.cpp:139-224): parlay_2 sums each row in source order like omp_2 => same bits;
    # parlay_1 privatises per worker (scheduling-dependent partition) => re-association noise only
    assert np.array_equal(reference.brute_force(4, b), reference.brute_force(2, b))
    p1 = reference.brute_force(3, b)
    assert np.allclose(p1, r1, rtol=0, atol=1e-9 * np.abs(r1).max())
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
