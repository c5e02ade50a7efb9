"""CPU: host-side logic of the package that needs no GPU: the numpy body generator reproduces the
reference generator's stream (= the oracle's = libstdc++'s), Plummer sampling is sane, the ctypes
layer validates shapes."""
import numpy as np
import pytest

from conftest import golden


@pytest.mark.parametrize("dim,n", [(2, 64), (3, 1024), (3, 3)])
def test_uniform_bodies_match_reference_stream(nbx, oracle, dim, n):
    g = golden(f"bf_D{dim}_N{n}.npz")
    b = nbx.uniform_bodies(n, dim, int(g["seed"]))
    assert np.array_equal(b, g["bodies"])
    assert np.array_equal(nbx.uniform_bodies(5000, dim, 99), oracle.generate(99, 5000, dim))


def test_plummer_bodies(nbx):
    b = nbx.plummer_bodies(20000, 3, seed=4, a=1e5)
    r = np.sqrt(((b[:, :3] - 5e6) ** 2).sum(axis=1))
    assert r.max() < 1e6 and 0.6e5 < np.median(r) < 2.0e5      # half-mass radius of a Plummer sphere = 1.305 a
    assert np.allclose(b[:, 6], 1e12 / 20000)
    assert abs(b[:, 3:6].mean()) < 0.05 * np.abs(b[:, 3:6]).max()
    assert np.abs(b[:, :3]).min() > 8192, "stays clear of the close-set region"


def test_body_stride_and_shapes(nbx):
    assert nbx.body_stride(3) == 7 and nbx.body_stride(2) == 5
    with pytest.raises(ValueError):
        nbx.uniform_bodies(4, 4)
