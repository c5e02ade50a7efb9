"""Synthetic initial conditions for benchmarks and examples (host side, numpy).

`uniform_bodies` reproduces the reference generator (nbody-sim-new/utils.h:107-135) with an explicit
seed: std::mt19937(seed) feeding libstdc++'s std::uniform_real_distribution<double>, ranges
position U[1,1e7), velocity U[-10,10), mass U[1,1e8), draw order per body p0,v0,p1,v1,(p2,v2,) mass.
The stream is bit-identical to the C++ harness's `--seed` bodies (tests/test_host_logic.py)."""
from __future__ import annotations

import numpy as np


def _canonical53(raw: np.ndarray) -> np.ndarray:
    """libstdc++ generate_canonical<double,53> over mt19937: two 32-bit draws, low word first."""
    lo = raw[0::2].astype(np.float64)
    hi = raw[1::2].astype(np.float64)
    r = (lo + hi * 4294967296.0) / 18446744073709551616.0
    return np.where(r >= 1.0, np.nextafter(1.0, 0.0), r)


def uniform_bodies(n: int, dim: int = 3, seed: int = 1) -> np.ndarray:
    """float64 [n, 2*dim+1] array of Body<dim> = (position[dim], velocity[dim], mass)."""
    if dim not in (2, 3):
        raise ValueError("dim must be 2 or 3")
    bg = np.random.MT19937()
    bg._legacy_seeding(int(seed) & 0xFFFFFFFF)  # init_genrand(seed) == std::mt19937(seed)
    per_body = 2 * dim + 1
    u = _canonical53(bg.random_raw(2 * per_body * n).astype(np.uint64)).reshape(n, per_body)
    out = np.empty((n, per_body), dtype=np.float64)
    for d in range(dim):
        out[:, d] = u[:, 2 * d] * (10000000.0 - 1.0) + 1.0
        out[:, dim + d] = u[:, 2 * d + 1] * (10.0 - (-10.0)) + (-10.0)
    out[:, 2 * dim] = u[:, 2 * dim] * (100000000.0 - 1.0) + 1.0
    return out


def plummer_bodies(n: int, dim: int = 3, seed: int = 1, a: float = 1.0e5, total_mass: float = 1.0e12,
                   centre: float = 5.0e6, G: float = 4.471e-21) -> np.ndarray:
    """Plummer sphere (not in the reference; BASELINE config 5): equal masses, scale radius a, radii cut
    at 10 a, isotropic velocities from the Plummer distribution function (rejection sampling) scaled
    for the Newtonian potential.  Centred in the reference's box so coordinates stay far from 0."""
    rng = np.random.default_rng(seed)
    r = np.empty(n)
    todo = np.arange(n)
    while todo.size:
        u = rng.random(todo.size)
        rr = a / np.sqrt(np.maximum(u, 1e-300) ** (-2.0 / 3.0) - 1.0)
        ok = rr < 10.0 * a
        r[todo[ok]] = rr[ok]
        todo = todo[~ok]
    q = np.empty(n)
    todo = np.arange(n)
    while todo.size:
        x, y = rng.random(todo.size), 0.1 * rng.random(todo.size)
        ok = y <= x * x * (1.0 - x * x) ** 3.5
        q[todo[ok]] = x[ok]
        todo = todo[~ok]

    def directions(length):
        v = rng.normal(size=(n, dim))
        v /= np.sqrt((v ** 2).sum(axis=1))[:, None]
        return v * length[:, None]

    vesc = np.sqrt(2.0 * G * total_mass / np.sqrt(r * r + a * a))
    out = np.empty((n, 2 * dim + 1))
    out[:, :dim] = directions(r) + centre
    out[:, dim:2 * dim] = directions(q * vesc)
    out[:, 2 * dim] = total_mass / n
    return out
