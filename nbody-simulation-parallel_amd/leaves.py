"""Leaf lists for the leaf-pair direct-sum entry point (nbx_leaf_pair_forces).

The reference's tree codes build their leaves by recursive subdivision and collect, per leaf, the adjacent
leaves whose bodies are summed directly (FMM neighbour lists: nbody-sim-new/fmm.cpp:455-476,
fmm_parlay.cpp:369-390).  The device entry point takes that structure as two CSR arrays; this module builds the
simplest instance of it -- a fixed-depth subdivision of the bounding box (a uniform grid of 2^depth cells per
axis; the non-empty cells are the leaves) with the 3^D adjacent cells as each leaf's list -- for tests, examples
and timing.  Pure integer/numpy host logic; any other tree can feed the same arrays."""
from __future__ import annotations

import numpy as np


def uniform_grid_leaves(bodies: np.ndarray, dim: int, depth: int):
    """Returns (leaf_offsets, leaf_bodies, list_offsets, list_sources), all uint32.
    Leaf = non-empty cell of the 2^depth-per-axis grid over the bodies' bounding box (padded by 1 %, like the
    reference's root box, fmm.cpp:386-387); list = the leaf itself first, then its non-empty adjacent cells."""
    pos = np.asarray(bodies)[:, :dim]
    n = pos.shape[0]
    g = 1 << depth
    if n == 0:
        z = np.zeros(1, dtype=np.uint32)
        return z, np.zeros(0, dtype=np.uint32), z.copy(), np.zeros(0, dtype=np.uint32)
    lo, hi = pos.min(axis=0), pos.max(axis=0)
    centre, half = (lo + hi) / 2.0, max(float((hi - lo).max()) / 2.0 * 1.01, 1e-300)
    cell = np.clip(np.floor((pos - (centre - half)) / (2.0 * half) * g).astype(np.int64), 0, g - 1)
    key = np.zeros(n, dtype=np.int64)
    for d in range(dim):
        key = key * g + cell[:, d]
    order = np.argsort(key, kind="stable")
    keys, first = np.unique(key[order], return_index=True)
    leaf_offsets = np.append(first, n).astype(np.uint32)
    leaf_bodies = order.astype(np.uint32)
    index_of = {int(k): i for i, k in enumerate(keys)}
    # decode the cell coordinates of every leaf and look its 3^dim neighbours up
    coords = np.zeros((keys.size, dim), dtype=np.int64)
    rest = keys.copy()
    for d in range(dim - 1, -1, -1):
        coords[:, d] = rest % g
        rest //= g
    offs = np.stack(np.meshgrid(*[np.arange(-1, 2)] * dim, indexing="ij"), -1).reshape(-1, dim)
    offs = offs[np.any(offs != 0, axis=1)]
    list_offsets = [0]
    list_sources = []
    for l in range(keys.size):
        list_sources.append(l)                       # own bodies first (fmm_parlay.cpp:973-974)
        for o in offs:
            c = coords[l] + o
            if np.any(c < 0) or np.any(c >= g):
                continue
            k = 0
            for d in range(dim):
                k = k * g + int(c[d])
            j = index_of.get(k)
            if j is not None:
                list_sources.append(j)
        list_offsets.append(len(list_sources))
    return (leaf_offsets, leaf_bodies, np.asarray(list_offsets, dtype=np.uint32), np.asarray(list_sources, dtype=np.uint32))


def all_pairs_leaves(n: int, leaf_size: int):
    """Contiguous leaves of `leaf_size` bodies, every leaf on every list: the leaf-pair sum then equals the
    all-pairs sum (used to tie the leaf kernel to the brute-force oracle)."""
    n_leaves = max(1, -(-n // leaf_size))
    leaf_offsets = np.minimum(np.arange(n_leaves + 1) * leaf_size, n).astype(np.uint32)
    leaf_bodies = np.arange(n, dtype=np.uint32)
    list_offsets = (np.arange(n_leaves + 1) * n_leaves).astype(np.uint32)
    list_sources = np.tile(np.arange(n_leaves, dtype=np.uint32), n_leaves)
    return leaf_offsets, leaf_bodies, list_offsets, list_sources


def median_split_leaves(bodies: np.ndarray, dim: int, max_leaf_size: int = 16, reach: float = 1.0):
    """Leaves the way the reference's BVH makes them -- BVH<D>::build_recursive (nbody-sim-new/bvh.cpp:34-73): split the bodies at
    the median along the longest axis of their bounding box until a node holds at most max_leaf_size (methods.h:57: 16) -- with a
    near-field list per leaf: the leaf itself first, then every other leaf whose bounding box comes within `reach` x the leaf's own
    box diagonal of its box (box-to-box distance, the kind of acceptance test a traversal applies).  Leaves come out in tree
    order (depth first, lower half first), which keeps spatial neighbours close in leaf order.  Host-side numpy; a stand-in for a
    tree builder in tests and timing, not a port of the reference's pointer tree."""
    pos = np.asarray(bodies)[:, :dim]
    n = pos.shape[0]
    if n == 0:
        z = np.zeros(1, dtype=np.uint32)
        return z, np.zeros(0, dtype=np.uint32), z.copy(), np.zeros(0, dtype=np.uint32)
    order = np.arange(n)
    leaves = []
    stack = [(0, n)]
    while stack:
        lo, hi = stack.pop()
        if hi - lo <= max_leaf_size:
            leaves.append((lo, hi))
            continue
        idx = order[lo:hi]
        p = pos[idx]
        axis = int(np.argmax(p.max(axis=0) - p.min(axis=0)))
        mid = (hi - lo) // 2
        part = np.argpartition(p[:, axis], mid)
        order[lo:hi] = idx[part]
        stack.append((lo + mid, hi))      # popped second: the lower half's subtree comes out first
        stack.append((lo, lo + mid))
    leaves.sort()
    leaf_offsets = np.array([l for l, _ in leaves] + [n], dtype=np.uint32)
    nl = len(leaves)
    bmin = np.array([pos[order[l:h]].min(axis=0) for l, h in leaves])
    bmax = np.array([pos[order[l:h]].max(axis=0) for l, h in leaves])
    diag = np.linalg.norm(bmax - bmin, axis=1)
    centre = 0.5 * (bmin + bmax)
    from scipy.spatial import cKDTree
    tree = cKDTree(centre)
    # candidates by centre distance (a superset), then the exact box-to-box distance
    half = 0.5 * diag
    cand = tree.query_ball_point(centre, r=(1.0 + reach) * diag + half.max())
    list_offsets, list_sources = [0], []
    for l in range(nl):
        c = np.asarray(cand[l], dtype=np.int64)
        c = c[c != l]
        gap = np.maximum(0.0, np.maximum(bmin[c] - bmax[l], bmin[l] - bmax[c]))
        near = c[np.linalg.norm(gap, axis=1) <= reach * diag[l]]
        list_sources.append(l)
        list_sources.extend(np.sort(near).tolist())
        list_offsets.append(len(list_sources))
    return (leaf_offsets, order.astype(np.uint32), np.asarray(list_offsets, dtype=np.uint32), np.asarray(list_sources, dtype=np.uint32))
