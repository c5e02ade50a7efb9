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
    # decode the cell coordinates of every leaf and look its 3^dim neighbours up (keys is sorted: a binary search per neighbour cell)
    coords = np.zeros((keys.size, dim), dtype=np.int64)
    rest = keys.copy()
    for d in range(dim - 1, -1, -1):
        coords[:, d] = rest % g
        rest //= g
    offs = np.stack(np.meshgrid(*[np.arange(-1, 2)] * dim, indexing="ij"), -1).reshape(-1, dim)
    offs = np.concatenate([np.zeros((1, dim), dtype=offs.dtype), offs[np.any(offs != 0, axis=1)]])   # own bodies first (fmm_parlay.cpp:973-974)
    found = np.full((keys.size, offs.shape[0]), -1, dtype=np.int64)
    for c, o in enumerate(offs):
        nb = coords + o
        inside = np.all((nb >= 0) & (nb < g), axis=1)
        k = np.zeros(keys.size, dtype=np.int64)
        for d in range(dim):
            k = k * g + nb[:, d]
        at = np.minimum(np.searchsorted(keys, k), keys.size - 1)
        hit = inside & (keys[at] == k)
        found[hit, c] = at[hit]
    present = found >= 0
    list_offsets = np.concatenate([[0], np.cumsum(present.sum(axis=1))])
    list_sources = found[present]                                    # row-major: every leaf's cells in the order of `offs`
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
    # level by level: the nodes of a level have at most two different sizes (a node of s bodies splits into s // 2 and s - s // 2),
    # and all nodes of one size are split in one batched call
    order = np.arange(n)
    nodes = np.array([[0, n]], dtype=np.int64)              # [lo, hi) of the current level's nodes, in tree order
    done = []
    while nodes.size:
        size = nodes[:, 1] - nodes[:, 0]
        leaf = size <= max_leaf_size
        done.append(nodes[leaf])
        nodes = nodes[~leaf]
        size = size[~leaf]
        nxt = np.empty((2 * nodes.shape[0], 2), dtype=np.int64)
        for s_ in np.unique(size):
            sel = np.nonzero(size == s_)[0]
            at = nodes[sel, 0][:, None] + np.arange(s_)[None, :]          # [k, s] positions in `order`
            idx = order[at]
            p = pos[idx]                                                # [k, s, dim]
            axis = np.argmax(p.max(axis=1) - p.min(axis=1), axis=1)     # longest axis of every node's box
            val = np.take_along_axis(p, axis[:, None, None], axis=2)[:, :, 0]
            mid = int(s_) // 2
            part = np.argpartition(val, mid, axis=1)
            order[at] = np.take_along_axis(idx, part, axis=1)
            nxt[2 * sel, 0] = nodes[sel, 0]; nxt[2 * sel, 1] = nodes[sel, 0] + mid          # lower half first
            nxt[2 * sel + 1, 0] = nodes[sel, 0] + mid; nxt[2 * sel + 1, 1] = nodes[sel, 1]
        nodes = nxt
    leaves = np.concatenate(done)
    leaves = leaves[np.argsort(leaves[:, 0], kind="stable")]
    leaf_offsets = np.append(leaves[:, 0], n).astype(np.uint32)
    nl = leaves.shape[0]
    sorted_pos = pos[order]
    bmin = np.minimum.reduceat(sorted_pos, leaves[:, 0], axis=0)
    bmax = np.maximum.reduceat(sorted_pos, leaves[:, 0], axis=0)
    diag = np.linalg.norm(bmax - bmin, axis=1)
    centre = 0.5 * (bmin + bmax)
    from scipy.spatial import cKDTree
    # candidates by centre distance (a superset: box gap >= centre distance - the two half diagonals), then the exact box-to-box
    # distance.  Leaves are put into buckets of like diagonal (ratio 1.25) and every pair of buckets is queried once with the
    # radius its largest members need -- the result is arrays (a ball query per leaf returns Python lists: slow for a million bodies)
    half = 0.5 * diag
    d_min = max(float(diag.min()), 1e-300 + float(diag.max()) * 1e-6)
    bucket = np.floor(np.log(np.maximum(diag, d_min) / d_min) / np.log(1.25)).astype(np.int64)
    ids_of = [np.nonzero(bucket == b_)[0] for b_ in np.unique(bucket)]
    trees = [cKDTree(centre[ids]) for ids in ids_of]
    rows_parts, c_parts = [np.zeros(0, dtype=np.int64)], [np.zeros(0, dtype=np.int64)]
    for ia, ta in zip(ids_of, trees):
        for ib, tb in zip(ids_of, trees):
            found = ta.sparse_distance_matrix(tb, float((0.5 + reach) * diag[ia].max() + half[ib].max()), output_type="ndarray")
            rows_parts.append(ia[found["i"]])
            c_parts.append(ib[found["j"]])
    rows, c = np.concatenate(rows_parts), np.concatenate(c_parts)
    other = c != rows
    rows, c = rows[other], c[other]
    gap = np.maximum(0.0, np.maximum(bmin[c] - bmax[rows], bmin[rows] - bmax[c]))
    near = np.sqrt((gap * gap).sum(axis=1)) <= reach * diag[rows]
    rows, c = rows[near], c[near]
    by_leaf = np.lexsort((c, rows))                          # every leaf's neighbours in ascending order
    rows, c = rows[by_leaf], c[by_leaf]
    counts = np.bincount(rows, minlength=nl)
    list_offsets = np.concatenate([[0], np.cumsum(counts + 1)])
    list_sources = np.empty(int(list_offsets[-1]), dtype=np.int64)
    list_sources[list_offsets[:-1]] = np.arange(nl)          # the leaf itself first
    rank = np.arange(rows.size) - np.repeat(np.cumsum(counts) - counts, counts)
    list_sources[list_offsets[rows] + 1 + rank] = c
    return (leaf_offsets, order.astype(np.uint32), np.asarray(list_offsets, dtype=np.uint32), np.asarray(list_sources, dtype=np.uint32))
