"""CPU: the host side of the leaf-pair path (csrc/leaf_plan.h) -- padded source pairs, copy runs, the cut of every leaf into its
workgroup's pieces, the launch order -- compiled with g++ under AddressSanitizer and UBSan (sanitizers run on the CPU build only) and checked for its invariants on ragged
structures, without a GPU.
What the pair kernel computes from the plan is the GPU tests' business (tests/test_gpu_leaf_pairs.py)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAD = 0xFFFFFFFF


def _build_planner(tmp_path_factory, name, *defines):
    exe = str(tmp_path_factory.mktemp(name) / "leaf_plan_check")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", *defines,
                    os.path.join(ROOT, "tests", "leaf_plan_check.cpp"), "-o", exe], check=True)

    def run(workdir, lo, lb, so, ss):
        for name, a in (("leaf_offsets", lo), ("leaf_bodies", lb), ("list_offsets", so), ("list_sources", ss)):
            np.asarray(a, dtype=np.uint32).tofile(os.path.join(workdir, name + ".u32"))
        p = subprocess.run([exe, workdir], capture_output=True, text=True)
        assert p.returncode == 0, p.stdout + p.stderr
        out = {k: np.fromfile(os.path.join(workdir, k + ".u32"), dtype=np.uint32)
               for k in ("unit_off", "pslot_body", "ops", "op_off", "blocks", "pack_subs", "pack_blocks")}
        out["pack_subs"] = out["pack_subs"].reshape(-1, 4)      # op_lo, op_n, first, count
        out["pack_blocks"] = out["pack_blocks"].reshape(-1, 8)  # sub_lo, n_sub, w, P, trips, inv_w, longest, shape
        out["waves"] = int(p.stdout.split()[1])
        out["ops"] = out["ops"].reshape(-1, 2)          # (end, base)
        out["blocks"] = out["blocks"].reshape(-1, 8)    # op_lo, op_n, key, -, first0, count0, first1, count1
        return out
    return run


@pytest.fixture(scope="module")
def planner(tmp_path_factory):
    return _build_planner(tmp_path_factory, "leaf_plan")


@pytest.fixture(scope="module")
def planner_threads(tmp_path_factory):
    """The same planner with the layout's threads switched on from 64 list entries (the library: from 200,000)."""
    return _build_planner(tmp_path_factory, "leaf_plan_threads", "-DNBX_PLAN_THREADS_FROM=64", "-pthread")


@pytest.fixture(scope="module")
def planner_xcd(tmp_path_factory):
    """The same planner with the XCD-aware order switched on from 16 workgroups (the library: from 4,096), so that structures of test size reach it."""
    return _build_planner(tmp_path_factory, "leaf_plan_xcd", "-DNBX_XCD_ORDER_FROM=16")


def _structure(seed, sizes, list_len):
    rng = np.random.default_rng(seed)
    sizes = np.asarray(sizes)
    lo = np.concatenate([[0], np.cumsum(sizes)])
    n = int(lo[-1]) + 11
    lb = rng.permutation(n)[:lo[-1]]
    lists = []
    for t in range(sizes.size):
        k = list_len(t)
        l = rng.integers(0, sizes.size, k)
        if t % 2 == 0 and k >= 3:                       # runs of consecutive leaves: these must be merged
            start = int(rng.integers(0, max(1, sizes.size - 3)))
            l[:3] = [start, start + 1, start + 2]
        lists.append(l)
    so = np.concatenate([[0], np.cumsum([len(l) for l in lists])])
    ss = np.concatenate(lists) if lists else np.zeros(0, dtype=np.int64)
    return lo, lb, so, ss


def packs_inv_w(plan, sub_lo):
    rows = plan["pack_blocks"]
    return int(rows[rows[:, 0] == sub_lo][0, 5])


def n_ops_of_leaf(op_off, l):
    return int(op_off[l + 1]) - int(op_off[l])


def _check(plan, lo, lb, so, ss):
    sizes = np.diff(lo)
    unit_off, pslot_body, ops, op_off, blocks = (plan[k] for k in ("unit_off", "pslot_body", "ops", "op_off", "blocks"))
    # padded slots: every leaf a whole number of pairs, bodies in leaf order, one pad at the end of an odd leaf
    psz = np.diff(unit_off.astype(np.int64))
    assert unit_off[0] == 0 and (psz == (sizes + 1) // 2 * 2).all()
    for l in range(sizes.size):
        seg = pslot_body[unit_off[l]:unit_off[l + 1]]
        assert (seg[:sizes[l]] == lb[lo[l]:lo[l + 1]]).all() and (seg[sizes[l]:] == PAD).all()
    # copy runs: expanded, a leaf's runs are its list's leaves (empty ones dropped) unit by unit, in list order; `end` is the running length
    n_merged = 0
    for l in range(sizes.size):
        want = [np.arange(unit_off[s], unit_off[s + 1]) for s in ss[so[l]:so[l + 1]] if psz[s]]
        want = np.concatenate(want) if want else np.zeros(0, dtype=np.int64)
        runs = ops[op_off[l]:op_off[l + 1]].astype(np.int64)
        begin, got = 0, []
        for end, base in runs:
            assert end > begin
            got.append((np.arange(begin, end) + base) % (1 << 32))
            begin = end
        got = np.concatenate(got) if got else np.zeros(0, dtype=np.int64)
        assert got.size == want.size and (got == want).all(), l
        n_entries = sum(1 for s in ss[so[l]:so[l + 1]] if psz[s])
        n_merged += n_entries - runs.shape[0]
    # workgroups: every body a target of exactly one piece, pieces inside one leaf and at most 64 targets, the leaf's runs attached
    hit = np.zeros(unit_off[-1], dtype=np.int64)
    leaf_of_unit = np.repeat(np.arange(sizes.size), psz)
    waves = plan["waves"]
    nonempty = sizes[sizes > 0]
    assert waves == (1 if nonempty.size and nonempty.sum() // nonempty.size <= 20 else 2)
    for op_lo, op_n, key, _, f0, c0, f1, c1 in blocks.astype(np.int64):
        assert c0 <= 64 and c1 <= 64 and c0 + c1 >= 1 and (waves == 2 or c1 == 0)
        l = leaf_of_unit[f0 if c0 else f1]
        assert sizes[l] > 16 or n_ops_of_leaf(op_off, l) > 32 or not (nonempty.sum() <= 8 * nonempty.size)   # tiny-leaf structures pack those
        assert op_lo == op_off[l] and op_n == op_off[l + 1] - op_off[l]
        for f, c in ((f0, c0), (f1, c1)):
            if c:
                assert (leaf_of_unit[f:f + c] == l).all() and (pslot_body[f:f + c] != PAD).all()
                hit[f:f + c] += 1
        if c0 and c1:
            assert f1 == f0 + c0
    # packed waves (leaves of <= 16 bodies with <= 32 runs): K = 64 / w leaves side by side, each on its own w lanes
    subs, packs = plan["pack_subs"].astype(np.int64), plan["pack_blocks"].astype(np.int64)
    n_ops_of = np.diff(op_off.astype(np.int64))
    seen_sub = np.zeros(subs.shape[0], dtype=np.int64)
    for sub_lo, n_sub, w, P, trips, _, longest, _ in packs:
        assert w in (4, 6, 8, 16) and 1 <= n_sub <= 64 // w and 1 <= P <= 8
        mine = subs[sub_lo:sub_lo + n_sub]
        seen_sub[sub_lo:sub_lo + n_sub] += 1
        assert (mine[:, 3] >= 1).all() and (mine[:, 3] <= w).all() and (mine[:, 1] <= 32).all()
        # leaves share a wave with leaves of their size class (a lane holds two targets): 1-2 and 3-4 bodies on 4 lanes (4 and 2 lane
        # groups), 5-6 bodies on 6 lanes, 7-8 bodies on 8 lanes, 9-16 bodies on 16 lanes (2 groups each)
        cls = lambda c: 0 if c <= 2 else 1 if c <= 4 else 2 if c <= 6 else 3 if c <= 8 else 4
        k = cls(mine[0, 3])
        assert all(cls(c) == k for c in mine[:, 3]) and w == (4, 4, 6, 8, 16)[k] and P == (4, 2, 2, 2, 2)[k]
        assert (-(-mine[:, 3] // 2) * P <= w).all() and packs_inv_w(plan, sub_lo) == -(-65536 // w)
        assert all((lane * packs_inv_w(plan, sub_lo)) >> 16 == lane // w for lane in range(64))
        streams = [int(ops[o + k - 1][0]) if k else 0 for o, k, _, _ in mine]
        # every lane group walks the same number of source pairs: its share of the longest stream, an even number (the loop takes two at a time)
        assert longest == max(streams) and longest % 2 == 0
        assert trips == -(-(-(-(longest // 2) // P)) // 2) * 2 and trips * P * 2 >= longest
        for o, k, f, c in mine:
            l = leaf_of_unit[f]
            assert sizes[l] == c and f == unit_off[l] and o == op_off[l] and k == n_ops_of[l]
            assert (pslot_body[f:f + c] != PAD).all()
            hit[f:f + c] += 1
    assert (seen_sub == 1).all()
    if packs.shape[0] > 1:
        assert (np.diff(packs[:, 4]) <= 0).all()                # longest first
    # a leaf is packed exactly when the structure's leaves are small on average (<= 8 bodies), it is small and its list short
    packed_leaves = set(leaf_of_unit[subs[:, 2]].tolist()) if subs.shape[0] else set()
    tiny = bool(nonempty.size) and nonempty.sum() <= 8 * nonempty.size
    for l in range(sizes.size):
        assert (l in packed_leaves) == bool(tiny and 1 <= sizes[l] <= 16 and n_ops_of[l] <= 32), l
    assert (hit == (pslot_body != PAD)).all()
    # launch order: longest first (1024 duration classes)
    key = blocks[:, 2].astype(np.int64)
    if key.size > 1:
        cls = key * 1023 // max(int(key.max()), 1)
        assert (np.diff(cls) <= 0).all()
    return n_merged


def test_plan_of_ragged_structures(planner, tmp_path):
    cases = [
        (1, list(range(1, 71)) + [0, 0], lambda t: [1, 2, 3, 27, 64, 65, 129, 300][t % 8]),          # every piece cut, long lists
        (2, list(range(1, 260, 3)) + [0], lambda t: [5, 27, 0][t % 3]),                                # several workgroups per leaf, empty lists
        (3, list(range(1, 25)) * 3 + [30, 27], lambda t: 9),                                           # small leaves: one wave per workgroup
        (4, [0, 0, 0], lambda t: 2),                                                                   # nothing but empty leaves
        (5, [64, 128, 129, 1, 65], lambda t: 5),
        (6, [4, 3, 5, 0, 1, 2, 8, 9, 16, 7, 4, 4, 6, 2, 1, 3] * 9 + [40], lambda t: [3, 9, 27, 17][t % 4]),            # tiny leaves: packed by size class
        (7, [1, 2, 3, 4] * 20, lambda t: 2),
    ]
    merged = 0
    for seed, sizes, list_len in cases:
        d = tmp_path / f"case{seed}"
        d.mkdir()
        lo, lb, so, ss = _structure(seed, sizes, list_len)
        merged += _check(planner(str(d), lo, lb, so, ss), lo, lb, so, ss)
    assert merged > 50, "consecutive leaves on a list must be merged into one run"


def test_grid_neighbourhoods_become_a_few_runs(planner, tmp_path):
    """A 3D grid's 27-cell list is 9 runs of 3 consecutive cells; the builder names the leaf itself first, which splits one of them:
    at most 11 runs (list order is kept), fewer at the faces -- the list walk the kernel no longer does."""
    import sys
    sys.path.insert(0, ROOT)
    import nbody_amd as nbx
    b = nbx.uniform_bodies(20000, 3, 7)
    lo, lb, so, ss = nbx.leaves.uniform_grid_leaves(b, 3, 3)
    plan = planner(str(tmp_path), lo, lb, so, ss)
    _check(plan, lo, lb, so, ss)
    runs = np.diff(plan["op_off"].astype(np.int64))
    assert runs.max() <= 11 and runs.min() >= 4 and np.diff(so).max() == 27, (runs.min(), runs.max())
    assert plan["waves"] == 2


def _check_xcd_runs(keys, where):
    """Within a duration class of >= 16 workgroups, the workgroups at indices x mod 8 are the x-th of eight consecutive runs of the
    class in leaf order (`where`: a position that grows with the leaf), run lengths differing by at most one."""
    cls = keys * 1023 // max(int(keys.max()), 1)
    assert (np.diff(cls) <= 0).all()
    checked = 0
    edges = np.flatnonzero(np.diff(cls)) + 1
    for b0, b1 in zip(np.r_[0, edges], np.r_[edges, cls.size]):
        idx = np.arange(b0, b1)
        if b1 - b0 < 16:
            assert (np.diff(where[idx]) > 0).all()              # small classes stay in leaf order
            continue
        runs = [where[idx[idx % 8 == x]] for x in range(8)]
        for x in range(8):
            assert (np.diff(runs[x]) > 0).all()
            if x:
                assert runs[x - 1].max() < runs[x].min()
        assert max(len(r) for r in runs) - min(len(r) for r in runs) <= 1
        checked += 1
    return checked


def test_workgroups_of_a_class_are_dealt_to_the_xcds_in_runs(planner_xcd, tmp_path):
    """csrc/leaf_plan.h order_launch: workgroup i runs on XCD i % 8; every XCD takes one contiguous eighth (in leaf order) of each
    duration class.  Checked for the one-leaf workgroups (larger leaves) and the packed waves (tiny leaves), with every other
    invariant of the plan."""
    big = [40] * 700                                     # equal leaves: a list's length alone decides the duration -> a few large classes
    d = tmp_path / "big"
    d.mkdir()
    lo, lb, so, ss = _structure(11, big, lambda t: [3, 9, 27, 5][t % 4])
    plan = planner_xcd(str(d), lo, lb, so, ss)
    _check(plan, lo, lb, so, ss)
    blocks = plan["blocks"].astype(np.int64)
    assert blocks.shape[0] >= 700 and plan["pack_blocks"].shape[0] == 0
    assert _check_xcd_runs(blocks[:, 2], blocks[:, 4]) >= 3
    tiny = [4] * 6000
    d = tmp_path / "tiny"
    d.mkdir()
    lo, lb, so, ss = _structure(12, tiny, lambda t: [3, 9, 12][(t // 16) % 3])      # sixteen 4-body leaves to a wave: waves of three durations
    plan = planner_xcd(str(d), lo, lb, so, ss)
    _check(plan, lo, lb, so, ss)
    packs, subs = plan["pack_blocks"].astype(np.int64), plan["pack_subs"].astype(np.int64)
    assert packs.shape[0] >= 300
    # packed waves of one duration class can belong to different size classes (built class by class, each in leaf order): the runs are
    # consecutive in BUILD order, which is the order of their first sub-leaf
    assert _check_xcd_runs(packs[:, 4], packs[:, 0]) >= 2


def test_launch_order_is_lexicographic_in_duration_level_and_shape(planner, tmp_path):
    """the order key used to be trips * 8 + shape quantised into 1,024 classes, so from trips >= 128 on waves of different
    shapes shared a class (and were dealt to the XCDs side by side).  Now: (duration level, shape), lexicographic.  Long streams
    (one merged run of ~140 consecutive leaves) on leaves of two size classes."""
    rng = np.random.default_rng(5)
    sizes = np.where(np.arange(1200) % 3 == 0, 8, 4)                     # size classes 3 (7-8 bodies) and 1 (3-4 bodies)
    lo = np.concatenate([[0], np.cumsum(sizes)])
    lb = rng.permutation(int(lo[-1]))
    lists = []
    for t in range(sizes.size):
        start = int(rng.integers(0, sizes.size - 160))
        lists.append(np.arange(start, start + 120 + (t * 7) % 40))       # consecutive leaves: ONE run, 600-900 units, trips 150-230
    so = np.concatenate([[0], np.cumsum([len(l) for l in lists])])
    ss = np.concatenate(lists)
    plan = planner(str(tmp_path), lo, lb, so, ss)
    _check(plan, lo, lb, so, ss)
    packs = plan["pack_blocks"].astype(np.int64)                         # sub_lo, n_sub, w, P, trips, inv_w, longest, shape
    assert packs.shape[0] > 50 and packs[:, 4].min() >= 128 and set(packs[:, 7]) == {1, 3}
    longest = int(packs[:, 4].max())
    level = 1023 - packs[:, 4] * 1023 // longest
    key = level * 8 + packs[:, 7]
    assert (np.diff(key) >= 0).all(), "waves must be ordered by (duration level, shape)"
    for k in np.unique(key):                                             # a class never mixes shapes
        assert len(set(packs[key == k, 7])) == 1


def test_the_layout_does_not_depend_on_the_number_of_threads(planner, planner_threads, tmp_path):
    """csrc/leaf_plan.h lays the copy runs and the packed waves out on up to 8 threads (ranges of leaves of equal list length; windows
    of packed leaves); every array of the plan must come out as from one thread.  (Under ASan + UBSan, like the rest of this file.)"""
    rng = np.random.default_rng(9)
    for seed, sizes, list_len in ((21, rng.integers(0, 12, 3000).tolist(), lambda t: [3, 9, 27, 0, 14][t % 5]),
                                  (22, rng.integers(1, 90, 500).tolist(), lambda t: [5, 40, 1][t % 3]),
                                  (23, [4] * 70 + [0] * 5 + [9] * 30, lambda t: 9)):
        lo, lb, so, ss = _structure(seed, sizes, list_len)
        da, db = tmp_path / f"one{seed}", tmp_path / f"many{seed}"
        da.mkdir(); db.mkdir()
        one, many = planner(str(da), lo, lb, so, ss), planner_threads(str(db), lo, lb, so, ss)
        _check(many, lo, lb, so, ss)
        for k in one:
            assert np.array_equal(one[k], many[k]), k


def test_the_threaded_layout_is_race_free(tmp_path):
    """The same driver under ThreadSanitizer with the threads (and the XCD order) switched on from small sizes: a launch of one-leaf
    workgroups and a launch of packed waves, no report.  (Skipped where the sanitizer's runtime cannot start.)"""
    import sys
    sys.path.insert(0, ROOT)
    import nbody_amd as nbx
    exe = str(tmp_path / "leaf_plan_tsan")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-DNBX_PLAN_THREADS_FROM=64", "-DNBX_XCD_ORDER_FROM=16", "-pthread",
                            os.path.join(ROOT, "tests", "leaf_plan_check.cpp"), "-o", exe], capture_output=True, text=True)
    if build.returncode:
        pytest.skip("no ThreadSanitizer build here: " + build.stderr[-200:])
    cases = {"grid": (nbx.uniform_bodies(60000, 3, 3), lambda b: nbx.leaves.uniform_grid_leaves(b, 3, 4)),
             "bvh": (nbx.uniform_bodies(40000, 3, 4), lambda b: nbx.leaves.median_split_leaves(b, 3, 8, reach=0.5))}
    for name, (b, make) in cases.items():
        d = tmp_path / name
        d.mkdir()
        for n, a in zip(("leaf_offsets", "leaf_bodies", "list_offsets", "list_sources"), make(b)):
            np.asarray(a, dtype=np.uint32).tofile(str(d / (n + ".u32")))
        p = subprocess.run([exe, str(d)], capture_output=True, text=True)
        if p.returncode and "ThreadSanitizer" not in p.stderr and ("FATAL" in p.stderr or "unexpected memory mapping" in p.stderr):
            pytest.skip("ThreadSanitizer cannot run here: " + p.stderr[-200:])
        assert p.returncode == 0 and "ThreadSanitizer" not in p.stderr, p.stderr[-2000:]
        assert "blocks" in p.stdout
