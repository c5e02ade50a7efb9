"""GPU: the leaf plan laid out on the device (csrc/leaf_plan_device.h) is, array for array and word for word, the plan the host
planner makes (csrc/leaf_plan.h; the sanitizer builds' subject in tests/test_leaf_plan_cpu.py).  tests/device_plan_check.hip is
compiled here with hipcc (the GPU box has the same toolchain) and run on ragged structures, on structures large enough for the
XCD-aware launch order, on the reference trees' shapes, and on refused structures.  The forces' bit-identity through both planners is
tests/test_gpu_leaf_pairs.py's business (every case there runs through both)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("device_plan") / "device_plan_check")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "device_plan_check.hip"), "-o", exe], check=True)

    def run(workdir, leaves, n_bodies=None, reps=0, refuse=False):
        os.makedirs(workdir, exist_ok=True)
        for name, a in zip(("leaf_offsets", "leaf_bodies", "list_offsets", "list_sources"), leaves):
            np.asarray(a, dtype=np.uint32).tofile(os.path.join(workdir, name + ".u32"))
        if n_bodies is not None:
            open(os.path.join(workdir, "n_bodies.txt"), "w").write(str(int(n_bodies)))
        return subprocess.run([exe, workdir] + (["refuse"] if refuse else [str(reps)] if reps else []), capture_output=True, text=True, timeout=600)
    return run


def _structure(seed, sizes, list_len):
    """tests/test_leaf_plan_cpu.py's generator: random lists with runs of consecutive leaves that must be merged."""
    rng = np.random.default_rng(seed)
    sizes = np.asarray(sizes)
    lo = np.concatenate([[0], np.cumsum(sizes)])
    n = int(lo[-1]) + 11
    lb = rng.permutation(n)[:lo[-1]]
    lists = []
    for t in range(sizes.size):
        k = list_len(t)
        l = rng.integers(0, sizes.size, k)
        if t % 2 == 0 and k >= 3:
            start = int(rng.integers(0, max(1, sizes.size - 3)))
            l[:3] = [start, start + 1, start + 2]
        lists.append(l)
    so = np.concatenate([[0], np.cumsum([len(l) for l in lists])])
    ss = np.concatenate(lists) if lists else np.zeros(0, dtype=np.int64)
    return (lo, lb, so, ss), n


def test_ragged_structures_word_for_word(check, tmp_path):
    rng = np.random.default_rng(9)
    cases = {
        "tiny_mixed": (rng.integers(0, 12, 3000).tolist(), lambda t: [3, 9, 27, 0, 14][t % 5]),          # every packed class, empty leaves, empty lists
        "fmm_sized": (rng.integers(1, 90, 500).tolist(), lambda t: [5, 40, 1][t % 3]),                   # two-wave workgroups, cut pieces
        "holes": ([4] * 70 + [0] * 5 + [9] * 30, lambda t: 9),
        "long_lists": (rng.integers(1, 9, 400).tolist(), lambda t: [70, 130, 40][t % 3]),                # lists beyond one wave's 64 entries; > 32 runs: not packed
        "big_leaves": (rng.integers(100, 300, 60).tolist(), lambda t: 7),                                # several workgroups per leaf
        "one_leaf": ([5], lambda t: 1),
        "xcd_blocks": ([40] * 5000, lambda t: [3, 9, 27, 5][t % 4]),                                     # >= 4,096 one-leaf workgroups: a class's eighths to the XCDs
        "xcd_packs": ([4] * 70000, lambda t: [3, 9, 12][(t // 16) % 3]),                                 # >= 4,096 packed waves of three durations
        "xcd_mixed": (np.r_[rng.integers(1, 17, 60000), rng.integers(17, 70, 6000)].tolist(), lambda t: [27, 9, 40, 3][t % 4]),
    }
    for k, (name, (sizes, list_len)) in enumerate(cases.items()):
        leaves, n = _structure(100 + k, sizes, list_len)
        p = check(str(tmp_path / name), leaves, n)
        assert p.returncode == 0 and p.stdout.startswith("identical"), f"{name}: {p.stdout[-600:]} {p.stderr[-300:]}"


def test_random_structures_word_for_word(check, tmp_path):
    """Thirty random structures (sizes, list lengths, share of empty leaves and of merged runs drawn per case): the device plan equals
    the host plan word for word on every one."""
    rng = np.random.default_rng(2025)
    for k in range(30):
        n_leaves = int(rng.integers(1, 4000))
        top = int(rng.choice([3, 9, 17, 40, 130, 300]))
        sizes = rng.integers(0 if rng.random() < 0.5 else 1, top + 1, n_leaves)
        if sizes.sum() == 0:
            sizes[0] = 1
        longest = int(rng.choice([1, 5, 30, 90]))
        lens = rng.integers(0, longest + 1, n_leaves)
        leaves, n = _structure(1000 + k, sizes.tolist(), lambda t: int(lens[t]))
        p = check(str(tmp_path / f"case{k}"), leaves, n)
        assert p.returncode == 0 and p.stdout.startswith("identical"), f"case {k} ({n_leaves} leaves of up to {top}, lists up to {longest}): {p.stdout[-600:]} {p.stderr[-300:]}"


def test_the_reference_trees_shapes_word_for_word_and_timed(check, tmp_path, nbx):
    """N = 2^20: the BVH's 16-body leaves, 8-body leaves and 4-body grid cells -- the rows of VERDICT r4 item 2 -- identical, and the
    device layout's stream time (copies of the four arrays included) printed for the record."""
    b = nbx.uniform_bodies(1 << 20, 3, 5)
    for name, leaves in (("bvh16", nbx.leaves.median_split_leaves(b, 3, 16, reach=0.5)), ("bvh8", nbx.leaves.median_split_leaves(b, 3, 8, reach=0.5)),
                         ("grid4", nbx.leaves.uniform_grid_leaves(b, 3, 6)), ("grid32", nbx.leaves.uniform_grid_leaves(b, 3, 5))):
        p = check(str(tmp_path / name), leaves, 1 << 20, reps=5)
        assert p.returncode == 0 and p.stdout.startswith("identical"), f"{name}: {p.stdout[-600:]} {p.stderr[-300:]}"
        print(name, p.stdout.strip().replace("\n", " | "))


def test_refused_structures_are_refused_on_the_device(check, tmp_path):
    """The host planner follows every index unchecked (validate_csr runs before it); the device planner checks as it goes and must
    never dereference anything out of range."""
    good = (np.array([0, 5, 10]), np.arange(10), np.array([0, 1, 2]), np.array([0, 1]))
    p = check(str(tmp_path / "good"), good, 10)
    assert p.returncode == 0 and p.stdout.startswith("identical"), p.stdout
    for name, leaves, text in (("bad_source", (good[0], good[1], good[2], np.array([0, 2])), "list_sources entry out of range"),
                               ("bad_body", (good[0], np.r_[np.arange(9), 10], good[2], good[3]), "leaf_bodies entry out of range"),
                               ("twice", (good[0], np.r_[np.arange(9), 0], good[2], good[3]), "a body may belong to at most one leaf")):
        p = check(str(tmp_path / name), leaves, 10, refuse=True)
        assert p.returncode == 0 and text in p.stdout, f"{name}: {p.stdout} {p.stderr[-300:]}"
