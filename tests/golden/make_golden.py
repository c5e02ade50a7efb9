"""Generate the golden vectors under tests/golden/ from the REFERENCE's own object code.

Run in the development container only (needs /root/reference to build oracle/_ref/libnbody_ref.so):
    python tests/golden/make_golden.py
Every array below is an OUTPUT of the reference's functions (or the seeded replica of its generator,
see oracle/ref_driver.cpp) -- data, not source.  Files are numpy .npz (no pickle).

  bf_D{2,3}_N{2,3,64,1024}.npz   bodies (seed 12345, reference ranges), forces of
                                 brute_force_seq_n_body / _omp_n_body_2 / _omp_n_body_1 (8 threads),
                                 and the same for the fp32-rounded inputs the device path consumes
  traj_D{2,3}_N64.npz            5 x { seq forces; update_body_velocities; update_body_positions }
  kat.npz                        known-answer cases: two-body, coincident bodies, r^2 guard either side
  octree_direct_D{2,3}_N512.npz  (python tests/golden/make_golden.py octree) fp32-representable bodies and the forces of
                                 the reference's Barnes-Hut octree walked with theta = 0 (octree.cpp:105-125 reached for
                                 every pair): the tree codes' attractive leaf law, SURVEY 8f-4
  bvh_leaves_D{2,3}_N4096.npz,   (python tests/golden/make_golden.py bvh) fp32-representable bodies, the LEAVES of the reference's
  bvh_leaves_D3_N3000.npz        own BVH<D>(bodies, 16) read through its public root (bvh.h:91-103, bvh.cpp:16-126) as CSR arrays,
                                 and per body the sum over all leaves of BVH<D>::calculate_force(body, leaf) (bvh.cpp:143-176):
                                 the tree codes' near-field sums on a reference-built tree (N = 3000: ragged leaves of 11-12)
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Reference  # noqa: E402

SEED = 12345


def round_f32(b, dim):
    r = b.copy()
    r[:, :dim] = r[:, :dim].astype(np.float32).astype(np.float64)
    r[:, -1] = r[:, -1].astype(np.float32).astype(np.float64)
    return r


def octree(ref):
    for dim in (2, 3):
        n = 512
        b = round_f32(ref.generate(SEED + 2, n, dim), dim)
        b[11, :dim] = (50.0, 60.0, 70.0)[:dim]    # small coordinates: fp32 spacing 3.8e-6, so the offset below survives
        b[10, :dim] = b[11, :dim]
        b[10, 0] += 2.0e-5                        # r^2 = 3.6e-10: below the leaf law's 1e-9 skip, above the brute-force law's 1e-10
        b = round_f32(b, dim)
        assert 1e-10 < ((b[10, :dim] - b[11, :dim]) ** 2).sum() < 1e-9
        np.savez_compressed(os.path.join(HERE, f"octree_direct_D{dim}_N{n}.npz"), bodies_f32=b, G=np.float64(ref.G()),
                            forces_octree_theta0=ref.octree_direct_forces(b), forces_brute_seq=ref.brute_force(0, b))


def bvh(ref):
    for dim, n in ((2, 4096), (3, 4096), (3, 3000)):
        b = round_f32(ref.generate(SEED + 3 + dim, n, dim), dim)
        b[21, :dim] = (50.0, 60.0, 70.0)[:dim]
        b[20, :dim] = b[21, :dim]
        b[20, 0] += 2.0e-5                        # a pair at r^2 = 3.6e-10: skipped by the leaf law (< 1e-9, bvh.cpp:164)
        b[30, :dim] = b[31, :dim]                 # and an exact duplicate (the "same position" skip, bvh.cpp:151-158)
        b = round_f32(b, dim)
        offs, idx = ref.bvh_leaves(b, 16)
        assert offs[-1] == n and np.array_equal(np.sort(idx), np.arange(n))
        np.savez_compressed(os.path.join(HERE, f"bvh_leaves_D{dim}_N{n}.npz"), bodies_f32=b, G=np.float64(ref.G()),
                            max_bodies_per_leaf=np.int64(16), leaf_offsets=offs, leaf_bodies=idx,
                            forces_bvh_all_leaves=ref.bvh_leaf_forces(b, 16))


def main():
    subprocess.check_call([os.path.join(ROOT, "oracle", "build_ref.sh")])
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    ref = Reference()
    if len(sys.argv) > 1 and sys.argv[1] == "octree":
        octree(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "bvh":
        bvh(ref)
        return
    assert ref.sizeof_body(3) == 56 and ref.sizeof_body(2) == 40
    for dim in (2, 3):
        for n in (2, 3, 64, 1024):
            b = ref.generate(SEED, n, dim)
            br = round_f32(b, dim)
            np.savez_compressed(
                os.path.join(HERE, f"bf_D{dim}_N{n}.npz"),
                seed=np.int64(SEED), bodies=b, G=np.float64(ref.G()),
                forces_seq=ref.brute_force(0, b), forces_omp_1=ref.brute_force(1, b), forces_omp_2=ref.brute_force(2, b),
                omp_1_threads=np.int64(int(os.environ["OMP_NUM_THREADS"])),
                bodies_f32=br, forces_seq_f32=ref.brute_force(0, br), forces_omp_2_f32=ref.brute_force(2, br))
        # trajectory: dt chosen so that positions move by O(1e-3) of the box per step
        n, dt, steps = 64, 1.0e3, 5
        b = ref.generate(SEED + 1, n, dim)
        states, forces = [b.copy()], []
        cur = b.copy()
        for _ in range(steps):
            f = ref.brute_force(0, cur)
            ref.update_body_velocities(cur, f, dt)
            ref.update_body_positions(cur, dt)
            forces.append(f)
            states.append(cur.copy())
        np.savez_compressed(os.path.join(HERE, f"traj_D{dim}_N{n}.npz"), dt=np.float64(dt), steps=np.int64(steps),
                            states=np.stack(states), forces=np.stack(forces))

    # known-answer cases
    def body(p, m=1.0, v=(0, 0, 0)):
        return list(p) + list(v) + [m]
    two = np.array([body((0, 0, 0)), body((2, 0, 0))], dtype=np.float64)
    coincident = np.array([body((5, 5, 5), 3.0), body((5, 5, 5), 7.0), body((6, 5, 5), 2.0)], dtype=np.float64)
    near_skip = np.array([body((1, 1, 1)), body((1 + 9.9e-6, 1, 1))], dtype=np.float64)      # r^2 = 9.8e-11 < 1e-10
    near_keep = np.array([body((1, 1, 1)), body((1 + 1.1e-5, 1, 1))], dtype=np.float64)      # r^2 = 1.21e-10
    out = {}
    for name, b in (("two", two), ("coincident", coincident), ("near_skip", near_skip), ("near_keep", near_keep)):
        out[name + "_bodies"] = b
        out[name + "_forces_seq"] = ref.brute_force(0, b)
        out[name + "_forces_omp_2"] = ref.brute_force(2, b)
    s = two.copy()
    f = ref.brute_force(0, s)
    ref.update_body_velocities(s, f, 1.0)
    ref.update_body_positions(s, 1.0)
    out["two_after_step_dt1"] = s
    np.savez_compressed(os.path.join(HERE, "kat.npz"), **out)
    octree(ref)
    bvh(ref)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
