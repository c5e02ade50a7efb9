"""Measurement (GPU box): every body of BASELINE's large inputs against the strict fp64 kernel, with the data the mixed
mode's selection rule is calibrated on.  Uses the oracle as the checker, hence under tests/.
    python tests/measure/all_bodies_survey.py [--variant NAME] [--sigma S] [--out FILE.jsonl] [--dump] INPUT...
INPUT: uniform20 uniform20b uniform20s1 uniform16 uniform20_2d uniform22 uniform22_2d plummer22 blobs20 lattice20 lattice20eq
Appends JSON lines to gpurun_out/<FILE> (default accuracy_all_bodies.jsonl); --dump also writes gpurun_out/calib_<name>.npz."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import all_bodies  # noqa: E402
import nbody_amd as nbx  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


def make_input(o, w):
    gen = lambda seed, n, dim: o.round_inputs_to_f32(o.generate(seed, n, dim))
    if w == "uniform20":
        return gen(3, 1 << 20, 3), "uniform 3D N=2^20 (BASELINE config 3 input, seed 3)"
    if w == "uniform20b":
        return gen(4, 1 << 20, 3), "uniform 3D N=2^20 (seed 4)"
    if w == "uniform20s1":
        return o.round_inputs_to_f32(nbx.uniform_bodies(1 << 20, 3, 1)), "uniform 3D N=2^20 (bench.py's input, seed 1)"
    if w == "uniform16":
        return gen(2, 1 << 16, 3), "uniform 3D N=65,536 (BASELINE config 2 input, seed 2)"
    if w == "uniform20_2d":
        return gen(3, 1 << 20, 2), "uniform 2D N=2^20 (seed 3)"
    if w == "uniform22":
        return gen(6, 1 << 22, 3), "uniform 3D N=2^22 (seed 6)"
    if w == "uniform22_2d":
        return gen(6, 1 << 22, 2), "uniform 2D N=2^22 (seed 6)"
    if w == "plummer22":
        return (o.round_inputs_to_f32(nbx.plummer_bodies(1 << 22, 3, seed=5, a=1.0e5, total_mass=1.0e12)),
                "Plummer N=2^22 (BASELINE config 5 input, seed 5)")
    if w == "blobs20":
        # clustered: the `blobs` generator of tests/test_gpu_parity.py (test_structured_distributions) at N = 2^20 -- 64 Gaussian
        # clumps of ~16,384 bodies each (sigma 2e4) in the reference's box, bodies in generation order (clumps interleaved)
        n, rng = 1 << 20, np.random.default_rng(17)
        b = o.generate(40, n, 3)
        centres = rng.uniform(1e6, 9e6, size=(64, 3))
        b[:, :3] = centres[rng.integers(0, 64, n)] + rng.normal(scale=2.0e4, size=(n, 3))
        return o.round_inputs_to_f32(b), "clustered 3D N=2^20 (64 Gaussian clumps, sigma 2e4)"
    if w in ("lattice20", "lattice20eq"):
        # a 128 x 128 x 64 lattice (spacing 512, offset 1e6) IN INDEX ORDER: every source slice is a slab, the slabs' pulls on an
        # interior body cancel against each other.  lattice20: the generator's random masses; lattice20eq: equal masses -- every
        # interior body is then a near-total cancellation (kappa up to ~1e5), the selection rule's worst case
        n = 1 << 20
        b = o.generate(41, n, 3)
        g = np.stack(np.meshgrid(np.arange(128), np.arange(128), np.arange(64), indexing="ij"), -1).reshape(-1, 3).astype(float)
        b[:, :3] = 1.0e6 + 512.0 * g
        if w == "lattice20eq":
            b[:, -1] = 1.0e6
        return o.round_inputs_to_f32(b), ("lattice 128x128x64 in index order, " + ("equal masses" if w == "lattice20eq" else "random masses"))
    raise SystemExit("unknown input " + w)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("inputs", nargs="*", default=["uniform20", "plummer22"])
    ap.add_argument("--variant", default=None, help="fast kernel variant to survey (default: the library's)")
    ap.add_argument("--sigma", type=float, default=0.0, help="sigma factor of the mixed-mode run (0: the library's)")
    ap.add_argument("--out", default="accuracy_all_bodies.jsonl")
    ap.add_argument("--dump", action="store_true")
    args = ap.parse_args()
    o = Oracle()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for w in args.inputs:
        b, label = make_input(o, w)
        rec = all_bodies.survey(nbx, o, b, label, sigma_factor=args.sigma, variant=args.variant,
                                sigmas=(4.0, 6.0, 8.0, 12.0, 16.0, 20.0, 24.0, 32.0, 40.0, 48.0, 64.0, 96.0),
                                dump=os.path.join(ROOT, "gpurun_out", f"calib_{w}.npz") if args.dump else None)
        rec["input"] = w
        all_bodies.write_record(rec, args.out)
        keep = {k: rec[k] for k in ("what", "default_variant", "default_kernel_ms", "mixed_kernel_ms", "mixed_refine_ms", "default", "mixed") if k in rec}
        print(w, keep, flush=True)


if __name__ == "__main__":
    main()
