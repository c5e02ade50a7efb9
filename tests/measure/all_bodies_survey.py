"""Measurement (GPU box): every body of BASELINE's large inputs against the strict fp64 kernel, with the data the mixed
mode's selection rule is calibrated on.  Uses the oracle as the checker, hence under tests/.
    python tests/measure/all_bodies_survey.py [uniform20] [plummer22] [uniform20_2d] [uniform16]
Appends JSON lines to gpurun_out/accuracy_all_bodies.jsonl and writes gpurun_out/calib_<name>.npz."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import all_bodies  # noqa: E402
import nbody_amd as nbx  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


def main():
    o = Oracle()
    which = sys.argv[1:] or ["uniform20", "plummer22"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for w in which:
        if w == "uniform20":
            b, label = o.round_inputs_to_f32(o.generate(3, 1 << 20, 3)), "uniform 3D N=2^20 (BASELINE config 3 input, seed 3)"
        elif w == "uniform20b":
            b, label = o.round_inputs_to_f32(o.generate(4, 1 << 20, 3)), "uniform 3D N=2^20 (seed 4)"
        elif w == "uniform16":
            b, label = o.round_inputs_to_f32(o.generate(2, 1 << 16, 3)), "uniform 3D N=65,536 (BASELINE config 2 input, seed 2)"
        elif w == "uniform20_2d":
            b, label = o.round_inputs_to_f32(o.generate(3, 1 << 20, 2)), "uniform 2D N=2^20 (seed 3)"
        elif w == "uniform22":
            b, label = o.round_inputs_to_f32(o.generate(6, 1 << 22, 3)), "uniform 3D N=2^22 (seed 6)"
        elif w == "plummer22":
            b = o.round_inputs_to_f32(nbx.plummer_bodies(1 << 22, 3, seed=5, a=1.0e5, total_mass=1.0e12))
            label = "Plummer N=2^22 (BASELINE config 5 input, seed 5)"
        else:
            raise SystemExit("unknown input " + w)
        rec = all_bodies.survey(nbx, o, b, label, dump=os.path.join(ROOT, "gpurun_out", f"calib_{w}.npz"))
        all_bodies.write_record(rec)
        print(w, rec, flush=True)


if __name__ == "__main__":
    main()
