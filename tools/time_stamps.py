"""Interleaved A/B: the default force kernel with the in-kernel clock stamps off / on (nbx_ctx_enable_clock_stamps), N = 2^20.
   python tools/time_stamps.py [N] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nbx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
b = nbx.uniform_bodies(n, 3, 1)
with nbx.Context(n, 3) as c:
    c.upload(b)
    res = {False: [], True: []}
    clocks = []
    for r in range(rounds + 1):
        for on in (False, True):
            c.enable_clock_stamps(on)
            c.compute_accel()
            ms, _ = c.kernel_time()
            if r:
                res[on].append(ms)
                if on:
                    clocks.append(c.shader_clock())
    for on in (False, True):
        print(f"stamps {'on ' if on else 'off'}: best {min(res[on]):8.3f} ms  median {sorted(res[on])[len(res[on]) // 2]:8.3f} ms  all {[round(x, 2) for x in res[on]]}")
    for k in clocks:
        print("held clock:", {a: round(v, 1) for a, v in k.items()})
    tf, mhz = nbx.capi.measure_valu_ceiling(0, 50.0)
    print(f"pure v_pk_fma_f32 stream right after: {tf:.1f} TFLOP/s at {mhz:.0f} MHz")
