"""What the harness's single timed call sees: a fresh process, nbx_warmup, then ONE nbx_brute_force_forces
(mode "phases": the same call taken apart through the context API, each phase for the first time in the process)."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 3 and sys.argv[1] == "child":
    import nbody_amd as nbx
    n = int(sys.argv[2])
    b = nbx.uniform_bodies(n, 3, 1)
    lib = nbx.load_library()
    t = time.perf_counter(); lib.nbx_warmup(0); w = time.perf_counter() - t
    if sys.argv[3] == "call":
        t = time.perf_counter(); nbx.brute_force_hip_n_body(b); first = time.perf_counter() - t
        t = time.perf_counter(); nbx.brute_force_hip_n_body(b); second = time.perf_counter() - t
        print(f"N={n}: warmup {w*1e3:.2f} ms, first call {first*1e3:.3f} ms, second call {second*1e3:.3f} ms", flush=True)
    else:
        for rep in ("first", "second"):
            ph = {}
            t = time.perf_counter(); c = nbx.Context(n, 3); ph["create"] = time.perf_counter() - t
            t = time.perf_counter(); c.upload(b); ph["upload"] = time.perf_counter() - t
            t = time.perf_counter(); c.compute_accel(nbx.SRC_ALL); c.synchronize(); ph["accel"] = time.perf_counter() - t
            t = time.perf_counter(); c.forces(); ph["forces"] = time.perf_counter() - t
            t = time.perf_counter(); c.close(); ph["destroy"] = time.perf_counter() - t
            print(f"N={n} {rep}: " + "  ".join(f"{k} {v*1e3:.3f}" for k, v in ph.items()), flush=True)
else:
    for n in [int(a) for a in sys.argv[1:]] or [1000, 100000, 1 << 20]:
        for mode in ("call", "phases"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n), mode], check=True)
