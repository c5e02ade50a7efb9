import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nbody_amd as nbx
n = 1 << 20
b = nbx.uniform_bodies(n, 3, 1)
nbx.load_library().nbx_warmup(0)
for R in (2, 8):
    for rep in range(3):
        t0 = time.perf_counter(); node = nbx.Node(n, 3, [0] * R, nbx.EXCHANGE_PEER_COPY); t1 = time.perf_counter()
        node.upload(b); t2 = time.perf_counter()
        f = node.forces(); t3 = time.perf_counter()
        node.close(); t4 = time.perf_counter()
        print(f"R={R} rep {rep}: create {1e3*(t1-t0):.2f}  upload {1e3*(t2-t1):.2f}  forces {1e3*(t3-t2):.2f}  destroy {1e3*(t4-t3):.2f} ms", flush=True)
