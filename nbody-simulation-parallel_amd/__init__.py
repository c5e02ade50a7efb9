"""nbody-simulation-parallel_amd: MI355X-native all-pairs force + kick/drift hot path.

The directory name carries a hyphen (it mirrors the reference repository's name), so import it
through the root-level shim:  `import nbody_amd as nbx`  (or importlib with the name
`nbody_simulation_parallel_amd`).  Contents:
  csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/nbody_hip.h)
  host/      C++ mirror of the reference's methods.h entry points + benchmark harness
  capi.py    ctypes binding of the C ABI (same entry-point names for Python callers / tests)
  sharding.py, dist.py   one-process-per-GPU sharding over torch.distributed (RCCL)
  leaves.py  CSR leaf lists for the leaf-pair direct-sum entry point (tree codes' near field)
"""
from .capi import (  # noqa: F401
    ABI, EXCHANGE_AUTO, EXCHANGE_PEER_COPY, EXCHANGE_RCCL, LIB_PATH, REFERENCE_G, SRC_ALL, SRC_LOCAL, SRC_REMOTE,
    FORCE_LAW_NEWTON, FORCE_LAW_REFERENCE, LAW_BRUTE, LAW_FMM_P2P, LAW_TREE_LEAF,
    Context, EvalInfo, LeafPlan, NbxError, Node,
    get_default_refine, refine_sigma_default, set_default_refine,
    body_stride, brute_force_hip_n_body, device_count, leaf_pair_forces_hip, leapfrog_hip_n_body, load_library, variants,
)
from . import generate, leaves, sharding  # noqa: E402,F401
from .generate import plummer_bodies, uniform_bodies  # noqa: E402,F401


def __getattr__(name):
    # dist.py imports torch; load it only when asked for
    if name == "dist":
        import importlib
        return importlib.import_module(__name__ + ".dist")
    raise AttributeError(name)
