"""ctypes binding of libnbody_hip.so (C ABI: include/nbody_hip.h).

Host-side mirror of the reference's solver entry points for the brute-force path
(nbody-sim-new/methods.h:29-37, :85-91) for Python callers and for the tests: same names,
same argument meaning (an array of Body<D> in, an array of Vector<D> forces out), same error
behaviour (a failure raises, like the C++ wrappers throw into the harness's safely_execute,
utils.h:87-104).  Nothing here computes: every call goes through the C ABI into the HIP library,
and a missing library or missing GPU raises NbxError -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np

NBX_OK = 0
NBX_ERR_INVALID, NBX_ERR_NO_DEVICE, NBX_ERR_HIP, NBX_ERR_ALLOC, NBX_ERR_STATE = 1, 2, 3, 4, 5
SRC_ALL, SRC_LOCAL, SRC_REMOTE = 0, 1, 2
REFERENCE_G = 4.471e-21  # nbody-sim-new/utils.h:21

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NBODY_HIP_LIBRARY") or os.path.join(_PKG_DIR, "libnbody_hip.so")   # the override is for A/B builds (tools/)

# every symbol include/nbody_hip.h declares: (name, restype, argtypes)
_c = ctypes
_vp, _sz, _i, _d = _c.c_void_p, _c.c_size_t, _c.c_int, _c.c_double
_pf, _pd, _pi = _c.POINTER(_c.c_float), _c.POINTER(_c.c_double), _c.POINTER(_c.c_int)
ABI = [
    ("nbx_abi_version", _i, []),
    ("nbx_strerror", _c.c_char_p, [_i]),
    ("nbx_last_error_detail", _c.c_char_p, []),
    ("nbx_device_count", _i, [_pi]),
    ("nbx_warmup", _i, [_i]),
    ("nbx_release_cached", _i, []),
    ("nbx_set_default_refine", _i, [_d, _d]),
    ("nbx_get_default_refine", _i, [_pd, _pd]),
    ("nbx_refine_sigma_default", _d, [_i]),
    ("nbx_brute_force_forces", _i, [_vp, _sz, _i, _sz, _d, _i, _vp, _pf]),
    ("nbx_brute_force_forces_ex", _i, [_vp, _sz, _i, _sz, _d, _i, _d, _vp, _vp]),
    ("nbx_leapfrog", _i, [_vp, _sz, _i, _sz, _d, _d, _i, _i, _pf]),
    ("nbx_leaf_pair_forces", _i, [_vp, _sz, _i, _sz, _vp, _vp, _sz, _vp, _vp, _i, _d, _i, _vp, _pf]),
    ("nbx_leaf_plan_create", _i, [_c.POINTER(_vp), _i, _i, _sz, _vp, _vp, _sz, _vp, _vp]),
    ("nbx_leaf_plan_destroy", _i, [_vp]),
    ("nbx_leaf_plan_forces", _i, [_vp, _vp, _sz, _i, _d, _vp, _pf]),
    ("nbx_leaf_plan_forces_ctx", _i, [_vp, _vp, _i, _d, _vp, _pf]),
    ("nbx_leaf_plan_get_forces", _i, [_vp, _vp]),
    ("nbx_leaf_plan_kick_drift", _i, [_vp, _vp, _d]),
    ("nbx_leaf_plan_step", _i, [_vp, _vp, _i, _d, _d, _i]),
    ("nbx_leaf_plan_time_kernel", _i, [_vp, _i, _i, _pf]),
    ("nbx_leaf_plan_info", _i, [_vp, _c.POINTER(_sz), _c.POINTER(_sz), _c.POINTER(_sz), _pi]),
    ("nbx_ctx_create", _i, [_c.POINTER(_vp), _i, _i, _sz, _i, _i]),
    ("nbx_ctx_destroy", _i, [_vp]),
    ("nbx_ctx_set_stream", _i, [_vp, _vp]),
    ("nbx_ctx_set_gather_buffers", _i, [_vp, _vp, _vp]),
    ("nbx_ctx_gather_layout", _i, [_vp, _c.POINTER(_sz), _c.POINTER(_sz), _c.POINTER(_vp), _c.POINTER(_vp)]),
    ("nbx_ctx_upload_bodies", _i, [_vp, _vp, _sz]),
    ("nbx_ctx_upload_shard", _i, [_vp, _vp, _sz, _pd, _pd]),
    ("nbx_ctx_upload_finish", _i, [_vp, _d, _d]),
    ("nbx_ctx_compute_accel", _i, [_vp, _i]),
    ("nbx_ctx_kick_drift", _i, [_vp, _d, _d]),
    ("nbx_ctx_kick_drift2", _i, [_vp, _d, _d, _d]),
    ("nbx_ctx_step_kdk", _i, [_vp, _d, _d, _i]),
    ("nbx_ctx_step", _i, [_vp, _d, _d, _i]),
    ("nbx_ctx_get_forces", _i, [_vp, _d, _vp]),
    ("nbx_ctx_accuracy", _i, [_vp, _d, _vp, _pd]),
    ("nbx_ctx_get_accel", _i, [_vp, _vp]),
    ("nbx_ctx_download_bodies", _i, [_vp, _vp, _sz]),
    ("nbx_ctx_energy", _i, [_vp, _d, _pd, _pd]),
    ("nbx_ctx_synchronize", _i, [_vp]),
    ("nbx_ctx_set_tuning", _i, [_vp, _i, _i]),
    ("nbx_ctx_set_softening", _i, [_vp, _d]),
    ("nbx_ctx_set_law", _i, [_vp, _i]),
    ("nbx_ctx_set_refine", _i, [_vp, _d, _d]),
    ("nbx_ctx_refine_stats", _i, [_vp, _c.POINTER(_c.c_uint), _c.POINTER(_c.c_uint)]),
    ("nbx_ctx_get_aux", _i, [_vp, _vp]),
    ("nbx_ctx_close_set_mode", _i, [_vp, _pi, _c.POINTER(_c.c_uint), _c.POINTER(_c.c_uint)]),
    ("nbx_ctx_effective_tuning", _i, [_vp, _pi, _pi]),
    ("nbx_num_variants", _i, []),
    ("nbx_variant_name", _c.c_char_p, [_i]),
    ("nbx_default_variant", _i, []),
    ("nbx_ctx_kernel_time", _i, [_vp, _pf, _pi]),
    ("nbx_ctx_refine_time", _i, [_vp, _pf]),
    ("nbx_variant_kernel_symbol", _i, [_i, _i, _i, _c.c_char_p, _sz]),
    ("nbx_ctx_enable_clock_stamps", _i, [_vp, _i]),
    ("nbx_ctx_shader_clock", _i, [_vp, _pd, _pd, _pd, _pi]),
    ("nbx_measure_valu_ceiling", _i, [_i, _d, _pd, _pd]),
    ("nbx_node_create", _i, [_c.POINTER(_vp), _i, _pi, _i, _sz, _i]),
    ("nbx_node_destroy", _i, [_vp]),
    ("nbx_node_exchange_mode", _i, [_vp, _pi]),
    ("nbx_node_upload_bodies", _i, [_vp, _vp, _sz]),
    ("nbx_node_verify_exchange", _i, [_vp, _c.POINTER(_sz)]),
    ("nbx_node_set_tuning", _i, [_vp, _i, _i]),
    ("nbx_node_set_softening", _i, [_vp, _d]),
    ("nbx_node_set_law", _i, [_vp, _i]),
    ("nbx_node_set_refine", _i, [_vp, _d, _d]),
    ("nbx_node_refine_stats", _i, [_vp, _c.POINTER(_c.c_uint), _c.POINTER(_c.c_uint)]),
    ("nbx_node_compute_forces", _i, [_vp, _d, _vp]),
    ("nbx_node_step", _i, [_vp, _d, _d, _i]),
    ("nbx_node_step_kdk", _i, [_vp, _d, _d, _i]),
    ("nbx_node_synchronize", _i, [_vp]),
    ("nbx_node_download_bodies", _i, [_vp, _vp, _sz]),
    ("nbx_node_energy", _i, [_vp, _d, _pd, _pd]),
    ("nbx_node_kernel_time", _i, [_vp, _pf, _pi]),
    ("nbx_node_enable_timing", _i, [_vp, _i]),
    ("nbx_node_pass_times", _i, [_vp, _i, _pi, _c.POINTER(_sz), _pf, _pf, _pf, _pi]),
]
EXCHANGE_AUTO, EXCHANGE_PEER_COPY, EXCHANGE_RCCL = 0, 1, 2


class NbxError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: nbx status {status}" + (f" -- {detail}" if detail else ""))


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen libnbody_hip.so and type every entry point.  Raises if the library is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise NbxError(-1, "load_library", f"{p} not found: build it with `make lib` (hipcc, gfx950); "
                       "there is no CPU fallback")
    lib = ctypes.CDLL(p)
    for name, res, args in ABI:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
        import atexit
        atexit.register(lib.nbx_release_cached)   # parked streams / RCCL communicators go back before the runtime does
    return lib


def _check(lib, rc: int, where: str):
    if rc != NBX_OK:
        msg = lib.nbx_strerror(rc).decode()
        det = lib.nbx_last_error_detail().decode()
        raise NbxError(rc, where, f"{msg}: {det}" if det else msg)


def body_stride(dim: int) -> int:
    """doubles per Body<dim> (nbody-sim-new/body.h:8-11): position[D], velocity[D], mass."""
    return 2 * dim + 1


def _as_bodies(bodies: np.ndarray, writable: bool = False) -> Tuple[np.ndarray, int]:
    b = np.asarray(bodies)
    if b.dtype != np.float64 or b.ndim != 2 or b.shape[1] not in (5, 7):
        raise ValueError("bodies must be float64 [n, 2*D+1] rows of Body<D> = (position[D], velocity[D], mass)")
    if not b.flags["C_CONTIGUOUS"] or (writable and not b.flags["WRITEABLE"]):
        if writable:
            raise ValueError("bodies must be a writable C-contiguous array")
        b = np.ascontiguousarray(b)
    return b, (b.shape[1] - 1) // 2


def device_count() -> int:
    lib = load_library()
    n = ctypes.c_int(0)
    _check(lib, lib.nbx_device_count(ctypes.byref(n)), "nbx_device_count")
    return n.value


def measure_valu_ceiling(device: int = 0, target_ms: float = 50.0) -> Tuple[float, float]:
    """(TFLOP/s, shader MHz) of a pure v_pk_fma_f32 stream on `device` for ~target_ms: this box's fp32 VALU ceiling."""
    lib = load_library()
    tf, mhz = ctypes.c_double(0.0), ctypes.c_double(0.0)
    _check(lib, lib.nbx_measure_valu_ceiling(int(device), float(target_ms), ctypes.byref(tf), ctypes.byref(mhz)), "nbx_measure_valu_ceiling")
    return tf.value, mhz.value


def variant_kernel_symbol(variant, dim: int = 3, mixed_mode: bool = True) -> str:
    """Demangled symbol (as rocprofv3 prints it) of the force kernel `variant` (id or name) launches; needs no device."""
    lib = load_library()
    if isinstance(variant, str):
        variant = variants().index(variant)
    buf = ctypes.create_string_buffer(512)
    _check(lib, lib.nbx_variant_kernel_symbol(int(variant), int(dim), 1 if mixed_mode else 0, buf, 512), "nbx_variant_kernel_symbol")
    return buf.value.decode()


def variants():
    lib = load_library()
    return [lib.nbx_variant_name(i).decode() for i in range(lib.nbx_num_variants())]


class EvalInfo(ctypes.Structure):
    """nbx_eval_info of include/nbody_hip.h: what the one-shot evaluation ran."""
    _fields_ = [("kernel_ms", _c.c_float), ("refine_ms", _c.c_float), ("refine_tolerance", _d), ("refine_selected", _c.c_uint), ("refine_refined", _c.c_uint),
                ("variant", _i), ("close_set_mode", _i)]


def set_default_refine(rel_tolerance: float, sigma_factor: float = 0.0):
    """Process-wide precision default of contexts / nodes / one-shot calls made afterwards (nbx_set_default_refine):
    0 = plain fp32, otherwise the mixed mode's relative tolerance.  The library starts with 1e-5."""
    lib = load_library()
    _check(lib, lib.nbx_set_default_refine(float(rel_tolerance), float(sigma_factor)), "nbx_set_default_refine")


def get_default_refine() -> Tuple[float, float]:
    lib = load_library()
    a, b = ctypes.c_double(0.0), ctypes.c_double(0.0)
    _check(lib, lib.nbx_get_default_refine(ctypes.byref(a), ctypes.byref(b)), "nbx_get_default_refine")
    return a.value, b.value


def refine_sigma_default(dim: int) -> float:
    return float(load_library().nbx_refine_sigma_default(int(dim)))


def brute_force_hip_n_body(bodies: np.ndarray, G: float = REFERENCE_G, device: int = 0,
                           return_kernel_ms: bool = False, rel_tolerance: Optional[float] = None, return_info: bool = False):
    """forces = brute_force_hip_n_body(bodies): drop-in for brute_force_seq_n_body<D> /
    brute_force_omp_n_body_{1,2}<D> (nbody-sim-new/methods.h:29-37).  bodies: float64 [n, 2D+1];
    returns float64 [n, D] forces (Vector<D> per body).  rel_tolerance: None = the process default (mixed mode, 1e-5),
    0 = plain fp32, > 0 = mixed mode with this tolerance (nbx_brute_force_forces_ex); return_info adds the EvalInfo."""
    lib = load_library()
    b, dim = _as_bodies(bodies)
    n = b.shape[0]
    out = np.empty((n, dim), dtype=np.float64)
    if rel_tolerance is None and not return_info:
        ms = ctypes.c_float(0.0)
        rc = lib.nbx_brute_force_forces(b.ctypes.data, n, dim, b.shape[1] * 8, G, device, out.ctypes.data,
                                        ctypes.byref(ms))
        _check(lib, rc, "nbx_brute_force_forces")
        return (out, ms.value) if return_kernel_ms else out
    info = EvalInfo()
    rc = lib.nbx_brute_force_forces_ex(b.ctypes.data, n, dim, b.shape[1] * 8, G, device,
                                       -1.0 if rel_tolerance is None else float(rel_tolerance), out.ctypes.data, ctypes.byref(info))
    _check(lib, rc, "nbx_brute_force_forces_ex")
    if return_info:
        return out, info
    return (out, info.kernel_ms) if return_kernel_ms else out


def leapfrog_hip_n_body(bodies: np.ndarray, dt: float, nsteps: int, G: float = REFERENCE_G, device: int = 0):
    """In-place nsteps x { forces; update_body_velocities; update_body_positions }
    (nbody-sim-new/methods.cpp:425-450), state resident on the device between steps."""
    lib = load_library()
    b, dim = _as_bodies(bodies, writable=True)
    ms = ctypes.c_float(0.0)
    rc = lib.nbx_leapfrog(b.ctypes.data, b.shape[0], dim, b.shape[1] * 8, G, dt, nsteps, device, ctypes.byref(ms))
    _check(lib, rc, "nbx_leapfrog")
    return ms.value


LAW_BRUTE, LAW_TREE_LEAF, LAW_FMM_P2P = 0, 1, 2
FORCE_LAW_REFERENCE, FORCE_LAW_NEWTON = 0, 1


def leaf_pair_forces_hip(bodies: np.ndarray, leaf_offsets, leaf_bodies, list_offsets, list_sources, law: int = LAW_FMM_P2P,
                         G: float = REFERENCE_G, device: int = 0, return_kernel_ms: bool = False):
    """Direct (near-field) sums of the reference's tree codes over CSR leaf lists -- FMM_Parlay<D>::p2p_phase
    (nbody-sim-new/fmm_parlay.cpp:916-1022), the BVH leaf loop (bvh.cpp:150-176), the octree leaf term
    (octree.cpp:105-125) -- on the device; see nbx_leaf_pair_forces in include/nbody_hip.h."""
    lib = load_library()
    b, dim = _as_bodies(bodies)
    arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a in (leaf_offsets, leaf_bodies, list_offsets, list_sources)]
    if arrs[0].size < 1 or arrs[2].size != arrs[0].size:
        raise ValueError("leaf_offsets and list_offsets must both have n_leaves + 1 entries")
    out = np.empty((b.shape[0], dim), dtype=np.float64)
    ms = ctypes.c_float(0.0)
    _check(lib, lib.nbx_leaf_pair_forces(b.ctypes.data, b.shape[0], dim, b.shape[1] * 8, arrs[0].ctypes.data, arrs[1].ctypes.data,
                                         arrs[0].size - 1, arrs[2].ctypes.data, arrs[3].ctypes.data, law, G, device,
                                         out.ctypes.data, ctypes.byref(ms)), "nbx_leaf_pair_forces")
    return (out, ms.value) if return_kernel_ms else out


class LeafPlan:
    """Device-resident leaf structure of a tree code (nbx_leaf_plan_* of include/nbody_hip.h): the CSR arrays are validated,
    laid out and uploaded once; every evaluation re-gathers positions (from host bodies, or from a resident Context) and runs the
    pair kernel -- the call pattern of bvh.cpp:143-176 / fmm_parlay.cpp:916-1022, whose trees stand while the bodies move."""

    def __init__(self, n_bodies: int, dim: int, leaf_offsets, leaf_bodies, list_offsets, list_sources, device: int = 0):
        self.lib = load_library()
        self.n, self.dim, self.device = int(n_bodies), int(dim), device
        arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a in (leaf_offsets, leaf_bodies, list_offsets, list_sources)]
        if arrs[0].size < 1 or arrs[2].size != arrs[0].size:
            raise ValueError("leaf_offsets and list_offsets must both have n_leaves + 1 entries")
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.nbx_leaf_plan_create(ctypes.byref(h), device, dim, self.n, arrs[0].ctypes.data, arrs[1].ctypes.data,
                                                       arrs[0].size - 1, arrs[2].ctypes.data, arrs[3].ctypes.data), "nbx_leaf_plan_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.nbx_leaf_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc, where):
        _check(self.lib, rc, where)

    def info(self):
        """(padded slots, copy runs, workgroups, wave64 per workgroup) of the resident layout."""
        a, b, c, w = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_int()
        self._ck(self.lib.nbx_leaf_plan_info(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(w)), "nbx_leaf_plan_info")
        return a.value, b.value, c.value, w.value

    def forces(self, bodies: np.ndarray, law: int = 2, G: float = REFERENCE_G, return_kernel_ms: bool = False):
        """Host bodies in, host forces out (nbx_leaf_plan_forces)."""
        b, dim = _as_bodies(bodies)
        if dim != self.dim or b.shape[0] != self.n:
            raise ValueError("bodies shape does not match the plan")
        out = np.empty((self.n, dim), dtype=np.float64)
        ms = ctypes.c_float(0.0)
        self._ck(self.lib.nbx_leaf_plan_forces(self.h, b.ctypes.data, b.shape[1] * 8, law, G, out.ctypes.data, ctypes.byref(ms)), "nbx_leaf_plan_forces")
        return (out, ms.value) if return_kernel_ms else out

    def forces_ctx(self, ctx: "Context", law: int = 2, G: float = REFERENCE_G, fetch: bool = True, timed: bool = False):
        """Bodies resident in `ctx` (nbx_leaf_plan_forces_ctx).  fetch=False: the sums stay on the device (asynchronous unless
        timed); returns forces, (forces, kernel_ms), kernel_ms or None accordingly."""
        out = np.empty((self.n, self.dim), dtype=np.float64) if fetch else None
        ms = ctypes.c_float(0.0)
        self._ck(self.lib.nbx_leaf_plan_forces_ctx(self.h, ctx.h, law, G, out.ctypes.data if fetch else None,
                                                   ctypes.byref(ms) if timed else None), "nbx_leaf_plan_forces_ctx")
        if fetch:
            return (out, ms.value) if timed else out
        return ms.value if timed else None

    def get_forces(self) -> np.ndarray:
        out = np.empty((self.n, self.dim), dtype=np.float64)
        self._ck(self.lib.nbx_leaf_plan_get_forces(self.h, out.ctypes.data), "nbx_leaf_plan_get_forces")
        return out

    def kick_drift(self, ctx: "Context", dt: float):
        self._ck(self.lib.nbx_leaf_plan_kick_drift(self.h, ctx.h, float(dt)), "nbx_leaf_plan_kick_drift")

    def step(self, ctx: "Context", law: int, G: float, dt: float, nsteps: int):
        """nsteps x {forces_ctx(fetch=False); kick_drift} in one call (nbx_leaf_plan_step)."""
        self._ck(self.lib.nbx_leaf_plan_step(self.h, ctx.h, int(law), float(G), float(dt), int(nsteps)), "nbx_leaf_plan_step")

    def time_kernel(self, law: int, reps: int) -> float:
        """Measurement: mean ms of the second half of `reps` back-to-back launches of the pair kernel."""
        ms = ctypes.c_float(0.0)
        self._ck(self.lib.nbx_leaf_plan_time_kernel(self.h, law, reps, ctypes.byref(ms)), "nbx_leaf_plan_time_kernel")
        return ms.value


class Context:
    """Device-resident shard of an N-body system (nbx_ctx_* of include/nbody_hip.h)."""

    def __init__(self, n_total: int, dim: int = 3, device: int = 0, n_shards: int = 1, shard: int = 0):
        self.lib = load_library()
        self.n_total, self.dim, self.device, self.n_shards, self.shard = n_total, dim, device, n_shards, shard
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.nbx_ctx_create(ctypes.byref(h), device, dim, n_total, n_shards, shard),
               "nbx_ctx_create")
        self.h = h
        sl, sp = ctypes.c_size_t(), ctypes.c_size_t()
        _check(self.lib, self.lib.nbx_ctx_gather_layout(self.h, ctypes.byref(sl), ctypes.byref(sp), None, None),
               "nbx_ctx_gather_layout")
        self.shard_len, self.shard_pad = sl.value, sp.value
        lo = shard * self.shard_len
        self.count = max(0, min(self.shard_len, n_total - lo))

    def close(self):
        if getattr(self, "h", None):
            self.lib.nbx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc, where):
        _check(self.lib, rc, where)

    def set_stream(self, raw_stream: int):
        self._ck(self.lib.nbx_ctx_set_stream(self.h, ctypes.c_void_p(raw_stream)), "nbx_ctx_set_stream")

    def set_gather_buffers(self, pos_ptr: int, mass_ptr: int):
        self._ck(self.lib.nbx_ctx_set_gather_buffers(self.h, ctypes.c_void_p(pos_ptr), ctypes.c_void_p(mass_ptr)),
                 "nbx_ctx_set_gather_buffers")

    def set_tuning(self, source_splits: int = 0, variant: int = -1):
        self._ck(self.lib.nbx_ctx_set_tuning(self.h, source_splits, variant), "nbx_ctx_set_tuning")

    def set_softening(self, epsilon: float):
        """Extension: Plummer-softened pair law (epsilon = 0 restores the reference's unsoftened law)."""
        self._ck(self.lib.nbx_ctx_set_softening(self.h, float(epsilon)), "nbx_ctx_set_softening")

    def close_set_mode(self) -> Tuple[str, int, int]:
        """(mode, candidates seen, bad targets seen): how the fast path currently keeps the reference's skip rule."""
        m, a, b = ctypes.c_int(0), ctypes.c_uint(0), ctypes.c_uint(0)
        self._ck(self.lib.nbx_ctx_close_set_mode(self.h, ctypes.byref(m), ctypes.byref(a), ctypes.byref(b)), "nbx_ctx_close_set_mode")
        return ("candidate_pairs", "sorted_cells", "guarded_kernel", "none")[m.value], a.value, b.value

    def set_law(self, law: int):
        """Extension: FORCE_LAW_REFERENCE (default) or FORCE_LAW_NEWTON (attractive, softened; needs set_softening > 0)."""
        self._ck(self.lib.nbx_ctx_set_law(self.h, int(law)), "nbx_ctx_set_law")

    def set_refine(self, rel_tolerance: float, sigma_factor: float = 0.0):
        """Mixed mode: fp32 for all targets, the strict fp64 kernel for those whose fp32 sum cannot be trusted to
        rel_tolerance (0 switches it off); see nbx_ctx_set_refine in include/nbody_hip.h."""
        self._ck(self.lib.nbx_ctx_set_refine(self.h, float(rel_tolerance), float(sigma_factor)), "nbx_ctx_set_refine")

    def refine_stats(self) -> Tuple[int, int]:
        """(selected, refined) of the last mixed-mode evaluation."""
        a, b = ctypes.c_uint(0), ctypes.c_uint(0)
        self._ck(self.lib.nbx_ctx_refine_stats(self.h, ctypes.byref(a), ctypes.byref(b)), "nbx_ctx_refine_stats")
        return a.value, b.value

    def aux(self) -> np.ndarray:
        """Per-target statistic of the last evaluation: Q_i (mixed mode) or S_i = sum_j |a_ij| (strict_f64_t4_mag)."""
        out = np.empty(self.count, dtype=np.float64)
        self._ck(self.lib.nbx_ctx_get_aux(self.h, out.ctypes.data), "nbx_ctx_get_aux")
        return out

    def effective_tuning(self) -> Tuple[str, int]:
        v, s = ctypes.c_int(0), ctypes.c_int(0)
        self._ck(self.lib.nbx_ctx_effective_tuning(self.h, ctypes.byref(v), ctypes.byref(s)), "nbx_ctx_effective_tuning")
        return self.lib.nbx_variant_name(v.value).decode(), s.value

    def upload(self, bodies: np.ndarray):
        b, dim = _as_bodies(bodies)
        if dim != self.dim or b.shape[0] != self.n_total:
            raise ValueError("bodies shape does not match the context")
        self._ck(self.lib.nbx_ctx_upload_bodies(self.h, b.ctypes.data, b.shape[1] * 8), "nbx_ctx_upload_bodies")

    def upload_shard(self, shard_bodies: np.ndarray) -> Tuple[float, float]:
        """Step 1 of the sharded upload (nbx_ctx_upload_shard): this context's own bodies only; returns (max |mass|, max |coordinate|)
        of the shard.  The caller then fills the other chunks of the exchange buffers and calls upload_finish."""
        b, dim = _as_bodies(shard_bodies)
        if dim != self.dim or b.shape[0] != self.count:
            raise ValueError("shard_bodies must hold exactly this shard's bodies")
        m, x = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._ck(self.lib.nbx_ctx_upload_shard(self.h, b.ctypes.data if self.count else None, b.shape[1] * 8, ctypes.byref(m), ctypes.byref(x)),
                 "nbx_ctx_upload_shard")
        return m.value, x.value

    def upload_finish(self, max_abs_mass_all: float, max_abs_coord_all: float):
        self._ck(self.lib.nbx_ctx_upload_finish(self.h, float(max_abs_mass_all), float(max_abs_coord_all)), "nbx_ctx_upload_finish")

    def compute_accel(self, which: int = SRC_ALL):
        self._ck(self.lib.nbx_ctx_compute_accel(self.h, which), "nbx_ctx_compute_accel")

    def kick_drift(self, dt: float, G: float = REFERENCE_G):
        self._ck(self.lib.nbx_ctx_kick_drift(self.h, G, dt), "nbx_ctx_kick_drift")

    def kick_drift2(self, dt_kick: float, dt_drift: float, G: float = REFERENCE_G):
        self._ck(self.lib.nbx_ctx_kick_drift2(self.h, G, dt_kick, dt_drift), "nbx_ctx_kick_drift2")

    def step_kdk(self, dt: float, nsteps: int = 1, G: float = REFERENCE_G):
        """Extension: synchronised kick-drift-kick leapfrog (second order) from the same two helpers."""
        self._ck(self.lib.nbx_ctx_step_kdk(self.h, G, dt, nsteps), "nbx_ctx_step_kdk")

    def step(self, dt: float, nsteps: int = 1, G: float = REFERENCE_G):
        self._ck(self.lib.nbx_ctx_step(self.h, G, dt, nsteps), "nbx_ctx_step")

    def forces(self, G: float = REFERENCE_G) -> np.ndarray:
        out = np.empty((self.count, self.dim), dtype=np.float64)
        self._ck(self.lib.nbx_ctx_get_forces(self.h, G, out.ctypes.data), "nbx_ctx_get_forces")
        return out

    def accuracy(self, reference_forces: np.ndarray, G: float = REFERENCE_G) -> float:
        """Percent of this shard's bodies within the reference's 1 % rule of reference_forces (utils.h:170-219),
        computed on the device."""
        r = np.ascontiguousarray(reference_forces, dtype=np.float64)
        if r.shape != (self.count, self.dim):
            raise ValueError("reference_forces must be [count, dim]")
        pct = ctypes.c_double(0.0)
        self._ck(self.lib.nbx_ctx_accuracy(self.h, G, r.ctypes.data, ctypes.byref(pct)), "nbx_ctx_accuracy")
        return pct.value

    def accel(self) -> np.ndarray:
        out = np.empty((self.dim, self.count), dtype=np.float32)
        self._ck(self.lib.nbx_ctx_get_accel(self.h, out.ctypes.data), "nbx_ctx_get_accel")
        return out

    def download(self, bodies: np.ndarray):
        b, dim = _as_bodies(bodies, writable=True)
        if dim != self.dim or b.shape[0] != self.n_total:
            raise ValueError("bodies shape does not match the context")
        self._ck(self.lib.nbx_ctx_download_bodies(self.h, b.ctypes.data, b.shape[1] * 8), "nbx_ctx_download_bodies")

    def energy(self, G: float = REFERENCE_G) -> Tuple[float, float]:
        """(kinetic, potential) share of this shard under the potential that matches the reference law."""
        ke, pe = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._ck(self.lib.nbx_ctx_energy(self.h, G, ctypes.byref(ke), ctypes.byref(pe)), "nbx_ctx_energy")
        return ke.value, pe.value

    def synchronize(self):
        self._ck(self.lib.nbx_ctx_synchronize(self.h), "nbx_ctx_synchronize")

    def kernel_time(self) -> Tuple[float, int]:
        ms, cnt = ctypes.c_float(0.0), ctypes.c_int(0)
        self._ck(self.lib.nbx_ctx_kernel_time(self.h, ctypes.byref(ms), ctypes.byref(cnt)), "nbx_ctx_kernel_time")
        return ms.value, cnt.value


    def enable_clock_stamps(self, on: bool = True):
        self._ck(self.lib.nbx_ctx_enable_clock_stamps(self.h, 1 if on else 0), "nbx_ctx_enable_clock_stamps")

    def shader_clock(self) -> dict:
        """Shader clock held by the last stamped force launch: median / min / max MHz over its workgroups."""
        med, lo, hi, n = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int(0)
        self._ck(self.lib.nbx_ctx_shader_clock(self.h, ctypes.byref(med), ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(n)), "nbx_ctx_shader_clock")
        return {"median_mhz": med.value, "min_mhz": lo.value, "max_mhz": hi.value, "workgroups": n.value}

    def refine_time(self) -> float:
        """Total ms of the mixed mode's kernels behind the evaluations the last kernel_time() call covered."""
        ms = ctypes.c_float(0.0)
        self._ck(self.lib.nbx_ctx_refine_time(self.h, ctypes.byref(ms)), "nbx_ctx_refine_time")
        return ms.value


class Node:
    """Single-process multi-GPU node (nbx_node_* of include/nbody_hip.h): one rank per entry of `devices`
    (a device may repeat: virtual ranks on one GPU)."""

    def __init__(self, n_total: int, dim: int, devices, exchange: int = EXCHANGE_AUTO):
        self.lib = load_library()
        self.n_total, self.dim, self.devices = n_total, dim, list(devices)
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.nbx_node_create(ctypes.byref(h), len(self.devices), arr, dim, n_total, exchange), "nbx_node_create")
        self.h = h
        m = ctypes.c_int(0)
        _check(self.lib, self.lib.nbx_node_exchange_mode(self.h, ctypes.byref(m)), "nbx_node_exchange_mode")
        self.exchange = m.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.nbx_node_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc, where):
        _check(self.lib, rc, where)

    def upload(self, bodies: np.ndarray):
        b, dim = _as_bodies(bodies)
        if dim != self.dim or b.shape[0] != self.n_total:
            raise ValueError("bodies shape does not match the node")
        self._ck(self.lib.nbx_node_upload_bodies(self.h, b.ctypes.data, b.shape[1] * 8), "nbx_node_upload_bodies")

    def verify_exchange(self) -> int:
        """Poisoned-buffer self-check of the position exchange; returns the number of fp32 values that did not arrive."""
        bad = ctypes.c_size_t(0)
        self._ck(self.lib.nbx_node_verify_exchange(self.h, ctypes.byref(bad)), "nbx_node_verify_exchange")
        return bad.value

    def set_tuning(self, source_splits: int = 0, variant: int = -1):
        self._ck(self.lib.nbx_node_set_tuning(self.h, source_splits, variant), "nbx_node_set_tuning")

    def set_softening(self, epsilon: float):
        self._ck(self.lib.nbx_node_set_softening(self.h, float(epsilon)), "nbx_node_set_softening")

    def set_law(self, law: int):
        self._ck(self.lib.nbx_node_set_law(self.h, int(law)), "nbx_node_set_law")

    def set_refine(self, rel_tolerance: float, sigma_factor: float = 0.0):
        self._ck(self.lib.nbx_node_set_refine(self.h, float(rel_tolerance), float(sigma_factor)), "nbx_node_set_refine")

    def enable_timing(self, on: bool = True):
        self._ck(self.lib.nbx_node_enable_timing(self.h, 1 if on else 0), "nbx_node_enable_timing")

    def pass_times(self, rank: int) -> dict:
        """Per-rank figures of the last timed evaluation (nbx_node_pass_times)."""
        dev, tg = ctypes.c_int(0), ctypes.c_size_t(0)
        l, r, x, h = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int(0)
        self._ck(self.lib.nbx_node_pass_times(self.h, rank, ctypes.byref(dev), ctypes.byref(tg), ctypes.byref(l), ctypes.byref(r),
                                              ctypes.byref(x), ctypes.byref(h)), "nbx_node_pass_times")
        return {"rank": rank, "device": dev.value, "targets": tg.value, "local_ms": l.value, "remote_ms": r.value,
                "exchange_ms": x.value, "exchange_hidden": bool(h.value)}

    def refine_stats(self) -> Tuple[int, int]:
        """(selected, refined) of the last mixed-mode evaluation, summed over the ranks."""
        a, b = ctypes.c_uint(0), ctypes.c_uint(0)
        self._ck(self.lib.nbx_node_refine_stats(self.h, ctypes.byref(a), ctypes.byref(b)), "nbx_node_refine_stats")
        return a.value, b.value

    def forces(self, G: float = REFERENCE_G) -> np.ndarray:
        out = np.empty((self.n_total, self.dim), dtype=np.float64)
        self._ck(self.lib.nbx_node_compute_forces(self.h, G, out.ctypes.data), "nbx_node_compute_forces")
        return out

    def step_kdk(self, dt: float, nsteps: int = 1, G: float = REFERENCE_G):
        self._ck(self.lib.nbx_node_step_kdk(self.h, G, dt, nsteps), "nbx_node_step_kdk")

    def step(self, dt: float, nsteps: int = 1, G: float = REFERENCE_G):
        self._ck(self.lib.nbx_node_step(self.h, G, dt, nsteps), "nbx_node_step")

    def synchronize(self):
        self._ck(self.lib.nbx_node_synchronize(self.h), "nbx_node_synchronize")

    def download(self, bodies: np.ndarray):
        b, dim = _as_bodies(bodies, writable=True)
        self._ck(self.lib.nbx_node_download_bodies(self.h, b.ctypes.data, b.shape[1] * 8), "nbx_node_download_bodies")

    def energy(self, G: float = REFERENCE_G) -> Tuple[float, float]:
        ke, pe = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._ck(self.lib.nbx_node_energy(self.h, G, ctypes.byref(ke), ctypes.byref(pe)), "nbx_node_energy")
        return ke.value, pe.value

    def kernel_time(self) -> Tuple[float, int]:
        ms, cnt = ctypes.c_float(0.0), ctypes.c_int(0)
        self._ck(self.lib.nbx_node_kernel_time(self.h, ctypes.byref(ms), ctypes.byref(cnt)), "nbx_node_kernel_time")
        return ms.value, cnt.value
