"""CPU: the C-ABI library loads without a GPU, exports every symbol include/nbody_hip.h declares,
validates arguments, and FAILS LOUDLY (no CPU fallback) when no HIP device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nbody_hip.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nbx_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree(nbx):
    decl = declared_symbols()
    assert len(decl) >= 20
    bound = sorted(name for name, _, _ in nbx.ABI)
    assert decl == bound, "capi.ABI must type exactly the entry points the header declares"


def test_library_exports_every_symbol(nbx):
    lib = ctypes.CDLL(nbx.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/nbody_hip.h but not exported"
    assert nbx.load_library().nbx_abi_version() == 5


def test_library_exports_nothing_but_the_header(nbx):
    """The other direction: `nm -D --defined-only` of the shared library lists the header's entry points and nothing else
    (built with -fvisibility=hidden and a linker version script, csrc/libnbody_hip.map) -- no mangled internals, no helper
    that happens to sit in an extern "C" block."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", nbx.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip()})
    assert exported == declared_symbols(), sorted(set(exported) ^ set(declared_symbols()))


def test_header_cites_reference_interfaces():
    txt = open(HEADER).read()
    for cite in ("methods.h:29-37", "methods.h:85-91", "methods.cpp:425-450", "body.h:8-11", "utils.h:87-104"):
        assert cite in txt


def test_strerror_and_variants(nbx):
    lib = nbx.load_library()
    assert lib.nbx_strerror(0) == b"ok"
    assert b"no CPU fallback" in lib.nbx_strerror(2)
    names = nbx.variants()
    assert len(names) == len(set(names)) and 0 <= lib.nbx_default_variant() < len(names)
    assert "strict_f64_t4" in names and "strict_f64_t4_mag" in names and names[lib.nbx_default_variant()].startswith("fastpk")
    assert lib.nbx_variant_name(len(names)) == b"?"


def _no_gpu(nbx):
    try:
        return nbx.device_count() == 0
    except nbx.NbxError:
        return True


def test_no_device_is_a_loud_failure(nbx):
    if not _no_gpu(nbx):
        pytest.skip("a GPU is present")
    b = np.zeros((4, 7))
    with pytest.raises(nbx.NbxError) as e:
        nbx.brute_force_hip_n_body(b)
    assert e.value.status == 2 and "no CPU fallback" in str(e.value)
    with pytest.raises(nbx.NbxError):
        nbx.leapfrog_hip_n_body(b, 1.0, 1)
    with pytest.raises(nbx.NbxError):
        nbx.Context(4, 3)


def test_argument_validation(nbx):
    lib = nbx.load_library()
    h = ctypes.c_void_p()
    assert lib.nbx_ctx_create(None, 0, 3, 8, 1, 0) == 1
    assert lib.nbx_ctx_create(ctypes.byref(h), 0, 4, 8, 1, 0) == 1          # dim must be 2 or 3
    assert lib.nbx_ctx_create(ctypes.byref(h), 0, 3, 8, 2, 2) == 1          # shard out of range
    assert lib.nbx_ctx_create(ctypes.byref(h), 0, 3, 8, 0, 0) == 1
    assert lib.nbx_device_count(None) == 1
    for fn, args in ((lib.nbx_ctx_compute_accel, (None, 0)), (lib.nbx_ctx_kick_drift, (None, 1.0, 1.0)),
                     (lib.nbx_ctx_step, (None, 1.0, 1.0, 1)), (lib.nbx_ctx_synchronize, (None,)),
                     (lib.nbx_ctx_set_tuning, (None, 0, 0)), (lib.nbx_ctx_set_refine, (None, 1e-5, 0.0)),
                     (lib.nbx_ctx_refine_stats, (None, None, None)), (lib.nbx_ctx_get_aux, (None, None)),
                     (lib.nbx_node_set_refine, (None, 1e-5, 0.0))):
        assert fn(*args) == 1
    assert lib.nbx_ctx_destroy(None) == 0
    assert lib.nbx_brute_force_forces(None, 4, 3, 56, 1.0, 0, None, None) == 1
    with pytest.raises(ValueError):
        nbx.brute_force_hip_n_body(np.zeros((4, 6)))
    with pytest.raises(ValueError):
        nbx.brute_force_hip_n_body(np.zeros((4, 7), dtype=np.float32))


def test_product_does_not_touch_the_oracle():
    """The shipped path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "nbody-simulation-parallel_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in txt and "oracle_lib" not in txt and "nbody_oracle" not in txt, f
    out = os.popen(f"ldd '{os.path.join(pkg, 'libnbody_hip.so')}'").read()
    assert "oracle" not in out and "nbody_ref" not in out
    # product-side tools and the harness sources stay clear of it too (checker-using scripts live under tests/measure/)
    for d in ("tools", "include"):
        for f in os.listdir(os.path.join(ROOT, d)):
            if f.endswith((".py", ".sh", ".h", ".hip")):
                txt = open(os.path.join(ROOT, d, f)).read()
                assert "oracle_lib" not in txt and "liboracle" not in txt and "oracle/" not in txt, f"{d}/{f}"


def test_leaf_pair_entry_validates_before_it_needs_a_device(nbx):
    """nbx_leaf_pair_forces checks every index array on the host first: malformed CSR input is NBX_ERR_INVALID even on a
    box without a GPU, a well-formed call then fails loudly with NO_DEVICE (no CPU fallback)."""
    b = np.zeros((10, 7))
    b[:, :3] = np.arange(30).reshape(10, 3)
    b[:, 6] = 1.0
    ok = (np.array([0, 5, 10]), np.arange(10), np.array([0, 1, 2]), np.array([0, 1]))
    bad = [
        (np.array([0, 5, 10]), np.arange(10), np.array([0, 1, 2]), np.array([0, 2])),            # source leaf out of range
        (np.array([0, 5, 10]), np.r_[np.arange(9), 10], np.array([0, 1, 2]), np.array([0, 1])),  # body out of range
        (np.array([0, 5, 10]), np.r_[np.arange(9), 0], np.array([0, 1, 2]), np.array([0, 1])),   # body in two leaves
        (np.array([0, 6, 5]), np.arange(6), np.array([0, 1, 2]), np.array([0, 1])),              # decreasing offsets
    ]
    for leaves in bad:
        with pytest.raises(nbx.NbxError) as e:
            nbx.leaf_pair_forces_hip(b, *leaves)
        assert e.value.status == 1, e.value          # NBX_ERR_INVALID
    with pytest.raises(nbx.NbxError) as e:
        nbx.leaf_pair_forces_hip(b, *ok, law=9)
    assert e.value.status == 1
    if _no_gpu(nbx):
        with pytest.raises(nbx.NbxError) as e:
            nbx.leaf_pair_forces_hip(b, *ok)
        assert e.value.status in (2, 3) and "no CPU fallback" in str(e.value)          # NO_DEVICE / HIP


def test_new_entry_points_validate_and_fail_loudly_without_a_device(nbx):
    """ABI 4's additions: the process-wide precision default needs no device; the leaf plan validates its CSR arrays on the host
    before it asks for a device; every plan / node / context entry rejects a null handle; without a GPU a well-formed plan is
    NO_DEVICE (no CPU fallback)."""
    lib = nbx.load_library()
    assert nbx.get_default_refine() == (1e-5, 0.0)
    try:
        nbx.set_default_refine(1e-6, 12.0)
        assert nbx.get_default_refine() == (1e-6, 12.0)
        for bad in ((1e-9, 0.0), (0.1, 0.0), (1e-5, -1.0), (float("nan"), 0.0)):
            with pytest.raises(nbx.NbxError) as e:
                nbx.set_default_refine(*bad)
            assert e.value.status == 1
        assert nbx.get_default_refine() == (1e-6, 12.0)           # a refused call changes nothing
    finally:
        nbx.set_default_refine(1e-5, 0.0)
    assert lib.nbx_refine_sigma_default(3) > 0 and lib.nbx_refine_sigma_default(2) >= lib.nbx_refine_sigma_default(3)
    ok = (np.array([0, 5, 10]), np.arange(10), np.array([0, 1, 2]), np.array([0, 1]))
    bad = [
        (np.array([0, 5, 10]), np.arange(10), np.array([0, 1, 2]), np.array([0, 2])),            # source leaf out of range
        (np.array([0, 5, 10]), np.r_[np.arange(9), 0], np.array([0, 1, 2]), np.array([0, 1])),   # body in two leaves
        (np.array([1, 5, 10]), np.arange(10), np.array([0, 1, 2]), np.array([0, 1])),            # offsets not from 0
    ]
    for leaves in bad:
        with pytest.raises(nbx.NbxError) as e:
            nbx.LeafPlan(10, 3, *leaves)
        assert e.value.status == 1
    with pytest.raises(nbx.NbxError) as e:
        nbx.LeafPlan(10, 4, *ok)                                  # dim
    assert e.value.status == 1
    if _no_gpu(nbx):
        with pytest.raises(nbx.NbxError) as e:
            nbx.LeafPlan(10, 3, *ok)
        assert e.value.status in (2, 3) and "no CPU fallback" in str(e.value)
        with pytest.raises(nbx.NbxError) as e:
            nbx.brute_force_hip_n_body(np.zeros((4, 7)), rel_tolerance=0.0)
        assert e.value.status == 2
    z = ctypes.c_float(0.0)
    for fn, args in ((lib.nbx_leaf_plan_forces, (None, None, 56, 1, 1.0, None, None)), (lib.nbx_leaf_plan_forces_ctx, (None, None, 1, 1.0, None, None)),
                     (lib.nbx_leaf_plan_get_forces, (None, None)), (lib.nbx_leaf_plan_kick_drift, (None, None, 1.0)), (lib.nbx_leaf_plan_step, (None, None, 1, 1.0, 1.0, 1)),
                     (lib.nbx_leaf_plan_time_kernel, (None, 1, 3, ctypes.byref(z))), (lib.nbx_leaf_plan_info, (None, None, None, None, None)),
                     (lib.nbx_ctx_upload_shard, (None, None, 56, None, None)), (lib.nbx_ctx_upload_finish, (None, 1.0, 1.0)),
                     (lib.nbx_ctx_refine_time, (None, ctypes.byref(z))), (lib.nbx_node_refine_stats, (None, None, None)),
                     (lib.nbx_node_enable_timing, (None, 1)), (lib.nbx_node_pass_times, (None, 0, None, None, None, None, None, None))):
        assert fn(*args) == 1, fn
    assert lib.nbx_leaf_plan_destroy(None) == 0
    assert lib.nbx_brute_force_forces_ex(None, 4, 3, 56, 1.0, 0, -1.0, None, None) == 1
