"""The stated tolerances, pinned: any change to one of these constants is a one-line diff in THIS file, named for what it is.

TOL_REL / TOL_BACKWARD / KAPPA_WELL were sized on 12.6 million bodies in round 3 (tests/oracle_lib.py) and
frozen there; TOL_BACKWARD_SMALL_N keeps round 2's tighter bound for sums over <= 65,536 sources.  The mixed mode's sigma
factors are the library's calibration of its selection rule (csrc/nbx_api.hip kRefineSigmaDefault*) and
are read back through the C ABI (nbx_refine_sigma_default) -- no GPU needed.  The default per-body tolerance of every entry
point is the north star's 1e-5 (nbx_get_default_refine)."""
import all_bodies
import oracle_lib


def test_parity_tolerances_are_the_frozen_ones():
    assert oracle_lib.TOL_REL == 1.0e-5            # north star: accelerations within 1e-5 relative of the sequential reference
    assert oracle_lib.TOL_BACKWARD == 1.0e-5       # (T1) every body, every input, against the magnitude sum
    assert oracle_lib.KAPPA_WELL == 4.0            # (T2) plain 1e-5 for kappa <= 4 in plain fp32
    assert oracle_lib.TOL_BACKWARD_SMALL_N == 4.0e-6 and oracle_lib.SMALL_N == 65536
    assert all_bodies.TOL_STRICT_REL == 1.0e-9 and all_bodies.TOL_STRICT_BACKWARD == 2.0e-12   # strict fp64 kernel vs oracle


def test_mixed_mode_calibration_is_the_frozen_one(nbx):
    lib = nbx.load_library()
    assert lib.nbx_refine_sigma_default(3) == SIGMA_3D
    assert lib.nbx_refine_sigma_default(2) == SIGMA_2D
    assert nbx.get_default_refine() == (1.0e-5, 0.0)   # every entry point: mixed mode at the north star's tolerance


# sigma factors of the selection rule  tol |a_i| < sigma u sqrt(Q_i)  (include/nbody_hip.h nbx_ctx_set_refine)
# round 4: calibrated for the three-level default kernel on ten all-bodies surveys (profiles/r4/all_bodies_3l.jsonl); round 3's
# 48 / 64 belonged to the two-level kernel and still apply to it when it is selected by name
SIGMA_3D = 24.0
SIGMA_2D = 32.0
