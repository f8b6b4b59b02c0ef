"""ctypes mirror of include/cmlhip.h (struct layouts + function prototypes).

The C header is authoritative; tests/test_abi.py checks that every symbol declared there is
exported by libcmlhip.so and that the struct sizes here match the compiled ones
(cmlhip_sizeof_*).
"""
import ctypes as C

PATTERN = 8
CPARS = 4
MAX_FRAMES = 32
RJ_FLOATS = 74

OK, ERR_INVALID, ERR_HIP, ERR_NONFINITE, ERR_NOT_FOUND, ERR_STATE, ERR_TIMEOUT = range(7)
RES_IN, RES_OOB, RES_OUTLIER = 0, 1, 2
MODE_ACTIVE, MODE_LINEARIZED, MODE_MARGINALIZED = 0, 1, 2
TEXEL_F32, TEXEL_F16 = 0, 1

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_ubyte_p = C.POINTER(C.c_ubyte)


class Limits(C.Structure):
    _fields_ = [("device_id", C.c_int), ("max_frames", C.c_int), ("max_points", C.c_int),
                ("max_residuals", C.c_int), ("max_tracker_points", C.c_int),
                ("max_reproj_obs", C.c_int), ("texel_format", C.c_int)]


class TrackerParams(C.Structure):
    _fields_ = [("huber", C.c_float), ("cutoff", C.c_float), ("cutoff_base", C.c_float),
                ("scale_rot", C.c_float), ("scale_trans", C.c_float), ("scale_a", C.c_float),
                ("scale_b", C.c_float)]


class TrackerResult(C.Structure):
    _fields_ = [("E", C.c_float), ("numTermsInE", C.c_int), ("numSaturated", C.c_int),
                ("numRobust", C.c_int), ("numWarped", C.c_int), ("flow", C.c_float * 3),
                ("H", C.c_double * 64), ("b", C.c_double * 8), ("H9", C.c_float * 81)]


TRACKER_MAX_STEPS = 256


class TrackerHypothesis(C.Structure):
    _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3)]


class TrackerOptResult(C.Structure):
    _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3), ("a", C.c_double), ("b", C.c_double),
                ("isCorrect", C.c_int), ("tooManySaturated", C.c_int),
                ("E", C.c_float * 5), ("numTermsInE", C.c_int * 5), ("numSaturated", C.c_int * 5), ("numRobust", C.c_int * 5), ("iterations", C.c_int * 5),
                ("levelCutoffRepeat", C.c_double * 5), ("relAff", C.c_double * 2), ("covariance", C.c_double * 6),
                ("flow", C.c_float * 3),
                ("n_pass", C.c_int), ("pass_level", C.c_int * 8), ("pass_rmse", C.c_double * 8),
                ("n_steps", C.c_int), ("step_level", C.c_ubyte * TRACKER_MAX_STEPS), ("step_accept", C.c_ubyte * TRACKER_MAX_STEPS),
                ("eval_us", C.c_double), ("algebra_us", C.c_double)]


class BAParams(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("w", C.c_int), ("h", C.c_int), ("huber", C.c_float), ("outlier_th_sum", C.c_float),
                ("scale_f", C.c_double), ("scale_c", C.c_double),
                ("optimize_a", C.c_int), ("optimize_b", C.c_int)]


class BAFrame(C.Structure):
    _fields_ = [("image_id", C.c_uint64), ("frame_energy_th", C.c_float), ("b0", C.c_float)]


class BAPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("idepth", C.c_double),
                ("idepth_zero", C.c_float), ("prior", C.c_float),
                ("colors", C.c_float * PATTERN), ("weights", C.c_float * PATTERN), ("host", C.c_int)]


class BAResidual(C.Structure):
    _fields_ = [("point", C.c_int), ("target", C.c_int), ("state", C.c_int), ("is_linearized", C.c_int)]


class BAPair(C.Structure):
    _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3), ("R0", C.c_double * 9),
                ("t0", C.c_double * 3), ("aff_a", C.c_double), ("aff_b", C.c_double)]


class BALinResult(C.Structure):
    _fields_ = [("energy", C.c_double), ("n_in", C.c_int), ("n_oob", C.c_int), ("n_outlier", C.c_int),
                ("new_frame_energy_th", C.c_float)]


class BAAccumIn(C.Structure):
    _fields_ = [("adHost", c_double_p), ("adTarget", c_double_p), ("adHTdeltaF", c_float_p),
                ("cdelta", c_double_p), ("prior", c_double_p), ("delta_prior", c_double_p),
                ("cprior", c_double_p)]


class ReprojObs(C.Structure):
    _fields_ = [("frame", C.c_int), ("point", C.c_int), ("gx", C.c_double), ("gy", C.c_double)]


# numpy dtypes with the same layout (align=True reproduces the C padding)
import numpy as np  # noqa: E402

BA_POINT_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("idepth", "f8"), ("idepth_zero", "f4"),
                           ("prior", "f4"), ("colors", "f4", (PATTERN,)), ("weights", "f4", (PATTERN,)),
                           ("host", "i4")], align=True)
BA_RESIDUAL_DTYPE = np.dtype([("point", "i4"), ("target", "i4"), ("state", "i4"), ("is_linearized", "i4")],
                             align=True)
BA_FRAME_DTYPE = np.dtype([("image_id", "u8"), ("frame_energy_th", "f4"), ("b0", "f4")], align=True)
BA_PAIR_DTYPE = np.dtype([("R", "f8", (9,)), ("t", "f8", (3,)), ("R0", "f8", (9,)), ("t0", "f8", (3,)),
                          ("aff_a", "f8"), ("aff_b", "f8")], align=True)
REPROJ_OBS_DTYPE = np.dtype([("frame", "i4"), ("point", "i4"), ("gx", "f8"), ("gy", "f8")], align=True)

assert BA_POINT_DTYPE.itemsize == C.sizeof(BAPoint)
assert BA_RESIDUAL_DTYPE.itemsize == C.sizeof(BAResidual)
assert BA_FRAME_DTYPE.itemsize == C.sizeof(BAFrame)
assert BA_PAIR_DTYPE.itemsize == C.sizeof(BAPair)
assert REPROJ_OBS_DTYPE.itemsize == C.sizeof(ReprojObs)


def default_ba_params(fx, fy, cx, cy, w, h):
    """Appendix A of SURVEY.md / BA.h:235-288 defaults."""
    return BAParams(fx=fx, fy=fy, cx=cx, cy=cy, w=w, h=h, huber=9.0, outlier_th_sum=2500.0,
                    scale_f=50.0, scale_c=50.0, optimize_a=1, optimize_b=1)


def default_tracker_params(cutoff_repeat=1.0):
    """TR.h:473-520 defaults."""
    return TrackerParams(huber=9.0, cutoff=20.0 * cutoff_repeat, cutoff_base=20.0,
                         scale_rot=1.0, scale_trans=0.5, scale_a=10.0, scale_b=1000.0)


class BAFrameState(C.Structure):
    """cmlhip_ba_frame_state (device-resident iterations)."""
    _fields_ = [("eval_q", C.c_double * 4), ("eval_t", C.c_double * 3), ("state", C.c_double * 10), ("state_zero", C.c_double * 10),
                ("prior_zero", C.c_double * 8), ("ab_exposure", C.c_double), ("fix_pose", C.c_int), ("pad", C.c_int)]


class BAResidentOut(C.Structure):
    """cmlhip_ba_resident_out (cmlhip_ba_finish_run: the resident loop's results, read back with the closing pass in one copy)."""
    _fields_ = [("frames", C.POINTER(BAFrameState)), ("pre_w2c", c_double_p), ("first", C.POINTER(BALinResult)), ("last", C.POINTER(BALinResult)),
                ("iterations", C.POINTER(C.c_int)), ("energies", c_double_p), ("capacity", C.c_int), ("x", c_double_p),
                ("state_good", C.POINTER(C.c_ubyte)), ("hdi", c_float_p)]


# ---- immature points (DSOTracer)
IPS_GOOD, IPS_OOB, IPS_OUTLIER, IPS_SKIPPED, IPS_BADCONDITION, IPS_UNINITIALIZED = range(6)
IMMATURE_POINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("host", "<i4"), ("last_status", "<i4"), ("idepth_min", "<f8"), ("idepth_max", "<f8"),
                                 ("gradH", "<f8", (4,)), ("energy_th", "<f8"), ("quality", "<f8"), ("last_uv", "<f8", (2,)),
                                 ("last_pixel_interval", "<f8"), ("gray", "<f4", (8,)), ("dpatch", "<f4", (24,))])
IMMATURE_STATE_DTYPE = np.dtype([("idepth_min", "<f8"), ("idepth_max", "<f8"), ("quality", "<f8"), ("last_uv", "<f8", (2,)), ("last_pixel_interval", "<f8"),
                                 ("last_status", "<i4"), ("pad", "<i4")])        # cmlhip_immature_state
TRACE_PAIR_DTYPE = np.dtype([("KRKi", "<f8", (9,)), ("Kt", "<f8", (3,)), ("aff_a", "<f8"), ("aff_b", "<f8")])
ACTIVATION_PAIR_DTYPE = np.dtype([("R", "<f8", (9,)), ("t", "<f8", (3,)), ("aff_a", "<f8"), ("aff_b", "<f8")])


INIT_POINT_DTYPE = np.dtype([("p_pattern", "<f4", (8, 3)), ("color", "<f4", (8,)), ("idepth_new", "<f4"), ("iR", "<f4"), ("outlier_th", "<f4"),
                             ("energy", "<f4", (2,)), ("is_good", "<i4"), ("is_good_new", "<i4"), ("energy_new", "<f4", (2,)),
                             ("maxstep", "<f4"), ("last_hessian_new", "<f4"), ("jb", "<f4", (10,)), ("pad", "<f4", (3,))])
assert INIT_POINT_DTYPE.itemsize == 224


class InitParams(C.Structure):
    """cmlhip_init_params: the per-evaluation constants of DSOInitializer::calcResAndGS (DSOInitializer.cpp:451-480)."""
    _fields_ = [("RKi", C.c_float * 9), ("t", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("aff_a", C.c_float), ("aff_b", C.c_float), ("huber", C.c_float), ("alpha_w", C.c_float), ("alpha_k", C.c_float),
                ("coupling_weight", C.c_float), ("tlog", C.c_float * 3), ("pad", C.c_float), ("t_sqnorm", C.c_double)]


PNP_MATCH_DTYPE = np.dtype([("X", "<f8", (3,)), ("obs", "<f8", (2,)), ("inv_sigma2", "<f8"), ("info", "<f8")])
assert PNP_MATCH_DTYPE.itemsize == 56
PNP_LEVENBERG, PNP_GAUSS_NEWTON = 0, 1


class PnpResult(C.Structure):
    """cmlhip_pnp_result: IndirectCameraOptimizerResult (IndirectCameraOptimizer.h) + per-round diagnostics."""
    _fields_ = [("is_ok", C.c_int), ("rounds", C.c_int), ("n_bad", C.c_int), ("lm_iterations", C.c_int * 4), ("pad", C.c_int),
                ("R", C.c_double * 9), ("t", C.c_double * 3), ("covariance", C.c_double * 6), ("chi2", C.c_double * 4)]


LBA_FRAME_DTYPE = np.dtype([("R", "<f8", (9,)), ("t", "<f8", (3,)), ("K", "<f8", (4,)), ("fixed", "<i4"), ("pad", "<i4")])
LBA_EDGE_DTYPE = np.dtype([("frame", "<i4"), ("pad", "<i4"), ("obs", "<f8", (2,)), ("inv_sigma2", "<f8")])
assert LBA_FRAME_DTYPE.itemsize == 136 and LBA_EDGE_DTYPE.itemsize == 32


class LbaResult(C.Structure):
    """cmlhip_lba_result"""
    _fields_ = [("ok", C.c_int), ("n_bad", C.c_int), ("iterations_done", C.c_int * 2), ("chi2", C.c_double * 2)]


class TracerParams(C.Structure):
    _fields_ = [("max_pix_search", C.c_double), ("max_slack_interval", C.c_double), ("trace_step_size", C.c_double),
                ("min_improvement_factor", C.c_double), ("min_trace_test_radius", C.c_double), ("extra_slack_on_th", C.c_double),
                ("huber_th", C.c_double), ("outlier_th_sum_component", C.c_double), ("min_idepth_h_act", C.c_double),
                ("gn_its_on_activation", C.c_int), ("pad", C.c_int)]


def default_tracer_params():
    """DSOTracer.h:188-206: Parameters hold doubles initialised from float literals."""
    f = lambda v: float(np.float32(v))
    return TracerParams(f(0.027), f(1.5), f(1.0), f(2.0), f(2.0), f(1.2), f(9.0), f(50.0 * 50.0), f(100.0), 3, 0)
