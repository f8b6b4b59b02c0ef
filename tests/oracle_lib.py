"""ctypes access to the oracle (oracle/libcml_oracle.so) and, when built, to oracle/_ref.

Checker only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from libcml_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def _ensure_built():
    so = os.path.join(ORACLE_DIR, "libcml_oracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("orc_base.c", "orc_ba.c", "orc_tracker.c", "cml_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libcml_oracle.so"], stdout=subprocess.DEVNULL)
    return so


class OrcSE3(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3)]


class OrcTrkStep(C.Structure):
    _fields_ = [("level", C.c_int), ("iteration", C.c_int), ("accept", C.c_int), ("lambda_", C.c_double), ("E_new", C.c_double),
                ("E_old", C.c_double), ("n_new", C.c_int), ("n_old", C.c_int)]


class OrcTrkProblem(C.Structure):
    _fields_ = [("levels", C.c_int), ("aos3", C.POINTER(C.c_float) * 5), ("w", C.c_int * 5), ("h", C.c_int * 5),
                ("uvic", C.POINTER(C.c_float) * 5), ("n", C.c_int * 5), ("K", C.c_double * 4),
                ("ref_a", C.c_double), ("ref_b", C.c_double), ("ref_t", C.c_double), ("new_t", C.c_double),
                ("prm", abi.TrackerParams), ("optimize_a", C.c_int), ("optimize_b", C.c_int), ("saturated_ratio_th", C.c_double),
                ("have_last", C.c_int), ("last_rmse", C.c_double * 5)]


class OrcTrkResult(C.Structure):
    _fields_ = [("isCorrect", C.c_int), ("tooManySaturated", C.c_int), ("E", C.c_double * 5), ("numTermsInE", C.c_int * 5),
                ("numSaturated", C.c_int * 5), ("numRobust", C.c_int * 5), ("levelCutoffRepeat", C.c_double * 5),
                ("relAff", C.c_double * 2), ("covariance", C.c_double * 6), ("flow", C.c_double * 3), ("n_steps", C.c_int)]


class OrcFrame(C.Structure):
    _fields_ = [("w2c_eval", OrcSE3), ("state", C.c_double * 10), ("state_zero", C.c_double * 10),
                ("state_scaled", C.c_double * 10), ("step", C.c_double * 10), ("state_backup", C.c_double * 10),
                ("ab_exposure", C.c_double), ("PRE_w2c", OrcSE3), ("PRE_c2w", OrcSE3),
                ("prior", C.c_double * 8), ("delta", C.c_double * 8), ("delta_prior", C.c_double * 8),
                ("prior_zero", C.c_double * 10), ("ns_pose", C.c_double * 36), ("ns_scale", C.c_double * 6),
                ("ns_affine", C.c_double * 8), ("keyid", C.c_int)]


class OrcScales(C.Structure):
    _fields_ = [("trans", C.c_double), ("rot", C.c_double), ("a", C.c_double), ("b", C.c_double)]


DEFAULT_SCALES = dict(trans=0.5, rot=1.0, a=10.0, b=1000.0)   # BA.h:246-249


class OrcBAWindow(C.Structure):
    """Mirror of orc_ba_window (oracle/cml_oracle.h) for field access from tests."""
    _fields_ = [("prm", abi.BAParams), ("N", C.c_int), ("P", C.c_int), ("R", C.c_int),
                ("image", C.POINTER(C.c_float) * abi.MAX_FRAMES),
                ("frame_energy_th", C.c_float * abi.MAX_FRAMES), ("b0", C.c_float * abi.MAX_FRAMES),
                ("pairs", C.POINTER(abi.BAPair)), ("points", C.POINTER(abi.BAPoint)),
                ("idepth_backup", abi.c_float_p),
                ("Hdd_accAF", abi.c_float_p), ("bd_accAF", abi.c_float_p), ("Hcd_accAF", abi.c_float_p),
                ("Hdd_accLF", abi.c_float_p), ("bd_accLF", abi.c_float_p), ("Hcd_accLF", abi.c_float_p),
                ("HdiF", abi.c_float_p), ("bdSumF", abi.c_float_p), ("step", abi.c_double_p),
                ("r_point", abi.c_int_p), ("r_target", abi.c_int_p), ("r_state", abi.c_int_p),
                ("r_new_state", abi.c_int_p), ("r_lin", abi.c_ubyte_p), ("r_good", abi.c_ubyte_p),
                ("r_energy", abi.c_float_p), ("r_new_energy", abi.c_float_p), ("r_new_energy_wo", abi.c_float_p),
                ("r_center", abi.c_float_p), ("rJ", abi.c_float_p), ("efsJ", abi.c_float_p),
                ("JpJdF", abi.c_float_p), ("res_toZeroF", abi.c_float_p),
                ("pair_of", abi.c_int_p), ("by_point_off", abi.c_int_p), ("by_point", abi.c_int_p),
                ("by_pair_off", abi.c_int_p), ("by_pair", abi.c_int_p),
                ("accA", abi.c_float_p), ("accL", abi.c_float_p), ("accA_num", abi.c_int_p), ("accL_num", abi.c_int_p)]


def _np(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype) if False else np.frombuffer(
        (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(C.addressof(ptr.contents)), dtype=dtype, count=n)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_ensure_built())
        L.orc_ba_create.restype = C.POINTER(OrcBAWindow)
        L.orc_ba_linearize_one.restype = C.c_double
        L.orc_ba_calc_l_energy.restype = C.c_double
        L.orc_ba_calc_m_energy.restype = C.c_double
        _lib = L
    return _lib


def ref():
    """oracle/_ref/libcml_ref.so or None. Built from /root/reference when that tree is present."""
    global _ref
    if _ref is None:
        so = os.path.join(ORACLE_DIR, "_ref", "libcml_ref.so")
        if not os.path.exists(so) and os.path.isdir("/root/reference/thirdparty/eigen"):
            subprocess.call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
        _ref = C.CDLL(so) if os.path.exists(so) else False
    return _ref or None


# ----------------------------------------------------------------------------- thin numpy wrappers
def se3_exp(xi):
    T = OrcSE3()
    lib().orc_se3_exp(ptr(f64(xi), C.c_double), C.byref(T))
    return T


def se3_from_Rt(R, t):
    T = OrcSE3()
    lib().orc_se3_from_Rt(ptr(f64(R).ravel(), C.c_double), ptr(f64(t), C.c_double), C.byref(T))
    return T


def se3_log(T):
    out = np.zeros(6)
    lib().orc_se3_log(C.byref(T), ptr(out, C.c_double))
    return out


def se3_matrix(T):
    R = np.zeros(9)
    lib().orc_se3_matrix(C.byref(T), ptr(R, C.c_double))
    return R.reshape(3, 3), np.array(T.t[:])


def se3_mul(A, B):
    T = OrcSE3()
    lib().orc_se3_mul(C.byref(A), C.byref(B), C.byref(T))
    return T


def se3_inv(A):
    T = OrcSE3()
    lib().orc_se3_inv(C.byref(A), C.byref(T))
    return T


def se3_adj(T):
    A = np.zeros(36)
    lib().orc_se3_adj(C.byref(T), ptr(A, C.c_double))
    return A.reshape(6, 6)


def se3_dx_exp_x(xi):
    J = np.zeros(42)
    lib().orc_se3_dx_exp_x(ptr(f64(xi), C.c_double), ptr(J, C.c_double))
    return J.reshape(7, 6)


def ldlt_solve(A, b):
    A = f64(A); b = f64(b)
    x = np.zeros(len(b))
    rc = lib().orc_ldlt_solve(ptr(A.ravel(), C.c_double), ptr(b, C.c_double), len(b), ptr(x, C.c_double))
    return x, rc


def inverse(A):
    A = f64(A); n = A.shape[0]
    out = np.zeros(n * n)
    lib().orc_inverse(ptr(A.ravel(), C.c_double), n, ptr(out, C.c_double))
    return out.reshape(n, n)


def orthogonalize(b, Ncols, delta=1e-5):
    b = f64(b).copy(); Ncols = f64(Ncols)   # Ncols: (m, n) rows = nullspace vectors
    m, n = Ncols.shape
    lib().orc_orthogonalize(ptr(b, C.c_double), n, ptr(Ncols.ravel(), C.c_double), m, C.c_double(delta))
    return b


def pyramid_sizes(w, h, max_levels=8):
    ws = (C.c_int * max_levels)(); hs = (C.c_int * max_levels)()
    n = lib().orc_pyramid_sizes(w, h, ws, hs, max_levels)
    return list(ws[:n]), list(hs[:n])


def build_pyramid(gray, levels=None):
    """gray (h,w) f32 -> (gray levels, AoS3 gradient levels) by the reference rules."""
    gray = f32(gray)
    h, w = gray.shape
    ws, hs = pyramid_sizes(w, h)
    if levels is not None:
        ws, hs = ws[:levels], hs[:levels]
    grays, grads = [gray], []
    for l in range(1, len(ws)):
        prev = grays[-1]
        out = np.zeros((hs[l], ws[l]), np.float32)
        lib().orc_reduce_by_two(ptr(prev, C.c_float), prev.shape[1], prev.shape[0], ptr(out, C.c_float))
        grays.append(out)
    for g in grays:
        o = np.zeros((g.shape[0], g.shape[1], 3), np.float32)
        lib().orc_gradient_image(ptr(g, C.c_float), g.shape[1], g.shape[0], ptr(o, C.c_float))
        grads.append(o)
    return grays, grads


def interpolate3(aos3, x, y):
    out = np.zeros(3, np.float32)
    lib().orc_interpolate3(ptr(aos3, C.c_float), aos3.shape[1], C.c_float(x), C.c_float(y), ptr(out, C.c_float))
    return out


def marginalize_frame(HM, bM, N, frame, prior, delta_prior):
    """orc_ba_marginalize_frame (BA.cpp:483-558): returns the (8N-4)^2 prior and its rhs."""
    n = 8 * N + 4
    H = f64(np.array(HM, np.float64).reshape(n, n).copy()); b = f64(np.array(bM, np.float64).copy())
    lib().orc_ba_marginalize_frame(ptr(H, C.c_double), ptr(b, C.c_double), N, frame, ptr(f64(prior), C.c_double), ptr(f64(delta_prior), C.c_double))
    nd = n - 8
    return H.ravel()[:nd * nd].reshape(nd, nd).copy(), b[:nd].copy()


def m_energy(HM, bM, delta):
    n = len(delta)
    return lib().orc_ba_calc_m_energy(ptr(f64(HM), C.c_double), ptr(f64(bM), C.c_double), n, ptr(f64(delta), C.c_double))
