"""Coarse-initializer (DSOInitializer::calcResAndGS) test inputs from a synthetic window, and the ctypes wrapper of the
oracle restatement."""
import ctypes as C

import numpy as np

from libcml_amd import abi, synth
from tests import oracle_lib as O

STAR8 = synth.STAR8


def make_points(ref_grad, step=7, seed=3, bad_fraction=0.05, border=6):
    """What setFirst (DSOInitializer.cpp:40-100) stores per selected pixel at one level: the homogeneous pattern pixels, the
    reference gray at them, idepth 1 (perturbed here so that the idepth terms are exercised), outlierTH = 8 * 12^2."""
    h, w = ref_grad.shape[:2]
    rng = np.random.default_rng(seed)
    xs, ys = np.meshgrid(np.arange(border, w - border, step), np.arange(border, h - border, step))
    xs = xs.ravel().astype(np.float32); ys = ys.ravel().astype(np.float32)
    n = len(xs)
    pts = np.zeros(n, abi.INIT_POINT_DTYPE)
    for k, (dx, dy) in enumerate(STAR8):
        pts["p_pattern"][:, k, 0] = xs + dx; pts["p_pattern"][:, k, 1] = ys + dy; pts["p_pattern"][:, k, 2] = 1
        for i in range(n):
            pts["color"][i, k] = O.interpolate3(ref_grad, float(xs[i] + dx), float(ys[i] + dy))[0]
    pts["idepth_new"] = (1.0 + 0.2 * rng.standard_normal(n)).astype(np.float32)
    pts["iR"] = (1.0 + 0.05 * rng.standard_normal(n)).astype(np.float32)
    pts["outlier_th"] = np.float32(8 * 12.0 * 12.0)
    pts["energy"] = rng.uniform(0, 50, (n, 2)).astype(np.float32)
    pts["is_good"] = (rng.uniform(size=n) > bad_fraction).astype(np.int32)
    pts["jb"] = rng.standard_normal((n, 10)).astype(np.float32)          # stale rows: untouched for points that are not good
    return pts


def make_params(K, level, R, t, exposure_ratio, w2c_log3, alpha_w=150.0 * 150.0, alpha_k=2.5 * 2.5, coupling=1.0, huber=9.0):
    """The constants calcResAndGS forms at :455-480 from refToNew = (R, t) and K(level) (pinhole level scaling as
    InternalCalibration.h:116-127: fx/2^l, (cx+0.5)/2^l-0.5)."""
    fx, fy, cx, cy = K
    s = 2.0 ** level
    Kl = np.array([[fx / s, 0, (cx + 0.5) / s - 0.5], [0, fy / s, (cy + 0.5) / s - 0.5], [0, 0, 1.0]])
    RKi = (np.asarray(R, np.float64) @ np.linalg.inv(Kl)).astype(np.float32)
    P = abi.InitParams()
    P.RKi[:] = [float(v) for v in RKi.ravel()]
    P.t[:] = [float(np.float32(v)) for v in t]
    P.fx, P.fy, P.cx, P.cy = [float(np.float32(v)) for v in (Kl[0, 0], Kl[1, 1], Kl[0, 2], Kl[1, 2])]
    P.aff_a = float(np.float32(exposure_ratio)); P.aff_b = 0.0
    P.huber = huber; P.alpha_w = alpha_w; P.alpha_k = alpha_k; P.coupling_weight = coupling
    P.tlog[:] = [float(np.float32(v)) for v in w2c_log3]
    P.t_sqnorm = float(np.dot(t, t))
    return P


def oracle_calc(grad, prm, points):
    """orc_init_calc_res_and_gs on a copy of the points."""
    pts = np.ascontiguousarray(points.copy())
    img = np.ascontiguousarray(grad, np.float32)
    h, w = img.shape[:2]
    H = np.zeros((8, 8), np.float32); b = np.zeros(8, np.float32); Hsc = np.zeros((8, 8), np.float32); bsc = np.zeros(8, np.float32)
    res = np.zeros(3, np.float32)
    L = O.lib()
    L.orc_init_calc_res_and_gs.restype = None
    L.orc_init_calc_res_and_gs(O.ptr(img, C.c_float), w, h, C.byref(prm), len(pts), C.c_void_p(pts.ctypes.data),
                               O.ptr(H, C.c_float), O.ptr(b, C.c_float), O.ptr(Hsc, C.c_float), O.ptr(bsc, C.c_float), O.ptr(res, C.c_float))
    return pts, H, b, Hsc, bsc, res


def scene(level=1, config="small", trans_scale=1.0):
    """Reference = frame 0 of a synthetic window, tracked = frame 1, at pyramid level `level`."""
    W = synth.make_window(config, eval_noise=0.0, idepth_noise=0.0, state_noise=0.0)
    g0 = O.build_pyramid(W.gray[0], level + 1)[1][level]
    g1 = O.build_pyramid(W.gray[1], level + 1)[1][level]
    R = W.R_eval[1] @ W.R_eval[0].T
    # the initializer works in the scale where the scene sits at inverse depth ~1 (setFirst: idepth = 1)
    t = (W.t_eval[1] - R @ W.t_eval[0]) * float(np.median(W.pts["idepth_true"])) * trans_scale
    xi = O.se3_log(O.se3_from_Rt(W.R_eval[1], W.t_eval[1]))
    ratio = W.ab_exposure[1] / W.ab_exposure[0]
    return W, g0, g1, R, t, ratio, xi[:3]
