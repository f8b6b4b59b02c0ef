"""Immature-point (DSOTracer) test inputs from a synthetic window, and ctypes wrappers of the oracle restatement."""
import ctypes as C

import numpy as np

from libcml_amd import abi, synth
from tests import oracle_lib as O

STAR8 = synth.STAR8


def make_immature(W, grads0):
    """What makeNewTraces (DSOTracer.cpp:496-541) stores per point: gradH from the interpolated gradients at the pattern
    pixels, energyTH = 8 * outlierTH, and the MapPoint patches (integer-pixel lookups, MapObject.h:392-412)."""
    n = W.P
    pts = np.zeros(n, abi.IMMATURE_POINT_DTYPE)
    pts["x"] = W.pts["x"]; pts["y"] = W.pts["y"]; pts["host"] = W.pts["host"]
    pts["last_status"] = abi.IPS_UNINITIALIZED
    pts["idepth_min"] = 1.0 / 1000.0
    pts["idepth_max"] = np.nan
    pts["energy_th"] = 8 * float(np.float32(12.0 * 12.0))
    pts["quality"] = 10000
    pts["last_uv"] = -1; pts["last_pixel_interval"] = -1
    for i in range(n):
        g = grads0[int(pts["host"][i])]
        x, y = float(pts["x"][i]), float(pts["y"][i])
        G = np.zeros((2, 2))
        for k, (dx, dy) in enumerate(STAR8):
            v = O.interpolate3(g, x + dx, y + dy)
            grad = v[1:3].astype(np.float64)
            G += np.outer(grad, grad)
            px = g[int(y) + dy, int(x) + dx]
            pts["gray"][i, k] = px[0]
            pts["dpatch"][i, 3 * k:3 * k + 3] = px
        pts["gradH"][i] = G.ravel()
    return pts


def trace_pairs(W, f):
    """host -> frame f: K R K^-1, K t (DSOTracer.cpp:608-610) and Exposure::to (Exposure.h:119-123)."""
    fx, fy, cx, cy = W.K
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    Ki = np.linalg.inv(K)
    pr = np.zeros(W.N, abi.TRACE_PAIR_DTYPE)
    for h in range(W.N):
        R = W.R_eval[f] @ W.R_eval[h].T
        t = W.t_eval[f] - R @ W.t_eval[h]
        pr["KRKi"][h] = ((K @ R) @ Ki).ravel()
        pr["Kt"][h] = K @ t
        a_h, b_h = W.aff_eval[h]; a_f, b_f = W.aff_eval[f]
        a = np.exp(a_f - a_h) * W.ab_exposure[f] / W.ab_exposure[h]
        pr["aff_a"][h] = a; pr["aff_b"][h] = b_f - a * b_h
    return pr


def activation_pairs(W):
    N = W.N
    pr = np.zeros(N * N, abi.ACTIVATION_PAIR_DTYPE)
    for h in range(N):
        for t in range(N):
            R = W.R_eval[t] @ W.R_eval[h].T
            tt = W.t_eval[t] - R @ W.t_eval[h]
            a_h, b_h = W.aff_eval[h]; a_t, b_t = W.aff_eval[t]
            a = np.exp(a_t - a_h) * W.ab_exposure[t] / W.ab_exposure[h]
            pr["R"][h * N + t] = R.ravel(); pr["t"][h * N + t] = tt
            pr["aff_a"][h * N + t] = a; pr["aff_b"][h * N + t] = b_t - a * b_h
    return pr


def oracle_trace(grad, pairs, prm, points):
    """orc_trace_point over the points (in place on a copy)."""
    out = points.copy()
    img = np.ascontiguousarray(grad, np.float32)
    h, w = img.shape[:2]
    pr = np.ascontiguousarray(pairs)
    L = O.lib()
    for i in range(len(out)):
        L.orc_trace_point(O.ptr(img, C.c_float), w, h, C.c_void_p(pr.ctypes.data + int(out["host"][i]) * pr.itemsize), C.byref(prm),
                          C.c_void_p(out.ctypes.data + i * out.itemsize))
    return out


def oracle_optimize(grads0, K, pairs, prm, min_obs, points):
    N = len(grads0)
    imgs = [np.ascontiguousarray(g, np.float32) for g in grads0]
    h, w = imgs[0].shape[:2]
    arr = (C.POINTER(C.c_float) * N)(*[O.ptr(im, C.c_float) for im in imgs])
    pr = np.ascontiguousarray(pairs); pts = np.ascontiguousarray(points)
    Kd = np.ascontiguousarray(K, np.float64)
    n = len(pts)
    res = np.zeros(n, np.int32); idp = np.zeros(n, np.float32); st = np.zeros((n, N), np.int32)
    L = O.lib()
    for i in range(n):
        v = C.c_float(0)
        res[i] = L.orc_optimize_immature_point(N, arr, w, h, O.ptr(Kd, C.c_double), C.c_void_p(pr.ctypes.data), C.byref(prm), int(min_obs),
                                               C.c_void_p(pts.ctypes.data + i * pts.itemsize), C.byref(v), O.ptr(st[i], C.c_int))
        idp[i] = v.value if res[i] == 1 else 0.0
    return res, idp, st
