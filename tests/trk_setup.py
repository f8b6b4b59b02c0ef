"""Tracker / coarse-depth / reprojection test inputs from a synthetic window (checker side)."""
import ctypes as C

import numpy as np

from libcml_amd import abi, synth
from tests import oracle_lib as O


class Scene:
    pass


def make_scene(config="small", seed=0xC0FFEE, **kw):
    W = synth.make_window(config, seed=seed, **kw)
    s = Scene()
    s.W = W
    s.levels = min(len(O.pyramid_sizes(W.w, W.h)[0]), 5)
    s.ref = W.N - 2          # last keyframe = tracking reference
    s.new = W.N - 1          # frame to track
    s.grays, s.grads = {}, {}
    for k in (s.ref, s.new):
        s.grays[k], s.grads[k] = O.build_pyramid(W.gray[k], s.levels)
    # host part of makeCoarseDepthL0 (TR.cpp:521-540): project every active point into the reference frame
    fx, fy, cx, cy = W.K
    pts = []
    rng = np.random.default_rng(seed + 5)
    for i in range(W.P):
        h = int(W.pts["host"][i])
        x, y, idp = float(W.pts["x"][i]), float(W.pts["y"][i]), float(W.pts["idepth"][i])
        Rht = W.R_eval[s.ref] @ W.R_eval[h].T
        tht = W.t_eval[s.ref] - Rht @ W.t_eval[h]
        p = Rht @ np.array([(x - cx) * (1.0 / fx), (y - cy) * (1.0 / fy), 1.0]) + tht * idp
        Ku = (p[0] / p[2]) * fx + cx; Kv = (p[1] / p[2]) * fy + cy
        new_id = (1.0 / p[2]) * idp
        unc = rng.uniform(0.5, 2.0) * 1e-3
        weight = np.float32(np.sqrt(np.float32(1e-3 / (unc + 1e-12))))
        pts.append((Ku, Kv, new_id, float(weight)))
    s.cd_pts = np.array(pts, np.float64)
    return s


def oracle_coarse_depth(s):
    L = s.levels
    ws = (C.c_int * L)(*[s.grays[s.ref][l].shape[1] for l in range(L)])
    hs = (C.c_int * L)(*[s.grays[s.ref][l].shape[0] for l in range(L)])
    gl = [np.ascontiguousarray(s.grays[s.ref][l]) for l in range(L)]
    gp = (C.POINTER(C.c_float) * L)(*[O.ptr(g, C.c_float) for g in gl])
    lists = [np.zeros((gl[l].size, 4), np.float32) for l in range(L)]
    lp = (C.POINTER(C.c_float) * L)(*[O.ptr(a, C.c_float) for a in lists])
    nout = (C.c_int * L)()
    pts = np.ascontiguousarray(s.cd_pts)
    O.lib().orc_tracker_make_coarse_depth(O.ptr(pts, C.c_double), len(pts), L, ws, hs, gp, lp, nout)
    return lists, list(nout[:L])


def tracker_inputs(s, level):
    W = s.W
    fx, fy, cx, cy = W.K
    d = float(1 << level)
    K = np.array([fx / d, fy / d, (cx + 0.5) / d - 0.5, (cy + 0.5) / d - 0.5])     # InternalCalibration.h:116-127
    # refToNew with a small error so that the Jacobian is non-trivial
    Rrn = W.R_true[s.new] @ W.R_true[s.ref].T
    trn = W.t_true[s.new] - Rrn @ W.t_true[s.ref]
    Rrn = synth.so3_exp(np.array([0.002, -0.001, 0.0015])) @ Rrn
    trn = trn + np.array([0.01, -0.005, 0.008])
    a_r, b_r = W.aff_true[s.ref]; a_n, b_n = W.aff_true[s.new]
    a = np.exp(a_n - a_r); b = b_n - a * b_r                                        # Exposure::to
    return Rrn, trn, K, np.array([a, b]), float(b_r)


def oracle_tracker_eval(s, level, uvic, R, t, K, aff, b0, prm):
    img = np.ascontiguousarray(s.grads[s.new][level])
    out = abi.TrackerResult()
    cap = len(uvic) + 4
    warped = np.zeros((8, cap), np.float32)
    uv = np.ascontiguousarray(uvic, np.float32)
    Rr = np.ascontiguousarray(R, np.float64).ravel()
    O.lib().orc_tracker_eval(O.ptr(img, C.c_float), img.shape[1], img.shape[0], O.ptr(uv, C.c_float), len(uv), level,
                             O.ptr(Rr, C.c_double), O.ptr(O.f64(t), C.c_double), O.ptr(O.f64(K), C.c_double),
                             O.ptr(O.f64(aff), C.c_double), C.c_double(b0), C.byref(prm), 1, C.byref(out),
                             O.ptr(warped, C.c_float), cap)
    return out, warped


def reproj_inputs(s, n_obs=1000, n_pts=300, seed=3):
    W = s.W
    poses, pts, obs = synth.indirect_observations(W, n_obs=n_obs, n_pts=n_pts, seed=seed)
    o = np.zeros(n_obs, abi.REPROJ_OBS_DTYPE)
    for f in ("frame", "point", "gx", "gy"):
        o[f] = obs[f]
    return poses, pts, o, W.K[0], W.K[1]


def oracle_reproj(poses, points, obs, fx, fy):
    N, M, n = len(poses), len(points), len(obs)
    M6 = np.zeros((6 * N, 6 * N)); b6 = np.zeros(6 * N); Jp = np.zeros((M, 3)); used = np.zeros(n, np.uint8)
    p = np.ascontiguousarray(poses); q = np.ascontiguousarray(points); o = np.ascontiguousarray(obs)
    O.lib().orc_reproj_accumulate(N, O.ptr(p, C.c_double), M, O.ptr(q, C.c_double), n, o.ctypes.data_as(C.POINTER(abi.ReprojObs)),
                                  C.c_double(fx), C.c_double(fy), O.ptr(M6, C.c_double), O.ptr(b6, C.c_double),
                                  O.ptr(Jp, C.c_double), O.ptr(used, C.c_ubyte))
    return M6, b6, Jp, used
