"""Timing of the per-frame tracker evaluation (computeResidual + computeHessian, TR.cpp:248-492) on a synthetic scene."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libcml_amd import abi, device, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "B"
W = synth.make_window(cfg)
fx, fy, cx, cy = W.K
ctx = device.Ctx(max_frames=8)
ref, new = W.N - 2, W.N - 1
L = 4
ctx.pyramid_build(1, W.gray[ref], L); ctx.pyramid_build(2, W.gray[new], L)
pts = []
for i in range(W.P):
    h = int(W.pts["host"][i]); x, y, idp = float(W.pts["x"][i]), float(W.pts["y"][i]), float(W.pts["idepth"][i])
    Rht = W.R_eval[ref] @ W.R_eval[h].T; tht = W.t_eval[ref] - Rht @ W.t_eval[h]
    p = Rht @ np.array([(x - cx) / fx, (y - cy) / fy, 1.0]) + tht * idp
    pts.append(((p[0] / p[2]) * fx + cx, (p[1] / p[2]) * fy + cy, idp / p[2], 1.0))
import time as _t
_g = np.ascontiguousarray(W.gray[new])
for _ in range(3): ctx.pyramid_build(777, _g, L); ctx.pyramid_drop(777)
_t0 = _t.perf_counter()
for _ in range(10): ctx.pyramid_build(777, _g, L); ctx.pyramid_drop(777)
ctx.sync()
print("pyramid_build + pyramid_drop of a %dx%d frame, %d levels: %.0f us per frame (1.9 MB gray upload + reduce + gradient; levels come from the pool)" % (W.w, W.h, L, (_t.perf_counter() - _t0) / 10 * 1e6))
_pts = np.array(pts)
nout = ctx.tracker_make_coarse_depth(1, L, _pts)
_t0 = _t.perf_counter()
for _ in range(10): nout = ctx.tracker_make_coarse_depth(1, L, _pts)
print("makeCoarseDepthL0 on the device: %d points -> lists %s, %.0f us per synchronous call (ordered splat, %d levels)" % (len(pts), nout, (_t.perf_counter() - _t0) / 10 * 1e6, L))
Rrn = W.R_true[new] @ W.R_true[ref].T; trn = W.t_true[new] - Rrn @ W.t_true[ref]
prm = abi.default_tracker_params()
for lvl in range(L):
    d = float(1 << lvl)
    K = np.array([fx / d, fy / d, (cx + 0.5) / d - 0.5, (cy + 0.5) / d - 0.5])
    for _ in range(20): ctx.tracker_eval(2, lvl, Rrn, trn, K, np.array([1.0, 0.0]), 0.0, prm, 1)
    t0 = time.perf_counter()
    n = 200
    for _ in range(n): r = ctx.tracker_eval(2, lvl, Rrn, trn, K, np.array([1.0, 0.0]), 0.0, prm, 1)
    dt = (time.perf_counter() - t0) / n
    print("level %d: n=%d  %.1f us per synchronous eval (launch + 1 D2H of the 9x9 system)  numTerms=%d" % (lvl, nout[lvl], dt * 1e6, (r[0] if isinstance(r, tuple) else r).numTermsInE))
