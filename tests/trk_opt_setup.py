"""Inputs of DSOTracker::optimize / trackWithMotionModel from a synthetic scene, for the oracle and for the host mirror (checker side)."""
import ctypes as C

import numpy as np

from libcml_amd import abi, synth
from tests import oracle_lib as O
from tests import trk_setup as T


class Problem:
    pass


def make_problem(config="small", **kw):
    s = T.make_scene(config, eval_noise=0.0, idepth_noise=0.0, state_noise=0.0, **kw)   # exact geometry: the optimum is the true motion
    W = s.W
    P = Problem()
    P.s, P.W = s, W
    P.levels = s.levels
    P.lists, P.n = T.oracle_coarse_depth(s)
    P.uvic = [np.ascontiguousarray(P.lists[l][:P.n[l]], np.float32) for l in range(P.levels)]
    P.imgs = [np.ascontiguousarray(s.grads[s.new][l], np.float32) for l in range(P.levels)]
    P.Rt = W.R_true[s.new] @ W.R_true[s.ref].T
    P.tt = W.t_true[s.new] - P.Rt @ W.t_true[s.ref]
    a_r, b_r = W.aff_true[s.ref]
    P.ref_exp = [a_r, b_r, float(W.ab_exposure[s.ref])]
    P.init_exp = [a_r, b_r, float(W.ab_exposure[s.new])]       # the new frame starts from the reference's parameters
    P.prm = abi.default_tracker_params()
    return P


def orc_problem(P, optimize_a=1, optimize_b=1, have_last=0, last_rmse=None):
    q = O.OrcTrkProblem()
    q.levels = P.levels
    for l in range(P.levels):
        q.aos3[l] = O.ptr(P.imgs[l], C.c_float); q.w[l] = P.imgs[l].shape[1]; q.h[l] = P.imgs[l].shape[0]
        q.uvic[l] = O.ptr(P.uvic[l], C.c_float); q.n[l] = len(P.uvic[l])
    for k in range(4):
        q.K[k] = P.W.K[k]
    q.ref_a, q.ref_b, q.ref_t = P.ref_exp
    q.new_t = P.init_exp[2]
    q.prm = P.prm
    q.optimize_a = optimize_a; q.optimize_b = optimize_b; q.saturated_ratio_th = 0.33
    q.have_last = have_last
    if last_rmse is not None:
        for l in range(len(last_rmse)):
            q.last_rmse[l] = last_rmse[l]
    return q


def perturbed(P, w, dt):
    return synth.so3_exp(np.asarray(w, float)) @ P.Rt, P.tt + np.asarray(dt, float)


def oracle_optimize(P, R0, t0, q=None, log_cap=512):
    q = q or orc_problem(P)
    T_ = O.se3_from_Rt(R0, t0)
    a, b = C.c_double(P.init_exp[0]), C.c_double(P.init_exp[1])
    out = O.OrcTrkResult()
    log = (O.OrcTrkStep * log_cap)()
    O.lib().orc_tracker_optimize(C.byref(q), C.byref(T_), C.byref(a), C.byref(b), C.byref(out), log, log_cap)
    R = np.zeros(9)
    O.lib().orc_se3_matrix(C.byref(T_), O.ptr(R, C.c_double))
    n = min(out.n_steps, log_cap)
    steps = [(log[i].level, log[i].iteration, log[i].accept, log[i].lambda_) for i in range(n)]
    return dict(R=R.reshape(3, 3), t=np.array(T_.t[:]), a=a.value, b=b.value, out=out, steps=steps)


def oracle_eval_fn(P):
    """computeResidual + computeHessian of the oracle with the mirror's EvalFn signature."""
    def fn(level, R, t, K, aff, b0, prm, out):
        img = P.imgs[level]; uv = P.uvic[level]
        O.lib().orc_tracker_eval(O.ptr(img, C.c_float), img.shape[1], img.shape[0], O.ptr(uv, C.c_float), len(uv), level,
                                 O.ptr(O.f64(R), C.c_double), O.ptr(O.f64(t), C.c_double), O.ptr(O.f64(K), C.c_double),
                                 O.ptr(O.f64(aff), C.c_double), C.c_double(b0), C.byref(prm), 1, out, None, 0)
        return 0
    return fn


def oracle_track(P, hyps, last_coarse_rmse=100.0, failure_mode=0):
    q = orc_problem(P)
    H = (O.OrcSE3 * len(hyps))(*[O.se3_from_Rt(R, t) for R, t in hyps])
    best = O.OrcSE3(); ba, bb = C.c_double(), C.c_double(); res = O.OrcTrkResult(); ach = C.c_double(); win, tries = C.c_int(), C.c_int()
    ok = O.lib().orc_tracker_track_with_motion_model(C.byref(q), len(hyps), H, C.c_double(P.init_exp[0]), C.c_double(P.init_exp[1]),
                                                     C.c_double(last_coarse_rmse), failure_mode, C.byref(best), C.byref(ba), C.byref(bb),
                                                     C.byref(res), C.byref(ach), C.byref(win), C.byref(tries))
    R = np.zeros(9)
    O.lib().orc_se3_matrix(C.byref(best), O.ptr(R, C.c_double))
    return dict(ok=bool(ok), R=R.reshape(3, 3), t=np.array(best.t[:]), a=ba.value, b=bb.value, res=res, achieved=ach.value,
                winner=win.value, tries=tries.value)
