"""Build bundle-adjustment inputs from a synthetic window the way the reference's host code would
(DSOFrame state algebra, computeAdjoints, computeDelta), using the ORACLE's frame algebra.
Shared by the parity tests, smoke() and bench.py's cpu_baseline leg — checker side only.
"""
import ctypes as C

import numpy as np

from libcml_amd import abi, synth
from tests import oracle_lib as O


class BAInputs:
    pass


def make_scales():
    return O.OrcScales(**O.DEFAULT_SCALES)


def build_frames(W, scales):
    frames = (O.OrcFrame * W.N)()
    for k in range(W.N):
        f = frames[k]
        f.ab_exposure = float(W.ab_exposure[k])
        f.keyid = int(W.keyid[k])
        T = O.se3_from_Rt(W.R_eval[k], W.t_eval[k])
        a, b = W.aff_eval[k]
        O.lib().orc_frame_set_evalpt_scaled(C.byref(f), C.byref(T), C.c_double(a), C.c_double(b), C.byref(scales))
        st = O.f64(W.state[k])
        O.lib().orc_frame_set_state(C.byref(f), O.ptr(st, C.c_double), C.byref(scales))
    return frames


def frame_pairs(frames, N):
    pairs = np.zeros(N * N, abi.BA_PAIR_DTYPE)
    for h in range(N):
        for t in range(N):
            p = abi.BAPair()
            O.lib().orc_frame_precompute(C.byref(frames[h]), C.byref(frames[t]), C.byref(p))
            pairs[h * N + t] = np.frombuffer(bytes(p), abi.BA_PAIR_DTYPE)[0]
    return pairs


def adjoints_and_delta(frames, N, scales, optimize_a=1, optimize_b=1):
    adH = np.zeros(N * N * 64); adT = np.zeros(N * N * 64)
    O.lib().orc_ba_compute_adjoints(frames, N, C.byref(scales), O.ptr(adH, C.c_double), O.ptr(adT, C.c_double))
    adHTd = np.zeros(N * N * 8, np.float32)
    O.lib().orc_ba_compute_delta(frames, N, O.ptr(adH, C.c_double), O.ptr(adT, C.c_double), optimize_a, optimize_b,
                                 O.ptr(adHTd, C.c_float))
    prior = np.array([list(frames[k].prior) for k in range(N)]).ravel()
    dprior = np.array([list(frames[k].delta_prior) for k in range(N)]).ravel()
    return adH, adT, adHTd, prior, dprior


def make_inputs(config="small", seed=0xC0FFEE, shard=0, W=None, **kw):
    """Everything needed to drive either the oracle or the device BA path on one synthetic window (W: a window synth.make_window already built)."""
    if W is None:
        W = synth.make_window(config, seed=seed, shard=shard, **kw)
    I = BAInputs()
    I.W = W
    I.N, I.P = W.N, W.P
    I.grays, I.grads = [], []
    for k in range(W.N):
        g, d = O.build_pyramid(W.gray[k], W.levels)
        I.grays.append(g); I.grads.append(d)
    grads0 = [I.grads[k][0] for k in range(W.N)]
    colors, weights = synth.point_colors_weights(W, grads0)
    I.scales = make_scales()
    I.frames = build_frames(W, I.scales)
    I.prm = abi.default_ba_params(*W.K, W.w, W.h)
    pts = np.zeros(W.P, abi.BA_POINT_DTYPE)
    pts["x"] = W.pts["x"]; pts["y"] = W.pts["y"]; pts["idepth"] = W.pts["idepth"]
    pts["idepth_zero"] = W.pts["idepth"].astype(np.float32)
    pts["prior"] = 0.0
    pts["colors"] = colors; pts["weights"] = weights; pts["host"] = W.pts["host"]
    I.points = pts
    res = synth.residual_list(W, W.R_eval, W.t_eval)
    I.residuals = np.zeros(len(res), abi.BA_RESIDUAL_DTYPE)
    for f in ("point", "target", "state", "is_linearized"):
        I.residuals[f] = res[f]
    I.R = len(res)
    fr = np.zeros(W.N, abi.BA_FRAME_DTYPE)
    fr["image_id"] = np.arange(W.N) + 1000 * (shard + 1)
    fr["frame_energy_th"] = W.frame_energy_th
    for k in range(W.N):   # getB0 = (float)(state_zero[7] * scaleB), DSOFrame.h:197-199
        fr["b0"][k] = np.float32(I.frames[k].state_zero[7] * np.float32(I.scales.b))
    I.frames_dev = fr
    I.pairs = frame_pairs(I.frames, W.N)
    I.adH, I.adT, I.adHTd, I.prior, I.dprior = adjoints_and_delta(I.frames, W.N, I.scales)
    I.cdelta = np.zeros(4); I.cprior = np.full(4, 5e9)
    return I


def accum_in(I):
    a = abi.BAAccumIn()
    a.adHost = O.ptr(I.adH, C.c_double); a.adTarget = O.ptr(I.adT, C.c_double)
    a.adHTdeltaF = O.ptr(I.adHTd, C.c_float); a.cdelta = O.ptr(I.cdelta, C.c_double)
    a.prior = O.ptr(I.prior, C.c_double); a.delta_prior = O.ptr(I.dprior, C.c_double)
    a.cprior = O.ptr(I.cprior, C.c_double)
    return a


class OracleBA:
    """The oracle window plus numpy views of its state."""

    def __init__(self, I):
        self.I = I
        N = I.N
        self._imgs = [np.ascontiguousarray(I.grads[k][0]) for k in range(N)]
        arr = (C.POINTER(C.c_float) * N)(*[O.ptr(im, C.c_float) for im in self._imgs])
        self.w = O.lib().orc_ba_create(C.byref(I.prm), N, I.frames_dev.ctypes.data_as(C.POINTER(abi.BAFrame)), arr,
                                       I.P, I.points.ctypes.data_as(C.POINTER(abi.BAPoint)), I.R,
                                       I.residuals.ctypes.data_as(C.POINTER(abi.BAResidual)))
        self.set_pairs(I.pairs)

    def __del__(self):
        try:
            O.lib().orc_ba_destroy(self.w)
        except Exception:
            pass

    def set_pairs(self, pairs):
        self._pairs = np.ascontiguousarray(pairs)
        O.lib().orc_ba_set_pairs(self.w, self._pairs.ctypes.data_as(C.POINTER(abi.BAPair)))

    def view(self, name, n, dtype):
        p = getattr(self.w.contents, name)
        return np.ctypeslib.as_array(p, shape=(n,)) if n else np.zeros(0, dtype)

    def linearize(self):
        out = abi.BALinResult()
        O.lib().orc_ba_linearize_all(self.w, C.byref(out))
        return out

    def apply(self, copy=1):
        O.lib().orc_ba_apply(self.w, copy)

    def accumulate(self):
        n = 8 * self.I.N + 4
        self._ain = accum_in(self.I)
        HA = np.zeros((n, n)); bA = np.zeros(n); HL = np.zeros((n, n)); bL = np.zeros(n)
        Hsc = np.zeros((n, n)); bsc = np.zeros(n)
        d = C.c_double
        O.lib().orc_ba_accumulate(self.w, C.byref(self._ain), O.ptr(HA, d), O.ptr(bA, d), O.ptr(HL, d), O.ptr(bL, d),
                                  O.ptr(Hsc, d), O.ptr(bsc, d))
        return HA, bA, HL, bL, Hsc, bsc

    def solve(self, lam, HA, bA, HL, bL, Hsc, bsc, HM=None, bM=None, optcal=0):
        n = 8 * self.I.N + 4
        x = np.zeros(n)
        d = C.c_double
        rc = O.lib().orc_ba_solve(self.w, d(lam), O.ptr(HA, d), O.ptr(bA, d), O.ptr(HL, d), O.ptr(bL, d),
                                  O.ptr(HM, d) if HM is not None else None, O.ptr(bM, d) if bM is not None else None,
                                  O.ptr(Hsc, d), O.ptr(bsc, d), optcal, O.ptr(x, d))
        return x, rc

    def backsub(self, x):
        self._ain = accum_in(self.I)
        x = O.f64(x)
        rc = O.lib().orc_ba_backsub(self.w, C.byref(self._ain), O.ptr(x, C.c_double))
        return self.view("step", self.I.P, np.float64).copy(), rc

    # ---- marginalisation
    def relinearize_points(self, pts):
        self._ain = accum_in(self.I)
        pts = np.ascontiguousarray(pts, np.int32)
        return O.lib().orc_ba_relinearize_points(self.w, len(pts), O.ptr(pts, C.c_int), C.byref(self._ain))

    def marginalize_points(self, pts):
        n = 8 * self.I.N + 4
        self._ain = accum_in(self.I)
        pts = np.ascontiguousarray(pts, np.int32)
        M = np.zeros((n, n)); Mb = np.zeros(n); Msc = np.zeros((n, n)); Mbsc = np.zeros(n)
        d = C.c_double
        O.lib().orc_ba_marginalize_points(self.w, len(pts), O.ptr(pts, C.c_int), C.byref(self._ain), O.ptr(M, d), O.ptr(Mb, d),
                                          O.ptr(Msc, d), O.ptr(Mbsc, d))
        return M, Mb, Msc, Mbsc

    def l_energy(self):
        self._ain = accum_in(self.I)
        num = C.c_int()
        O.lib().orc_ba_calc_l_energy.restype = C.c_double
        return O.lib().orc_ba_calc_l_energy(self.w, C.byref(self._ain), C.byref(num)), num.value

    def rJ(self, which=0):
        return self.view("efsJ" if which else "rJ", self.I.R * 74, np.float32).reshape(-1, 74).copy()

    def states(self):
        R = self.I.R
        return dict(state=self.view("r_state", R, np.int32).copy(), new_state=self.view("r_new_state", R, np.int32).copy(),
                    energy=self.view("r_energy", R, np.float32).copy(),
                    new_energy=self.view("r_new_energy", R, np.float32).copy(),
                    new_energy_wo=self.view("r_new_energy_wo", R, np.float32).copy(),
                    good=self.view("r_good", R, np.uint8).copy())
