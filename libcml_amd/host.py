"""ctypes access to the C++ host mirror (libcmlhost.so: cml_amd::DSOBundleAdjustment / DSOTracker over the C ABI).
Product-side plumbing for tests and bench.py; a C++ caller would use the classes in libcml_amd/host/*.h directly."""
import ctypes as C
import time
import os

import numpy as np

from . import abi, device, synth

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcmlhost.so")
_lib = None
_d, _f, _i, _u8, _vp = C.c_double, C.c_float, C.c_int, C.c_ubyte, C.c_void_p
_P = C.POINTER


def lib():
    global _lib
    if _lib is None:
        device.lib()     # libcmlhip.so first (rpath $ORIGIN also covers it)
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libcml_amd/libcmlhost.so is missing: run `python -m libcml_amd.build`")
        L = C.CDLL(LIB_PATH)
        L.cmlhost_ba_create.restype = _vp; L.cmlhost_ba_create.argtypes = [_vp]
        L.cmlhost_ba_destroy.argtypes = [_vp]
        L.cmlhost_ba_set_calibration.argtypes = [_vp, _d, _d, _d, _d, _i, _i]
        L.cmlhost_ba_set_param.argtypes = [_vp, C.c_char_p, _d]
        L.cmlhost_ba_add_frame.argtypes = [_vp, C.c_uint64, _P(_d), _P(_d), _d, _d, _d]
        L.cmlhost_ba_set_frame_state.argtypes = [_vp, _i, _P(_d)]
        L.cmlhost_ba_set_frame_energy_th.argtypes = [_vp, _i, _d]
        L.cmlhost_ba_add_point.argtypes = [_vp, _f, _f, _d, _i, _P(_f), _P(_f), _i]
        L.cmlhost_ba_run.argtypes = [_vp, _i]
        L.cmlhost_ba_run_resident.argtypes = [_vp, _i]
        L.cmlhost_ba_run_host_loop.argtypes = [_vp, _i]
        L.cmlhost_ba_begin_resident.argtypes = [_vp, _i]
        L.cmlhost_ba_iterate_resident.argtypes = [_vp, _i, _d]
        L.cmlhost_ba_end_resident.argtypes = [_vp, _P(_d)]
        L.cmlhost_ba_flag_frame.argtypes = [_vp, _i, _i]
        L.cmlhost_ba_flag_frames_for_marginalization.argtypes = [_vp, _i]
        L.cmlhost_ba_try_marginalize.argtypes = [_vp]
        L.cmlhost_ba_flag_frames_for_marginalization_v.argtypes = [_vp, _i, _P(_i)]
        L.cmlhost_ba_export.argtypes = [_vp, _vp, _vp, _vp]
        L.cmlhost_ba_marginalize_points.argtypes = [_vp]
        L.cmlhost_ba_marginalize_frames.argtypes = [_vp, _P(_i), _i]
        L.cmlhost_ba_get_prior.argtypes = [_vp, _P(_d), _P(_d)]
        L.cmlhost_ba_get_point_flags.argtypes = [_vp, _P(_u8), _P(_u8), _P(_f)]
        L.cmlhost_ba_rejected.argtypes = [_vp]
        L.cmlhost_ba_run_timing.argtypes = [_vp, _P(_d)]
        L.cmlhost_ba_coarse_depth_points.argtypes = [_vp, _i, _P(_d), _P(_d), _i]
        L.cmlhost_ba_last_lambda.restype = _d; L.cmlhost_ba_last_lambda.argtypes = [_vp]
        L.cmlhost_ba_calc_m_energy.restype = _d; L.cmlhost_ba_calc_m_energy.argtypes = [_vp]
        L.cmlhost_ba_calc_l_energy.restype = _d; L.cmlhost_ba_calc_l_energy.argtypes = [_vp]
        L.cmlhost_ba_last_error.restype = C.c_char_p; L.cmlhost_ba_last_error.argtypes = [_vp]
        L.cmlhost_ba_counts.argtypes = [_vp] + [_P(_i)] * 5
        L.cmlhost_ba_get_frame.argtypes = [_vp, _i, _P(_d), _P(_d), _P(_d), _P(_d), _P(_d)]
        L.cmlhost_ba_get_points.argtypes = [_vp, _P(_d), _P(_u8), _P(_i)]
        L.cmlhost_ba_get_residual_states.argtypes = [_vp, _P(_i), _P(_u8), _P(_u8)]
        L.cmlhost_ba_get_outliers.argtypes = [_vp, _P(_i)]
        L.cmlhost_ba_get_algebra.argtypes = [_vp, _P(_d), _P(_d), _P(_f), _P(abi.BAPair), _P(_d), _P(_d), _P(_d)]
        L.cmlhost_ba_orthogonalize.argtypes = [_vp, _P(_d), _i]
        L.cmlhost_ba_set_indirect_points.argtypes = [_vp, _i, _P(_d), _i, _P(abi.ReprojObs)]
        L.cmlhost_ba_get_indirect.argtypes = [_vp, _P(_d), _P(_d), _P(_d)]
        L.cmlhost_ba_stats.argtypes = [_vp, _P(_d), _i]
        L.cmlhost_tracker_create.restype = _vp; L.cmlhost_tracker_create.argtypes = [_vp]
        L.cmlhost_tracker_destroy.argtypes = [_vp]
        L.cmlhost_tracker_set_calibration.argtypes = [_vp, _d, _d, _d, _d]
        L.cmlhost_tracker_set_param.argtypes = [_vp, C.c_char_p, _d]
        L.cmlhost_tracker_make_coarse_depth.argtypes = [_vp, C.c_uint64, _i, _P(_d), _i, _P(_i)]
        L.cmlhost_tracker_optimize.argtypes = [_vp, C.c_uint64, _i, _P(_d), _P(_d), _P(_d), _P(_d), _P(_d), _P(_i), _P(_i), _P(_d),
                                               _P(_d), _P(_d), _P(_i), _P(_i), _P(_i)]
        L.cmlhost_tracker_last_error.restype = C.c_char_p; L.cmlhost_tracker_last_error.argtypes = [_vp]
        L.cmlhost_tracker_set_eval.argtypes = [_vp, _vp, _vp]
        L.cmlhost_tracker_steps.argtypes = [_vp, _i, _P(_i), _P(_i), _P(_i), _P(_d)]
        L.cmlhost_tracker_set_last_residual.argtypes = [_vp, _i, _i, _P(_d)]
        L.cmlhost_tracker_track_with_motion_model.argtypes = [_vp, C.c_uint64, _i, _i, _P(_d), _P(_d), _P(_d), _P(_d), _P(_d), _P(_d), _P(_d), _P(_i), _P(_i),
                                                              _P(_i), _P(_i), _P(_i), _P(_i), _P(_d), _i]
        L.cmlhost_frame_track_and_trace.argtypes = [_vp, _vp, C.c_uint64, _i, _i, _P(_d), _P(_d), _P(_d), _i, _i, _P(_i), _P(_d), _i, _P(_d),
                                                    _P(_d), _P(_d), _P(_d), _P(_d), _P(_i), _P(_i), _P(_i), _P(_i), _P(_i), _P(_i), _P(_d), _P(_i), _P(_i), _vp]
        L.cmlhost_tracer_create.restype = _vp; L.cmlhost_tracer_create.argtypes = [_vp]
        L.cmlhost_tracer_destroy.argtypes = [_vp]
        L.cmlhost_tracer_add_point.argtypes = [_vp, _f, _f, _i, _P(_f), _P(_f), _P(_d), _f]
        L.cmlhost_tracer_add_points.argtypes = [_vp, _i, _P(_f), _i, _P(_f), _P(_f), _P(_d)]
        L.cmlhost_tracer_compact.argtypes = [_vp]
        L.cmlhost_tracer_prepare_resident.argtypes = [_vp, _i, _P(_i)]
        L.cmlhost_tracer_get_frame_ids.argtypes = [_vp, _P(_i)]
        L.cmlhost_tracer_immature_counts.argtypes = [_vp, _i, _P(_i), _P(_i)]
        L.cmlhost_tracer_add_activated_to_ba.argtypes = [_vp, _vp, _i, _P(_i), _i, _P(_i), _P(_i)]
        L.cmlhost_ba_add_points.argtypes = [_vp, _i, _P(_f), _P(_d), _P(_i), _P(_f), _P(_f), _i]
        L.cmlhost_tracer_trace.argtypes = [_vp, C.c_uint64, _i, _i, _P(_i), _vp, _P(_i)]
        L.cmlhost_tracer_activate.argtypes = [_vp, _i, _P(_i), _P(C.c_uint64), _P(_d), _i, _i, _vp, _P(_i), _i]
        L.cmlhost_tracer_count.argtypes = [_vp]
        L.cmlhost_tracer_get_points.argtypes = [_vp, _vp, _P(_u8), _P(_u8), _P(_f)]
        L.cmlhost_tracer_last_error.restype = C.c_char_p; L.cmlhost_tracer_last_error.argtypes = [_vp]
        L.cmlhost_pnp_optimize.argtypes = [_vp, _i, _P(_d), _P(_d), _vp, _vp, _P(_d), _i, _vp, _P(_u8), _i, _i, _P(_i), _P(_d), _P(_d), _P(_d)]
        L.cmlhost_pnp_optimize_points.argtypes = [_vp, _i, _P(_d), _P(_d), _P(_d), _i, _vp, _P(_i), _P(_i), _i, _P(_i), _P(_d), _P(_d), _P(_d)]
        L.cmlhost_lba_create.restype = _vp; L.cmlhost_lba_create.argtypes = [_vp]
        L.cmlhost_lba_destroy.argtypes = [_vp]
        L.cmlhost_lba_set_params.argtypes = [_vp, _i, _i, _i]
        L.cmlhost_lba_local_optimize.argtypes = [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp]
        L.cmlhost_lba_apply.argtypes = [_vp, _i, _vp, _i, _P(_d), _P(_i), _i, _P(abi.LbaResult)]
        L.cmlhost_lba_last_error.restype = C.c_char_p; L.cmlhost_lba_last_error.argtypes = [_vp]
        L.cmlhost_init_create.restype = _vp; L.cmlhost_init_create.argtypes = [_vp]
        L.cmlhost_init_destroy.argtypes = [_vp]
        L.cmlhost_init_set_first.argtypes = [_vp, _i, _P(_i), _P(_i), _P(_d), _P(_vp), _P(_i), _P(_i), _P(_i), _P(_d), _d]
        L.cmlhost_init_try.argtypes = [_vp, C.c_uint64, _P(_d), _d]
        L.cmlhost_init_state.argtypes = [_vp, _P(_d), _P(_i), _P(_i), _P(_i), _P(_f)]
        L.cmlhost_init_level_size.argtypes = [_vp, _i]
        L.cmlhost_init_get_points.argtypes = [_vp, _i, _P(_f), _P(_f), _P(_f), _P(_u8), _P(_f), _P(_i), _P(_i)]
        L.cmlhost_init_last_error.restype = C.c_char_p; L.cmlhost_init_last_error.argtypes = [_vp]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(_P(t))


class HostBA:
    """cml_amd::DSOBundleAdjustment (flat-window mirror of the reference class)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.L = lib()
        self.h = self.L.cmlhost_ba_create(ctx.h)

    def close(self):
        if self.h:
            self.L.cmlhost_ba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_calibration(self, fx, fy, cx, cy, w, h):
        self.L.cmlhost_ba_set_calibration(self.h, fx, fy, cx, cy, w, h)

    def set_param(self, name, value):
        if self.L.cmlhost_ba_set_param(self.h, name.encode(), float(value)) != 0:
            raise KeyError(name)

    def add_frame(self, image_id, R, t, a, b, exposure=1.0):
        R = np.ascontiguousarray(R, np.float64).ravel(); t = np.ascontiguousarray(t, np.float64)
        return self.L.cmlhost_ba_add_frame(self.h, int(image_id), _p(R, _d), _p(t, _d), a, b, exposure)

    def set_frame_state(self, f, state):
        s = np.ascontiguousarray(state, np.float64)
        self.L.cmlhost_ba_set_frame_state(self.h, f, _p(s, _d))

    def add_point(self, x, y, idepth, host, colors, weights, prior=False):
        c = np.ascontiguousarray(colors, np.float32); w = np.ascontiguousarray(weights, np.float32)
        return self.L.cmlhost_ba_add_point(self.h, float(x), float(y), float(idepth), int(host), _p(c, _f), _p(w, _f), int(prior))

    def add_points(self, xy, idepth, host, colors, weights, prior=False):
        """addPoints for n points in one call: xy (n, 2), idepth (n), host (n), colors / weights (n, 8)"""
        xy = np.ascontiguousarray(xy, np.float32); idp = np.ascontiguousarray(idepth, np.float64); hs = np.ascontiguousarray(host, np.int32)
        c = np.ascontiguousarray(colors, np.float32); w = np.ascontiguousarray(weights, np.float32)
        return self.L.cmlhost_ba_add_points(self.h, len(xy), _p(xy, _f), _p(idp, _d), _p(hs, _i), _p(c, _f), _p(w, _f), int(prior))

    def run(self, update_points_only=False):
        return bool(self.L.cmlhost_ba_run(self.h, int(update_points_only)))

    def run_host_loop(self, update_points_only=False):
        """The literal loop of BA::run: one synchronous device call per reference statement."""
        return bool(self.L.cmlhost_ba_run_host_loop(self.h, int(update_points_only)))

    def run_resident(self, update_points_only=False):
        """run() with the iteration loop resident on the device (forceAccept + fixLambda, no early break)."""
        return bool(self.L.cmlhost_ba_run_resident(self.h, int(update_points_only)))

    def begin_resident(self, update_points_only=False):
        return bool(self.L.cmlhost_ba_begin_resident(self.h, int(update_points_only)))

    def iterate_resident(self, k, lam):
        return bool(self.L.cmlhost_ba_iterate_resident(self.h, int(k), float(lam)))

    def end_resident(self):
        e = _d()
        ok = bool(self.L.cmlhost_ba_end_resident(self.h, C.byref(e)))
        return ok, e.value

    # ---- marginalisation (reference BA.h:34-46)
    def flag_frame(self, f, flag=True):
        self.L.cmlhost_ba_flag_frame(self.h, int(f), int(flag))

    def flag_frames_for_marginalization(self, immature=0):
        self.L.cmlhost_ba_flag_frames_for_marginalization(self.h, int(immature))

    def flag_frames_for_marginalization_v(self, immature_per_frame):
        """flagFramesForMarginalization with every frame's own immature count (BA.cpp:617); call BEFORE add_frame, as addNewFrame does (:428)."""
        a = np.ascontiguousarray(immature_per_frame, np.int32)
        self.L.cmlhost_ba_flag_frames_for_marginalization_v(self.h, len(a), _p(a, _i))

    def export(self):
        """(frames, points, residuals) of the mirror as structured arrays (HOST_BA_*_DTYPE): everything an independent checker needs to
        rebuild the window the mirror holds."""
        c = self.counts()
        fr = np.zeros(c["frames"], HOST_BA_FRAME_DTYPE); pt = np.zeros(c["points"], HOST_BA_POINT_DTYPE); rs = np.zeros(c["residuals"], HOST_BA_RESIDUAL_DTYPE)
        sz = self.L.cmlhost_ba_export(self.h, fr.ctypes.data, pt.ctypes.data, rs.ctypes.data)
        assert (sz & 1023, (sz >> 10) & 1023, sz >> 20) == (HOST_BA_FRAME_DTYPE.itemsize, HOST_BA_POINT_DTYPE.itemsize, HOST_BA_RESIDUAL_DTYPE.itemsize), sz
        return fr, pt, rs

    def try_marginalize(self):
        return bool(self.L.cmlhost_ba_try_marginalize(self.h))

    def marginalize_points(self):
        return bool(self.L.cmlhost_ba_marginalize_points(self.h))

    def marginalize_frames(self):
        out = np.zeros(64, np.int32)
        n = self.L.cmlhost_ba_marginalize_frames(self.h, _p(out, _i), 64)
        return out[:n].copy()

    def prior(self):
        n = self.L.cmlhost_ba_get_prior(self.h, None, None)
        H = np.zeros((n, n)); b = np.zeros(n)
        self.L.cmlhost_ba_get_prior(self.h, _p(H, _d), _p(b, _d))
        return H, b

    def point_flags(self):
        n = self.counts()["points"]
        tm = np.zeros(n, np.uint8); mg = np.zeros(n, np.uint8); ih = np.zeros(n, np.float32)
        self.L.cmlhost_ba_get_point_flags(self.h, _p(tm, _u8), _p(mg, _u8), _p(ih, _f))
        return tm, mg, ih

    def coarse_depth_points(self, kf_index, K):
        """getGoodPointsForTracking + the host half of makeCoarseDepthL0 (TR.cpp:521-553): (n, 4) array of (u, v, idepth, weight) in keyframe kf_index"""
        cap = max(self.counts()["points"], 1)
        out = np.zeros((cap, 4))
        k = np.ascontiguousarray(K, np.float64)
        n = self.L.cmlhost_ba_coarse_depth_points(self.h, int(kf_index), _p(k, _d), _p(out, _d), cap)
        if n < 0:
            raise RuntimeError("cmlhost_ba_coarse_depth_points: %d" % n)
        return out[:n]

    def run_timing(self):
        """host clock of the last resident run(), microseconds: window commit (edits handed over + index positions + pair records staged) | preamble pass
        enqueued | resident state staged | iterations enqueued | cmlhip_ba_finish_run (the ONE host wait: loop, re-anchoring, closing pass, readback) |
        host bookkeeping behind it"""
        t = np.zeros(6)
        self.L.cmlhost_ba_run_timing(self.h, _p(t, _d))
        return dict(zip(("commit_window", "enqueue_first_pass", "resident_state", "enqueue_iterations", "wait_and_readback", "bookkeeping"), [float(x) for x in t]))

    def rejected(self):
        return self.L.cmlhost_ba_rejected(self.h)

    def last_lambda(self):
        return self.L.cmlhost_ba_last_lambda(self.h)

    def m_energy(self):
        return self.L.cmlhost_ba_calc_m_energy(self.h)

    def l_energy(self):
        return self.L.cmlhost_ba_calc_l_energy(self.h)

    def last_error(self):
        return (self.L.cmlhost_ba_last_error(self.h) or b"").decode()

    def counts(self):
        v = [_i() for _ in range(5)]
        self.L.cmlhost_ba_counts(self.h, *[C.byref(x) for x in v])
        return dict(zip(("frames", "points", "residuals", "outliers", "iterations"), [x.value for x in v]))

    def frame(self, f):
        R = np.zeros(9); t = np.zeros(3); ab = np.zeros(2); st = np.zeros(10); th = _d()
        self.L.cmlhost_ba_get_frame(self.h, f, _p(R, _d), _p(t, _d), _p(ab, _d), _p(st, _d), C.byref(th))
        return dict(R=R.reshape(3, 3), t=t, ab=ab, state=st, th=th.value)

    def points(self):
        n = self.counts()["points"]
        idp = np.zeros(n); alive = np.zeros(n, np.uint8); ng = np.zeros(n, np.int32)
        self.L.cmlhost_ba_get_points(self.h, _p(idp, _d), _p(alive, _u8), _p(ng, _i))
        return idp, alive, ng

    def residual_states(self):
        n = self.counts()["residuals"]
        st = np.zeros(n, np.int32); alive = np.zeros(n, np.uint8); good = np.zeros(n, np.uint8)
        self.L.cmlhost_ba_get_residual_states(self.h, _p(st, _i), _p(alive, _u8), _p(good, _u8))
        return st, alive, good

    def outliers(self):
        n = self.counts()["outliers"]
        o = np.zeros(max(n, 1), np.int32)
        self.L.cmlhost_ba_get_outliers(self.h, _p(o, _i))
        return o[:n]

    def algebra(self):
        N = self.counts()["frames"]
        n = 8 * N + 4
        adH = np.zeros(N * N * 64); adT = np.zeros(N * N * 64); adHTd = np.zeros(N * N * 8, np.float32)
        pairs = np.zeros(N * N, abi.BA_PAIR_DTYPE); prior = np.zeros(8 * N); dprior = np.zeros(8 * N); ns = np.zeros(7 * n)
        self.L.cmlhost_ba_get_algebra(self.h, _p(adH, _d), _p(adT, _d), _p(adHTd, _f), pairs.ctypes.data_as(_P(abi.BAPair)),
                                      _p(prior, _d), _p(dprior, _d), _p(ns, _d))
        return dict(adH=adH, adT=adT, adHTd=adHTd, pairs=pairs, prior=prior, dprior=dprior, nullspaces=ns.reshape(7, n))

    def orthogonalize(self, x):
        x = np.ascontiguousarray(x, np.float64).copy()
        self.L.cmlhost_ba_orthogonalize(self.h, _p(x, _d), len(x))
        return x

    def set_indirect_points(self, xyz, obs):
        """The INDIRECTGROUP map points of the window's frames and their observations (abi.REPROJ_OBS_DTYPE), addIndirectToProblem's inputs."""
        xyz = np.ascontiguousarray(xyz, np.float64); obs = np.ascontiguousarray(obs, abi.REPROJ_OBS_DTYPE)
        self._n_indirect = len(xyz)
        self.L.cmlhost_ba_set_indirect_points(self.h, len(xyz), _p(xyz, _d), len(obs), obs.ctypes.data_as(_P(abi.ReprojObs)))

    def indirect(self):
        """(indirectX of the last solve [6N], point uncertainties [M], x of the last solve [8N+4])"""
        N = self.counts()["frames"]
        x6 = np.zeros(6 * N); unc = np.zeros(max(getattr(self, "_n_indirect", 0), 1)); x = np.zeros(8 * N + 4)
        n = self.L.cmlhost_ba_get_indirect(self.h, _p(x6, _d), _p(unc, _d), _p(x, _d))
        return (x6 if n else None), unc[:getattr(self, "_n_indirect", 0)], x

    def energies(self, cap=16):
        e = np.zeros(cap)
        n = self.L.cmlhost_ba_stats(self.h, _p(e, _d), cap)
        return e[:n]


# cmlhost_ba_export records (libcml_amd/host/capi.cpp)
HOST_BA_FRAME_DTYPE = np.dtype([("eval_q", "<f8", (4,)), ("eval_t", "<f8", (3,)), ("pre_q", "<f8", (4,)), ("pre_t", "<f8", (3,)), ("state", "<f8", (10,)),
                                ("state_zero", "<f8", (10,)), ("prior_zero", "<f8", (10,)), ("ab_exposure", "<f8"), ("frameEnergyTH", "<f8"), ("image_id", "<u8"),
                                ("id", "<i4"), ("keyid", "<i4"), ("flagged", "<i4"), ("numMarginalized", "<i4"), ("numResidualsOut", "<i4"), ("pad", "<i4")])
HOST_BA_POINT_DTYPE = np.dtype([("idepth", "<f8"), ("x", "<f4"), ("y", "<f4"), ("colors", "<f4", (8,)), ("weights", "<f4", (8,)), ("idepth_zero", "<f4"),
                                ("priorF", "<f4"), ("idepth_hessian", "<f4"), ("pad0", "<f4"), ("host", "<i4"), ("hasDepthPrior", "<i4"), ("numGoodResiduals", "<i4"),
                                ("lastResidual", "<i4", (2,)), ("lastResidualState", "<i4", (2,)), ("toMarginalize", "<i4"), ("marginalized", "<i4"), ("alive", "<i4")])
HOST_BA_RESIDUAL_DTYPE = np.dtype([("state_energy", "<f8"), ("state_NewEnergy", "<f8"), ("point", "<i4"), ("target", "<i4"), ("state_state", "<i4"),
                                   ("state_NewState", "<i4"), ("isLinearized", "<i4"), ("good", "<i4"), ("alive", "<i4"), ("pad", "<i4")])


# computeResidual + computeHessian provider signature of cml_amd::DSOTracker::EvalFn
TRACKER_EVAL_FN = C.CFUNCTYPE(_i, _vp, _i, _P(_d), _P(_d), _P(_d), _P(_d), _d, _P(abi.TrackerParams), _P(abi.TrackerResult))


class HostTracker:
    def __init__(self, ctx):
        """ctx = device.Ctx, or None for a tracker whose evaluations come from set_eval() (control-flow tests without a GPU)."""
        self.ctx = ctx
        self.L = lib()
        self.h = self.L.cmlhost_tracker_create(ctx.h if ctx is not None else None)
        self._eval_keep = None
        self.last_call_s = None                                 # seconds inside the last C call of track_with_motion_model / track_and_trace (without this wrapper's array packing)

    def set_eval(self, fn):
        """fn(level, R[9], t[3], K[4], aff[2], b0, prm, result) -> int; None restores the device evaluation."""
        if fn is None:
            self._eval_keep = None
            self.L.cmlhost_tracker_set_eval(self.h, None, None)
            return

        def tramp(user, level, R, t, K, aff, b0, prm, out):
            return int(fn(level, np.array(R[:9]), np.array(t[:3]), np.array(K[:4]), np.array(aff[:2]), float(b0), prm.contents, out))
        self._eval_keep = TRACKER_EVAL_FN(tramp)
        self.L.cmlhost_tracker_set_eval(self.h, C.cast(self._eval_keep, _vp), None)

    def steps(self, cap=512):
        lv = np.zeros(cap, np.int32); it = np.zeros(cap, np.int32); ac = np.zeros(cap, np.int32); lam = np.zeros(cap)
        n = self.L.cmlhost_tracker_steps(self.h, cap, _p(lv, _i), _p(it, _i), _p(ac, _i), _p(lam, _d))
        n = min(n, cap)
        return lv[:n].copy(), it[:n].copy(), ac[:n].copy(), lam[:n].copy()

    def set_last_residual(self, is_correct, rmse):
        r = np.ascontiguousarray(rmse, np.float64)
        self.L.cmlhost_tracker_set_last_residual(self.h, int(is_correct), len(r), _p(r, _d))

    def track_with_motion_model(self, new_image, levels, hyps, ref_exp, init_exp, batched=False):
        """hyps: list of (R, t) refToNew candidates.  Returns the adopted try (DSOTracker.h:238-383)."""
        H = np.zeros((len(hyps), 12))
        for i, (R, t) in enumerate(hyps):
            H[i, :9] = np.asarray(R, np.float64).ravel(); H[i, 9:] = t
        re = np.ascontiguousarray(ref_exp, np.float64); ie = np.ascontiguousarray(init_exp, np.float64)
        R = np.zeros(9); t = np.zeros(3); oe = np.zeros(2); E = np.zeros(8); nt = np.zeros(8, np.int32); ns = np.zeros(8, np.int32)
        ok, sat, win, tries = _i(), _i(), _i(), _i()
        lcr = _d()
        t_c = time.perf_counter()
        good = self.L.cmlhost_tracker_track_with_motion_model(self.h, int(new_image), levels, len(hyps), _p(H, _d), _p(re, _d), _p(ie, _d), _p(R, _d), _p(t, _d),
                                                              _p(oe, _d), _p(E, _d), _p(nt, _i), _p(ns, _i), C.byref(ok), C.byref(sat), C.byref(win),
                                                              C.byref(tries), C.byref(lcr), int(batched))
        self.last_call_s = time.perf_counter() - t_c          # inside the C call alone
        return dict(haveOneGood=bool(good), R=R.reshape(3, 3), t=t, exposure=oe, E=E, numTerms=nt, numSat=ns, isCorrect=bool(ok.value),
                    tooManySaturated=bool(sat.value), winner=win.value, tries=tries.value, lastCoarseRMSE=lcr.value)

    def track_and_trace(self, tracer, new_image, levels, hyps, ref_exp, init_exp, traced_frame_id, frame_ids, host_poses, ref_index, K):
        """cmlhost_frame_track_and_trace: trackWithMotionModel (batched) and, behind it in the same enqueue, traceNewCoarse of `tracer`'s resident set
        against the first hypothesis' result — one host wait.  host_poses: list of (R, t, a, b) world -> camera per window keyframe.
        Returns (tracking result dict as track_with_motion_model, kept, counts[6], pairs used [TRACE_PAIR_DTYPE] or None)."""
        H = np.zeros((len(hyps), 12))
        for i, (R, t) in enumerate(hyps):
            H[i, :9] = np.asarray(R, np.float64).ravel(); H[i, 9:] = t
        re = np.ascontiguousarray(ref_exp, np.float64); ie = np.ascontiguousarray(init_exp, np.float64)
        ids = np.ascontiguousarray(frame_ids, np.int32)
        hp = np.zeros((len(host_poses), 14))
        for i, (R, t, a, b) in enumerate(host_poses):
            hp[i, :9] = np.asarray(R, np.float64).ravel(); hp[i, 9:12] = t; hp[i, 12] = a; hp[i, 13] = b
        Kd = np.ascontiguousarray(K, np.float64)
        R = np.zeros(9); t = np.zeros(3); oe = np.zeros(2); E = np.zeros(8); nt = np.zeros(8, np.int32); ns = np.zeros(8, np.int32)
        ok, sat, win, tries, kept = _i(), _i(), _i(), _i(), _i()
        lcr = _d()
        counts = np.zeros(6, np.int32); pairs = np.zeros(len(host_poses), abi.TRACE_PAIR_DTYPE)
        t_c = time.perf_counter()
        good = self.L.cmlhost_frame_track_and_trace(self.h, tracer.h, int(new_image), levels, len(hyps), _p(H, _d), _p(re, _d), _p(ie, _d), int(traced_frame_id),
                                                    len(ids), _p(ids, _i), _p(hp, _d), int(ref_index), _p(Kd, _d), _p(R, _d), _p(t, _d), _p(oe, _d), _p(E, _d),
                                                    _p(nt, _i), _p(ns, _i), C.byref(ok), C.byref(sat), C.byref(win), C.byref(tries), C.byref(lcr), C.byref(kept),
                                                    _p(counts, _i), pairs.ctypes.data)
        self.last_call_s = time.perf_counter() - t_c          # inside the C call alone (the array packing above is this Python wrapper's, not the library's)
        if good < 0:
            raise RuntimeError("cmlhost_frame_track_and_trace: " + self.L.cmlhost_tracker_last_error(self.h).decode() + " / " + self.L.cmlhost_tracer_last_error(tracer.h).decode())
        res = dict(haveOneGood=bool(good), R=R.reshape(3, 3), t=t, exposure=oe, E=E, numTerms=nt, numSat=ns, isCorrect=bool(ok.value),
                   tooManySaturated=bool(sat.value), winner=win.value, tries=tries.value, lastCoarseRMSE=lcr.value)
        return res, bool(kept.value), counts, (pairs if kept.value else None)

    def close(self):
        if self.h:
            self.L.cmlhost_tracker_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_calibration(self, fx, fy, cx, cy):
        self.L.cmlhost_tracker_set_calibration(self.h, fx, fy, cx, cy)

    def set_param(self, name, value):
        if self.L.cmlhost_tracker_set_param(self.h, name.encode(), float(value)) != 0:
            raise KeyError(name)

    def make_coarse_depth(self, ref_image, levels, pts):
        a = np.ascontiguousarray(pts, np.float64)
        nout = (C.c_int * 8)()
        ok = self.L.cmlhost_tracker_make_coarse_depth(self.h, int(ref_image), levels, _p(a, _d), len(a), nout)
        if not ok:
            raise RuntimeError(self.L.cmlhost_tracker_last_error(self.h).decode())
        return list(nout[:levels])

    def optimize(self, new_image, levels, R, t, ref_exp, cur_exp):
        R = np.ascontiguousarray(R, np.float64).ravel().copy(); t = np.ascontiguousarray(t, np.float64).copy()
        re = np.ascontiguousarray(ref_exp, np.float64); ce = np.ascontiguousarray(cur_exp, np.float64).copy()
        E = np.zeros(8); nt = np.zeros(8, np.int32); nsat = np.zeros(8, np.int32); flow = np.zeros(3); rel = np.zeros(2); cov = np.zeros(6)
        ok, sat = _i(), _i(); its = np.zeros(8, np.int32)
        self.L.cmlhost_tracker_optimize(self.h, int(new_image), levels, _p(R, _d), _p(t, _d), _p(re, _d), _p(ce, _d), _p(E, _d), _p(nt, _i),
                                        _p(nsat, _i), _p(flow, _d), _p(rel, _d), _p(cov, _d), C.byref(ok), C.byref(sat), _p(its, _i))
        return dict(R=R.reshape(3, 3), t=t, exposure=ce, E=E, numTerms=nt, numSat=nsat, flow=flow, relAff=rel, covariance=cov,
                    isCorrect=bool(ok.value), tooManySaturated=bool(sat.value), iterations=its)


class HostTracer:
    """cml_amd::DSOTracer (flat mirror of the reference's immature-point tracer)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.L = lib()
        self.h = self.L.cmlhost_tracer_create(ctx.h)

    def close(self):
        if self.h:
            self.L.cmlhost_tracer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_point(self, x, y, host_frame_id, gray, dpatch, gradH, type=1.0):
        g = np.ascontiguousarray(gray, np.float32); d = np.ascontiguousarray(dpatch, np.float32); G = np.ascontiguousarray(gradH, np.float64)
        return self.L.cmlhost_tracer_add_point(self.h, float(x), float(y), int(host_frame_id), _p(g, _f), _p(d, _f), _p(G, _d), float(type))

    def add_points(self, xy, host_frame_id, gray, dpatch, gradH):
        """makeNewTraces' records of one keyframe in one call: xy (n, 2), gray (n, 8), dpatch (n, 24), gradH (n, 4)"""
        xy = np.ascontiguousarray(xy, np.float32); g = np.ascontiguousarray(gray, np.float32); d = np.ascontiguousarray(dpatch, np.float32); G = np.ascontiguousarray(gradH, np.float64)
        return self.L.cmlhost_tracer_add_points(self.h, len(xy), _p(xy, _f), int(host_frame_id), _p(g, _f), _p(d, _f), _p(G, _d))

    def compact(self):
        self.L.cmlhost_tracer_compact(self.h)

    def prepare_resident(self, frame_ids):
        """the device-resident immature set brought up to date for this frame list (a keyframe's closing act: the next frame's trace then starts at once)"""
        ids = np.ascontiguousarray(frame_ids, np.int32)
        if not self.L.cmlhost_tracer_prepare_resident(self.h, len(ids), _p(ids, _i)):
            raise RuntimeError(self.L.cmlhost_tracer_last_error(self.h).decode())

    def frame_ids(self):
        out = np.zeros(max(self.L.cmlhost_tracer_count(self.h), 1), np.int32)
        self.L.cmlhost_tracer_get_frame_ids(self.h, _p(out, _i))
        return out[:self.L.cmlhost_tracer_count(self.h)]

    def add_activated_to_ba(self, ba, activated, frame_ids):
        """the activated points handed to BA::addPoints (colours = gray patch, gradient weights, host = window index of the point's keyframe);
        returns (first BA point index, (n, 2) int pixels)"""
        idx = np.ascontiguousarray(activated, np.int32); ids = np.ascontiguousarray(frame_ids, np.int32)
        xy = np.zeros((max(len(idx), 1), 2), np.int32)
        first = self.L.cmlhost_tracer_add_activated_to_ba(self.h, ba.h, len(idx), _p(idx, _i), len(ids), _p(ids, _i), _p(xy, _i))
        if first < 0:
            raise RuntimeError("cmlhost_tracer_add_activated_to_ba: bad point index / frame id")
        return first, xy[:len(idx)]

    def immature_counts(self, frame_ids):
        """live immature points per frame id"""
        ids = np.ascontiguousarray(frame_ids, np.int32); out = np.zeros(max(len(ids), 1), np.int32)
        self.L.cmlhost_tracer_immature_counts(self.h, len(ids), _p(ids, _i), _p(out, _i))
        return [int(x) for x in out[:len(ids)]]

    def trace_new_coarse(self, image_id, traced_frame_id, frame_ids, pairs):
        ids = np.ascontiguousarray(frame_ids, np.int32); pr = np.ascontiguousarray(pairs, abi.TRACE_PAIR_DTYPE)
        counts = np.zeros(6, np.int32)
        ok = self.L.cmlhost_tracer_trace(self.h, int(image_id), int(traced_frame_id), len(ids), _p(ids, _i), pr.ctypes.data, _p(counts, _i))
        if not ok:
            raise RuntimeError(self.L.cmlhost_tracer_last_error(self.h).decode())
        return counts

    def activate_points(self, frame_ids, image_ids, K, w, h, pairs):
        ids = np.ascontiguousarray(frame_ids, np.int32); im = np.ascontiguousarray(image_ids, np.uint64)
        Kd = np.ascontiguousarray(K, np.float64); pr = np.ascontiguousarray(pairs, abi.ACTIVATION_PAIR_DTYPE)
        out = np.zeros(self.L.cmlhost_tracer_count(self.h) + 1, np.int32)
        n = self.L.cmlhost_tracer_activate(self.h, len(ids), _p(ids, _i), _p(im, C.c_uint64), _p(Kd, _d), int(w), int(h), pr.ctypes.data, _p(out, _i), len(out))
        if n < 0:
            raise RuntimeError(self.L.cmlhost_tracer_last_error(self.h).decode())
        return out[:n].copy()

    def points(self):
        n = self.L.cmlhost_tracer_count(self.h)
        pts = np.zeros(n, abi.IMMATURE_POINT_DTYPE); alive = np.zeros(n, np.uint8); act = np.zeros(n, np.uint8); idp = np.zeros(n, np.float32)
        self.L.cmlhost_tracer_get_points(self.h, pts.ctypes.data, _p(alive, _u8), _p(act, _u8), _p(idp, _f))
        return pts, alive, act, idp


def window_to_host_ba(ctx, W, image_id_base=1000, levels=1):
    """Upload a synthetic window (libcml_amd.synth) and register it with a HostBA the way Hybrid::directMap would:
    device-side pyramid build, addNewFrame per keyframe, addPoint per active point (colours / gradient weights read
    back from the device pyramid, DSOContext.h:87-91 / BA.cpp:405-411)."""
    ba = HostBA(ctx)
    fx, fy, cx, cy = W.K
    ba.set_calibration(fx, fy, cx, cy, W.w, W.h)
    grads0 = []
    for k in range(W.N):
        ctx.pyramid_build(image_id_base + k, W.gray[k], levels)
        grads0.append(ctx.pyramid_get(image_id_base + k, 0))
    for k in range(W.N):
        a, b = W.aff_eval[k]
        ba.add_frame(image_id_base + k, W.R_eval[k], W.t_eval[k], a, b, float(W.ab_exposure[k]))
    for k in range(W.N):
        ba.set_frame_state(k, W.state[k])
    colors, weights = synth.point_colors_weights(W, grads0)
    for i in range(W.P):
        ba.add_point(W.pts["x"][i], W.pts["y"][i], W.pts["idepth"][i], W.pts["host"][i], colors[i], weights[i])
    ba.synth_inputs = {"colors": colors, "weights": weights, "grads0": grads0}      # what was registered (for checkers that replay the window)
    return ba


# ---- flat records of the IndirectCameraOptimizer / IndirectBundleAdjustment mirrors (libcml_amd/host/capi.cpp)
HOST_MATCHING_DTYPE = np.dtype([("has_map_point", "<i4"), ("level", "<i4"), ("X", "<f8", (3,)), ("obs", "<f8", (2,)), ("scale_factor_base", "<f8"),
                                ("descriptor_distance", "<f8")])
HOST_LBA_FRAME_DTYPE = np.dtype([("id", "<i4"), ("pad", "<i4"), ("R", "<f8", (9,)), ("t", "<f8", (3,)), ("K", "<f8", (4,))])
HOST_LBA_POINT_DTYPE = np.dtype([("id", "<i4"), ("reference_frame_id", "<i4"), ("X", "<f8", (3,))])
HOST_LBA_APPARITION_DTYPE = np.dtype([("point", "<i4"), ("frame_id", "<i4"), ("obs", "<f8", (2,)), ("level", "<i4"), ("pad", "<i4"), ("scale_factor_base", "<f8")])


class HostCameraOptimizer:
    """cml_amd::IndirectCameraOptimizer (flat mirror of the reference's g2o pose optimiser)."""

    def __init__(self, ctx, check_outliers=True):
        self.ctx = ctx; self.L = lib(); self.check = bool(check_outliers)

    def optimize(self, frameR, frameT, K, matchings, outliers, camera=None, compute_covariance=False):
        """The Levenberg overload.  outliers: uint8 array (len(matchings) or any other length = "reset"), returns (isOk, R, t, cov, outliers)."""
        m = np.ascontiguousarray(matchings, HOST_MATCHING_DTYPE); n = len(m)
        fr = np.ascontiguousarray(frameR, np.float64); ft = np.ascontiguousarray(frameT, np.float64); Kd = np.ascontiguousarray(K, np.float64)
        out = np.zeros(max(n, len(outliers), 1), np.uint8); out[:len(outliers)] = outliers
        cr = ct = None
        if camera is not None:
            cr = np.ascontiguousarray(camera[0], np.float64); ct = np.ascontiguousarray(camera[1], np.float64)
        ok = _i(0); R = np.zeros(9); t = np.zeros(3); cov = np.zeros(6)
        rc = self.L.cmlhost_pnp_optimize(self.ctx.h, int(self.check), _p(fr, _d), _p(ft, _d), cr.ctypes.data if cr is not None else None,
                                         ct.ctypes.data if ct is not None else None, _p(Kd, _d), n, m.ctypes.data, _p(out, _u8), len(outliers),
                                         int(bool(compute_covariance)), C.byref(ok), _p(R, _d), _p(t, _d), _p(cov, _d))
        if rc:
            raise RuntimeError("cmlhost_pnp_optimize failed")
        return bool(ok.value), R.reshape(3, 3), t, cov, out[:n].copy()

    def optimize_points(self, frameR, frameT, K, points, compute_covariance=False):
        """The Gauss-Newton overload; returns (isOk, R, t, cov, indices of the outlier points)."""
        m = np.ascontiguousarray(points, HOST_MATCHING_DTYPE); n = len(m)
        fr = np.ascontiguousarray(frameR, np.float64); ft = np.ascontiguousarray(frameT, np.float64); Kd = np.ascontiguousarray(K, np.float64)
        idx = np.zeros(max(n, 1), np.int32); nidx = _i(0); ok = _i(0); R = np.zeros(9); t = np.zeros(3); cov = np.zeros(6)
        rc = self.L.cmlhost_pnp_optimize_points(self.ctx.h, int(self.check), _p(fr, _d), _p(ft, _d), _p(Kd, _d), n, m.ctypes.data, _p(idx, _i), C.byref(nidx),
                                                int(bool(compute_covariance)), C.byref(ok), _p(R, _d), _p(t, _d), _p(cov, _d))
        if rc:
            raise RuntimeError("cmlhost_pnp_optimize_points failed")
        return bool(ok.value), R.reshape(3, 3), t, cov, idx[:nidx.value].copy()


class HostLocalBA:
    """cml_amd::IndirectBundleAdjustment (flat mirror of the reference's g2o local bundle adjustment)."""

    def __init__(self, ctx, num_iteration=5, refine_iteration=0, remove_edge=True):
        self.ctx = ctx; self.L = lib()
        self.h = self.L.cmlhost_lba_create(ctx.h)
        self.L.cmlhost_lba_set_params(self.h, int(num_iteration), int(refine_iteration), int(bool(remove_edge)))

    def close(self):
        if self.h:
            self.L.cmlhost_lba_destroy(self.h); self.h = None

    def local_optimize(self, local, fixed, points, apparitions, fix_frames, stop_flag=None):
        """stop_flag: a 1-element uint8 array = the reference's pbStopFlag (IndirectBundleAdjustment.h:27), or None."""
        self._nl, self._np = len(local), len(points)
        lo = np.ascontiguousarray(local, HOST_LBA_FRAME_DTYPE); fx = np.ascontiguousarray(fixed, HOST_LBA_FRAME_DTYPE)
        pt = np.ascontiguousarray(points, HOST_LBA_POINT_DTYPE); ap = np.ascontiguousarray(apparitions, HOST_LBA_APPARITION_DTYPE)
        return bool(self.L.cmlhost_lba_local_optimize(self.h, len(lo), lo.ctypes.data, len(fx), fx.ctypes.data, len(pt), pt.ctypes.data, len(ap), ap.ctypes.data,
                                                       int(bool(fix_frames)), stop_flag.ctypes.data if stop_flag is not None else None))

    def last_error(self):
        return self.L.cmlhost_lba_last_error(self.h).decode()

    def apply(self, cap=1 << 20):
        lo = np.zeros(self._nl, HOST_LBA_FRAME_DTYPE); X = np.zeros((self._np, 3)); rem = np.zeros(2 * cap, np.int32); res = abi.LbaResult()
        n = self.L.cmlhost_lba_apply(self.h, self._nl, lo.ctypes.data, self._np, _p(X, _d), _p(rem, _i), cap, C.byref(res))
        return lo, X, rem[:2 * min(n, cap)].reshape(-1, 2).copy(), res


class HostInitializer:
    """cml_amd::DSOInitializer (flat mirror of the reference's coarse initializer; calcResAndGS runs on the device)."""

    def __init__(self, ctx):
        self.ctx = ctx; self.L = lib()
        self.h = self.L.cmlhost_init_create(ctx.h)

    def close(self):
        if self.h:
            self.L.cmlhost_init_destroy(self.h); self.h = None

    def set_first(self, grays, Ks, pixels, ref_qt, ref_exposure=1.0):
        """grays: per level float32 (h, w); Ks: per level (fx, fy, cx, cy); pixels: per level (x int array, y int array) in raster order."""
        n = len(grays)
        self._keep = [np.ascontiguousarray(g, np.float32) for g in grays]
        w = np.array([g.shape[1] for g in self._keep], np.int32); h = np.array([g.shape[0] for g in self._keep], np.int32)
        K4 = np.ascontiguousarray(np.array(Ks, np.float64).reshape(n, 4))
        ptrs = (_vp * n)(*[g.ctypes.data for g in self._keep])
        npx = np.array([len(p[0]) for p in pixels], np.int32)
        px = np.ascontiguousarray(np.concatenate([np.asarray(p[0], np.int32) for p in pixels])); py = np.ascontiguousarray(np.concatenate([np.asarray(p[1], np.int32) for p in pixels]))
        qt = np.ascontiguousarray(ref_qt, np.float64)
        ok = self.L.cmlhost_init_set_first(self.h, n, _p(w, _i), _p(h, _i), _p(K4, _d), ptrs, _p(npx, _i), _p(px, _i), _p(py, _i), _p(qt, _d), float(ref_exposure))
        self.n_levels = n
        return bool(ok)

    def try_initialize(self, image_id, frame_qt, exposure=1.0):
        qt = np.ascontiguousarray(frame_qt, np.float64)
        return self.L.cmlhost_init_try(self.h, int(image_id), _p(qt, _d), float(exposure))

    def state(self):
        qt = np.zeros(7); sn = _i(0); fid = _i(0); cnt = np.zeros(3, np.int32); rs = _f(0)
        self.L.cmlhost_init_state(self.h, _p(qt, _d), C.byref(sn), C.byref(fid), _p(cnt, _i), C.byref(rs))
        return dict(qt=qt, snapped=bool(sn.value), frame_id=fid.value, calc_calls=int(cnt[0]), accepted=int(cnt[1]), rejected=int(cnt[2]), rescale=rs.value)

    def points(self, lvl):
        n = self.L.cmlhost_init_level_size(self.h, lvl)
        xy = np.zeros((n, 2), np.float32); iR = np.zeros(n, np.float32); idp = np.zeros(n, np.float32); good = np.zeros(n, np.uint8)
        lh = np.zeros(n, np.float32); par = np.zeros(n, np.int32); nb = np.zeros((n, 10), np.int32)
        self.L.cmlhost_init_get_points(self.h, lvl, _p(xy, _f), _p(iR, _f), _p(idp, _f), _p(good, _u8), _p(lh, _f), _p(par, _i), _p(nb, _i))
        return dict(xy=xy, iR=iR, idepth=idp, good=good, last_hessian=lh, parent=par, neighbours=nb)

    def last_error(self):
        return self.L.cmlhost_init_last_error(self.h).decode()
