"""ctypes binding of libcmlhip.so (the C ABI of include/cmlhip.h) + a thin numpy convenience class.

This is plumbing for tests/bench/smoke: every call goes through the extern "C" boundary a C++ host would use.
There is no CPU fallback: loading fails loudly when the HIP library is missing, and Ctx() fails when no
gfx950 device is usable.
"""
import ctypes as C
import os

import numpy as np

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcmlhip.so")

_lib = None

_d, _f, _i, _u8 = C.c_double, C.c_float, C.c_int, C.c_ubyte
_P = C.POINTER
_ctx = C.c_void_p

PROTOTYPES = {
    "cmlhip_abi_version": (C.c_int, []),
    "cmlhip_device_count": (C.c_int, []),
    "cmlhip_create": (C.c_int, [_P(_ctx), _P(abi.Limits)]),
    "cmlhip_destroy": (None, [_ctx]),
    "cmlhip_last_error": (C.c_char_p, [_ctx]),
    "cmlhip_synchronize": (C.c_int, [_ctx]),
    "cmlhip_stream": (C.c_void_p, [_ctx]),
    "cmlhip_pyramid_put": (C.c_int, [_ctx, C.c_uint64, _i, _P(_f), _i, _i]),
    "cmlhip_pyramid_build": (C.c_int, [_ctx, C.c_uint64, _P(_f), _i, _i, _i]),
    "cmlhip_pyramid_build_async": (C.c_int, [_ctx, C.c_uint64, _P(_f), _i, _i, _i]),
    "cmlhip_pyramid_drop": (C.c_int, [_ctx, C.c_uint64]),
    "cmlhip_pyramid_level_size": (C.c_int, [_ctx, C.c_uint64, _i, _P(_i), _P(_i)]),
    "cmlhip_pyramid_get": (C.c_int, [_ctx, C.c_uint64, _i, _P(_f)]),
    "cmlhip_tracker_set_reference": (C.c_int, [_ctx, _i, _P(_f), _i]),
    "cmlhip_tracker_make_coarse_depth": (C.c_int, [_ctx, C.c_uint64, _i, _P(_d), _i, _P(_i)]),
    "cmlhip_tracker_get_reference": (C.c_int, [_ctx, _i, _P(_f), _P(_i)]),
    "cmlhip_tracker_eval": (C.c_int, [_ctx, C.c_uint64, _i, _P(_d), _P(_d), _P(_d), _P(_d), _d,
                                      _P(abi.TrackerParams), _i, _P(abi.TrackerResult)]),
    "cmlhip_tracker_get_warped": (C.c_int, [_ctx, _P(_f), _i, _P(_i)]),
    "cmlhip_tracker_optimize_batch": (C.c_int, [_ctx, C.c_uint64, _i, _P(_d), _P(_d), _P(_d), _P(abi.TrackerParams), _i, _i, _d, _i,
                                                _P(abi.TrackerHypothesis), _P(abi.TrackerOptResult)]),
    "cmlhip_set_device_share": (C.c_int, [_ctx, _i]),
    "cmlhip_tracker_optimize_batch_async": (C.c_int, [_ctx, C.c_uint64, _i, _P(_d), _P(_d), _P(_d), _P(abi.TrackerParams), _i, _i, _d, _i,
                                                      _P(abi.TrackerHypothesis)]),
    "cmlhip_tracker_optimize_wait": (C.c_int, [_ctx, _P(abi.TrackerOptResult)]),
    "cmlhip_tracer_trace_resident_tracked_async": (C.c_int, [_ctx, C.c_uint64, _P(abi.TracerParams), _i, C.c_void_p, C.c_void_p, _P(_d), _i]),
    "cmlhip_tracer_trace_resident_finish": (C.c_int, [_ctx, _i, _P(C.c_int), C.c_void_p]),
    "cmlhip_tracer_tracked_prepare": (C.c_int, [_ctx, _i, C.c_void_p, C.c_void_p, _P(_d)]),
    "cmlhip_ba_set_resident_indirect": (C.c_int, [_ctx, _i, _P(_d), _i, _P(abi.ReprojObs), _d, _d]),
    "cmlhip_ba_get_resident_indirect": (C.c_int, [_ctx, _P(_d), _P(_d), _P(_d)]),
    "cmlhip_ba_set_resident_prior": (C.c_int, [_ctx, _P(_d), _P(_d)]),
    "cmlhip_ba_set_params": (C.c_int, [_ctx, _P(abi.BAParams)]),
    "cmlhip_ba_upload_window": (C.c_int, [_ctx, _i, _P(abi.BAFrame), _i, _P(abi.BAPoint), _i, _P(abi.BAResidual)]),
    "cmlhip_ba_set_pairs": (C.c_int, [_ctx, _P(abi.BAPair)]),
    "cmlhip_ba_window_size": (C.c_int, [_ctx, _P(_i), _P(_i), _P(_i)]),
    "cmlhip_profile_enable": (C.c_int, [_ctx, _i]),
    "cmlhip_debug_timestamps": (C.c_int, [_ctx, _i, _P(C.c_longlong)]),
    "cmlhip_profile_stride": (C.c_int, [_ctx, _i]),
    "cmlhip_profile_select": (C.c_int, [_ctx, _i]),
    "cmlhip_trace_points": (C.c_int, [_ctx, C.c_uint64, _P(abi.TracerParams), _i, C.c_void_p, _i, C.c_void_p]),
    "cmlhip_tracer_set_points": (C.c_int, [_ctx, _i, C.c_void_p]),
    "cmlhip_tracer_trace_resident": (C.c_int, [_ctx, C.c_uint64, _P(abi.TracerParams), _i, C.c_void_p, _i, _P(C.c_int)]),
    "cmlhip_tracer_get_points": (C.c_int, [_ctx, _i, C.c_void_p]),
    "cmlhip_tracer_edit_points": (C.c_int, [_ctx, _i, _P(_i), _P(_i), _i, C.c_void_p]),
    "cmlhip_tracer_get_state": (C.c_int, [_ctx, _i, C.c_void_p]),
    "cmlhip_initializer_calc_res_and_gs": (C.c_int, [_ctx, C.c_uint64, _i, _P(abi.InitParams), _i, C.c_void_p, _P(C.c_float), _P(C.c_float), _P(C.c_float), _P(C.c_float), _P(C.c_float)]),
    "cmlhip_ba_finish_keyframe": (C.c_int, [_ctx, C.c_void_p, _P(C.c_int), _P(C.c_int), _P(C.c_float), _P(C.c_float), _P(C.c_float), _P(C.c_ubyte), _P(C.c_double), _P(C.c_float)]),
    "cmlhip_upload_scope_begin": (C.c_int, [_ctx]),
    "cmlhip_upload_scope_end": (C.c_int, [_ctx]),
    "cmlhip_ba_window_reset": (C.c_int, [_ctx]),
    "cmlhip_ba_window_append_points": (C.c_int, [_ctx, _i, C.c_void_p]),
    "cmlhip_ba_window_append_residuals": (C.c_int, [_ctx, _i, C.c_void_p]),
    "cmlhip_ba_window_retire_frame": (C.c_int, [_ctx, _i]),
    "cmlhip_ba_window_compact": (C.c_int, [_ctx, _i, _P(C.c_ubyte), _i, _P(C.c_ubyte)]),
    "cmlhip_ba_window_counts": (C.c_int, [_ctx, _P(C.c_int), _P(C.c_int)]),
    "cmlhip_ba_window_generation": (C.c_int, [_ctx, _P(C.c_uint)]),
    "cmlhip_ba_window_commit": (C.c_int, [_ctx, _i, C.c_void_p, _P(C.c_double), _P(C.c_float), _P(C.c_float), _i, _i, _P(C.c_int), _P(C.c_int)]),
    "cmlhip_ba_finish_run": (C.c_int, [_ctx, _i, _P(abi.BAResidentOut), C.c_void_p, _P(C.c_int), _P(C.c_int), _P(C.c_float), _P(C.c_float), _P(C.c_float), _P(C.c_ubyte), _P(C.c_double), _P(C.c_float)]),
    "cmlhip_pnp_optimize": (C.c_int, [_ctx, _P(C.c_double), _P(C.c_double), _P(C.c_double), _i, C.c_void_p, _P(C.c_ubyte), _i, _i, _i, _P(abi.PnpResult)]),
    "cmlhip_lba_set_stop_flag": (C.c_int, [_ctx, C.c_void_p]),
    "cmlhip_lba_optimize": (C.c_int, [_ctx, _i, C.c_void_p, _i, _P(C.c_double), _P(C.c_int), C.c_void_p, _i, _i, _i, _P(C.c_ubyte), _P(abi.LbaResult)]),
    "cmlhip_optimize_immature_points": (C.c_int, [_ctx, _i, _P(C.c_uint64), _P(C.c_double), C.c_void_p, _P(abi.TracerParams), _i, _i, C.c_void_p, _P(C.c_int), _P(C.c_float), _P(C.c_int)]),
    "cmlhip_optimize_immature_points_resident": (C.c_int, [_ctx, _i, _P(C.c_uint64), _P(C.c_double), C.c_void_p, _P(abi.TracerParams), _i, _i, _P(C.c_int), _P(C.c_int), _P(C.c_float), _P(C.c_int)]),
    "cmlhip_ba_relinearize_points": (C.c_int, [_ctx, _P(abi.BAAccumIn), _i, _P(C.c_int), _P(C.c_int)]),
    "cmlhip_ba_relinearize_points_packed": (C.c_int, [_ctx, _P(abi.BAAccumIn), _i, _P(_i), _P(_i), _P(C.c_ubyte), _P(_f), _P(_f), _P(_f)]),
    "cmlhip_ba_marginalize_points": (C.c_int, [_ctx, _P(abi.BAAccumIn), _i, _P(C.c_int), _P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    "cmlhip_ba_lin_energy": (C.c_int, [_ctx, _P(abi.BAAccumIn), _P(C.c_double), _P(C.c_int)]),
    "cmlhip_ba_get_res_to_zero": (C.c_int, [_ctx, _P(C.c_float), _P(C.c_ubyte)]),
    "cmlhip_ba_set_resident_state": (C.c_int, [_ctx, _P(abi.BAAccumIn), _P(abi.BAFrameState), _P(C.c_double), _P(C.c_double)]),
    "cmlhip_ba_resident_convergence": (C.c_int, [_ctx, C.c_double]),
    "cmlhip_ba_get_resident_log": (C.c_int, [_ctx, _P(C.c_int), _P(C.c_double), _i]),
    "cmlhip_ba_get_resident_state": (C.c_int, [_ctx, _P(abi.BAFrameState), _P(C.c_double), _P(abi.BALinResult)]),
    "cmlhip_profile_read": (C.c_int, [_ctx, _P(_f), _P(_f), _P(_f), _P(_i)]),
    "cmlhip_ba_set_frame_energy_th": (C.c_int, [_ctx, _P(_f)]),
    "cmlhip_ba_set_frame_b0": (C.c_int, [_ctx, _P(_f)]),
    "cmlhip_ba_set_arithmetic": (C.c_int, [_ctx, _i]),
    "cmlhip_ba_set_resident_outputs": (C.c_int, [_ctx, _i]),
    "cmlhip_ba_get_resident_outputs": (C.c_int, [_ctx, _P(_i)]),
    "cmlhip_tracker_set_early_exit": (C.c_int, [_ctx, _d]),
    "cmlhip_ba_set_idepth": (C.c_int, [_ctx, _P(_d), _P(_f)]),
    "cmlhip_ba_get_idepth": (C.c_int, [_ctx, _P(_d)]),
    "cmlhip_ba_linearize": (C.c_int, [_ctx, _P(abi.BALinResult)]),
    "cmlhip_ba_apply": (C.c_int, [_ctx, _i]),
    "cmlhip_ba_linearize_apply": (C.c_int, [_ctx, _P(abi.BALinResult)]),
    "cmlhip_ba_accumulate": (C.c_int, [_ctx, _P(abi.BAAccumIn), _P(_d), _P(_d), _P(_d), _P(_d), _P(_d), _P(_d)]),
    "cmlhip_ba_solve": (C.c_int, [_ctx, _d, _P(_d), _P(_d), _i, _P(_d)]),
    "cmlhip_ba_backsub": (C.c_int, [_ctx, _P(_d), _P(_d)]),
    "cmlhip_ba_backup_points": (C.c_int, [_ctx]),
    "cmlhip_ba_step_points": (C.c_int, [_ctx, _P(_f)]),
    "cmlhip_ba_restore_points": (C.c_int, [_ctx]),
    "cmlhip_ba_get_states": (C.c_int, [_ctx, _P(_i), _P(_i), _P(_f), _P(_f), _P(_f), _P(_u8)]),
    "cmlhip_ba_get_rj": (C.c_int, [_ctx, _i, _P(_f)]),
    "cmlhip_ba_get_jpjdf": (C.c_int, [_ctx, _P(_f)]),
    "cmlhip_ba_get_center_projected": (C.c_int, [_ctx, _P(_f)]),
    "cmlhip_ba_get_point_acc": (C.c_int, [_ctx, _P(_f)]),
    "cmlhip_ba_get_pair_acc": (C.c_int, [_ctx, _i, _P(_f)]),
    "cmlhip_ba_get_index_maps": (C.c_int, [_ctx, _P(_i), _P(_i), _P(_i), _P(_i), _P(_i)]),
    "cmlhip_reproj_accumulate": (C.c_int, [_ctx, _i, _P(_d), _i, _P(_d), _i, _P(abi.ReprojObs), _d, _d, _P(_d), _P(_d),
                                           _P(_d), _P(_u8)]),
    "cmlhip_reproj_solve": (C.c_int, [_ctx, _i, _d, _P(_d)]),
    "cmlhip_event_mark": (C.c_int, [_ctx, _i]),
    "cmlhip_event_elapsed_ms": (C.c_int, [_ctx, _P(_f)]),
    "cmlhip_profile_next_launch": (C.c_int, [_ctx]),
    "cmlhip_ba_get_pairs": (C.c_int, [_ctx, C.c_void_p, _P(_f), _P(_f)]),
    "cmlhip_ba_linearize_async": (C.c_int, [_ctx]),
    "cmlhip_ba_iteration_async": (C.c_int, [_ctx, _d]),
    "cmlhip_ba_iteration_batch": (C.c_int, [_P(_ctx), C.c_int, _d]),
}


def lib():
    """Load libcmlhip.so (built in-tree by libcml_amd.build). Raises when it is missing — no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libcml_amd/libcmlhip.so is missing: run `python -m libcml_amd.build` "
                               "(hipcc --offload-arch=gfx950). There is no CPU fallback for the device path.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def ba_iteration_batch(ctxs, lam):
    """One resident iteration of every window in `ctxs` (device.Ctx objects) in five launches; see cmlhip_ba_iteration_batch."""
    arr = (_ctx * len(ctxs))(*[c.h for c in ctxs])
    ctxs[0].ck(lib().cmlhip_ba_iteration_batch(arr, len(ctxs), lam))


class CmlHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cmlhip status %d: %s" % (code, msg))
        self.code = code


def _p(a, t):
    return None if a is None else a.ctypes.data_as(_P(t))


class Ctx:
    """One device context (one HIP stream)."""

    def __init__(self, device_id=0, max_frames=8, max_points=4096, max_residuals=65536, max_tracker_points=1 << 20,
                 max_reproj_obs=1 << 16, texel_format=abi.TEXEL_F32):
        self.L = lib()
        self.h = _ctx()
        lim = abi.Limits(device_id, max_frames, max_points, max_residuals, max_tracker_points, max_reproj_obs, texel_format)
        rc = self.L.cmlhip_create(C.byref(self.h), C.byref(lim))
        if rc != 0:
            raise CmlHipError(rc, "cmlhip_create failed (no usable gfx950 device? there is no CPU fallback)")
        self.N = self.P = self.R = 0

    def close(self):
        if self.h:
            self.L.cmlhip_destroy(self.h)
            self.h = _ctx()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ck(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise CmlHipError(rc, (self.L.cmlhip_last_error(self.h) or b"").decode())
        return rc

    def refresh_window_size(self):
        """N, P, R of the window currently uploaded (it may have been uploaded by the C++ host mirror)."""
        n, p, r = _i(), _i(), _i()
        self.ck(self.L.cmlhip_ba_window_size(self.h, C.byref(n), C.byref(p), C.byref(r)))
        self.N, self.P, self.R = n.value, p.value, r.value
        return self.N, self.P, self.R

    def profile_stride(self, stride):
        self.ck(self.L.cmlhip_profile_stride(self.h, stride))

    def profile_select(self, mask):
        self.ck(self.L.cmlhip_profile_select(self.h, mask))

    def profile_enable(self, max_iterations):
        self.ck(self.L.cmlhip_profile_enable(self.h, max_iterations))

    def profile_read(self):
        a, b, e, n = _f(), _f(), _f(), _i()
        self.ck(self.L.cmlhip_profile_read(self.h, C.byref(a), C.byref(b), C.byref(e), C.byref(n)))
        return a.value, b.value, e.value, n.value

    # ------------------------------------------------------------------ pyramids
    def pyramid_put(self, image_id, level, aos3):
        a = np.ascontiguousarray(aos3, np.float32)
        self.ck(self.L.cmlhip_pyramid_put(self.h, image_id, level, _p(a, _f), a.shape[1], a.shape[0]))

    def pyramid_build(self, image_id, gray, levels):
        g = np.ascontiguousarray(gray, np.float32)
        self.ck(self.L.cmlhip_pyramid_build(self.h, image_id, _p(g, _f), g.shape[1], g.shape[0], levels))

    def pyramid_build_async(self, image_id, gray, levels):
        """cmlhip_pyramid_build_async: returns at once; the array is kept alive here until the image is dropped or rebuilt"""
        g = np.ascontiguousarray(gray, np.float32)
        if not hasattr(self, "_async_images"):
            self._async_images = {}
        self._async_images[int(image_id)] = g
        self.ck(self.L.cmlhip_pyramid_build_async(self.h, image_id, _p(g, _f), g.shape[1], g.shape[0], levels))

    def pyramid_get(self, image_id, level):
        w, h = _i(), _i()
        self.ck(self.L.cmlhip_pyramid_level_size(self.h, image_id, level, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value, 3), np.float32)
        self.ck(self.L.cmlhip_pyramid_get(self.h, image_id, level, _p(out, _f)))
        return out

    def pyramid_drop(self, image_id):
        rc = self.L.cmlhip_pyramid_drop(self.h, image_id)
        getattr(self, "_async_images", {}).pop(int(image_id), None)
        return rc

    # ------------------------------------------------------------------ BA
    def ba_set_params(self, prm):
        self.ck(self.L.cmlhip_ba_set_params(self.h, C.byref(prm)))

    def ba_upload_window(self, frames, points, residuals):
        frames = np.ascontiguousarray(frames); points = np.ascontiguousarray(points); residuals = np.ascontiguousarray(residuals)
        self.N, self.P, self.R = len(frames), len(points), len(residuals)
        self.ck(self.L.cmlhip_ba_upload_window(self.h, self.N, frames.ctypes.data_as(_P(abi.BAFrame)), self.P,
                                                points.ctypes.data_as(_P(abi.BAPoint)), self.R,
                                                residuals.ctypes.data_as(_P(abi.BAResidual))))

    def ba_set_pairs(self, pairs):
        pairs = np.ascontiguousarray(pairs)
        self.ck(self.L.cmlhip_ba_set_pairs(self.h, pairs.ctypes.data_as(_P(abi.BAPair))))

    def ba_set_frame_energy_th(self, th):
        th = np.ascontiguousarray(th, np.float32)
        self.ck(self.L.cmlhip_ba_set_frame_energy_th(self.h, _p(th, _f)))

    def tracker_set_early_exit(self, rmse_bar):
        """cmlhip_tracker_set_early_exit: > 0: hypothesis 0 of the following batches may end them (results given up carry n_steps = -1); 0: off"""
        self.ck(self.L.cmlhip_tracker_set_early_exit(self.h, float(rmse_bar)))

    def set_device_share(self, n_contexts):
        """cmlhip_set_device_share: how many contexts launch on this device beside each other (sequence shards per GPU)"""
        self.ck(self.L.cmlhip_set_device_share(self.h, int(n_contexts)))

    def ba_set_resident_outputs(self, lean):
        """cmlhip_ba_set_resident_outputs: False = CMLHIP_RESIDENT_OUTPUTS_FULL (default), True = LEAN (see include/cmlhip.h)"""
        self.ck(self.L.cmlhip_ba_set_resident_outputs(self.h, 1 if lean else 0))

    def ba_resident_outputs_lean(self):
        m = _i(0)
        self.ck(self.L.cmlhip_ba_get_resident_outputs(self.h, C.byref(m)))
        return bool(m.value)

    def ba_set_arithmetic(self, relaxed):
        """cmlhip_ba_set_arithmetic: False = CMLHIP_ARITH_EXACT (default), True = CMLHIP_ARITH_RELAXED (throughput-regime residual kernel only)"""
        self.ck(self.L.cmlhip_ba_set_arithmetic(self.h, 1 if relaxed else 0))

    def ba_get_idepth(self):
        out = np.zeros(self.P)
        self.ck(self.L.cmlhip_ba_get_idepth(self.h, _p(out, _d)))
        return out

    def ba_set_idepth(self, idepth, idepth_zero=None):
        a = np.ascontiguousarray(idepth, np.float64)
        z = None if idepth_zero is None else np.ascontiguousarray(idepth_zero, np.float32)
        self.ck(self.L.cmlhip_ba_set_idepth(self.h, _p(a, _d), _p(z, _f)))

    def ba_linearize(self):
        out = abi.BALinResult()
        self.ck(self.L.cmlhip_ba_linearize(self.h, C.byref(out)), allow=(abi.ERR_NONFINITE,))
        return out

    def ba_apply(self, copy=1):
        self.ck(self.L.cmlhip_ba_apply(self.h, copy))

    def ba_accumulate(self, adH, adT, adHTd, cdelta, prior, dprior, cprior):
        n = 8 * self.N + 4
        self._keep = [np.ascontiguousarray(adH, np.float64), np.ascontiguousarray(adT, np.float64),
                      np.ascontiguousarray(adHTd, np.float32), np.ascontiguousarray(cdelta, np.float64),
                      np.ascontiguousarray(prior, np.float64), np.ascontiguousarray(dprior, np.float64),
                      np.ascontiguousarray(cprior, np.float64)]
        k = self._keep
        ain = abi.BAAccumIn(_p(k[0], _d), _p(k[1], _d), _p(k[2], _f), _p(k[3], _d), _p(k[4], _d), _p(k[5], _d), _p(k[6], _d))
        HA = np.zeros((n, n)); bA = np.zeros(n); HL = np.zeros((n, n)); bL = np.zeros(n); Hsc = np.zeros((n, n)); bsc = np.zeros(n)
        self.ck(self.L.cmlhip_ba_accumulate(self.h, C.byref(ain), _p(HA, _d), _p(bA, _d), _p(HL, _d), _p(bL, _d), _p(Hsc, _d), _p(bsc, _d)))
        return HA, bA, HL, bL, Hsc, bsc

    def _accum_in(self, adH, adT, adHTd, cdelta, prior, dprior, cprior):
        self._keep = [np.ascontiguousarray(adH, np.float64), np.ascontiguousarray(adT, np.float64),
                      np.ascontiguousarray(adHTd, np.float32), np.ascontiguousarray(cdelta, np.float64),
                      np.ascontiguousarray(prior, np.float64), np.ascontiguousarray(dprior, np.float64),
                      np.ascontiguousarray(cprior, np.float64)]
        k = self._keep
        return abi.BAAccumIn(_p(k[0], _d), _p(k[1], _d), _p(k[2], _f), _p(k[3], _d), _p(k[4], _d), _p(k[5], _d), _p(k[6], _d))

    # ---- marginalisation
    def ba_relinearize_points(self, pts, *accum_in):
        ain = self._accum_in(*accum_in)
        pts = np.ascontiguousarray(pts, np.int32)
        ng = _i()
        self.ck(self.L.cmlhip_ba_relinearize_points(self.h, C.byref(ain), len(pts), _p(pts, C.c_int), C.byref(ng)))
        return ng.value

    def ba_marginalize_points(self, pts, *accum_in):
        ain = self._accum_in(*accum_in)
        pts = np.ascontiguousarray(pts, np.int32)
        n = 8 * self.N + 4
        M = np.zeros((n, n)); Mb = np.zeros(n); Msc = np.zeros((n, n)); Mbsc = np.zeros(n)
        self.ck(self.L.cmlhip_ba_marginalize_points(self.h, C.byref(ain), len(pts), _p(pts, C.c_int), _p(M, _d), _p(Mb, _d), _p(Msc, _d), _p(Mbsc, _d)))
        return M, Mb, Msc, Mbsc

    def ba_lin_energy(self, *accum_in):
        ain = self._accum_in(*accum_in)
        e = _d(); k = _i()
        self.ck(self.L.cmlhip_ba_lin_energy(self.h, C.byref(ain), C.byref(e), C.byref(k)))
        return e.value, k.value

    def ba_res_to_zero(self):
        rtz = np.zeros((self.R, 8), np.float32); lin = np.zeros(self.R, np.uint8)
        self.ck(self.L.cmlhip_ba_get_res_to_zero(self.h, _p(rtz, _f), _p(lin, C.c_ubyte)))
        return rtz, lin

    def ba_solve(self, lam, HM=None, bM=None, optcal=0):
        n = 8 * self.N + 4
        x = np.zeros(n)
        HM = None if HM is None else np.ascontiguousarray(HM, np.float64)
        bM = None if bM is None else np.ascontiguousarray(bM, np.float64)
        rc = self.ck(self.L.cmlhip_ba_solve(self.h, lam, _p(HM, _d), _p(bM, _d), optcal, _p(x, _d)), allow=(abi.ERR_NONFINITE,))
        return x, rc

    def ba_backsub(self, x=None):
        step = np.zeros(self.P)
        xx = None if x is None else np.ascontiguousarray(x, np.float64)
        rc = self.ck(self.L.cmlhip_ba_backsub(self.h, _p(xx, _d), _p(step, _d)), allow=(abi.ERR_NONFINITE,))
        return step, rc

    def ba_backup_points(self):
        self.ck(self.L.cmlhip_ba_backup_points(self.h))

    def ba_step_points(self):
        s = np.zeros(3, np.float32)
        self.ck(self.L.cmlhip_ba_step_points(self.h, _p(s, _f)))
        return s

    def ba_states(self):
        R = self.R
        st = np.zeros(R, np.int32); ns = np.zeros(R, np.int32); e = np.zeros(R, np.float32); ne = np.zeros(R, np.float32)
        nw = np.zeros(R, np.float32); g = np.zeros(R, np.uint8)
        self.ck(self.L.cmlhip_ba_get_states(self.h, _p(st, _i), _p(ns, _i), _p(e, _f), _p(ne, _f), _p(nw, _f), _p(g, _u8)))
        return dict(state=st, new_state=ns, energy=e, new_energy=ne, new_energy_wo=nw, good=g)

    def ba_rj(self, which=0):
        out = np.zeros((self.R, abi.RJ_FLOATS), np.float32)
        self.ck(self.L.cmlhip_ba_get_rj(self.h, which, _p(out, _f)))
        return out

    def ba_jpjdf(self):
        out = np.zeros((self.R, 8), np.float32)
        self.ck(self.L.cmlhip_ba_get_jpjdf(self.h, _p(out, _f)))
        return out

    def ba_center(self):
        out = np.zeros((self.R, 3), np.float32)
        self.ck(self.L.cmlhip_ba_get_center_projected(self.h, _p(out, _f)))
        return out

    def ba_pairs(self):
        """(pairs N*N, frame_energy_th N, b0 N) as the device holds them now — no side effect (cmlhip_ba_get_pairs)."""
        pairs = np.zeros(self.N * self.N, abi.BA_PAIR_DTYPE); th = np.zeros(self.N, np.float32); b0 = np.zeros(self.N, np.float32)
        self.ck(self.L.cmlhip_ba_get_pairs(self.h, pairs.ctypes.data_as(C.c_void_p), _p(th, _f), _p(b0, _f)))
        return pairs, th, b0

    def ba_point_acc(self):
        out = np.zeros((self.P, 14), np.float32)
        self.ck(self.L.cmlhip_ba_get_point_acc(self.h, _p(out, _f)))
        return out

    def ba_pair_acc(self, mode=0):
        out = np.zeros((self.N * self.N, 13, 13), np.float32)
        self.ck(self.L.cmlhip_ba_get_pair_acc(self.h, mode, _p(out, _f)))
        return out

    def ba_index_maps(self):
        R, P, NN = self.R, self.P, self.N * self.N
        a = np.zeros(R, np.int32); b = np.zeros(P + 1, np.int32); c = np.zeros(R, np.int32)
        d = np.zeros(NN + 1, np.int32); e = np.zeros(R, np.int32)
        self.ck(self.L.cmlhip_ba_get_index_maps(self.h, _p(a, _i), _p(b, _i), _p(c, _i), _p(d, _i), _p(e, _i)))
        return dict(pair_of=a, by_point_off=b, by_point=c, by_pair_off=d, by_pair=e)

    def ba_linearize_async(self):
        self.ck(self.L.cmlhip_ba_linearize_async(self.h))

    def ba_iteration_async(self, lam):
        self.ck(self.L.cmlhip_ba_iteration_async(self.h, lam))

    def ba_resident_state(self):
        """(frame states [abi.BAFrameState x N], PRE_worldToCam N x 7 (q w,x,y,z | t)) after the iterations enqueued so far; synchronises."""
        fs = (abi.BAFrameState * max(self.N, 1))(); pre = np.zeros((max(self.N, 1), 7))
        self.ck(self.L.cmlhip_ba_get_resident_state(self.h, fs, _p(pre, _d), None))
        return fs, pre[:self.N]

    def ba_resident_indirect(self, n_points=0):
        """(x of the last iteration [8N+4], the last indirect solution [6N], per-point Jacobian sums [M x 3]) of the hybrid term inside the resident loop."""
        x = np.zeros(8 * self.N + 4); x6 = np.zeros(6 * self.N); jp = np.zeros((max(n_points, 1), 3))
        self.ck(self.L.cmlhip_ba_get_resident_indirect(self.h, _p(x, _d), _p(x6, _d), _p(jp, _d) if n_points else None))
        return x, x6, jp[:n_points]

    # ------------------------------------------------------------------ tracker
    def tracker_set_reference(self, level, uvic):
        a = np.ascontiguousarray(uvic, np.float32)
        self.ck(self.L.cmlhip_tracker_set_reference(self.h, level, _p(a, _f), len(a)))

    def tracker_get_reference(self, level):
        n = _i()
        self.ck(self.L.cmlhip_tracker_get_reference(self.h, level, None, C.byref(n)))
        out = np.zeros((n.value, 4), np.float32)
        if n.value:
            self.ck(self.L.cmlhip_tracker_get_reference(self.h, level, _p(out, _f), C.byref(n)))
        return out

    def tracker_make_coarse_depth(self, image_id, levels, pts):
        a = np.ascontiguousarray(pts, np.float64)
        nout = (C.c_int * 8)()
        self.ck(self.L.cmlhip_tracker_make_coarse_depth(self.h, image_id, levels, _p(a, _d), len(a), nout))
        return list(nout[:levels])

    def tracker_eval(self, image_id, level, R, t, K, aff, b0, prm, want_hessian=1):
        R = np.ascontiguousarray(R, np.float64).ravel(); t = np.ascontiguousarray(t, np.float64)
        K = np.ascontiguousarray(K, np.float64); aff = np.ascontiguousarray(aff, np.float64)
        out = abi.TrackerResult()
        rc = self.ck(self.L.cmlhip_tracker_eval(self.h, image_id, level, _p(R, _d), _p(t, _d), _p(K, _d), _p(aff, _d), b0,
                                                C.byref(prm), want_hessian, C.byref(out)), allow=(abi.ERR_NONFINITE,))
        return out, rc

    def tracker_optimize_batch(self, image_id, levels, K0, ref_exp, init_exp, prm, hyps, optimize_a=1, optimize_b=1, sat_th=0.33):
        """hyps: list of (R, t).  Returns a list of abi.TrackerOptResult."""
        n = len(hyps)
        H = (abi.TrackerHypothesis * max(n, 1))()
        for i, (R, t) in enumerate(hyps):
            Rr = np.asarray(R, np.float64).ravel()
            for k in range(9):
                H[i].R[k] = Rr[k]
            for k in range(3):
                H[i].t[k] = float(t[k])
        out = (abi.TrackerOptResult * max(n, 1))()
        K = np.ascontiguousarray(K0, np.float64); re = np.ascontiguousarray(ref_exp, np.float64); ie = np.ascontiguousarray(init_exp, np.float64)
        self.ck(self.L.cmlhip_tracker_optimize_batch(self.h, C.c_uint64(int(image_id)), int(levels), _p(K, _d), _p(re, _d), _p(ie, _d), C.byref(prm),
                                                     int(optimize_a), int(optimize_b), C.c_double(sat_th), n, H, out))
        return [out[i] for i in range(n)]

    def tracker_optimize_batch_async(self, image_id, levels, K0, ref_exp, init_exp, prm, hyps, optimize_a=1, optimize_b=1, sat_th=0.33):
        """cmlhip_tracker_optimize_batch_async: enqueue only; tracker_optimize_wait() hands the results over."""
        n = len(hyps)
        H = (abi.TrackerHypothesis * max(n, 1))()
        for i, (R, t) in enumerate(hyps):
            Rr = np.asarray(R, np.float64).ravel()
            for k in range(9):
                H[i].R[k] = Rr[k]
            for k in range(3):
                H[i].t[k] = float(t[k])
        K = np.ascontiguousarray(K0, np.float64); re = np.ascontiguousarray(ref_exp, np.float64); ie = np.ascontiguousarray(init_exp, np.float64)
        self.ck(self.L.cmlhip_tracker_optimize_batch_async(self.h, C.c_uint64(int(image_id)), int(levels), _p(K, _d), _p(re, _d), _p(ie, _d), C.byref(prm),
                                                           int(optimize_a), int(optimize_b), C.c_double(sat_th), n, H))
        self._trk_pending = n

    def tracker_optimize_wait(self):
        n = self._trk_pending
        out = (abi.TrackerOptResult * max(n, 1))()
        self.ck(self.L.cmlhip_tracker_optimize_wait(self.h, out))
        self._trk_pending = 0
        return [out[i] for i in range(n)]

    @staticmethod
    def _pack_poses(ps):
        a = np.zeros((len(ps), 14))
        for i, (R, t, ea, eb) in enumerate(ps):
            a[i, :9] = np.asarray(R, np.float64).ravel(); a[i, 9:12] = t; a[i, 12] = ea; a[i, 13] = eb
        return a

    def tracer_tracked_prepare(self, host_poses, reference, K):
        """cmlhip_tracer_tracked_prepare, ahead of tracker_optimize_batch_async: the window the tracked trace will pass"""
        hp = self._pack_poses(host_poses); rf = self._pack_poses([reference]); Kd = np.ascontiguousarray(K, np.float64)
        self.ck(self.L.cmlhip_tracer_tracked_prepare(self.h, len(host_poses), hp.ctypes.data, rf.ctypes.data, _p(Kd, _d)))

    def tracer_trace_resident_tracked_async(self, image_id, prm, host_poses, reference, K, skip_host=-2):
        """cmlhip_tracer_trace_resident_tracked_async behind tracker_optimize_batch_async: host_poses / reference = (R, t, a, b) world -> camera"""
        hp = self._pack_poses(host_poses); rf = self._pack_poses([reference]); Kd = np.ascontiguousarray(K, np.float64)
        self._tr_hosts = len(host_poses)
        self.ck(self.L.cmlhip_tracer_trace_resident_tracked_async(self.h, int(image_id), C.byref(prm), len(host_poses), hp.ctypes.data, rf.ctypes.data, _p(Kd, _d), int(skip_host)))

    def tracer_trace_resident_finish(self, keep):
        counts = np.zeros(6, np.int32); pairs = np.zeros(self._tr_hosts, abi.TRACE_PAIR_DTYPE)
        self.ck(self.L.cmlhip_tracer_trace_resident_finish(self.h, 1 if keep else 0, _p(counts, C.c_int), pairs.ctypes.data))
        return counts, pairs

    def tracker_get_warped(self, capacity):
        out = np.zeros((8, capacity), np.float32)
        n = _i()
        self.ck(self.L.cmlhip_tracker_get_warped(self.h, _p(out, _f), capacity, C.byref(n)))
        return out[:, :min(n.value, capacity)], n.value

    # ------------------------------------------------------------------ immature points (DSOTracer)
    def trace_points(self, image_id, prm, pairs, points):
        """points: IMMATURE_POINT_DTYPE array, updated in place and returned."""
        pairs = np.ascontiguousarray(pairs, abi.TRACE_PAIR_DTYPE); points = np.ascontiguousarray(points, abi.IMMATURE_POINT_DTYPE)
        self.ck(self.L.cmlhip_trace_points(self.h, int(image_id), C.byref(prm), len(pairs), pairs.ctypes.data, len(points), points.ctypes.data))
        return points

    def tracer_set_points(self, points):
        points = np.ascontiguousarray(points, abi.IMMATURE_POINT_DTYPE)
        self._tr_n = len(points)
        self.ck(self.L.cmlhip_tracer_set_points(self.h, len(points), points.ctypes.data))

    def tracer_trace_resident(self, image_id, prm, pairs, skip_host):
        pairs = np.ascontiguousarray(pairs, abi.TRACE_PAIR_DTYPE)
        counts = np.zeros(6, np.int32)
        self.ck(self.L.cmlhip_tracer_trace_resident(self.h, int(image_id), C.byref(prm), len(pairs), pairs.ctypes.data, int(skip_host), _p(counts, C.c_int)))
        return counts

    def tracer_edit_points(self, keep, hosts, new_points):
        """cmlhip_tracer_edit_points: kept old slots (with their host index in the current frame list) first, then the new records"""
        k = np.ascontiguousarray(keep, np.int32); h = np.ascontiguousarray(hosts, np.int32)
        npnt = np.ascontiguousarray(new_points, abi.IMMATURE_POINT_DTYPE)
        self.ck(self.L.cmlhip_tracer_edit_points(self.h, len(k), _p(k, _i), _p(h, _i), len(npnt), npnt.ctypes.data))
        self._tr_n = len(k) + len(npnt)

    def tracer_get_state(self):
        out = np.zeros(self._tr_n, abi.IMMATURE_STATE_DTYPE)
        self.ck(self.L.cmlhip_tracer_get_state(self.h, self._tr_n, out.ctypes.data))
        return out

    def tracer_get_points(self):
        out = np.zeros(self._tr_n, abi.IMMATURE_POINT_DTYPE)
        self.ck(self.L.cmlhip_tracer_get_points(self.h, self._tr_n, out.ctypes.data))
        return out

    def optimize_immature_points(self, image_ids, K, pairs, prm, min_obs, points):
        N = len(image_ids)
        ids = np.ascontiguousarray(image_ids, np.uint64); K = np.ascontiguousarray(K, np.float64)
        pairs = np.ascontiguousarray(pairs, abi.ACTIVATION_PAIR_DTYPE); points = np.ascontiguousarray(points, abi.IMMATURE_POINT_DTYPE)
        n = len(points)
        res = np.zeros(n, np.int32); idp = np.zeros(n, np.float32); st = np.zeros((n, N), np.int32)
        self.ck(self.L.cmlhip_optimize_immature_points(self.h, N, _p(ids, C.c_uint64), _p(K, _d), pairs.ctypes.data, C.byref(prm), int(min_obs), n,
                                                        points.ctypes.data, _p(res, C.c_int), _p(idp, _f), _p(st, C.c_int)))
        return res, idp, st

    def optimize_immature_points_resident(self, image_ids, K, pairs, prm, min_obs, slots):
        """cmlhip_optimize_immature_points_resident: the candidates are slots of the device-resident set"""
        N = len(image_ids)
        ids = np.ascontiguousarray(image_ids, np.uint64); K = np.ascontiguousarray(K, np.float64)
        pairs = np.ascontiguousarray(pairs, abi.ACTIVATION_PAIR_DTYPE); sl = np.ascontiguousarray(slots, np.int32)
        n = len(sl)
        res = np.zeros(n, np.int32); idp = np.zeros(n, np.float32); st = np.zeros((n, N), np.int32)
        self.ck(self.L.cmlhip_optimize_immature_points_resident(self.h, N, _p(ids, C.c_uint64), _p(K, _d), pairs.ctypes.data, C.byref(prm), int(min_obs), n,
                                                                 _p(sl, C.c_int), _p(res, C.c_int), _p(idp, _f), _p(st, C.c_int)))
        return res, idp, st

    # ------------------------------------------------------------------ coarse initializer (DSOInitializer::calcResAndGS)
    def initializer_calc_res_and_gs(self, image_id, level, prm, points):
        """points: INIT_POINT_DTYPE array, updated in place.  Returns (H 8x8, b 8, Hsc 8x8, bsc 8, res 3) as float32."""
        assert points.dtype == abi.INIT_POINT_DTYPE and points.flags.c_contiguous
        H = np.zeros((8, 8), np.float32); b = np.zeros(8, np.float32); Hsc = np.zeros((8, 8), np.float32); bsc = np.zeros(8, np.float32)
        res = np.zeros(3, np.float32)
        self.ck(self.L.cmlhip_initializer_calc_res_and_gs(self.h, int(image_id), int(level), C.byref(prm), len(points), points.ctypes.data,
                                                           _p(H, _f), _p(b, _f), _p(Hsc, _f), _p(bsc, _f), _p(res, _f)))
        return H, b, Hsc, bsc, res

    # ------------------------------------------------------------------ ORB side: pose-only optimisation (IndirectCameraOptimizer)
    def pnp_optimize(self, R, t, K, matches, outliers, algorithm=abi.PNP_LEVENBERG, check_outliers=True, compute_covariance=False):
        """matches: PNP_MATCH_DTYPE array; outliers: uint8 array, updated in place.  Returns abi.PnpResult."""
        assert matches.dtype == abi.PNP_MATCH_DTYPE and matches.flags.c_contiguous
        assert outliers.dtype == np.uint8 and len(outliers) == len(matches) and outliers.flags.c_contiguous
        R = np.ascontiguousarray(R, np.float64); t = np.ascontiguousarray(t, np.float64); K = np.ascontiguousarray(K, np.float64)
        out = abi.PnpResult()
        self.ck(self.L.cmlhip_pnp_optimize(self.h, _p(R, _d), _p(t, _d), _p(K, _d), len(matches), matches.ctypes.data, _p(outliers, C.c_ubyte),
                                            int(algorithm), int(bool(check_outliers)), int(bool(compute_covariance)), C.byref(out)))
        return out

    # ------------------------------------------------------------------ ORB side: local bundle adjustment (IndirectBundleAdjustment)
    def lba_set_stop_flag(self, flag):
        """flag: a 1-element uint8 array the caller keeps alive (pbStopFlag), or None."""
        self.ck(self.L.cmlhip_lba_set_stop_flag(self.h, flag.ctypes.data if flag is not None else None))

    def lba_optimize(self, frames, points, point_offsets, edges, fix_frames=True, num_iterations=5, refine_iterations=0):
        """frames (LBA_FRAME_DTYPE) and points (n x 3 float64) are updated in place.  Returns (edge_bad uint8, abi.LbaResult)."""
        assert frames.dtype == abi.LBA_FRAME_DTYPE and edges.dtype == abi.LBA_EDGE_DTYPE and frames.flags.c_contiguous and edges.flags.c_contiguous
        assert points.dtype == np.float64 and points.flags.c_contiguous and points.shape == (len(point_offsets) - 1, 3)
        off = np.ascontiguousarray(point_offsets, np.int32)
        assert off[-1] == len(edges)
        bad = np.zeros(len(edges), np.uint8)
        out = abi.LbaResult()
        self.ck(self.L.cmlhip_lba_optimize(self.h, len(frames), frames.ctypes.data, len(points), _p(points, _d), _p(off, C.c_int), edges.ctypes.data,
                                            int(bool(fix_frames)), int(num_iterations), int(refine_iterations), _p(bad, C.c_ubyte), C.byref(out)))
        return bad, out

    # ------------------------------------------------------------------ reproj
    def reproj_accumulate(self, poses, points, obs, fx, fy):
        poses = np.ascontiguousarray(poses, np.float64); points = np.ascontiguousarray(points, np.float64)
        obs = np.ascontiguousarray(obs)
        N, M, n = len(poses), len(points), len(obs)
        M6 = np.zeros((6 * N, 6 * N)); b6 = np.zeros(6 * N); Jp = np.zeros((M, 3)); used = np.zeros(n, np.uint8)
        self.ck(self.L.cmlhip_reproj_accumulate(self.h, N, _p(poses, _d), M, _p(points, _d), n,
                                                 obs.ctypes.data_as(_P(abi.ReprojObs)), fx, fy, _p(M6, _d), _p(b6, _d), _p(Jp, _d),
                                                 _p(used, _u8)))
        return M6, b6, Jp, used

    def reproj_solve(self, N, lam):
        x = np.zeros(6 * N)
        rc = self.ck(self.L.cmlhip_reproj_solve(self.h, N, lam, _p(x, _d)), allow=(abi.ERR_NONFINITE,))
        return x, rc

    # ------------------------------------------------------------------ timing
    def sync(self):
        self.ck(self.L.cmlhip_synchronize(self.h))

    def mark(self, which):
        self.ck(self.L.cmlhip_event_mark(self.h, which))

    def profile_next_launch(self):
        self.ck(self.L.cmlhip_profile_next_launch(self.h))

    def elapsed_ms(self):
        ms = _f()
        self.ck(self.L.cmlhip_event_elapsed_ms(self.h, C.byref(ms)))
        return ms.value
