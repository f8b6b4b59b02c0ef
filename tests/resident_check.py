"""Replay ONE residual pass of the device-resident loop on the oracle, from exactly the state the device had, and compare bit for bit.

The kernels bench.py times are the resident ones (k_ba_lin_rs4: 4 lanes per residual, small windows; k_ba_lin_rs: a lane per
residual over the tiled fp16 level 0, large windows).  This checker holds THEM against oracle/orc_ba.c directly (VERDICT round 2,
item 1) instead of through the record-writing kernel:

    pre  = residual states / energies before the iteration            (what linearize's OOB early-out and applyRes read, BA.cpp:68-72)
    one cmlhip_ba_iteration_async (K3..K6 step the points and frames, then the residual kernel under test runs)
    post = device pairs (DSOFramePrecomputed as the device frame step wrote them), frameEnergyTH / b0 per frame, inverse depths,
           new_state / new_energy / new_energy_wo / state / energy / good / JpJdF / centerProjectedTo (+ re-materialised records)
    oracle window := same points with the device's inverse depths, residuals with the pre states / energies, the device's pairs and
           thresholds, the same level-0 texels -> orc_ba_linearize_all + orc_ba_apply(1)                       (BA.cpp:62-316, 2051-2093)

Everything per residual must be IDENTICAL IN EVERY BIT.  Checker side only (tests/, bench.py's parity gate)."""
import ctypes as C

import numpy as np

from libcml_amd import abi
from tests import oracle_lib as O


def _u32(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


class ResidentReplay:
    """Oracle window over (prm, frames_dev, images, points, residuals) whose mutable state is overwritten from the device before every replay."""

    def __init__(self, prm, frames_dev, images, points, residuals):
        self.N, self.P, self.R = len(frames_dev), len(points), len(residuals)
        self._prm = prm
        self._frames = np.ascontiguousarray(frames_dev, abi.BA_FRAME_DTYPE)
        self._imgs = [np.ascontiguousarray(im, np.float32) for im in images]
        self._points = np.ascontiguousarray(points, abi.BA_POINT_DTYPE).copy()
        self._res = np.ascontiguousarray(residuals, abi.BA_RESIDUAL_DTYPE).copy()
        arr = (C.POINTER(C.c_float) * self.N)(*[O.ptr(im, C.c_float) for im in self._imgs])
        self.w = O.lib().orc_ba_create(C.byref(prm), self.N, self._frames.ctypes.data_as(C.POINTER(abi.BAFrame)), arr, self.P,
                                       self._points.ctypes.data_as(C.POINTER(abi.BAPoint)), self.R,
                                       self._res.ctypes.data_as(C.POINTER(abi.BAResidual)))

    def close(self):
        if self.w:
            O.lib().orc_ba_destroy(self.w); self.w = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _view(self, name, n):
        return np.ctypeslib.as_array(getattr(self.w.contents, name), shape=(n,))

    def replay(self, pre, pairs, frame_energy_th, idepth):
        """Oracle pass from the given state.  Returns the oracle's per-residual outputs."""
        w = self.w.contents
        R = self.R
        self._view("r_state", R)[:] = pre["state"]
        self._view("r_energy", R)[:] = pre["energy"]
        self._view("r_new_state", R)[:] = pre["new_state"]
        self._view("r_new_energy", R)[:] = pre["new_energy"]
        self._view("r_new_energy_wo", R)[:] = pre["new_energy_wo"]
        self._view("r_good", R)[:] = pre["good"]
        for k in range(self.N):
            w.frame_energy_th[k] = float(frame_energy_th[k])
        pts = np.ctypeslib.as_array(C.cast(w.points, C.POINTER(C.c_ubyte)), shape=(self.P * abi.BA_POINT_DTYPE.itemsize,)).view(abi.BA_POINT_DTYPE)
        pts["idepth"] = idepth
        self._pairs = np.ascontiguousarray(pairs, abi.BA_PAIR_DTYPE)
        O.lib().orc_ba_set_pairs(self.w, self._pairs.ctypes.data_as(C.POINTER(abi.BAPair)))
        out = abi.BALinResult()
        O.lib().orc_ba_linearize_all(self.w, C.byref(out))
        rj = self._view("rJ", R * 74).reshape(-1, 74).copy()          # the pass's records, before applyRes swaps them into efsJ
        center = self._view("r_center", 3 * R).reshape(-1, 3).copy()
        new_state = self._view("r_new_state", R).copy()
        O.lib().orc_ba_apply(self.w, 1)
        return dict(new_state=new_state, state=self._view("r_state", R).copy(), energy=self._view("r_energy", R).copy(),
                    new_energy=self._view("r_new_energy", R).copy(), new_energy_wo=self._view("r_new_energy_wo", R).copy(),
                    good=self._view("r_good", R).copy(), jpjdf=self._view("JpJdF", 8 * R).reshape(-1, 8).copy(), center=center,
                    efsj=rj, lin=out)


def make_replay(ctx, ba, W):
    """Oracle replay window for a synthetic window registered through libcml_amd.host.window_to_host_ba (the bench.py set-up): same
    points (colours / weights as registered), the residual list in the mirror's order (= synth.residual_list), the device's level-0
    texels read back through the ABI (for fp16 contexts: the stored halves widened), thresholds / b0 as uploaded."""
    from libcml_amd import synth
    si = ba.synth_inputs
    pts = np.zeros(W.P, abi.BA_POINT_DTYPE)
    pts["x"] = W.pts["x"]; pts["y"] = W.pts["y"]; pts["idepth"] = W.pts["idepth"]; pts["idepth_zero"] = W.pts["idepth"].astype(np.float32)
    pts["colors"] = si["colors"]; pts["weights"] = si["weights"]; pts["host"] = W.pts["host"]
    res = synth.residual_list(W, W.R_eval, W.t_eval)
    rs = np.zeros(len(res), abi.BA_RESIDUAL_DTYPE)
    for f in ("point", "target", "state", "is_linearized"):
        rs[f] = res[f]
    _, th, b0 = ctx.ba_pairs()
    fr = np.zeros(W.N, abi.BA_FRAME_DTYPE)
    fr["frame_energy_th"] = th; fr["b0"] = b0
    prm = abi.default_ba_params(*W.K, W.w, W.h)
    return ResidentReplay(prm, fr, si["grads0"], pts, rs)


def check_one_pass(ctx, replay, lam=1e-5, with_records=False):
    """Enqueue one resident iteration on `ctx`, replay its residual pass on the oracle, compare.  Returns a report dict; report["ok"] is
    True when every compared value is bit-identical.  The device state advances by one iteration (as in any other step)."""
    ctx.sync()
    pre = ctx.ba_states()
    ctx.ba_iteration_async(lam)
    ctx.sync()
    return compare_pass(ctx, replay, pre, with_records)


def check_one_batched_pass(ctxs, replays, lam=1e-5, with_records=False):
    """The same for S windows stepped by ONE cmlhip_ba_iteration_batch: a report per window."""
    from libcml_amd import device
    ctxs[0].sync()
    pres = [c.ba_states() for c in ctxs]
    device.ba_iteration_batch(ctxs, lam)
    ctxs[0].sync()
    return [compare_pass(c, r, p, with_records) for c, r, p in zip(ctxs, replays, pres)]


def compare_pass(ctx, replay, pre, with_records=False):
    """Replay the residual pass the device just ran (from `pre` = the residual states before it and the device's current pairs /
    thresholds / inverse depths) and compare every per-residual output bit for bit."""
    pairs, th, _b0 = ctx.ba_pairs()
    idepth = ctx.ba_get_idepth()
    post = ctx.ba_states()
    jp = ctx.ba_jpjdf(); ce = ctx.ba_center()
    o = replay.replay(pre, pairs, th, idepth)
    rep = {"R": int(replay.R)}
    rep["new_state_mismatch"] = int((o["new_state"] != post["new_state"]).sum())
    rep["state_mismatch"] = int((o["state"] != post["state"]).sum())
    rep["good_mismatch"] = int((o["good"] != post["good"]).sum())
    # CMLHIP_RESIDENT_OUTPUTS_LEAN (what the host mirror's run() sets, include/cmlhip.h): centerProjectedTo is not stored and
    # state_NewEnergyWithOutlier only for the residuals into the newest frame (setNewFrameEnergyTH's input) — those are then the compared ones
    lean = ctx.ba_resident_outputs_lean()
    rep["lean_outputs"] = bool(lean)
    for k in ("energy", "new_energy"):
        rep[k + "_mismatch"] = int((_u32(o[k]) != _u32(post[k])).sum())
    wo = (replay._res["target"] == replay.N - 1) if lean else np.ones(replay.R, bool)
    rep["new_energy_wo_mismatch"] = int((_u32(o["new_energy_wo"])[wo] != _u32(post["new_energy_wo"])[wo]).sum())
    g = o["good"] == 1
    IN = o["new_state"] == 0
    rep["n_good"] = int(g.sum()); rep["n_in"] = int(IN.sum())
    rep["n_sampled"] = int((pre["state"] != 1).sum())          # residuals that enter the pixel loop at all (not absorbed as OOB, BA.cpp:68-72)
    rep["jpjdf_mismatch"] = int((_u32(o["jpjdf"])[g] != _u32(jp)[g]).any(axis=1).sum())
    rep["center_mismatch"] = 0 if lean else int((_u32(o["center"])[IN] != _u32(ce)[IN]).any(axis=1).sum())
    if with_records:                                            # 74-float records the resident kernel never wrote, re-created on demand
        rj = ctx.ba_rj(1)
        rep["record_mismatch"] = int((_u32(o["efsj"])[g] != _u32(rj)[g]).any(axis=1).sum())
    rep["ok"] = all(v == 0 for k, v in rep.items() if k.endswith("_mismatch"))
    return rep


def compare_pass_tolerant(ctx, replay, pre):
    """The same replay for a pass computed in CMLHIP_ARITH_RELAXED: how far the per-residual outputs are from the oracle's, and how many
    residuals were classified differently (tests/test_relaxed_arithmetic_gpu.py states the bars)."""
    pairs, th, _ = ctx.ba_pairs()
    post = ctx.ba_states(); jp = ctx.ba_jpjdf(); ce = ctx.ba_center()
    o = replay.replay(pre, pairs, th, ctx.ba_get_idepth())
    rep = {"R": int(replay.R)}
    for k in ("new_state", "state", "good"):
        rep[k + "_flips"] = int((o[k] != post[k]).sum())
    same = (o["new_state"] == post["new_state"]) & (o["state"] == post["state"]) & (o["good"] == post["good"])
    m = same & (pre["state"] != 1)
    lean = ctx.ba_resident_outputs_lean()                     # (see compare_pass)
    for k in ("energy", "new_energy", "new_energy_wo"):
        mk = m & (replay._res["target"] == replay.N - 1) if (lean and k == "new_energy_wo") else m
        a, b = o[k][mk].astype(np.float64), post[k][mk].astype(np.float64)
        rep[k + "_rel"] = float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1.0))) if mk.any() else 0.0      # (relative, energies below 1 — 0.35 grey levels per pattern pixel — absolute)
    g = m & (o["good"] == 1)
    a, b = o["jpjdf"][g].astype(np.float64), jp[g].astype(np.float64)
    # per row, against the row's largest entry.  The entries are J^T J d products whose two terms can cancel (g = JIdx2 * Jpdd): such rows
    # amplify the rounding of the fp32 pattern sums — in the exact mode against real arithmetic just as in the relaxed mode against the exact
    # one — so the statement is a distribution: median, 99.9th percentile, worst row
    rel = np.abs(a - b).max(axis=1) / np.maximum(np.abs(a).max(axis=1), 1e-6) if g.any() else np.zeros(1)
    rep["jpjdf_rel"] = float(rel.max()); rep["jpjdf_rel_p999"] = float(np.percentile(rel, 99.9)); rep["jpjdf_rel_median"] = float(np.median(rel))
    IN = m & (o["new_state"] == 0)
    rep["center_abs"] = float(np.abs(o["center"][IN].astype(np.float64) - ce[IN]).max()) if (IN.any() and not lean) else 0.0
    rep["n_in"] = int(IN.sum())
    return rep
