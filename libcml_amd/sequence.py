"""A sequence shard, end to end, through the host mirror — the unit north_star fans out across GPUs.

Order of calls = the reference's direct pipeline:
  per frame     Hybrid::trackWithDso (slam/modslam/Hybrid.cpp:431-458): DSOTracker::trackWithMotionModel (DSOTracker.h:238-383)
  non-keyframe  Hybrid::directMakeNonKeyFrame (direct/Mapping.cpp:43-45): DSOTracer::traceNewCoarse
  keyframe      Hybrid::directMap (direct/Mapping.cpp:47-134): traceNewCoarse -> addNewFrame (which flags frames for marginalisation first,
                BA.cpp:428) -> activatePoints -> addPoints -> run -> makeCoarseDepthL0(getGoodPointsForTracking) -> tryMarginalize ->
                [outliers dropped] -> marginalizePointsF -> makeNewTraces -> marginalizeFrames -> free / makeUnactive of the marginalised
                frames (here: cmlhip_pyramid_drop, the image id goes back to the pool and is handed to a later frame).

What stays outside (SURVEY §2 OUT OF SCOPE, stood in for by seeded synthetic inputs): the map graph, the PixelSelector (best gradient
of 12 random candidates per pick), the DistanceMap spacing of activatePoints (none), the coarse initializer's bootstrap (frame 0 is
registered with noisy true inverse depths, hasDepthPrior as the initializer's points have), the motion-model hypothesis list (a short
constant-velocity list).  Product-side plumbing for tests and bench.py: it talks to the C++ host mirror (libcmlhost.so) and the C ABI
only — no oracle.  A checker can subscribe to every stage (`observer`) and is handed that stage's inputs and outputs."""
import os
import time

import numpy as np

from . import abi, host, synth

STAR8 = synth.STAR8


class Sequence:
    pass


def select_pixels(gray, n, rng, margin=10, taken=None):
    """stand-in for Features::PixelSelector (out of scope): n distinct integer pixels, each the best-gradient one of 12 random candidates
    (drawn in rounds of 2n picks; a pick is kept when its gradient clears a floor and its pixel is still free)"""
    h, w = gray.shape
    out = []
    seen = set() if taken is None else taken
    for _round in range(20):
        m = 2 * n
        cx = rng.integers(margin, w - margin, size=(m, 12)); cy = rng.integers(margin, h - margin, size=(m, 12))
        mag = np.abs(gray[cy, cx + 1] - gray[cy, cx - 1]) + np.abs(gray[cy + 1, cx] - gray[cy - 1, cx])
        b = np.argmax(mag, axis=1); rows = np.arange(m)
        bx, by, bm = cx[rows, b], cy[rows, b], mag[rows, b]
        for x, y, g in zip(bx.tolist(), by.tolist(), bm.tolist()):
            if g < 4.0 or (x, y) in seen:
                continue
            seen.add((x, y)); out.append((x, y))
            if len(out) == n:
                return np.array(out, np.int32).reshape(-1, 2)
    return np.array(out, np.int32).reshape(-1, 2)


def make_sequence(n_frames=48, seed=0x5EED, config="B", shard=0, kf_gap=(3, 5), n_bootstrap=1500):
    """Seeded synthetic sequence of the BASELINE image shape: one textured plane (the scene of synth.make_window's config B: band-limited
    edges, value noise with fine octaves) seen from a camera that translates mostly sideways (≈ 12 px of parallax per frame at the plane
    distance) with a little forward motion and smooth rotation / exposure drift.  Keyframes every kf_gap[0]..kf_gap[1] frames (seeded)."""
    _N, _P, w, h, levels, fx, fy, cx, cy = synth.CONFIGS[config] if isinstance(config, str) else config
    rng = np.random.default_rng(seed + 7919 * shard)
    S = Sequence()
    S.w, S.h, S.levels, S.K, S.n_frames = w, h, levels, (fx, fy, cx, cy), n_frames
    n = np.array([0.12, -0.08, 1.0]); n /= np.linalg.norm(n)
    d = 9.0
    tex = synth.Texture(rng, scale=0.07 * fx / (d * 32.0), edge_width=4.0 * d / fx, octave_gain=0.7)
    ph = rng.uniform(0, 2 * np.pi, 6)
    S.R_true, S.t_true, S.aff_true = [], [], []
    for k in range(n_frames):
        c = np.array([0.15 * k + 0.02 * np.sin(0.31 * k + ph[0]), 0.03 * np.sin(0.23 * k + ph[1]), 0.04 * k + 0.02 * np.sin(0.17 * k + ph[2])])
        wv = np.deg2rad(np.array([0.4 * np.sin(0.19 * k + ph[3]), 0.6 * np.sin(0.13 * k + ph[4]), 0.3 * np.sin(0.29 * k + ph[5])]))
        Rk = synth.so3_exp(wv)
        S.R_true.append(Rk); S.t_true.append(-Rk @ c)
        S.aff_true.append((0.03 * np.sin(0.11 * k + ph[0]), 3.0 * np.sin(0.07 * k + ph[1])))

    def _render(k):
        img, s = synth.render(tex, S.K, S.R_true[k], S.t_true[k], w, h, n, d)
        a, b = S.aff_true[k]
        return (np.exp(a) * img + b).astype(np.float32), (s if k == 0 else None)
    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        out = list(ex.map(_render, range(n_frames)))
    S.gray = [o[0] for o in out]
    depth0 = out[0][1]
    # keyframe schedule: frame 0 (bootstrap), then every 3-5 frames
    S.keyframes = [0]
    k = 0
    while True:
        k += int(rng.integers(kf_gap[0], kf_gap[1] + 1))
        if k >= n_frames:
            break
        S.keyframes.append(k)
    # bootstrap points of frame 0 (what the coarse initializer hands over): pixels + inverse depths accurate to 2 %
    px = select_pixels(S.gray[0], n_bootstrap, rng)
    idp = 1.0 / depth0[px[:, 1], px[:, 0]]
    S.boot_px = px
    S.boot_idepth = idp * (1 + rng.normal(0, 0.02, len(px)))
    S.seed = seed
    return S


def _patches(grad0, px):
    """MapPoint::getGrayPatch / getDerivativePatch at the pattern pixels (integer-pixel lookups, MapObject.h:392-412) and gradH
    (DSOTracer.cpp:516-521) of n integer pixels at once: gray (n, 8), dpatch (n, 24), gradH (n, 4)"""
    px = np.asarray(px, np.int64).reshape(-1, 2)
    xs = px[:, 0:1] + STAR8[None, :, 0]; ys = px[:, 1:2] + STAR8[None, :, 1]
    t = grad0[ys, xs]                                                 # (n, 8, 3)
    gray = np.ascontiguousarray(t[:, :, 0], np.float32)
    dp = np.ascontiguousarray(t.reshape(len(px), 24), np.float32)
    g = t[:, :, 1:3].astype(np.float64)
    G = np.zeros((len(px), 4))
    for k in range(8):                                                # summed in pattern order, like the reference's loop
        G[:, 0] += g[:, k, 0] * g[:, k, 0]; G[:, 1] += g[:, k, 0] * g[:, k, 1]; G[:, 2] += g[:, k, 1] * g[:, k, 0]; G[:, 3] += g[:, k, 1] * g[:, k, 1]
    return gray, dp, G


def _weights(dpatch):
    """BA::addPoints gradient weights sqrt(c / (c + |grad|^2)), c = 50^2 (BA.cpp:405-411), (n, 24) -> (n, 8)"""
    d = np.asarray(dpatch, np.float32).reshape(-1, 8, 3).astype(np.float64)
    return np.sqrt(2500.0 / (2500.0 + (d[:, :, 1] * d[:, :, 1] + d[:, :, 2] * d[:, :, 2]))).astype(np.float32)


def _rel(Rh, th, Rt, tt):
    """host -> target: Camera::to (src/cml/map/Camera.h)"""
    R = Rt @ Rh.T
    return R, tt - R @ th


class DirectPipeline:
    """Drives cml_amd::DSOTracker / DSOTracer / DSOBundleAdjustment (host mirror over the C ABI) in the reference's order on ONE context.
    `observer(stage, info)` — when given — is called after every stage with that stage's inputs and outputs (a checker replays them)."""

    def __init__(self, ctx, K, w, h, levels, n_immature=500, seed=1, observer=None, max_frames=6, id_pool=16, mapper_only=False):
        self.ctx, self.K, self.w, self.h, self.levels = ctx, tuple(K), w, h, levels
        self.ba = host.HostBA(ctx); self.ba.set_calibration(*K, w, h)
        self.ba.set_param("disableMarginalization", 0)               # the marginalisation prior is live (BA.cpp:1389-1401)
        self.ba.set_param("maxFrames", max_frames)
        # mapper_only: the mapping half of the shard alone (SplitPipeline: Hybrid::directMappingLoop on its own context / thread) — frames arrive already
        # tracked (`injected`), and what the tracker needs from a keyframe's mapping (reference lists, optimised pose) is left in `handover`
        self.mapper_only = mapper_only
        self.injected = None
        self.handover = None
        self.trk = None
        if not mapper_only:
            self.trk = host.HostTracker(ctx); self.trk.set_calibration(*K)
        self.trc = host.HostTracer(ctx)
        self.rng = np.random.default_rng(seed)
        self.n_immature = n_immature
        self.obs = observer
        self.free_ids = list(range(1, id_pool + 1))                  # image ids: taken smallest first, recycled through cmlhip_pyramid_drop
        self.kfs = []                                                # window keyframes in DSOFrame::id order: dict(fid, image_id, gray, grad0, taken)
        self.history = []                                            # world->cam (R, t) of the tracked frames (motion model)
        self.last_exposure = (0.0, 0.0)
        self.last_coarse_rmse = 100.0                                # DSOTracker.h:470
        self.n_fid = 0
        self.times = {}                                              # stage -> list of seconds
        self._poses = None
        self._prefetched = None                                      # (image id, gray) of the next frame, handed to the image worker
        self.prefetch = True
        # one enqueue and one host wait per tracked frame (cmlhost_frame_track_and_trace: the immature points are traced behind the tracker batch, on the
        # first hypothesis' result); False restores the two calls (trackWithMotionModel, then traceNewCoarse with pairs formed here)
        self.fused = (not mapper_only) and not os.environ.get("CML_SEQ_UNFUSED")
        self.lib_times = {}                                          # stage -> seconds inside the library's calls of that stage (see _c)
        self.run_split = []                                          # per keyframe: the host clock of run()'s phases (HostBA.run_timing)
        self.tprm = abi.default_tracer_params()
        self.stats = {"frames": 0, "keyframes": 0, "tracking_lost": 0, "ids_recycled": 0, "max_window": 0, "marginalized_frames": 0}

    def close(self):
        self.trc.close()
        if self.trk is not None:
            self.trk.close()
        self.ba.close()

    # ------------------------------------------------------------------ helpers
    def _t(self, stage, t0):
        self.ctx.sync()
        self.times.setdefault(stage, []).append(time.perf_counter() - t0)

    def _c(self, stage, fn, *a, **kw):
        """a call INTO the library (C++ host mirror / C ABI), timed on its own with a device sync behind it: `lib_times` separates the product from
        this Python driver, which stands in for the reference's own host code around these calls (map accessors, motion model, pixel selector)"""
        owner = getattr(fn, "__self__", None)
        if owner is not None and hasattr(owner, "last_call_s"):
            owner.last_call_s = None
        t0 = time.perf_counter()
        r = fn(*a, **kw)
        t1 = time.perf_counter()
        self.ctx.sync()
        t2 = time.perf_counter()
        inner = getattr(owner, "last_call_s", None) if owner is not None else None      # wrappers that pack arrays in Python report the C call's own time
        self.lib_times.setdefault(stage, []).append((inner if inner is not None else t1 - t0) + (t2 - t1))
        return r

    def _emit(self, stage, **info):
        if self.obs is not None:
            self.obs(stage, info)

    def _take_id(self):
        if not self.free_ids:
            raise RuntimeError("image id pool exhausted")
        return self.free_ids.pop(0)

    def _drop(self, image_id):
        self.ctx.pyramid_drop(image_id)
        self.free_ids.append(image_id); self.free_ids.sort()
        self.stats["ids_recycled"] += 1

    def kf_poses(self):
        """current (R, t, a, b) of the window's keyframes = frame->getCamera() / getExposure() (PRE_worldToCam, aff_g2l)"""
        if self._poses is not None and len(self._poses) == len(self.kfs):     # (they move in run() and when frames join / leave: invalidated there)
            return self._poses
        out = []
        for i in range(len(self.kfs)):
            f = self.ba.frame(i)
            out.append((f["R"].copy(), f["t"].copy(), float(f["ab"][0]), float(f["ab"][1])))
        self._poses = out
        return out

    def _trace_pairs(self, poses, Rn, tn, an, bn):
        """host h -> traced frame, every keyframe at once: K R K^-1, K t, affine (Exposure::to with exposure times 1, Exposure.h:119-123)"""
        fx, fy, cx, cy = self.K
        Km = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]]); Ki = np.linalg.inv(Km)
        Rh = np.stack([p[0] for p in poses]); th = np.stack([p[1] for p in poses])
        ah = np.array([p[2] for p in poses]); bh = np.array([p[3] for p in poses])
        R = np.einsum("ij,hkj->hik", Rn, Rh)                          # Rn Rh^T (Camera::to)
        t = tn[None, :] - np.einsum("hij,hj->hi", R, th)
        pr = np.zeros(len(poses), abi.TRACE_PAIR_DTYPE)
        pr["KRKi"] = np.einsum("ij,hjk,kl->hil", Km, R, Ki).reshape(len(poses), 9)
        pr["Kt"] = np.einsum("ij,hj->hi", Km, t)
        a = np.exp(an - ah)
        pr["aff_a"] = a; pr["aff_b"] = bn - a * bh
        return pr

    def _activation_pairs(self, poses):
        """host h -> target t for every ordered pair of the window (row h * N + t)"""
        N = len(poses)
        Rw = np.stack([p[0] for p in poses]); tw = np.stack([p[1] for p in poses])
        aw = np.array([p[2] for p in poses]); bw = np.array([p[3] for p in poses])
        R = np.einsum("tij,hkj->htik", Rw, Rw)                        # Rt Rh^T
        t = tw[None, :, :] - np.einsum("htij,hj->hti", R, tw)
        a = np.exp(aw[None, :] - aw[:, None])
        pr = np.zeros(N * N, abi.ACTIVATION_PAIR_DTYPE)
        pr["R"] = R.reshape(N * N, 9); pr["t"] = t.reshape(N * N, 3); pr["aff_a"] = a.reshape(-1); pr["aff_b"] = (bw[None, :] - a * bw[:, None]).reshape(-1)
        return pr

    def _immature_counts(self):
        return self.trc.immature_counts([kf["fid"] for kf in self.kfs])

    def _make_new_traces(self, kf):
        """DSOTracer::makeNewTraces (DSOTracer.cpp:496-541) with the stand-in pixel selector"""
        self._c("makeNewTraces", self.trc.compact)                      # activated / removed points leave the list (removeMapPoint): indices change here only
        px = select_pixels(kf["gray"], self.n_immature, self.rng, taken=kf["taken"])
        g, dp, G = _patches(kf["grad0"], px)
        self._c("makeNewTraces", self.trc.add_points, px.astype(np.float32), kf["fid"], g, dp, G)
        return px

    def _coarse_depth(self, kf_index):
        """host half of makeCoarseDepthL0 (DSOTracker.cpp:521-553) over getGoodPointsForTracking (BA.h:76-85): in the host mirror (capi.cpp)"""
        return self.ba.coarse_depth_points(kf_index, self.K)

    # ------------------------------------------------------------------ stages
    def bootstrap(self, gray, R, t, px, idepth):
        """Frame 0 as the first keyframe with the points an initializer would hand over (hasDepthPrior, BA.cpp:1182)."""
        t0 = time.perf_counter()
        iid = self._take_id()
        self.ctx.pyramid_build(iid, gray, self.levels)
        grad0 = self.ctx.pyramid_get(iid, 0)
        kf = {"fid": self.n_fid, "image_id": iid, "gray": gray, "grad0": grad0, "taken": set()}
        self.n_fid += 1
        self.ba.add_frame(iid, R, t, 0.0, 0.0, 1.0)
        self.kfs.append(kf)
        kf["taken"].update((int(x), int(y)) for x, y in px)
        g, dp, _G = _patches(grad0, px)
        self.ba.add_points(px.astype(np.float32), idepth, np.zeros(len(px), np.int32), g, _weights(dp), prior=True)
        # tracking reference lists straight from the bootstrap points (uniform weights: no Hessian yet)
        pts = np.stack([px[:, 0].astype(np.float64), px[:, 1].astype(np.float64), np.asarray(idepth, np.float64), np.ones(len(px))], 1)
        if self.mapper_only:
            nout = None
            self.handover = {"cd": pts, "R": np.asarray(R, float).copy(), "t": np.asarray(t, float).copy(), "ab": (0.0, 0.0)}
        else:
            nout = self.trk.make_coarse_depth(iid, self.levels, pts)
        self._make_new_traces(kf)
        self.ref = 0
        self.history.append((np.asarray(R, float).copy(), np.asarray(t, float).copy()))
        self.last_exposure = (0.0, 0.0)
        self.stats["frames"] += 1; self.stats["keyframes"] += 1
        self._t("bootstrap", t0)
        if not self.mapper_only:                                      # (the reference lists are the tracker front's stage there)
            self._emit("bootstrap", image_id=iid, gray=gray, pts=pts, n_lists=nout)
        return nout

    def _hypotheses(self):
        """a short Map::multiConstantVelocityMotionModel stand-in (world->cam candidates): constant velocity, no motion, half, double, and the
        constant-velocity pose with six small extra rotations"""
        R1, t1 = self.history[-1]
        if len(self.history) < 2:
            return [(R1, t1)]
        R0, t0 = self.history[-2]
        dR = R1 @ R0.T; dt = t1 - dR @ t0                           # last motion: prev -> last (world->cam composition)
        w = _so3_log(dR)
        out = [(dR @ R1, dR @ t1 + dt), (R1, t1)]
        Rh = synth.so3_exp(0.5 * w); out.append((Rh @ R1, Rh @ t1 + 0.5 * dt))
        R2 = dR @ dR; out.append((R2 @ R1, R2 @ t1 + (dR @ dt + dt)))
        Rc, tc = out[0]
        for ax in range(3):
            for sg in (1.0, -1.0):
                e = np.zeros(3); e[ax] = sg * 0.01
                Re = synth.so3_exp(e)
                out.append((Re @ Rc, Re @ tc))
        return out

    def track(self, gray, next_gray=None):
        """pyramid of the new frame + trackWithMotionModel against the newest keyframe.  Returns (image_id, R, t, a, b, ok).
        next_gray: the frame after this one, when the reader already has it — its pyramid is handed to the context's image worker
        (cmlhip_pyramid_build_async) BEFORE this frame is tracked, as the reference's capture thread builds pyramids ahead of the SLAM thread
        (capture/CaptureImage.cpp): staging copy, transfer and level kernels then run beside the tracker instead of in front of the next one."""
        if self.mapper_only:                                          # tracked elsewhere (SplitPipeline's tracker front): only the frame's pyramid is built here
            t0 = time.perf_counter()
            iid = self._take_id()
            self._c("pyramid_build", self.ctx.pyramid_build, iid, gray, self.levels)
            self._t("pyramid_build", t0)
            Rn, tn, a, b, ok = self.injected
            self.history.append((Rn.copy(), tn.copy())); self.last_exposure = (a, b)
            self.stats["frames"] += 1
            if not ok:
                self.stats["tracking_lost"] += 1
            return iid, Rn, tn, a, b, ok
        t0 = time.perf_counter()
        if self._prefetched is not None and self._prefetched[1] is gray:
            iid = self._prefetched[0]                                 # built (or being built) by the image worker: the first call that names it waits on the device
        else:
            if self._prefetched is not None:                          # a prefetched image that is not this frame (the caller passed another array): its id and
                self._drop(self._prefetched[0])                       # device pyramid go back to the pool instead of leaking
            iid = self._take_id()
            self._c("pyramid_build", self.ctx.pyramid_build, iid, gray, self.levels)
        self._prefetched = None
        if next_gray is not None and self.prefetch:
            nid = self._take_id()
            self._c("pyramid_build", self.ctx.pyramid_build_async, nid, next_gray, self.levels)
            self._prefetched = (nid, next_gray)
        self._t("pyramid_build", t0)
        t0 = time.perf_counter()
        ref = self.kfs[self.ref]
        Rr, tr, ar, br = self.kf_poses()[self.ref]
        hyps_w = self._hypotheses()
        hyps = [_rel(Rr, tr, Rw, tw) for Rw, tw in hyps_w]           # reference->getCamera().to(camera)
        ref_exp = [ar, br, 1.0]; init_exp = [self.last_exposure[0], self.last_exposure[1], 1.0]
        res = self._c("trackWithMotionModel", self.trk.track_with_motion_model, iid, self.levels, hyps, ref_exp, init_exp, batched=True)
        self._t("trackWithMotionModel", t0)
        ok = bool(res["haveOneGood"])
        if ok:
            Rn = res["R"] @ Rr; tn = res["R"] @ tr + res["t"]          # frame->setCamera(reference.compose(refToNew))
            a, b = float(res["exposure"][0]), float(res["exposure"][1])
        else:                                                         # tracking lost: keep the constant-velocity guess (the reference would relocalise)
            Rn, tn = hyps_w[0]; a, b = self.last_exposure
            self.stats["tracking_lost"] += 1
        self._emit("track", image_id=iid, gray=gray, ref_image_id=ref["image_id"], hyps=hyps, ref_exp=ref_exp, init_exp=init_exp, result=res,
                   last_coarse_rmse=self.last_coarse_rmse, levels=self.levels)
        if ok:
            self.last_coarse_rmse = float(res["lastCoarseRMSE"])
        self.history.append((Rn.copy(), tn.copy())); self.last_exposure = (a, b)
        self.stats["frames"] += 1
        return iid, Rn, tn, a, b, ok

    def trace(self, iid, Rn, tn, a, b, traced_fid):
        """DSOTracer::traceNewCoarse(frame, ACTIVEKEYFRAME)"""
        t0 = time.perf_counter()
        poses = self.kf_poses()
        pairs = self._trace_pairs(poses, Rn, tn, a, b)
        fids = [kf["fid"] for kf in self.kfs]
        before = self.trc.points() if self.obs is not None else None
        counts = self._c("traceNewCoarse", self.trc.trace_new_coarse, iid, traced_fid, fids, pairs)
        self._t("traceNewCoarse", t0)
        if self.obs is not None:
            self._emit("trace", image_id=iid, pairs=pairs, frame_ids=fids, traced_fid=traced_fid, before=before, after=self.trc.points(), counts=counts,
                       tracer_fids=[int(f) for f in self.trc.frame_ids()])
        return counts

    def track_and_trace(self, gray, next_gray, traced_fid):
        """the frame's two stages.  Fused (default): cmlhost_frame_track_and_trace — the hypothesis batch and, behind it on the stream, traceNewCoarse against the
        first hypothesis' result: ONE host wait; when the replayed selection adopts another try (or tracking fails) the trace was rolled back by the library and
        runs again here with the pose the driver goes on with.  Returns what track() returns."""
        if not self.fused:
            r = self.track(gray, next_gray)
            self.trace(r[0], r[1], r[2], r[3], r[4], traced_fid)
            return r
        t0 = time.perf_counter()
        if self._prefetched is not None and self._prefetched[1] is gray:
            iid = self._prefetched[0]
        else:
            if self._prefetched is not None:
                self._drop(self._prefetched[0])
            iid = self._take_id()
            self._c("pyramid_build", self.ctx.pyramid_build, iid, gray, self.levels)
        self._prefetched = None
        if next_gray is not None and self.prefetch:
            nid = self._take_id()
            self._c("pyramid_build", self.ctx.pyramid_build_async, nid, next_gray, self.levels)
            self._prefetched = (nid, next_gray)
        self._t("pyramid_build", t0)
        t0 = time.perf_counter()
        ref = self.kfs[self.ref]
        poses = self.kf_poses()
        Rr, tr, ar, br = poses[self.ref]
        hyps_w = self._hypotheses()
        hyps = [_rel(Rr, tr, Rw, tw) for Rw, tw in hyps_w]
        ref_exp = [ar, br, 1.0]; init_exp = [self.last_exposure[0], self.last_exposure[1], 1.0]
        fids = [kf["fid"] for kf in self.kfs]
        before = self.trc.points() if self.obs is not None else None
        res, kept, counts, pairs = self._c("trackAndTrace", self.trk.track_and_trace, self.trc, iid, self.levels, hyps, ref_exp, init_exp, traced_fid, fids, poses,
                                           self.ref, self.K)
        self._t("trackAndTrace", t0)
        ok = bool(res["haveOneGood"])
        if ok:
            Rn = res["R"] @ Rr; tn = res["R"] @ tr + res["t"]
            a, b = float(res["exposure"][0]), float(res["exposure"][1])
        else:
            Rn, tn = hyps_w[0]; a, b = self.last_exposure
            self.stats["tracking_lost"] += 1
        self._emit("track", image_id=iid, gray=gray, ref_image_id=ref["image_id"], hyps=hyps, ref_exp=ref_exp, init_exp=init_exp, result=res,
                   last_coarse_rmse=self.last_coarse_rmse, levels=self.levels)
        if ok:
            self.last_coarse_rmse = float(res["lastCoarseRMSE"])
        self.history.append((Rn.copy(), tn.copy())); self.last_exposure = (a, b)
        self.stats["frames"] += 1
        if not kept:
            self.stats["traces_redone"] = self.stats.get("traces_redone", 0) + 1
            self.trace(iid, Rn, tn, a, b, traced_fid)
        elif self.obs is not None:
            self._emit("trace", image_id=iid, pairs=pairs, frame_ids=fids, traced_fid=traced_fid, before=before, after=self.trc.points(), counts=counts,
                       tracer_fids=[int(f) for f in self.trc.frame_ids()])
        return iid, Rn, tn, a, b, ok

    def non_keyframe(self, gray, next_gray=None):
        iid, Rn, tn, a, b, ok = self.track_and_trace(gray, next_gray, traced_fid=-1)
        t0 = time.perf_counter()
        self._drop(iid)                                               # CaptureImage::makeUnactive: the id is free for the next frame
        self._t("pyramid_drop", t0)
        return ok

    def keyframe(self, gray, next_gray=None):
        """Hybrid::directMap (direct/Mapping.cpp:47-134)"""
        ctx, ba = self.ctx, self.ba
        fid = self.n_fid; self.n_fid += 1
        iid, Rn, tn, a, b, ok = self.track_and_trace(gray, next_gray, traced_fid=fid)
        # ---- addNewFrame: flagFramesForMarginalization first (BA.cpp:428), then the frame, the prior block, residuals of the old points
        t0 = time.perf_counter()
        counts = self._immature_counts()
        exp0 = ba.export() if self.obs is not None else None
        self._c("addNewFrame", ba.flag_frames_for_marginalization_v, counts)
        self._c("addNewFrame", ba.add_frame, iid, Rn, tn, a, b, 1.0)
        self._poses = None
        grad0 = ctx.pyramid_get(iid, 0)
        kf = {"fid": fid, "image_id": iid, "gray": gray, "grad0": grad0, "taken": set()}
        self.kfs.append(kf)
        self._t("addNewFrame", t0)
        if self.obs is not None:
            self._emit("flag", immature=counts, before=exp0, after=ba.export())
        # ---- activatePoints + addPoints
        t0 = time.perf_counter()
        poses = self.kf_poses()
        apairs = self._activation_pairs(poses)
        fids = [k_["fid"] for k_ in self.kfs]; iids = [k_["image_id"] for k_ in self.kfs]
        before = self.trc.points() if self.obs is not None else None
        activated = self._c("activatePoints+addPoints", self.trc.activate_points, fids, iids, self.K, self.w, self.h, apairs)
        if len(activated):                                            # DSOTracer::activatePoints -> BA::addPoints, in the host mirror (capi.cpp)
            _first, xy = self._c("activatePoints+addPoints", self.trc.add_activated_to_ba, ba, activated, fids)
            tf = self.trc.frame_ids()[np.asarray(activated, np.int64)]
            for (x_, y_), f_ in zip(xy.tolist(), tf.tolist()):
                self.kfs[fids.index(int(f_))]["taken"].add((x_, y_))
        if self.obs is not None:
            pts, alive, act, idp = self.trc.points()
            tfids = self.trc.frame_ids()
        self._t("activatePoints+addPoints", t0)
        if self.obs is not None:
            self._emit("activate", frame_ids=fids, image_ids=iids, pairs=apairs, before=before, after=(pts, alive, act, idp), activated=activated,
                       tracer_fids=[int(f) for f in tfids], grads0=[k_["grad0"] for k_ in self.kfs])
        # ---- run
        exp0 = (ba.export(), ba.prior()) if self.obs is not None else None
        t0 = time.perf_counter()
        ok_run = self._c("run", ba.run)
        self._poses = None
        self._t("run", t0)
        self.run_split.append(ba.run_timing())
        if not ok_run:
            raise RuntimeError("BA run failed: " + ba.last_error())
        self.stats["max_window"] = max(self.stats["max_window"], len(self.kfs))
        if self.obs is not None:
            self._emit("run", before=exp0[0], prior=exp0[1], after=ba.export(), energies=ba.energies(64), iterations=ba.counts()["iterations"],
                       grads0=[k_["grad0"] for k_ in self.kfs], outliers=ba.outliers().copy())
        # ---- makeCoarseDepthL0 on the new keyframe
        t0 = time.perf_counter()
        cd = self._c("makeCoarseDepthL0", self._coarse_depth, len(self.kfs) - 1)
        nout = None if self.mapper_only else self._c("makeCoarseDepthL0", self.trk.make_coarse_depth, iid, self.levels, cd)      # (mapper alone: the tracker front builds the lists at the hand-over)
        self._t("makeCoarseDepthL0", t0)
        if not self.mapper_only:
            self._emit("coarse", image_id=iid, gray=gray, pts=cd, n_lists=nout, levels=self.levels)
        # ---- tryMarginalize, marginalizePointsF
        exp0 = (ba.export(), ba.algebra()) if self.obs is not None else None
        t0 = time.perf_counter()
        if not self._c("tryMarginalize", ba.try_marginalize):
            raise RuntimeError("tryMarginalize failed: " + ba.last_error())
        self._t("tryMarginalize", t0)
        if self.obs is not None:
            self._emit("try_marginalize", before=exp0[0], algebra=exp0[1], after=ba.export(), grads0=[k_["grad0"] for k_ in self.kfs])
        exp0 = (ba.export(), ba.prior(), ba.algebra()) if self.obs is not None else None
        t0 = time.perf_counter()
        if not self._c("marginalizePointsF", ba.marginalize_points):
            raise RuntimeError("marginalizePointsF failed: " + ba.last_error())
        self._t("marginalizePointsF", t0)
        if self.obs is not None:
            self._emit("marginalize_points", before=exp0[0], prior_before=exp0[1], algebra=exp0[2], prior_after=ba.prior(), after=ba.export(),
                       grads0=[k_["grad0"] for k_ in self.kfs])
        # ---- makeNewTraces on the new keyframe
        t0 = time.perf_counter()
        self._make_new_traces(kf)
        self._t("makeNewTraces", t0)
        # ---- marginalizeFrames + release of their images
        exp0 = (ba.export(), ba.prior(), ba.algebra()) if self.obs is not None else None
        t0 = time.perf_counter()
        removed = list(self._c("marginalizeFrames", ba.marginalize_frames))
        self._poses = None
        for idx in sorted(removed, reverse=True):
            self._drop(self.kfs[idx]["image_id"])
            del self.kfs[idx]
        self._t("marginalizeFrames", t0)
        self.stats["marginalized_frames"] += len(removed)
        if self.obs is not None:
            self._emit("marginalize_frames", before=exp0[0], prior_before=exp0[1], algebra=exp0[2], removed=removed, prior_after=ba.prior(), after=ba.export())
        self.ref = len(self.kfs) - 1
        self._c("makeNewTraces", self.trc.prepare_resident, [k_["fid"] for k_ in self.kfs])      # the immature set on the device follows the keyframe's edits now, not in front of the next frame
        # the tracked pose of the newest frame is now the optimised one (the next motion model starts from it)
        f = ba.frame(self.ref)
        self.history[-1] = (f["R"].copy(), f["t"].copy()); self.last_exposure = (float(f["ab"][0]), float(f["ab"][1]))
        self.stats["keyframes"] += 1
        if self.mapper_only:
            self.handover = {"cd": cd, "R": f["R"].copy(), "t": f["t"].copy(), "ab": (float(f["ab"][0]), float(f["ab"][1])),
                             "energies": self.ba.energies(64).copy(), "iterations": self.ba.counts()["iterations"], "outliers": self.ba.outliers().copy(),
                             "window": [(p[0].copy(), p[1].copy(), p[2], p[3]) for p in self.kf_poses()]}
        return ok

    def run(self, seq, n_frames=None):
        """the whole shard: bootstrap on frame 0, then every frame in order"""
        self.bootstrap(seq.gray[0], seq.R_true[0], seq.t_true[0], seq.boot_px, seq.boot_idepth)
        kfset = set(seq.keyframes)
        last = n_frames or seq.n_frames
        for k in range(1, last):
            nxt = seq.gray[k + 1] if k + 1 < last else None
            if k in kfset:
                self.keyframe(seq.gray[k], nxt)
            else:
                self.non_keyframe(seq.gray[k], nxt)
        return self.stats

    def library_summary(self):
        """per stage: seconds spent inside the library's calls (summed over the calls of one stage invocation), as timing_summary reports the stage's wall clock"""
        out = {}
        for k, v in self.lib_times.items():
            n = max(len(self.times.get(k, [])), 1)                      # several calls of one stage invocation are one entry
            tot = float(np.sum(v))
            out[k] = {"calls": int(len(v)), "total_ms": 1e3 * tot, "mean_ms_per_stage": 1e3 * tot / n}
        return out

    def timing_summary(self):
        out = {}
        for k, v in self.times.items():
            a = np.array(v)
            out[k] = {"calls": int(len(a)), "mean_ms": float(1e3 * a.mean()), "median_ms": float(1e3 * np.median(a)), "max_ms": float(1e3 * a.max())}
        return out


class TrackerFront:
    """The tracking half of the shard on its OWN context: pyramid of the frame, DSOTracker::trackWithMotionModel against the reference keyframe the
    last hand-over installed (DSOTracker::getLastComputed, Hybrid.cpp:431-458), motion-model history."""

    def __init__(self, ctx, K, levels, id_pool=16, observer=None):
        self.ctx, self.K, self.levels = ctx, tuple(K), levels
        self.obs = observer                                          # stages "bootstrap" / "coarse" (reference lists) and "track", as DirectPipeline emits them
        self.grays = {}
        self.trk = host.HostTracker(ctx); self.trk.set_calibration(*K)
        self.free_ids = list(range(1, id_pool + 1))
        self.history = []                                            # world->cam (R, t) per frame index
        self.last_exposure = (0.0, 0.0)
        self.last_coarse_rmse = 100.0
        self.ref = None                                              # dict(image_id, R, t, a, b)
        self.images = {}                                             # frame index -> image id of the frames whose pyramid this context still holds
        self._prefetched = None
        self.lib_s = 0.0
        self.results = []

    def close(self):
        self.trk.close()

    def _take(self):
        return self.free_ids.pop(0)

    def drop_frame(self, k):
        iid = self.images.pop(k, None)
        if iid is not None:
            self.ctx.pyramid_drop(iid); self.free_ids.append(iid); self.free_ids.sort()

    def build(self, k, gray, next_k=None, next_gray=None):
        t0 = time.perf_counter()
        if self._prefetched is not None and self._prefetched[0] == k:
            iid = self._prefetched[1]
        else:
            iid = self._take(); self.ctx.pyramid_build(iid, gray, self.levels)
        self._prefetched = None
        self.images[k] = iid
        if self.obs is not None:
            self.grays[k] = gray
        if next_gray is not None:
            nid = self._take(); self.ctx.pyramid_build_async(nid, next_gray, self.levels)
            self._prefetched = (next_k, nid)
        self.lib_s += time.perf_counter() - t0
        return iid

    def adopt(self, k, ho):
        """hand-over of keyframe k's mapping: reference lists (makeCoarseDepthL0 on this context's pyramid of the frame), the optimised pose"""
        t0 = time.perf_counter()
        iid = self.images[k]
        nout = self.trk.make_coarse_depth(iid, self.levels, ho["cd"])
        self.ctx.sync()
        self.lib_s += time.perf_counter() - t0
        if self.obs is not None:
            self.obs("bootstrap" if self.ref is None else "coarse", dict(image_id=iid, gray=self.grays[k], pts=ho["cd"], n_lists=nout, levels=self.levels))
            self.grays = {k: self.grays[k]}
        old = self.ref
        self.ref = {"k": k, "image_id": iid, "R": ho["R"], "t": ho["t"], "a": ho["ab"][0], "b": ho["ab"][1]}
        if old is not None and old["k"] != k:
            self.drop_frame(old["k"])
        self.history[k] = (ho["R"].copy(), ho["t"].copy())            # the keyframe's pose is the optimised one from here on
        if k == len(self.history) - 1:
            self.last_exposure = ho["ab"]

    def _hypotheses(self):
        return DirectPipeline._hypotheses(self)

    def track(self, k, gray, next_k=None, next_gray=None):
        iid = self.build(k, gray, next_k, next_gray)
        Rr, tr, ar, br = self.ref["R"], self.ref["t"], self.ref["a"], self.ref["b"]
        hyps_w = self._hypotheses()
        hyps = [_rel(Rr, tr, Rw, tw) for Rw, tw in hyps_w]
        ref_exp = [ar, br, 1.0]; init_exp = [self.last_exposure[0], self.last_exposure[1], 1.0]
        t0 = time.perf_counter()
        res = self.trk.track_with_motion_model(iid, self.levels, hyps, ref_exp, init_exp, batched=True)
        self.ctx.sync()
        self.lib_s += time.perf_counter() - t0
        ok = bool(res["haveOneGood"])
        if self.obs is not None:
            self.obs("track", dict(image_id=iid, gray=gray, ref_image_id=self.ref["image_id"], hyps=hyps, ref_exp=ref_exp, init_exp=init_exp, result=res,
                                   last_coarse_rmse=self.last_coarse_rmse, levels=self.levels))
        if ok:
            Rn = res["R"] @ Rr; tn = res["R"] @ tr + res["t"]
            a, b = float(res["exposure"][0]), float(res["exposure"][1])
            self.last_coarse_rmse = float(res["lastCoarseRMSE"])
        else:
            Rn, tn = hyps_w[0]; a, b = self.last_exposure
        self.history.append((Rn.copy(), tn.copy())); self.last_exposure = (a, b)
        self.results.append((k, Rn.copy(), tn.copy(), a, b, ok, float(res["lastCoarseRMSE"]) if ok else None))
        return Rn, tn, a, b, ok


class SplitPipeline:
    """The shard as the reference runs it with linearizeDirect off: the tracker on the SLAM thread, Hybrid::directMappingLoop on a thread of its own
    (Hybrid.cpp:103-106, direct/Mapping.cpp:3-41) — two contexts, two streams, no shared mutable state.  Frames are tracked against the reference
    keyframe of the last hand-over while the mapper works through its queue (traceNewCoarse of non-keyframes, directMap of keyframes, in frame order).
    The hand-over of keyframe j (reference lists + optimised pose) is adopted before frame j + 1 + lag is tracked (lag = 1: frame j + 1 is tracked
    beside the mapping of j, against the previous reference) — a fixed schedule, so that `threaded=False` (the same jobs run inline on the calling
    thread at the moment they are queued) is the same computation, stage for stage and bit for bit."""

    def __init__(self, ctx_tracker, ctx_mapper, K, w, h, levels, threaded=True, lag=1, front_observer=None, **kw):
        self.front = TrackerFront(ctx_tracker, K, levels, observer=front_observer)
        self.mapper = DirectPipeline(ctx_mapper, K, w, h, levels, mapper_only=True, **kw)
        self.threaded, self.lag = threaded, lag
        self.done = {}                                               # keyframe index -> hand-over
        self.kf_log = []
        self.stall_s = 0.0
        self._err = None
        if threaded:
            import queue
            import threading
            self._q = queue.Queue()
            self._cv = threading.Condition()
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()

    def close(self):
        if self.threaded:
            self._q.put(None); self._th.join()
        self.front.close(); self.mapper.close()

    # ---- the mapper's side
    def _job(self, job):
        kind, k, gray, pose = job
        m = self.mapper
        if kind == "boot":
            m.bootstrap(*gray)
        else:
            m.injected = pose
            (m.keyframe if kind == "kf" else m.non_keyframe)(gray)
        if kind != "nonkf":
            ho = m.handover; m.handover = None
            self.kf_log.append((k, ho))
            return ho
        return None

    def _loop(self):
        while True:
            job = self._q.get()
            if job is None:
                return
            try:
                ho = self._job(job)
            except Exception as e:                                   # the front sees it at its next wait
                with self._cv:
                    self._err = e; self._cv.notify_all()
                return
            if ho is not None:
                with self._cv:
                    self.done[job[1]] = ho; self._cv.notify_all()

    def _submit(self, job):
        if self.threaded:
            self._q.put(job)
        else:
            ho = self._job(job)
            if ho is not None:
                self.done[job[1]] = ho

    def _wait(self, k):
        if self.threaded:
            t0 = time.perf_counter()
            with self._cv:
                while k not in self.done and self._err is None:
                    self._cv.wait(0.5)
                    if time.perf_counter() - t0 > 300.0:               # never wait forever for a mapper that stopped answering
                        raise RuntimeError("SplitPipeline: the mapper thread did not hand keyframe %d over within 300 s" % k)
            self.stall_s += time.perf_counter() - t0
            if self._err is not None:
                raise self._err
        return self.done.pop(k)

    def run(self, seq, n_frames=None):
        f = self.front
        last = n_frames or seq.n_frames
        kfset = set(seq.keyframes)
        f.build(0, seq.gray[0])
        f.history.append((np.asarray(seq.R_true[0], float).copy(), np.asarray(seq.t_true[0], float).copy()))
        self._submit(("boot", 0, (seq.gray[0], seq.R_true[0], seq.t_true[0], seq.boot_px, seq.boot_idepth), None))
        f.adopt(0, self._wait(0))
        pending = []                                                 # keyframes whose hand-over is still to be adopted, oldest first
        for k in range(1, last):
            while pending and k >= pending[0] + 1 + self.lag:
                j = pending.pop(0)
                f.adopt(j, self._wait(j))
            nxt = seq.gray[k + 1] if k + 1 < last else None
            pose = f.track(k, seq.gray[k], k + 1, nxt)
            if k in kfset:
                pending.append(k)
                self._submit(("kf", k, seq.gray[k], pose))
            else:
                self._submit(("nonkf", k, seq.gray[k], pose))
                f.drop_frame(k)
        for j in pending:
            f.adopt(j, self._wait(j))
        if self.threaded:                                            # the queue drained: the last non-keyframes' traces are part of the shard
            self._q.put(None); self._th.join(); self.threaded = False
            if self._err is not None:
                raise self._err
        return dict(self.mapper.stats, tracker_stall_s=self.stall_s)


def _so3_log(R):
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    if th < 1e-9:
        return np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    return th / (2 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
