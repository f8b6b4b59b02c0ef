"""The host mirrors of the reference's g2o classes (libcml_amd/host/IndirectG2O.{h,cpp}) over the C ABI: what
IndirectCameraOptimizer::optimize and IndirectBundleAdjustment::localOptimize / apply keep on the host — null map points,
information weights from the pyramid level, the early returns, the write-back and the edge-removal policy — checked against the
oracle fed with the arrays the reference would have put into its g2o graph."""
import numpy as np
import pytest

from libcml_amd import abi, device, host
from tests import lba_setup as LS
from tests import pnp_setup as PS

pytestmark = pytest.mark.gpu


def _matchings(S, rng):
    m = S["matches"]; n = len(m)
    hm = np.zeros(n, host.HOST_MATCHING_DTYPE)
    hm["has_map_point"] = 1; hm["X"] = m["X"]; hm["obs"] = m["obs"]; hm["scale_factor_base"] = 1.2
    hm["level"] = rng.integers(0, 8, n); hm["descriptor_distance"] = rng.integers(10, 60, n)
    return hm


def test_camera_optimizer_mirror():
    S = PS.scene(n=500, seed=21)
    rng = np.random.default_rng(3)
    hm = _matchings(S, rng)
    null = rng.uniform(size=len(hm)) < 0.05                     # matchings whose map point is null: flagged, not optimised (:57-62)
    hm["has_map_point"][null] = 0
    ctx = device.Ctx(max_frames=2)
    try:
        opt = host.HostCameraOptimizer(ctx)
        ok, R, t, cov, out = opt.optimize(S["R0"], S["t0"], S["K"], hm, np.zeros(0, np.uint8), compute_covariance=True)
        # a camera argument overrides the frame camera as the starting pose (:132-135)
        ok2, R2, t2, _, out2 = opt.optimize(np.eye(3), np.zeros(3), S["K"], hm, np.zeros(0, np.uint8), camera=(S["R0"], S["t0"]))
        okp, Rp, tp, _, idx = opt.optimize_points(S["R0"], S["t0"], S["K"], hm)
    finally:
        ctx.close()
    # the oracle on the arrays the reference would have built
    keep = ~null
    m = np.zeros(int(keep.sum()), abi.PNP_MATCH_DTYPE)
    m["X"] = hm["X"][keep]; m["obs"] = hm["obs"][keep]
    m["inv_sigma2"] = 1.0 / hm["descriptor_distance"][keep]; m["info"] = 1.0 / (1.2 ** hm["level"][keep]) ** 2
    oo = np.zeros(len(m), np.uint8)
    ro = PS.oracle_pnp(S["R0"], S["t0"], S["K"], m, oo, algorithm=abi.PNP_LEVENBERG, compute_covariance=True)
    assert ok and ro.is_ok == 1
    assert out[null].all() and np.array_equal(out[keep], oo)
    assert np.abs(R.ravel() - np.array(list(ro.R))).max() < 1e-9 and np.abs(t - np.array(list(ro.t))).max() < 1e-9
    assert np.abs(cov - np.array(list(ro.covariance))).max() <= 1e-8 * np.abs(cov).max()
    assert ok2 and np.array_equal(out2, out) and np.array_equal(R2, R) and np.array_equal(t2, t)
    m2 = m.copy(); m2["inv_sigma2"] = m2["info"]                                    # the Gauss-Newton overload weighs edges by level (:280-285)
    o2 = np.zeros(len(m2), np.uint8)
    r2 = PS.oracle_pnp(S["R0"], S["t0"], S["K"], m2, o2, algorithm=abi.PNP_GAUSS_NEWTON)
    assert okp and r2.is_ok == 1
    assert np.array_equal(np.flatnonzero(keep)[o2 == 1], idx)
    assert np.abs(Rp.ravel() - np.array(list(r2.R))).max() < 1e-9


@pytest.mark.parametrize("fix_frames", [True, False])
def test_local_ba_mirror(fix_frames):
    S = LS.scene(pose_noise=0.0 if fix_frames else 0.02, n_points=500, seed=12)
    fr = S["frames"]; nl = int((fr["fixed"] == 0).sum())
    ids = 100 + 7 * np.arange(len(fr))                                              # arbitrary frame ids
    hf = np.zeros(len(fr), host.HOST_LBA_FRAME_DTYPE)
    hf["id"] = ids; hf["R"] = fr["R"]; hf["t"] = fr["t"]; hf["K"] = fr["K"]
    npts = len(S["points"])
    hp = np.zeros(npts, host.HOST_LBA_POINT_DTYPE)
    hp["id"] = 5000 + np.arange(npts); hp["X"] = S["points"]
    E = S["edges"]; off = S["off"]
    rng = np.random.default_rng(1)
    level = rng.integers(0, 8, len(E))
    first = off[:-1]
    hp["reference_frame_id"] = ids[E["frame"][first]]                               # the first observing frame is the reference frame
    ap = np.zeros(len(E) + 40, host.HOST_LBA_APPARITION_DTYPE)
    ap["point"][:len(E)] = np.repeat(np.arange(npts), np.diff(off)); ap["frame_id"][:len(E)] = ids[E["frame"]]
    ap["obs"][:len(E)] = E["obs"]; ap["level"][:len(E)] = level; ap["scale_factor_base"] = 1.2
    ap["point"][len(E):] = rng.integers(0, npts, 40); ap["frame_id"][len(E):] = 9999        # apparitions in frames outside both sets: ignored (:131)
    order = np.argsort(ap["point"], kind="stable"); ap = ap[order]
    ctx = device.Ctx(max_frames=2)
    try:
        ba = host.HostLocalBA(ctx, num_iteration=5, refine_iteration=0)
        assert not ba.local_optimize(hf[:2], hf[nl:], hp, ap, fix_frames) and "Not enough frames" in ba.last_error()       # :43-46
        assert not ba.local_optimize(hf[:nl], hf[nl:nl + 2], hp, ap, fix_frames) and "fixed cameras" in ba.last_error()    # :95-98
        lo0, X0, rem0, _ = ba.apply()
        assert len(rem0) == 0                                                       # nothing optimised yet: apply() is a no-op
        stop = np.ones(1, np.uint8)                                                 # pbStopFlag set before the call: refused, :173-178
        assert not ba.local_optimize(hf[:nl], hf[nl:], hp, ap, fix_frames, stop_flag=stop) and "Stop flag" in ba.last_error()
        assert ba.local_optimize(hf[:nl], hf[nl:], hp, ap, fix_frames), ba.last_error()
        lo, X, rem, res = ba.apply()
        ba.close()
    finally:
        ctx.close()
    Eo = E.copy(); Eo["inv_sigma2"] = 1.0 / (1.2 ** level) ** 2
    fo = fr.copy()
    fr_o, pts_o, bad_o, r_o = LS.oracle_lba(fo, S["points"], off, Eo, fix_frames, 5, 0)
    tol = 1e-9 if fix_frames else 1e-6          # (the level weights are formed by std::pow on one side, numpy's power on the other)
    assert np.abs(X - pts_o).max() <= tol * max(1.0, np.abs(pts_o).max())
    assert np.abs(lo["R"] - fr_o["R"][:nl]).max() <= tol and np.abs(lo["t"] - fr_o["t"][:nl]).max() <= tol
    assert abs(res.n_bad - r_o.n_bad) <= 2
    # removal policy (:325-334): flagged edges, except the observation in the point's reference frame
    pt_of = np.repeat(np.arange(npts), np.diff(off))
    want = {(int(ids[E["frame"][e]]), int(hp["id"][pt_of[e]])) for e in np.flatnonzero(bad_o) if ids[E["frame"][e]] != hp["reference_frame_id"][pt_of[e]]}
    got = {(int(a), int(b)) for a, b in rem}
    assert len(got ^ want) <= 2
    assert len(want) > 0
