"""The host mirror of DSOInitializer (libcml_amd/host/DSOInitializer.{h,cpp}): setFirst's point records, makeNN, and
tryInitialize's per-level Levenberg loop around the device calcResAndGS, run over a synthetic sequence with a growing
baseline.  There is no second implementation of this control flow to compare with (calcResAndGS itself is parity-tested in
test_initializer_gpu.py), so the bar is functional: the initializer snaps, succeeds 6 frames later like the reference's
`mFrameID > mSnappedAt + 5`, and what it delivers — relative pose up to scale, inverse depths up to the same scale — is
the scene it was shown."""
import numpy as np
import pytest

from libcml_amd import device, host, synth

pytestmark = pytest.mark.gpu

W_, H_, K_ = 320, 240, (260.0, 260.0, 159.5, 119.5)
LEVELS = 4


def _q_from_R(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    return np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])


def _R_from_q(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _sequence(n_frames, seed=4):
    rng = np.random.default_rng(seed)
    n = np.array([0.25, -0.15, 1.0]); n /= np.linalg.norm(n)
    d = 9.0
    tex = synth.Texture(rng, scale=0.07 * K_[0] / (d * 32.0))
    frames = []
    for k in range(n_frames):
        c = k * np.array([0.05, 0.012, 0.03])                      # camera centre: mostly sideways, a little forward
        R = synth.so3_exp(np.deg2rad([0.05, -0.08, 0.03]) * k)
        t = -R @ c
        img, depth = synth.render(tex, K_, R, t, W_, H_, n, d)
        frames.append(dict(R=R, t=t, gray=img.astype(np.float32), depth=depth))
    return frames


def test_initializer_snaps_and_recovers_the_scene():
    F = _sequence(14)
    ctx = device.Ctx(max_frames=2)
    try:
        ids = [3000 + k for k in range(len(F))]
        for k, f in enumerate(F):
            ctx.pyramid_build(ids[k], f["gray"], LEVELS)
        grays, Ks, pixels = [], [], []
        for lvl in range(LEVELS):
            g3 = ctx.pyramid_get(ids[0], lvl)
            gray = np.ascontiguousarray(g3[..., 0]); mag = np.hypot(g3[..., 1], g3[..., 2])
            s = 2.0 ** lvl
            grays.append(gray); Ks.append((K_[0] / s, K_[1] / s, (K_[2] + 0.5) / s - 0.5, (K_[3] + 0.5) / s - 0.5))
            step = max(1, 3 - lvl)
            ys, xs = np.mgrid[4:gray.shape[0] - 5:step, 4:gray.shape[1] - 5:step]
            keep = mag[ys, xs] > np.percentile(mag, 55)
            pixels.append((xs[keep].astype(np.int32), ys[keep].astype(np.int32)))          # raster order
        init = host.HostInitializer(ctx)
        ident = np.array([1.0, 0, 0, 0, 0, 0, 0])
        assert init.set_first(grays, Ks, pixels, ident), init.last_error()
        P0 = init.points(0)
        n0 = len(P0["iR"])
        assert n0 > 1500 and np.all(P0["neighbours"] >= 0) and np.all(P0["parent"] >= 0)
        # makeNN: the first neighbour of a point is the point itself (distance 0), the others are near
        assert np.array_equal(P0["neighbours"][:, 0], np.arange(n0))
        dn = np.linalg.norm(P0["xy"][P0["neighbours"][:, 5]] - P0["xy"], axis=1)
        assert np.median(dn) < 8
        results = []
        for k in range(1, len(F)):
            # the caller hands the previous estimate over as the frame's camera only on the first call (mCurrentCamera is kept inside)
            r = init.try_initialize(ids[k], ident)
            st = init.state()
            results.append((r, st["snapped"], st["frame_id"]))
            assert r in (0, 1), (k, init.last_error())
            if r == 1:
                break
        st = init.state()
        assert st["snapped"] and results[-1][0] == 1, results
        snapped_at = next(i for i, x in enumerate(results) if x[1]) + 1
        assert len(results) == snapped_at + 6, (results, snapped_at)                       # mFrameID > mSnappedAt + 5
        assert st["accepted"] > 10 and st["calc_calls"] == st["accepted"] + st["rejected"] + LEVELS * len(results)
        k = len(results)                                                                    # the frame the initializer ended on
        R_est = _R_from_q(st["qt"][:4]); t_est = st["qt"][4:]
        R_true = F[k]["R"]; t_true = F[k]["t"]
        ang = np.arccos(np.clip((np.trace(R_est @ R_true.T) - 1) / 2, -1, 1))
        cosang = float(t_est @ t_true / (np.linalg.norm(t_est) * np.linalg.norm(t_true)))
        assert ang < np.deg2rad(0.5) and cosang > 0.95, (np.rad2deg(ang), cosang)
        # inverse depths against the true inverse depth of the reference pixel, up to one global scale.  `idepth` is what the
        # photometric optimisation estimates; `iR` is the regularised value, and optReg (DSOInitializer.cpp:810-842) is mirrored
        # literally: its neighbourhood median reads the level's FIRST ten points (`mPoints[lvl][j]`, :825) rather than the
        # point's neighbours, so where those ten are not good — here they hug the image corner and leave the image — only the
        # pull towards initialiR = 1 remains and iR sits at 1
        P0 = init.points(0)
        good = P0["good"] == 1
        xi = np.clip(np.round(P0["xy"][:, 0]).astype(int), 0, W_ - 1); yi = np.clip(np.round(P0["xy"][:, 1]).astype(int), 0, H_ - 1)
        true_id = 1.0 / F[0]["depth"][yi, xi]
        assert good.mean() > 0.8
        ratio = P0["idepth"][good] / true_id[good]
        scale = np.median(ratio)
        assert np.corrcoef(P0["idepth"][good], true_id[good])[0, 1] > 0.8
        assert np.median(np.abs(ratio / scale - 1)) < 0.08, float(np.median(np.abs(ratio / scale - 1)))
        # that scale is the one of the translation: lengths in the initializer's units are true lengths / scale, and
        # onInitializationSuccess has divided the translation by `rescale`
        assert abs(np.linalg.norm(t_est) * st["rescale"] * scale / np.linalg.norm(t_true) - 1) < 0.1
        assert abs(np.median(P0["iR"][good]) * st["rescale"] - 0.5) < 0.02                 # median inverse depth 0.5 after rescaling
        init.close()
    finally:
        ctx.close()
