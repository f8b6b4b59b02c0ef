"""CPU checks of the sequence-shard plumbing (libcml_amd/sequence.py) and of the checker's restatements of the reference's host logic
(tests/sequence_check.py) — what can be held without a GPU: determinism of the synthetic sequence, the stand-in selector, the patch /
weight helpers against their scalar definitions, and the marginalisation / flagging rules on hand-made window states."""
import numpy as np

from libcml_amd import host, sequence, synth
from tests import sequence_check as SC

SMALL = (0, 0, 320, 240, 3, 260.0, 260.0, 159.5, 119.5)


def test_synthetic_sequence_is_deterministic_and_scheduled():
    a = sequence.make_sequence(n_frames=12, config=SMALL, n_bootstrap=200)
    b = sequence.make_sequence(n_frames=12, config=SMALL, n_bootstrap=200)
    assert all(np.array_equal(x, y) for x, y in zip(a.gray, b.gray)) and np.array_equal(a.boot_px, b.boot_px) and a.keyframes == b.keyframes
    gaps = np.diff(a.keyframes)
    assert a.keyframes[0] == 0 and np.all((gaps >= 3) & (gaps <= 5))
    c = sequence.make_sequence(n_frames=12, config=SMALL, n_bootstrap=200, shard=1)
    assert not np.array_equal(a.gray[3], c.gray[3])                  # another shard is another sequence
    assert len(set(map(tuple, a.boot_px.tolist()))) == len(a.boot_px) == 200
    assert np.all(a.boot_idepth > 0)
    # consecutive frames overlap (the tracker's premise): mean absolute difference well below the image contrast
    assert np.abs(a.gray[1] - a.gray[0]).mean() < 0.5 * a.gray[0].std()


def test_patches_and_weights_match_their_scalar_definitions():
    rng = np.random.default_rng(3)
    g = rng.uniform(0, 255, size=(60, 80, 3)).astype(np.float32)
    px = np.array([[10, 12], [40, 30], [70, 50]])
    gray, dp, G = sequence._patches(g, px)
    w = sequence._weights(dp)
    for i, (x, y) in enumerate(px):
        H = np.zeros((2, 2))
        for k, (dx, dy) in enumerate(synth.STAR8):
            t = g[y + dy, x + dx]
            assert gray[i, k] == t[0] and np.array_equal(dp[i, 3 * k:3 * k + 3], t)
            gr = t[1:3].astype(np.float64)
            H += np.outer(gr, gr)
            assert w[i, k] == np.float32(np.sqrt(2500.0 / (2500.0 + gr @ gr)))       # BA.cpp:405-411
        assert np.allclose(G[i], H.ravel(), rtol=1e-15)


def _export(N, pts, res, flagged=()):
    fr = np.zeros(N, host.HOST_BA_FRAME_DTYPE)
    fr["eval_q"][:, 0] = 1; fr["pre_q"][:, 0] = 1; fr["ab_exposure"] = 1
    fr["keyid"] = np.arange(N); fr["id"] = np.arange(N)
    for k in range(N):
        fr["pre_t"][k] = (-0.5 * k, 0, 0)
    for f in flagged:
        fr["flagged"][f] = 1
    pt = np.zeros(len(pts), host.HOST_BA_POINT_DTYPE)
    for i, (hst, idepth, ngood, last0, last1, hess) in enumerate(pts):
        pt["host"][i] = hst; pt["idepth"][i] = idepth; pt["numGoodResiduals"][i] = ngood; pt["alive"][i] = 1
        pt["lastResidualState"][i] = (last0, last1); pt["idepth_hessian"][i] = hess
    rs = np.zeros(len(res), host.HOST_BA_RESIDUAL_DTYPE)
    for i, (p, t, st) in enumerate(res):
        rs["point"][i] = p; rs["target"][i] = t; rs["state_state"][i] = st; rs["alive"][i] = 1
    return fr, pt, rs


def test_try_marginalize_classification_rules():
    """BA.cpp:2240-2363 / isOOB :2515-2554 on a hand-made window of 5 frames (a point with only 3 residuals IN of which one goes into the
    flagged frame would itself be OOB by the first rule of isOOB: the healthy point needs four)"""
    IN, OOB, OUT = 0, 1, 2
    pts = [(0, 0.5, 20, IN, IN, 100.0),      # 0: healthy, host not flagged -> stays
           (0, 0.5, 20, OOB, IN, 100.0),     # 1: last residual OOB, 3 residuals, 20 good -> candidate
           (0, -0.1, 20, IN, IN, 100.0),     # 2: negative inverse depth -> dropped
           (1, 0.5, 20, IN, IN, 100.0),      # 3: host flagged -> candidate
           (1, 0.5, 2, IN, IN, 100.0),       # 4: host flagged but too few good residuals -> dropped
           (0, 0.5, 20, OUT, OUT, 100.0),    # 5: two outliers in a row with >= 2 residuals IN -> candidate
           (0, 0.5, 20, IN, IN, 100.0)]      # 6: no residual at all -> dropped
    res = []
    for p in (0, 1, 3, 4, 5):
        h = pts[p][0]
        res += [(p, t, IN) for t in range(5) if t != h]
    fr, pt, rs = _export(5, pts, res, flagged=(1,))
    cand, drop = SC.try_marginalize_sets(fr, pt, rs)
    assert cand == [1, 3, 5] and drop == [2, 4, 6]


def test_flag_frames_rules():
    """BA.cpp:603-716: a frame with < 5 % of its points left is flagged; with maxFrames reached the distance score picks one more"""
    pts = [(0, 0.5, 20, 0, 0, 100.0)]
    res = [(0, t, 0) for t in range(1, 6)]
    fr, pt, rs = _export(6, pts, res)
    fr["numResidualsOut"][2] = 1000                                   # frame 2: 1 residual in, 1000 out
    flags = SC.flag_frames(fr, pt, rs, immature=[0] * 6, max_frames=6)
    assert flags[2] and sum(flags) == 1                               # 6 - 1 flagged = 5 < maxFrames: nobody else goes
    fr["numResidualsOut"][2] = 0
    flags = SC.flag_frames(fr, pt, rs, immature=[50] * 6, max_frames=6)
    assert sum(flags) == 1 and not flags[0] and not flags[5]          # the score never picks keyid 0 or the newest frame (minFrameAge)
