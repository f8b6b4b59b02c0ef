"""One enqueue, one host wait per tracked frame (Hybrid.cpp:383-442): cmlhip_tracker_optimize_batch_async + cmlhip_tracer_trace_resident_tracked_async
+ cmlhip_tracker_optimize_wait + cmlhip_tracer_trace_resident_finish through the C ABI.

  * the batch's results equal the synchronous call's in every bit (same kernel, same launch);
  * the pairs host -> frame the device forms from the first hypothesis' result agree with the caller-side formula (DSOTracer.cpp:606-608,
    Exposure.h:119-123) to rounding, and the trace that used them equals oracle/orc_tracer.c on THOSE pairs in every bit;
  * keep = 0 restores every field trace() writes (the journal);
  * the host mirror's fused call returns what its two separate calls return."""
import numpy as np
import pytest

from libcml_amd import abi, device, host
from tests import oracle_lib as O
from tests import tracer_setup as TS
from tests import trk_opt_setup as TO

pytestmark = pytest.mark.gpu

FIELDS = ("last_status", "idepth_min", "idepth_max", "quality", "last_uv", "last_pixel_interval")


def _bits_equal(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    if a.dtype.kind == "f":
        u = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
        return bool(((a.view(u) == b.view(u)) | (np.isnan(a) & np.isnan(b))).all())
    return bool(np.array_equal(a, b))


def _host_pairs(K, poses, Rn, tn, an, bn):
    fx, fy, cx, cy = K
    Km = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]]); Ki = np.linalg.inv(Km)
    pr = np.zeros(len(poses), abi.TRACE_PAIR_DTYPE)
    for h, (Rh, th, ah, bh) in enumerate(poses):
        R = Rn @ Rh.T; t = tn - R @ th
        pr["KRKi"][h] = (Km @ R @ Ki).ravel(); pr["Kt"][h] = Km @ t
        a = np.exp(an - ah)
        pr["aff_a"][h] = a; pr["aff_b"][h] = bn - a * bh
    return pr


@pytest.fixture(scope="module")
def scene():
    P = TO.make_problem("small")
    W, s = P.W, P.s
    ctx = device.Ctx(max_frames=W.N)
    ctx.pyramid_build(501, W.gray[s.new], P.levels)
    for l in range(P.levels):
        ctx.tracker_set_reference(l, P.uvic[l])
    grads0 = [O.build_pyramid(W.gray[k], 1)[1][0] for k in range(W.N)]
    pts = TS.make_immature(W, grads0)
    pts = pts[pts["host"] < s.new].copy()
    hosts = [(W.R_eval[h].copy(), W.t_eval[h].copy(), float(W.aff_eval[h][0]), float(W.aff_eval[h][1])) for h in range(s.new)]
    hyps = [TO.perturbed(P, (0.004, -0.003, 0.002), (0.03, -0.02, 0.025)), TO.perturbed(P, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)), TO.perturbed(P, (-0.01, 0.004, 0.0), (0.05, 0.0, -0.04))]
    yield P, ctx, pts, hosts, hyps
    ctx.close()


def test_async_batch_equals_the_synchronous_call(scene):
    P, ctx, pts, hosts, hyps = scene
    a = ctx.tracker_optimize_batch(501, P.levels, P.W.K, P.ref_exp, P.init_exp, P.prm, hyps)
    ctx.tracker_optimize_batch_async(501, P.levels, P.W.K, P.ref_exp, P.init_exp, P.prm, hyps)
    b = ctx.tracker_optimize_wait()
    for x, y in zip(a, b):
        assert np.array_equal(np.array(x.R[:]).view(np.uint64), np.array(y.R[:]).view(np.uint64)) and np.array_equal(np.array(x.t[:]).view(np.uint64), np.array(y.t[:]).view(np.uint64))
        assert x.a == y.a and x.b == y.b and x.n_steps == y.n_steps and list(x.E[:]) == list(y.E[:])


def test_speculative_trace_against_the_oracle_and_rollback(scene):
    P, ctx, pts, hosts, hyps = scene
    W, s = P.W, P.s
    prm = abi.default_tracer_params()
    ref = hosts[s.ref]
    ctx.tracer_set_points(pts)
    ctx.tracker_optimize_batch_async(501, P.levels, W.K, P.ref_exp, P.init_exp, P.prm, hyps)
    ctx.tracer_trace_resident_tracked_async(501, prm, hosts, ref, W.K, skip_host=-2)
    res = ctx.tracker_optimize_wait()                          # the ONE wait
    counts, pairs_dev = ctx.tracer_trace_resident_finish(keep=True)
    after = ctx.tracer_get_points()
    # the pairs the device formed from result 0, against the caller-side formula on the same result
    R0 = np.array(res[0].R[:]).reshape(3, 3); t0 = np.array(res[0].t[:])
    Rn = R0 @ ref[0]; tn = R0 @ ref[1] + t0
    pr_h = _host_pairs(W.K, hosts, Rn, tn, res[0].a, res[0].b)
    for name in ("KRKi", "Kt", "aff_a", "aff_b"):
        d = np.abs(np.asarray(pairs_dev[name], np.float64) - pr_h[name]).max() / max(np.abs(pr_h[name]).max(), 1e-300)
        assert d < 1e-12, (name, d)
    # the trace itself: bit for bit the oracle's on those pairs and the device's own level 0
    grad_new = ctx.pyramid_get(501, 0)
    o = TS.oracle_trace(grad_new, pairs_dev, prm, pts.copy())
    for name in FIELDS:
        assert _bits_equal(o[name], after[name]), name
    assert np.array_equal(counts, np.bincount(o["last_status"], minlength=6)[:6])
    assert counts[abi.IPS_GOOD] > 20
    # ... and equal to the plain resident trace on the same pairs
    ctx.tracer_set_points(pts)
    c2 = ctx.tracer_trace_resident(501, prm, pairs_dev, -2)
    plain = ctx.tracer_get_points()
    for name in FIELDS:
        assert _bits_equal(plain[name], after[name]), name
    assert np.array_equal(c2, counts)
    # rollback: every traced point gets back what it held
    ctx.tracer_set_points(pts)
    ctx.tracker_optimize_batch_async(501, P.levels, W.K, P.ref_exp, P.init_exp, P.prm, hyps)
    ctx.tracer_trace_resident_tracked_async(501, prm, hosts, ref, W.K, skip_host=-2)
    ctx.tracker_optimize_wait()
    ctx.tracer_trace_resident_finish(keep=False)
    back = ctx.tracer_get_points()
    for name in FIELDS:
        assert _bits_equal(back[name], pts[name]), name


def test_pairs_formed_at_the_tracker_tail_equal_the_trace_launch_ones(scene):
    """cmlhip_tracer_tracked_prepare: the batch's launch carries the window and forms the pairs behind hypothesis 0 — same function, same bits as the
    trace launch forming them; a prepared window that is not the one the trace then passes is ignored"""
    P, ctx, pts, hosts, hyps = scene
    W, s = P.W, P.s
    prm = abi.default_tracer_params()
    ref = hosts[s.ref]

    def frame(prepare, trace_hosts=None):
        ctx.tracer_set_points(pts)
        if prepare is not None:
            ctx.tracer_tracked_prepare(prepare, ref, W.K)
        ctx.tracker_optimize_batch_async(501, P.levels, W.K, P.ref_exp, P.init_exp, P.prm, hyps)
        ctx.tracer_trace_resident_tracked_async(501, prm, trace_hosts or hosts, ref, W.K, skip_host=-2)
        res = ctx.tracker_optimize_wait()
        counts, pairs = ctx.tracer_trace_resident_finish(keep=True)
        return res, counts, pairs, ctx.tracer_get_points()
    r0, c0, p0, a0 = frame(None)
    r1, c1, p1, a1 = frame(hosts)
    assert _bits_equal(np.array(r0[0].R[:]), np.array(r1[0].R[:])) and _bits_equal(np.array(r0[0].t[:]), np.array(r1[0].t[:]))
    assert p0.tobytes() == p1.tobytes()
    assert np.array_equal(c0, c1)
    for name in FIELDS:
        assert _bits_equal(a0[name], a1[name]), name
    # a stale request (another window's poses): the trace forms its own pairs, as without it
    other = [(R, t + 0.01, a, b) for (R, t, a, b) in hosts]
    r2, c2, p2, a2 = frame(other)
    assert p0.tobytes() == p2.tobytes() and np.array_equal(c0, c2)
    for name in FIELDS:
        assert _bits_equal(a0[name], a2[name]), name
    # the request is one shot: a second batch without prepare does not reuse it
    r3, c3, p3, a3 = frame(None)
    assert p0.tobytes() == p3.tobytes() and np.array_equal(c0, c3)


def test_host_mirror_fused_call_equals_the_two_calls(scene):
    P, ctx, pts, hosts, hyps = scene
    W, s = P.W, P.s
    fids = list(range(s.new))

    def fresh():
        trk = host.HostTracker(ctx); trk.set_calibration(*W.K)
        trc = host.HostTracer(ctx)
        for i in range(len(pts)):
            trc.add_point(pts["x"][i], pts["y"][i], int(pts["host"][i]), pts["gray"][i], pts["dpatch"][i], pts["gradH"][i])
        return trk, trc
    trk, trc = fresh()
    res, kept, counts, pairs = trk.track_and_trace(trc, 501, P.levels, hyps, P.ref_exp, P.init_exp, -1, fids, hosts, s.ref, W.K)
    assert res["haveOneGood"] and res["winner"] == 0 and kept
    a_pts = trc.points()[0]
    trk.close(); trc.close()
    trk, trc = fresh()
    res2 = trk.track_with_motion_model(501, P.levels, hyps, P.ref_exp, P.init_exp, batched=True)
    assert np.array_equal(res["R"].view(np.uint64), res2["R"].view(np.uint64)) and np.array_equal(res["t"].view(np.uint64), res2["t"].view(np.uint64))
    assert res["winner"] == res2["winner"] and res["tries"] == res2["tries"]
    c2 = trc.trace_new_coarse(501, -1, fids, pairs)
    b_pts = trc.points()[0]
    for name in FIELDS:
        assert _bits_equal(a_pts[name], b_pts[name]), name
    assert np.array_equal(counts, c2)
    trk.close(); trc.close()
    # a batch whose first hypothesis is far off: another try wins, the library rolls the trace back (kept = False) and the points are untouched
    trk, trc = fresh()
    bad_first = [TO.perturbed(P, (0.3, -0.25, 0.2), (2.0, -1.5, 1.0))] + hyps
    res3, kept3, _c, _p = trk.track_and_trace(trc, 501, P.levels, bad_first, P.ref_exp, P.init_exp, -1, fids, hosts, s.ref, W.K)
    if res3["haveOneGood"] and res3["winner"] != 0:
        assert not kept3
        u = trc.points()[0]
        for name in FIELDS:
            assert _bits_equal(u[name], pts[name]), name
    trk.close(); trc.close()


def test_histogram_survives_an_activation_between_two_frames(scene):
    """the activation's results share a scratch buffer with the plain resident trace: the tracked trace keeps its status histogram in one of its own
    (the second frame's counts must not start from another call's leftovers); and the resident-slot activation equals the record one"""
    P, ctx, pts, hosts, hyps = scene
    W, s = P.W, P.s
    prm = abi.default_tracer_params()
    ref = hosts[s.ref]

    def frame():
        ctx.tracker_optimize_batch_async(501, P.levels, W.K, P.ref_exp, P.init_exp, P.prm, hyps)
        ctx.tracer_trace_resident_tracked_async(501, prm, hosts, ref, W.K, skip_host=-2)
        ctx.tracker_optimize_wait()
        return ctx.tracer_trace_resident_finish(keep=True)
    ctx.tracer_set_points(pts)
    c1, _ = frame()
    cur = ctx.tracer_get_points()
    assert int(c1.sum()) == len(pts)
    # an activation over the first frames of the window (images of the hosts), by records and by resident slots
    ids = [900 + k for k in range(s.new)]
    for k in range(s.new):
        ctx.pyramid_build(ids[k], W.gray[k], 1)
    cand = np.flatnonzero(np.isfinite(cur["idepth_max"]) & (cur["last_status"] != abi.IPS_OOB))[:200]
    assert len(cand) > 20
    apr = TS.activation_pairs(W)
    N = W.N
    sub = np.zeros(s.new * s.new, abi.ACTIVATION_PAIR_DTYPE)
    for h in range(s.new):
        sub[h * s.new:(h + 1) * s.new] = apr[h * N:h * N + s.new]
    ra, ia, sa = ctx.optimize_immature_points(ids, W.K, sub, prm, 1, cur[cand])
    rb, ib, sb = ctx.optimize_immature_points_resident(ids, W.K, sub, prm, 1, cand)
    assert np.array_equal(ra, rb) and np.array_equal(ia.view(np.uint32), ib.view(np.uint32)) and np.array_equal(sa, sb)
    c2, _ = frame()
    assert int(c2.sum()) == len(pts), (c1, c2)
    for k in ids:
        ctx.pyramid_drop(k)
