"""Edge cases of the BA path on the device against the oracle: the smallest window, a window without residuals, residuals
that all leave the image, ragged points (some without any residual), non-finite inputs, and the limits of the layer."""
import numpy as np
import pytest

from libcml_amd import abi, device
from tests import ba_setup as S
from tests import dev_setup as D

pytestmark = pytest.mark.gpu


def _pipeline(I):
    """linearize -> apply -> accumulate -> solve -> backsub on both sides; returns what to compare."""
    ob = S.OracleBA(I)
    ctx = D.make_ctx(I)
    try:
        ro, rd = ob.linearize(), ctx.ba_linearize()
        so, sd = ob.states(), ctx.ba_states()
        assert np.array_equal(so["new_state"], sd["new_state"]) and np.array_equal(so["state"], sd["state"])
        assert (ro.n_in, ro.n_oob, ro.n_outlier) == (rd.n_in, rd.n_oob, rd.n_outlier)
        assert np.float32(ro.new_frame_energy_th).view(np.uint32) == np.float32(rd.new_frame_energy_th).view(np.uint32)
        IN = so["new_state"] == 0
        if IN.any():
            assert np.array_equal(ob.rJ(0)[IN].view(np.uint32), ctx.ba_rj(0)[IN].view(np.uint32))
        ob.apply(1); ctx.ba_apply(1)
        Ho = ob.accumulate(); Hd = D.accumulate(ctx, I)
        for a, b in zip(Ho, Hd):
            assert np.all(np.isfinite(b))
            assert np.abs(a - b).max() <= 5e-5 * max(np.abs(a).max(), 1e-30)
        xd, rc = ctx.ba_solve(1e-5)
        xo, rco = ob.solve(1e-5, *Hd)
        assert rc == 0 and rco == 0 and np.all(np.isfinite(xd))
        assert np.abs(xd - xo).max() <= 1e-7 * max(np.abs(xo).max(), 1e-30)
        sto, _ = ob.backsub(xd)
        std, rcb = ctx.ba_backsub(xd)
        assert rcb == 0 and np.abs(sto - std).max() <= 5e-5 * max(np.abs(sto).max(), 1e-30)
        return ro, rd, so
    finally:
        ctx.close()


def test_smallest_window():
    ro, rd, so = _pipeline(S.make_inputs((2, 3, 160, 120, 2, 140.0, 140.0, 79.5, 59.5)))
    assert ro.n_in + ro.n_oob + ro.n_outlier == 3


def test_all_residuals_out_of_bounds():
    I = S.make_inputs("tiny")
    I.residuals["state"][:] = abi.RES_OOB if hasattr(abi, "RES_OOB") else 1          # DSOResidualState::DSORES_OOB
    ro, rd, so = _pipeline(I)
    assert rd.n_in == 0 and rd.energy == 0
    assert rd.new_frame_energy_th == np.float32(12 * 12 * 8)                        # BA.cpp:2432-2436


def test_ragged_points_and_empty_pairs():
    """Half of the points keep a single residual, a quarter none at all; several (host,target) pairs end up empty."""
    I = S.make_inputs("small")
    keep = np.ones(I.R, bool)
    pt = I.residuals["point"]
    first_of_point = np.r_[True, pt[1:] != pt[:-1]]
    keep[(pt % 4 == 1) & ~first_of_point] = False        # a single residual
    keep[pt % 4 == 2] = False                            # none
    I.residuals = I.residuals[keep].copy(); I.R = int(keep.sum())
    ro, rd, so = _pipeline(I)
    assert rd.n_in > 10


def test_window_without_residuals():
    I = S.make_inputs("tiny")
    I.residuals = I.residuals[:0].copy(); I.R = 0
    ctx = D.make_ctx(I)
    try:
        r = ctx.ba_linearize()
        assert (r.n_in, r.n_oob, r.n_outlier) == (0, 0, 0) and r.energy == 0
        ctx.ba_apply(1)
        HA, bA, HL, bL, Hsc, bsc = D.accumulate(ctx, I)
        assert not HA.any() and not Hsc.any() and not bA.any() and not bsc.any()
        assert np.allclose(np.diag(HL)[4:], I.prior)
        x, rc = ctx.ba_solve(1e-5)
        assert rc == 0 and np.all(np.isfinite(x))
    finally:
        ctx.close()


def test_non_finite_point_is_contained():
    """A NaN inverse depth poisons exactly its own residuals (BA.cpp:115-118,297-300: new state OOB), nothing else, on both sides."""
    I = S.make_inputs("small")
    I.points["idepth"][5] = np.nan
    ob = S.OracleBA(I)
    ctx = D.make_ctx(I)
    try:
        ro, rd = ob.linearize(), ctx.ba_linearize()
        so, sd = ob.states(), ctx.ba_states()
        assert np.array_equal(so["new_state"], sd["new_state"]) and np.array_equal(so["state"], sd["state"])
        mine = I.residuals["point"] == 5
        assert np.all(sd["new_state"][mine] != 0) and (sd["new_state"][~mine] == 0).sum() > 50
        assert (ro.n_in, ro.n_oob, ro.n_outlier) == (rd.n_in, rd.n_oob, rd.n_outlier)
        assert np.isfinite(rd.energy) and abs(ro.energy - rd.energy) <= 1e-12 * abs(ro.energy)
    finally:
        ctx.close()


def test_limits_are_errors_not_crashes():
    L = device.lib()
    with pytest.raises(device.CmlHipError):
        device.Ctx(max_frames=abi.MAX_FRAMES + 1 if hasattr(abi, "MAX_FRAMES") else 33)
    I = S.make_inputs("tiny")
    ctx = device.Ctx(max_frames=2, max_points=I.P, max_residuals=I.R)          # the window has 3 frames
    try:
        for k in range(I.N):
            ctx.pyramid_put(int(I.frames_dev["image_id"][k]), 0, I.grads[k][0])
        ctx.ba_set_params(I.prm)
        with pytest.raises(device.CmlHipError):
            ctx.ba_upload_window(I.frames_dev, I.points, I.residuals)
        with pytest.raises(device.CmlHipError):
            ctx.ba_linearize()                                                  # nothing uploaded: call-order error
    finally:
        ctx.close()
    # a window wider than the LDS-resident solver takes (N > 20) is not refused any more: it factorises in global memory
    # (tests/test_ba_parity_gpu.py::test_system_and_solver_wide_window checks its result); the hybrid term and the batched
    # iteration stay limited to 20 frames and say so
    N = 22
    I = S.make_inputs((N, 60, 160, 120, 2, 140.0, 140.0, 79.5, 59.5))
    ctx = D.make_ctx(I)
    try:
        ctx.ba_linearize(); ctx.ba_apply(1)
        Hd = D.accumulate(ctx, I)
        assert all(np.all(np.isfinite(h)) for h in Hd)
        x, rc = ctx.ba_solve(1e-5)
        assert rc == 0 and np.all(np.isfinite(x))
    finally:
        ctx.close()


def test_tracer_edge_cases():
    """Empty immature set, a set that is entirely out of bounds, and a NaN interval: device == oracle, no hang."""
    from libcml_amd import synth
    from tests import oracle_lib as O
    from tests import tracer_setup as TS
    W = synth.make_window("tiny")
    grads0 = [O.build_pyramid(W.gray[k], 1)[1][0] for k in range(W.N)]
    ctx = device.Ctx(max_frames=W.N)
    try:
        ids = [300 + k for k in range(W.N)]
        for k in range(W.N):
            ctx.pyramid_put(ids[k], 0, grads0[k])
        prm = abi.default_tracer_params()
        pr = TS.trace_pairs(W, 1)
        pts = TS.make_immature(W, grads0)
        assert len(ctx.trace_points(ids[1], prm, pr, pts[:0])) == 0
        ctx.tracer_set_points(pts[:0])
        assert ctx.tracer_trace_resident(ids[1], prm, pr, 1).sum() == 0
        sel = pts[pts["host"] == 0].copy()
        far = sel.copy(); far["x"] += 10000                          # projects far outside: OOB at the first test (DSOTracer.cpp:620-626)
        o = TS.oracle_trace(grads0[1], pr, prm, far); d = ctx.trace_points(ids[1], prm, pr, far)
        assert np.all(o["last_status"] == abi.IPS_OOB) and np.array_equal(o["last_status"], d["last_status"])
        bad = sel.copy(); bad["idepth_min"][::2] = np.nan
        o = TS.oracle_trace(grads0[1], pr, prm, bad); d = ctx.trace_points(ids[1], prm, pr, bad)
        assert np.array_equal(o["last_status"], d["last_status"])
        ok = np.isfinite(o["idepth_min"])
        assert np.array_equal(o["idepth_min"][ok].view(np.uint64), d["idepth_min"][ok].view(np.uint64))
        # activation with a single other frame and with every residual out of bounds
        apr = TS.activation_pairs(W)
        cand = sel.copy(); cand["idepth_min"] = 0.05; cand["idepth_max"] = 0.2
        ro, io, so = TS.oracle_optimize(grads0, W.K, apr, prm, 1, cand)
        rd, idd, sd = ctx.optimize_immature_points(ids, W.K, apr, prm, 1, cand)
        assert np.array_equal(ro, rd) and np.array_equal(io.view(np.uint32), idd.view(np.uint32))
        cand["x"] += 10000
        ro, io, so = TS.oracle_optimize(grads0, W.K, apr, prm, 1, cand)
        rd, idd, sd = ctx.optimize_immature_points(ids, W.K, apr, prm, 1, cand)
        assert np.array_equal(ro, rd) and np.all(rd != 1)
    finally:
        ctx.close()


def test_host_mirror_error_returns():
    """The reference's `return false` paths: no points (BA.cpp:759-762) and a missing calibration."""
    from libcml_amd import host
    ctx = device.Ctx(max_frames=4)
    ba = host.HostBA(ctx)
    try:
        ba.set_calibration(140.0, 140.0, 79.5, 59.5, 160, 120)
        assert not ba.run() and "No points" in ba.last_error()
        assert not ba.run_host_loop()
        with pytest.raises(KeyError):
            ba.set_param("no such parameter", 1.0)
    finally:
        ba.close(); ctx.close()


def test_reproj_without_observations():
    ctx = device.Ctx(max_frames=4)
    try:
        poses = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0]), (3, 1))
        pts = np.zeros((5, 3)); pts[:, 2] = 4.0
        obs = np.zeros(0, abi.REPROJ_OBS_DTYPE) if hasattr(abi, "REPROJ_OBS_DTYPE") else np.zeros(0, np.dtype([("frame", "i4"), ("point", "i4"), ("u", "f8"), ("v", "f8")]))
        M6, b6, Jp, used = ctx.reproj_accumulate(poses, pts, obs, 500.0, 500.0)
        assert not M6.any() and not b6.any() and len(used) == 0
    finally:
        ctx.close()


def test_window_is_invalidated_when_its_images_go_away():
    """An uploaded window points into the level-0 images it names: rebuilding, resizing or dropping one of them must make the BA entry
    points fail with CMLHIP_ERR_STATE (round 3, ADVICE: they used to read the recycled blocks), a same-size put must not."""
    I = S.make_inputs("tiny")
    ctx = D.make_ctx(I)
    try:
        r0 = ctx.ba_linearize()
        img = int(I.frames_dev["image_id"][1])
        ctx.pyramid_put(img, 0, I.grads[1][0])                    # same size: texels rewritten in place, the window stays valid
        r1 = ctx.ba_linearize()
        assert r1.n_in > 0
        pairs, th, b0 = ctx.ba_pairs()                              # (the readback the replay checker uses)
        assert np.array_equal(pairs["R"], I.pairs["R"]) and np.array_equal(b0, I.frames_dev["b0"])
        ctx.pyramid_drop(img)
        with pytest.raises(device.CmlHipError) as e:
            ctx.ba_linearize()
        assert e.value.code == abi.ERR_STATE
        ctx.pyramid_put(img, 0, I.grads[1][0])
        with pytest.raises(device.CmlHipError):                     # still invalid until the window is uploaded again
            ctx.ba_linearize()
        ctx.ba_upload_window(I.frames_dev, I.points, I.residuals)
        ctx.ba_set_pairs(I.pairs)
        r2 = ctx.ba_linearize()
        assert (r2.n_in, r2.n_oob, r2.n_outlier) == (r0.n_in, r0.n_oob, r0.n_outlier)        # a fresh upload: the first pass again
        ctx.pyramid_build(img, I.W.gray[1], 1)                      # rebuilt from gray: new blocks
        with pytest.raises(device.CmlHipError):
            ctx.ba_linearize()
    finally:
        ctx.close()


def _resident_window(config="small", shard=0):
    from libcml_amd import host, synth
    W = synth.make_window(config, shard=shard)
    ctx = device.Ctx(max_frames=max(W.N, 2), max_points=W.P, max_residuals=W.P * W.N)
    ba = host.window_to_host_ba(ctx, W, image_id_base=100 * (shard + 1), levels=1)
    ba.set_param("iterations", 1)
    assert ba.run(), ba.last_error()
    ctx.refresh_window_size()
    assert ba.begin_resident(), ba.last_error()
    return W, ctx, ba


def _snapshot(ctx):
    st = ctx.ba_states()
    fs, pre = ctx.ba_resident_state()
    return (st["state"].tobytes(), st["energy"].tobytes(), ctx.ba_get_idepth().tobytes(), ctx.ba_jpjdf().tobytes(), bytes(fs), pre.tobytes())


def test_resident_prior_edge_cases():
    """cmlhip_ba_set_resident_prior (round 4): refused before the resident state exists; NULL switches it off; an all-zero prior is the
    prior-free iteration in every bit ((bL + 0) + bA and (HL + 0) + HA are exact); a real prior changes the step; cmlhip_ba_set_resident_state
    switches it off again."""
    import ctypes as C
    W, ctx, ba = _resident_window()
    n = 8 * W.N + 4
    try:
        base = []
        for _ in range(3):
            ctx.ba_iteration_async(1e-5)
        base = _snapshot(ctx)
    finally:
        ba.close(); ctx.close()
    W, ctx, ba = _resident_window()
    try:
        Z = np.zeros((n, n)); z = np.zeros(n)
        ctx.ck(ctx.L.cmlhip_ba_set_resident_prior(ctx.h, Z.ctypes.data_as(C.POINTER(C.c_double)), z.ctypes.data_as(C.POINTER(C.c_double))))
        for _ in range(3):
            ctx.ba_iteration_async(1e-5)
        assert _snapshot(ctx) == base, "a zero prior must not change a bit"
    finally:
        ba.close(); ctx.close()
    W, ctx, ba = _resident_window()
    try:
        rng = np.random.default_rng(5)
        Q = rng.standard_normal((n, n)) * 300.0
        HM = Q @ Q.T; bM = rng.standard_normal(n) * 1e3
        ctx.ck(ctx.L.cmlhip_ba_set_resident_prior(ctx.h, HM.ctypes.data_as(C.POINTER(C.c_double)), bM.ctypes.data_as(C.POINTER(C.c_double))))
        ctx.ck(ctx.L.cmlhip_ba_set_resident_prior(ctx.h, None, None))                       # ... and off again
        for _ in range(3):
            ctx.ba_iteration_async(1e-5)
        assert _snapshot(ctx) == base, "NULL must switch the prior off"
        ctx.ck(ctx.L.cmlhip_ba_set_resident_prior(ctx.h, HM.ctypes.data_as(C.POINTER(C.c_double)), bM.ctypes.data_as(C.POINTER(C.c_double))))
        ctx.ba_iteration_async(1e-5)
        assert _snapshot(ctx) != base
    finally:
        ba.close(); ctx.close()
    # call order: the resident state first
    I = S.make_inputs("tiny")
    ctx = D.make_ctx(I)
    try:
        Z = np.zeros((8 * I.N + 4, 8 * I.N + 4)); z = np.zeros(8 * I.N + 4)
        rc = ctx.L.cmlhip_ba_set_resident_prior(ctx.h, Z.ctypes.data_as(C.POINTER(C.c_double)), z.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == abi.ERR_STATE
        rc = ctx.L.cmlhip_ba_set_frame_b0(ctx.h, None)
        assert rc == abi.ERR_INVALID
    finally:
        ctx.close()
    c2 = device.Ctx(max_frames=4, max_points=10, max_residuals=10)
    try:
        b0 = np.zeros(4, np.float32)
        assert c2.L.cmlhip_ba_set_frame_b0(c2.h, b0.ctypes.data_as(C.POINTER(C.c_float))) == abi.ERR_STATE      # no window uploaded
    finally:
        c2.close()
    assert abi.ERR_TIMEOUT == 6


def test_closing_pass_retires_what_it_removes():
    """cmlhip_ba_finish_keyframe = linearizeAll(true): every active residual that is not good afterwards is removed (BA.cpp:1595-1598) — on the
    device too (state OOB, absorbing), so that tryMarginalize's pass over a point cannot revive it (found by tests/test_sequence_gpu.py)."""
    from libcml_amd import host, synth
    W = synth.make_window("medium", idepth_noise=0.2)            # poor depths: a good share of the residuals ends OUTLIER / OOB
    ctx = device.Ctx(max_frames=W.N, max_points=W.P, max_residuals=W.P * W.N)
    ba = host.window_to_host_ba(ctx, W)
    try:
        assert ba.run(), ba.last_error()
        st, alive, good = ba.residual_states()
        dead = alive == 0
        assert dead.sum() > 20 and (alive == 1).sum() > 100
        ctx.refresh_window_size()
        dev = ctx.ba_states()
        assert np.all(dev["state"][dead] == 1) and np.all(dev["good"][dead] == 0)          # retired = OOB on the device
        # a pass over EVERY point (what tryMarginalize does for its candidates): the retired residuals stay out
        A = ba.algebra()
        ain = (A["adH"], A["adT"], A["adHTd"], np.zeros(4), A["prior"], A["dprior"], np.full(4, 5e9))
        ng = ctx.ba_relinearize_points(np.arange(ctx.P, dtype=np.int32), *ain)
        dev2 = ctx.ba_states()
        assert np.all(dev2["state"][dead] == 1) and np.all(dev2["good"][dead] == 0)
        assert ng == int(dev2["good"].sum()) and ng <= int((alive == 1).sum())
    finally:
        ba.close(); ctx.close()
