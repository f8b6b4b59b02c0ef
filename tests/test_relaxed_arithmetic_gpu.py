"""CMLHIP_ARITH_RELAXED (cmlhip_ba_set_arithmetic, include/cmlhip.h) against the oracle — the opt-in arithmetic of the resident
residual kernels (k_ba_lin_rs<..., RELAX = true>: ba_linearize_rs_body.inc; k_ba_lin_rs4*<..., RELAX = true>: ba_linearize_rs4.hip): fused multiply-adds, one Newton step on the projection's
reciprocal, the photometric terms and pattern sums of a pixel in fp32.  It is NOT bit-exact; this file states what it is instead, in the
terms SURVEY §7 uses for a kernel that does not reproduce the reference's rounding: per-residual energies within 1e-4 relative of the
oracle's (observed 2e-7), Jacobian products JpJdF within 1e-4 of the row's largest entry for 99.9 % of the residuals (median below 1e-6; rows
whose two terms cancel reach a few 1e-4: bar 1e-3) (the oracle = the contraction-free statement-for-statement reading, DSOBundleAdjustment.cpp:62-316), the
classification (IN / OOB / OUTLIER, isActiveAndIsGoodNEW) identical except for a REPORTED and bounded count of residuals that sit on a
threshold, and the loop it drives converging to the same window.  The exact mode stays the default and the regression instrument
(tests/test_resident_oracle_gpu.py: every bit)."""
import os

import numpy as np
import pytest

from libcml_amd import abi, device, host, synth
from tests import resident_check as RC

pytestmark = pytest.mark.gpu

REL = 1e-4            # the bar; observed values are asserted an order of magnitude below it where they are stable


def _window(config, relaxed, force_tile64=False):
    W = synth.make_window(config)
    half = config == "E"
    old = os.environ.get("CMLHIP_RS_TILE")
    if force_tile64:
        os.environ["CMLHIP_RS_TILE"] = "64"          # (read at every upload: the lane-per-residual kernel for a window below its regime)
    try:
        ctx = device.Ctx(max_frames=max(W.N, 2), max_points=W.P, max_residuals=W.P * W.N,
                         texel_format=abi.TEXEL_F16 if half else abi.TEXEL_F32)
        ba = host.window_to_host_ba(ctx, W, image_id_base=7000, levels=1)
        ba.set_param("iterations", 1)
        assert ba.run(), ba.last_error()
        ctx.refresh_window_size()
        assert ba.begin_resident(), ba.last_error()
    finally:
        if force_tile64:
            if old is None:
                os.environ.pop("CMLHIP_RS_TILE", None)
            else:
                os.environ["CMLHIP_RS_TILE"] = old
    ctx.ba_set_arithmetic(relaxed)
    return W, ctx, ba


@pytest.mark.parametrize("config,force", [("E", False), ("B", True), ("B", False), ("medium", False)])      # lane-per-residual kernel (E; B forced into it), 4-lane kernel (B, medium)
def test_relaxed_pass_against_the_oracle(config, force):
    """each relaxed residual pass replayed on the oracle FROM THE DEVICE'S OWN STATE (so differences do not accumulate across passes)"""
    W, ctx, ba = _window(config, True, force)
    replay = RC.make_replay(ctx, ba, W)
    try:
        worst = {}
        for it in range(5):
            ctx.sync()
            pre = ctx.ba_states()
            ctx.ba_iteration_async(1e-5)
            ctx.sync()
            rep = RC.compare_pass_tolerant(ctx, replay, pre)
            for k, v in rep.items():
                worst[k] = max(worst.get(k, 0), v)
            # classification: identical except residuals whose energy sits on the outlier threshold / whose projection sits on the image border
            assert rep["new_state_flips"] <= max(2, replay.R // 5000), rep
            assert rep["state_flips"] <= max(2, replay.R // 5000) and rep["good_flips"] <= max(2, replay.R // 5000), rep
            assert rep["energy_rel"] < REL and rep["new_energy_rel"] < REL and rep["new_energy_wo_rel"] < REL, rep
            assert rep["jpjdf_rel_median"] < 1e-6 and rep["jpjdf_rel_p999"] < REL and rep["jpjdf_rel"] < 1e-3, rep      # (worst row: cancelling terms, see compare_pass_tolerant)
            assert rep["center_abs"] < 1e-3, rep                 # pixels (fp32 of an fp64 projection that differs in its last bits)
            assert rep["n_in"] > 0.5 * replay.R, rep
        print("relaxed arithmetic, config %s, worst over 5 passes: %s" % (config, worst))
        assert worst["new_energy_rel"] < 5e-6, worst                # where the energies actually are (fp32 sums of 8 terms: observed 2e-7)
    finally:
        replay.close(); ba.close(); ctx.close()


def test_relaxed_loop_converges_to_the_exact_loop():
    """the same window iterated in both modes: frame states and the energy after 6 iterations agree far inside the bars of the host-mirror test"""
    res = []
    for relaxed in (False, True):
        W, ctx, ba = _window("E", relaxed)
        try:
            for _ in range(6):
                ctx.ba_iteration_async(1e-5)
            ctx.sync()
            fs, pre = ctx.ba_resident_state()
            st = ctx.ba_states()
            res.append((np.array(pre), st["energy"].astype(np.float64).sum(), st["state"].copy(), ctx.ba_get_idepth()))
        finally:
            ba.close(); ctx.close()
    (p0, e0, s0, d0), (p1, e1, s1, d1) = res
    assert np.abs(p0 - p1).max() < 1e-5, np.abs(p0 - p1).max()              # quaternion / translation entries of PRE_worldToCam (observed 1.4e-6; the host-mirror test's bar against the ORACLE loop is 1e-3)
    assert abs(e0 - e1) <= 1e-4 * abs(e0), (e0, e1)
    assert (s0 != s1).sum() <= max(4, len(s0) // 2000), (s0 != s1).sum()
    assert np.percentile(np.abs(d0 - d1) / np.maximum(np.abs(d0), 1e-6), 99) < 1e-4


def test_arithmetic_mode_is_validated_and_batch_refuses_mixed_modes():
    W, ctx, ba = _window("small", False)
    W2, ctx2, ba2 = _window("small", True)
    try:
        assert ctx.L.cmlhip_ba_set_arithmetic(ctx.h, 2) == abi.ERR_INVALID
        assert ctx.L.cmlhip_ba_set_arithmetic(ctx.h, 1) == abi.OK and ctx.L.cmlhip_ba_set_arithmetic(ctx.h, 0) == abi.OK
        with pytest.raises(device.CmlHipError) as e:                # one exact, one relaxed window in a batch
            device.ba_iteration_batch([ctx, ctx2], 1e-5)
        assert e.value.code == abi.ERR_INVALID and "arithmetic" in str(e.value)
        ctx.ba_set_arithmetic(True)
        device.ba_iteration_batch([ctx, ctx2], 1e-5)               # both relaxed: accepted
        ctx.sync()
    finally:
        ba.close(); ctx.close(); ba2.close(); ctx2.close()


def test_relaxed_batch_equals_relaxed_solo():
    """the relaxed mode keeps the batched launch's property: a window stepped in a batch ends bit-identical to the same window stepped alone"""
    outs = []
    for batched in (False, True):
        Ws = [_window("small", True) for _ in range(3)]
        try:
            ctxs = [w[1] for w in Ws]
            for _ in range(4):
                if batched:
                    device.ba_iteration_batch(ctxs, 1e-5)
                else:
                    for c in ctxs:
                        c.ba_iteration_async(1e-5)
            for c in ctxs:
                c.sync()
            outs.append([(c.ba_states()["energy"].tobytes(), c.ba_get_idepth().tobytes(), c.ba_jpjdf().tobytes()) for c in ctxs])
        finally:
            for w in Ws:
                w[2].close(); w[1].close()
    assert outs[0] == outs[1]


def test_host_mirror_run_with_the_relaxed_parameter():
    """cml_amd::DSOBundleAdjustment::run with `relaxedArithmetic` (the mirror's switch for cmlhip_ba_set_arithmetic): same iterations, energies and poses
    as the exact run (energies 1e-4, poses 2e-4)"""
    res = []
    for relaxed in (0, 1):
        W = synth.make_window("medium")
        ctx = device.Ctx(max_frames=W.N, max_points=W.P, max_residuals=W.P * W.N)
        ba = host.window_to_host_ba(ctx, W, image_id_base=7100, levels=1)
        try:
            ba.set_param("iterations", 4)
            ba.set_param("relaxedArithmetic", relaxed)
            assert ba.run(), ba.last_error()
            res.append((ba.energies(16).copy(), ba.counts()["iterations"], np.concatenate([np.r_[ba.frame(i)["R"].ravel(), ba.frame(i)["t"]] for i in range(W.N)])))
        finally:
            ba.close(); ctx.close()
    (e0, it0, p0), (e1, it1, p1) = res
    assert it0 == it1 and len(e0) == len(e1)
    assert np.all(np.abs(e0 - e1) <= 1e-4 * np.abs(e0)), (e0, e1)
    assert np.abs(p0 - p1).max() < 2e-4, np.abs(p0 - p1).max()      # (observed 3e-5 in the translation of the last frame of this 2 400-residual window; the bars of the exact run against the ORACLE loop are 1e-3 / 5e-3)


def test_relaxed_mode_contains_a_non_finite_point_like_the_exact_mode():
    """a NaN inverse depth poisons exactly its own residuals (new state OOB, BA.cpp:115-118) and nothing else — in both arithmetic modes, with the same states"""
    out = []
    for relaxed in (False, True):
        W, ctx, ba = _window("small", relaxed)
        try:
            idp = ctx.ba_get_idepth().copy()
            idp[5] = np.nan
            ctx.ba_set_idepth(idp)
            for _ in range(2):
                ctx.ba_iteration_async(1e-5)
            ctx.sync()
            st = ctx.ba_states()
            out.append((st["state"].copy(), st["new_state"].copy(), st["good"].copy(), st["energy"].copy()))
        finally:
            ba.close(); ctx.close()
    (s0, n0, g0, e0), (s1, n1, g1, e1) = out
    assert np.array_equal(s0, s1) and np.array_equal(n0, n1) and np.array_equal(g0, g1)
    assert (n0 != 0).sum() >= 1 and (n0 == 0).sum() > 50
    ok = g0 == 1
    # (this 628-residual window amplifies a 1e-6 difference of the first step to 2e-4 in pose after two: the energies are compared loosely, the statement is containment)
    assert np.all(np.isfinite(e1[ok])) and np.abs(e0[ok] - e1[ok]).max() <= 1e-2 * max(np.abs(e0[ok]).max(), 1.0)


def test_run_preamble_stays_exact_under_the_relaxed_flag():
    """include/cmlhip.h: only cmlhip_ba_iteration_async / _batch honour CMLHIP_ARITH_RELAXED.  cmlhip_ba_linearize_apply — run()'s preamble, which commits states
    and frameEnergyTH through applyRes(true) — goes through the resident residual kernel when the window is armed; with the sticky flag set it must still be the
    exact kernel: same energy bits, same states, same thresholds as on a context that never saw the flag (round-4 advisor finding)."""
    import ctypes as C
    outs = []
    for relaxed in (False, True):
        W, ctx, ba = _window("medium", relaxed)
        try:
            lr = abi.BALinResult()
            ctx.ck(ctx.L.cmlhip_ba_linearize_apply(ctx.h, C.byref(lr)))
            st = ctx.ba_states()
            outs.append((np.float64(lr.energy).view(np.uint64), lr.n_in, lr.n_oob, lr.n_outlier, np.float32(lr.new_frame_energy_th).view(np.uint32),
                         st["state"].copy(), st["energy"].view(np.uint32).copy(), st["good"].copy()))
        finally:
            ba.close(); ctx.close()
    a, b = outs
    assert a[:4] == b[:4], (a[:4], b[:4])
    for x, y in zip(a[5:], b[5:]):
        assert np.array_equal(x, y)
