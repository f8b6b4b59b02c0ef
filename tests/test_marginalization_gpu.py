"""SURVEY §8 a15 on the device, through the C ABI, against the oracle: tryMarginalize's residual loop (resetOOB,
linearize, applyRes, fixLinearization), MARGINALIZED-mode accumulation (marginalizePointsF), the linearized energy, and
the regular accumulation once LINEARIZED residuals exist.  Bars: states / records / res_toZero bit-exact (same statement
order, fp contraction off); sums over residuals at fp32 accumulation-order tolerance."""
import numpy as np
import pytest

from tests import ba_setup as S
from tests import dev_setup as D

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config", ["small", "medium"])
def test_marginalize_points_path(config):
    I = S.make_inputs(config, state_noise=0.3)          # state != state_zero: J*delta is exercised
    ob = S.OracleBA(I)
    ctx = D.make_ctx(I)
    try:
        ob.linearize(); ctx.ba_linearize()
        ob.apply(1); ctx.ba_apply(1)
        ain = (I.adH, I.adT, I.adHTd, I.cdelta, I.prior, I.dprior, I.cprior)
        sel = np.arange(1, I.P, 3, dtype=np.int32)
        # ---- tryMarginalize residual loop
        ngo = ob.relinearize_points(sel)
        ngd = ctx.ba_relinearize_points(sel, *ain)
        assert ngo == ngd and ngo > 30
        so, sd = ob.states(), ctx.ba_states()
        for k in ("state", "new_state", "good"):
            assert np.array_equal(so[k], sd[k]), k
        assert np.array_equal(so["energy"].view(np.uint32), sd["energy"].view(np.uint32))
        rtz_o = ob.view("res_toZeroF", 8 * I.R, np.float32).reshape(-1, 8).copy()
        lin_o = ob.view("r_lin", I.R, np.uint8).copy()
        rtz_d, lin_d = ctx.ba_res_to_zero()
        assert np.array_equal(lin_o, lin_d) and lin_o.sum() == ngo
        L = lin_o == 1
        assert np.array_equal(rtz_o[L].view(np.uint32), rtz_d[L].view(np.uint32)), "res_toZero differs"
        g = so["good"] == 1
        assert np.array_equal(ob.rJ(1)[g].view(np.uint32), ctx.ba_rj(1)[g].view(np.uint32))
        # ---- linearized energy (calcLEnergy)
        eo, no = ob.l_energy()
        ed, nd = ctx.ba_lin_energy(*ain)
        assert no == nd == ngo
        assert abs(ed - eo) <= 1e-5 * abs(eo)
        # ---- the regular accumulation now has ACTIVE and LINEARIZED parts
        HAo, bAo, HLo, bLo, Hso, bso = ob.accumulate()
        HAd, bAd, HLd, bLd, Hsd, bsd = D.accumulate(ctx, I)
        assert D.rel(HAd, HAo) < 2e-5 and D.rel(bAd, bAo) < 2e-5
        assert D.rel(HLd, HLo) < 2e-5 and D.rel(bLd, bLo) < 5e-5
        assert D.rel(Hsd, Hso) < 5e-5 and D.rel(bsd, bso) < 1e-4
        # ---- marginalizePointsF
        Mo, Mbo, Msco, Mbsco = ob.marginalize_points(sel)
        Md, Mbd, Mscd, Mbscd = ctx.ba_marginalize_points(sel, *ain)
        assert D.rel(Md, Mo) < 2e-5 and D.rel(Mbd, Mbo) < 5e-5
        assert D.rel(Mscd, Msco) < 5e-5 and D.rel(Mbscd, Mbsco) < 1e-4
        assert np.abs(Md - Md.T).max() <= 1e-9 * np.abs(Md).max()
        # nothing outside the selection contributes: an empty selection gives zero blocks
        Z = ctx.ba_marginalize_points(np.zeros(0, np.int32), *ain)
        assert all(np.abs(z).max() == 0 for z in Z)
    finally:
        ctx.close()


def test_host_marginalisation_flow():
    """The keyframe epilogue of direct/Mapping.cpp:89-100 through the host mirror: run -> flag -> tryMarginalize ->
    marginalizePointsF -> marginalizeFrames -> run again with the prior.  The device pieces are pinned above; here the host
    plumbing (slot mapping, prior growth/permutation, frame renumbering) is checked against the oracle's dense algebra."""
    from libcml_amd import device, host
    from tests import oracle_lib as O
    I = S.make_inputs("medium")
    ctx = device.Ctx(max_frames=I.N + 1, max_points=I.P, max_residuals=I.P * (I.N + 1))
    ba = host.window_to_host_ba(ctx, I.W)
    try:
        ba.set_param("Minimum iDepth Hessian Marginlaization", 1.0)
        assert ba.run(), ba.last_error()
        N = I.N
        ba.flag_frame(1)
        idp, alive0, ng = ba.points()
        assert ba.try_marginalize(), ba.last_error()
        tm, mg, ih = ba.point_flags()
        idp, alive1, ng = ba.points()
        hosted = (I.W.pts["host"] == 1) & (alive0 == 1)
        assert tm.sum() > 10, tm.sum()
        # every surviving point hosted by the flagged frame is going to be marginalised; dropped ones are dead
        assert np.all((tm[hosted] == 1) | (alive1[hosted] == 0))
        assert np.all(ih[tm == 1] > 1.0)
        st, ralive, good = ba.residual_states()
        # ---- marginalizePointsF: prior += 0.25 (M - Msc) of exactly those points
        H0, b0 = ba.prior()
        assert np.all(H0 == 0) and H0.shape == (8 * N + 4, 8 * N + 4)
        A = ba.algebra()
        assert ba.marginalize_points(), ba.last_error()
        H1, b1 = ba.prior()
        assert np.abs(H1 - H1.T).max() <= 1e-9 * np.abs(H1).max() and np.abs(H1).max() > 0
        ev = np.linalg.eigvalsh(0.5 * (H1 + H1.T)[4:, 4:])
        assert ev.min() > -1e-6 * ev.max()                       # Schur complement of a Gram matrix
        tm2, mg2, _ = ba.point_flags()
        assert np.array_equal(mg2, tm) and tm2.sum() == 0
        _, alive2, _ = ba.points()
        assert np.all(alive2[mg2 == 1] == 0)
        # ---- marginalizeFrames: dense algebra against the oracle
        Aa = ba.algebra()
        removed = ba.marginalize_frames()
        assert list(removed) == [1]
        H2, b2 = ba.prior()
        Ho, bo = O.marginalize_frame(H1, b1, N, 1, Aa["prior"][8:16], Aa["dprior"][8:16])
        assert H2.shape == (8 * N - 4, 8 * N - 4)
        assert np.abs(H2 - Ho).max() <= 1e-10 * np.abs(Ho).max() and np.abs(b2 - bo).max() <= 1e-10 * np.abs(bo).max()
        c = ba.counts()
        assert c["frames"] == N - 1
        _, alive3, _ = ba.points()
        assert np.all(alive3[I.W.pts["host"] == 1] == 0)          # the removed frame takes its points with it
        # ---- next keyframe cycle with the prior active
        ba.set_param("disableMarginalization", 0)
        assert ba.run(), ba.last_error()
        e = ba.energies(32)
        assert np.all(np.isfinite(e))
        H3, b3 = ba.prior()
        assert np.array_equal(H3, H2)                             # run() reads the prior, it does not change it
    finally:
        ba.close(); ctx.close()


def test_resident_loop_carries_the_marginalisation_prior():
    """run() with the prior ENABLED keeps the loop on the device (cmlhip_ba_set_resident_prior: HM resident, bM_top = bM + HM * delta
    re-formed by the frame step after every iteration, BA.cpp:1389-1401).  Same keyframe cycle twice; the closing run once through the
    literal host loop (solveSystem hands HM / bM_top to cmlhip_ba_solve each iteration) and once resident: same arithmetic in another
    place, so the results agree far below the fp32 accumulation noise — and the prior must actually matter (a third run without it
    lands somewhere else)."""
    from libcml_amd import device, host
    res = {}
    for mode in ("host", "resident", "noprior"):
        I = S.make_inputs("medium")
        ctx = device.Ctx(max_frames=I.N + 1, max_points=I.P, max_residuals=I.P * (I.N + 1))
        ba = host.window_to_host_ba(ctx, I.W)
        try:
            ba.set_param("Minimum iDepth Hessian Marginlaization", 1.0)
            assert ba.run(), ba.last_error()
            ba.flag_frame(1)
            assert ba.try_marginalize(), ba.last_error()
            assert ba.marginalize_points(), ba.last_error()
            assert list(ba.marginalize_frames()) == [1]
            H, b = ba.prior()
            assert np.abs(H).max() > 0
            # move the states off their linearisation points so that HM * delta is not zero in the first iteration
            for k in range(1, I.N - 1):
                st = ba.frame(k)["state"].copy(); st[:6] += 1e-3 * np.cos(np.arange(6) + k); ba.set_frame_state(k, st)
            ba.set_param("disableMarginalization", 1 if mode == "noprior" else 0)
            ba.set_param("iterations", 5); ba.set_param("ThOptIterations", 0.0)
            ok = ba.run_host_loop() if mode == "host" else ba.run()
            assert ok, ba.last_error()
            assert ba.counts()["iterations"] == 5
            idp, alive, _ = ba.points()
            res[mode] = (idp[alive == 1].copy(), [ba.frame(k) for k in range(I.N - 1)], ba.energies(16).copy(), ba.residual_states()[2].copy())
        finally:
            ba.close(); ctx.close()
    (idp_h, fr_h, e_h, g_h), (idp_r, fr_r, e_r, g_r) = res["host"], res["resident"]
    assert int((g_h != g_r).sum()) <= max(1, len(g_h) // 2000)
    for a, b in zip(fr_h, fr_r):
        assert np.abs(a["state"] - b["state"]).max() < 1e-7 * max(1.0, np.abs(a["state"]).max())
        assert np.abs(a["R"] - b["R"]).max() < 1e-8 and np.abs(a["t"] - b["t"]).max() < 1e-7
    assert np.abs(idp_h / idp_r - 1).max() < 1e-5
    assert len(e_h) == len(e_r) and np.abs(e_h[-5:] / e_r[-5:] - 1).max() < 1e-6, (e_h, e_r)
    d_prior = max(np.abs(a["state"] - b["state"]).max() for a, b in zip(fr_h, res["noprior"][1]))
    assert d_prior > 1e-5, d_prior                       # the prior is not a no-op on this window
