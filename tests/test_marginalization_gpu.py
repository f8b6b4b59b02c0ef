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
