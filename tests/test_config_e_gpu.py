"""BASELINE.json configs[4] at FULL size against the oracle: 20-keyframe window, 8000 active points (R = 152 000 point-residuals),
1920x1080 level-0 gradient images stored as four halves per texel (CMLHIP_TEXEL_F16), fp32 arithmetic / accumulation.
The oracle is fed the same images rounded to half precision, so the per-residual outputs must again be bit-exact; the
reductions carry the fp32 accumulation-order tolerances of tests/test_ba_parity_gpu.py; N = 20 is the widest window the
LDS-resident factorisation takes (160 x 160 trailing block of the 164 x 164 system)."""
import numpy as np
import pytest

from libcml_amd import abi
from tests import ba_setup as S
from tests import dev_setup as D

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def window_e():
    I = S.make_inputs("E")
    assert (I.N, I.P, I.R) == (20, 8000, 152000) and (I.W.w, I.W.h) == (1920, 1080)
    for k in range(I.N):
        for lvl in range(len(I.grads[k])):
            I.grads[k][lvl] = I.grads[k][lvl].astype(np.float16).astype(np.float32)
    ob = S.OracleBA(I)
    ctx = D.make_ctx(I, texel_format=abi.TEXEL_F16)
    yield I, ob, ctx
    ctx.close()


def test_config_e_linearize_bit_exact(window_e):
    I, ob, ctx = window_e
    ro = ob.linearize()
    rd = ctx.ba_linearize()
    so, sd = ob.states(), ctx.ba_states()
    assert np.array_equal(so["new_state"], sd["new_state"]) and np.array_equal(so["state"], sd["state"])        # all R
    assert np.array_equal(so["new_energy_wo"].view(np.uint32), sd["new_energy_wo"].view(np.uint32))
    IN = so["new_state"] == 0
    assert IN.sum() > I.R // 4
    assert np.array_equal(so["new_energy"][IN].view(np.uint32), sd["new_energy"][IN].view(np.uint32))
    assert np.array_equal(ob.rJ(0)[IN].view(np.uint32), ctx.ba_rj(0)[IN].view(np.uint32)), "raw Jacobian records differ"
    co = ob.view("r_center", 3 * I.R, np.float32).reshape(-1, 3)
    assert np.array_equal(co[IN].view(np.uint32), ctx.ba_center()[IN].view(np.uint32))
    assert (ro.n_in, ro.n_oob, ro.n_outlier) == (rd.n_in, rd.n_oob, rd.n_outlier)
    assert np.float32(ro.new_frame_energy_th).view(np.uint32) == np.float32(rd.new_frame_energy_th).view(np.uint32)
    assert abs(ro.energy - rd.energy) <= 1e-12 * abs(ro.energy)


def test_config_e_accumulate_schur_solve(window_e):
    I, ob, ctx = window_e
    ob.apply(1); ctx.ba_apply(1)
    so, sd = ob.states(), ctx.ba_states()
    assert np.array_equal(so["state"], sd["state"]) and np.array_equal(so["good"], sd["good"])
    HAo, bAo, HLo, bLo, Hso, bso = ob.accumulate()
    HAd, bAd, HLd, bLd, Hsd, bsd = D.accumulate(ctx, I)
    assert D.rel(HAd, HAo) < 2e-5 and D.rel(bAd, bAo) < 2e-5
    assert D.rel(HLd, HLo) < 1e-12 and D.rel(bLd, bLo) < 1e-12
    assert D.rel(Hsd, Hso) < 5e-5 and D.rel(bsd, bso) < 5e-5
    assert np.array_equal(Hsd, Hsd.T)
    lam = 1e-5
    xd, rcd = ctx.ba_solve(lam)
    xo2, rco = ob.solve(lam, HAd, bAd, HLd, bLd, Hsd, bsd)             # the 164 x 164 factorisation on the same matrices
    assert rcd == 0 and rco == 0
    assert D.rel(xd, xo2) < 1e-7, D.rel(xd, xo2)
    sto, _ = ob.backsub(xd)
    std, rc = ctx.ba_backsub(xd)
    assert rc == 0
    assert np.abs(sto - std).max() <= 1e-4 * np.abs(sto).max()          # up to 19 residuals per point at N = 20: the fp32 point sums (Hdd, bd, Hcd) that enter the step carry 19-term accumulation noise
