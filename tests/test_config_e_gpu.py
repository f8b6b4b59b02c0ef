"""BASELINE.json configs[4] at FULL size against the oracle: 20-keyframe window, 8000 active points (R = 152 000 point-residuals),
1920x1080 level-0 gradient images stored as four halves per texel (CMLHIP_TEXEL_F16), fp32 arithmetic / accumulation.
The oracle is fed the same images rounded to half precision, so the per-residual outputs must again be bit-exact; the
reductions carry the fp32 accumulation-order tolerances of tests/test_ba_parity_gpu.py; N = 20 is the widest window the
LDS-resident factorisation takes (160 x 160 trailing block of the 164 x 164 system).
test_config_e_against_fp32_texels states the OTHER distance: from the fp16-texel device run to the oracle on the unrounded fp32 images —
what the reference (fp32 `Vector3f` texels, image/Array2D.h:265-286) would compute on the same window."""
import json
import os

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
    I.grads_fp32_level0 = [I.grads[k][0] for k in range(I.N)]          # the UNROUNDED level 0 (test_config_e_against_fp32_texels)
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


def test_config_e_against_fp32_texels(window_e):
    """north_star: "match the reference CPU path's pose-Hessian and pose updates within a stated fp32 tolerance".  Config E is the one place the
    product's INPUT is narrower than the reference's: texels are halves (intensities in [128, 255] have an fp16 ulp of 0.125 grey levels against a
    Huber threshold of 9).  The same window goes through the oracle on the unrounded fp32 images (DSOBundleAdjustment.cpp:214-271 on fp32
    texels) and the device's fp16-texel results are held against it.  Measured (seed 0xC0FFEE, first linearisation, 68 % IN): 29 of 152 000
    residuals classified differently (1.9e-4), per-residual energies median 3e-3 / p99 5e-2 relative, total energy 2e-5, H_A 6.5e-4, b_A 2.6e-3,
    H_sc 5.1e-4, b_sc 2.4e-3 of the largest entry; Jacobi-scaled (BA.cpp:1312-1316) H_A 1.4e-3 and H_sc 1.2e-3 — ABOVE the 1e-3 reference point
    of north_star: config E's texel format costs that much and DESIGN section 5 says so; gauge-free pose update 1e-2 (max) / 6e-3 (l2): the
    matrix difference times the window's conditioning.  Bars = 2 x measured."""
    from tests import e_texel_check as T
    I, ob, ctx = window_e
    sd = ctx.ba_states()
    if not sd["good"].any():                                            # run alone: bring the device window to the state behind applyRes
        ctx.ba_linearize(); ctx.ba_apply(1)
        sd = ctx.ba_states()
    Hd = D.accumulate(ctx, I)
    rounded = [I.grads[k][0] for k in range(I.N)]
    try:
        for k in range(I.N):
            I.grads[k][0] = I.grads_fp32_level0[k]
        st32, Hs32 = T.oracle_side(I)
    finally:
        for k in range(I.N):
            I.grads[k][0] = rounded[k]
    rep = T.compare(I, sd, Hd, st32, Hs32)
    print("config E, fp16-texel device run vs oracle on fp32 texels:", json.dumps(rep))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "config_e_vs_fp32_texels.json"), "w") as f:
            json.dump(rep, f, indent=1)
    assert rep["class_flips"] <= 60 and rep["good_flips"] <= 60                     # 29 measured: 4e-4 of R
    assert rep["energy_rel_median"] < 7e-3 and rep["energy_rel_p99"] < 0.11 and rep["total_energy_rel"] < 1e-4
    assert rep["H_A_rel"] < 1.3e-3 and rep["H_sc_rel"] < 1.1e-3 and rep["b_A_rel"] < 5.5e-3 and rep["b_sc_rel"] < 5e-3
    assert rep["H_A_jacobi_rel"] < 3e-3 and rep["H_sc_jacobi_rel"] < 2.5e-3         # measured 1.4e-3 / 1.2e-3: above north_star's 1e-3, stated
    assert rep["x_gauge_free_rel"] < 2e-2 and rep["x_gauge_free_rel_l2"] < 1.2e-2
