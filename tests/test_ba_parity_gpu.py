"""GPU parity of the bundle-adjustment hot path against the oracle, through the C ABI.
Bars: per-residual records, states, energies, thresholds: BIT-EXACT (the kernel keeps the reference's
statement order, fp contraction off).  Reductions over residuals/points (13x13 pair blocks, Schur, solve,
back-substitution): fp32 accumulation-order tolerance, stated per test."""
import numpy as np
import pytest

from tests import ba_setup as S
from tests import dev_setup as D

pytestmark = pytest.mark.gpu


def gauge_free_pose_update_error(I, xd, xo):
    """max |P xd - P xo| / max |P xo| over the frame blocks, P = the reference's nullspace projection (computeNullspaces BA.cpp:2365-2417,
    orthogonalize :1196-1261) as the oracle restates them."""
    import ctypes as C
    from tests import oracle_lib as O
    n = 8 * I.N + 4
    ns = np.zeros(7 * n)
    O.lib().orc_ba_nullspaces(I.frames, I.N, C.byref(I.scales), O.ptr(ns, C.c_double))
    pd, po = O.orthogonalize(xd, ns.reshape(7, n), 1e-5), O.orthogonalize(xo, ns.reshape(7, n), 1e-5)
    return float(np.abs(pd[4:] - po[4:]).max() / max(np.abs(po[4:]).max(), 1e-300))


def reduced_system_conditioning(I, Ho, lam=1e-5):
    """(cancellation, kappa) of the system the pose update solves: H = (H_L + H_A) diag*(1+lam) - H_sc/(1+lam), Jacobi-scaled as
    BA.cpp:1312-1316, restricted to the complement of the 7 gauge directions.  cancellation = |H_A| / |H| (how much of H_A the Schur
    complement cancels), kappa = largest / smallest eigenvalue.  A relative difference eps of the accumulated matrices can move the
    gauge-free update by up to about eps * cancellation * kappa."""
    import ctypes as C
    from tests import oracle_lib as O
    HAo, bAo, HLo, bLo, Hso, bso = Ho
    n = 8 * I.N + 4
    H = HLo + HAo
    H[np.diag_indices(n)] *= (1 + lam)
    H = H - Hso / (1 + lam)
    Sv = 1.0 / np.sqrt(np.diag(H) + 10.0)
    Hs = (Sv[:, None] * H * Sv[None, :])[4:, 4:]
    ns = np.zeros(7 * n)
    O.lib().orc_ba_nullspaces(I.frames, I.N, C.byref(I.scales), O.ptr(ns, C.c_double))
    Nn = (ns.reshape(7, n)[:, 4:] / Sv[None, 4:]).T                      # gauge directions in the scaled coordinates
    Q, _ = np.linalg.qr(Nn)
    Pm = np.eye(n - 4) - Q @ Q.T
    ev = np.linalg.eigvalsh(Pm @ Hs @ Pm)
    ev = ev[np.abs(ev) > 1e-12 * np.abs(ev).max()]
    HAs = (Sv[:, None] * HAo * Sv[None, :])[4:, 4:]
    return float(np.abs(HAs).max() / np.abs(Hs).max()), float(np.abs(ev).max() / np.abs(ev).min())


@pytest.fixture(scope="module", params=["tiny", "small", "A", "B"])     # A, B: BASELINE.json configs[0] and configs[1] at full size
def pair(request):
    I = S.make_inputs(request.param)
    ob = S.OracleBA(I)
    ctx = D.make_ctx(I)
    yield I, ob, ctx
    ctx.close()


def test_index_maps_exact(pair):
    I, ob, ctx = pair
    m = ctx.ba_index_maps()
    w = ob.w.contents
    assert np.array_equal(m["pair_of"], ob.view("pair_of", I.R, np.int32))
    assert np.array_equal(m["by_point_off"], np.ctypeslib.as_array(w.by_point_off, shape=(I.P + 1,)))
    assert np.array_equal(m["by_point"], ob.view("by_point", I.R, np.int32))
    assert np.array_equal(m["by_pair_off"], np.ctypeslib.as_array(w.by_pair_off, shape=(I.N * I.N + 1,)))
    assert np.array_equal(m["by_pair"], ob.view("by_pair", I.R, np.int32))
    hosts = I.points["host"][I.residuals["point"]]
    assert np.array_equal(m["pair_of"], hosts + I.residuals["target"] * I.N)      # htIDX, BA.cpp:1677


def test_linearize_bit_exact(pair):
    I, ob, ctx = pair
    ro = ob.linearize()
    rd = ctx.ba_linearize()
    so, sd = ob.states(), ctx.ba_states()
    assert np.array_equal(so["new_state"], sd["new_state"])
    assert np.array_equal(so["state"], sd["state"])
    assert np.array_equal(so["new_energy_wo"].view(np.uint32), sd["new_energy_wo"].view(np.uint32))
    IN = so["new_state"] == 0
    assert IN.sum() > 10
    assert np.array_equal(so["new_energy"][IN].view(np.uint32), sd["new_energy"][IN].view(np.uint32))
    jo, jd = ob.rJ(0), ctx.ba_rj(0)
    assert np.array_equal(jo[IN].view(np.uint32), jd[IN].view(np.uint32)), "raw Jacobian records differ"
    co = ob.view("r_center", 3 * I.R, np.float32).reshape(-1, 3)
    assert np.array_equal(co[IN].view(np.uint32), ctx.ba_center()[IN].view(np.uint32))
    assert (ro.n_in, ro.n_oob, ro.n_outlier) == (rd.n_in, rd.n_oob, rd.n_outlier)
    assert np.float32(ro.new_frame_energy_th).view(np.uint32) == np.float32(rd.new_frame_energy_th).view(np.uint32)
    assert abs(ro.energy - rd.energy) <= 1e-12 * abs(ro.energy)      # fp64 sum, order differs


def test_apply_and_accumulate(pair):
    I, ob, ctx = pair
    ob.apply(1); ctx.ba_apply(1)
    so, sd = ob.states(), ctx.ba_states()
    for k in ("state", "good"):
        assert np.array_equal(so[k], sd[k])
    assert np.array_equal(so["energy"].view(np.uint32), sd["energy"].view(np.uint32))
    g = so["good"] == 1
    assert np.array_equal(ob.rJ(1)[g].view(np.uint32), ctx.ba_rj(1)[g].view(np.uint32))
    jo = ob.view("JpJdF", 8 * I.R, np.float32).reshape(-1, 8)
    assert np.abs(jo[g] - ctx.ba_jpjdf()[g]).max() <= 2e-6 * np.abs(jo[g]).max()      # fma contraction in apply
    HAo, bAo, HLo, bLo, Hso, bso = ob.accumulate()
    HAd, bAd, HLd, bLd, Hsd, bsd = D.accumulate(ctx, I)
    # raw fp32 13x13 pair accumulators: tiered sequential sums (oracle) vs tree sums (device)
    ao = ob.view("accA", 169 * I.N * I.N, np.float32).reshape(-1, 13, 13)
    ad = ctx.ba_pair_acc(0)
    for q in range(I.N * I.N):
        assert np.abs(ao[q] - ad[q]).max() <= 2e-5 * max(np.abs(ao[q]).max(), 1e-30), q
    # (bars at ~10x the worst deviation of the seed sweeps — 1.3e-7 ... 5.5e-7, profiles/round*_parity_soak_tolerance.txt — so that a
    #  10x loss of accumulation accuracy fails here; the fp32 pair accumulators above keep the looser per-block bar)
    print("matrices, device vs oracle: H_A %.2e b_A %.2e H_sc %.2e b_sc %.2e" % (D.rel(HAd, HAo), D.rel(bAd, bAo), D.rel(Hsd, Hso), D.rel(bsd, bso)))
    assert D.rel(HAd, HAo) < 3e-6 and D.rel(bAd, bAo) < 3e-6
    assert D.rel(HLd, HLo) < 1e-12 and D.rel(bLd, bLo) < 1e-12          # priors only
    assert D.rel(Hsd, Hso) < 5e-6 and D.rel(bsd, bso) < 5e-6
    assert np.abs(HAd - HAd.T).max() <= 1e-9 * np.abs(HAd).max()
    # per-point scalars
    pa = ctx.ba_point_acc()
    P = I.P
    for name, col in (("Hdd_accAF", 0), ("bd_accAF", 1), ("HdiF", 12), ("bdSumF", 13)):
        o = ob.view(name, P, np.float32)
        assert np.abs(o - pa[:, col]).max() <= 2e-5 * max(np.abs(o).max(), 1e-30), name
    # solve + back-substitution
    lam = 1e-5
    xo, rco = ob.solve(lam, HAo, bAo, HLo, bLo, Hso, bso)
    xd, rcd = ctx.ba_solve(lam)
    assert rco == 0 and rcd == 0
    # (a) the factorisation in isolation: the oracle's pivoted LDLT (Eigen semantics) on the DEVICE's matrices
    xo2, _ = ob.solve(lam, HAd, bAd, HLd, bLd, Hsd, bsd)
    assert D.rel(xd, xo2) < 1e-7, D.rel(xd, xo2)
    # (b) end to end: the monocular window has a near-singular scale gauge, so fp32-level differences of H are
    # amplified in x; the meaningful bar is the backward error of the device x in the ORACLE's scaled system
    n = 8 * I.N + 4
    H = HLo + HAo
    H[np.diag_indices(n)] *= (1 + lam)
    H = H - Hso / (1 + lam)
    b = bLo + bAo - bso
    Sv = 1.0 / np.sqrt(np.diag(H) + 10.0)
    r = Sv[4:] * (H[4:, 4:] @ xd[4:] - b[4:])
    assert np.linalg.norm(r) <= 2e-4 * np.linalg.norm(Sv[4:] * b[4:]), np.linalg.norm(r) / np.linalg.norm(Sv[4:] * b[4:])
    assert D.rel(xd, xo) < 5e-2
    # (c) the POSE UPDATE with the gauge removed (orthogonalize, BA.cpp:1196-1261: x minus its component in the 7-dimensional
    # nullspace of global pose + scale).  The raw x differs in the weakly determined gauge directions (b); what the frames are
    # actually stepped by is compared here, device against oracle, relative to the largest component of the update:
    gf = gauge_free_pose_update_error(I, xd, xo)
    # The size of gf is the window's conditioning times the fp32 accumulation noise of the matrices (measured 1.6e-4 ... 1.4e-3 at config B
    # over seeds and scene variants, up to 9.8e-3 on 300-point windows: profiles/round2_parity_soak.txt), so a fixed bar on it says little
    # (ADVICE round 2).  The tight, conditioning-independent statement is in two parts: the accumulated matrices agree (bars above,
    # measured ~1e-7), and on the DEVICE's matrices the device's assemble / factorise / substitute path gives the pose update an
    # independent fp64 solve gives (numpy, LU with partial pivoting) — compared with the gauge removed, relative to the update:
    Hd = HLd + HAd
    Hd[np.diag_indices(n)] *= (1 + lam)
    Hd = Hd - Hsd / (1 + lam)
    bd_ = bLd + bAd - bsd
    Svd = 1.0 / np.sqrt(np.diag(Hd) + 10.0)
    xn = np.zeros(n)
    xn[4:] = Svd[4:] * np.linalg.solve(Svd[4:, None] * Hd[4:, 4:] * Svd[None, 4:], Svd[4:] * bd_[4:])
    gf_solver = gauge_free_pose_update_error(I, xd, xn)
    print("gauge-free pose update: device vs oracle %.3g; device solve vs numpy solve on the device's matrices %.3g" % (gf, gf_solver))
    assert gf_solver < 1e-9, (gf_solver, gf)                                    # measured 1e-13 ... 1e-12
    # ... and the device-vs-oracle difference itself is bounded FROM THE WINDOW (ADVICE rounds 2 / 3): it must be what the measured
    # difference of the two sets of matrices propagates to — the same independent fp64 solve on the ORACLE's matrices gives the update
    # the oracle's matrices imply; the distance between the two numpy updates is this window's (conditioning x accumulation noise), and
    # the device-vs-oracle figure may exceed it only by the two solvers' own errors (1e-9 each, checked above / in (a))
    Ho_ = HLo + HAo
    Ho_[np.diag_indices(n)] *= (1 + lam)
    Ho_ = Ho_ - Hso / (1 + lam)
    bo_ = bLo + bAo - bso
    Svo = 1.0 / np.sqrt(np.diag(Ho_) + 10.0)
    xno = np.zeros(n)
    xno[4:] = Svo[4:] * np.linalg.solve(Svo[4:, None] * Ho_[4:, 4:] * Svo[None, 4:], Svo[4:] * bo_[4:])
    gf_window = gauge_free_pose_update_error(I, xn, xno)
    gf_oracle_solver = gauge_free_pose_update_error(I, xo, xno)                  # the oracle's pivoted LDLT against numpy on the ORACLE's matrices
    print("   window bound (numpy on the device's matrices vs numpy on the oracle's): %.3g; oracle LDLT vs numpy on its own matrices %.3g" % (gf_window, gf_oracle_solver))
    # triangle: device -> numpy(device matrices) -> numpy(oracle matrices) -> oracle
    assert gf <= 1.1 * (gf_solver + gf_window + gf_oracle_solver) + 1e-9, (gf, gf_solver, gf_window, gf_oracle_solver)
    assert gf < (2e-3 if I.P >= 2000 else 2e-2), gf                              # outer sanity bar by window size (measured <= 1.4e-3 at config B, <= 9.8e-3 on 300-point windows)
    # same x into both back-substitutions isolates that kernel
    sto, _ = ob.backsub(xo)
    std, rc = ctx.ba_backsub(xo)
    assert rc == 0
    assert np.abs(sto - std).max() <= 5e-5 * np.abs(sto).max()


@pytest.mark.parametrize("N,P", [(12, 1100), (9, 700), (14, 2100), (21, 300), (26, 600), (32, 400)])
def test_system_and_solver_wide_window(N, P):
    """Windows wider than the default: several Schur slices per tile (P > 512), block columns whose panel does not fit
    one wave (8N+4 > 64+...), systems assembled by k_ba_assemble (more than 16 blocks), with and without the calibration block and a
    marginalisation prior; from 21 frames on (8N > 160) the factorisation leaves the LDS and runs in global memory (k_ba_solve_global).
    Bars as above:
    Schur/Hessian blocks at fp32 accumulation tolerance, the factorisation isolated on the DEVICE's matrices at 1e-7."""
    I = S.make_inputs((N, P, 320, 240, 3, 260.0, 260.0, 159.5, 119.5))
    ob = S.OracleBA(I)
    ctx = D.make_ctx(I)
    try:
        ob.linearize(); ctx.ba_linearize()
        ob.apply(1); ctx.ba_apply(1)
        HAo, bAo, HLo, bLo, Hso, bso = ob.accumulate()
        HAd, bAd, HLd, bLd, Hsd, bsd = D.accumulate(ctx, I)
        assert D.rel(HAd, HAo) < 2e-5 and D.rel(bAd, bAo) < 2e-5
        assert D.rel(Hsd, Hso) < 5e-5 and D.rel(bsd, bso) < 5e-5
        assert np.array_equal(Hsd, Hsd.T)
        n = 8 * N + 4
        rng = np.random.default_rng(7)
        Q = rng.standard_normal((n, n)) * 30.0
        HM = Q @ Q.T
        bM = rng.standard_normal(n) * 100.0
        for optcal in (0, 1):
            for hm, bm in ((None, None), (HM, bM)):
                xd, rcd = ctx.ba_solve(1e-4, hm, bm, optcal)
                xo, rco = ob.solve(1e-4, HAd, bAd, HLd, bLd, Hsd, bsd, hm, bm, optcal)
                assert rcd == 0 and rco == 0
                assert D.rel(xd, xo) < 1e-7, (optcal, hm is not None, D.rel(xd, xo))
                if not optcal:
                    assert np.all(xd[:4] == 0)
        # the back-substitution on the same x (from 12 frames x 2048 points on, its x . adjoint table comes from k_ba_xad)
        sto, _ = ob.backsub(xo)
        std, rc = ctx.ba_backsub(xo)
        assert rc == 0
        assert np.abs(sto - std).max() <= 5e-5 * np.abs(sto).max()
    finally:
        ctx.close()


def test_linearize_fp16_texels_bit_exact():
    """Config E stores the pyramids as four halves per texel (CMLHIP_TEXEL_F16) and computes in fp32.  With the oracle fed
    the same images rounded to half precision, records, states and energies must again be bit-exact: the only difference
    of the mode is the storage format of the texel."""
    from libcml_amd import abi
    I = S.make_inputs("small")
    for k in range(I.N):
        for lvl in range(len(I.grads[k])):
            I.grads[k][lvl] = I.grads[k][lvl].astype(np.float16).astype(np.float32)
    ob = S.OracleBA(I)
    ctx = D.make_ctx(I, texel_format=abi.TEXEL_F16)
    try:
        ro = ob.linearize()
        rd = ctx.ba_linearize()
        so, sd = ob.states(), ctx.ba_states()
        assert np.array_equal(so["new_state"], sd["new_state"])
        assert np.array_equal(so["new_energy_wo"].view(np.uint32), sd["new_energy_wo"].view(np.uint32))
        IN = so["new_state"] == 0
        assert IN.sum() > 10
        assert np.array_equal(ob.rJ(0)[IN].view(np.uint32), ctx.ba_rj(0)[IN].view(np.uint32)), "raw Jacobian records differ"
        assert (ro.n_in, ro.n_oob, ro.n_outlier) == (rd.n_in, rd.n_oob, rd.n_outlier)
        assert np.float32(ro.new_frame_energy_th).view(np.uint32) == np.float32(rd.new_frame_energy_th).view(np.uint32)
        ob.apply(1); ctx.ba_apply(1)
        Ho = ob.accumulate(); Hd = D.accumulate(ctx, I)
        assert D.rel(Hd[0], Ho[0]) < 2e-5 and D.rel(Hd[4], Ho[4]) < 5e-5
    finally:
        ctx.close()
