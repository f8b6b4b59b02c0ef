"""CPU tests of the oracle: pinned against the committed golden vectors (outputs of the reference's vendored
Eigen 3.4.0 / Sophus 1.1.0, tests/golden/make_thirdparty_vectors.py), against oracle/_ref when it is built, and
against mathematical identities for the DSO-specific restatements (no reference outputs exist for those)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import ba_setup as S
from tests import oracle_lib as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "thirdparty_vectors.npz"))


def test_se3_against_golden_sophus():
    for i in range(len(G["xi"])):
        T = O.se3_exp(G["xi"][i])
        assert np.abs(np.array(T.q[:]) - G["q"][i]).max() < 1e-14
        assert np.abs(np.array(T.t[:]) - G["t"][i]).max() < 1e-13
        assert np.abs(O.se3_log(T) - G["log"][i]).max() < 1e-12
        assert np.abs(O.se3_adj(T).ravel() - G["adj"][i]).max() < 1e-13
        R, t = O.se3_matrix(T)
        assert np.abs(R.ravel() - G["R"][i]).max() < 1e-14
        Ti = O.se3_inv(T)
        assert np.abs(np.array(Ti.q[:]) - G["q_inv"][i]).max() < 1e-14 and np.abs(np.array(Ti.t[:]) - G["t_inv"][i]).max() < 1e-13
        j = (i + 7) % len(G["xi"])
        Tm = O.se3_mul(T, O.se3_exp(G["xi"][j]))
        assert np.abs(np.array(Tm.q[:]) - G["q_mul"][i]).max() < 1e-14 and np.abs(np.array(Tm.t[:]) - G["t_mul"][i]).max() < 1e-12
        Tr = O.se3_from_Rt(R, t)
        q = np.array(Tr.q[:]); qg = G["q_from_R"][i]
        assert min(np.abs(q - qg).max(), np.abs(q + qg).max()) < 1e-13
        D = O.se3_dx_exp_x(G["xi"][i])
        assert np.abs(D.ravel() - G["dx_exp_x"][i]).max() < 1e-9 * max(1.0, np.abs(G["dx_exp_x"][i]).max())


def test_eigen_expression_shapes_bitwise_against_golden():
    """The projection arithmetic of the path is written with small fixed-size Eigen products whose evaluation order is Eigen's,
    not left-to-right (float 3x3 * 3-vector: e0 + (e1 + e2); double: packet rows (e0 + e1) + e2, scalar row e0 + (e1 + e2);
    homogeneous product; cofactor inverse).  The oracle's orc_eig_* helpers — the formulas the tracker, tracer and initializer
    restatements and the device kernels use — must reproduce the vendored Eigen 3.4.0 BIT FOR BIT."""
    L = O.lib()
    f = C.c_float; d = C.c_double; P = O.ptr
    def same32(a, b): return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    def same64(a, b): return np.array_equal(a.view(np.uint64), b.view(np.uint64))
    n = len(G["eig_M3f"])
    for i in range(n):
        M = G["eig_M3f"][i].copy(); v = G["eig_v3f"][i].copy(); t = G["eig_t3f"][i].copy(); s = float(G["eig_s3f"][i])
        o = np.zeros(3, np.float32)
        L.orc_eig_matvec3f_affine(P(M, f), P(v, f), P(t, f), f(s), 1, P(o, f)); assert same32(o, G["eig_affine_plus"][i]), i
        L.orc_eig_matvec3f_affine(P(M, f), P(v, f), P(t, f), f(s), -1, P(o, f)); assert same32(o, G["eig_affine_minus"][i]), i
        L.orc_eig_matvec3f_noalias(P(M, f), P(v, f), P(t, f), f(s), P(o, f)); assert same32(o, G["eig_noalias"][i]), i
        o9 = np.zeros(9, np.float32)
        L.orc_eig_matmul3f(P(M, f), P(G["eig_B3f"][i].copy(), f), P(o9, f)); assert same32(o9, G["eig_matmul3f"][i]), i
        L.orc_eig_inverse3f(P(M, f), P(o9, f)); assert same32(o9, G["eig_inverse3f"][i]), i
        L.orc_eig_inverse3f(P(G["eig_K3f"][i].copy(), f), P(o9, f)); assert same32(o9, G["eig_inverseK3f"][i]), i
        Md = G["eig_M3d"][i].copy(); vd = G["eig_v3d"][i].copy(); td = G["eig_t3d"][i].copy(); sd = float(G["eig_s3d"][i])
        od = np.zeros(3)
        L.orc_eig_homog3d(P(Md, d), P(vd[:2].copy(), d), P(td, d), d(sd), P(od, d)); assert same64(od, G["eig_homog3d"][i]), i
        L.orc_eig_matvec3d(P(Md, d), P(vd, d), P(od, d)); assert same64(od, G["eig_matvec3d"][i]), i
        o9d = np.zeros(9)
        L.orc_eig_matmul3d(P(Md, d), P(G["eig_B3d"][i].copy(), d), P(o9d, d)); assert same64(o9d, G["eig_matmul3d"][i]), i
        assert same32(o9d.astype(np.float32), G["eig_matmul3d_cast"][i])
        L.orc_eig_inverse3d(P(Md, d), P(o9d, d)); assert same64(o9d, G["eig_inverse3d"][i]), i
        Kd = G["eig_K3f"][i].astype(np.float64); KR = np.zeros(9); Ki = np.zeros(9); out = np.zeros(9); kt = np.zeros(3)
        L.orc_eig_matmul3d(P(Kd, d), P(Md, d), P(KR, d)); L.orc_eig_inverse3d(P(Kd, d), P(Ki, d)); L.orc_eig_matmul3d(P(KR, d), P(Ki, d), P(out, d))
        L.orc_eig_matvec3d(P(Kd, d), P(td, d), P(kt, d))
        assert same64(out, G["eig_krki"][i]) and same64(kt, G["eig_kt"][i]), i            # K * R * K.inverse(), K * t (DSOTracer.cpp:606-607)
    # dot-product shapes of the linearised-residual algebra and of the back-substitution (BA.cpp:1470, 1699, 2166, 2219)
    L.orc_eig_jp_delta.restype = C.c_float; L.orc_eig_calib_dot.restype = C.c_double; L.orc_eig_row8_dot_cast.restype = C.c_double
    for i in range(len(G["jp_Jxi"])):
        a = (P(G["jp_Jxi"][i].copy(), f), P(G["jp_dp"][i].copy(), f), P(G["jp_Jc"][i].copy(), f), P(G["jp_cdelta"][i].copy(), d),
             f(float(G["jp_Jpdd"][i])), f(float(G["jp_dd"][i])))
        assert np.float32(L.orc_eig_jp_delta(*a, 0)) == G["jp_delta_vec4f"][i], i
        assert np.float32(L.orc_eig_jp_delta(*a, 1)) == G["jp_delta_cast"][i], i
        assert L.orc_eig_calib_dot(P(G["calib_step"][i].copy(), d), P(G["calib_A"][i].copy(), f), P(G["calib_L"][i].copy(), f)) == G["calib_dot"][i], i
    assert (G["jp_delta_vec4f"] != G["jp_delta_cast"]).any()          # the two forms of the same formula really differ
    # the naive left-to-right sum is NOT what Eigen computes: the pin has teeth
    M = G["eig_M3f"]; v = G["eig_v3f"]; t = G["eig_t3f"]; s = G["eig_s3f"]
    naive = ((M[:, 0::3] * v[:, :1] + M[:, 1::3] * v[:, 1:2]) + M[:, 2::3] * v[:, 2:3]) + t * s[:, None]
    assert (naive.astype(np.float32) != G["eig_affine_plus"]).any()


def test_g2o_arithmetic_against_golden_eigen():
    """The SE3Quat / Eigen arithmetic of the g2o-based restatements (oracle/orc_g2o.h: pose-only optimisation, local BA) against
    the outputs of the same Eigen calls made by the reference's vendored Eigen 3.4.0 (oracle/ref_thirdparty.cpp)."""
    L = O.lib()
    d = C.c_double
    P = O.ptr
    n = len(G["g2o_u"])
    q = np.zeros((n, 4)); t = np.zeros((n, 3))
    for i in range(n):
        L.orc_g2o_exp(P(G["g2o_u"][i].copy(), d), P(q[i], d), P(t[i], d))
        assert np.abs(q[i] - G["g2o_q"][i]).max() < 1e-14 and np.abs(t[i] - G["g2o_t"][i]).max() < 1e-13, i
        R = np.zeros(9); L.orc_g2o_to_matrix(P(G["g2o_q"][i].copy(), d), P(R, d))
        assert np.abs(R - G["g2o_R"][i]).max() < 1e-15
        q2 = np.zeros(4); t2 = np.zeros(3)
        L.orc_g2o_from_Rt(P(G["g2o_R"][i].copy(), d), P(G["g2o_t"][i].copy(), d), P(q2, d), P(t2, d))
        assert np.abs(q2 - G["g2o_q_from_R"][i]).max() < 1e-14
        m = np.zeros(3); L.orc_g2o_map(P(G["g2o_q"][i].copy(), d), P(G["g2o_t"][i].copy(), d), P(G["g2o_X"][i].copy(), d), P(m, d))
        assert np.abs(m - G["g2o_map"][i]).max() < 1e-13
        j = (i + 11) % n
        qm = np.zeros(4); tm = np.zeros(3)
        L.orc_g2o_mul(P(G["g2o_q"][i].copy(), d), P(G["g2o_t"][i].copy(), d), P(G["g2o_q"][j].copy(), d), P(G["g2o_t"][j].copy(), d), P(qm, d), P(tm, d))
        assert np.abs(qm - G["g2o_q_mul"][i]).max() < 1e-14 and np.abs(tm - G["g2o_t_mul"][i]).max() < 1e-13
    for i in range(len(G["ldlt3_A"])):
        A = G["ldlt3_A"][i].copy(); x = np.zeros(3)
        pos = L.orc_ldlt3(P(A, d), P(G["ldlt3_b"][i].copy(), d), P(x, d))
        assert pos == int(G["ldlt3_pos"][i]), i
        assert np.abs(x - G["ldlt3_x"][i]).max() <= 1e-9 * max(1e-300, np.abs(G["ldlt3_x"][i]).max()), i
        Ai = np.zeros(9); L.orc_g2o_inv3(P(A, d), P(Ai, d))
        assert np.abs(Ai - G["inv3"][i]).max() <= 1e-10 * np.abs(G["inv3"][i]).max()
    assert int(G["ldlt3_pos"][:16].sum()) == 16 and int(G["ldlt3_pos"][16:].sum()) < 8
    for nn in (6, 36, 126):
        A = np.ascontiguousarray(G[f"llt_A{nn}"]); x = np.zeros(nn)
        ok = L.orc_g2o_llt_solve(P(A, d), nn, P(G[f"llt_b{nn}"].copy(), d), P(x, d))
        assert ok == int(G[f"llt_ok{nn}"]) == 1
        assert np.abs(x - G[f"llt_x{nn}"]).max() <= 1e-9 * np.abs(G[f"llt_x{nn}"]).max()
    x = np.zeros(6)
    assert L.orc_g2o_llt_solve(P(np.ascontiguousarray(G["llt_indef_A"]), d), 6, P(np.ones(6), d), P(x, d)) == int(G["llt_indef_ok"]) == 0
    if O.ref():                                             # where oracle/_ref is built: fresh random cases straight against it
        Rf = O.ref(); rng = np.random.default_rng(1)
        for _ in range(200):
            u = rng.normal(size=6) * rng.choice([1e-8, 0.3, 3.0])
            a = np.zeros(4); b = np.zeros(3); c2 = np.zeros(4); d2 = np.zeros(3)
            L.orc_g2o_exp(P(u, d), P(a, d), P(b, d)); Rf.ref_g2o_exp(P(u, d), P(c2, d), P(d2, d))
            assert np.abs(a - c2).max() < 1e-14 and np.abs(b - d2).max() < 1e-13


def test_ldlt_inverse_orthogonalize_against_golden_eigen():
    for n in (6, 7, 8, 64, 160):
        A, b, x = G[f"ldlt_A{n}"], G[f"ldlt_b{n}"], G[f"ldlt_x{n}"]
        xo, rc = O.ldlt_solve(A, b)
        assert rc == 0
        assert np.abs(xo - x).max() <= 1e-9 * np.abs(x).max()
        assert np.abs(O.inverse(A) - G[f"inv{n}"]).max() <= 1e-8 * np.abs(G[f"inv{n}"]).max()
    xo, rc = O.ldlt_solve(G["ldlt_indef_A"], G["ldlt_indef_b"])
    assert np.abs(xo - G["ldlt_indef_x"]).max() <= 1e-10 * np.abs(G["ldlt_indef_x"]).max()
    xo, rc = O.ldlt_solve(G["ldlt_sing_A"], G["ldlt_sing_b"])         # rank-deficient: same pivot order, same answer class
    r = G["ldlt_sing_A"] @ xo - G["ldlt_sing_b"]
    rg = G["ldlt_sing_A"] @ G["ldlt_sing_x"] - G["ldlt_sing_b"]
    assert np.linalg.norm(r) <= 10 * np.linalg.norm(rg) + 1e-9
    for name in ("orth68", "orth164", "orth_rankdef"):
        out = O.orthogonalize(G[name + "_b"], G[name + "_N"], 1e-5)
        assert np.abs(out - G[name + "_out"]).max() < 1e-12


@pytest.mark.skipif(O.ref() is None, reason="oracle/_ref not built (needs /root/reference)")
def test_against_live_reference_thirdparty():
    R = O.ref()
    rng = np.random.default_rng(7)
    d = C.c_double
    for _ in range(50):
        xi = rng.normal(size=6) * np.array([2, 2, 2, 1, 1, 1])
        T = O.se3_exp(xi)
        q = np.zeros(4); t = np.zeros(3)
        R.ref_se3_exp(O.ptr(xi, d), O.ptr(q, d), O.ptr(t, d))
        assert np.abs(np.array(T.q[:]) - q).max() < 1e-14 and np.abs(np.array(T.t[:]) - t).max() < 1e-13
        D2 = np.zeros(42)
        R.ref_se3_dx_exp_x(O.ptr(xi, d), O.ptr(D2, d))
        assert np.abs(O.se3_dx_exp_x(xi).ravel() - D2).max() < 1e-9 * max(1, np.abs(D2).max())


def test_pyramid_rules():
    ws, hs = O.pyramid_sizes(1241, 376)
    assert ws == [1241, 620, 310, 155, 77] and hs == [376, 188, 94, 47, 23]
    ws, hs = O.pyramid_sizes(640, 480)
    assert ws == [640, 320, 160, 80, 40] and hs == [480, 240, 120, 60, 30]
    rng = np.random.default_rng(0)
    g = rng.uniform(0, 255, size=(9, 11)).astype(np.float32)
    grays, grads = O.build_pyramid(g, 2)
    assert grays[1].shape == (4, 5)
    assert grays[1][1, 2] == np.float32((((g[2, 4] + g[2, 5]) + g[3, 4]) + g[3, 5]) / np.float32(4))
    assert np.all(grads[0][0] == 0) and np.all(grads[0][:, 0] == 0) and np.all(grads[0][-1] == 0)
    assert grads[0][3, 4, 1] == np.float32((g[3, 5] - g[3, 3]) * np.float32(0.5))
    assert grads[0][3, 4, 2] == np.float32((g[4, 4] - g[2, 4]) * np.float32(0.5))
    v = O.interpolate3(grads[0], 4.25, 3.5)
    a = grads[0]
    expect = a[3, 4] * np.float32(1 - .25 - .5 + .125) + a[3, 5] * np.float32(.25 - .125) + a[4, 4] * np.float32(.5 - .125) + a[4, 5] * np.float32(.125)
    assert np.allclose(v, expect, rtol=1e-6)


def _dense_system(I, ob):
    """Build J (one row per pattern pixel) from the oracle's raw records and the adjoints: the textbook form of the
    system that addToHessianTop / addToHessianSC / stitch* assemble blockwise."""
    N, P, R = I.N, I.P, I.R
    st = ob.states(); J = ob.rJ(1)
    n = 8 * N
    AH = I.adH.reshape(N * N, 8, 8); AT = I.adT.reshape(N * N, 8, 8)
    rows, rhs = [], []
    for r in range(R):
        if not st["good"][r]:
            continue
        p = I.residuals["point"][r]; t = I.residuals["target"][r]; h = I.points["host"][p]
        j = J[r]
        for k in range(8):
            jp = np.zeros(8)
            jp[:6] = j[30 + k] * j[8:14] + j[38 + k] * j[14:20]
            jp[6] = j[46 + k]; jp[7] = j[54 + k]
            row = np.zeros(n + P)
            row[8 * h:8 * h + 8] += AH[h + t * N] @ jp
            row[8 * t:8 * t + 8] += AT[h + t * N] @ jp
            row[n + p] = j[30 + k] * j[28] + j[38 + k] * j[29]
            rows.append(row); rhs.append(j[k])
    return np.array(rows), np.array(rhs)


@pytest.mark.parametrize("config", ["tiny", "small"])
def test_accumulate_schur_equal_dense_normal_equations(config):
    I = S.make_inputs(config)
    ob = S.OracleBA(I)
    ob.linearize(); ob.apply(1)
    HA, bA, HL, bL, Hsc, bsc = ob.accumulate()
    Jm, rv = _dense_system(I, ob)
    n = 8 * I.N
    H = Jm.T @ Jm; b = Jm.T @ rv
    assert np.abs(HA[4:, 4:] - H[:n, :n]).max() <= 2e-6 * np.abs(H[:n, :n]).max()
    assert np.abs(bA[4:] - b[:n]).max() <= 2e-6 * np.abs(b[:n]).max()
    Hdd = np.diag(H[n:, n:]).copy()
    Hdi = np.where(Hdd > 0, 1 / np.maximum(Hdd, 1e-10), 0)
    Hpd = H[:n, n:]
    assert np.abs(Hsc[4:, 4:] - (Hpd * Hdi) @ Hpd.T).max() <= 5e-6 * np.abs(Hsc).max()
    assert np.abs(bsc[4:] - (Hpd * Hdi) @ b[n:]).max() <= 5e-6 * np.abs(bsc).max()
    assert np.allclose(HL[4:, 4:], np.diag(I.prior)) and np.allclose(bL[4:], I.prior * I.dprior)
    # back-substitution = the point rows of the full solve
    x, rc = ob.solve(1e-5, HA, bA, HL, bL, Hsc, bsc)
    step, rc = ob.backsub(x)
    sd = -Hdi * (b[n:] - Hpd.T @ x[4:])
    assert np.abs(step - sd).max() <= 1e-4 * np.abs(sd).max()


def test_geometric_jacobians_match_finite_differences():
    """At a relative pose with unit depth scale (q_z = 1) the reference's un-normalised u,v (BA.cpp:121-122) coincide with
    the normalised ones, so Jpdxi / Jpdd / the image gradient must be the true derivatives there."""
    I = S.make_inputs("tiny", eval_noise=0.0, idepth_noise=0.0, state_noise=0.0)
    ob = S.OracleBA(I)
    fx, fy, cx, cy = I.W.K
    base = I.pairs.copy()
    hst = int(I.points["host"][I.residuals["point"][0]]); tgt = int(I.residuals["target"][0])
    q = hst * I.N + tgt
    base["R"][q] = np.eye(3).ravel(); base["t"][q] = [0.2, -0.1, 0.0]
    base["R0"][q] = base["R"][q]; base["t0"][q] = base["t"][q]
    base["aff_a"][q] = 1.0; base["aff_b"][q] = 0.0

    def centre(pairs):
        ob.set_pairs(pairs)
        w = ob.w.contents
        w.r_state[0] = 0
        O.lib().orc_ba_linearize_one(ob.w, 0)
        return np.array([w.r_center[0], w.r_center[1]], np.float64), ob.rJ(0)[0].copy()

    c0, rj = centre(base)
    assert ob.w.contents.r_new_state[0] != 1, "test residual must project inside"
    eps = 2e-3
    for k in range(6):
        xi = np.zeros(6); xi[k] = eps
        T = O.se3_mul(O.se3_exp(xi), O.se3_from_Rt(np.eye(3), base["t"][q]))
        Rm, tm = O.se3_matrix(T)
        pp = base.copy(); pp["R"][q] = Rm.ravel(); pp["t"][q] = tm
        cp, _ = centre(pp)
        xi[k] = -eps
        T = O.se3_mul(O.se3_exp(xi), O.se3_from_Rt(np.eye(3), base["t"][q]))
        Rm, tm = O.se3_matrix(T)
        pm = base.copy(); pm["R"][q] = Rm.ravel(); pm["t"][q] = tm
        cm, _ = centre(pm)
        fd = (cp - cm) / (2 * eps)
        an = np.array([rj[8 + k], rj[14 + k]])
        assert np.abs(fd - an).max() <= 2e-2 * max(1.0, np.abs(an).max()), (k, fd, an)
    # d(Ku,Kv)/d(idepth) = Jpdd
    p = int(I.residuals["point"][0])
    id0 = ob.w.contents.points[p].idepth
    ob.w.contents.points[p].idepth = id0 * (1 + 1e-3); cp, _ = centre(base)
    ob.w.contents.points[p].idepth = id0 * (1 - 1e-3); cm, _ = centre(base)
    ob.w.contents.points[p].idepth = id0
    fd = (cp - cm) / (2e-3 * id0)
    assert np.abs(fd - rj[28:30]).max() <= 2e-2 * max(1.0, np.abs(rj[28:30]).max())


def test_linearize_edge_cases():
    I = S.make_inputs("tiny")
    ob = S.OracleBA(I)
    w = ob.w.contents
    # a residual that enters OOB stays OOB, keeps its energy, record untouched (BA.cpp:68-72)
    w.r_state[3] = 1; w.r_energy[3] = 42.0
    before = ob.rJ(0)[3].copy()
    e = O.lib().orc_ba_linearize_one(ob.w, 3)
    assert e == 42.0 and w.r_new_energy_wo[3] == -1 and np.array_equal(ob.rJ(0)[3], before)
    # centre projected out of the image -> NewState OOB (BA.cpp:115-118)
    p = int(I.residuals["point"][5])
    w.points[p].x = np.float32(I.W.w + 50.0)
    w.r_state[5] = 0
    O.lib().orc_ba_linearize_one(ob.w, 5)
    assert w.r_new_state[5] == 1
    # empty window
    ob.linearize()


def test_marginalisation_restatement():
    """SURVEY §8 a15, pinned against the textbook forms: marginalizeFrame = Schur complement of the frame's 8 variables
    (with its prior folded in first); fixLinearization + MARGINALIZED-mode accumulation = normal equations of the selected
    points' residuals with res_toZero as the residual vector, Schur-reduced over their inverse depths."""
    # ---- marginalizeFrame
    rng = np.random.default_rng(5)
    N = 5; n = 8 * N + 4
    Q = rng.standard_normal((n, n + 6)) * rng.uniform(1, 1e3, (n, 1))
    HM = Q @ Q.T; bM = rng.standard_normal(n) * 1e3
    prior = rng.uniform(1e3, 1e6, 8); dprior = rng.standard_normal(8) * 1e-3
    for frame in (1, N - 1, 0):
        Hn, bn = O.marginalize_frame(HM, bM, N, frame, prior, dprior)
        io = 4 + 8 * frame
        keep = [i for i in range(n) if not (io <= i < io + 8)]
        drop = list(range(io, io + 8))
        A = HM[np.ix_(keep, keep)]; B = HM[np.ix_(keep, drop)]; D = HM[np.ix_(drop, drop)] + np.diag(prior)
        bd = bM[drop] + prior * dprior
        Di = np.linalg.inv(D)
        He = A - B @ Di @ B.T; be = bM[keep] - B @ Di @ bd
        assert np.abs(Hn - 0.5 * (He + He.T)).max() <= 1e-9 * np.abs(He).max()
        assert np.abs(bn - be).max() <= 1e-9 * np.abs(be).max()
        assert np.array_equal(Hn, Hn.T)
    d = rng.standard_normal(n)
    assert abs(O.m_energy(HM, bM, d) - abs(d @ (2 * bM + HM @ d))) <= 1e-12 * abs(d @ HM @ d)
    # ---- points
    I = S.make_inputs("small", state_noise=0.3)            # non-zero state - state_zero, so that J*delta is exercised
    ob = S.OracleBA(I)
    ob.linearize(); ob.apply(1)
    sel = np.arange(0, I.P, 3, dtype=np.int32)
    ngood = ob.relinearize_points(sel)
    st = ob.states()
    lin = ob.view("r_lin", I.R, np.uint8).copy()
    in_sel = np.isin(I.residuals["point"], sel)
    assert ngood == int(st["good"][in_sel].sum()) and ngood > 30
    assert np.array_equal(lin.astype(bool), in_sel & (st["good"] == 1))
    # res_toZero = resF - J * delta (BA.cpp:2224-2233)
    J = ob.rJ(1); rtz = ob.view("res_toZeroF", 8 * I.R, np.float32).reshape(-1, 8)
    for r in np.flatnonzero(lin)[:40]:
        p = I.residuals["point"][r]; t = I.residuals["target"][r]; h = I.points["host"][p]
        dp = I.adHTd.reshape(-1, 8)[h + t * I.N].astype(np.float64)
        dd = np.float32(I.points["idepth"][p] - np.float64(I.points["idepth_zero"][p]))
        j = J[r].astype(np.float64)
        jpx = j[8:14] @ dp[:6] + j[20:24] @ I.cdelta + j[28] * dd
        jpy = j[14:20] @ dp[:6] + j[24:28] @ I.cdelta + j[29] * dd
        expect = j[0:8] - j[30:38] * jpx - j[38:46] * jpy - j[46:54] * dp[6] - j[54:62] * dp[7]
        assert np.abs(rtz[r] - expect).max() <= 2e-5 * max(1.0, np.abs(expect).max())
    M, Mb, Msc, Mbsc = ob.marginalize_points(sel)
    # dense check: rows of the selected points' good residuals, residual vector = res_toZero
    N_, P = I.N, I.P
    nn = 8 * N_
    AH = I.adH.reshape(N_ * N_, 8, 8); AT = I.adT.reshape(N_ * N_, 8, 8)
    rows, rhs = [], []
    for r in np.flatnonzero(in_sel & (st["good"] == 1)):
        p = I.residuals["point"][r]; t = I.residuals["target"][r]; h = I.points["host"][p]
        j = J[r]
        for k in range(8):
            jp = np.zeros(8)
            jp[:6] = j[30 + k] * j[8:14] + j[38 + k] * j[14:20]
            jp[6] = j[46 + k]; jp[7] = j[54 + k]
            row = np.zeros(nn + P)
            row[8 * h:8 * h + 8] += AH[h + t * N_] @ jp
            row[8 * t:8 * t + 8] += AT[h + t * N_] @ jp
            row[nn + p] = j[30 + k] * j[28] + j[38 + k] * j[29]
            rows.append(row); rhs.append(rtz[r][k])
    Jm = np.array(rows); rv = np.array(rhs, np.float64)
    H = Jm.T @ Jm; b = Jm.T @ rv
    assert np.abs(M[4:, 4:] - H[:nn, :nn]).max() <= 2e-6 * np.abs(H[:nn, :nn]).max()
    assert np.abs(Mb[4:] - b[:nn]).max() <= 5e-6 * np.abs(b[:nn]).max()
    Hdd = np.diag(H[nn:, nn:]).copy()
    Hdi = np.where(Hdd > 0, 1 / np.maximum(Hdd, 1e-10), 0)
    Hpd = H[:nn, nn:]
    assert np.abs(Msc[4:, 4:] - (Hpd * Hdi) @ Hpd.T).max() <= 5e-6 * np.abs(Msc).max()
    assert np.abs(Mbsc[4:] - (Hpd * Hdi) @ b[nn:]).max() <= 2e-5 * np.abs(Mbsc).max()
    # linearized energy: sum over the LINEARIZED good residuals of (2 res_toZero + J delta) . J delta + priors (BA.cpp:2119-2208)
    e, num = ob.l_energy()
    assert num == int(lin.sum())


def test_tracer_restatement_recovers_depth():
    """SURVEY §8 f1 (DSOTracer).  The restatement is pinned functionally: on exact synthetic geometry the epipolar search
    must bracket the true inverse depth, a second trace from another frame must shrink the interval, and the activation
    Gauss-Newton must land on the true inverse depth."""
    from libcml_amd import abi, synth
    from tests import tracer_setup as TS
    W = synth.make_window("small", eval_noise=0.0, idepth_noise=0.0, state_noise=0.0)
    grads0 = [O.build_pyramid(W.gray[k], 1)[1][0] for k in range(W.N)]
    prm = abi.default_tracer_params()
    pts = TS.make_immature(W, grads0)
    f1, f2 = 1, 2                      # a fresh point is first traced in the frames right after its host (small baseline:
    sel = pts["host"] == 0             # the search is limited to maxPixSearch = 2.7 % of (w+h) pixels, DSOTracer.cpp:611)
    p0 = pts[sel]
    truth = W.pts["idepth_true"][sel]
    p1 = TS.oracle_trace(grads0[f1], TS.trace_pairs(W, f1), prm, p0)
    good1 = p1["last_status"] == abi.IPS_GOOD
    assert good1.sum() > 0.5 * len(p1), np.bincount(p1["last_status"], minlength=6)
    inside1 = (p1["idepth_min"] <= truth * 1.02) & (p1["idepth_max"] >= truth * 0.98)
    assert inside1[good1].mean() > 0.8            # the error bound of the search is a heuristic (DSOTracer.cpp:690-697)
    assert np.all(p1["idepth_max"][good1] >= p1["idepth_min"][good1])
    assert np.all((p1["last_status"] != abi.IPS_GOOD) | (p1["last_pixel_interval"] > 0))
    p2 = TS.oracle_trace(grads0[f2], TS.trace_pairs(W, f2), prm, p1)
    both = good1 & (p2["last_status"] == abi.IPS_GOOD)
    assert both.sum() > 20
    w1 = p1["idepth_max"][both] - p1["idepth_min"][both]; w2 = p2["idepth_max"][both] - p2["idepth_min"][both]
    assert np.median(w2 / w1) < 1.0
    # activation: per-point Gauss-Newton over the window
    cand = p2[np.isfinite(p2["idepth_max"]) & (p2["last_status"] != abi.IPS_OOB)]
    tr = W.pts["idepth_true"][sel][np.isfinite(p2["idepth_max"]) & (p2["last_status"] != abi.IPS_OOB)]
    res, idp, st = TS.oracle_optimize(grads0, W.K, TS.activation_pairs(W), prm, 1, cand)
    ok = res == 1
    assert ok.sum() > 20, np.bincount(res + 1, minlength=3)
    assert np.median(np.abs(idp[ok] / tr[ok] - 1)) < 0.02
    assert np.all(st[ok][np.arange(ok.sum()), cand["host"][ok]] == -1)


def test_initializer_restatement_dense_form_and_derivative():
    """SURVEY §8 f3 (DSOInitializer::calcResAndGS).  The restatement against an independent float64 numpy form of the same
    sums (H = sum J^T J over the inlier residuals, Hsc = sum_p w_p jb_p jb_p^T), and its idepth Jacobian against a finite
    difference of the per-point energy (dE/didepth = 2 sum r dd)."""
    from tests import initializer_setup as IS
    level = 1
    W, g0, g1, R, t, ratio, tlog = IS.scene(level=level)
    pts = IS.make_points(g0, step=3)
    prm = IS.make_params(W.K, level, R, t, ratio, tlog)
    o, H, b, Hsc, bsc, res = IS.oracle_calc(g1, prm, pts)
    n = len(pts)
    inl = o["is_good_new"] == 1
    assert 0.3 * n < inl.sum() < n
    assert res[2] == 2 * n
    assert res[1] == np.float32(prm.alpha_k * n)                          # |t|^2 * alphaW > alphaK here: alphaOpt = 0 (coupling branch)
    # --- independent dense form, float64
    RKi = np.array(list(prm.RKi), np.float64).reshape(3, 3); tt = np.array(list(prm.t), np.float64)
    Hn = np.zeros((9, 9)); Hs = np.zeros((9, 9)); E = 0.0
    for i in range(n):
        if not inl[i]:
            E += float(pts["energy"][i, 0]); continue
        q = pts["p_pattern"][i].astype(np.float64) @ RKi.T + tt * float(pts["idepth_new"][i])
        u = q[:, 0] / q[:, 2]; v = q[:, 1] / q[:, 2]; nid = float(pts["idepth_new"][i]) / q[:, 2]
        J = np.zeros((8, 9)); dd = np.zeros(8); e = 0.0
        for k in range(8):
            hit = O.interpolate3(g1, float(np.float32(prm.fx * u[k] + prm.cx)), float(np.float32(prm.fy * v[k] + prm.cy))).astype(np.float64)
            r = hit[0] - prm.aff_a * float(pts["color"][i, k]) - prm.aff_b
            hw = 1.0 if abs(r) < prm.huber else prm.huber / abs(r)
            e += hw * r * r * (2 - hw)
            s = np.sqrt(hw) if hw < 1 else 1.0
            dx = s * hit[1] * prm.fx; dy = s * hit[2] * prm.fy
            J[k, :8] = [nid[k] * dx, nid[k] * dy, -nid[k] * (u[k] * dx + v[k] * dy), -u[k] * v[k] * dx - (1 + v[k] ** 2) * dy,
                        (1 + u[k] ** 2) * dx + u[k] * v[k] * dy, -v[k] * dx + u[k] * dy, -s * prm.aff_a * float(pts["color"][i, k]), -s]
            J[k, 8] = s * r
            dd[k] = dx * (tt[0] - tt[2] * u[k]) / q[k, 2] + dy * (tt[1] - tt[2] * v[k]) / q[k, 2]
        E += e
        Hn += J.T @ J
        jb = J.T @ dd
        jb[8] += prm.coupling_weight * (float(pts["idepth_new"][i]) - float(pts["iR"][i]))
        w = 1.0 / (1.0 + dd @ dd + prm.coupling_weight)
        Hs += w * np.outer(jb, jb)
        assert abs(o["energy_new"][i, 0] - e) <= 1e-4 * max(e, 1.0)
        assert abs(o["jb"][i, 9] - w) <= 1e-4 * w and abs(o["last_hessian_new"][i] - dd @ dd) <= 1e-3 * max(dd @ dd, 1e-4)
    for got, want, name in ((H, Hn[:8, :8], "H"), (b, Hn[:8, 8], "b"), (Hsc, Hs[:8, :8], "Hsc"), (bsc, Hs[:8, 8], "bsc")):
        assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max(), name
    assert abs(res[0] - E) <= 1e-4 * E
    # --- dE/didepth by central differences (holds in the Huber band too: d(2k|r| - k^2) = 2 hw r dr = 2 (sqrt(hw) r)(sqrt(hw) dr))
    eps = 1e-3
    hi = pts.copy(); hi["idepth_new"] += np.float32(eps)
    lo = pts.copy(); lo["idepth_new"] -= np.float32(eps)
    oh = IS.oracle_calc(g1, prm, hi)[0]; ol = IS.oracle_calc(g1, prm, lo)[0]
    step = (hi["idepth_new"].astype(np.float64) - lo["idepth_new"].astype(np.float64))
    num = (oh["energy_new"][:, 0].astype(np.float64) - ol["energy_new"][:, 0]) / step
    # jb[8] after the Schur loop = sum r dd + coupling * (idepth - iR)
    rdd = o["jb"][:, 8].astype(np.float64) - prm.coupling_weight * (pts["idepth_new"].astype(np.float64) - pts["iR"])
    ana = 2 * rdd
    sel = inl & (oh["is_good_new"] == 1) & (ol["is_good_new"] == 1)
    assert sel.sum() > 50
    # the analytic form uses the interpolated central-difference gradient, the finite difference sees the bilinear facet the
    # sample sits on: they agree as a population (slope and correlation), not per point
    slope = float(np.dot(num[sel], ana[sel]) / np.dot(ana[sel], ana[sel]))
    corr = float(np.corrcoef(num[sel], ana[sel])[0, 1])
    assert 0.9 < slope < 1.1 and corr > 0.9, (slope, corr)


@pytest.mark.parametrize("algorithm", [0, 1])
def test_pnp_restatement_recovers_pose_and_flags_outliers(algorithm):
    """SURVEY §8 f4 (IndirectCameraOptimizer over g2o).  Functional pin of the restatement: from a perturbed start the 4
    rounds land on the true pose, the planted gross outliers are flagged, the covariance is the diagonal of (J^T W J)^-1."""
    from tests import pnp_setup as PS
    S = PS.scene()
    m = S["matches"]
    if algorithm == 1:
        m = m.copy(); m["inv_sigma2"] = m["info"]                 # the Gauss-Newton overload uses 1/scaleFactor^2 for both
    out = np.zeros(len(m), np.uint8)
    r = PS.oracle_pnp(S["R0"], S["t0"], S["K"], m, out, algorithm=algorithm, compute_covariance=True)
    assert r.is_ok == 1 and r.rounds == 4
    R = np.array(list(r.R)).reshape(3, 3); t = np.array(list(r.t))
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12
    ang = np.arccos(np.clip((np.trace(R @ S["R_true"].T) - 1) / 2, -1, 1))
    assert ang < 2e-3 and np.linalg.norm(t - S["t_true"]) < 2e-2, (ang, t - S["t_true"])
    planted = S["planted"]
    assert out[planted].mean() > 0.98 and out[~planted].mean() < 0.15
    assert r.n_bad == int(out.sum())
    # start pose error was ~10x larger
    ang0 = np.arccos(np.clip((np.trace(S["R0"] @ S["R_true"].T) - 1) / 2, -1, 1))
    assert ang0 > 5 * ang
    cov = np.array(list(r.covariance))
    assert np.all(cov > 0) and np.all(cov < 1e-2)
    # chi2 decreases over the robust rounds as outliers are excluded
    assert r.chi2[1] <= r.chi2[0] * 1.0001


def test_local_ba_structure_only_restatement():
    """SURVEY §8 f4 (IndirectBundleAdjustment, fixFrames: g2o StructureOnlySolver).  Functional pin: the points move towards
    the truth, the robust cost of every track does not increase, the planted gross outliers fail apply()'s chi2 test; the
    3x3 LDLT equals the restatement that is pinned on the vendored Eigen."""
    from tests import lba_setup as LS
    rng = np.random.default_rng(0)
    L = O.lib()
    for _ in range(50):
        A = rng.normal(size=(3, 3)); A = A @ A.T + 1e-3 * np.eye(3); b = rng.normal(size=3)
        x = np.zeros(3)
        pos = L.orc_ldlt3(O.ptr(np.ascontiguousarray(A), C.c_double), O.ptr(b, C.c_double), O.ptr(x, C.c_double))
        assert pos == 1 and np.array_equal(x, O.ldlt_solve(A, b)[0])
    ind = np.diag([1.0, -2.0, 3.0]); x = np.zeros(3)
    assert L.orc_ldlt3(O.ptr(ind, C.c_double), O.ptr(np.ones(3), C.c_double), O.ptr(x, C.c_double)) == 0
    S = LS.scene()
    fr, pts, bad, r = LS.oracle_lba(S["frames"], S["points"], S["off"], S["edges"], fix_frames=True, num_iterations=5)
    assert r.ok == 1 and r.iterations_done[0] == 5 and r.iterations_done[1] == 0
    assert np.array_equal(fr, S["frames"])                                   # poses untouched
    e0 = np.linalg.norm(S["points"] - S["truth"], axis=1); e1 = np.linalg.norm(pts - S["truth"], axis=1)
    clean = np.array([not S["planted"][S["off"][p]:S["off"][p + 1]].any() for p in range(len(pts))])
    assert np.median(e1[clean]) < 0.5 * np.median(e0[clean]), (np.median(e0[clean]), np.median(e1[clean]))
    assert bad[S["planted"]].mean() > 0.9 and bad[~S["planted"]].mean() < 0.1
    assert r.n_bad == int(bad.sum())
    # the refinement pass (kernel off, bad edges at level 1) keeps converging
    fr2, pts2, bad2, r2 = LS.oracle_lba(S["frames"], S["points"], S["off"], S["edges"], fix_frames=True, num_iterations=5, refine_iterations=3)
    assert r2.iterations_done[1] == 3
    assert np.isfinite(pts2).all()


def test_local_ba_levenberg_restatement():
    """SURVEY §8 f4 (IndirectBundleAdjustment, free poses: g2o Levenberg + Schur).  Functional pin: from perturbed local
    keyframes and points the optimisation lowers the robust cost monotonically over the passes, pulls the local poses back to
    where the observations were made (the gauge is held by the fixed keyframes) and leaves the fixed keyframes alone."""
    from tests import lba_setup as LS
    S = LS.scene(pose_noise=0.02, n_points=600, seed=3)
    nl = int((S["frames"]["fixed"] == 0).sum())
    fr, pts, bad, r = LS.oracle_lba(S["frames"], S["points"], S["off"], S["edges"], fix_frames=False, num_iterations=10, refine_iterations=5)
    assert r.ok == 1 and 1 <= r.iterations_done[0] <= 10 and 1 <= r.iterations_done[1] <= 5
    assert np.array_equal(fr[nl:], S["frames"][nl:])
    t_err0 = np.linalg.norm(S["frames"]["t"][:nl] - S["frames_true"]["t"][:nl], axis=1)
    t_err1 = np.linalg.norm(fr["t"][:nl] - S["frames_true"]["t"][:nl], axis=1)
    assert np.median(t_err1) < 0.2 * np.median(t_err0), (t_err0, t_err1)
    for f in range(nl):
        R = fr["R"][f].reshape(3, 3)
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12
    e0 = np.linalg.norm(S["points"] - S["truth"], axis=1); e1 = np.linalg.norm(pts - S["truth"], axis=1)
    assert np.median(e1) < 0.5 * np.median(e0)
    assert bad[S["planted"]].mean() > 0.9 and bad[~S["planted"]].mean() < 0.1
    # one Levenberg pass alone: chi2 of the pass is below the starting cost of a pass with zero iterations' worth of progress
    _, _, _, r1 = LS.oracle_lba(S["frames"], S["points"], S["off"], S["edges"], fix_frames=False, num_iterations=1, refine_iterations=0)
    assert r.chi2[0] < r1.chi2[0]
