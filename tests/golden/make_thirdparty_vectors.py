"""Generate tests/golden/thirdparty_vectors.npz from oracle/_ref (the reference's vendored Eigen 3.4.0 and
Sophus 1.1.0, compiled where they lie under /root/reference/thirdparty by oracle/Makefile `ref`).

Run in the build container only:  python tests/golden/make_thirdparty_vectors.py
The .npz holds inputs and the outputs of the REAL third-party code for the calls the hot path makes
(SE3::exp/log/Adj/Dx_exp_x/product/inverse, LDLT solve, inverse, the JacobiSVD nullspace projection),
so the oracle's plain-C restatements stay pinned on the GPU box where neither /root/reference nor
(necessarily) oracle/_ref exist.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from tests import oracle_lib as O  # noqa: E402

R = O.ref()
assert R, "oracle/_ref not built (needs /root/reference)"
d = C.c_double
rng = np.random.default_rng(20260929)
P = O.ptr

xi = rng.normal(size=(48, 6)) * np.array([1, 1, 1, .6, .6, .6])
xi[:8, 3:] *= 1e-7          # small-angle branch
xi[8:12, 3:] = 0            # exact zero rotation
xi[12:16, 3:] *= 4.5        # angles near/over pi
q = np.zeros((48, 4)); t = np.zeros((48, 3)); lg = np.zeros((48, 6)); adj = np.zeros((48, 36)); dx = np.zeros((48, 42))
Rm = np.zeros((48, 9)); qi = np.zeros((48, 4)); ti = np.zeros((48, 3)); qm = np.zeros((48, 4)); tm = np.zeros((48, 3))
qr = np.zeros((48, 4))
for i in range(48):
    R.ref_se3_exp(P(xi[i], d), P(q[i], d), P(t[i], d))
    R.ref_se3_log(P(q[i], d), P(t[i], d), P(lg[i], d))
    R.ref_se3_adj(P(q[i], d), P(t[i], d), P(adj[i], d))
    R.ref_se3_dx_exp_x(P(xi[i], d), P(dx[i], d))
    R.ref_se3_matrix(P(q[i], d), P(Rm[i], d))
    R.ref_se3_inv(P(q[i], d), P(t[i], d), P(qi[i], d), P(ti[i], d))
for i in range(48):   # products need all q/t filled first
    j = (i + 7) % 48
    R.ref_se3_mul(P(q[i], d), P(t[i], d), P(q[j], d), P(t[j], d), P(qm[i], d), P(tm[i], d))
    R.ref_se3_from_Rt(P(Rm[i], d), P(t[i], d), P(qr[i], d))

out = dict(xi=xi, q=q, t=t, log=lg, adj=adj, dx_exp_x=dx, R=Rm, q_inv=qi, t_inv=ti, q_mul=qm, t_mul=tm, q_from_R=qr)
for n in (6, 7, 8, 64, 160):
    M = rng.normal(size=(n, n + 3))
    A = M @ M.T
    if n >= 64:   # badly scaled like the Jacobi-scaled BA system with gauge priors
        s = 10.0 ** rng.uniform(-3, 3, size=n)
        A = A * s[:, None] * s[None, :]
    b = rng.normal(size=n)
    x = np.zeros(n); Ai = np.zeros(n * n)
    R.ref_ldlt_solve(P(A.ravel(), d), P(b, d), n, P(x, d))
    R.ref_inverse(P(A.ravel(), d), n, P(Ai, d))
    out[f"ldlt_A{n}"] = A; out[f"ldlt_b{n}"] = b; out[f"ldlt_x{n}"] = x; out[f"inv{n}"] = Ai.reshape(n, n)
# indefinite + singular LDLT cases (pivoting / zero-pivot paths)
A = rng.normal(size=(8, 8)); A = A + A.T; b = rng.normal(size=8); x = np.zeros(8)
R.ref_ldlt_solve(P(A.ravel(), d), P(b, d), 8, P(x, d)); out["ldlt_indef_A"] = A; out["ldlt_indef_b"] = b; out["ldlt_indef_x"] = x
v = rng.normal(size=(8, 5)); A = v @ v.T; b = A @ rng.normal(size=8); x = np.zeros(8)
R.ref_ldlt_solve(P(A.ravel(), d), P(b, d), 8, P(x, d)); out["ldlt_sing_A"] = A; out["ldlt_sing_b"] = b; out["ldlt_sing_x"] = x
for name, n, dup in (("orth68", 68, False), ("orth164", 164, False), ("orth_rankdef", 68, True)):
    Nc = rng.normal(size=(7, n)); Nc[:, :4] = 0
    if dup:
        Nc[6] = 3.0 * Nc[1]
    b = rng.normal(size=n); b2 = b.copy()
    R.ref_orthogonalize(P(b2, d), n, P(Nc.ravel(), d), 7, d(1e-5))
    out[name + "_N"] = Nc; out[name + "_b"] = b; out[name + "_out"] = b2
# ---- the Eigen arithmetic behind g2o's SE3Quat, Matrix3d::inverse, LDLT<Matrix3d>, LLT (pose-only optimisation, local BA)
u = rng.normal(size=(40, 6)) * np.array([.5, .5, .5, 1, 1, 1])
u[:6, :3] *= 1e-7; u[6:9, :3] = 0; u[9:14, :3] *= 5.5
gq = np.zeros((40, 4)); gt = np.zeros((40, 3)); gR = np.zeros((40, 9)); gq2 = np.zeros((40, 4)); gt2 = np.zeros((40, 3))
gqm = np.zeros((40, 4)); gtm = np.zeros((40, 3)); gX = rng.normal(size=(40, 3)) * 5; gmap = np.zeros((40, 3))
for i in range(40):
    R.ref_g2o_exp(P(u[i], d), P(gq[i], d), P(gt[i], d))
    R.ref_quat_to_matrix(P(gq[i], d), P(gR[i], d))
    R.ref_g2o_from_Rt(P(gR[i], d), P(gt[i], d), P(gq2[i], d), P(gt2[i], d))
    R.ref_g2o_map(P(gq[i], d), P(gt[i], d), P(gX[i], d), P(gmap[i], d))
for i in range(40):
    j = (i + 11) % 40
    R.ref_g2o_mul(P(gq[i], d), P(gt[i], d), P(gq[j], d), P(gt[j], d), P(gqm[i], d), P(gtm[i], d))
out.update(g2o_u=u, g2o_q=gq, g2o_t=gt, g2o_R=gR, g2o_q_from_R=gq2, g2o_X=gX, g2o_map=gmap, g2o_q_mul=gqm, g2o_t_mul=gtm)
A3 = np.zeros((24, 9)); b3 = rng.normal(size=(24, 3)); x3 = np.zeros((24, 3)); pos3 = np.zeros(24, np.int32); inv3 = np.zeros((24, 9))
for i in range(24):
    M = rng.normal(size=(3, 3))
    A = M @ M.T + 1e-3 * np.eye(3) if i < 16 else (M + M.T)                  # SPD, then indefinite
    if i in (14, 15):
        A = A * np.array([1e4, 1.0, 1e-4])[:, None] * np.array([1e4, 1.0, 1e-4])[None, :]
    A3[i] = A.ravel()
    pos3[i] = R.ref_ldlt3(P(A3[i], d), P(b3[i], d), P(x3[i], d))
    R.ref_mat3_inverse(P(A3[i], d), P(inv3[i], d))
out.update(ldlt3_A=A3, ldlt3_b=b3, ldlt3_x=x3, ldlt3_pos=pos3, inv3=inv3)
for n in (6, 36, 126):
    M = rng.normal(size=(n, n + 4)); A = M @ M.T; b = rng.normal(size=n); x = np.zeros(n)
    ok = R.ref_llt_solve(P(A.ravel(), d), n, P(b, d), P(x, d))
    out[f"llt_A{n}"] = A; out[f"llt_b{n}"] = b; out[f"llt_x{n}"] = x; out[f"llt_ok{n}"] = np.int32(ok)
A = rng.normal(size=(6, 6)); A = A + A.T; x = np.zeros(6)
out["llt_indef_A"] = A; out["llt_indef_ok"] = np.int32(R.ref_llt_solve(P(A.ravel(), d), 6, P(np.ones(6), d), P(x, d)))
# ---- Eigen expression shapes of the projection arithmetic (float / double 3x3 products, homogeneous product, 3x3 inverse)
f = C.c_float
ne = 400
eM = rng.normal(size=(ne, 9)).astype(np.float32); ev = (rng.normal(size=(ne, 3)) * 100).astype(np.float32); ev[::2, 2] = 1
et = rng.normal(size=(ne, 3)).astype(np.float32); es = rng.normal(size=ne).astype(np.float32)
e_aff_p = np.zeros((ne, 3), np.float32); e_aff_m = np.zeros((ne, 3), np.float32); e_noal = np.zeros((ne, 3), np.float32)
eB = rng.normal(size=(ne, 9)).astype(np.float32); e_mm3f = np.zeros((ne, 9), np.float32); e_inv3f = np.zeros((ne, 9), np.float32)
eK = np.zeros((ne, 9), np.float32); e_invK = np.zeros((ne, 9), np.float32)
dM = rng.normal(size=(ne, 9)); dv = rng.normal(size=(ne, 3)) * 100; dt = rng.normal(size=(ne, 3)); ds = rng.normal(size=ne)
d_hom = np.zeros((ne, 3)); d_mv = np.zeros((ne, 3)); dB = rng.normal(size=(ne, 9)); d_mm = np.zeros((ne, 9)); d_mmf = np.zeros((ne, 9), np.float32)
d_inv = np.zeros((ne, 9)); d_krki = np.zeros((ne, 9)); d_kt = np.zeros((ne, 3))
for i in range(ne):
    eK[i] = [300 + 100 * rng.normal(), 0, 300 + 50 * rng.normal(), 0, 300 + 100 * rng.normal(), 200 + 50 * rng.normal(), 0, 0, 1]
    R.ref_eig_matvec3f_affine(P(eM[i], f), P(ev[i], f), P(et[i], f), f(es[i]), 1, P(e_aff_p[i], f))
    R.ref_eig_matvec3f_affine(P(eM[i], f), P(ev[i], f), P(et[i], f), f(es[i]), -1, P(e_aff_m[i], f))
    R.ref_eig_matvec3f_noalias(P(eM[i], f), P(ev[i], f), P(et[i], f), f(es[i]), P(e_noal[i], f))
    R.ref_eig_matmul3f(P(eM[i], f), P(eB[i], f), P(e_mm3f[i], f))
    R.ref_eig_inverse3f(P(eM[i], f), P(e_inv3f[i], f)); R.ref_eig_inverse3f(P(eK[i], f), P(e_invK[i], f))
    R.ref_eig_homog3d(P(dM[i], d), P(dv[i, :2].copy(), d), P(dt[i], d), d(ds[i]), P(d_hom[i], d))
    R.ref_eig_matvec3d(P(dM[i], d), P(dv[i], d), P(d_mv[i], d))
    R.ref_eig_matmul3d(P(dM[i], d), P(dB[i], d), P(d_mm[i], d), P(d_mmf[i], f))
    R.ref_eig_inverse3d(P(dM[i], d), P(d_inv[i], d))
    R.ref_eig_krki(P(eK[i].astype(np.float64), d), P(dM[i], d), P(dt[i], d), P(d_krki[i], d), P(d_kt[i], d))
out.update(eig_M3f=eM, eig_v3f=ev, eig_t3f=et, eig_s3f=es, eig_affine_plus=e_aff_p, eig_affine_minus=e_aff_m, eig_noalias=e_noal,
           eig_B3f=eB, eig_matmul3f=e_mm3f, eig_inverse3f=e_inv3f, eig_K3f=eK, eig_inverseK3f=e_invK,
           eig_M3d=dM, eig_v3d=dv, eig_t3d=dt, eig_s3d=ds, eig_homog3d=d_hom, eig_matvec3d=d_mv, eig_B3d=dB, eig_matmul3d=d_mm,
           eig_matmul3d_cast=d_mmf, eig_inverse3d=d_inv, eig_krki=d_krki, eig_kt=d_kt)
# ---- dot-product shapes of the linearised-residual algebra and of the back-substitution (BA.cpp:1470,1478,1699,2166,2219)
R.ref_eig_jp_delta.restype = C.c_float; R.ref_eig_calib_dot.restype = C.c_double
nd = 400
jx = rng.normal(size=(nd, 6)).astype(np.float32); jdp = rng.normal(size=(nd, 8)).astype(np.float32); jc = rng.normal(size=(nd, 4)).astype(np.float32)
jcd = rng.normal(size=(nd, 4)); jdd = rng.normal(size=nd).astype(np.float32); jde = rng.normal(size=nd).astype(np.float32)
jp0 = np.zeros(nd, np.float32); jp1 = np.zeros(nd, np.float32)
cst = rng.normal(size=(nd, 4)); ca = rng.normal(size=(nd, 4)).astype(np.float32); cl = rng.normal(size=(nd, 4)).astype(np.float32); cdot = np.zeros(nd)
for i in range(nd):
    jp0[i] = R.ref_eig_jp_delta(P(jx[i], f), P(jdp[i], f), P(jc[i], f), P(jcd[i], d), f(jdd[i]), f(jde[i]), 0)
    jp1[i] = R.ref_eig_jp_delta(P(jx[i], f), P(jdp[i], f), P(jc[i], f), P(jcd[i], d), f(jdd[i]), f(jde[i]), 1)
    cdot[i] = R.ref_eig_calib_dot(P(cst[i], d), P(ca[i], f), P(cl[i], f))
out.update(jp_Jxi=jx, jp_dp=jdp, jp_Jc=jc, jp_cdelta=jcd, jp_Jpdd=jdd, jp_dd=jde, jp_delta_vec4f=jp0, jp_delta_cast=jp1,
           calib_step=cst, calib_A=ca, calib_L=cl, calib_dot=cdot)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "thirdparty_vectors.npz"), **out)
print("wrote", len(out), "arrays")
