// reproj_dev.h — device code of the hybrid ORB term (reproj.hip), shared with the solve kernel of the resident iteration
// (ba_accumulate.hip), in whose launch the per-frame workgroups of the term run beside the factorisation.
#pragma once
#include "cmlhip_internal.h"
#include "../host/se3.h"

struct FramePre { double q[4]; double D[42]; };      // CML quaternion (w,x,y,z) of R, and Dx_exp_x(log(T)) 7x6

__device__ inline void d_hat(const double w[3], double O[9]) {
    O[0] = 0; O[1] = -w[2]; O[2] = w[1]; O[3] = w[2]; O[4] = 0; O[5] = -w[0]; O[6] = -w[1]; O[7] = w[0]; O[8] = 0;
}
__device__ inline void d_mm3(const double A[9], const double B[9], double C[9]) {
    double r[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    for (int i = 0; i < 9; i++) C[i] = r[i];
}
__device__ inline void d_mv3(const double A[9], const double v[3], double o[3]) {
    double r[3];
    for (int i = 0; i < 3; i++) r[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}

// SE3(R,t).log() — rotation matrix -> unit quaternion (Shoemake) -> atan-based log, then V^-1 t (Sophus 1.1.0 semantics)
__device__ inline void d_se3_log(const double R[9], const double t[3], double xi[6]) {
    double q[4];
    double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double s = sqrt(tr + 1.0);
        q[0] = 0.5 * s; s = 0.5 / s;
        q[1] = (R[7] - R[5]) * s; q[2] = (R[2] - R[6]) * s; q[3] = (R[3] - R[1]) * s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        q[1 + i] = 0.5 * s; s = 0.5 / s;
        q[0] = (R[k * 3 + j] - R[j * 3 + k]) * s;
        q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
        q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
    }
    const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= nq;
    const double eps = 1e-10;
    const double sq = q[1] * q[1] + q[2] * q[2] + q[3] * q[3], w = q[0];
    double f, theta;
    if (sq < eps * eps) {
        f = 2.0 / w - (2.0 / 3.0) * sq / (w * w * w);
        theta = 2.0 * sq / w;
    } else {
        const double n = sqrt(sq);
        const double at = (w < 0) ? atan2(-n, -w) : atan2(n, w);
        f = 2.0 * at / n;
        theta = f * n;
    }
    const double om[3] = {f * q[1], f * q[2], f * q[3]};
    double O[9], O2[9], Vi[9];
    d_hat(om, O);
    d_mm3(O, O, O2);
    double c;
    if (fabs(theta) < eps) c = 1.0 / 12.0;
    else { const double ht = 0.5 * theta; c = (1.0 - theta * cos(ht) / (2.0 * sin(ht))) / (theta * theta); }
    for (int i = 0; i < 9; i++) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * O[i] + c * O2[i];
    d_mv3(Vi, t, xi);
    xi[3] = om[0]; xi[4] = om[1]; xi[5] = om[2];
}

// d[qx qy qz qw tx ty tz]/d[upsilon omega] of SE3::exp (rows in Sophus parameter order), from the definitions
// q = (cos(th/2), sin(th/2)/th w), t = V(w) u, V = I + B W + C W^2.
__device__ inline void d_dx_exp_x(const double xi[6], double J[42]) {
    const double* u = xi; const double* w = xi + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    for (int i = 0; i < 42; i++) J[i] = 0;
    if (th2 < 1e-10) {
        J[0 * 6 + 3] = 0.5; J[1 * 6 + 4] = 0.5; J[2 * 6 + 5] = 0.5;
        J[4 * 6 + 0] = 1; J[5 * 6 + 1] = 1; J[6 * 6 + 2] = 1;
        const double ux = 0.5 * u[0], uy = 0.5 * u[1], uz = 0.5 * u[2];
        J[4 * 6 + 4] = uz; J[4 * 6 + 5] = -uy; J[5 * 6 + 3] = -uz; J[5 * 6 + 5] = ux; J[6 * 6 + 3] = uy; J[6 * 6 + 4] = -ux;
        return;
    }
    const double th = sqrt(th2), hth = 0.5 * th;
    const double a = sin(hth) / th, c = cos(hth), da = (0.5 * c - a) / th;
    for (int j = 0; j < 3; j++) {
        for (int i = 0; i < 3; i++) J[i * 6 + 3 + j] = (i == j ? a : 0.0) + w[i] * w[j] * da / th;
        J[3 * 6 + 3 + j] = -0.5 * a * w[j];
    }
    const double B = (1.0 - cos(th)) / th2, C = (th - sin(th)) / (th2 * th);
    const double dB = (th * sin(th) - 2.0 * (1.0 - cos(th))) / (th2 * th);
    const double dC = ((1.0 - cos(th)) * th - 3.0 * (th - sin(th))) / (th2 * th2);
    double W[9], W2[9];
    d_hat(w, W);
    d_mm3(W, W, W2);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) J[(4 + i) * 6 + j] = ((i == j) ? 1.0 : 0.0) + B * W[i * 3 + j] + C * W2[i * 3 + j];
    for (int j = 0; j < 3; j++) {
        double e[3] = {0, 0, 0}, G[9], GW[9], WG[9], dV[9], dt[3];
        e[j] = 1;
        d_hat(e, G);
        d_mm3(G, W, GW);
        d_mm3(W, G, WG);
        for (int i = 0; i < 9; i++) dV[i] = dB * (w[j] / th) * W[i] + B * G[i] + dC * (w[j] / th) * W2[i] + C * (GW[i] + WG[i]);
        d_mv3(dV, u, dt);
        for (int i = 0; i < 3; i++) J[(4 + i) * 6 + 3 + j] = dt[i];
    }
}

__device__ inline void reproj_frame_pre(const double* R, const double* t, FramePre& P) {
    // Quaternion::logHati + normalise, Rotation.cpp:205-221, Rotation.h:246-252
    double q[4];
    q[0] = sqrt(fmax(0.0, 1.0 + R[0] + R[4] + R[8])) / 2.0;
    q[1] = sqrt(fmax(0.0, 1.0 + R[0] - R[4] - R[8])) / 2.0;
    q[2] = sqrt(fmax(0.0, 1.0 - R[0] + R[4] - R[8])) / 2.0;
    q[3] = sqrt(fmax(0.0, 1.0 - R[0] - R[4] + R[8])) / 2.0;
    q[1] = copysign(q[1], R[7] - R[5]); q[2] = copysign(q[2], R[2] - R[6]); q[3] = copysign(q[3], R[3] - R[1]);
    const double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; k++) P.q[k] = q[k] / nn;
    double xi[6];
    d_se3_log(R, t, xi);                // BA.cpp:2621-2622
    d_dx_exp_x(xi, P.D);                // BA.cpp:2623
}
__device__ __forceinline__ double d_tukey(double v, double th) {            // Derivative.h:35-39
    if (fabs(v) > th) return 0;
    const double l = 1.0 - (v * v) / (th * th);
    return v * l * l;
}
__device__ __forceinline__ double d_dtukey(double v, double d, double th) { // Derivative.h:108-113 (v is the loss value, literal)
    if (fabs(v) > th) return 0;
    const double v2 = v * v, o = 1.0 - v2;
    return d * (-4.0 * v2 * o + o * o);
}


// indirectX = M.ldlt().solve(-bM) with M(i,i) *= (1+lambda) (BA.cpp:2695-2700); M is block diagonal, so one 6x6
// diagonally-pivoted LDL^T per frame (Eigen LDLT.h:300-396 / :560-600 semantics incl. the zero-pivot rule).  One lane.
// A (36), x (6), tr (6): the caller's work storage — the pivoting indexes them dynamically, so private arrays live in a scratch frame (memory
// round trips on a single lane, and a kernel with a scratch frame pays for it at every dispatch): the frame workgroup passes LDS.
__device__ inline void reproj_solve6(const double* __restrict__ Min /* 6x6 row-major */, const double* __restrict__ bin, double lambda, double* __restrict__ xout,
                                     double* __restrict__ A, double* __restrict__ x, int* __restrict__ tr) {
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j < 6; j++) A[i * 6 + j] = Min[i * 6 + j];
        A[i * 6 + i] *= (1 + lambda);
        x[i] = -bin[i];
    }
    for (int k = 0; k < 6; k++) {
        int big = k; double best = fabs(A[k * 6 + k]);
        for (int i = k + 1; i < 6; i++) if (fabs(A[i * 6 + i]) > best) { best = fabs(A[i * 6 + i]); big = i; }
        tr[k] = big;
        if (k != big) {
            for (int j = 0; j < k; j++) { double t = A[k * 6 + j]; A[k * 6 + j] = A[big * 6 + j]; A[big * 6 + j] = t; }
            for (int i = big + 1; i < 6; i++) { double t = A[i * 6 + k]; A[i * 6 + k] = A[i * 6 + big]; A[i * 6 + big] = t; }
            { double t = A[k * 6 + k]; A[k * 6 + k] = A[big * 6 + big]; A[big * 6 + big] = t; }
            for (int i = k + 1; i < big; i++) { double t = A[i * 6 + k]; A[i * 6 + k] = A[big * 6 + i]; A[big * 6 + i] = t; }
        }
        if (k > 0) {
            double temp[6], s = 0;
            for (int j = 0; j < k; j++) { temp[j] = A[j * 6 + j] * A[k * 6 + j]; s += A[k * 6 + j] * temp[j]; }
            A[k * 6 + k] -= s;
            for (int i = k + 1; i < 6; i++) {
                double s2 = 0;
                for (int j = 0; j < k; j++) s2 += A[i * 6 + j] * temp[j];
                A[i * 6 + k] -= s2;
            }
        }
        const double akk = A[k * 6 + k];
        if (k == 0 && !(fabs(akk) > 0.0)) { for (int j = 0; j < 6; j++) tr[j] = j; break; }
        if (fabs(akk) > 0.0) for (int i = k + 1; i < 6; i++) A[i * 6 + k] /= akk;
    }
    for (int k = 0; k < 6; k++) if (tr[k] != k) { double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
    for (int i = 0; i < 6; i++) { double s = x[i]; for (int j = 0; j < i; j++) s -= A[i * 6 + j] * x[j]; x[i] = s; }
    for (int i = 0; i < 6; i++) x[i] = (fabs(A[i * 6 + i]) > 2.2250738585072014e-308) ? x[i] / A[i * 6 + i] : 0.0;
    for (int i = 5; i >= 0; i--) { double s = x[i]; for (int j = i + 1; j < 6; j++) s -= A[j * 6 + i] * x[j]; x[i] = s; }
    for (int k = 5; k >= 0; k--) if (tr[k] != k) { double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
    for (int i = 0; i < 6; i++) xout[i] = x[i];
}

// One observation (BA.cpp:2607-2660, Residual.h:59-100): f = J_cam^T D (6), the Tukey loss value, the point Jacobian; false when the
// observation is not used (:2630 or a non-finite derivative).
__device__ __forceinline__ bool reproj_one(const double* __restrict__ R, const double* __restrict__ t, const FramePre& pre, const double* __restrict__ X,
                                           double gx, double gy, double fx, double fy, double f[6], double& res_out, double Jp[3]) {
    double T[3];
    for (int a = 0; a < 3; a++) T[a] = (R[a * 3] * X[0] + R[a * 3 + 1] * X[1] + R[a * 3 + 2] * X[2]) + t[a];
    const double dx = T[0] / T[2] - gx, dy = T[1] / T[2] - gy;
    const double norm = sqrt(dx * dx + dy * dy);
    const double th = 3.0 / sqrt(fx * fx + fy * fy);
    const double res = d_tukey(norm, th);
    double cam[7];
    bool ok = true;
    for (int a = 0; a < 3; a++) {
        const double d[3] = {R[a], R[3 + a], R[6 + a]};                                   // Camera.h:317-321: R e_a
        const double hx = (d[0] * T[2] - T[0] * d[2]) / (T[2] * T[2]), hy = (d[1] * T[2] - T[1] * d[2]) / (T[2] * T[2]);
        const double dsq = 2.0 * hx * dx + 2.0 * hy * dy;
        double v = (norm == 0) ? 0 : dsq / (2.0 * norm);
        v = d_dtukey(res, v, th);
        ok = ok && isfinite(v);
        cam[a] = v; Jp[a] = -v;
    }
    const double* q = pre.q;
    const double Pt[3] = {X[0] + t[0], X[1] + t[1], X[2] + t[2]};                         // Camera.h:323-325: R'(q)_a (P + t)
    for (int a = 0; a < 4; a++) {
        const double qa = q[0], qb = q[1], qc = q[2], qd = q[3];
        double _2b2 = 0, _2c2 = 0, _2d2 = 0, _2bc = 0, _2ad = 0, _2bd = 0, _2ac = 0, _2cd = 0, _2ab = 0;
        if (a == 0) { _2ad = 2 * qd; _2ac = 2 * qc; _2ab = 2 * qb; }
        else if (a == 1) { _2b2 = 4 * qb; _2bc = 2 * qc; _2bd = 2 * qd; _2ab = 2 * qa; }
        else if (a == 2) { _2c2 = 4 * qc; _2bc = 2 * qb; _2ac = 2 * qa; _2cd = 2 * qd; }
        else { _2d2 = 4 * qd; _2ad = 2 * qa; _2bd = 2 * qb; _2cd = 2 * qc; }
        const double D[9] = {-_2c2 - _2d2, _2bc - _2ad, _2bd + _2ac, _2bc + _2ad, -_2b2 - _2d2, _2cd - _2ab,
                             _2bd - _2ac, _2cd + _2ab, -_2b2 - _2c2};
        double d[3];
        for (int b = 0; b < 3; b++) d[b] = D[b * 3] * Pt[0] + D[b * 3 + 1] * Pt[1] + D[b * 3 + 2] * Pt[2];
        const double hx = (d[0] * T[2] - T[0] * d[2]) / (T[2] * T[2]), hy = (d[1] * T[2] - T[1] * d[2]) / (T[2] * T[2]);
        const double dsq = 2.0 * hx * dx + 2.0 * hy * dy;
        double v = (norm == 0) ? 0 : dsq / (2.0 * norm);
        v = d_dtukey(res, v, th);
        ok = ok && isfinite(v);
        cam[3 + a] = v;
    }
    for (int c = 0; c < 6; c++) {                                                         // BA.cpp:2641
        double s = 0;
        for (int r = 0; r < 7; r++) s += cam[r] * pre.D[r * 6 + c];
        f[c] = s;
    }
    res_out = res;
    return ok && !(res > 4 * 4);                                                          // BA.cpp:2630
}

// Frame-sorted observation list of one call: off[N+1] into obs / orig (orig[k] = the caller's index of sorted observation k).
struct ReprojArgs {
    int N;
    const double* poses;                       // N x 12 (R, t) given by the caller, or null: taken from the resident frame states
    const cmlhip_ba_frame_state* fs; double sc_t, sc_r;
    const int* off; const cmlhip_reproj_obs* obs; const int* orig; const double* points;
    double fx, fy, lambda;
    double* M6; double* b6;                    // 6N x 6N block-diagonal matrix and 6N vector (host path), may be null
    double* x6;                                // per-frame solution of the damped 6x6 system, may be null
    double* jp_obs; unsigned char* used;       // per observation, in the caller's numbering: point Jacobian (3) and the use flag
    int* ready; int ticket;                    // non-null: per-frame completion tickets for a consumer inside the same launch
};
#define RP_THREADS 512             // lanes of a frame's workgroup (the reduction order depends on it: the standalone launch and the solve launch use the same)
#define RP_LDS_DOUBLES (12 + 46 + (RP_THREADS / 64) * 27 + 27 + 36 + 6 + 36 + 6 + 3 + 6 + 1)      // ... + lane 0's M, b and the work storage of the 6x6 solve

// the whole term of frame `i` by one workgroup of RP_THREADS lanes; lds: RP_LDS_DOUBLES doubles
__device__ inline void reproj_frame_block(const ReprojArgs& a, int i, double* __restrict__ lds) {
    const int tid = threadIdx.x;
    double* sR = lds; FramePre* sPre = reinterpret_cast<FramePre*>(lds + 12); double* sW = lds + 58; double* sTot = sW + (RP_THREADS / 64) * 27;
    // a lane's first observation and its point are requested BEFORE lane 0 composes the pose (two dependent memory trips that need
    // nothing from it): they land while the SE(3) chain runs
    const int k0 = a.off[i] + tid, kend = a.off[i + 1];
    const bool has0 = k0 < kend;
    const cmlhip_reproj_obs o0 = a.obs[has0 ? k0 : 0];
    const int orig0 = a.orig[has0 ? k0 : 0];
    double X0[3];
    for (int c = 0; c < 3; c++) X0[c] = a.points[3 * (size_t)o0.point + c];
    if (tid == 0) {
        double R[9], t[3];
        if (a.poses) {
            for (int k = 0; k < 9; k++) R[k] = a.poses[12 * (size_t)i + k];
            for (int k = 0; k < 3; k++) t[k] = a.poses[12 * (size_t)i + 9 + k];
        } else {                               // PRE_worldToCam from the resident frame state (frame_step_block, ba_frames.h): exp(scaled state) * evaluation point
            using cml_amd::SE3;
            const cmlhip_ba_frame_state& S = a.fs[i];
            const double ss[6] = {a.sc_t * S.state[0], a.sc_t * S.state[1], a.sc_t * S.state[2], a.sc_r * S.state[3], a.sc_r * S.state[4], a.sc_r * S.state[5]};
            SE3 ev;
            for (int k = 0; k < 4; k++) ev.q[k] = S.eval_q[k];
            for (int k = 0; k < 3; k++) ev.t[k] = S.eval_t[k];
            const SE3 W = SE3::exp(ss) * ev;
            W.matrix(R);
            for (int k = 0; k < 3; k++) t[k] = W.t[k];
        }
        for (int k = 0; k < 9; k++) sR[k] = R[k];
        for (int k = 0; k < 3; k++) sR[9 + k] = t[k];
        reproj_frame_pre(R, t, *sPre);
    }
    __syncthreads();
    double acc[27];
    for (int e = 0; e < 27; e++) acc[e] = 0.0;
    for (int k = k0; k < kend; k += RP_THREADS) {                               // a lane's share of the list, in list order
        const bool first = k == k0;
        const cmlhip_reproj_obs o = first ? o0 : a.obs[k];
        double Xk[3];
        for (int c = 0; c < 3; c++) Xk[c] = first ? X0[c] : a.points[3 * (size_t)o.point + c];
        double f[6], res, Jp[3];
        const bool use = reproj_one(sR, sR + 9, *sPre, Xk, o.gx, o.gy, a.fx, a.fy, f, res, Jp);
        const int ok = first ? orig0 : a.orig[k];
        if (a.used) a.used[ok] = use ? 1 : 0;
        if (a.jp_obs) for (int c = 0; c < 3; c++) a.jp_obs[3 * (size_t)ok + c] = use ? Jp[c] : 0.0;
        if (use) {
            int idx = 0;
            for (int r = 0; r < 6; r++) for (int c = r; c < 6; c++) { acc[idx] += f[r] * f[c]; idx++; }
            for (int r = 0; r < 6; r++) acc[21 + r] += f[r] * res;                  // BA.cpp:2655
        }
    }
    // fixed-order sums inside the wave on the DPP path: row_shr 1 / 2 / 4 / 8 inside each 16-lane row, then row_bcast:15 and row_bcast:31
    // carry the row totals forward — the total is valid in lane 63.  (Round 3, first form: six __shfl_down steps per sum, i.e. 27 x 6
    // dependent LDS-crossbar round trips of two ds_bpermute each: 6.9 of the frame workgroup's 19.6 us; in-kernel stamps.)  The 27 chains
    // are independent: unrolled, they interleave.
#pragma unroll
    for (int e = 0; e < 27; e++) {
        double v = acc[e];
#define RP_DPP_ADD(ctrl, rmask) do { \
            const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xf, false); \
            const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xf, false); \
            v += __hiloint2double(hi_, lo_); } while (0)
        RP_DPP_ADD(0x111, 0xf);      // row_shr:1   (lanes without a source add +0.0)
        RP_DPP_ADD(0x112, 0xf);      // row_shr:2
        RP_DPP_ADD(0x114, 0xf);      // row_shr:4
        RP_DPP_ADD(0x118, 0xf);      // row_shr:8
        RP_DPP_ADD(0x142, 0xa);      // row_bcast:15 into rows 1 and 3
        RP_DPP_ADD(0x143, 0xc);      // row_bcast:31 into rows 2 and 3
#undef RP_DPP_ADD
        if ((tid & 63) == 63) sW[(tid >> 6) * 27 + e] = v;
    }
    __syncthreads();
    if (tid < 27) { double v = sW[tid]; for (int wq = 1; wq < RP_THREADS / 64; wq++) v += sW[27 * wq + tid]; sTot[tid] = v; }      // the waves in sequence
    __syncthreads();
    if (tid == 0) {
        double* Mf = sTot + 27; double* bf = Mf + 36;          // lane 0's scratchpads in LDS (see reproj_solve6)
        int idx = 0;
        for (int r = 0; r < 6; r++) for (int c = r; c < 6; c++) { Mf[r * 6 + c] = Mf[c * 6 + r] = sTot[idx]; idx++; }
        for (int r = 0; r < 6; r++) bf[r] = sTot[21 + r];
        const int m = 6 * a.N;
        if (a.M6) for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) a.M6[(size_t)(6 * i + r) * m + 6 * i + c] = Mf[r * 6 + c];
        if (a.b6) for (int r = 0; r < 6; r++) a.b6[6 * i + r] = bf[r];
        if (a.x6) {
            double* xs = bf + 6 + 36 + 6 + 3;
            reproj_solve6(Mf, bf, a.lambda, xs, bf + 6, bf + 6 + 36, reinterpret_cast<int*>(bf + 6 + 36 + 6));
            if (a.ready) {
                // consumed by ANOTHER workgroup of the same launch (the solve workgroup of the resident iteration): device-scope stores,
                // then, once they are acknowledged, the frame's ticket — the consumer polls it and reads the solution with device-scope loads
                for (int r = 0; r < 6; r++) __hip_atomic_store(a.x6 + 6 * (size_t)i + r, xs[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // device-scope stores acknowledged: no release fence (an L2 write-back) needed
                __hip_atomic_store(a.ready + i, a.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                for (int r = 0; r < 6; r++) a.x6[6 * (size_t)i + r] = xs[r];
            }
        }
    }
}

