// pnp.hip — the ORB side's pose-only optimisation as ONE launch.
// Replaces CML::Optimization::G2O::IndirectCameraOptimizer::optimize + evaluateOutliers
// (src/cml/optimization/g2o/IndirectCameraOptimizer.cpp:4-195 Levenberg, :197-382 Gauss-Newton, :384-427) and the slice of
// the vendored g2o it drives for one free VertexSE3Expmap over fixed points: EdgeSE3ProjectXYZ (g2o/types/sba/
// edge_project_xyz.cpp:44-95), SE3Quat (g2o/types/slam3d/se3quat.h), RobustKernelHuber (g2o/core/robust_kernel_impl.cpp:60-74),
// constructQuadraticForm (g2o/core/base_fixed_sized_edge.hpp:49-133), OptimizationAlgorithmLevenberg::solve
// (g2o/core/optimization_algorithm_levenberg.cpp:58-175), SparseOptimizer::optimize (g2o/core/sparse_optimizer.cpp:392-455).
//
// Design.  The reference builds a graph, and every one of its <= 40 iterations (x up to 10 Levenberg trials) walks the edge
// list twice on one core.  Here the whole 4-round optimisation is one workgroup that never leaves the CU: the matches sit in
// LDS (48 B each), every evaluation of the active set — errors, Huber weights, chi2 and the 6x6 system J^T W J | J^T W e — is
// one pass with the 2n rows reduced on the matrix cores (v_mfma_f64_16x16x4_f64, D = sum_k (w_k J_k) [J_k | -e_k]^T), and
// the 6x6 Cholesky, the SE(3) exponential and the Levenberg bookkeeping run on one lane between two barriers.  The system
// evaluated at an accepted trial pose IS the next iteration's buildSystem (same pose, same edges, same order), so each
// Levenberg trial costs one pass instead of g2o's two.  No host round trip until the result struct is read back.
#include "cmlhip_internal.h"
#include <cfloat>

#pragma clang fp contract(off)

#define PNP_THREADS 256
#define PNP_WAVES (PNP_THREADS / 64)
#define PNP_MAX_MATCHES CMLHIP_PNP_MAX_MATCHES             // 48 B of LDS per match (120 KB) + tiles: under the 160 KB of a CU
typedef double pnp_double4 __attribute__((ext_vector_type(4)));

struct PnpPose { double x, y, z, w, t[3]; };                 // Eigen coefficient order
struct PnpArgs {
    const cmlhip_pnp_match* m; int n;
    unsigned char* outliers;
    double R0[9], t0[3], K[4];
    int algorithm, check, cov;
    cmlhip_pnp_result* out;
};

// v_rcp_f64 / v_rsq_f64 seeds + two Newton steps with explicit fmas: <= 1 ulp, a third of the instructions of the IEEE
// division / square root sequences.  The parity bar of this path is 1e-9 (the reduction order differs from g2o's anyway).
__device__ __forceinline__ double pnp_rcp(double d) {
    double x = __builtin_amdgcn_rcp(d);
    x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
    x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
    return x;
}
__device__ __forceinline__ double pnp_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = __builtin_fma(0.5 * y, __builtin_fma(-d * y, y, 1.0), y);
    y = __builtin_fma(0.5 * y, __builtin_fma(-d * y, y, 1.0), y);
    return y;
}

// ---------------------------------------------------------------------------------------------- SE3Quat on one lane
__device__ static void pq_from_matrix(const double m[9], PnpPose& q) {     // Eigen Quaternion(Matrix3)
    const double tr = m[0] + m[4] + m[8];
    if (tr > 0) {
        const double r = pnp_rsqrt(tr + 1.0);
        double t = (tr + 1.0) * r;
        q.w = 0.5 * t; t = 0.5 * r;
        q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        double v[3];
        v[i] = 0.5 * t; t = 0.5 / t;
        q.w = (m[k * 3 + j] - m[j * 3 + k]) * t;
        v[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        v[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
}
__device__ static void pq_normalize(PnpPose& q) {                          // se3quat.h normalizeRotation
    if (q.w < 0) { q.x *= -1; q.y *= -1; q.z *= -1; q.w *= -1; }
    const double n = pnp_rsqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x *= n; q.y *= n; q.z *= n; q.w *= n;
}
__device__ __forceinline__ void pq_rotate(const PnpPose& q, const double v[3], double o[3]) {
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
    o[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
    o[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
__device__ static void pq_to_matrix(const PnpPose& q, double R[9]) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x,
                 tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__device__ static void pq_exp(const double u[6], PnpPose& T) {             // se3quat.h:201-229
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    const double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    const double theta = sqrt(th2);
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9], R[9], V[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
    double a, b, c, d;
    if (theta < 0.00001) { a = 1; b = 0.5; c = 0.5; d = 1. / 6.; }
    else {
        double sn, cs;
        sincos(theta, &sn, &cs);
        const double it = pnp_rcp(theta), it2 = it * it;
        a = sn * it; b = (1 - cs) * it2;
        c = b; d = (theta - sn) * (it2 * it);
    }
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + a * O[i] + b * O2[i];
        V[i] = I + c * O[i] + d * O2[i];
    }
    pq_from_matrix(R, T); pq_normalize(T);
    for (int i = 0; i < 3; i++) T.t[i] = V[i * 3] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
}
__device__ static void pq_mul(const PnpPose& A, const PnpPose& B, PnpPose& C) {   // se3quat.h:96-102
    double rt[3];
    pq_rotate(A, B.t, rt);
    PnpPose r;
    r.t[0] = A.t[0] + rt[0]; r.t[1] = A.t[1] + rt[1]; r.t[2] = A.t[2] + rt[2];
    r.w = A.w * B.w - A.x * B.x - A.y * B.y - A.z * B.z;
    r.x = A.w * B.x + A.x * B.w + A.y * B.z - A.z * B.y;
    r.y = A.w * B.y + A.y * B.w + A.z * B.x - A.x * B.z;
    r.z = A.w * B.z + A.z * B.w + A.x * B.y - A.y * B.x;
    pq_normalize(r);
    C = r;
}
// SimplicialLLT on the 6x6 block (+ lambda on the diagonal): false on a pivot that is not positive
__device__ static bool pnp_llt_solve(const double* Hin, double lambda, const double* b, double* x) {
    double L[36];
    for (int i = 0; i < 36; i++) L[i] = Hin[i];
    for (int i = 0; i < 6; i++) L[i * 6 + i] += lambda;
    double inv[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = L[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= L[j * 6 + k] * L[j * 6 + k];
        if (!(d > 0)) return false;
        inv[j] = pnp_rsqrt(d); L[j * 6 + j] = d * inv[j];
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double s = L[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = s * inv[j];
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { double s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k]; y[i] = s * inv[i]; }
#pragma unroll
    for (int i = 5; i >= 0; i--) { double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k]; x[i] = s * inv[i]; }
    return true;
}

// ---------------------------------------------------------------------------------------------- one pass over the edges
struct PnpShared {
    double tile[PNP_WAVES][64][9];   // per wave: one Jacobian row of each lane's edge, [6] = -e, [7] = rho' * omega
    double part[2 * PNP_WAVES][44];  // per wave and 8 x 8 block: 6x7 sums + chi
    double sys[2][44];           // [0] the system the solver works on, [1] the system at the trial pose (H 36 | b 6 | chi)
    PnpPose T, Ttrial, T0;
    // the rejection cascade: the trials that would follow a rejection, prepared side by side (PNP_SPEC = at most 9 of them)
    PnpPose specT[9]; double specLambda[9], specNi[9], specScale[9], specChi[9], specPart[PNP_WAVES][9]; int specOk[9];
    int ctl[8];                  // [0] loop again, [1] stop, [2] nBad, [3] adopt the trial system, [4] next trial chi2 only, [5] rebuild at the accepted pose
    unsigned char lvl[PNP_MAX_MATCHES];
};

// computeError, edge_project_xyz.cpp:44-50: p = (x/z, y/z, 1/z) of the mapped point
__device__ __forceinline__ void pnp_edge(const double* sm, int i, const PnpPose& T, const double K[4], double e[2], double p[3], double& om) {
    const double X[3] = {sm[6 * i], sm[6 * i + 1], sm[6 * i + 2]};
    double r[3];
    pq_rotate(T, X, r);
    const double iz = pnp_rcp(r[2] + T.t[2]);
    p[0] = (r[0] + T.t[0]) * iz; p[1] = (r[1] + T.t[1]) * iz; p[2] = iz;
    e[0] = sm[6 * i + 3] - (p[0] * K[0] + K[2]);
    e[1] = sm[6 * i + 4] - (p[1] * K[1] + K[3]);
    om = sm[6 * i + 5];
}

// computeActiveErrors + activeRobustChi2 + buildSystem at pose `T` into S.sys[dst]
// want_H == false: computeActiveErrors + activeRobustChi2 only (what a REJECTED Levenberg trial needs); the chi2 sum is the same
// code either way, so a trial evaluated light and then accepted is rebuilt in full at the same pose with an identical chi2
__device__ static void pnp_evaluate(PnpShared& S, const double* sm, int n, const PnpPose& T, const double K[4], bool robust, double delta, int dst, bool want_H = true) {
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, col = l & 15, kq = l >> 4, pk = col >> 3, c7 = col & 7;
    pnp_double4 acc0 = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.};
    double chi = 0.0;
    for (int base = 0; base < n; base += PNP_THREADS) {
        const int i = base + tid;
        const bool valid = i < n && !S.lvl[i < n ? i : 0];
        double J0[7], J1[7], w = 0.0;
#pragma unroll
        for (int k = 0; k < 7; k++) { J0[k] = 0.0; J1[k] = 0.0; }
        if (valid) {
            double e[2], p[3], om;
            pnp_edge(sm, i, T, K, e, p, om);
            const double chi2 = e[0] * (om * e[0]) + e[1] * (om * e[1]);
            double rho0 = chi2, rho1 = 1.0;
            if (robust) {                                                      // robust_kernel_impl.cpp:60-74
                const double dsqr = delta * delta;
                if (!(chi2 <= dsqr)) { const double rs = pnp_rsqrt(chi2), sq = chi2 * rs; rho0 = 2 * sq * delta - dsqr; rho1 = delta * rs; }
            }
            chi += rho0;
            if (want_H) {
            const double u = p[0], v = p[1], iz = p[2], fx = K[0], fy = K[1];                       // u = x/z, v = y/z
            J0[0] = u * v * fx; J0[1] = -(1 + u * u) * fx; J0[2] = v * fx;                          // edge_project_xyz.cpp:80-94
            J0[3] = -iz * fx; J0[4] = 0; J0[5] = u * iz * fx;
            J1[0] = (1 + v * v) * fy; J1[1] = -u * v * fy; J1[2] = -u * fy;
            J1[3] = 0; J1[4] = -iz * fy; J1[5] = v * iz * fy;
            J0[6] = -e[0]; J1[6] = -e[1];
            w = rho1 * om;
            }
        }
        if (!want_H) continue;                                                 // (workgroup-uniform)
        // row 0 of every edge of this wave, then row 1: D += sum_k (w_k J_k) [J_k | -e_k]^T.  Only 6 x 7 of the 16 x 16 tile is
        // wanted, so two edges ride in one K step: A rows / B columns 0..7 carry edge 2k, 8..15 carry edge 2k+1 (the two
        // off-diagonal 8 x 8 blocks hold cross terms nobody reads): 8 edge rows per instruction
#pragma unroll
        for (int k = 0; k < 7; k++) S.tile[wv][l][k] = J0[k];
        S.tile[wv][l][7] = w;
        // (LDS accesses of one wave are ordered: no barrier between the stores above and the loads below)
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const double* row = S.tile[wv][8 * m + 2 * kq + pk];
            const double v = c7 < 7 ? row[c7] : 0.0;
            const double av = c7 < 6 ? v * row[7] : 0.0;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, v, acc0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 7; k++) S.tile[wv][l][k] = J1[k];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const double* row = S.tile[wv][8 * m + 2 * kq + pk];
            const double v = c7 < 7 ? row[c7] : 0.0;
            const double av = c7 < 6 ? v * row[7] : 0.0;
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, v, acc1, 0, 0, 0);
        }
    }
    // D: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
        const int row = kq + 4 * rg;
        if ((row >> 3) == pk && (row & 7) < 6 && c7 < 7) S.part[2 * wv + pk][(row & 7) * 7 + c7] = acc0[rg] + acc1[rg];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) chi += __shfl_xor(chi, off, 64);
    if (l == 0) { S.part[2 * wv][42] = chi; S.part[2 * wv + 1][42] = 0.0; }
    __syncthreads();
    if (tid < 43) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 2 * PNP_WAVES; k++) v += S.part[k][tid];
        int at = 42;                                                           // chi
        if (tid < 42) { const int r = tid / 7, c = tid % 7; at = c < 6 ? r * 6 + c : 36 + r; }
        S.sys[dst][at] = v;
    }
    __syncthreads();
}

// activeRobustChi2 of the active set at m candidate poses in ONE pass over the edges (the poses of the trials a rejection cascade
// would try one after the other); the per-pose sums use the reduction of pnp_evaluate, so each equals what that trial would get
__device__ static void pnp_chi2_multi(PnpShared& S, const double* sm, int n, int m, const double K[4], bool robust, double delta) {
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    double chi[9];
#pragma unroll
    for (int j = 0; j < 9; j++) chi[j] = 0.0;
    for (int base = 0; base < n; base += PNP_THREADS) {
        const int i = base + tid;
        const bool valid = i < n && !S.lvl[i < n ? i : 0];
        if (!valid) continue;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            if (j < m) {                                                       // (uniform; no break, so that chi[] stays in registers)
                double e[2], p[3], om;
                pnp_edge(sm, i, S.specT[j], K, e, p, om);
                const double chi2 = e[0] * (om * e[0]) + e[1] * (om * e[1]);
                double rho0 = chi2;
                if (robust) {
                    const double dsqr = delta * delta;
                    if (!(chi2 <= dsqr)) { const double rs = pnp_rsqrt(chi2), sq = chi2 * rs; rho0 = 2 * sq * delta - dsqr; }
                }
                chi[j] += rho0;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 9; j++) {
        double v = chi[j];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (l == 0) S.specPart[wv][j] = v;
    }
    __syncthreads();
    if (tid < 9) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < PNP_WAVES; k++) { v += S.specPart[k][tid]; v += 0.0; }     // (+ 0.0: the empty second block of pnp_evaluate's sum)
        S.specChi[tid] = v;
    }
    __syncthreads();
}

__global__ __launch_bounds__(PNP_THREADS) void k_pnp_optimize(PnpArgs A) {
    extern __shared__ double s_m[];                   // n x {X(3), obs(2), inv_sigma2}
    __shared__ PnpShared S;
    const int tid = threadIdx.x, n = A.n;
    for (int i = tid; i < n; i += PNP_THREADS) {
        const cmlhip_pnp_match& M = A.m[i];
        s_m[6 * i] = M.X[0]; s_m[6 * i + 1] = M.X[1]; s_m[6 * i + 2] = M.X[2];
        s_m[6 * i + 3] = M.obs[0]; s_m[6 * i + 4] = M.obs[1]; s_m[6 * i + 5] = M.inv_sigma2;
        S.lvl[i] = A.outliers[i] ? 1 : 0;
    }
    if (tid == 0) {
        PnpPose T;
        pq_from_matrix(A.R0, T); pq_normalize(T);                              // SE3Quat(R, t)
        T.t[0] = A.t0[0]; T.t[1] = A.t0[1]; T.t[2] = A.t0[2];
        S.T0 = T; S.T = T;
    }
    __syncthreads();
    const double delta = (double)sqrtf(5.991f);                                // const float deltaMono = sqrt(5.991), :40
    bool robust = true;
    // thread-0 state of the Levenberg algorithm
    double lambda = 0.0, ni = 2.0, x[6] = {0, 0, 0, 0, 0, 0};
    int rounds = 0, nBad = 0, lmits[4] = {0, 0, 0, 0};
    double chis[4] = {0, 0, 0, 0};
    bool failed = false;
    for (int round = 0; round < 4; round++) {                                  // :139-168
        if (tid == 0) S.T = S.T0;
        __syncthreads();
        int done = 0;
        if (A.algorithm == CMLHIP_PNP_LEVENBERG) {
            pnp_evaluate(S, s_m, n, S.T, A.K, robust, delta, 0);
            for (int it = 0; it < 10; it++) {
                double currentChi = 0.0, rho = 0.0;
                int qmax = 0;
                bool accepted = false;
                if (tid == 0) {
                    currentChi = S.sys[0][42];
                    if (it == 0) {                                             // computeLambdaInit
                        double mx = 0.0;
                        for (int j = 0; j < 6; j++) mx = fmax(fabs(S.sys[0][j * 6 + j]), mx);
                        lambda = 1e-5 * mx; ni = 2.0;
                    }
                }
                bool again, accepted_light = false;
                do {
                    bool ok2 = false;
                    if (tid == 0) {
                        for (int j = 0; j < 6; j++) x[j] = 0.0;
                        ok2 = pnp_llt_solve(S.sys[0], lambda, S.sys[0] + 36, x);
                        PnpPose E, Tn;
                        pq_exp(x, E); pq_mul(E, S.T, Tn);                      // oplusImpl: exp(update) * estimate
                        S.Ttrial = Tn;
                    }
                    __syncthreads();
                    pnp_evaluate(S, s_m, n, S.Ttrial, A.K, robust, delta, 1);
                    if (tid == 0) {
                        double tempChi = S.sys[1][42];
                        if (!ok2) tempChi = DBL_MAX;
                        rho = currentChi - tempChi;
                        double scale = 0.0;
                        for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + S.sys[0][36 + j]);
                        scale += 1e-3;
                        rho /= scale;
                        bool brk = false;
                        if (rho > 0 && isfinite(tempChi)) {
                            const double r21 = 2 * rho - 1;
                            double alpha = 1. - r21 * r21 * r21;
                            alpha = fmin(alpha, 2. / 3.);
                            const double sf = fmax(1. / 3., alpha);
                            lambda *= sf; ni = 2.0; currentChi = tempChi;
                            S.T = S.Ttrial; accepted = true;
                        } else {
                            lambda *= ni; ni *= 2.0;                           // pop: S.T stays
                            if (!isfinite(lambda)) brk = true;
                        }
                        if (!brk) qmax++;
                        const bool more = !brk && rho < 0 && qmax < 10;
                        // A rejection is usually followed by more (g2o's cascade of up to ten at convergence is most of the
                        // evaluations of a call).  The trials that would follow only differ by lambda, so they are prepared side by
                        // side — one lane each for the solve and the exponential — their chi2 come from ONE pass over the edges, and
                        // the loop below replays g2o's decisions on them in order: same trials, same outcome, a fifth of the time.
                        S.ctl[6] = more ? 10 - qmax : 0;
                        S.specLambda[0] = lambda; S.specNi[0] = ni;
                        S.ctl[0] = 0;
                    }
                    __syncthreads();
                    const int m = S.ctl[6];
                    if (m > 0) {
                        if (tid < m) {
                            double lj = S.specLambda[0], nj = S.specNi[0];
                            for (int k = 0; k < tid; k++) { lj *= nj; nj *= 2.0; }
                            double xj[6] = {0, 0, 0, 0, 0, 0};
                            const bool okj = pnp_llt_solve(S.sys[0], lj, S.sys[0] + 36, xj);
                            PnpPose E, Tn;
                            pq_exp(xj, E); pq_mul(E, S.T, Tn);
                            double sc = 0.0;
                            for (int j = 0; j < 6; j++) sc += xj[j] * (lj * xj[j] + S.sys[0][36 + j]);
                            sc += 1e-3;
                            S.specT[tid] = Tn; S.specScale[tid] = sc; S.specOk[tid] = okj ? 1 : 0;
                            if (tid > 0) { S.specLambda[tid] = lj; S.specNi[tid] = nj; }
                        }
                        __syncthreads();
                        pnp_chi2_multi(S, s_m, n, m, A.K, robust, delta);
                        if (tid == 0) {
                            for (int j = 0; j < m; j++) {                      // the remaining trials, decided exactly as above
                                double tempChi = S.specOk[j] ? S.specChi[j] : DBL_MAX;
                                rho = (currentChi - tempChi) / S.specScale[j];
                                bool brk = false;
                                if (rho > 0 && isfinite(tempChi)) {
                                    const double r21 = 2 * rho - 1;
                                    double alpha = 1. - r21 * r21 * r21;
                                    alpha = fmin(alpha, 2. / 3.);
                                    const double sf = fmax(1. / 3., alpha);
                                    lambda *= sf; ni = 2.0; currentChi = tempChi;
                                    S.T = S.specT[j]; accepted = true; accepted_light = true;
                                } else {
                                    lambda *= ni; ni *= 2.0;
                                    if (!isfinite(lambda)) brk = true;
                                }
                                if (!brk) qmax++;
                                if (!(!brk && rho < 0 && qmax < 10)) break;
                            }
                        }
                        __syncthreads();
                    }
                    again = S.ctl[0] != 0;
                } while (again);
                if (tid == 0) {
                    done++;
                    chis[round] = currentChi;
                    const bool terminate = qmax == 10 || rho == 0 || !isfinite(lambda);
                    // the accepted trial system is the next iteration's buildSystem; keep the solver's system otherwise
                    S.ctl[1] = terminate ? 1 : 0;
                    S.ctl[3] = (accepted && !terminate && it < 9) ? 1 : 0;
                    S.ctl[5] = (S.ctl[3] && accepted_light) ? 1 : 0;          // accepted on a chi2-only evaluation: build its system now
                }
                __syncthreads();
                if (S.ctl[5]) pnp_evaluate(S, s_m, n, S.T, A.K, robust, delta, 1, true);
                if (S.ctl[3] && tid < 43) S.sys[0][tid] = S.sys[1][tid];
                const bool stop = S.ctl[1] != 0;
                __syncthreads();
                if (stop) break;
            }
        } else {                                                               // Gauss-Newton, optimization_algorithm_gauss_newton.cpp:47-94
            if (tid == 0) for (int j = 0; j < 6; j++) x[j] = 0.0;
            for (int it = 0; it < 10; it++) {
                pnp_evaluate(S, s_m, n, S.T, A.K, robust, delta, 0);
                if (tid == 0) {
                    chis[round] = S.sys[0][42];
                    const bool ok = pnp_llt_solve(S.sys[0], 0.0, S.sys[0] + 36, x);
                    PnpPose E, Tn;
                    pq_exp(x, E); pq_mul(E, S.T, Tn);
                    S.T = Tn;
                    done++;
                    S.ctl[1] = ok ? 0 : 1;
                }
                __syncthreads();
                const bool stop = S.ctl[1] != 0;
                __syncthreads();
                if (stop) break;
            }
        }
        // evaluateOutliers, :384-427
        if (tid == 0) S.ctl[2] = 0;
        __syncthreads();
        int bad = 0;
        for (int i = tid; i < n; i += PNP_THREADS) {
            unsigned char flag = 0;
            if (A.check) {
                double e[2], p[3], om;
                pnp_edge(s_m, i, S.T, A.K, e, p, om);
                const double info = A.m[i].info;
                const float chi2 = (float)(e[0] * (info * e[0]) + e[1] * (info * e[1]));
                flag = (!isfinite(chi2) || (double)chi2 > 5.991) ? 1 : 0;
            }
            S.lvl[i] = flag;
            bad += flag;
        }
        if (bad) atomicAdd(&S.ctl[2], bad);
        __syncthreads();
        const int nb = S.ctl[2];
        if (tid == 0) { lmits[round] = done; nBad = nb; rounds = round + 1; }
        if ((n - nb) < 5) { failed = true; break; }
        if (round == 2) robust = false;
        if (n < 10) { failed = true; break; }                                  // optimizer.edges().size() < 10
        __syncthreads();
    }
    __syncthreads();
    for (int i = tid; i < n; i += PNP_THREADS) A.outliers[i] = S.lvl[i];
    if (tid == 0) {
        cmlhip_pnp_result& R = *A.out;
        bool ok = !failed;
        for (int k = 0; k < 6; k++) R.covariance[k] = 0.0;
        if (ok && A.cov) {                                                     // :177-190: diag(Hpp^-1), Hpp as the last buildSystem left it
            for (int cidx = 0; cidx < 6 && ok; cidx++) {
                double e[6] = {0, 0, 0, 0, 0, 0}, y[6] = {0, 0, 0, 0, 0, 0};
                e[cidx] = 1.0;
                ok = pnp_llt_solve(S.sys[0], 0.0, e, y);
                R.covariance[cidx] = y[cidx];
            }
        }
        R.is_ok = ok ? 1 : 0; R.rounds = rounds; R.n_bad = nBad; R.pad = 0;
        for (int k = 0; k < 4; k++) { R.lm_iterations[k] = lmits[k]; R.chi2[k] = chis[k]; }
        pq_to_matrix(S.T, R.R);
        for (int k = 0; k < 3; k++) R.t[k] = S.T.t[k];
    }
}

extern "C" {

int cmlhip_pnp_optimize(cmlhip_ctx* c, const double R[9], const double t[3], const double K[4], int n, const cmlhip_pnp_match* matches,
                        unsigned char* outliers, int algorithm, int check_outliers, int compute_covariance, cmlhip_pnp_result* out) { CML_DEV(c);
    if (!c || !R || !t || !K || n < 0 || (n > 0 && (!matches || !outliers)) || !out) return CMLHIP_ERR_INVALID;
    if (algorithm != CMLHIP_PNP_LEVENBERG && algorithm != CMLHIP_PNP_GAUSS_NEWTON) return CMLHIP_ERR_INVALID;
    CML_REQUIRE(c, n <= PNP_MAX_MATCHES, CMLHIP_ERR_INVALID, "more matches than the LDS-resident pose optimiser holds (2560)");
    *out = cmlhip_pnp_result{};
    for (int k = 0; k < 9; k++) out->R[k] = R[k];
    for (int k = 0; k < 3; k++) out->t[k] = t[k];
    int nBad = 0;
    for (int i = 0; i < n; i++) nBad += outliers[i] ? 1 : 0;
    out->n_bad = nBad;
    // "Not enough initial correspondences" / "Too few initial inliers", :121-129 (the Gauss-Newton overload only has the first, :312-314)
    if (n < 3 || (algorithm == CMLHIP_PNP_LEVENBERG && (n - nBad) < 5)) return CMLHIP_OK;
    int rc;
    if ((rc = cml_ensure(c, c->pnp_matches, sizeof(cmlhip_pnp_match) * (size_t)n))) return rc;
    if ((rc = cml_ensure(c, c->pnp_flags, (size_t)n))) return rc;
    if ((rc = cml_ensure(c, c->pnp_out, sizeof(cmlhip_pnp_result)))) return rc;
    if ((rc = cml_h2d(c, c->pnp_matches.p, matches, sizeof(cmlhip_pnp_match) * (size_t)n))) return rc;
    if ((rc = cml_h2d(c, c->pnp_flags.p, outliers, (size_t)n))) return rc;
    PnpArgs A;
    A.m = c->pnp_matches.as<cmlhip_pnp_match>(); A.n = n; A.outliers = c->pnp_flags.as<unsigned char>();
    for (int k = 0; k < 9; k++) A.R0[k] = R[k];
    for (int k = 0; k < 3; k++) A.t0[k] = t[k];
    for (int k = 0; k < 4; k++) A.K[k] = K[k];
    A.algorithm = algorithm; A.check = check_outliers ? 1 : 0; A.cov = compute_covariance ? 1 : 0;
    A.out = c->pnp_out.as<cmlhip_pnp_result>();
    const size_t dyn = sizeof(double) * 6 * (size_t)n;
    if (!(c->attr_done & (1u << 16))) {                              // per context (= per device), not a process-wide static
        CML_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pnp_optimize), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * 6 * PNP_MAX_MATCHES)));
        c->attr_done |= 1u << 16;
    }
    k_pnp_optimize<<<1, PNP_THREADS, dyn, c->stream>>>(A);
    CML_CHECK(c, hipGetLastError());
    cml_d2h_batch_begin(c);                                             // result + flags: one round trip
    cml_d2h(c, out, c->pnp_out.p, sizeof(cmlhip_pnp_result));
    cml_d2h(c, outliers, c->pnp_flags.p, (size_t)n);
    return cml_d2h_batch_flush(c);
}

}  // extern "C"
