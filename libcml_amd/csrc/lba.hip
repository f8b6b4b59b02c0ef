// lba.hip — the ORB side's local bundle adjustment.
// Replaces CML::Optimization::G2O::IndirectBundleAdjustment::localOptimize / startOptimization and the removal test of
// apply() (src/cml/optimization/g2o/IndirectBundleAdjustment.cpp:7-236,:322-334) for the graph handed over as arrays, and the
// slice of the vendored g2o it runs: StructureOnlySolver<3>::calc (g2o/solvers/structure_only/structure_only_solver.h:66-217),
// EdgeSE3ProjectXYZ (g2o/types/sba/edge_project_xyz.cpp:44-95), RobustKernelHuber, constructQuadraticForm, Eigen::LDLT 3x3.
//
// fixFrames (mBaMode != BAINDIRECT, the mode MODSLAM runs in while the photometric BA owns the poses): every point is an
// independent 3-unknown damped Gauss-Newton problem over its own track, so the device form is a lane per point that walks its
// point-major edge range exactly as g2o walks v->edges() — same statements, same order, fp64, no fused multiply-adds — and ALL
// optimize() iterations of a pass run inside one launch (g2o re-enters calc() per iteration only to do the same thing again).
// The results are bit-identical to the oracle restatement.
#include "cmlhip_internal.h"

#pragma clang fp contract(off)

struct LbaCam { double qx, qy, qz, qw, t[3], R[9], K[4]; };

// SE3Quat(R, t): Eigen Quaternion(Matrix3) + normalizeRotation (g2o/types/slam3d/se3quat.h:52-58), then toRotationMatrix
__global__ void k_lba_cams(const cmlhip_lba_frame* __restrict__ frames, int n, LbaCam* __restrict__ cams) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const double* m = frames[f].R;
    double x, y, z, w;
    const double tr = m[0] + m[4] + m[8];
    if (tr > 0) {
        double t = sqrt(tr + 1.0);
        w = 0.5 * t; t = 0.5 / t;
        x = (m[7] - m[5]) * t; y = (m[2] - m[6]) * t; z = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        double v[3];
        v[i] = 0.5 * t; t = 0.5 / t;
        w = (m[k * 3 + j] - m[j * 3 + k]) * t;
        v[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        v[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        x = v[0]; y = v[1]; z = v[2];
    }
    if (w < 0) { x *= -1; y *= -1; z *= -1; w *= -1; }
    const double nn = sqrt(x * x + y * y + z * z + w * w);
    x /= nn; y /= nn; z /= nn; w /= nn;
    LbaCam& C = cams[f];
    C.qx = x; C.qy = y; C.qz = z; C.qw = w;
    for (int k = 0; k < 3; k++) C.t[k] = frames[f].t[k];
    for (int k = 0; k < 4; k++) C.K[k] = frames[f].K[k];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    C.R[0] = 1 - (tyy + tzz); C.R[1] = txy - twz; C.R[2] = txz + twy;
    C.R[3] = txy + twz; C.R[4] = 1 - (txx + tzz); C.R[5] = tyz - twx;
    C.R[6] = txz - twy; C.R[7] = tyz + twx; C.R[8] = 1 - (txx + tyy);
}

__device__ __forceinline__ void lba_map(const LbaCam& C, const double v[3], double p[3]) {     // SE3Quat::map: _r * xyz + _t
    double uv[3] = {C.qy * v[2] - C.qz * v[1], C.qz * v[0] - C.qx * v[2], C.qx * v[1] - C.qy * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const double r0 = v[0] + C.qw * uv[0] + (C.qy * uv[2] - C.qz * uv[1]);
    const double r1 = v[1] + C.qw * uv[1] + (C.qz * uv[0] - C.qx * uv[2]);
    const double r2 = v[2] + C.qw * uv[2] + (C.qx * uv[1] - C.qy * uv[0]);
    p[0] = r0 + C.t[0]; p[1] = r1 + C.t[1]; p[2] = r2 + C.t[2];
}
__device__ __forceinline__ void lba_error(const LbaCam& C, const double X[3], const cmlhip_lba_edge& E, double e[2], double p[3]) {
    lba_map(C, X, p);                                                      // computeError, edge_project_xyz.cpp:44-50
    e[0] = E.obs[0] - (p[0] / p[2] * C.K[0] + C.K[2]);
    e[1] = E.obs[1] - (p[1] / p[2] * C.K[1] + C.K[3]);
}
__device__ __forceinline__ double lba_chi2(const double e[2], double om) { return e[0] * (om * e[0]) + e[1] * (om * e[1]); }
__device__ __forceinline__ void lba_huber(double e, double delta, double& rho0, double& rho1) {    // robust_kernel_impl.cpp:60-74
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho0 = e; rho1 = 1.; }
    else { const double sq = sqrt(e); rho0 = 2 * sq * delta - dsqr; rho1 = delta / sq; }
}

// Eigen::LDLT<Matrix3d>: diagonal pivoting, isPositive(), solve (Eigen/src/Cholesky/LDLT.h:300-396,560-600)
__device__ static bool lba_ldlt3(const double Ain[9], const double b[3], double x[3]) {
    double A[9];
    int tr[3];
    for (int i = 0; i < 9; i++) A[i] = Ain[i];
    int sign = 0;                                                          // 0 zero, 1 positive, 2 negative, 3 indefinite
#define M(i, j) A[(i) * 3 + (j)]
    for (int k = 0; k < 3; k++) {
        int big = k;
        double best = fabs(M(k, k));
        for (int i = k + 1; i < 3; i++) if (fabs(M(i, i)) > best) { best = fabs(M(i, i)); big = i; }
        tr[k] = big;
        if (k != big) {
            const int s = 3 - big - 1;
            for (int j = 0; j < k; j++) { const double t = M(k, j); M(k, j) = M(big, j); M(big, j) = t; }
            for (int i = 0; i < s; i++) { const double t = M(big + 1 + i, k); M(big + 1 + i, k) = M(big + 1 + i, big); M(big + 1 + i, big) = t; }
            { const double t = M(k, k); M(k, k) = M(big, big); M(big, big) = t; }
            for (int i = k + 1; i < big; i++) { const double t = M(i, k); M(i, k) = M(big, i); M(big, i) = t; }
        }
        const int rs = 3 - k - 1;
        if (k > 0) {
            double temp[3];
            for (int j = 0; j < k; j++) temp[j] = M(j, j) * M(k, j);
            double s = 0;
            for (int j = 0; j < k; j++) s += M(k, j) * temp[j];
            M(k, k) -= s;
            for (int i = 0; i < rs; i++) {
                double s2 = 0;
                for (int j = 0; j < k; j++) s2 += M(k + 1 + i, j) * temp[j];
                M(k + 1 + i, k) -= s2;
            }
        }
        const double akk = M(k, k);
        const bool valid = fabs(akk) > 0.0;
        if (k == 0 && !valid) { sign = 0; for (int j = 0; j < 3; j++) tr[j] = j; break; }
        if (rs > 0 && valid) for (int i = 0; i < rs; i++) M(k + 1 + i, k) /= akk;
        if (sign == 1) { if (akk < 0) sign = 3; }
        else if (sign == 2) { if (akk > 0) sign = 3; }
        else if (sign == 0) { if (akk > 0) sign = 1; else if (akk < 0) sign = 2; }
    }
    for (int i = 0; i < 3; i++) x[i] = b[i];
    for (int k = 0; k < 3; k++) if (tr[k] != k) { const double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
    for (int i = 0; i < 3; i++) { double s = x[i]; for (int j = 0; j < i; j++) s -= M(i, j) * x[j]; x[i] = s; }
    for (int i = 0; i < 3; i++) { if (fabs(M(i, i)) > 2.2250738585072014e-308) x[i] /= M(i, i); else x[i] = 0; }
    for (int i = 2; i >= 0; i--) { double s = x[i]; for (int j = i + 1; j < 3; j++) s -= M(j, i) * x[j]; x[i] = s; }
    for (int k = 2; k >= 0; k--) if (tr[k] != k) { const double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
#undef M
    return sign == 1 || sign == 0;
}

struct LbaArgs {
    const LbaCam* cams; const cmlhip_lba_edge* edges; const int* off;
    double* points; double* err; unsigned char* level1;
    int n_points, iterations, robust;
    double delta;
};

__device__ static double lba_track_chi2(const LbaArgs& A, int e0, int n, const double X[3]) {
    double chi2 = 0;
    for (int k = 0; k < n; k++) {
        const cmlhip_lba_edge E = A.edges[e0 + k];
        double e[2], p[3];
        lba_error(A.cams[E.frame], X, E, e, p);
        A.err[2 * (size_t)(e0 + k)] = e[0]; A.err[2 * (size_t)(e0 + k) + 1] = e[1];
        const double c = lba_chi2(e, E.inv_sigma2);
        if (A.robust) { double r0, r1; lba_huber(c, A.delta, r0, r1); chi2 += r0; }
        else chi2 += c;
    }
    return chi2;
}

// optimize(iterations) of the structure-only solver: a lane per point, structure_only_solver.h:74-215 per iteration
__global__ __launch_bounds__(64) void k_lba_structure_only(LbaArgs A) {
    const int pt = blockIdx.x * 64 + threadIdx.x;
    if (pt >= A.n_points) return;
    const int e0 = A.off[pt], n = A.off[pt + 1] - e0;
    bool active = false;                                                   // activeVertices: a level-0 edge is attached
    for (int k = 0; k < n; k++) active = active || !A.level1[e0 + k];
    if (!active || n == 0) return;
    double X[3] = {A.points[3 * (size_t)pt], A.points[3 * (size_t)pt + 1], A.points[3 * (size_t)pt + 2]};
    for (int it = 0; it < A.iterations; it++) {
        double chi2 = lba_track_chi2(A, e0, n, X);
        double mu = 0.01, nu = 2;
        double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
        for (int k = 0; k < n; k++) {
            const cmlhip_lba_edge E = A.edges[e0 + k];
            const LbaCam& C = A.cams[E.frame];
            double e[2], p[3], J[2][3];
            lba_error(C, X, E, e, p);
            A.err[2 * (size_t)(e0 + k)] = e[0]; A.err[2 * (size_t)(e0 + k) + 1] = e[1];
            {                                                              // _jacobianOplusXi = -1./z * tmp * R, edge_project_xyz.cpp:68-78
                const double x = p[0], y = p[1], z = p[2], fx = C.K[0], fy = C.K[1];
                const double tmp[2][3] = {{fx, 0, -x / z * fx}, {0, fy, -y / z * fy}};
                const double s = -1. / z;
                for (int r = 0; r < 2; r++)
                    for (int c = 0; c < 3; c++)
                        J[r][c] = ((s * tmp[r][0]) * C.R[c] + (s * tmp[r][1]) * C.R[3 + c]) + (s * tmp[r][2]) * C.R[6 + c];
            }
            const double om = E.inv_sigma2;
            double rho0 = 0, rho1 = 1.;
            if (A.robust) lba_huber(lba_chi2(e, om), A.delta, rho0, rho1);
            const double w = rho1 * om;
            const double we[2] = {(-om * e[0]) * rho1, (-om * e[1]) * rho1};
            for (int j = 0; j < 3; j++) {
                b[j] += J[0][j] * we[0] + J[1][j] * we[1];
                const double a0 = J[0][j] * w, a1 = J[1][j] * w;
                for (int c = 0; c < 3; c++) H[j * 3 + c] += a0 * J[0][c] + a1 * J[1][c];
            }
        }
        if (sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]) < 0.001) continue;      // stop: this calc() is over, the next starts afresh
        int trial = 0;
        for (;;) {
            double Hmu[9], dp[3];
            for (int i = 0; i < 9; i++) Hmu[i] = H[i];
            Hmu[0] += mu; Hmu[4] += mu; Hmu[8] += mu;
            bool good = false;
            if (lba_ldlt3(Hmu, b, dp)) {
                const double Xn[3] = {X[0] + dp[0], X[1] + dp[1], X[2] + dp[2]};
                const double new_chi2 = lba_track_chi2(A, e0, n, Xn);
                const double rho = chi2 - new_chi2;
                if (rho > 0 && isfinite(new_chi2)) { good = true; chi2 = new_chi2; X[0] = Xn[0]; X[1] = Xn[1]; X[2] = Xn[2]; }
            }
            if (good) break;                                               // mu *= 1/3 is dead: mu is re-initialised by the next calc()
            mu *= nu; nu *= 2.; ++trial;
            if (trial >= 10) break;
        }
    }
    A.points[3 * (size_t)pt] = X[0]; A.points[3 * (size_t)pt + 1] = X[1]; A.points[3 * (size_t)pt + 2] = X[2];
}

// the level test of the refinement pass (:210-221) and apply()'s removal test (:327): chi2() of the stored error, isDepthPositive()
__global__ void k_lba_edge_test(const LbaCam* __restrict__ cams, const cmlhip_lba_edge* __restrict__ edges, const int* __restrict__ off,
                                const double* __restrict__ points, const double* __restrict__ err, int n_points, unsigned char* __restrict__ flag) {
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n_points) return;
    const double X[3] = {points[3 * (size_t)pt], points[3 * (size_t)pt + 1], points[3 * (size_t)pt + 2]};
    for (int k = off[pt]; k < off[pt + 1]; k++) {
        const cmlhip_lba_edge E = edges[k];
        double p[3];
        lba_map(cams[E.frame], X, p);
        const double e[2] = {err[2 * (size_t)k], err[2 * (size_t)k + 1]};
        flag[k] = (lba_chi2(e, E.inv_sigma2) > 5.991 || !(p[2] > 0.0)) ? 1 : 0;
    }
}


// ================================================================================================ free poses
// fixFrames == false: g2o's OptimizationAlgorithmLevenberg over BlockSolver_6_3 with the points marginalised
// (g2o/core/optimization_algorithm_levenberg.cpp:58-175, g2o/core/block_solver.hpp:329-479,495-587).  One Levenberg trial is
//   k_lba_dinv   lane per point      D = Hll + lambda I, D^-1, D^-1 bl
//   k_lba_schur  wave per 6x6 block  S(i1,i2) = [Hpp + lambda I] - sum_p Hpl D^-1 Hpl^T over the edges of pose i1, each paired
//                                    through a point x pose -> edge table; bs(i1) = bp - sum Hpl D^-1 bl
//                                    (fixed order: deterministic, no atomics)
//   k_lba_chol   one workgroup       dense LL^T of the reduced pose system in LDS (packed lower), forward/backward substitution
//   k_lba_update lane per point / per pose   xl = D^-1 (bl - Hpl^T xp), X + xl, exp(xp) T, the terms of computeScale
//   k_lba_eval_points / k_lba_eval_frames    errors, Huber, chi2 and the next system AT THE TRIAL STATE (if the trial is
//                                    accepted that system is the next iteration's buildSystem: same state, same edges)
//   k_lba_reduce one workgroup       chi2, scale, max |diag| in a fixed order -> 4 doubles for the host
// and the host reads 32 bytes and takes g2o's accept / reject decision.
struct LbaSys { double *Hpp, *bp, *Hll, *bl, *Hpl; };     // Hpp 36/pose, bp 6/pose, Hll 9/point, bl 3/point, Hpl 18/edge (6x3)

struct LbaLmArgs {
    const LbaCam* cams; const double* points; const cmlhip_lba_edge* edges; const int* off;
    const unsigned char* level1; const int* pose_of_frame;     // frame -> pose block or -1
    double* err; LbaSys sys;
    double* part_chi; double* part_max;                        // per workgroup of the point kernel
    int n_points, robust; double delta;
};

__device__ __forceinline__ void lba_jac_point(const LbaCam& C, const double p[3], double J[2][3]) {
    const double x = p[0], y = p[1], z = p[2], fx = C.K[0], fy = C.K[1];
    const double tmp[2][3] = {{fx, 0, -x / z * fx}, {0, fy, -y / z * fy}};
    const double s = -1. / z;
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++)
            J[r][c] = ((s * tmp[r][0]) * C.R[c] + (s * tmp[r][1]) * C.R[3 + c]) + (s * tmp[r][2]) * C.R[6 + c];
}
__device__ __forceinline__ void lba_jac_pose(const LbaCam& C, const double p[3], double J[2][6]) {      // edge_project_xyz.cpp:80-94
    const double x = p[0], y = p[1], z = p[2], z_2 = z * z, fx = C.K[0], fy = C.K[1];
    J[0][0] = x * y / z_2 * fx; J[0][1] = -(1 + (x * x / z_2)) * fx; J[0][2] = y / z * fx;
    J[0][3] = -1. / z * fx; J[0][4] = 0; J[0][5] = x / z_2 * fx;
    J[1][0] = (1 + y * y / z_2) * fy; J[1][1] = -x * y / z_2 * fy; J[1][2] = -x / z * fy;
    J[1][3] = 0; J[1][4] = -1. / z * fy; J[1][5] = y / z_2 * fy;
}

// computeActiveErrors + activeRobustChi2 + the point half of buildSystem (Hll, bl, Hpl), a lane per point
__global__ __launch_bounds__(64) void k_lba_eval_points(LbaLmArgs A) {
    const int pt = blockIdx.x * 64 + threadIdx.x;
    double chi = 0.0, mx = 0.0;
    if (pt < A.n_points) {
        const double X[3] = {A.points[3 * (size_t)pt], A.points[3 * (size_t)pt + 1], A.points[3 * (size_t)pt + 2]};
        double Hll[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
        for (int k = A.off[pt]; k < A.off[pt + 1]; k++) {
            double* Hpl = A.sys.Hpl + 18 * (size_t)k;
            if (A.level1[k]) { for (int i = 0; i < 18; i++) Hpl[i] = 0.0; continue; }
            const cmlhip_lba_edge E = A.edges[k];
            const LbaCam& C = A.cams[E.frame];
            double e[2], p[3], Jl[2][3];
            lba_error(C, X, E, e, p);
            A.err[2 * (size_t)k] = e[0]; A.err[2 * (size_t)k + 1] = e[1];
            const double om = E.inv_sigma2;
            double rho0 = lba_chi2(e, om), rho1 = 1.;
            if (A.robust) lba_huber(rho0, A.delta, rho0, rho1);
            chi += rho0;
            lba_jac_point(C, p, Jl);
            const double w = rho1 * om;
            const double we[2] = {(-om * e[0]) * rho1, (-om * e[1]) * rho1};
            for (int j = 0; j < 3; j++) {
                bl[j] += Jl[0][j] * we[0] + Jl[1][j] * we[1];
                for (int c = 0; c < 3; c++) Hll[j * 3 + c] += (Jl[0][j] * w) * Jl[0][c] + (Jl[1][j] * w) * Jl[1][c];
            }
            if (A.pose_of_frame[E.frame] >= 0) {
                double Jp[2][6];
                lba_jac_pose(C, p, Jp);
                for (int j = 0; j < 6; j++)
                    for (int c = 0; c < 3; c++) Hpl[j * 3 + c] = (Jp[0][j] * w) * Jl[0][c] + (Jp[1][j] * w) * Jl[1][c];
            } else for (int i = 0; i < 18; i++) Hpl[i] = 0.0;
        }
        for (int i = 0; i < 9; i++) A.sys.Hll[9 * (size_t)pt + i] = Hll[i];
        for (int i = 0; i < 3; i++) A.sys.bl[3 * (size_t)pt + i] = bl[i];
        mx = fmax(fmax(fabs(Hll[0]), fabs(Hll[4])), fabs(Hll[8]));
    }
    for (int o = 32; o >= 1; o >>= 1) { chi += __shfl_xor(chi, o, 64); mx = fmax(mx, __shfl_xor(mx, o, 64)); }
    if (threadIdx.x == 0) { A.part_chi[blockIdx.x] = chi; A.part_max[blockIdx.x] = mx; }
}

// the pose half of buildSystem: Hpp, bp of one free pose over the edges that observe from it (fixed list order)
__global__ __launch_bounds__(256) void k_lba_eval_frames(LbaLmArgs A, const int* __restrict__ fe_off, const int* __restrict__ fe_edge,
                                                         const int* __restrict__ edge_point, const int* __restrict__ frame_of_pose, double* __restrict__ pose_max) {
    __shared__ double s_red[4][42];
    const int pi = blockIdx.x, tid = threadIdx.x, f = frame_of_pose[pi];
    const LbaCam& C = A.cams[f];
    double acc[42];
#pragma unroll
    for (int i = 0; i < 42; i++) acc[i] = 0.0;
    for (int q = fe_off[pi] + tid; q < fe_off[pi + 1]; q += 256) {
        const int k = fe_edge[q];
        if (A.level1[k]) continue;
        const cmlhip_lba_edge E = A.edges[k];
        const int pt = edge_point[k];
        const double X[3] = {A.points[3 * (size_t)pt], A.points[3 * (size_t)pt + 1], A.points[3 * (size_t)pt + 2]};
        double e[2], p[3], Jp[2][6];
        lba_error(C, X, E, e, p);
        const double om = E.inv_sigma2;
        double rho0 = lba_chi2(e, om), rho1 = 1.;
        if (A.robust) lba_huber(rho0, A.delta, rho0, rho1);
        lba_jac_pose(C, p, Jp);
        const double w = rho1 * om;
        const double we[2] = {(-om * e[0]) * rho1, (-om * e[1]) * rho1};
#pragma unroll
        for (int j = 0; j < 6; j++) {
            acc[36 + j] += Jp[0][j] * we[0] + Jp[1][j] * we[1];
#pragma unroll
            for (int c = 0; c < 6; c++) acc[j * 6 + c] += (Jp[0][j] * w) * Jp[0][c] + (Jp[1][j] * w) * Jp[1][c];
        }
    }
#pragma unroll
    for (int i = 0; i < 42; i++) {
        double v = acc[i];
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((tid & 63) == 0) s_red[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid < 42) {
        const double v = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
        if (tid < 36) A.sys.Hpp[36 * (size_t)pi + tid] = v; else A.sys.bp[6 * (size_t)pi + tid - 36] = v;
        s_red[0][tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        double mx = 0.0;
        for (int j = 0; j < 6; j++) mx = fmax(mx, fabs(s_red[0][j * 7]));
        pose_max[pi] = mx;
    }
}

__global__ void k_lba_dinv(const double* __restrict__ Hll, const double* __restrict__ bl, int n_points, double lambda, double* __restrict__ Dinv, double* __restrict__ db) {
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n_points) return;
    double A[9];
    for (int i = 0; i < 9; i++) A[i] = Hll[9 * (size_t)pt + i];
    A[0] += lambda; A[4] += lambda; A[8] += lambda;
    const double c00 = A[4] * A[8] - A[5] * A[7], c10 = A[5] * A[6] - A[3] * A[8], c20 = A[3] * A[7] - A[4] * A[6];
    const double det = c00 * A[0] + c10 * A[1] + c20 * A[2];
    const double id = 1.0 / det;
    double o[9];
    o[0] = c00 * id; o[3] = c10 * id; o[6] = c20 * id;
    o[1] = (A[2] * A[7] - A[1] * A[8]) * id; o[4] = (A[0] * A[8] - A[2] * A[6]) * id; o[7] = (A[1] * A[6] - A[0] * A[7]) * id;
    o[2] = (A[1] * A[5] - A[2] * A[4]) * id; o[5] = (A[2] * A[3] - A[0] * A[5]) * id; o[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    for (int i = 0; i < 9; i++) Dinv[9 * (size_t)pt + i] = o[i];
    const double* b = bl + 3 * (size_t)pt;
    for (int r = 0; r < 3; r++) db[3 * (size_t)pt + r] = o[r * 3] * b[0] + o[r * 3 + 1] * b[1] + o[r * 3 + 2] * b[2];
}

// one wave per upper block (i1 <= i2) of the reduced system, block_solver.hpp:357-432: the edges of pose i1 in list order, each
// paired with the edge of the same point into pose i2 (edge_of: n_points x nfree table, -1 = not observed)
__global__ __launch_bounds__(64) void k_lba_schur(LbaSys sys, const double* __restrict__ Dinv, const double* __restrict__ db,
                                                  const int* __restrict__ blk_i1, const int* __restrict__ blk_i2, const int* __restrict__ fe_off,
                                                  const int* __restrict__ fe_edge, const int* __restrict__ edge_point, const int* __restrict__ edge_of,
                                                  int nfree, double lambda, int n, double* __restrict__ S, double* __restrict__ bs) {
    const int b = blockIdx.x, l = threadIdx.x, i1 = blk_i1[b], i2 = blk_i2[b];
    double acc[42];
#pragma unroll
    for (int i = 0; i < 42; i++) acc[i] = 0.0;
    for (int q = fe_off[i1] + l; q < fe_off[i1 + 1]; q += 64) {
        const int k1 = fe_edge[q], pt = edge_point[k1];
        const int k2 = i1 == i2 ? k1 : edge_of[(size_t)pt * nfree + i2];
        if (k2 < 0) continue;
        const double* Bi = sys.Hpl + 18 * (size_t)k1; const double* Bj = sys.Hpl + 18 * (size_t)k2;
        const double* Di = Dinv + 9 * (size_t)pt;
        double BD[18];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) BD[r * 3 + c] = Bi[r * 3] * Di[c] + Bi[r * 3 + 1] * Di[3 + c] + Bi[r * 3 + 2] * Di[6 + c];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 6; c++) acc[r * 6 + c] += BD[r * 3] * Bj[c * 3] + BD[r * 3 + 1] * Bj[c * 3 + 1] + BD[r * 3 + 2] * Bj[c * 3 + 2];
        if (i1 == i2) {
            const double* d = db + 3 * (size_t)pt;
#pragma unroll
            for (int r = 0; r < 6; r++) acc[36 + r] += Bi[r * 3] * d[0] + Bi[r * 3 + 1] * d[1] + Bi[r * 3 + 2] * d[2];
        }
    }
#pragma unroll
    for (int i = 0; i < 42; i++) for (int o = 32; o >= 1; o >>= 1) acc[i] += __shfl_xor(acc[i], o, 64);
    if (l < 36) {
        const int r = l / 6, c = l % 6;
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < 36; i++) if (i == l) v = acc[i];
        double h = 0.0;
        if (i1 == i2) { h = sys.Hpp[36 * (size_t)i1 + l]; if (r == c) h += lambda; }
        const double out = h - v;
        S[(size_t)(6 * i1 + r) * n + 6 * i2 + c] = out;
        if (i1 != i2) S[(size_t)(6 * i2 + c) * n + 6 * i1 + r] = out;
    } else if (l < 42 && i1 == i2) {
        double v = 0.0;
#pragma unroll
        for (int i = 36; i < 42; i++) if (i == l) v = acc[i];
        bs[6 * i1 + l - 36] = sys.bp[6 * (size_t)i1 + l - 36] - v;
    }
}

// Factorisation of S (n x n, symmetric positive definite, row-major in global) in LDS (packed lower) and the solve S xp = bs;
// flag[0] = 1 on success.  Same pivots as the LL^T of SimplicialLLT (it fails on the same non-positive pivot), carried in the
// square-root-free form S = L D L^T: column j is left unscaled while it updates the trailing matrix (a_ik -= a_ij a_kj / d_j),
// so a column costs ONE barrier, and both substitutions run on the unit-lower factor with one barrier per column as well.
#define LBA_CHOL_THREADS 512
__global__ __launch_bounds__(LBA_CHOL_THREADS) void k_lba_chol(const double* __restrict__ S, const double* __restrict__ bs, int n, double* __restrict__ xp, int* __restrict__ flag) {
    extern __shared__ double s_L[];                 // n (n + 1) / 2 packed lower, then n of the right-hand side, then n inverse pivots
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, NW = LBA_CHOL_THREADS / 64;
    double* rhs = s_L + (size_t)n * (n + 1) / 2;
    double* idg = rhs + n;
#define LI(i, j) s_L[(size_t)(i) * ((i) + 1) / 2 + (j)]
    for (int q = tid; q < n * n; q += LBA_CHOL_THREADS) { const int i = q / n, j = q % n; if (j <= i) LI(i, j) = S[q]; }
    for (int i = tid; i < n; i += LBA_CHOL_THREADS) rhs[i] = bs[i];
    __syncthreads();
    bool fail = false;
    for (int j = 0; j < n; j++) {
        const double d = LI(j, j);                                // final: every update of column j happened before the last barrier
        if (!(d > 0)) { fail = true; break; }                     // uniform: every thread reads the same LDS word
        const double id = 1.0 / d;
        for (int i = j + 1 + wv; i < n; i += NW) {                // a wave per row of the trailing matrix, lanes along the row
            const double aij = LI(i, j) * id;
            for (int k = j + 1 + l; k <= i; k += 64) LI(i, k) -= aij * LI(k, j);
        }
        __syncthreads();
    }
    if (!fail) {
        for (int j = 0; j < n; j++) {                             // L y = b with L = unit lower: L(i,j) = a_ij / d_j
            const double yj = rhs[j] / LI(j, j);                  // rhs[j] is final here; every thread forms the same value
            for (int i = j + 1 + tid; i < n; i += LBA_CHOL_THREADS) rhs[i] -= LI(i, j) * yj;
            __syncthreads();
        }
        for (int i = tid; i < n; i += LBA_CHOL_THREADS) { const double di = 1.0 / LI(i, i); idg[i] = di; rhs[i] = rhs[i] * di; }   // z = D^-1 y
        __syncthreads();
        for (int j = n - 1; j >= 0; j--) {                        // L^T x = z, L(j,i) = a_ji / d_i
            const double xj = rhs[j];
            for (int i = tid; i < j; i += LBA_CHOL_THREADS) rhs[i] -= (LI(j, i) * idg[i]) * xj;
            __syncthreads();
        }
        for (int i = tid; i < n; i += LBA_CHOL_THREADS) xp[i] = rhs[i];
    }
#undef LI
    if (tid == 0) flag[0] = fail ? 0 : 1;
}

// SE3Quat::exp(update) * estimate (vertex_se3_expmap.cpp:48-51, se3quat.h:96-102,201-229) for the free poses; copies the rest
__global__ void k_lba_update_poses(const LbaCam* __restrict__ cur, LbaCam* __restrict__ trial, int n_frames, const int* __restrict__ pose_of_frame,
                                   const double* __restrict__ xp, const double* __restrict__ bp, double lambda, double* __restrict__ pose_scale) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    LbaCam C = cur[f];
    const int pi = pose_of_frame[f];
    if (pi >= 0) {
        const double* u = xp + 6 * (size_t)pi;
        double sc = 0.0;
        for (int j = 0; j < 6; j++) sc += u[j] * (lambda * u[j] + bp[6 * (size_t)pi + j]);
        pose_scale[pi] = sc;
        const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
        const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
        const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
        double O2[9], R[9], V[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
        double a, b, c, d;
        if (theta < 0.00001) { a = 1; b = 0.5; c = 0.5; d = 1. / 6.; }
        else { a = sin(theta) / theta; b = (1 - cos(theta)) / (theta * theta); c = b; d = (theta - sin(theta)) / (theta * theta * theta); }
        for (int i = 0; i < 9; i++) { const double I = (i % 4 == 0) ? 1.0 : 0.0; R[i] = I + a * O[i] + b * O2[i]; V[i] = I + c * O[i] + d * O2[i]; }
        // Quaternion(R), normalizeRotation
        double ex, ey, ez, ew;
        {
            const double tr = R[0] + R[4] + R[8];
            if (tr > 0) { double t = sqrt(tr + 1.0); ew = 0.5 * t; t = 0.5 / t; ex = (R[7] - R[5]) * t; ey = (R[2] - R[6]) * t; ez = (R[3] - R[1]) * t; }
            else {
                int i = 0;
                if (R[4] > R[0]) i = 1;
                if (R[8] > R[i * 3 + i]) i = 2;
                const int j = (i + 1) % 3, k = (j + 1) % 3;
                double t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0), v[3];
                v[i] = 0.5 * t; t = 0.5 / t;
                ew = (R[k * 3 + j] - R[j * 3 + k]) * t; v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t; v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
                ex = v[0]; ey = v[1]; ez = v[2];
            }
            if (ew < 0) { ex *= -1; ey *= -1; ez *= -1; ew *= -1; }
            const double nn = sqrt(ex * ex + ey * ey + ez * ez + ew * ew);
            ex /= nn; ey /= nn; ez /= nn; ew /= nn;
        }
        double et[3];
        for (int i = 0; i < 3; i++) et[i] = V[i * 3] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
        // result = E * C: t = E.t + E.r * C.t ; r = E.r * C.r ; normalize
        LbaCam E; E.qx = ex; E.qy = ey; E.qz = ez; E.qw = ew; E.t[0] = 0; E.t[1] = 0; E.t[2] = 0;
        double rt[3];
        lba_map(E, C.t, rt);
        double w = ew * C.qw - ex * C.qx - ey * C.qy - ez * C.qz;
        double x = ew * C.qx + ex * C.qw + ey * C.qz - ez * C.qy;
        double y = ew * C.qy + ey * C.qw + ez * C.qx - ex * C.qz;
        double z = ew * C.qz + ez * C.qw + ex * C.qy - ey * C.qx;
        if (w < 0) { x *= -1; y *= -1; z *= -1; w *= -1; }
        const double nn = sqrt(x * x + y * y + z * z + w * w);
        x /= nn; y /= nn; z /= nn; w /= nn;
        C.qx = x; C.qy = y; C.qz = z; C.qw = w;
        for (int i = 0; i < 3; i++) C.t[i] = et[i] + rt[i];
        const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
        const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
        C.R[0] = 1 - (tyy + tzz); C.R[1] = txy - twz; C.R[2] = txz + twy;
        C.R[3] = txy + twz; C.R[4] = 1 - (txx + tzz); C.R[5] = tyz - twx;
        C.R[6] = txz - twy; C.R[7] = tyz + twx; C.R[8] = 1 - (txx + tyy);
    }
    trial[f] = C;
}

// xl = D^-1 (bl - Hpl^T xp) (block_solver.hpp:453-474), X_trial = X + xl, the point terms of computeScale
__global__ __launch_bounds__(64) void k_lba_update_points(LbaSys sys, const double* __restrict__ Dinv, const int* __restrict__ off, const cmlhip_lba_edge* __restrict__ edges,
                                                          const int* __restrict__ pose_of_frame, const double* __restrict__ xp, const double* __restrict__ cur,
                                                          double* __restrict__ trial, int n_points, double lambda, double* __restrict__ part_scale) {
    const int pt = blockIdx.x * 64 + threadIdx.x;
    double sc = 0.0;
    if (pt < n_points) {
        double cl[3] = {sys.bl[3 * (size_t)pt], sys.bl[3 * (size_t)pt + 1], sys.bl[3 * (size_t)pt + 2]};
        for (int k = off[pt]; k < off[pt + 1]; k++) {
            const int pi = pose_of_frame[edges[k].frame];
            if (pi < 0) continue;
            const double* B = sys.Hpl + 18 * (size_t)k;
            for (int c = 0; c < 3; c++)
                for (int r = 0; r < 6; r++) cl[c] += B[r * 3 + c] * (-xp[6 * (size_t)pi + r]);
        }
        const double* Di = Dinv + 9 * (size_t)pt;
        for (int r = 0; r < 3; r++) {
            const double x = Di[r * 3] * cl[0] + Di[r * 3 + 1] * cl[1] + Di[r * 3 + 2] * cl[2];
            trial[3 * (size_t)pt + r] = cur[3 * (size_t)pt + r] + x;
            sc += x * (lambda * x + sys.bl[3 * (size_t)pt + r]);
        }
    }
    for (int o = 32; o >= 1; o >>= 1) sc += __shfl_xor(sc, o, 64);
    if (threadIdx.x == 0) part_scale[blockIdx.x] = sc;
}

// fixed-order sums of the partials: out = {chi2, scale, max |diag|}
__global__ __launch_bounds__(64) void k_lba_reduce(const double* part_chi, const double* part_max, const double* part_scale, int nb,
                                                   const double* pose_scale, const double* pose_max, int nfree, const int* flag, double* out) {
    if (threadIdx.x != 0) return;
    double chi = 0.0, sc = 0.0, mx = 0.0;
    for (int i = 0; i < nfree; i++) { sc += pose_scale[i]; mx = fmax(mx, pose_max[i]); }
    for (int i = 0; i < nb; i++) { chi += part_chi[i]; sc += part_scale[i]; mx = fmax(mx, part_max[i]); }
    out[0] = chi; out[1] = sc; out[2] = mx; out[3] = flag ? (double)flag[0] : 1.0;      // [3]: the factorisation succeeded
}

// ---- host driver of the Levenberg mode: graph index lists, buffers, g2o's accept / reject loop
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

namespace {
struct Carver {
    char* base; size_t at = 0;
    template <class T> T* take(size_t n) { at = (at + 255) & ~(size_t)255; T* p = base ? reinterpret_cast<T*>(base + at) : nullptr; at += sizeof(T) * n; return p; }
};
}

static int lba_levenberg(cmlhip_ctx* c, int n_frames, cmlhip_lba_frame* frames, int n_points, double* points, const int* off,
                         const cmlhip_lba_edge* edges, int n_edges, int num_iterations, int refine_iterations, unsigned char* edge_bad,
                         cmlhip_lba_result* out) {
    // ---- index lists (the reference builds the g2o graph here, IndirectBundleAdjustment.cpp:69-165)
    std::vector<int> pose_of_frame(n_frames, -1), frame_of_pose;
    for (int f = 0; f < n_frames; f++) if (!frames[f].fixed) { pose_of_frame[f] = (int)frame_of_pose.size(); frame_of_pose.push_back(f); }
    const int nfree = (int)frame_of_pose.size(), n = 6 * nfree;
    CML_REQUIRE(c, nfree >= 1 && nfree <= 32, CMLHIP_ERR_INVALID, "local BA: 1..32 free keyframes (the reduced pose system is factorised in the LDS of one CU)");
    std::vector<int> edge_point(n_edges), fe_off(nfree + 1, 0), fe_edge;
    for (int p = 0; p < n_points; p++) for (int k = off[p]; k < off[p + 1]; k++) edge_point[k] = p;
    for (int k = 0; k < n_edges; k++) { const int pi = pose_of_frame[edges[k].frame]; if (pi >= 0) fe_off[pi + 1]++; }
    for (int i = 0; i < nfree; i++) fe_off[i + 1] += fe_off[i];
    fe_edge.resize(std::max(fe_off[nfree], 1));
    { std::vector<int> at(fe_off.begin(), fe_off.end() - 1);
      for (int k = 0; k < n_edges; k++) { const int pi = pose_of_frame[edges[k].frame]; if (pi >= 0) fe_edge[at[pi]++] = k; } }
    // every upper block (i1 <= i2) of the reduced system, and the point x pose -> edge table the Schur kernel pairs edges with
    std::vector<int> blk_i1, blk_i2, edge_of((size_t)std::max(n_points, 1) * nfree, -1);
    for (int a = 0; a < nfree; a++) for (int b = a; b < nfree; b++) { blk_i1.push_back(a); blk_i2.push_back(b); }
    for (int k = 0; k < n_edges; k++) { const int pi = pose_of_frame[edges[k].frame]; if (pi >= 0) edge_of[(size_t)edge_point[k] * nfree + pi] = k; }
    const int nblk = (int)blk_i1.size();
    // ---- device memory
    const int nb = cml_div_up(n_points, 64);
    // (one work buffer: sized by a dry run of the carving with a null base, then carved for real)
    auto carve = [&](Carver& Q, LbaSys sys[2], LbaCam* cams[2], double* pts[2], double*& Dinv, double*& db, double*& S, double*& bs, double*& xp,
                     double*& part_chi, double*& part_max, double*& part_scale, double*& pose_scale, double*& pose_max, double*& out4, int*& flag,
                     int*& d_pof, int*& d_fop, int*& d_ep, int*& d_feoff, int*& d_feedge, int*& d_b1, int*& d_b2, int*& d_eof) {
        for (int i = 0; i < 2; i++) {
            sys[i].Hpp = Q.take<double>(36 * (size_t)nfree); sys[i].bp = Q.take<double>(6 * (size_t)nfree);
            sys[i].Hll = Q.take<double>(9 * (size_t)n_points); sys[i].bl = Q.take<double>(3 * (size_t)n_points);
            sys[i].Hpl = Q.take<double>(18 * (size_t)n_edges);
            cams[i] = Q.take<LbaCam>(n_frames); pts[i] = Q.take<double>(3 * (size_t)n_points);
        }
        Dinv = Q.take<double>(9 * (size_t)n_points); db = Q.take<double>(3 * (size_t)n_points);
        S = Q.take<double>((size_t)n * n); bs = Q.take<double>(n); xp = Q.take<double>(n);
        part_chi = Q.take<double>(nb); part_max = Q.take<double>(nb); part_scale = Q.take<double>(nb);
        pose_scale = Q.take<double>(nfree); pose_max = Q.take<double>(nfree); out4 = Q.take<double>(4); flag = Q.take<int>(4);
        d_pof = Q.take<int>(n_frames); d_fop = Q.take<int>(nfree); d_ep = Q.take<int>(n_edges);
        d_feoff = Q.take<int>(nfree + 1); d_feedge = Q.take<int>(fe_edge.size());
        d_b1 = Q.take<int>(nblk); d_b2 = Q.take<int>(nblk); d_eof = Q.take<int>(edge_of.size());
    };
    LbaSys sys[2]; LbaCam* cams[2]; double* pts[2];
    double *Dinv, *db, *S, *bs, *xp, *part_chi, *part_max, *part_scale, *pose_scale, *pose_max, *out4;
    int *flag, *d_pof, *d_fop, *d_ep, *d_feoff, *d_feedge, *d_b1, *d_b2, *d_eof;
    Carver dry{nullptr};
    carve(dry, sys, cams, pts, Dinv, db, S, bs, xp, part_chi, part_max, part_scale, pose_scale, pose_max, out4, flag, d_pof, d_fop, d_ep, d_feoff, d_feedge, d_b1, d_b2, d_eof);
    int rc;
    if ((rc = cml_ensure(c, c->lba_work, dry.at + 512))) return rc;
    Carver real{c->lba_work.as<char>()};
    carve(real, sys, cams, pts, Dinv, db, S, bs, xp, part_chi, part_max, part_scale, pose_scale, pose_max, out4, flag, d_pof, d_fop, d_ep, d_feoff, d_feedge, d_b1, d_b2, d_eof);
#define LBA_UP(dst, vec) if ((rc = cml_h2d(c, dst, (vec).data(), sizeof((vec)[0]) * (vec).size()))) return rc
    LBA_UP(d_pof, pose_of_frame); LBA_UP(d_fop, frame_of_pose); LBA_UP(d_ep, edge_point); LBA_UP(d_feoff, fe_off); LBA_UP(d_feedge, fe_edge);
    LBA_UP(d_b1, blk_i1); LBA_UP(d_b2, blk_i2); LBA_UP(d_eof, edge_of);
#undef LBA_UP
    unsigned char* level1 = c->lba_flags.as<unsigned char>();
    unsigned char* bad = level1 + n_edges;
    CML_CHECK(c, hipMemcpyAsync(cams[0], c->lba_cams.p, sizeof(LbaCam) * (size_t)n_frames, hipMemcpyDeviceToDevice, c->stream));
    CML_CHECK(c, hipMemcpyAsync(pts[0], c->lba_points.p, sizeof(double) * 3 * (size_t)n_points, hipMemcpyDeviceToDevice, c->stream));
    CML_CHECK(c, hipMemsetAsync(xp, 0, sizeof(double) * (size_t)n, c->stream));
    CML_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_lba_chol), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * ((size_t)192 * 193 / 2 + 2 * 192))));
    const size_t chol_lds = sizeof(double) * ((size_t)n * (n + 1) / 2 + 2 * (size_t)n);
    const double delta = (double)sqrtf(5.991f);
    int cur = 0;                                                            // state (cams, points) and its system live in slot `cur`
    auto evaluate = [&](int slot, bool robust) {
        LbaLmArgs A;
        A.cams = cams[slot]; A.points = pts[slot]; A.edges = c->lba_edges.as<cmlhip_lba_edge>(); A.off = c->lba_off.as<int>();
        A.level1 = level1; A.pose_of_frame = d_pof; A.err = c->lba_err.as<double>(); A.sys = sys[slot];
        A.part_chi = part_chi; A.part_max = part_max; A.n_points = n_points; A.robust = robust ? 1 : 0; A.delta = delta;
        k_lba_eval_points<<<nb, 64, 0, c->stream>>>(A);
        k_lba_eval_frames<<<nfree, 256, 0, c->stream>>>(A, d_feoff, d_feedge, d_ep, d_fop, pose_max);
    };
    double h4[4];
    for (int phase = 0; phase < 2; phase++) {
        const int iters = phase == 0 ? num_iterations : refine_iterations;
        const bool robust = phase == 0;
        if (phase == 1) {
            if (refine_iterations <= 0) break;
            k_lba_edge_test<<<nb, 64, 0, c->stream>>>(cams[cur], c->lba_edges.as<cmlhip_lba_edge>(), c->lba_off.as<int>(), pts[cur], c->lba_err.as<double>(), n_points, level1);
        }
        if (iters <= 0) continue;
        CML_CHECK(c, hipMemsetAsync(part_scale, 0, sizeof(double) * (size_t)nb, c->stream));
        CML_CHECK(c, hipMemsetAsync(pose_scale, 0, sizeof(double) * (size_t)nfree, c->stream));
        evaluate(cur, robust);
        k_lba_reduce<<<1, 64, 0, c->stream>>>(part_chi, part_max, part_scale, nb, pose_scale, pose_max, nfree, nullptr, out4);
        if ((rc = cml_d2h(c, h4, out4, sizeof(double) * 4))) return rc;
        double currentChi = h4[0], maxdiag = h4[2], lambda = 0.0, ni = 2.0;
        int done = 0;
        bool ok = true;
        for (int it = 0; it < iters && ok && !(c->lba_stop && *c->lba_stop); it++) {      // sparse_optimizer.cpp: `i < iterations && !terminate() && ok`; optimization_algorithm_levenberg.cpp:58-160
            if (it == 0) { lambda = 1e-5 * maxdiag; ni = 2.0; }
            double rho = 0.0;
            int qmax = 0;
            do {
                const int tr = cur ^ 1;
                k_lba_dinv<<<cml_div_up(n_points, 64), 64, 0, c->stream>>>(sys[cur].Hll, sys[cur].bl, n_points, lambda, Dinv, db);
                k_lba_schur<<<nblk, 64, 0, c->stream>>>(sys[cur], Dinv, db, d_b1, d_b2, d_feoff, d_feedge, d_ep, d_eof, nfree, lambda, n, S, bs);
                k_lba_chol<<<1, LBA_CHOL_THREADS, chol_lds, c->stream>>>(S, bs, n, xp, flag);
                // (after a failed factorisation g2o applies the stale x of the previous solve and pops it again; here xp is
                //  stale too, the point part is recomputed from it: the state is restored either way)
                k_lba_update_poses<<<cml_div_up(n_frames, 64), 64, 0, c->stream>>>(cams[cur], cams[tr], n_frames, d_pof, xp, sys[cur].bp, lambda, pose_scale);
                k_lba_update_points<<<nb, 64, 0, c->stream>>>(sys[cur], Dinv, c->lba_off.as<int>(), c->lba_edges.as<cmlhip_lba_edge>(), d_pof, xp, pts[cur], pts[tr], n_points, lambda, part_scale);
                evaluate(tr, robust);
                k_lba_reduce<<<1, 64, 0, c->stream>>>(part_chi, part_max, part_scale, nb, pose_scale, pose_max, nfree, flag, out4);
                CML_CHECK(c, hipGetLastError());
                if ((rc = cml_d2h(c, h4, out4, sizeof(double) * 4))) return rc;
                double tempChi = h4[0];
                if (h4[3] == 0.0) tempChi = DBL_MAX;
                rho = currentChi - tempChi;
                const double scale = h4[1] + 1e-3;
                rho /= scale;
                if (rho > 0 && std::isfinite(tempChi)) {
                    double alpha = 1. - std::pow(2 * rho - 1, 3);
                    alpha = std::min(alpha, 2. / 3.);
                    lambda *= std::max(1. / 3., alpha); ni = 2.0; currentChi = tempChi;
                    cur = tr;                                               // discardTop: the trial state and its system become current
                } else {
                    lambda *= ni; ni *= 2.0;                                // pop: the current slot was never touched
                    if (!std::isfinite(lambda)) break;
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            done++;
            out->chi2[phase] = currentChi;
            if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) ok = false;
        }
        out->iterations_done[phase] = done;
    }
    k_lba_edge_test<<<nb, 64, 0, c->stream>>>(cams[cur], c->lba_edges.as<cmlhip_lba_edge>(), c->lba_off.as<int>(), pts[cur], c->lba_err.as<double>(), n_points, bad);
    CML_CHECK(c, hipGetLastError());
    std::vector<LbaCam> hc(n_frames);
    cml_d2h_batch_begin(c);                                             // points, edge flags, cameras: one round trip
    cml_d2h(c, points, pts[cur], sizeof(double) * 3 * (size_t)n_points);
    cml_d2h(c, edge_bad, bad, (size_t)n_edges);
    cml_d2h(c, hc.data(), cams[cur], sizeof(LbaCam) * (size_t)n_frames);
    if ((rc = cml_d2h_batch_flush(c))) return rc;
    for (int f = 0; f < n_frames; f++) {                                    // apply(): pKF->setCamera of the local keyframes, :301-305
        if (frames[f].fixed) continue;
        for (int k = 0; k < 9; k++) frames[f].R[k] = hc[f].R[k];
        for (int k = 0; k < 3; k++) frames[f].t[k] = hc[f].t[k];
    }
    int nbad = 0;
    for (int e = 0; e < n_edges; e++) nbad += edge_bad[e];
    out->n_bad = nbad; out->ok = 1;
    return CMLHIP_OK;
}

extern "C" {

int cmlhip_lba_set_stop_flag(cmlhip_ctx* c, const unsigned char* flag) {
    if (!c) return CMLHIP_ERR_INVALID;
    c->lba_stop = flag;
    return CMLHIP_OK;
}

int cmlhip_lba_optimize(cmlhip_ctx* c, int n_frames, cmlhip_lba_frame* frames, int n_points, double* points, const int* point_offsets,
                        const cmlhip_lba_edge* edges, int fix_frames, int num_iterations, int refine_iterations, unsigned char* edge_bad,
                        cmlhip_lba_result* out) { CML_DEV(c);
    if (!c || !out || n_frames < 1 || !frames || n_points < 0 || !point_offsets || num_iterations < 0) return CMLHIP_ERR_INVALID;
    *out = cmlhip_lba_result{};
    const int n_edges = point_offsets[n_points];
    if (n_points > 0 && (!points || (n_edges > 0 && (!edges || !edge_bad)))) return CMLHIP_ERR_INVALID;
    for (int p = 0; p < n_points; p++) if (point_offsets[p + 1] < point_offsets[p]) { c->err = "point_offsets must be non-decreasing"; return CMLHIP_ERR_INVALID; }
    for (int e = 0; e < n_edges; e++) if (edges[e].frame < 0 || edges[e].frame >= n_frames) { c->err = "edge frame index out of range"; return CMLHIP_ERR_INVALID; }
    if (n_points == 0 || n_edges == 0) { out->ok = 1; return CMLHIP_OK; }
    int rc;
    if ((rc = cml_ensure(c, c->lba_frames, sizeof(cmlhip_lba_frame) * (size_t)n_frames))) return rc;
    if ((rc = cml_ensure(c, c->lba_cams, sizeof(LbaCam) * (size_t)n_frames))) return rc;
    if ((rc = cml_ensure(c, c->lba_points, sizeof(double) * 3 * (size_t)n_points))) return rc;
    if ((rc = cml_ensure(c, c->lba_off, sizeof(int) * ((size_t)n_points + 1)))) return rc;
    if ((rc = cml_ensure(c, c->lba_edges, sizeof(cmlhip_lba_edge) * (size_t)n_edges))) return rc;
    if ((rc = cml_ensure(c, c->lba_err, sizeof(double) * 2 * (size_t)n_edges))) return rc;
    if ((rc = cml_ensure(c, c->lba_flags, 2 * (size_t)n_edges))) return rc;
    if ((rc = cml_h2d(c, c->lba_frames.p, frames, sizeof(cmlhip_lba_frame) * (size_t)n_frames))) return rc;
    if ((rc = cml_h2d(c, c->lba_points.p, points, sizeof(double) * 3 * (size_t)n_points))) return rc;
    if ((rc = cml_h2d(c, c->lba_off.p, point_offsets, sizeof(int) * ((size_t)n_points + 1)))) return rc;
    if ((rc = cml_h2d(c, c->lba_edges.p, edges, sizeof(cmlhip_lba_edge) * (size_t)n_edges))) return rc;
    CML_CHECK(c, hipMemsetAsync(c->lba_err.p, 0, sizeof(double) * 2 * (size_t)n_edges, c->stream));
    CML_CHECK(c, hipMemsetAsync(c->lba_flags.p, 0, 2 * (size_t)n_edges, c->stream));
    unsigned char* level1 = c->lba_flags.as<unsigned char>();
    unsigned char* bad = level1 + n_edges;
    k_lba_cams<<<cml_div_up(n_frames, 64), 64, 0, c->stream>>>(c->lba_frames.as<cmlhip_lba_frame>(), n_frames, c->lba_cams.as<LbaCam>());
    if (!fix_frames) return lba_levenberg(c, n_frames, frames, n_points, points, point_offsets, edges, n_edges, num_iterations, refine_iterations, edge_bad, out);
    LbaArgs A;
    A.cams = c->lba_cams.as<LbaCam>(); A.edges = c->lba_edges.as<cmlhip_lba_edge>(); A.off = c->lba_off.as<int>();
    A.points = c->lba_points.as<double>(); A.err = c->lba_err.as<double>(); A.level1 = level1;
    A.n_points = n_points; A.delta = (double)sqrtf(5.991f);             // const float thHuberIndirect = sqrt(5.991), :111
    const int nb = cml_div_up(n_points, 64);
    A.iterations = num_iterations; A.robust = 1;                        // startOptimization(mNumIteration, true, false), :193
    // forceStopFlag (setForceStopFlag(pbStopFlag), IBA.cpp:65-67): g2o tests it before every iteration; a pass of the structure-only
    // solver is ONE launch here, so the flag is tested before each pass
    const bool stop0 = c->lba_stop && *c->lba_stop;
    if (num_iterations > 0 && !stop0) k_lba_structure_only<<<nb, 64, 0, c->stream>>>(A);
    out->iterations_done[0] = stop0 ? 0 : num_iterations;
    if (refine_iterations > 0 && !(c->lba_stop && *c->lba_stop)) {                                        // startOptimization(mRefineIteration, true, true), :196-207
        k_lba_edge_test<<<nb, 64, 0, c->stream>>>(A.cams, A.edges, A.off, A.points, A.err, n_points, level1);
        A.iterations = refine_iterations; A.robust = 0;
        k_lba_structure_only<<<nb, 64, 0, c->stream>>>(A);
        out->iterations_done[1] = refine_iterations;
    }
    k_lba_edge_test<<<nb, 64, 0, c->stream>>>(A.cams, A.edges, A.off, A.points, A.err, n_points, bad);
    CML_CHECK(c, hipGetLastError());
    cml_d2h_batch_begin(c);
    cml_d2h(c, points, c->lba_points.p, sizeof(double) * 3 * (size_t)n_points);
    cml_d2h(c, edge_bad, bad, (size_t)n_edges);
    if ((rc = cml_d2h_batch_flush(c))) return rc;
    int nbad = 0;
    for (int e = 0; e < n_edges; e++) nbad += edge_bad[e];
    out->n_bad = nbad; out->ok = 1;
    return CMLHIP_OK;
}

}  // extern "C"
