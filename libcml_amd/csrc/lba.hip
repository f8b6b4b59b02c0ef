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

extern "C" {

int cmlhip_lba_optimize(cmlhip_ctx* c, int n_frames, cmlhip_lba_frame* frames, int n_points, double* points, const int* point_offsets,
                        const cmlhip_lba_edge* edges, int fix_frames, int num_iterations, int refine_iterations, unsigned char* edge_bad,
                        cmlhip_lba_result* out) {
    if (!c || !out || n_frames < 1 || !frames || n_points < 0 || !point_offsets || num_iterations < 0) return CMLHIP_ERR_INVALID;
    *out = cmlhip_lba_result{};
    CML_REQUIRE(c, fix_frames != 0, CMLHIP_ERR_INVALID, "local BA with free poses (fixFrames == false, g2o Levenberg + Schur) is not built yet: structure-only mode only");
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
    LbaArgs A;
    A.cams = c->lba_cams.as<LbaCam>(); A.edges = c->lba_edges.as<cmlhip_lba_edge>(); A.off = c->lba_off.as<int>();
    A.points = c->lba_points.as<double>(); A.err = c->lba_err.as<double>(); A.level1 = level1;
    A.n_points = n_points; A.delta = (double)sqrtf(5.991f);             // const float thHuberIndirect = sqrt(5.991), :111
    const int nb = cml_div_up(n_points, 64);
    A.iterations = num_iterations; A.robust = 1;                        // startOptimization(mNumIteration, true, false), :193
    if (num_iterations > 0) k_lba_structure_only<<<nb, 64, 0, c->stream>>>(A);
    out->iterations_done[0] = num_iterations;
    if (refine_iterations > 0) {                                        // startOptimization(mRefineIteration, true, true), :196-207
        k_lba_edge_test<<<nb, 64, 0, c->stream>>>(A.cams, A.edges, A.off, A.points, A.err, n_points, level1);
        A.iterations = refine_iterations; A.robust = 0;
        k_lba_structure_only<<<nb, 64, 0, c->stream>>>(A);
        out->iterations_done[1] = refine_iterations;
    }
    k_lba_edge_test<<<nb, 64, 0, c->stream>>>(A.cams, A.edges, A.off, A.points, A.err, n_points, bad);
    CML_CHECK(c, hipGetLastError());
    if ((rc = cml_d2h(c, points, c->lba_points.p, sizeof(double) * 3 * (size_t)n_points))) return rc;
    if ((rc = cml_d2h(c, edge_bad, bad, (size_t)n_edges))) return rc;
    int nbad = 0;
    for (int e = 0; e < n_edges; e++) nbad += edge_bad[e];
    out->n_bad = nbad; out->ok = 1;
    return CMLHIP_OK;
}

}  // extern "C"
