/* orc_lba.c — CPU restatement of the ORB side's local bundle adjustment:
 * CML::Optimization::G2O::IndirectBundleAdjustment::localOptimize / startOptimization / apply's edge test
 * (src/cml/optimization/g2o/IndirectBundleAdjustment.cpp:7-236,:325-337) over the vendored g2o (thirdparty/g2o; needs the
 * cmake-generated g2o/config.h: unbuildable here).  Restated slices of g2o:
 *   fixFrames == true  (mBaMode != BAINDIRECT, indirect/Mapping.cpp:89): StructureOnlySolver<3>::calc
 *                      g2o/solvers/structure_only/structure_only_solver.h:66-217 — every point on its own, one damped
 *                      Gauss-Newton iteration per optimize() iteration with up to 10 trials, Eigen::LDLT 3x3;
 *   EdgeSE3ProjectXYZ  g2o/types/sba/edge_project_xyz.cpp:44-95;  RobustKernelHuber  g2o/core/robust_kernel_impl.cpp:60-74;
 *   constructQuadraticForm  g2o/core/base_fixed_sized_edge.hpp:49-133;  Eigen LDLT  Eigen/src/Cholesky/LDLT.h:300-396,560-600.
 * TEST INFRASTRUCTURE ONLY (see cml_oracle.h).  Parity unpinned: the reference has no test or fixture for this path; checked
 * functionally (tests/test_oracle_cpu.py) and, for the 3x3 LDLT, against the restatement that IS pinned on the vendored Eigen
 * (orc_ldlt_solve).  Literal behaviour kept: an edge's chi2() is whatever its last computeError() left — after a rejected
 * trial that is the error at the rejected point position — and that is what the level test of the refinement pass (:214) and
 * apply()'s removal test (:327) read. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "cml_oracle.h"
#include "orc_g2o.h"

typedef struct { se3q T; double R[9]; double K[4]; } lba_cam;

static void edge_compute_error(const lba_cam* C, const double X[3], const cmlhip_lba_edge* E, double e[2], double p[3]) {
    double r[3];
    q_rotate(&C->T, X, r);
    p[0] = r[0] + C->T.t[0]; p[1] = r[1] + C->T.t[1]; p[2] = r[2] + C->T.t[2];
    e[0] = E->obs[0] - (p[0] / p[2] * C->K[0] + C->K[2]);
    e[1] = E->obs[1] - (p[1] / p[2] * C->K[1] + C->K[3]);
}
static double edge_chi2(const double e[2], double om) { return e[0] * (om * e[0]) + e[1] * (om * e[1]); }

/* _jacobianOplusXi = -1./z * tmp * R, edge_project_xyz.cpp:68-78 */
static void edge_jacobian_point(const lba_cam* C, const double p[3], double J[2][3]) {
    const double x = p[0], y = p[1], z = p[2], fx = C->K[0], fy = C->K[1];
    const double tmp[2][3] = {{fx, 0, -x / z * fx}, {0, fy, -y / z * fy}};
    const double s = -1. / z;
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++)
            J[r][c] = ((s * tmp[r][0]) * C->R[c] + (s * tmp[r][1]) * C->R[3 + c]) + (s * tmp[r][2]) * C->R[6 + c];
}

/* Eigen::LDLT<Matrix3d>(A): factor with diagonal pivoting, report isPositive(), solve A x = b (LDLT.h:300-396,560-600) */
int orc_ldlt3(const double Ain[9], const double b[3], double x[3]) {
    double A[9];
    int tr[3];
    memcpy(A, Ain, sizeof A);
    enum { ZERO, POS, NEG, INDEF } sign = ZERO;
#define M(i, j) A[(i) * 3 + (j)]
    for (int k = 0; k < 3; k++) {
        int big = k;
        double best = fabs(M(k, k));
        for (int i = k + 1; i < 3; i++) if (fabs(M(i, i)) > best) { best = fabs(M(i, i)); big = i; }
        tr[k] = big;
        if (k != big) {
            const int s = 3 - big - 1;
            for (int j = 0; j < k; j++) { double t = M(k, j); M(k, j) = M(big, j); M(big, j) = t; }
            for (int i = 0; i < s; i++) { double t = M(big + 1 + i, k); M(big + 1 + i, k) = M(big + 1 + i, big); M(big + 1 + i, big) = t; }
            { double t = M(k, k); M(k, k) = M(big, big); M(big, big) = t; }
            for (int i = k + 1; i < big; i++) { double t = M(i, k); M(i, k) = M(big, i); M(big, i) = t; }
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
        const int valid = fabs(akk) > 0.0;
        if (k == 0 && !valid) { sign = ZERO; for (int j = 0; j < 3; j++) tr[j] = j; break; }
        if (rs > 0 && valid) for (int i = 0; i < rs; i++) M(k + 1 + i, k) /= akk;
        if (sign == POS) { if (akk < 0) sign = INDEF; }
        else if (sign == NEG) { if (akk > 0) sign = INDEF; }
        else if (sign == ZERO) { if (akk > 0) sign = POS; else if (akk < 0) sign = NEG; }
    }
    for (int i = 0; i < 3; i++) x[i] = b[i];
    for (int k = 0; k < 3; k++) if (tr[k] != k) { double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
    for (int i = 0; i < 3; i++) { double s = x[i]; for (int j = 0; j < i; j++) s -= M(i, j) * x[j]; x[i] = s; }
    const double tol = 2.2250738585072014e-308;
    for (int i = 0; i < 3; i++) { if (fabs(M(i, i)) > tol) x[i] /= M(i, i); else x[i] = 0; }
    for (int i = 2; i >= 0; i--) { double s = x[i]; for (int j = i + 1; j < 3; j++) s -= M(j, i) * x[j]; x[i] = s; }
    for (int k = 2; k >= 0; k--) if (tr[k] != k) { double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
#undef M
    return sign == POS || sign == ZERO;
}

/* sum of the (robustified) chi2 of the point's track at X; every edge's stored error is refreshed (computeError) */
static double track_chi2(const lba_cam* cams, const cmlhip_lba_edge* E, int n, const double X[3], int robust, double delta, double* err) {
    double chi2 = 0;
    for (int k = 0; k < n; k++) {
        double p[3];
        edge_compute_error(&cams[E[k].frame], X, &E[k], &err[2 * k], p);
        const double c = edge_chi2(&err[2 * k], E[k].inv_sigma2);
        if (robust) { double rho[3]; orc_huber(c, delta, rho); chi2 += rho[0]; }
        else chi2 += c;
    }
    return chi2;
}

/* StructureOnlySolver<3>::calc(points, 1) for one point, structure_only_solver.h:74-215 */
static void structure_only_point(const lba_cam* cams, const cmlhip_lba_edge* E, int n, double X[3], int robust, double delta, double* err) {
    double chi2 = track_chi2(cams, E, n, X, robust, delta, err);
    double mu = 0.01, nu = 2;
    int stop = 0;
    for (int i_g = 0; i_g < 1; ++i_g) {
        double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
        for (int k = 0; k < n; k++) {
            const lba_cam* C = &cams[E[k].frame];
            double p[3], J[2][3];
            edge_compute_error(C, X, &E[k], &err[2 * k], p);
            edge_jacobian_point(C, p, J);
            const double om = E[k].inv_sigma2, *e = &err[2 * k];
            double rho[3] = {0, 1., 0};
            if (robust) orc_huber(edge_chi2(e, om), delta, rho);
            const double w = rho[1] * om;
            const double we[2] = {(-om * e[0]) * rho[1], (-om * e[1]) * rho[1]};
            for (int j = 0; j < 3; j++) {
                b[j] += J[0][j] * we[0] + J[1][j] * we[1];
                const double a0 = J[0][j] * w, a1 = J[1][j] * w;
                for (int c = 0; c < 3; c++) H[j * 3 + c] += a0 * J[0][c] + a1 * J[1][c];
            }
        }
        if (sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]) < 0.001) { stop = 1; break; }
        int trial = 0;
        do {
            double Hmu[9], dp[3];
            memcpy(Hmu, H, sizeof Hmu);
            Hmu[0] += mu; Hmu[4] += mu; Hmu[8] += mu;
            int good = 0;
            if (orc_ldlt3(Hmu, b, dp)) {
                const double Xn[3] = {X[0] + dp[0], X[1] + dp[1], X[2] + dp[2]};
                const double new_chi2 = track_chi2(cams, E, n, Xn, robust, delta, err);
                const double rho = chi2 - new_chi2;
                if (rho > 0 && isfinite(new_chi2)) { good = 1; chi2 = new_chi2; X[0] = Xn[0]; X[1] = Xn[1]; X[2] = Xn[2]; }
            }
            if (good) { mu *= 1. / 3.; nu = 2.; trial = 0; break; }
            mu *= nu; nu *= 2.; ++trial;
            if (trial >= 10) { stop = 1; break; }
        } while (!stop);
        if (stop) break;
    }
}

/* IndirectBundleAdjustment::localOptimize with the graph given as arrays: edges are point-major (the order :120-165 creates
 * them in), point p owns edges [off[p], off[p+1]).  fix_frames != 0: StructureOnlySolver (poses untouched).
 * edge_bad[e] = the removal test of apply() (:327). */
int orc_lba_optimize(int n_frames, cmlhip_lba_frame* frames, int n_points, double* points, const int* off,
                     const cmlhip_lba_edge* edges, int fix_frames, int num_iterations, int refine_iterations,
                     unsigned char* edge_bad, cmlhip_lba_result* out) {
    memset(out, 0, sizeof *out);
    if (!fix_frames) return CMLHIP_ERR_INVALID;
    const int n_edges = off[n_points];
    const double delta = (double)(float)sqrt(5.991);                    /* const float thHuberIndirect, :111 */
    lba_cam* cams = (lba_cam*)malloc(sizeof(lba_cam) * (size_t)(n_frames > 0 ? n_frames : 1));
    double* err = (double*)calloc((size_t)(n_edges > 0 ? n_edges : 1) * 2, sizeof(double));
    unsigned char* level1 = (unsigned char*)calloc((size_t)(n_edges > 0 ? n_edges : 1), 1);
    for (int f = 0; f < n_frames; f++) {
        se3q_from_Rt(frames[f].R, frames[f].t, &cams[f].T);
        q_to_matrix(&cams[f].T, cams[f].R);
        memcpy(cams[f].K, frames[f].K, sizeof cams[f].K);
    }
    for (int phase = 0; phase < 2; phase++) {                           /* startOptimization(num, .., false) then (refine, .., true) */
        const int iters = phase == 0 ? num_iterations : refine_iterations;
        if (phase == 1) {
            if (refine_iterations <= 0) break;                          /* :196 */
            for (int p = 0; p < n_points; p++)                          /* :210-221 */
                for (int k = off[p]; k < off[p + 1]; k++) {
                    double r[3];
                    q_rotate(&cams[edges[k].frame].T, &points[3 * p], r);
                    const int depth_pos = (r[2] + cams[edges[k].frame].T.t[2]) > 0.0;
                    level1[k] = (edge_chi2(&err[2 * k], edges[k].inv_sigma2) > 5.991 || !depth_pos) ? 1 : 0;
                }
        }
        for (int it = 0; it < iters; it++) {                            /* optimize(num): one calc(points, 1) per iteration */
            for (int p = 0; p < n_points; p++) {
                const int n = off[p + 1] - off[p];
                int active = 0;                                         /* activeVertices: a level-0 edge is attached */
                for (int k = off[p]; k < off[p + 1]; k++) active |= !level1[k];
                if (!active || n == 0) continue;
                structure_only_point(cams, edges + off[p], n, &points[3 * p], phase == 0, delta, err + 2 * (size_t)off[p]);
            }
            out->iterations_done[phase]++;
        }
    }
    int nbad = 0;
    for (int p = 0; p < n_points; p++)                                  /* apply(), :322-334 */
        for (int k = off[p]; k < off[p + 1]; k++) {
            double r[3];
            q_rotate(&cams[edges[k].frame].T, &points[3 * p], r);
            const int depth_pos = (r[2] + cams[edges[k].frame].T.t[2]) > 0.0;
            edge_bad[k] = (edge_chi2(&err[2 * k], edges[k].inv_sigma2) > 5.991 || !depth_pos) ? 1 : 0;
            nbad += edge_bad[k];
        }
    out->n_bad = nbad; out->ok = 1;
    free(cams); free(err); free(level1);
    return 0;
}
