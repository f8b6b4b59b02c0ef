/* orc_lba.c — CPU restatement of the ORB side's local bundle adjustment:
 * CML::Optimization::G2O::IndirectBundleAdjustment::localOptimize / startOptimization / apply's edge test
 * (src/cml/optimization/g2o/IndirectBundleAdjustment.cpp:7-236,:325-337) over the vendored g2o (thirdparty/g2o; needs the
 * cmake-generated g2o/config.h: unbuildable here).  Restated slices of g2o:
 *   fixFrames == true  (mBaMode != BAINDIRECT, indirect/Mapping.cpp:89): StructureOnlySolver<3>::calc
 *                      g2o/solvers/structure_only/structure_only_solver.h:66-217 — every point on its own, one damped
 *                      Gauss-Newton iteration per optimize() iteration with up to 10 trials, Eigen::LDLT 3x3;
 *   EdgeSE3ProjectXYZ  g2o/types/sba/edge_project_xyz.cpp:44-95;  RobustKernelHuber  g2o/core/robust_kernel_impl.cpp:60-74;
 *   constructQuadraticForm  g2o/core/base_fixed_sized_edge.hpp:49-133;  Eigen LDLT  Eigen/src/Cholesky/LDLT.h:300-396,560-600.
 * TEST INFRASTRUCTURE ONLY (see cml_oracle.h).  Pinning: the Eigen pieces (LDLT<Matrix3d> incl. isPositive(), Matrix3d::inverse,
 * LL^T, the SE3Quat arithmetic of orc_g2o.h) are pinned on the reference's vendored Eigen 3.4.0 (oracle/_ref, tests/golden/
 * thirdparty_vectors.npz).  The g2o control flow (structure-only calc, Levenberg, Schur solve) is PARITY UNPINNED: the reference
 * has no test or fixture for this path; checked functionally (tests/test_oracle_cpu.py).  Literal behaviour kept: an edge's chi2() is whatever its last computeError() left — after a rejected
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


/* ================================================================== free poses: g2o Levenberg + Schur (fixFrames == false)
 * OptimizationAlgorithmLevenberg::solve (g2o/core/optimization_algorithm_levenberg.cpp:58-175) over BlockSolver_6_3 with the
 * points marginalised (buildSystem / setLambda / solve, g2o/core/block_solver.hpp:329-479,495-587) and LinearSolverEigen
 * (SimplicialLLT) on the reduced pose system — restated densely: the sparse block structure only skips zero blocks. */

/* _jacobianOplusXj, edge_project_xyz.cpp:80-94 */
static void edge_jacobian_pose(const lba_cam* C, const double p[3], double J[2][6]) {
    const double x = p[0], y = p[1], z = p[2], z_2 = z * z, fx = C->K[0], fy = C->K[1];
    J[0][0] = x * y / z_2 * fx; J[0][1] = -(1 + (x * x / z_2)) * fx; J[0][2] = y / z * fx;
    J[0][3] = -1. / z * fx; J[0][4] = 0; J[0][5] = x / z_2 * fx;
    J[1][0] = (1 + y * y / z_2) * fy; J[1][1] = -x * y / z_2 * fy; J[1][2] = -x / z * fy;
    J[1][3] = 0; J[1][4] = -1. / z * fy; J[1][5] = y / z_2 * fy;
}

typedef struct {
    int n_frames, n_points, n_edges, nfree;
    const int* off; const cmlhip_lba_edge* edges;
    int* pidx;                    /* frame -> pose block, -1: fixed or without an active edge */
    unsigned char* pt_active;     /* point has a level-0 edge */
    unsigned char* level1;
    int robust; double delta;
    double *Hpp, *bp, *Hll, *bl, *Hpl, *err;
} lba_graph;

/* computeActiveErrors + activeRobustChi2 (+ buildSystem when build != 0) at (cams, points) */
static double lba_evaluate(lba_graph* G, const lba_cam* cams, const double* points, int build) {
    double chi = 0;
    if (build) {
        memset(G->Hpp, 0, sizeof(double) * 36 * (size_t)G->nfree); memset(G->bp, 0, sizeof(double) * 6 * (size_t)G->nfree);
        memset(G->Hll, 0, sizeof(double) * 9 * (size_t)G->n_points); memset(G->bl, 0, sizeof(double) * 3 * (size_t)G->n_points);
        memset(G->Hpl, 0, sizeof(double) * 18 * (size_t)G->n_edges);
    }
    for (int pt = 0; pt < G->n_points; pt++)
        for (int k = G->off[pt]; k < G->off[pt + 1]; k++) {
            if (G->level1[k]) continue;
            const cmlhip_lba_edge* E = &G->edges[k];
            const lba_cam* C = &cams[E->frame];
            double p[3], *e = &G->err[2 * (size_t)k];
            edge_compute_error(C, &points[3 * pt], E, e, p);
            const double om = E->inv_sigma2;
            double rho[3] = {edge_chi2(e, om), 1., 0};
            if (G->robust) orc_huber(rho[0], G->delta, rho);
            chi += rho[0];
            if (!build) continue;
            double Jl[2][3], Jp[2][6];
            edge_jacobian_point(C, p, Jl);
            const double w = rho[1] * om;
            const double we[2] = {(-om * e[0]) * rho[1], (-om * e[1]) * rho[1]};
            double* Hll = &G->Hll[9 * (size_t)pt]; double* bl = &G->bl[3 * (size_t)pt];
            for (int j = 0; j < 3; j++) {
                bl[j] += Jl[0][j] * we[0] + Jl[1][j] * we[1];
                for (int c = 0; c < 3; c++) Hll[j * 3 + c] += (Jl[0][j] * w) * Jl[0][c] + (Jl[1][j] * w) * Jl[1][c];
            }
            const int pi = G->pidx[E->frame];
            if (pi < 0) continue;
            edge_jacobian_pose(C, p, Jp);
            double* Hpp = &G->Hpp[36 * (size_t)pi]; double* bp = &G->bp[6 * (size_t)pi]; double* Hpl = &G->Hpl[18 * (size_t)k];
            for (int j = 0; j < 6; j++) {
                bp[j] += Jp[0][j] * we[0] + Jp[1][j] * we[1];
                for (int c = 0; c < 6; c++) Hpp[j * 6 + c] += (Jp[0][j] * w) * Jp[0][c] + (Jp[1][j] * w) * Jp[1][c];
                for (int c = 0; c < 3; c++) Hpl[j * 3 + c] += (Jp[0][j] * w) * Jl[0][c] + (Jp[1][j] * w) * Jl[1][c];
            }
        }
    return chi;
}

/* BlockSolver::solve with Schur (block_solver.hpp:343-478) at damping lambda: xp (6 nfree), xl (3 n_points) */
static int lba_schur_solve(const lba_graph* G, double lambda, double* xp, double* xl) {
    const int n = 6 * G->nfree;
    double* S = (double*)calloc((size_t)n * n, sizeof(double));
    double* bs = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double* Dinv = (double*)malloc(sizeof(double) * 9 * (size_t)G->n_points);
    for (int f = 0; f < G->nfree; f++) {
        for (int r = 0; r < 6; r++) {
            for (int c = 0; c < 6; c++) S[(size_t)(6 * f + r) * n + 6 * f + c] = G->Hpp[36 * (size_t)f + r * 6 + c];
            S[(size_t)(6 * f + r) * n + 6 * f + r] += lambda;
            bs[6 * f + r] = G->bp[6 * (size_t)f + r];
        }
    }
    for (int pt = 0; pt < G->n_points; pt++) {
        if (!G->pt_active[pt]) continue;
        double D[9], db[3];
        memcpy(D, &G->Hll[9 * (size_t)pt], sizeof D);
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        double* Di = &Dinv[9 * (size_t)pt];
        inv3(D, Di);
        const double* bl = &G->bl[3 * (size_t)pt];
        for (int r = 0; r < 3; r++) db[r] = Di[r * 3] * bl[0] + Di[r * 3 + 1] * bl[1] + Di[r * 3 + 2] * bl[2];
        for (int k1 = G->off[pt]; k1 < G->off[pt + 1]; k1++) {
            const int i1 = G->pidx[G->edges[k1].frame];
            if (i1 < 0 || G->level1[k1]) continue;
            const double* Bi = &G->Hpl[18 * (size_t)k1];
            double BD[18];
            for (int r = 0; r < 6; r++) {
                for (int c = 0; c < 3; c++) BD[r * 3 + c] = Bi[r * 3] * Di[c] + Bi[r * 3 + 1] * Di[3 + c] + Bi[r * 3 + 2] * Di[6 + c];
                bs[6 * i1 + r] -= Bi[r * 3] * db[0] + Bi[r * 3 + 1] * db[1] + Bi[r * 3 + 2] * db[2];
            }
            for (int k2 = G->off[pt]; k2 < G->off[pt + 1]; k2++) {
                const int i2 = G->pidx[G->edges[k2].frame];
                if (i2 < 0 || G->level1[k2]) continue;
                const double* Bj = &G->Hpl[18 * (size_t)k2];
                for (int r = 0; r < 6; r++)
                    for (int c = 0; c < 6; c++)
                        S[(size_t)(6 * i1 + r) * n + 6 * i2 + c] -= BD[r * 3] * Bj[c * 3] + BD[r * 3 + 1] * Bj[c * 3 + 1] + BD[r * 3 + 2] * Bj[c * 3 + 2];
            }
        }
    }
    const int ok = n == 0 ? 1 : chol_solve_dense(S, n, bs, xp);
    if (ok)
        for (int pt = 0; pt < G->n_points; pt++) {
            double* x = &xl[3 * (size_t)pt];
            x[0] = x[1] = x[2] = 0;
            if (!G->pt_active[pt]) continue;
            double cl[3] = {G->bl[3 * (size_t)pt], G->bl[3 * (size_t)pt + 1], G->bl[3 * (size_t)pt + 2]};
            for (int k = G->off[pt]; k < G->off[pt + 1]; k++) {
                const int i1 = G->pidx[G->edges[k].frame];
                if (i1 < 0 || G->level1[k]) continue;
                const double* B = &G->Hpl[18 * (size_t)k];
                for (int c = 0; c < 3; c++)
                    for (int r = 0; r < 6; r++) cl[c] += B[r * 3 + c] * (-xp[6 * i1 + r]);
            }
            const double* Di = &Dinv[9 * (size_t)pt];
            for (int r = 0; r < 3; r++) x[r] = Di[r * 3] * cl[0] + Di[r * 3 + 1] * cl[1] + Di[r * 3 + 2] * cl[2];
        }
    free(S); free(bs); free(Dinv);
    return ok;
}

/* SparseOptimizer::optimize(iterations) with Levenberg over the graph; returns the number of solve() calls */
static int lba_lm_optimize(lba_graph* G, lba_cam* cams, double* points, int iterations, double* last_chi) {
    const int np3 = 3 * G->n_points, n6 = 6 * G->nfree;
    double* xp = (double*)calloc((size_t)(n6 > 0 ? n6 : 1), sizeof(double));
    double* xl = (double*)calloc((size_t)(np3 > 0 ? np3 : 1), sizeof(double));
    lba_cam* cam_bak = (lba_cam*)malloc(sizeof(lba_cam) * (size_t)G->n_frames);
    double* pts_bak = (double*)malloc(sizeof(double) * (size_t)(np3 > 0 ? np3 : 1));
    double lambda = 0, ni = 2;
    int done = 0, ok = 1;
    for (int it = 0; it < iterations && ok; it++) {
        double currentChi = lba_evaluate(G, cams, points, 1);
        if (it == 0) {                                                  /* computeLambdaInit over every active vertex */
            double mx = 0;
            for (int f = 0; f < G->nfree; f++) for (int j = 0; j < 6; j++) mx = fmax(fabs(G->Hpp[36 * (size_t)f + j * 7]), mx);
            for (int pt = 0; pt < G->n_points; pt++) if (G->pt_active[pt]) for (int j = 0; j < 3; j++) mx = fmax(fabs(G->Hll[9 * (size_t)pt + j * 4]), mx);
            lambda = 1e-5 * mx; ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            memcpy(cam_bak, cams, sizeof(lba_cam) * (size_t)G->n_frames); memcpy(pts_bak, points, sizeof(double) * (size_t)np3);   /* push */
            const int ok2 = lba_schur_solve(G, lambda, xp, xl);
            for (int f = 0; f < G->n_frames; f++) {                     /* update: oplus on every active vertex */
                const int pi = G->pidx[f];
                if (pi < 0) continue;
                se3q E, Tn;
                se3q_exp(&xp[6 * pi], &E); se3q_mul(&E, &cams[f].T, &Tn);
                cams[f].T = Tn; q_to_matrix(&Tn, cams[f].R);
            }
            for (int pt = 0; pt < G->n_points; pt++) if (G->pt_active[pt]) for (int c = 0; c < 3; c++) points[3 * pt + c] += xl[3 * pt + c];
            double tempChi = lba_evaluate(G, cams, points, 0);
            if (!ok2) tempChi = 1.7976931348623157e308;
            rho = currentChi - tempChi;
            double scale = 0;                                           /* computeScale: poses, then points */
            for (int j = 0; j < n6; j++) scale += xp[j] * (lambda * xp[j] + G->bp[j]);
            for (int pt = 0; pt < G->n_points; pt++) if (G->pt_active[pt]) for (int c = 0; c < 3; c++) scale += xl[3 * pt + c] * (lambda * xl[3 * pt + c] + G->bl[3 * pt + c]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow(2 * rho - 1, 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                memcpy(cams, cam_bak, sizeof(lba_cam) * (size_t)G->n_frames); memcpy(points, pts_bak, sizeof(double) * (size_t)np3);   /* pop */
                if (!isfinite(lambda)) break;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        done++;
        *last_chi = currentChi;
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) ok = 0;
    }
    free(xp); free(xl); free(cam_bak); free(pts_bak);
    return done;
}

static void lba_activate(lba_graph* G, const cmlhip_lba_frame* frames) {  /* initializeOptimization(0): vertices with a level-0 edge */
    unsigned char* fa = (unsigned char*)calloc((size_t)G->n_frames, 1);
    for (int pt = 0; pt < G->n_points; pt++) {
        G->pt_active[pt] = 0;
        for (int k = G->off[pt]; k < G->off[pt + 1]; k++) if (!G->level1[k]) { G->pt_active[pt] = 1; fa[G->edges[k].frame] = 1; }
    }
    G->nfree = 0;
    for (int f = 0; f < G->n_frames; f++) G->pidx[f] = (!frames[f].fixed && fa[f]) ? G->nfree++ : -1;
    free(fa);
}

/* IndirectBundleAdjustment::localOptimize with the graph given as arrays: edges are point-major (the order :120-165 creates
 * them in), point p owns edges [off[p], off[p+1]).  fix_frames != 0: StructureOnlySolver (poses untouched).
 * edge_bad[e] = the removal test of apply() (:327). */
int orc_lba_optimize(int n_frames, cmlhip_lba_frame* frames, int n_points, double* points, const int* off,
                     const cmlhip_lba_edge* edges, int fix_frames, int num_iterations, int refine_iterations,
                     unsigned char* edge_bad, cmlhip_lba_result* out) {
    memset(out, 0, sizeof *out);
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
        if (!fix_frames) {                                              /* OptimizationAlgorithmLevenberg over BlockSolver_6_3 */
            lba_graph G;
            memset(&G, 0, sizeof G);
            G.n_frames = n_frames; G.n_points = n_points; G.n_edges = n_edges; G.off = off; G.edges = edges;
            G.level1 = level1; G.robust = phase == 0; G.delta = delta; G.err = err;
            G.pidx = (int*)malloc(sizeof(int) * (size_t)n_frames);
            G.pt_active = (unsigned char*)malloc((size_t)(n_points > 0 ? n_points : 1));
            lba_activate(&G, frames);
            G.Hpp = (double*)malloc(sizeof(double) * 36 * (size_t)(G.nfree > 0 ? G.nfree : 1)); G.bp = (double*)malloc(sizeof(double) * 6 * (size_t)(G.nfree > 0 ? G.nfree : 1));
            G.Hll = (double*)malloc(sizeof(double) * 9 * (size_t)(n_points > 0 ? n_points : 1)); G.bl = (double*)malloc(sizeof(double) * 3 * (size_t)(n_points > 0 ? n_points : 1));
            G.Hpl = (double*)malloc(sizeof(double) * 18 * (size_t)(n_edges > 0 ? n_edges : 1));
            out->iterations_done[phase] = lba_lm_optimize(&G, cams, points, iters, &out->chi2[phase]);
            free(G.pidx); free(G.pt_active); free(G.Hpp); free(G.bp); free(G.Hll); free(G.bl); free(G.Hpl);
            continue;
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
    if (!fix_frames)                                                    /* apply(): pKF->setCamera of the local keyframes, :301-305 */
        for (int f = 0; f < n_frames; f++) {
            if (frames[f].fixed) continue;
            memcpy(frames[f].R, cams[f].R, sizeof frames[f].R);
            memcpy(frames[f].t, cams[f].T.t, sizeof frames[f].t);
        }
    free(cams); free(err); free(level1);
    return 0;
}
