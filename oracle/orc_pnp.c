/* orc_pnp.c — CPU restatement of the ORB side's pose-only optimisation:
 * CML::Optimization::G2O::IndirectCameraOptimizer::optimize (src/cml/optimization/g2o/IndirectCameraOptimizer.cpp:4-195)
 * and evaluateOutliers (:384-427), together with the parts of the vendored g2o (thirdparty/g2o, g2o/config.h is
 * cmake-generated: unbuildable here) it runs through for a single free VertexSE3Expmap and fixed points:
 *   EdgeSE3ProjectXYZ::computeError / linearizeOplus      g2o/types/sba/edge_project_xyz.cpp:44-95
 *   SE3Quat (exp, operator*, map, normalizeRotation)      g2o/types/slam3d/se3quat.h:52-58,96-102,199-229
 *   VertexSE3Expmap::oplusImpl                            g2o/types/sba/vertex_se3_expmap.cpp:48-51
 *   RobustKernelHuber::robustify                          g2o/core/robust_kernel_impl.cpp:60-74
 *   BaseFixedSizedEdge::constructQuadraticForm            g2o/core/base_fixed_sized_edge.hpp:49-133
 *   OptimizationAlgorithmLevenberg::solve                 g2o/core/optimization_algorithm_levenberg.cpp:58-175
 *   SparseOptimizer::optimize                             g2o/core/sparse_optimizer.cpp:392-455
 *   LinearSolverEigen (SimplicialLLT, upper) on the one 6x6 pose block   g2o/solvers/eigen/linear_solver_eigen.h:57-130
 * and Eigen 3.4.0's Quaternion <-> matrix conversions (Eigen/src/Geometry/Quaternion.h).
 * TEST INFRASTRUCTURE ONLY (see cml_oracle.h).  Pinning: the SE3Quat / Eigen arithmetic (orc_g2o.h: exp, product, map,
 * quaternion <-> matrix, LL^T) is pinned on the reference's vendored Eigen 3.4.0 (oracle/_ref, tests/golden/
 * thirdparty_vectors.npz, 1e-14).  The g2o control flow (edge, Huber, Levenberg, the 4 rounds) is PARITY UNPINNED: the
 * reference has no test or fixture for this path; it follows the statements in order (sums over the edges in edge order)
 * and is checked functionally (tests/test_oracle_cpu.py: recovers the true pose, flags the planted outliers). */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "cml_oracle.h"
#include "orc_g2o.h"

typedef struct { const cmlhip_pnp_match* m; int n; const unsigned char* level1; double fx, fy, cx, cy; int robust; double delta; } pnp_graph;

static void edge_error(const pnp_graph* G, const se3q* T, int i, double e[2], double xyz[3]) {   /* computeError */
    double r[3];
    q_rotate(T, G->m[i].X, r);
    xyz[0] = r[0] + T->t[0]; xyz[1] = r[1] + T->t[1]; xyz[2] = r[2] + T->t[2];
    e[0] = G->m[i].obs[0] - (xyz[0] / xyz[2] * G->fx + G->cx);
    e[1] = G->m[i].obs[1] - (xyz[1] / xyz[2] * G->fy + G->cy);
}
/* computeActiveErrors + activeRobustChi2, and buildSystem when H != NULL */
static double evaluate(const pnp_graph* G, const se3q* T, double* H, double* b) {
    double chi = 0;
    if (H) { memset(H, 0, 36 * sizeof(double)); memset(b, 0, 6 * sizeof(double)); }
    for (int i = 0; i < G->n; i++) {
        if (G->level1[i]) continue;
        double e[2], p[3];
        edge_error(G, T, i, e, p);
        const double om = G->m[i].inv_sigma2;
        const double chi2 = e[0] * (om * e[0]) + e[1] * (om * e[1]);
        double rho[3] = {chi2, 1., 0.};
        if (G->robust) orc_huber(chi2, G->delta, rho);
        chi += rho[0];
        if (!H) continue;
        const double x = p[0], y = p[1], z = p[2], z_2 = z * z, fx = G->fx, fy = G->fy;
        double J[2][6];                                                     /* edge_project_xyz.cpp:80-94 */
        J[0][0] = x * y / z_2 * fx; J[0][1] = -(1 + (x * x / z_2)) * fx; J[0][2] = y / z * fx;
        J[0][3] = -1. / z * fx; J[0][4] = 0; J[0][5] = x / z_2 * fx;
        J[1][0] = (1 + y * y / z_2) * fy; J[1][1] = -x * y / z_2 * fy; J[1][2] = -x / z * fy;
        J[1][3] = 0; J[1][4] = -1. / z * fy; J[1][5] = y / z_2 * fy;
        const double w = rho[1] * om;
        const double we[2] = {(-om * e[0]) * rho[1], (-om * e[1]) * rho[1]};
        for (int j = 0; j < 6; j++) {
            b[j] += J[0][j] * we[0] + J[1][j] * we[1];
            const double a0 = J[0][j] * w, a1 = J[1][j] * w;
            for (int k = 0; k < 6; k++) H[j * 6 + k] += a0 * J[0][k] + a1 * J[1][k];
        }
    }
    return chi;
}
static int llt_solve6(const double* Hin, double lambda, const double* b, double* x) {   /* SimplicialLLT: fails on a pivot <= 0 */
    double L[36];
    for (int i = 0; i < 36; i++) L[i] = Hin[i];
    for (int i = 0; i < 6; i++) L[i * 6 + i] += lambda;
    for (int j = 0; j < 6; j++) {
        double d = L[j * 6 + j];
        for (int k = 0; k < j; k++) d -= L[j * 6 + k] * L[j * 6 + k];
        if (!(d > 0)) return 0;
        d = sqrt(d); L[j * 6 + j] = d;
        for (int i = j + 1; i < 6; i++) {
            double s = L[i * 6 + j];
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = s / d;
        }
    }
    double y[6];
    for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k]; y[i] = s / L[i * 6 + i]; }
    for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k]; x[i] = s / L[i * 6 + i]; }
    return 1;
}

typedef struct { double lambda, ni; double H[36], b[6]; } lm_state;

/* SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg; returns the number of solve() calls */
static int lm_optimize(const pnp_graph* G, se3q* T, int iterations, lm_state* S, double* last_chi) {
    int done = 0, ok = 1;
    for (int it = 0; it < iterations && ok; it++) {
        double currentChi = evaluate(G, T, S->H, S->b);
        if (it == 0) {                                                      /* computeLambdaInit */
            double mx = 0;
            for (int j = 0; j < 6; j++) mx = fmax(fabs(S->H[j * 6 + j]), mx);
            S->lambda = 1e-5 * mx; S->ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            const se3q backup = *T;                                         /* push */
            double x[6] = {0, 0, 0, 0, 0, 0};
            const int ok2 = llt_solve6(S->H, S->lambda, S->b, x);
            se3q E, Tn;
            se3q_exp(x, &E); se3q_mul(&E, T, &Tn); *T = Tn;                 /* oplusImpl */
            double tempChi = evaluate(G, T, NULL, NULL);
            if (!ok2) tempChi = 1.7976931348623157e308;
            rho = currentChi - tempChi;
            double scale = 0;
            for (int j = 0; j < 6; j++) scale += x[j] * (S->lambda * x[j] + S->b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow(2 * rho - 1, 3);
                alpha = fmin(alpha, 2. / 3.);
                const double sf = fmax(1. / 3., alpha);
                S->lambda *= sf; S->ni = 2; currentChi = tempChi;
            } else {
                S->lambda *= S->ni; S->ni *= 2;
                *T = backup;                                                /* pop */
                if (!isfinite(S->lambda)) break;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        done++;
        *last_chi = currentChi;
        if (qmax == 10 || rho == 0 || !isfinite(S->lambda)) ok = 0;         /* Terminate */
    }
    return done;
}

/* SparseOptimizer::optimize(iterations) with OptimizationAlgorithmGaussNewton (optimization_algorithm_gauss_newton.cpp:47-94):
 * build, solve, update; a failed factorisation leaves x as it was, the update is still applied and the loop ends (Fail). */
static int gn_optimize(const pnp_graph* G, se3q* T, int iterations, lm_state* S, double* last_chi) {
    int done = 0, ok = 1;
    double x[6] = {0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iterations && ok; it++) {
        *last_chi = evaluate(G, T, S->H, S->b);
        ok = llt_solve6(S->H, 0.0, S->b, x);
        se3q E, Tn;
        se3q_exp(x, &E); se3q_mul(&E, T, &Tn); *T = Tn;
        done++;
    }
    return done;
}

static int inv6_diag(const double* H, double* diag) {          /* computeMarginals on the one block: diag(Hpp^-1) */
    for (int c = 0; c < 6; c++) {
        double e[6] = {0, 0, 0, 0, 0, 0}, x[6];
        e[c] = 1;
        if (!llt_solve6(H, 0.0, e, x)) return 0;
        diag[c] = x[c];
    }
    return 1;
}

/* IndirectCameraOptimizer::optimize(frame, camera, matchings, outliers, computeCovariance), :4-195 (algorithm 0:
 * Levenberg) and optimize(frame, outliersPoints, computeCovariance), :197-382 (algorithm 1: Gauss-Newton, no initial
 * outliers, no inlier-count test before the first round).  The caller has dropped the matchings without a map point
 * (:57-62); outliers is in/out. */
void orc_pnp_optimize(const double R0[9], const double t0[3], const double K[4], int n, const cmlhip_pnp_match* m,
                      unsigned char* outliers, int algorithm, int check_outliers, int compute_covariance, cmlhip_pnp_result* out) {
    memset(out, 0, sizeof *out);
    pnp_graph G;
    G.m = m; G.n = n; G.fx = K[0]; G.fy = K[1]; G.cx = K[2]; G.cy = K[3];
    G.robust = 1; G.delta = (double)(float)sqrt(5.991);                     /* const float deltaMono, :40 */
    unsigned char* level1 = (unsigned char*)malloc((size_t)(n > 0 ? n : 1));
    G.level1 = level1;
    int nBad = 0;
    for (int i = 0; i < n; i++) { level1[i] = outliers[i] ? 1 : 0; nBad += level1[i]; }
    se3q T, T0;
    se3q_from_Rt(R0, t0, &T0); T = T0;
    q_to_matrix(&T, out->R); memcpy(out->t, T.t, sizeof out->t);
    out->n_bad = nBad;
    if (n < 3 || (algorithm == 0 && (n - nBad) < 5)) { free(level1); return; }   /* :121-129 / :312-314 */
    const double chi2Mono = 5.991;
    lm_state S;
    memset(&S, 0, sizeof S);
    for (int it = 0; it < 4; it++) {                                        /* :139-168 */
        T = T0;
        out->lm_iterations[it] = algorithm == 0 ? lm_optimize(&G, &T, 10, &S, &out->chi2[it]) : gn_optimize(&G, &T, 10, &S, &out->chi2[it]);
        nBad = 0;                                                           /* evaluateOutliers, :384-427 */
        for (int i = 0; i < n; i++) {
            if (!check_outliers) { outliers[i] = 0; level1[i] = 0; continue; }
            double e[2], p[3];
            edge_error(&G, &T, i, e, p);
            const float chi2 = (float)(e[0] * (m[i].info * e[0]) + e[1] * (m[i].info * e[1]));
            if (!isfinite(chi2) || (double)chi2 > chi2Mono) { outliers[i] = 1; level1[i] = 1; nBad++; }
            else { outliers[i] = 0; level1[i] = 0; }
        }
        out->n_bad = nBad; out->rounds = it + 1;
        q_to_matrix(&T, out->R); memcpy(out->t, T.t, sizeof out->t);
        if ((n - nBad) < 5) { free(level1); return; }
        if (it == 2) G.robust = 0;
        if (n < 10) { free(level1); return; }                               /* optimizer.edges().size() < 10 */
    }
    if (compute_covariance) {                                               /* :177-190: Hpp as the last buildSystem left it */
        if (!inv6_diag(S.H, out->covariance)) { free(level1); return; }
    }
    out->is_ok = 1;
    free(level1);
}
