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
 * TEST INFRASTRUCTURE ONLY (see cml_oracle.h).  Parity unpinned: the reference has no test or fixture for this path; the
 * restatement follows the statements in order (sums over the edges in edge order) and is checked functionally
 * (tests/test_oracle_cpu.py: recovers the true pose, flags the planted outliers). */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "cml_oracle.h"

typedef struct { double x, y, z, w; double t[3]; } se3q;       /* Eigen coefficient order */

static void q_from_matrix(const double m[9], se3q* q) {        /* Quaternion.h quaternionbase_assign_impl<Other,3,3> */
    double tr = m[0] + m[4] + m[8];
    if (tr > 0) {
        double t = sqrt(tr + 1.0);
        q->w = 0.5 * t; t = 0.5 / t;
        q->x = (m[7] - m[5]) * t; q->y = (m[2] - m[6]) * t; q->z = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        double v[3];
        v[i] = 0.5 * t; t = 0.5 / t;
        q->w = (m[k * 3 + j] - m[j * 3 + k]) * t;
        v[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        v[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        q->x = v[0]; q->y = v[1]; q->z = v[2];
    }
}
static void q_normalize(se3q* q) {                             /* se3quat.h normalizeRotation */
    if (q->w < 0) { q->x *= -1; q->y *= -1; q->z *= -1; q->w *= -1; }
    double n = sqrt(q->x * q->x + q->y * q->y + q->z * q->z + q->w * q->w);
    q->x /= n; q->y /= n; q->z /= n; q->w /= n;
}
static void q_rotate(const se3q* q, const double v[3], double o[3]) {   /* QuaternionBase::_transformVector */
    double uv[3] = {q->y * v[2] - q->z * v[1], q->z * v[0] - q->x * v[2], q->x * v[1] - q->y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q->w * uv[0] + (q->y * uv[2] - q->z * uv[1]);
    o[1] = v[1] + q->w * uv[1] + (q->z * uv[0] - q->x * uv[2]);
    o[2] = v[2] + q->w * uv[2] + (q->x * uv[1] - q->y * uv[0]);
}
static void q_to_matrix(const se3q* q, double R[9]) {          /* QuaternionBase::toRotationMatrix */
    const double tx = 2 * q->x, ty = 2 * q->y, tz = 2 * q->z;
    const double twx = tx * q->w, twy = ty * q->w, twz = tz * q->w, txx = tx * q->x, txy = ty * q->x, txz = tz * q->x,
                 tyy = ty * q->y, tyz = tz * q->y, tzz = tz * q->z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static void se3q_from_Rt(const double R[9], const double t[3], se3q* T) {   /* SE3Quat(R, t) */
    q_from_matrix(R, T); q_normalize(T);
    T->t[0] = t[0]; T->t[1] = t[1]; T->t[2] = t[2];
}
static void se3q_exp(const double u[6], se3q* T) {             /* se3quat.h:201-229 */
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9], R[9], V[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
    double a, b, c, d;
    if (theta < 0.00001) { a = 1; b = 0.5; c = 0.5; d = 1. / 6.; }
    else {
        a = sin(theta) / theta; b = (1 - cos(theta)) / (theta * theta);
        c = b; d = (theta - sin(theta)) / pow(theta, 3);
    }
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + a * O[i] + b * O2[i];
        V[i] = I + c * O[i] + d * O2[i];
    }
    q_from_matrix(R, T); q_normalize(T);
    for (int i = 0; i < 3; i++) T->t[i] = V[i * 3] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
}
static void se3q_mul(const se3q* A, const se3q* B, se3q* C) {  /* se3quat.h:96-102 */
    double rt[3];
    q_rotate(A, B->t, rt);
    se3q r;
    r.t[0] = A->t[0] + rt[0]; r.t[1] = A->t[1] + rt[1]; r.t[2] = A->t[2] + rt[2];
    r.w = A->w * B->w - A->x * B->x - A->y * B->y - A->z * B->z;
    r.x = A->w * B->x + A->x * B->w + A->y * B->z - A->z * B->y;
    r.y = A->w * B->y + A->y * B->w + A->z * B->x - A->x * B->z;
    r.z = A->w * B->z + A->z * B->w + A->x * B->y - A->y * B->x;
    q_normalize(&r);
    *C = r;
}

typedef struct { const cmlhip_pnp_match* m; int n; const unsigned char* level1; double fx, fy, cx, cy; int robust; double delta; } pnp_graph;

static void edge_error(const pnp_graph* G, const se3q* T, int i, double e[2], double xyz[3]) {   /* computeError */
    double r[3];
    q_rotate(T, G->m[i].X, r);
    xyz[0] = r[0] + T->t[0]; xyz[1] = r[1] + T->t[1]; xyz[2] = r[2] + T->t[2];
    e[0] = G->m[i].obs[0] - (xyz[0] / xyz[2] * G->fx + G->cx);
    e[1] = G->m[i].obs[1] - (xyz[1] / xyz[2] * G->fy + G->cy);
}
static void huber(double e, double delta, double rho[3]) {     /* robust_kernel_impl.cpp:60-74 */
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else { const double sq = sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; rho[2] = -0.5 * rho[1] / e; }
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
        if (G->robust) huber(chi2, G->delta, rho);
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
