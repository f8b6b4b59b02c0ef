/* orc_g2o.c — exported handles on the shared g2o / Eigen arithmetic of orc_g2o.h, so that tests can pin it against the
 * vendored Eigen (oracle/_ref, tests/golden/thirdparty_vectors.npz).  TEST INFRASTRUCTURE ONLY (see cml_oracle.h).
 * Quaternions cross this boundary as x, y, z, w. */
#include <stdlib.h>
#include <string.h>
#include "cml_oracle.h"
#include "orc_g2o.h"

static void out_qt(const se3q* T, double q[4], double t[3]) { q[0] = T->x; q[1] = T->y; q[2] = T->z; q[3] = T->w; memcpy(t, T->t, 3 * sizeof(double)); }
static void in_qt(const double q[4], const double t[3], se3q* T) { T->x = q[0]; T->y = q[1]; T->z = q[2]; T->w = q[3]; memcpy(T->t, t, 3 * sizeof(double)); }

void orc_g2o_from_Rt(const double R[9], const double t[3], double q[4], double tt[3]) { se3q T; se3q_from_Rt(R, t, &T); out_qt(&T, q, tt); }
void orc_g2o_exp(const double u[6], double q[4], double tt[3]) { se3q T; se3q_exp(u, &T); out_qt(&T, q, tt); }
void orc_g2o_mul(const double qa[4], const double ta[3], const double qb[4], const double tb[3], double q[4], double tt[3]) {
    se3q A, B, C; in_qt(qa, ta, &A); in_qt(qb, tb, &B); se3q_mul(&A, &B, &C); out_qt(&C, q, tt);
}
void orc_g2o_map(const double q[4], const double t[3], const double X[3], double out[3]) {
    se3q T; in_qt(q, t, &T);
    double r[3]; q_rotate(&T, X, r);
    for (int i = 0; i < 3; i++) out[i] = r[i] + T.t[i];
}
void orc_g2o_to_matrix(const double q[4], double R[9]) { se3q T; const double z[3] = {0, 0, 0}; in_qt(q, z, &T); q_to_matrix(&T, R); }
void orc_g2o_inv3(const double A[9], double Ai[9]) { inv3(A, Ai); }
int orc_g2o_llt_solve(const double* A, int n, const double* b, double* x) {
    double* W = (double*)malloc(sizeof(double) * (size_t)n * n);
    memcpy(W, A, sizeof(double) * (size_t)n * n);
    const int ok = chol_solve_dense(W, n, b, x);
    free(W);
    return ok;
}
