/* orc_g2o.h — the SE3Quat / Eigen quaternion arithmetic shared by the restatements of the g2o-based paths
 * (orc_pnp.c, orc_lba.c): g2o/types/slam3d/se3quat.h:52-58,96-102,199-229 and Eigen 3.4.0 Geometry/Quaternion.h.
 * TEST INFRASTRUCTURE ONLY (see cml_oracle.h). */
#ifndef ORC_G2O_H
#define ORC_G2O_H
#include <math.h>

typedef struct { double x, y, z, w; double t[3]; } se3q;       /* Eigen coefficient order */

static inline void q_from_matrix(const double m[9], se3q* q) {        /* Quaternion.h quaternionbase_assign_impl<Other,3,3> */
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
static inline void q_normalize(se3q* q) {                             /* se3quat.h normalizeRotation */
    if (q->w < 0) { q->x *= -1; q->y *= -1; q->z *= -1; q->w *= -1; }
    double n = sqrt(q->x * q->x + q->y * q->y + q->z * q->z + q->w * q->w);
    q->x /= n; q->y /= n; q->z /= n; q->w /= n;
}
static inline void q_rotate(const se3q* q, const double v[3], double o[3]) {   /* QuaternionBase::_transformVector */
    double uv[3] = {q->y * v[2] - q->z * v[1], q->z * v[0] - q->x * v[2], q->x * v[1] - q->y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q->w * uv[0] + (q->y * uv[2] - q->z * uv[1]);
    o[1] = v[1] + q->w * uv[1] + (q->z * uv[0] - q->x * uv[2]);
    o[2] = v[2] + q->w * uv[2] + (q->x * uv[1] - q->y * uv[0]);
}
static inline void q_to_matrix(const se3q* q, double R[9]) {          /* QuaternionBase::toRotationMatrix */
    const double tx = 2 * q->x, ty = 2 * q->y, tz = 2 * q->z;
    const double twx = tx * q->w, twy = ty * q->w, twz = tz * q->w, txx = tx * q->x, txy = ty * q->x, txz = tz * q->x,
                 tyy = ty * q->y, tyz = tz * q->y, tzz = tz * q->z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static inline void se3q_from_Rt(const double R[9], const double t[3], se3q* T) {   /* SE3Quat(R, t) */
    q_from_matrix(R, T); q_normalize(T);
    T->t[0] = t[0]; T->t[1] = t[1]; T->t[2] = t[2];
}
static inline void se3q_exp(const double u[6], se3q* T) {             /* se3quat.h:201-229 */
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
static inline void se3q_mul(const se3q* A, const se3q* B, se3q* C) {  /* se3quat.h:96-102 */
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


static inline void orc_huber(double e, double delta, double rho[3]) {     /* g2o/core/robust_kernel_impl.cpp:60-74 */
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else { const double sq = sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; rho[2] = -0.5 * rho[1] / e; }
}
static inline void inv3(const double* A, double* o) {                          /* Matrix3d::inverse(): adjugate / determinant */
    const double c00 = A[4] * A[8] - A[5] * A[7], c10 = A[5] * A[6] - A[3] * A[8], c20 = A[3] * A[7] - A[4] * A[6];
    const double det = c00 * A[0] + c10 * A[1] + c20 * A[2];
    const double id = 1.0 / det;
    o[0] = c00 * id; o[3] = c10 * id; o[6] = c20 * id;
    o[1] = (A[2] * A[7] - A[1] * A[8]) * id; o[4] = (A[0] * A[8] - A[2] * A[6]) * id; o[7] = (A[1] * A[6] - A[0] * A[7]) * id;
    o[2] = (A[1] * A[5] - A[2] * A[4]) * id; o[5] = (A[2] * A[3] - A[0] * A[5]) * id; o[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

static inline int chol_solve_dense(double* A, int n, const double* b, double* x) {     /* LLT, fails on a pivot <= 0 */
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0)) return 0;
        d = sqrt(d); A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * x[k]; x[i] = s / A[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * x[k]; x[i] = s / A[(size_t)i * n + i]; }
    return 1;
}

#endif
