/*
 * orc_base.c — oracle: images, SE3 algebra, small dense algebra.
 * TEST INFRASTRUCTURE ONLY (see cml_oracle.h).  Plain C restatement; every function
 * cites the reference file:line it follows.
 */
#include "cml_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ============================================================ images */

/* src/cml/capture/CaptureImage.cpp:39-78: s = (w,h) as doubles, halved each level,
 * truncated to int on push; stop when area <= 25*25 and already >= 5 levels. */
int orc_pyramid_sizes(int w, int h, int* ws, int* hs, int max_levels) {
    double sx = w, sy = h;
    int n = 0;
    while (n < max_levels) {
        double area = sx * sy;
        if (area <= 25.0 * 25.0 && n >= 5) break;
        ws[n] = (int)sx;
        hs[n] = (int)sy;
        n++;
        sx = sx / 2.0;
        sy = sy / 2.0;
    }
    return n;
}

/* src/cml/image/Array2D.h:388-401 */
void orc_reduce_by_two(const float* in, int w, int h, float* out) {
    int nw = w / 2, nh = h / 2;
    for (int y = 0; y < nh; y++)
        for (int x = 0; x < nw; x++) {
            float a = in[(2 * y) * w + 2 * x];
            float b = in[(2 * y) * w + 2 * x + 1];
            float c = in[(2 * y + 1) * w + 2 * x];
            float d = in[(2 * y + 1) * w + 2 * x + 1];
            out[y * nw + x] = (((a + b) + c) + d) / 4.0f;
        }
}

/* src/cml/image/Array2D.h:288-327 */
void orc_gradient_image(const float* g, int w, int h, float* o) {
    memset(o, 0, sizeof(float) * 3 * (size_t)w * h);
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            float* p = o + 3 * ((size_t)y * w + x);
            p[0] = g[y * w + x];
            p[1] = (g[y * w + x + 1] - g[y * w + x - 1]) * 0.5f;
            p[2] = (g[(y + 1) * w + x] - g[(y - 1) * w + x]) * 0.5f;
        }
}

/* src/cml/image/Array2D.h:265-286 (no bounds check, truncating (int) cast) */
void orc_interpolate3(const float* d, int w, float x, float y, float out[3]) {
    const int ix = (int)x, iy = (int)y;
    const float dx = x - (float)ix, dy = y - (float)iy;
    const float dxdy = dx * dy;
    const float w00 = 1 - dx - dy + dxdy, w01 = dx - dxdy, w10 = dy - dxdy, w11 = dxdy;
    const float* p1 = d + 3 * ((size_t)iy * w + ix);
    const float* p2 = p1 + 3 * (size_t)w;
    for (int c = 0; c < 3; c++)
        out[c] = p1[c] * w00 + p1[3 + c] * w01 + p2[c] * w10 + p2[3 + c] * w11;
}


/* ============================================================ Eigen expression shapes (pinned: tests/golden, oracle/_ref)
 * How the vendored Eigen 3.4.0 (non-FMA build) evaluates the small fixed-size products the path is written with.  The inner
 * sum of three products is NOT left-to-right everywhere: scalar (non-vectorisable) evaluation reduces as e0 + (e1 + e2)
 * (redux unroller splits 3 into 1 + 2), SSE2 double packets (rows 0-1 of a column-major 3-vector) accumulate column by column,
 * (e0 + e1) + e2; the homogeneous product adds the last column after the 2-column product. */
void orc_eig_matvec3f_affine(const float M[9], const float v[3], const float t[3], float s, int sign, float out[3]) {
    for (int i = 0; i < 3; i++) {                                   /* Vector3f pt = RKi * Vector3f(x, y, 1) +/- t*id, TR.cpp:306-330 */
        const float r = M[3 * i] * v[0] + (M[3 * i + 1] * v[1] + M[3 * i + 2] * v[2]);
        out[i] = sign >= 0 ? r + t[i] * s : r - t[i] * s;
    }
}
void orc_eig_matvec3f_noalias(const float M[9], const float v[3], const float t[3], float s, float out[3]) {
    for (int i = 0; i < 3; i++) {                                   /* setZero; noalias() += RKi * p; noalias() += t * id, DSOInitializer.cpp:490-493 */
        const float r = 0.0f + (M[3 * i] * v[0] + (M[3 * i + 1] * v[1] + M[3 * i + 2] * v[2]));
        out[i] = r + t[i] * s;
    }
}
void orc_eig_homog3d(const double R[9], const double v[2], const double t[3], double s, double out[3]) {
    for (int i = 0; i < 3; i++)                                     /* R * refcorner.homogeneous() + t * idepth, BA.cpp:109 */
        out[i] = ((R[3 * i] * v[0] + R[3 * i + 1] * v[1]) + R[3 * i + 2]) + t[i] * s;
}
void orc_eig_matvec3d(const double M[9], const double v[3], double out[3]) {
    for (int i = 0; i < 2; i++) out[i] = (M[3 * i] * v[0] + M[3 * i + 1] * v[1]) + M[3 * i + 2] * v[2];   /* packet rows */
    out[2] = M[6] * v[0] + (M[7] * v[1] + M[8] * v[2]);                                                    /* scalar row */
}
void orc_eig_matmul3f(const float A[9], const float B[9], float out[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[3 * i + j] = A[3 * i] * B[j] + (A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j]);
}
void orc_eig_matmul3d(const double A[9], const double B[9], double out[9]) {
    for (int j = 0; j < 3; j++) {
        for (int i = 0; i < 2; i++) out[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
        out[6 + j] = A[6] * B[j] + (A[7] * B[3 + j] + A[8] * B[6 + j]);
    }
}

/* Matrix3f / Matrix3d::inverse(), Eigen/src/LU/InverseImpl.h:139-176: cofactors, det = sum of cofactors_col0 .* col(0)
 * (float: e0 + (e1 + e2); double: the 2-packet first, (e0 + e1) + e2), result(j,i) = cofactor<i,j> * invdet */
#define EIG_COF(m, i, j) (m[(((i) + 1) % 3) * 3 + (((j) + 1) % 3)] * m[(((i) + 2) % 3) * 3 + (((j) + 2) % 3)] - m[(((i) + 1) % 3) * 3 + (((j) + 2) % 3)] * m[(((i) + 2) % 3) * 3 + (((j) + 1) % 3)])
void orc_eig_inverse3f(const float m[9], float o[9]) {
    const float c0 = EIG_COF(m, 0, 0), c1 = EIG_COF(m, 1, 0), c2 = EIG_COF(m, 2, 0);
    const float det = c0 * m[0] + (c1 * m[3] + c2 * m[6]);
    const float invdet = 1.0f / det;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) if (j != 0) o[j * 3 + i] = EIG_COF(m, i, j) * invdet;
    o[0] = c0 * invdet; o[1] = c1 * invdet; o[2] = c2 * invdet;
}
void orc_eig_inverse3d(const double m[9], double o[9]) {
    const double c0 = EIG_COF(m, 0, 0), c1 = EIG_COF(m, 1, 0), c2 = EIG_COF(m, 2, 0);
    const double det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
    const double invdet = 1.0 / det;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) if (j != 0) o[j * 3 + i] = EIG_COF(m, i, j) * invdet;
    o[0] = c0 * invdet; o[1] = c1 * invdet; o[2] = c2 * invdet;
}
#undef EIG_COF

/* Jp*delta of the linearised residuals: Vector6f.dot(Vector8f.head<6>()) + Vector4f.dot(.) + Jpdd * dd.  Eigen's SSE
 * evaluation: the 6-dot is one packet product-sum plus a scalar tail, ((x0 + x2) + (x1 + x3)) + (x4 + x5); the 4-dot against a
 * Vector4f is (c0 + c2) + (c1 + c3) (BA.cpp:1699-1700, 2166-2172: Vector4f dc = mCDeltaF.cast<float>()), against the
 * un-evaluated cast expression it is not vectorised and reduces by halves, (c0 + c1) + (c2 + c3) (fixLinearization,
 * BA.cpp:2219-2220: J.Jpdc[0].dot(mCDeltaF.cast<float>())). */
float orc_eig_jp_delta(const float Jxi[6], const float dp[8], const float Jc[4], const double cdelta[4], float Jpdd, float dd, int cast_in_dot) {
    float x[6], c[4];
    for (int i = 0; i < 6; i++) x[i] = Jxi[i] * dp[i];
    for (int i = 0; i < 4; i++) c[i] = Jc[i] * (float)cdelta[i];
    const float d6 = ((x[0] + x[2]) + (x[1] + x[3])) + (x[4] + x[5]);
    const float d4 = cast_in_dot ? (c[0] + c[1]) + (c[2] + c[3]) : (c[0] + c[2]) + (c[1] + c[3]);
    return (d6 + d4) + Jpdd * dd;
}
/* mCalibStep.dot(Hcd_accAF.cast<double>() + Hcd_accLF.cast<double>()), BA.cpp:1470: cast expression, reduced by halves */
double orc_eig_calib_dot(const double step[4], const float a[4], const float l[4]) {
    double z[4];
    for (int i = 0; i < 4; i++) z[i] = step[i] * ((double)a[i] + (double)l[i]);
    return (z[0] + z[1]) + (z[2] + z[3]);
}
/* Matrix<double,1,8> * Vector8f.cast<double>(), BA.cpp:1478: reduced by halves */
double orc_eig_row8_dot_cast(const double xa[8], const float J[8]) {
    double z[8];
    for (int i = 0; i < 8; i++) z[i] = xa[i] * (double)J[i];
    return ((z[0] + z[1]) + (z[2] + z[3])) + ((z[4] + z[5]) + (z[6] + z[7]));
}

/* ============================================================ SE3 */

static void q_normalize_approx(double q[4]) {
    /* sophus/so3.hpp SO3 product: first-order renormalisation after concatenation */
    double sn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (sn != 1.0) {
        double s = 2.0 / (1.0 + sn);
        for (int i = 0; i < 4; i++) q[i] *= s;
    }
}
static void q_mul(const double a[4], const double b[4], double c[4]) {
    /* Eigen quaternion product, (w,x,y,z) */
    double r[4];
    r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    r[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    r[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
    memcpy(c, r, sizeof r);
}
static void q_to_R(const double q[4], double R[9]) {
    /* Eigen::QuaternionBase::toRotationMatrix */
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
static void R_to_q(const double R[9], double q[4]) {
    /* Eigen quaternion from rotation matrix (Shoemake) */
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t;
        q[2] = (R[2] - R[6]) * t;
        q[3] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}
static void mat3_vec(const double R[9], const double v[3], double o[3]) {
    double r[3];
    for (int i = 0; i < 3; i++) r[i] = R[i * 3] * v[0] + R[i * 3 + 1] * v[1] + R[i * 3 + 2] * v[2];
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}
static void hat(const double w[3], double O[9]) {
    O[0] = 0;     O[1] = -w[2]; O[2] = w[1];
    O[3] = w[2];  O[4] = 0;     O[5] = -w[0];
    O[6] = -w[1]; O[7] = w[0];  O[8] = 0;
}
static void mat3_mul(const double A[9], const double B[9], double C[9]) {
    double r[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            r[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    memcpy(C, r, sizeof r);
}

void orc_se3_identity(orc_se3* T) {
    T->q[0] = 1; T->q[1] = T->q[2] = T->q[3] = 0;
    T->t[0] = T->t[1] = T->t[2] = 0;
}
void orc_se3_from_Rt(const double R[9], const double t[3], orc_se3* T) {
    R_to_q(R, T->q);
    double n = sqrt(T->q[0] * T->q[0] + T->q[1] * T->q[1] + T->q[2] * T->q[2] + T->q[3] * T->q[3]);
    for (int i = 0; i < 4; i++) T->q[i] /= n;
    memcpy(T->t, t, 3 * sizeof(double));
}
void orc_se3_matrix(const orc_se3* T, double R[9]) { q_to_R(T->q, R); }

/* sophus/so3.hpp:599-635 + se3.hpp:776-797 */
void orc_se3_exp(const double xi[6], orc_se3* T) {
    const double eps = 1e-10;
    const double* om = xi + 3;
    double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    double theta, imag, real;
    if (theta_sq < eps * eps) {
        theta = 0;
        double p4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * p4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * p4;
    } else {
        theta = sqrt(theta_sq);
        double ht = 0.5 * theta;
        imag = sin(ht) / theta;
        real = cos(ht);
    }
    T->q[0] = real; T->q[1] = imag * om[0]; T->q[2] = imag * om[1]; T->q[3] = imag * om[2];
    double O[9], O2[9], V[9];
    hat(om, O);
    mat3_mul(O, O, O2);
    if (theta < eps) {
        q_to_R(T->q, V);
    } else {
        double tsq = theta * theta;
        double a = (1.0 - cos(theta)) / tsq, b = (theta - sin(theta)) / (tsq * theta);
        for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * O[i] + b * O2[i];
    }
    mat3_vec(V, xi, T->t);
}

/* sophus/so3.hpp:248-294 + se3.hpp:224-257 */
void orc_se3_log(const orc_se3* T, double xi[6]) {
    const double eps = 1e-10;
    double sq = T->q[1] * T->q[1] + T->q[2] * T->q[2] + T->q[3] * T->q[3];
    double w = T->q[0], f, theta;
    if (sq < eps * eps) {
        double w2 = w * w;
        f = 2.0 / w - (2.0 / 3.0) * sq / (w * w2);
        theta = 2.0 * sq / w;
    } else {
        double n = sqrt(sq);
        double at = (w < 0) ? atan2(-n, -w) : atan2(n, w);
        f = 2.0 * at / n;
        theta = f * n;
    }
    double om[3] = {f * T->q[1], f * T->q[2], f * T->q[3]};
    double O[9], O2[9], Vi[9];
    hat(om, O);
    mat3_mul(O, O, O2);
    if (fabs(theta) < eps) {
        for (int i = 0; i < 9; i++) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * O[i] + (1.0 / 12.0) * O2[i];
    } else {
        double ht = 0.5 * theta;
        double c = (1.0 - theta * cos(ht) / (2.0 * sin(ht))) / (theta * theta);
        for (int i = 0; i < 9; i++) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * O[i] + c * O2[i];
    }
    mat3_vec(Vi, T->t, xi);
    xi[3] = om[0]; xi[4] = om[1]; xi[5] = om[2];
}

void orc_se3_mul(const orc_se3* A, const orc_se3* B, orc_se3* C) {
    orc_se3 r;
    q_mul(A->q, B->q, r.q);
    q_normalize_approx(r.q);
    double RA[9], v[3];
    q_to_R(A->q, RA);
    mat3_vec(RA, B->t, v);
    for (int i = 0; i < 3; i++) r.t[i] = A->t[i] + v[i];
    *C = r;
}
void orc_se3_inv(const orc_se3* A, orc_se3* C) {
    orc_se3 r;
    r.q[0] = A->q[0]; r.q[1] = -A->q[1]; r.q[2] = -A->q[2]; r.q[3] = -A->q[3];
    double Ri[9], v[3];
    q_to_R(r.q, Ri);
    mat3_vec(Ri, A->t, v);
    r.t[0] = -v[0]; r.t[1] = -v[1]; r.t[2] = -v[2];
    *C = r;
}
/* sophus/se3.hpp:104-112 */
void orc_se3_adj(const orc_se3* T, double A[36]) {
    double R[9], H[9], HR[9];
    q_to_R(T->q, R);
    hat(T->t, H);
    mat3_mul(H, R, HR);
    memset(A, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            A[i * 6 + j] = R[i * 3 + j];
            A[(i + 3) * 6 + j + 3] = R[i * 3 + j];
            A[i * 6 + j + 3] = HR[i * 3 + j];
        }
}

/* d[qx qy qz qw tx ty tz]/d[upsilon omega] of exp (sophus/se3.hpp:558-740 is a generated
 * closed form; this is the same derivative written from the definitions
 * q = (cos(th/2), sin(th/2)/th * w), t = V(w) u, V = I + B W + C W^2). */
void orc_se3_dx_exp_x(const double xi[6], double J[42]) {
    const double* u = xi;
    const double* w = xi + 3;
    double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    memset(J, 0, 42 * sizeof(double));
    if (th2 < 1e-10) { /* sophus/se3.hpp:576-586 */
        J[0 * 6 + 3] = 0.5; J[1 * 6 + 4] = 0.5; J[2 * 6 + 5] = 0.5;
        J[4 * 6 + 0] = 1; J[5 * 6 + 1] = 1; J[6 * 6 + 2] = 1;
        double ux = 0.5 * u[0], uy = 0.5 * u[1], uz = 0.5 * u[2];
        J[4 * 6 + 4] = uz;  J[4 * 6 + 5] = -uy;
        J[5 * 6 + 3] = -uz; J[5 * 6 + 5] = ux;
        J[6 * 6 + 3] = uy;  J[6 * 6 + 4] = -ux;
        return;
    }
    double th = sqrt(th2), hth = 0.5 * th;
    double a = sin(hth) / th, c = cos(hth);
    double da = (0.5 * c - a) / th; /* da/dth */
    for (int j = 0; j < 3; j++) {
        for (int i = 0; i < 3; i++) J[i * 6 + 3 + j] = (i == j ? a : 0.0) + w[i] * w[j] * da / th;
        J[3 * 6 + 3 + j] = -0.5 * a * w[j];
    }
    double B = (1.0 - cos(th)) / th2, C = (th - sin(th)) / (th2 * th);
    double dB = (th * sin(th) - 2.0 * (1.0 - cos(th))) / (th2 * th);
    double dC = ((1.0 - cos(th)) * th - 3.0 * (th - sin(th))) / (th2 * th2);
    double W[9], W2[9], V[9];
    hat(w, W);
    mat3_mul(W, W, W2);
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + B * W[i] + C * W2[i];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) J[(4 + i) * 6 + j] = V[i * 3 + j];
    for (int j = 0; j < 3; j++) {
        double e[3] = {0, 0, 0}, G[9], GW[9], WG[9], dV[9], dt[3];
        e[j] = 1;
        hat(e, G);
        mat3_mul(G, W, GW);
        mat3_mul(W, G, WG);
        for (int i = 0; i < 9; i++)
            dV[i] = dB * (w[j] / th) * W[i] + B * G[i] + dC * (w[j] / th) * W2[i] + C * (GW[i] + WG[i]);
        mat3_vec(dV, u, dt);
        for (int i = 0; i < 3; i++) J[(4 + i) * 6 + 3 + j] = dt[i];
    }
}

/* src/cml/map/Exposure.h:119-123: this.to(other) */
void orc_exposure_to(double a_from, double b_from, double t_from, double a_to, double b_to, double t_to,
                     double* a, double* b) {
    *a = exp(a_to - a_from) * t_to / t_from;
    *b = b_to - (*a) * b_from;
}

/* ============================================================ dense */

/* Eigen/src/Cholesky/LDLT.h:300-396 (unblocked, lower, diagonal pivoting) and :560-600 (solve) */
int orc_ldlt_solve(const double* Ain, const double* b, int n, double* x) {
    double* A = (double*)malloc(sizeof(double) * n * n);
    double* temp = (double*)malloc(sizeof(double) * n);
    int* tr = (int*)malloc(sizeof(int) * n);
    memcpy(A, Ain, sizeof(double) * n * n);
#define M(i, j) A[(size_t)(i) * n + (j)]
    if (n == 1) {
        tr[0] = 0;
    } else
        for (int k = 0; k < n; k++) {
            int big = k;
            double best = fabs(M(k, k));
            for (int i = k + 1; i < n; i++)
                if (fabs(M(i, i)) > best) { best = fabs(M(i, i)); big = i; }
            tr[k] = big;
            if (k != big) {
                int s = n - big - 1;
                for (int j = 0; j < k; j++) { double t = M(k, j); M(k, j) = M(big, j); M(big, j) = t; }
                for (int i = 0; i < s; i++) {
                    double t = M(big + 1 + i, k); M(big + 1 + i, k) = M(big + 1 + i, big); M(big + 1 + i, big) = t;
                }
                { double t = M(k, k); M(k, k) = M(big, big); M(big, big) = t; }
                for (int i = k + 1; i < big; i++) { double t = M(i, k); M(i, k) = M(big, i); M(big, i) = t; }
            }
            int rs = n - k - 1;
            if (k > 0) {
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
            double akk = M(k, k);
            int valid = fabs(akk) > 0.0;
            if (k == 0 && !valid) {
                for (int j = 0; j < n; j++) tr[j] = j;
                break;
            }
            if (rs > 0 && valid)
                for (int i = 0; i < rs; i++) M(k + 1 + i, k) /= akk;
        }
    /* solve: dst = P b */
    for (int i = 0; i < n; i++) x[i] = b[i];
    for (int k = 0; k < n; k++)
        if (tr[k] != k) { double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
    for (int i = 0; i < n; i++) { /* L^-1 (unit lower) */
        double s = x[i];
        for (int j = 0; j < i; j++) s -= M(i, j) * x[j];
        x[i] = s;
    }
    const double tol = 2.2250738585072014e-308; /* numeric_limits<double>::min() */
    for (int i = 0; i < n; i++) {
        if (fabs(M(i, i)) > tol) x[i] /= M(i, i);
        else x[i] = 0;
    }
    for (int i = n - 1; i >= 0; i--) { /* L^-T */
        double s = x[i];
        for (int j = i + 1; j < n; j++) s -= M(j, i) * x[j];
        x[i] = s;
    }
    for (int k = n - 1; k >= 0; k--)
        if (tr[k] != k) { double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
#undef M
    free(A); free(temp); free(tr);
    for (int i = 0; i < n; i++)
        if (!isfinite(x[i])) return CMLHIP_ERR_NONFINITE;
    return 0;
}

/* general inverse: partial-pivot Gauss-Jordan (Eigen uses PartialPivLU for dynamic / n>4) */
int orc_inverse(const double* Ain, int n, double* Ai) {
    double* A = (double*)malloc(sizeof(double) * n * n);
    memcpy(A, Ain, sizeof(double) * n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ai[i * n + j] = (i == j);
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int i = k + 1; i < n; i++)
            if (fabs(A[i * n + k]) > fabs(A[p * n + k])) p = i;
        if (p != k)
            for (int j = 0; j < n; j++) {
                double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t;
                t = Ai[k * n + j]; Ai[k * n + j] = Ai[p * n + j]; Ai[p * n + j] = t;
            }
        double d = A[k * n + k];
        for (int j = 0; j < n; j++) { A[k * n + j] /= d; Ai[k * n + j] /= d; }
        for (int i = 0; i < n; i++)
            if (i != k) {
                double f = A[i * n + k];
                if (f != 0)
                    for (int j = 0; j < n; j++) { A[i * n + j] -= f * A[k * n + j]; Ai[i * n + j] -= f * Ai[k * n + j]; }
            }
    }
    free(A);
    return 0;
}

/* symmetric Jacobi eigen-decomposition, m <= 16: A = V diag(e) V^T */
static void jacobi_eig(double* A, int m, double* V) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) V[i * m + j] = (i == j);
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0;
        for (int i = 0; i < m; i++)
            for (int j = i + 1; j < m; j++) off += A[i * m + j] * A[i * m + j];
        if (off < 1e-300) break;
        for (int p = 0; p < m; p++)
            for (int q = p + 1; q < m; q++) {
                double apq = A[p * m + q];
                if (fabs(apq) < 1e-300) continue;
                double tau = (A[q * m + q] - A[p * m + p]) / (2 * apq);
                double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
                double c = 1 / sqrt(1 + t * t), s = t * c;
                for (int k = 0; k < m; k++) {
                    double akp = A[k * m + p], akq = A[k * m + q];
                    A[k * m + p] = c * akp - s * akq;
                    A[k * m + q] = s * akp + c * akq;
                }
                for (int k = 0; k < m; k++) {
                    double apk = A[p * m + k], aqk = A[q * m + k];
                    A[p * m + k] = c * apk - s * aqk;
                    A[q * m + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < m; k++) {
                    double vkp = V[k * m + p], vkq = V[k * m + q];
                    V[k * m + p] = c * vkp - s * vkq;
                    V[k * m + q] = s * vkp + c * vkq;
                }
            }
    }
}

/* BA.cpp:1196-1261: N = normalized nullspace columns; Npi = U S^+ V^T (singular values
 * <= delta*max dropped); b -= 0.5 (N Npi^T + (N Npi^T)^T) b.
 * With N = U S V^T:  N Npi^T = U S V^T V S^+ U^T = U_kept U_kept^T (symmetric), so the
 * projector is assembled from the eigen-decomposition of N^T N = V S^2 V^T. */
void orc_orthogonalize(double* b, int n, const double* Ncols, int m, double delta) {
    double* Nn = (double*)malloc(sizeof(double) * n * m);
    for (int j = 0; j < m; j++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += Ncols[j * n + i] * Ncols[j * n + i];
        s = sqrt(s);
        for (int i = 0; i < n; i++) Nn[j * n + i] = Ncols[j * n + i] / s;
    }
    double G[16 * 16], V[16 * 16];
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += Nn[i * n + k] * Nn[j * n + k];
            G[i * m + j] = s;
        }
    jacobi_eig(G, m, V);
    double smax = 0;
    for (int i = 0; i < m; i++) {
        double sv = G[i * m + i] > 0 ? sqrt(G[i * m + i]) : 0;
        if (sv > smax) smax = sv;
    }
    double* proj = (double*)calloc(n, sizeof(double));
    for (int e = 0; e < m; e++) {
        double sv = G[e * m + e] > 0 ? sqrt(G[e * m + e]) : 0;
        if (!(sv > delta * smax)) continue;
        /* u_e = N v_e / sv */
        double dot = 0;
        double* ue = (double*)malloc(sizeof(double) * n);
        for (int i = 0; i < n; i++) {
            double s = 0;
            for (int j = 0; j < m; j++) s += Nn[j * n + i] * V[j * m + e];
            ue[i] = s / sv;
            dot += ue[i] * b[i];
        }
        for (int i = 0; i < n; i++) proj[i] += ue[i] * dot;
        free(ue);
    }
    for (int i = 0; i < n; i++) b[i] -= proj[i];
    free(proj);
    free(Nn);
}
