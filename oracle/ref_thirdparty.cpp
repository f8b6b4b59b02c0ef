/*
 * ref_thirdparty.cpp — oracle/_ref: the reference's OWN vendored third-party arithmetic at the
 * hot-path boundary (Eigen 3.4.0, Sophus 1.1.0), compiled from the headers where they lie under
 * /root/reference/thirdparty (see oracle/Makefile, target _ref).  TEST INFRASTRUCTURE ONLY.
 *
 * This file contains no reference code, only calls.  It exists to pin the oracle's plain-C
 * restatements (orc_se3_*, orc_ldlt_solve, orc_inverse, orc_orthogonalize) and to generate
 * tests/golden/thirdparty_vectors.json (tests/golden/make_thirdparty_vectors.py).
 *
 * The reference's own translation units cannot be built here: every one includes the
 * cmake-generated cml/config.h (src/cml/config.h.in) and src/cml/types/OS.cpp needs Qt.
 */
#include <Eigen/Dense>
#include <sophus/se3.hpp>

using SE3 = Sophus::SE3<double>;
typedef Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> RowMat;
typedef Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic> ColMat;
typedef Eigen::Matrix<double, Eigen::Dynamic, 1> Vec;

extern "C" {

/* SE3::exp -> quaternion (w,x,y,z) + translation; DSOFrame.h:119 */
void ref_se3_exp(const double xi[6], double q[4], double t[3]) {
    Eigen::Matrix<double, 6, 1> v = Eigen::Map<const Eigen::Matrix<double, 6, 1>>(xi);
    SE3 T = SE3::exp(v);
    q[0] = T.unit_quaternion().w(); q[1] = T.unit_quaternion().x(); q[2] = T.unit_quaternion().y(); q[3] = T.unit_quaternion().z();
    for (int i = 0; i < 3; i++) t[i] = T.translation()[i];
}
static SE3 make(const double q[4], const double t[3]) {
    return SE3(Eigen::Quaterniond(q[0], q[1], q[2], q[3]), Eigen::Vector3d(t[0], t[1], t[2]));
}
void ref_se3_log(const double q[4], const double t[3], double xi[6]) {
    Eigen::Matrix<double, 6, 1> v = make(q, t).log();
    for (int i = 0; i < 6; i++) xi[i] = v[i];
}
void ref_se3_adj(const double q[4], const double t[3], double A[36]) {   /* BA.cpp:1077 */
    Eigen::Matrix<double, 6, 6> M = make(q, t).Adj();
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) A[i * 6 + j] = M(i, j);
}
void ref_se3_mul(const double qa[4], const double ta[3], const double qb[4], const double tb[3], double q[4], double t[3]) {
    SE3 T = make(qa, ta) * make(qb, tb);
    q[0] = T.unit_quaternion().w(); q[1] = T.unit_quaternion().x(); q[2] = T.unit_quaternion().y(); q[3] = T.unit_quaternion().z();
    for (int i = 0; i < 3; i++) t[i] = T.translation()[i];
}
void ref_se3_inv(const double qa[4], const double ta[3], double q[4], double t[3]) {
    SE3 T = make(qa, ta).inverse();
    q[0] = T.unit_quaternion().w(); q[1] = T.unit_quaternion().x(); q[2] = T.unit_quaternion().y(); q[3] = T.unit_quaternion().z();
    for (int i = 0; i < 3; i++) t[i] = T.translation()[i];
}
void ref_se3_matrix(const double q[4], double R[9]) {
    Eigen::Matrix3d M = Eigen::Quaterniond(q[0], q[1], q[2], q[3]).toRotationMatrix();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = M(i, j);
}
void ref_se3_from_Rt(const double R[9], const double t[3], double q[4]) {   /* SE3(R, t) ctor, TR.cpp:45 */
    Eigen::Matrix3d M;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M(i, j) = R[i * 3 + j];
    SE3 T(M, Eigen::Vector3d(t[0], t[1], t[2]));
    q[0] = T.unit_quaternion().w(); q[1] = T.unit_quaternion().x(); q[2] = T.unit_quaternion().y(); q[3] = T.unit_quaternion().z();
}
void ref_se3_dx_exp_x(const double xi[6], double J[42]) {   /* BA.cpp:2623 */
    Eigen::Matrix<double, 6, 1> v = Eigen::Map<const Eigen::Matrix<double, 6, 1>>(xi);
    Eigen::Matrix<double, 7, 6> D = SE3::Dx_exp_x(v);
    for (int i = 0; i < 7; i++) for (int j = 0; j < 6; j++) J[i * 6 + j] = D(i, j);
}
/* x = A.ldlt().solve(b), BA.cpp:1317-1319, TR.cpp:97 */
int ref_ldlt_solve(const double* A, const double* b, int n, double* x) {
    ColMat M = Eigen::Map<const RowMat>(A, n, n);
    Vec rhs = Eigen::Map<const Vec>(b, n);
    Vec s = M.ldlt().solve(rhs);
    for (int i = 0; i < n; i++) x[i] = s[i];
    return s.allFinite() ? 0 : 3;
}
/* A.inverse(), BA.cpp:534, TR.cpp:243 */
void ref_inverse(const double* A, int n, double* Ai) {
    ColMat M = Eigen::Map<const RowMat>(A, n, n);
    ColMat I = M.inverse();
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) Ai[i * n + j] = I(i, j);
}
/* the JacobiSVD pseudo-inverse projection written as BA.cpp:1217-1251 writes it */
void ref_orthogonalize(double* b, int n, const double* Ncols, int m, double delta) {
    ColMat N(n, m);
    for (int j = 0; j < m; j++) {
        Vec c = Eigen::Map<const Vec>(Ncols + (size_t)j * n, n);
        N.col(j) = c.normalized();
    }
    Eigen::JacobiSVD<ColMat> svd(N, Eigen::ComputeThinU | Eigen::ComputeThinV);
    Vec S = svd.singularValues();
    double maxSv = 0;
    for (int i = 0; i < S.size(); i++) if (S[i] > maxSv) maxSv = S[i];
    for (int i = 0; i < S.size(); i++) S[i] = (S[i] > delta * maxSv) ? 1.0 / S[i] : 0.0;
    ColMat Npi = svd.matrixU() * S.asDiagonal() * svd.matrixV().transpose();
    ColMat NNpiT = N * Npi.transpose();
    ColMat NNpiTS = 0.5 * (NNpiT + NNpiT.transpose());
    Vec bv = Eigen::Map<Vec>(b, n);
    bv -= NNpiTS * bv;
    for (int i = 0; i < n; i++) b[i] = bv[i];
}

}  /* extern "C" */
