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

/* ---- the Eigen arithmetic behind g2o's SE3Quat (thirdparty/g2o/g2o/types/slam3d/se3quat.h cannot be compiled here: it pulls in
 * the cmake-generated g2o/config.h).  The functions below make the SAME Eigen calls its statements make, so that the oracle's
 * plain-C versions (orc_g2o.h) are pinned on Eigen's Quaternion(Matrix3), quaternion product, operator*(Vector3),
 * normalize(), toRotationMatrix(), Matrix3d::inverse(), LDLT<Matrix3d> and LLT.  Quaternions are passed as x, y, z, w. */
static void put(const Eigen::Quaterniond& r, const Eigen::Vector3d& t, double q[4], double tt[3]) {
    q[0] = r.x(); q[1] = r.y(); q[2] = r.z(); q[3] = r.w();
    for (int i = 0; i < 3; i++) tt[i] = t[i];
}
static void normalize_rotation(Eigen::Quaterniond& r) {              /* se3quat.h normalizeRotation */
    if (r.w() < 0) r.coeffs() *= -1;
    r.normalize();
}
void ref_g2o_from_Rt(const double R[9], const double t[3], double q[4], double tt[3]) {      /* SE3Quat(R, t), :52-54 */
    Eigen::Matrix3d M;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M(i, j) = R[i * 3 + j];
    Eigen::Quaterniond r(M);
    normalize_rotation(r);
    put(r, Eigen::Vector3d(t[0], t[1], t[2]), q, tt);
}
void ref_g2o_exp(const double u[6], double q[4], double tt[3]) {                              /* SE3Quat::exp, :201-229 */
    Eigen::Vector3d omega(u[0], u[1], u[2]), upsilon(u[3], u[4], u[5]);
    double theta = omega.norm();
    Eigen::Matrix3d Omega;
    Omega << 0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0;
    Eigen::Matrix3d R, V, Omega2 = Omega * Omega;
    if (theta < 0.00001) {
        R = (Eigen::Matrix3d::Identity() + Omega + 0.5 * Omega2);
        V = (Eigen::Matrix3d::Identity() + 0.5 * Omega + 1. / 6. * Omega2);
    } else {
        R = (Eigen::Matrix3d::Identity() + std::sin(theta) / theta * Omega + (1 - std::cos(theta)) / (theta * theta) * Omega2);
        V = (Eigen::Matrix3d::Identity() + (1 - std::cos(theta)) / (theta * theta) * Omega + (theta - std::sin(theta)) / (std::pow(theta, 3)) * Omega2);
    }
    Eigen::Quaterniond r(R);
    normalize_rotation(r);
    put(r, V * upsilon, q, tt);
}
void ref_g2o_mul(const double qa[4], const double ta[3], const double qb[4], const double tb[3], double q[4], double tt[3]) {   /* operator*, :96-102 */
    Eigen::Quaterniond ra(qa[3], qa[0], qa[1], qa[2]), rb(qb[3], qb[0], qb[1], qb[2]);
    Eigen::Vector3d t(ta[0], ta[1], ta[2]);
    t += ra * Eigen::Vector3d(tb[0], tb[1], tb[2]);
    ra *= rb;
    normalize_rotation(ra);
    put(ra, t, q, tt);
}
void ref_g2o_map(const double q[4], const double t[3], const double X[3], double out[3]) {   /* map, :199 */
    Eigen::Quaterniond r(q[3], q[0], q[1], q[2]);
    Eigen::Vector3d p = r * Eigen::Vector3d(X[0], X[1], X[2]) + Eigen::Vector3d(t[0], t[1], t[2]);
    for (int i = 0; i < 3; i++) out[i] = p[i];
}
void ref_quat_to_matrix(const double q[4], double R[9]) {
    Eigen::Matrix3d M = Eigen::Quaterniond(q[3], q[0], q[1], q[2]).toRotationMatrix();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = M(i, j);
}
void ref_mat3_inverse(const double A[9], double Ai[9]) {                                     /* D->inverse(), block_solver.hpp:369 */
    Eigen::Matrix3d M;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M(i, j) = A[i * 3 + j];
    Eigen::Matrix3d I = M.inverse();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ai[i * 3 + j] = I(i, j);
}
int ref_ldlt3(const double A[9], const double b[3], double x[3]) {                           /* structure_only_solver.h:166-173 */
    Eigen::Matrix3d M;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M(i, j) = A[i * 3 + j];
    Eigen::LDLT<Eigen::Matrix3d> chol(M);
    Eigen::Vector3d s = chol.solve(Eigen::Vector3d(b[0], b[1], b[2]));
    for (int i = 0; i < 3; i++) x[i] = s[i];
    return chol.isPositive() ? 1 : 0;
}
int ref_llt_solve(const double* A, int n, const double* b, double* x) {                      /* the dense equivalent of SimplicialLLT */
    ColMat M = Eigen::Map<const RowMat>(A, n, n);
    Eigen::LLT<ColMat> llt(M);
    if (llt.info() != Eigen::Success) return 0;
    Vec s = llt.solve(Eigen::Map<const Vec>(b, n));
    for (int i = 0; i < n; i++) x[i] = s[i];
    return 1;
}

/* ---- the Eigen expression SHAPES of the projection arithmetic on the path, evaluated by the vendored Eigen exactly as the
 * reference's statements are typed (same scalar types, same operand expressions), non-FMA build.  Eigen's evaluation order is
 * not the textbook left-to-right sum: e.g. a float 3x3 * 3-vector is e0 + (e1 + e2) per row, a double one is (e0 + e1) + e2 in
 * the SSE2 packet rows 0-1 and e0 + (e1 + e2) in the scalar row 2.  The oracle's orc_eig_* helpers are pinned on these. */
typedef Eigen::Matrix<float, 3, 3> M3f;
typedef Eigen::Matrix<double, 3, 3> M3d;
static M3f ldf(const float* a) { M3f m; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m(i, j) = a[i * 3 + j]; return m; }
static M3d ldd(const double* a) { M3d m; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m(i, j) = a[i * 3 + j]; return m; }
/* Vector3f pt = RKi * Vector3f(x, y, 1) + t*id;  (TR.cpp:306,316)   sign < 0: ... - t*id (TR.cpp:323,330) */
void ref_eig_matvec3f_affine(const float M[9], const float v[3], const float t[3], float s, int sign, float out[3]) {
    M3f RKi = ldf(M); Eigen::Vector3f tt(t[0], t[1], t[2]);
    Eigen::Vector3f pt;
    if (sign >= 0) pt = RKi * Eigen::Vector3f(v[0], v[1], v[2]) + tt * s;
    else pt = RKi * Eigen::Vector3f(v[0], v[1], v[2]) - tt * s;
    for (int i = 0; i < 3; i++) out[i] = pt[i];
}
/* tempPt.setZero(); tempPt.noalias() += RKi * pPattern; tempPt.noalias() += t * idepth_new;  (DSOInitializer.cpp:490-493) */
void ref_eig_matvec3f_noalias(const float M[9], const float v[3], const float t[3], float s, float out[3]) {
    M3f RKi = ldf(M); Eigen::Vector3f tt(t[0], t[1], t[2]), p(v[0], v[1], v[2]), tempPt;
    tempPt.setZero();
    tempPt.noalias() += RKi * p;
    tempPt.noalias() += tt * s;
    for (int i = 0; i < 3; i++) out[i] = tempPt[i];
}
/* projectedcurp.noalias() = R * refcorner.homogeneous() + t * pointIdepth;  (BA.cpp:109, DSOTracer.cpp:436) */
void ref_eig_homog3d(const double R[9], const double v[2], const double t[3], double s, double out[3]) {
    M3d Rm = ldd(R); Eigen::Vector2d refcorner(v[0], v[1]); Eigen::Vector3d tt(t[0], t[1], t[2]), projectedcurp;
    projectedcurp.noalias() = Rm * refcorner.homogeneous() + tt * s;
    for (int i = 0; i < 3; i++) out[i] = projectedcurp[i];
}
/* Vector3 pr = hostToFrame_KRKi * Vector3(x, y, 1);  (DSOTracer.cpp:608) */
void ref_eig_matvec3d(const double M[9], const double v[3], double out[3]) {
    M3d A = ldd(M);
    Eigen::Vector3d pr = A * Eigen::Vector3d(v[0], v[1], v[2]);
    for (int i = 0; i < 3; i++) out[i] = pr[i];
}
/* Matrix33f RKi = (R.cast<float>() * Ki);  (TR.cpp:270) */
void ref_eig_matmul3f(const float A[9], const float B[9], float out[9]) {
    M3f C = ldf(A) * ldf(B);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out[i * 3 + j] = C(i, j);
}
/* Matrix33f RKi = (R * Ki).cast<float>();  (DSOInitializer.cpp:471) and the plain double product */
void ref_eig_matmul3d(const double A[9], const double B[9], double out[9], float outf[9]) {
    M3d C = ldd(A) * ldd(B);
    M3f Cf = (ldd(A) * ldd(B)).cast<float>();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { out[i * 3 + j] = C(i, j); outf[i * 3 + j] = Cf(i, j); }
}
/* Matrix33f Ki = K.inverse() (TR.cpp:260-261) / Matrix33 Ki = K.inverse() (DSOInitializer.cpp:456) */
void ref_eig_inverse3f(const float A[9], float out[9]) {
    M3f I = ldf(A).inverse();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out[i * 3 + j] = I(i, j);
}
void ref_eig_inverse3d(const double A[9], double out[9]) {
    M3d I = ldd(A).inverse();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out[i * 3 + j] = I(i, j);
}
/* Matrix33 hostToFrame_KRKi = K * R * K.inverse();  Vector3 hostToFrame_Kt = K * t;  (DSOTracer.cpp:606-607) */
void ref_eig_krki(const double K[9], const double R[9], const double t[3], double out[9], double kt[3]) {
    M3d Km = ldd(K), Rm = ldd(R);
    M3d hostToFrame_KRKi = Km * Rm * Km.inverse();
    Eigen::Vector3d hostToFrame_Kt = Km * Eigen::Vector3d(t[0], t[1], t[2]);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) out[i * 3 + j] = hostToFrame_KRKi(i, j); kt[i] = hostToFrame_Kt[i]; }
}

/* Jp*delta of fixLinearization (BA.cpp:2219-2220) and of the LINEARIZED accumulation (BA.cpp:1699-1700):
 *   J.Jpdxi[0].dot(dp.head<6>()) + J.Jpdc[0].dot(mCDeltaF.cast<float>()) + J.Jpdd[0] * deltaF      (cast_in_dot != 0)
 *   rJ.Jpdxi[0].dot(dp.head<6>()) + rJ.Jpdc[0].dot(dc) + rJ.Jpdd[0] * dd,  Vector4f dc = mCDeltaF.cast<float>()  (cast_in_dot == 0) */
float ref_eig_jp_delta(const float Jpdxi[6], const float dp8[8], const float Jpdc[4], const double cdelta[4], float Jpdd, float deltaF, int cast_in_dot) {
    Eigen::Matrix<float, 6, 1> Jx; Eigen::Matrix<float, 8, 1> dp; Eigen::Matrix<float, 4, 1> Jc; Eigen::Matrix<double, 4, 1> mCDeltaF;
    for (int i = 0; i < 6; i++) Jx[i] = Jpdxi[i];
    for (int i = 0; i < 8; i++) dp[i] = dp8[i];
    for (int i = 0; i < 4; i++) { Jc[i] = Jpdc[i]; mCDeltaF[i] = cdelta[i]; }
    if (cast_in_dot) return Jx.dot(dp.head<6>()) + Jc.dot(mCDeltaF.cast<float>()) + Jpdd * deltaF;
    Eigen::Matrix<float, 4, 1> dc = mCDeltaF.cast<float>();
    return Jx.dot(dp.head<6>()) + Jc.dot(dc) + Jpdd * deltaF;
}
/* b -= mCalibStep.dot(Hcd_accAF.cast<scalar_t>() + Hcd_accLF.cast<scalar_t>())  (BA.cpp:1470) */
double ref_eig_calib_dot(const double step[4], const float a[4], const float l[4]) {
    Eigen::Matrix<double, 4, 1> mCalibStep; Eigen::Matrix<float, 4, 1> A, L;
    for (int i = 0; i < 4; i++) { mCalibStep[i] = step[i]; A[i] = a[i]; L[i] = l[i]; }
    return mCalibStep.dot(A.cast<double>() + L.cast<double>());
}

}  /* extern "C" */
