// se3.h — minimal fp64 SE(3) / exposure algebra for the host mirror (product code, independent of oracle/).
// Semantics follow what the reference host code gets from Sophus 1.1.0 (quaternion storage, (upsilon, omega)
// tangent order, atan-based log) — see thirdparty/Sophus/sophus/se3.hpp:104-112,224-257,776-797 and
// so3.hpp:248-294,599-635 in the reference tree — and src/cml/map/Exposure.h:119-123.
#pragma once
#include <cmath>
#include <cstring>

// the device layer steps the frame states with the same code (csrc/ba_frames.hip): one definition, host and gfx950
#ifdef __HIPCC__
#define CML_HD __host__ __device__
#else
#define CML_HD
#endif

namespace cml_amd {

struct SE3 {
    double q[4] = {1, 0, 0, 0};   // w x y z
    double t[3] = {0, 0, 0};

    CML_HD static void hat(const double w[3], double O[9]) {
        O[0] = 0; O[1] = -w[2]; O[2] = w[1]; O[3] = w[2]; O[4] = 0; O[5] = -w[0]; O[6] = -w[1]; O[7] = w[0]; O[8] = 0;
    }
    CML_HD static void mm(const double A[9], const double B[9], double C[9]) {
        double r[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
        for (int i = 0; i < 9; i++) C[i] = r[i];
    }
    CML_HD static void mv(const double A[9], const double v[3], double o[3]) {
        double r[3];
        for (int i = 0; i < 3; i++) r[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
        o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
    }
    CML_HD void matrix(double R[9]) const {
        const double w = q[0], x = q[1], y = q[2], z = q[3];
        const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
        const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
        R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
        R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
        R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
    }
    CML_HD static SE3 fromRt(const double R[9], const double tt[3]) {
        SE3 T;
        double tr = R[0] + R[4] + R[8];
        if (tr > 0) {
            double s = std::sqrt(tr + 1.0);
            T.q[0] = 0.5 * s; s = 0.5 / s;
            T.q[1] = (R[7] - R[5]) * s; T.q[2] = (R[2] - R[6]) * s; T.q[3] = (R[3] - R[1]) * s;
        } else {
            int i = 0;
            if (R[4] > R[0]) i = 1;
            if (R[8] > R[i * 3 + i]) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            double s = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
            T.q[1 + i] = 0.5 * s; s = 0.5 / s;
            T.q[0] = (R[k * 3 + j] - R[j * 3 + k]) * s;
            T.q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
            T.q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
        }
        const double n = std::sqrt(T.q[0] * T.q[0] + T.q[1] * T.q[1] + T.q[2] * T.q[2] + T.q[3] * T.q[3]);
        for (int i = 0; i < 4; i++) T.q[i] /= n;
        for (int i = 0; i < 3; i++) T.t[i] = tt[i];
        return T;
    }
    CML_HD static SE3 exp(const double xi[6]) {
        const double eps = 1e-10;
        const double* om = xi + 3;
        const double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
        double theta, imag, real;
        if (th2 < eps * eps) {
            theta = 0;
            const double p4 = th2 * th2;
            imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * p4;
            real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * p4;
        } else {
            theta = std::sqrt(th2);
#ifdef __HIP_DEVICE_COMPILE__
            double sh_, ch_;                          // (device: one argument reduction for the pair — the frame step is a single-lane chain
            ::sincos(0.5 * theta, &sh_, &ch_);         //  on the workgroup that ends the solve launch)
            imag = sh_ / theta; real = ch_;
#else
            imag = std::sin(0.5 * theta) / theta;
            real = std::cos(0.5 * theta);
#endif
        }
        SE3 T;
        T.q[0] = real; T.q[1] = imag * om[0]; T.q[2] = imag * om[1]; T.q[3] = imag * om[2];
        double O[9], O2[9], V[9];
        hat(om, O);
        mm(O, O, O2);
        if (theta < eps) T.matrix(V);
        else {
#ifdef __HIP_DEVICE_COMPILE__
            double st_, ct_;
            ::sincos(theta, &st_, &ct_);
            const double a = (1.0 - ct_) / (theta * theta), b = (theta - st_) / (theta * theta * theta);
#else
            const double a = (1.0 - std::cos(theta)) / (theta * theta), b = (theta - std::sin(theta)) / (theta * theta * theta);
#endif
            for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * O[i] + b * O2[i];
        }
        mv(V, xi, T.t);
        return T;
    }
    CML_HD void log(double xi[6]) const {
        const double eps = 1e-10;
        const double sq = q[1] * q[1] + q[2] * q[2] + q[3] * q[3], w = q[0];
        double f, theta;
        if (sq < eps * eps) {
            f = 2.0 / w - (2.0 / 3.0) * sq / (w * w * w);
            theta = 2.0 * sq / w;
        } else {
            const double n = std::sqrt(sq);
            const double at = (w < 0) ? std::atan2(-n, -w) : std::atan2(n, w);
            f = 2.0 * at / n;
            theta = f * n;
        }
        const double om[3] = {f * q[1], f * q[2], f * q[3]};
        double O[9], O2[9], Vi[9];
        hat(om, O);
        mm(O, O, O2);
        double c;
        if (std::fabs(theta) < eps) c = 1.0 / 12.0;
        else { const double ht = 0.5 * theta; c = (1.0 - theta * std::cos(ht) / (2.0 * std::sin(ht))) / (theta * theta); }
        for (int i = 0; i < 9; i++) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * O[i] + c * O2[i];
        mv(Vi, t, xi);
        xi[3] = om[0]; xi[4] = om[1]; xi[5] = om[2];
    }
    CML_HD SE3 operator*(const SE3& B) const {
        SE3 r;
        const double* a = q; const double* b = B.q;
        r.q[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
        r.q[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
        r.q[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
        r.q[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
        const double sn = r.q[0] * r.q[0] + r.q[1] * r.q[1] + r.q[2] * r.q[2] + r.q[3] * r.q[3];
        if (sn != 1.0) { const double s = 2.0 / (1.0 + sn); for (int i = 0; i < 4; i++) r.q[i] *= s; }
        double R[9], v[3];
        matrix(R);
        mv(R, B.t, v);
        for (int i = 0; i < 3; i++) r.t[i] = t[i] + v[i];
        return r;
    }
    CML_HD SE3 inverse() const {
        SE3 r;
        r.q[0] = q[0]; r.q[1] = -q[1]; r.q[2] = -q[2]; r.q[3] = -q[3];
        double R[9], v[3];
        r.matrix(R);
        mv(R, t, v);
        r.t[0] = -v[0]; r.t[1] = -v[1]; r.t[2] = -v[2];
        return r;
    }
    CML_HD void Adj(double A[36]) const {
        double R[9], H[9], HR[9];
        matrix(R);
        hat(t, H);
        mm(H, R, HR);
        for (int i = 0; i < 36; i++) A[i] = 0.0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { A[i * 6 + j] = R[i * 3 + j]; A[(i + 3) * 6 + j + 3] = R[i * 3 + j]; A[i * 6 + j + 3] = HR[i * 3 + j]; }
    }
};

// src/cml/map/Exposure.h:119-123
struct Exposure {
    double a = 0, b = 0, t = 1;
    CML_HD Exposure() {}
    CML_HD Exposure(double t_, double a_, double b_) : a(a_), b(b_), t(t_) {}
    CML_HD void to(const Exposure& o, double& A, double& B) const { A = std::exp(o.a - a) * o.t / t; B = o.b - A * b; }
};

}  // namespace cml_amd
