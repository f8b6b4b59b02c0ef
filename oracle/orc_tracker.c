/*
 * orc_tracker.c — oracle: coarse tracker (TR.cpp = src/cml/optimization/dso/DSOTracker.cpp)
 * and the hybrid ORB reprojection term (BA.cpp:2574-2729, src/cml/optimization/Residual.h).
 * TEST INFRASTRUCTURE ONLY (see cml_oracle.h).
 */
#include "cml_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Eigen compute_inverse_size3 (cofactors * 1/det), float */
/* TR.cpp:248-492 */
void orc_tracker_eval(const float* aos3, int wl, int hl, const float* uvic, int n, int level,
                      const double Rd[9], const double td[3], const double Kd[4], const double aff[2], double b0d,
                      const cmlhip_tracker_params* prm, int want_hessian,
                      cmlhip_tracker_result* out, float* warped, int cap) {
    float E = 0;
    int numTermsInE = 0, numWarped = 0, numSaturated = 0, numRobust = 0;
    float K[9] = {(float)Kd[0], 0, (float)Kd[2], 0, (float)Kd[1], (float)Kd[3], 0, 0, 1}, Ki[9];
    orc_eig_inverse3f(K, Ki);                       /* Matrix33f Ki = K.inverse(), :260-261 */
    float fxl = K[0], fyl = K[4], cxl = K[2], cyl = K[5];
    float R[9], RKi[9], t[3];
    for (int i = 0; i < 9; i++) R[i] = (float)Rd[i];
    orc_eig_matmul3f(R, Ki, RKi);                   /* Matrix33f RKi = R.cast<float>() * Ki, :270 (Eigen order: e0 + (e1 + e2)) */
    for (int i = 0; i < 3; i++) t[i] = (float)td[i];
    float a0 = (float)aff[0], a1 = (float)aff[1];   /* affLL, :272 */
    float sT = 0, sRT = 0, sN = 0;
    float maxEnergy = (float)(2.0f * (double)prm->huber * (double)prm->cutoff - (double)prm->huber * (double)prm->huber);   /* :278 */
    int own = 0;
    if (!warped) { cap = n + 4; warped = (float*)malloc(sizeof(float) * 8 * (size_t)cap); own = 1; }
#define WP(row, i) warped[(size_t)(row) * cap + (i)]
    for (int i = 0; i < n; i++) {
        float x = uvic[4 * i], y = uvic[4 * i + 1], id = uvic[4 * i + 2], refColor = uvic[4 * i + 3];
        if (!isfinite(refColor)) continue;
        float pt[3];
        for (int k = 0; k < 3; k++) pt[k] = (RKi[k * 3] * x + (RKi[k * 3 + 1] * y + RKi[k * 3 + 2] * 1.0f)) + t[k] * id;   /* :306, Eigen: e0 + (e1 + e2) */
        float u = pt[0] / pt[2], v = pt[1] / pt[2];
        float Ku = fxl * u + cxl, Kv = fyl * v + cyl;
        float new_idepth = id / pt[2];
        if (level == 0 && i % 32 == 0) {                /* :313-344 */
            float a[3], b[3], c[3];
            for (int k = 0; k < 3; k++) {
                float kp = Ki[k * 3] * x + (Ki[k * 3 + 1] * y + Ki[k * 3 + 2] * 1.0f);     /* Eigen: e0 + (e1 + e2) */
                a[k] = kp + t[k] * id;
                b[k] = kp - t[k] * id;
                c[k] = (RKi[k * 3] * x + (RKi[k * 3 + 1] * y + RKi[k * 3 + 2] * 1.0f)) - t[k] * id;
            }
            float KuT = fxl * (a[0] / a[2]) + cxl, KvT = fyl * (a[1] / a[2]) + cyl;
            float KuT2 = fxl * (b[0] / b[2]) + cxl, KvT2 = fyl * (b[1] / b[2]) + cyl;
            float Ku3 = fxl * (c[0] / c[2]) + cxl, Kv3 = fyl * (c[1] / c[2]) + cyl;
            sT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
            sT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
            sRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
            sRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
            sN += 2;
        }
        if (!(Ku > 2 && Kv > 2 && Ku < wl - 3 && Kv < hl - 3 && new_idepth > 0)) continue;     /* :346 */
        float hit[3];
        orc_interpolate3(aos3, wl, Ku, Kv, hit);
        if (!(isfinite(hit[0]) && isfinite(hit[1]) && isfinite(hit[2]))) continue;
        float residual = hit[0] - (float)(a0 * refColor + a1);
        float hw = fabs((double)residual) < (double)prm->huber ? 1.0f : (float)((double)prm->huber / fabs((double)residual));
        if (fabs((double)residual) > (double)prm->cutoff) {
            E += maxEnergy; numTermsInE++; numSaturated++;
        } else {
            E += hw * residual * residual * (2 - hw);
            numTermsInE++;
            if (numWarped < cap) {
                WP(0, numWarped) = new_idepth; WP(1, numWarped) = u; WP(2, numWarped) = v;
                WP(3, numWarped) = hit[1]; WP(4, numWarped) = hit[2]; WP(5, numWarped) = residual;
                WP(6, numWarped) = hw; WP(7, numWarped) = refColor;
            }
            numWarped++;
        }
        if (fabs((double)residual) <= (double)prm->cutoff_base) numRobust++;
    }
    out->E = E; out->numTermsInE = numTermsInE; out->numSaturated = numSaturated; out->numRobust = numRobust;
    out->numWarped = numWarped;
    out->flow[0] = sT / (sN + 0.1f); out->flow[1] = 0; out->flow[2] = sRT / (sN + 0.1f);   /* :412-414 */
    int npad = numWarped;
    while (npad % 4 != 0) {                         /* :391-402 */
        if (npad < cap) for (int k = 0; k < 8; k++) WP(k, npad) = 0;
        npad++;
    }
    if (want_hessian) {                             /* computeHessian, :421-492 + Accumulator9 ACC.h:1006-1211 */
        static float S[45][4], S1k[45][4], S1m[45][4];
        memset(S, 0, sizeof S); memset(S1k, 0, sizeof S1k); memset(S1m, 0, sizeof S1m);
        float numIn1 = 0, numIn1k = 0;
        float fx = (float)Kd[0], fy = (float)Kd[1], b0 = (float)b0d, a = (float)aff[0];
        for (int i = 0; i < npad; i += 4) {
            for (int l = 0; l < 4; l++) {
                int j = i + l;
                float dx = WP(3, j) * fx, dy = WP(4, j) * fy, u = WP(1, j), v = WP(2, j), id = WP(0, j);
                float J[9];
                J[0] = id * dx;
                J[1] = id * dy;
                J[2] = 0.0f - (id * (u * dx + v * dy));
                J[3] = 0.0f - ((u * v * dx) + dy * (1.0f + v * v));
                J[4] = (u * v * dy) + (dx * (1.0f + u * u));
                J[5] = u * dy - v * dx;
                J[6] = a * (b0 - WP(7, j));
                J[7] = -1.0f;
                J[8] = WP(5, j);
                float wgt = WP(6, j);
                int idx = 0;
                for (int r = 0; r < 9; r++) {
                    float Jw = J[r] * wgt;
                    for (int c = r; c < 9; c++) { S[idx][l] += Jw * J[c]; idx++; }
                }
            }
            numIn1++;
            if (numIn1 > 1000) {                    /* shiftUp, ACC.h:1364-1390 */
                for (int k = 0; k < 45; k++) for (int l = 0; l < 4; l++) { S1k[k][l] += S[k][l]; S[k][l] = 0; }
                numIn1k += numIn1; numIn1 = 0;
            }
            if (numIn1k > 1000) {
                for (int k = 0; k < 45; k++) for (int l = 0; l < 4; l++) { S1m[k][l] += S1k[k][l]; S1k[k][l] = 0; }
                numIn1k = 0;
            }
        }
        for (int k = 0; k < 45; k++) for (int l = 0; l < 4; l++) { S1k[k][l] += S[k][l]; S1m[k][l] += S1k[k][l]; }
        int idx = 0;
        for (int r = 0; r < 9; r++)
            for (int c = r; c < 9; c++) {
                float d = ((S1m[idx][0] + S1m[idx][1]) + S1m[idx][2]) + S1m[idx][3];
                out->H9[r * 9 + c] = out->H9[c * 9 + r] = d;
                idx++;
            }
        const double sc[8] = {prm->scale_rot, prm->scale_rot, prm->scale_rot, prm->scale_trans, prm->scale_trans,
                              prm->scale_trans, prm->scale_a, prm->scale_b};   /* the lane/scale quirk, :477-488 */
        for (int r = 0; r < 8; r++) {
            for (int c = 0; c < 8; c++) out->H[r * 8 + c] = ((double)out->H9[r * 9 + c] / (double)npad) * sc[c] * sc[r];
            out->b[r] = ((double)out->H9[r * 9 + 8] / (double)npad) * sc[r];
        }
    }
#undef WP
    if (own) free(warped);
}

/* makeCoarseDepthL0 from the splat on, TR.cpp:550-719 */
void orc_tracker_make_coarse_depth(const double* pts, int n, int levels, const int* ws, const int* hs,
                                   const float* const* gray, float** lists, int* n_out) {
    float* idepth[8]; float* wsum[8]; float* wbak[8];
    for (int l = 0; l < levels; l++) {
        size_t sz = (size_t)ws[l] * hs[l];
        idepth[l] = (float*)calloc(sz, 4); wsum[l] = (float*)calloc(sz, 4); wbak[l] = (float*)calloc(sz, 4);
    }
    int w0 = ws[0], h0 = hs[0];
    for (int i = 0; i < n; i++) {
        double Ku = pts[4 * i], Kv = pts[4 * i + 1], nid = pts[4 * i + 2];
        float weight = (float)pts[4 * i + 3];
        int u = (int)(Ku + 0.5), v = (int)(Kv + 0.5);
        if (u < 0 || u >= w0) continue;
        if (v < 0 || v >= h0) continue;
        idepth[0][u + w0 * v] = (float)((double)idepth[0][u + w0 * v] + nid * (double)weight);
        wsum[0][u + w0 * v] += weight;
    }
    for (int l = 1; l < levels; l++) {
        int wl = ws[l], hl = hs[l], wm = ws[l - 1];
        for (int y = 0; y < hl; y++)
            for (int x = 0; x < wl; x++) {
                int b = 2 * x + 2 * y * wm;
                idepth[l][x + y * wl] = ((idepth[l - 1][b] + idepth[l - 1][b + 1]) + idepth[l - 1][b + wm]) + idepth[l - 1][b + wm + 1];
                wsum[l][x + y * wl] = ((wsum[l - 1][b] + wsum[l - 1][b + 1]) + wsum[l - 1][b + wm]) + wsum[l - 1][b + wm + 1];
            }
    }
    for (int l = 0; l < levels; l++) {
        int wl = ws[l], hl = hs[l], wh = wl * hl - wl, size = wl * hl;
        memcpy(wbak[l], wsum[l], sizeof(float) * size);
        int d[4];
        if (l < 2) { d[0] = 1 + wl; d[1] = -1 - wl; d[2] = wl - 1; d[3] = -wl + 1; }   /* :616-619 */
        else { d[0] = 1; d[1] = -1; d[2] = wl; d[3] = -wl; }                          /* :655-658 */
        for (int i = wl; i < wh; i++) {
            if (wbak[l][i] <= 0) {
                float sum = 0, num = 0, numn = 0;
                for (int k = 0; k < 4; k++) {
                    int j = i + d[k];
                    if (j >= 0 && j < size && wbak[l][j] > 0) { sum += idepth[l][j]; num += wbak[l][j]; numn++; }
                }
                if (numn > 0) { idepth[l][i] = sum / numn; wsum[l][i] = num / numn; }
            }
        }
    }
    for (int l = 0; l < levels; l++) {
        int wl = ws[l], hl = hs[l], cnt = 0;
        for (int y = 2; y < hl - 2; y++)
            for (int x = 2; x < wl - 2; x++) {
                int i = x + y * wl;
                if (wsum[l][i] > 0) {
                    idepth[l][i] /= wsum[l][i];
                    float id = idepth[l][i];
                    float col = gray[l][i];
                    if (!isfinite(col) || !(id > 0)) { idepth[l][i] = -1; continue; }
                    lists[l][4 * cnt] = (float)x; lists[l][4 * cnt + 1] = (float)y;
                    lists[l][4 * cnt + 2] = id; lists[l][4 * cnt + 3] = col;
                    cnt++;
                } else
                    idepth[l][i] = -1;
                wsum[l][i] = 1;
            }
        n_out[l] = cnt;
        free(idepth[l]); free(wsum[l]); free(wbak[l]);
    }
}

/* ------------------------------------------------------------------ hybrid ORB term */

/* Quaternion::logHati, src/cml/maths/Rotation.cpp:205-221; then normalised (Rotation.h:246-252) */
static void cml_quat_from_R(const double R[9], double q[4]) {
    q[0] = sqrt(fmax(0.0, 1.0 + R[0] + R[4] + R[8])) / 2.0;
    q[1] = sqrt(fmax(0.0, 1.0 + R[0] - R[4] - R[8])) / 2.0;
    q[2] = sqrt(fmax(0.0, 1.0 - R[0] + R[4] - R[8])) / 2.0;
    q[3] = sqrt(fmax(0.0, 1.0 - R[0] - R[4] + R[8])) / 2.0;
    q[1] = copysign(q[1], R[7] - R[5]);
    q[2] = copysign(q[2], R[2] - R[6]);
    q[3] = copysign(q[3], R[3] - R[1]);
    double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= nn;
}
/* Quaternion::hatExpDerivative, Rotation.cpp:223-290: d(hatExp)/dq_i with the constant 1 dropped */
static void cml_quat_dR(const double q[4], int i, double D[9]) {
    double a = q[0], b = q[1], c = q[2], d = q[3];
    double _2b2 = 0, _2c2 = 0, _2d2 = 0, _2bc = 0, _2ad = 0, _2bd = 0, _2ac = 0, _2cd = 0, _2ab = 0;
    switch (i) {
        case 0: _2ad = 2 * d; _2ac = 2 * c; _2ab = 2 * b; break;
        case 1: _2b2 = 4 * b; _2bc = 2 * c; _2bd = 2 * d; _2ab = 2 * a; break;
        case 2: _2c2 = 4 * c; _2bc = 2 * b; _2ac = 2 * a; _2cd = 2 * d; break;
        case 3: _2d2 = 4 * d; _2ad = 2 * a; _2bd = 2 * b; _2cd = 2 * c; break;
    }
    D[0] = -_2c2 - _2d2; D[1] = _2bc - _2ad;   D[2] = _2bd + _2ac;
    D[3] = _2bc + _2ad;  D[4] = -_2b2 - _2d2;  D[5] = _2cd - _2ab;
    D[6] = _2bd - _2ac;  D[7] = _2cd + _2ab;   D[8] = -_2b2 - _2c2;
}
static double tukey(double v, double th) {            /* maths/Derivative.h:35-39 */
    if (fabs(v) > th) return 0;
    double l = 1.0 - (v * v) / (th * th);
    return v * l * l;
}
static double d_tukey(double v, double d, double th) { /* maths/Derivative.h:108-113 (literal: v is the LOSS value) */
    if (fabs(v) > th) return 0;
    double v2 = v * v, o = 1.0 - v2;
    return d * (-4.0 * v2 * o + o * o);
}

/* ReprojectionError::jacobian, src/cml/optimization/Residual.h:59-100 with the Camera derivative helpers
 * src/cml/map/Camera.h:317-386 (literal, including d/dt_i = R e_i and d/dq_i = R'(q)_i (P + t)). */
int orc_reproj_jacobian(const double R[9], const double t[3], const double X[3], double gx, double gy,
                        double fx, double fy, double* residual, double Jt[3], double Jq[4], double Jp[3]) {
    double T[3];
    for (int i = 0; i < 3; i++) T[i] = (R[i * 3] * X[0] + R[i * 3 + 1] * X[1] + R[i * 3 + 2] * X[2]) + t[i];
    double dx = T[0] / T[2] - gx, dy = T[1] / T[2] - gy;
    double sq = dx * dx + dy * dy, norm = sqrt(sq);
    double th = 3.0 / sqrt(fx * fx + fy * fy);
    *residual = tukey(norm, th);
    for (int i = 0; i < 3; i++) {
        double d[3] = {R[i], R[3 + i], R[6 + i]};                 /* R * e_i */
        double hx = (d[0] * T[2] - T[0] * d[2]) / (T[2] * T[2]);  /* Derivative::hnormalized, Derivative.h:45-47 */
        double hy = (d[1] * T[2] - T[1] * d[2]) / (T[2] * T[2]);
        double dsq = 2.0 * hx * dx + 2.0 * hy * dy;               /* Derivative::squaredNorm */
        double v = (norm == 0) ? 0 : dsq / (2.0 * norm);          /* Derivative::sqrt */
        v = d_tukey(*residual, v, th);
        if (!isfinite(v)) return 0;
        Jt[i] = v; Jp[i] = -v;
    }
    double q[4];
    cml_quat_from_R(R, q);
    double Pt[3] = {X[0] + t[0], X[1] + t[1], X[2] + t[2]};
    for (int i = 0; i < 4; i++) {
        double D[9], d[3];
        cml_quat_dR(q, i, D);
        for (int k = 0; k < 3; k++) d[k] = D[k * 3] * Pt[0] + D[k * 3 + 1] * Pt[1] + D[k * 3 + 2] * Pt[2];
        double hx = (d[0] * T[2] - T[0] * d[2]) / (T[2] * T[2]);
        double hy = (d[1] * T[2] - T[1] * d[2]) / (T[2] * T[2]);
        double dsq = 2.0 * hx * dx + 2.0 * hy * dy;
        double v = (norm == 0) ? 0 : dsq / (2.0 * norm);
        v = d_tukey(*residual, v, th);
        Jq[i] = v;
        if (!isfinite(v)) return 0;
    }
    return 1;
}

/* addIndirectToProblem, BA.cpp:2607-2687: pose block of J J^T and b, without forming the sparse J.
 * Column (frame i, point j) of J has 6 pose entries f = cameraDerivative^T * Dx_exp_x(log(T_i)) and 3 point
 * entries; the 6N x 6N pose block of J J^T is block-diagonal: M6[i,i] = sum_j f f^T. */
void orc_reproj_accumulate(int N, const double* poses, int M, const double* points, int n,
                           const cmlhip_reproj_obs* obs, double fx, double fy, double* M6, double* b6,
                           double* Jpoints, unsigned char* used) {
    const int m = 6 * N;
    memset(M6, 0, sizeof(double) * m * m);
    memset(b6, 0, sizeof(double) * m);
    if (Jpoints) memset(Jpoints, 0, sizeof(double) * 3 * M);
    for (int k = 0; k < n; k++) {
        int i = obs[k].frame, j = obs[k].point;
        const double* R = poses + 12 * i; const double* t = R + 9;
        orc_se3 T; double xi[6], D[42];
        orc_se3_from_Rt(R, t, &T);
        orc_se3_log(&T, xi);
        orc_se3_dx_exp_x(xi, D);
        double res, Jt[3], Jq[4], Jp[3];
        int ok = orc_reproj_jacobian(R, t, points + 3 * j, obs[k].gx, obs[k].gy, fx, fy, &res, Jt, Jq, Jp);
        if (used) used[k] = 0;
        if (!ok || res > 4 * 4) continue;                  /* :2630 */
        if (used) used[k] = 1;
        if (Jpoints) for (int c = 0; c < 3; c++) Jpoints[3 * j + c] += Jp[c];
        double cam[7] = {Jt[0], Jt[1], Jt[2], Jq[0], Jq[1], Jq[2], Jq[3]};    /* Vector7: head<3> t, tail<4> q, Residual.h:45-46 */
        double f[6];
        for (int c = 0; c < 6; c++) {                     /* cameraDerivative^T * expDerivative, :2641 (rows as Sophus orders them) */
            double s = 0;
            for (int r = 0; r < 7; r++) s += cam[r] * D[r * 6 + c];
            f[c] = s;
        }
        for (int a = 0; a < 6; a++) {
            for (int c = 0; c < 6; c++) M6[(size_t)(6 * i + a) * m + 6 * i + c] += f[a] * f[c];
            b6[6 * i + a] += f[a] * res;                   /* :2655 */
        }
    }
}
