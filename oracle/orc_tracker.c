/*
 * orc_tracker.c — oracle: coarse tracker (TR.cpp = src/cml/optimization/dso/DSOTracker.cpp)
 * and the hybrid ORB reprojection term (BA.cpp:2574-2729, src/cml/optimization/Residual.h).
 * TEST INFRASTRUCTURE ONLY (see cml_oracle.h).
 */
#include "cml_oracle.h"
#include <math.h>
#include <float.h>
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

/* ================================================================== DSOTracker::optimize, TR.cpp:15-246
 * Coarse-to-fine Levenberg-Marquardt over 8 parameters on top of orc_tracker_eval (computeResidual + computeHessian); every
 * statement of the reference's loop is restated in order: per-level iteration caps (:23), minimum term counts (:65,:77), the
 * saturation repeat (:71-75), the four optimizeA/B solver branches (:96-119), the extrapolation below lambda 1e-3 (:140-142),
 * the literal lane / scale pairing of incrementScaled (:144-148), accept / reject (:163-174), the increment-norm exit (:176),
 * the rmse test against mLastResidual (:183-189), the single level repeat (:192-195), light / saturation validity (:203-240). */
static void trk_level_K(const double K0[4], int level, double K[4]) {        /* InternalCalibration.h:116-127 */
    const double d = (double)(1 << level);
    K[0] = K0[0] / d; K[1] = K0[1] / d; K[2] = (K0[2] + 0.5) / d - 0.5; K[3] = (K0[3] + 0.5) / d - 0.5;
}
static void trk_eval(const orc_trk_problem* P, int level, const orc_se3* T, double a, double b, double cutoff_mult,
                     cmlhip_tracker_result* out) {
    double R[9], K[4], aff[2];
    orc_se3_matrix(T, R);
    trk_level_K(P->K, level, K);
    orc_exposure_to(P->ref_a, P->ref_b, P->ref_t, a, b, P->new_t, &aff[0], &aff[1]);      /* reference->getExposure().to(exposure), :269 */
    cmlhip_tracker_params prm = P->prm;
    prm.cutoff = (float)((double)P->prm.cutoff_base * cutoff_mult);                       /* mCutoffThreshold.f() * levelCutoffRepeat[level] */
    orc_tracker_eval(P->aos3[level], P->w[level], P->h[level], P->uvic[level], P->n[level], level, R, T->t, K, aff, P->ref_b, &prm, 1, out, NULL, 0);
}

int orc_tracker_optimize(const orc_trk_problem* P, orc_se3* refToNew, double* cur_a, double* cur_b, orc_trk_result* out,
                         orc_trk_step* log, int log_cap) {
    static const int maxIterations[5] = {10, 20, 50, 50, 50};                            /* :23 */
    int maxLevel = P->levels - 1 < 4 ? P->levels - 1 : 4;
    memset(out, 0, sizeof *out);
    for (int k = 0; k < 6; k++) out->covariance[k] = 999999;
    out->tooManySaturated = 1;
    double E[5] = {0}, levelCutoffRepeat[5] = {0};
    int nT[5] = {0}, nS[5] = {0}, nR[5] = {0};
    double E_new[5] = {0};                       /* newResidual: every level slot keeps the LAST trial evaluated there, accepted or not */
    int nT_new[5] = {0}, nS_new[5] = {0}, nR_new[5] = {0};
    double flow[3] = {0, 0, 0};
    int haveRepeated = 0, nlog = 0;
    orc_se3 cur = *refToNew, nw;
    double a = *cur_a, b = *cur_b, na, nb_;
    double H[64], bv[8];
    cmlhip_tracker_result tr;
#define TRK_FAIL() do { out->isCorrect = 0; goto fill; } while (0)
    for (int level = maxLevel; level >= 0; level--) {
        levelCutoffRepeat[level] = 1;
        trk_eval(P, level, &cur, a, b, levelCutoffRepeat[level], &tr);
        E[level] = tr.E; nT[level] = tr.numTermsInE; nS[level] = tr.numSaturated; nR[level] = tr.numRobust;
        for (int k = 0; k < 3; k++) flow[k] = tr.flow[k];
        if (nT[level] < 20) TRK_FAIL();                                                  /* :65-69 */
        while ((nS[level] / (double)nT[level]) > 0.6 && levelCutoffRepeat[level] < 50) { /* :71-75 */
            levelCutoffRepeat[level] *= 2;
            trk_eval(P, level, &cur, a, b, levelCutoffRepeat[level], &tr);
            E[level] = tr.E; nT[level] = tr.numTermsInE; nS[level] = tr.numSaturated; nR[level] = tr.numRobust;
            for (int k = 0; k < 3; k++) flow[k] = tr.flow[k];
        }
        if (nT[level] - nS[level] < 10) TRK_FAIL();                                      /* :77-81 */
        memcpy(H, tr.H, sizeof H); memcpy(bv, tr.b, sizeof bv);                          /* computeHessian, :85 */
        double lambda = 0.01;
        const double lambdaExtrapolationLimit = 0.001;
        for (int iteration = 0; iteration < maxIterations[level]; iteration++) {
            double D[64], inc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nbv[8];
            memcpy(D, H, sizeof D);
            for (int i = 0; i < 8; i++) { D[i * 8 + i] *= (1 + lambda); nbv[i] = -bv[i]; }
            int ok = 1;
            if (P->optimize_a && P->optimize_b) {
                ok = orc_ldlt_solve(D, nbv, 8, inc) == 0;                                 /* :96-98 */
            } else if (P->optimize_a && !P->optimize_b) {                                 /* :99-102 */
                double S[49], x7[7];
                for (int i = 0; i < 7; i++) for (int j = 0; j < 7; j++) S[i * 7 + j] = D[i * 8 + j];
                ok = orc_ldlt_solve(S, nbv, 7, x7) == 0;
                for (int i = 0; i < 7; i++) inc[i] = x7[i];
                inc[7] = 0;
            } else if (!P->optimize_a && P->optimize_b) {                                 /* :103-114 */
                double Hs[64], bs[8], S[49], nb7[7], x7[7];
                memcpy(Hs, D, sizeof Hs); memcpy(bs, bv, sizeof bs);
                for (int i = 0; i < 8; i++) Hs[i * 8 + 6] = Hs[i * 8 + 7];
                for (int j = 0; j < 8; j++) Hs[6 * 8 + j] = Hs[7 * 8 + j];
                bs[6] = bs[7];
                for (int i = 0; i < 7; i++) { for (int j = 0; j < 7; j++) S[i * 7 + j] = Hs[i * 8 + j]; nb7[i] = -bs[i]; }
                ok = orc_ldlt_solve(S, nb7, 7, x7) == 0;
                for (int i = 0; i < 6; i++) inc[i] = x7[i];
                inc[6] = 0; inc[7] = x7[6];
            } else {                                                                      /* :115-119 */
                double S[36], x6[6];
                for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) S[i * 6 + j] = D[i * 8 + j];
                ok = orc_ldlt_solve(S, nbv, 6, x6) == 0;
                for (int i = 0; i < 6; i++) inc[i] = x6[i];
            }
            for (int i = 0; i < 8; i++) if (!isfinite(inc[i])) ok = 0;
            if (!ok) TRK_FAIL();                                                          /* :121-138 (mBackupSolver off: the default) */
            double extrapFac = 1;
            if (lambda < lambdaExtrapolationLimit) extrapFac = sqrt(sqrt(lambdaExtrapolationLimit / lambda));   /* :140-142 */
            for (int i = 0; i < 8; i++) inc[i] *= extrapFac;
            double incS[8];
            memcpy(incS, inc, sizeof incS);
            for (int i = 0; i < 3; i++) { incS[i] *= (double)P->prm.scale_rot; incS[3 + i] *= (double)P->prm.scale_trans; }   /* :144-146, literal pairing */
            incS[6] *= (double)P->prm.scale_a; incS[7] *= (double)P->prm.scale_b;
            orc_se3 ex;
            orc_se3_exp(incS, &ex);
            orc_se3_mul(&ex, &cur, &nw);                                                  /* newRefToNew = se3 * currentRefToNew, :157 */
            na = a + incS[6]; nb_ = b + incS[7];                                          /* currentExposure.add(...), :159 */
            cmlhip_tracker_result tn;
            trk_eval(P, level, &nw, na, nb_, levelCutoffRepeat[level], &tn);
            E_new[level] = tn.E; nT_new[level] = tn.numTermsInE; nS_new[level] = tn.numSaturated; nR_new[level] = tn.numRobust;
            const int accept = (tn.E / (double)tn.numTermsInE) < (E[level] / (double)nT[level]);   /* :163 */
            if (log && nlog < log_cap) {
                orc_trk_step* s = &log[nlog];
                s->level = level; s->iteration = iteration; s->accept = accept; s->lambda = lambda;
                s->E_new = tn.E; s->E_old = E[level]; s->n_new = tn.numTermsInE; s->n_old = nT[level];
            }
            nlog++;
            if (accept) {
                memcpy(H, tn.H, sizeof H); memcpy(bv, tn.b, sizeof bv);                   /* computeHessian at the accepted state, :166 */
                /* oldResidual = newResidual, :167: a whole-struct copy — the slots of the coarser levels become those of the last TRIAL
                   evaluated there (literal behaviour; only reporting is affected, every test of the loop reads the current level) */
                for (int l = 0; l < 5; l++) { E[l] = E_new[l]; nT[l] = nT_new[l]; nS[l] = nS_new[l]; nR[l] = nR_new[l]; }
                for (int k = 0; k < 3; k++) flow[k] = tn.flow[k];
                cur = nw; a = na; b = nb_;
                lambda *= 0.5;
            } else {
                lambda *= 4;
            }
            double nrm = 0;
            for (int i = 0; i < 8; i++) nrm += inc[i] * inc[i];
            if (sqrt(nrm) < 1e-3) break;                                                  /* :176-179 */
        }
        if (P->have_last && (E[level] / (double)nT[level]) > 1.5 * P->last_rmse[level]) TRK_FAIL();   /* :183-189 */
        if (levelCutoffRepeat[level] > 1 && !haveRepeated) { level++; haveRepeated = 1; } /* :192-195 */
    }
    {
        double relA, relB;
        orc_exposure_to(P->ref_a, P->ref_b, P->ref_t, a, b, P->new_t, &relA, &relB);      /* :203 */
        int haveGoodLight = 1;
        if (P->optimize_a) { if (fabs(a) > 1.2) haveGoodLight = 0; }
        else if (fabs(logf((float)relA)) > 1.5) haveGoodLight = 0;
        if (P->optimize_b) { if (fabs(b) > 200) haveGoodLight = 0; }
        else if (fabs((float)relB) > 200) haveGoodLight = 0;
        int haveGoodPoints = 1;
        if ((double)nS[0] / (double)nT[0] > P->saturated_ratio_th) haveGoodPoints = 0;    /* :231-235 */
        out->isCorrect = haveGoodLight;
        out->tooManySaturated = haveGoodPoints;                                           /* sic, :240 */
        out->relAff[0] = relA; out->relAff[1] = relB;
        double Hi[64];
        orc_inverse(H, 8, Hi);                                                            /* :243 */
        for (int k = 0; k < 6; k++) out->covariance[k] = Hi[k * 8 + k];
        *refToNew = cur; *cur_a = a; *cur_b = b;                                          /* the caller composes reference * refToNew, :237 */
    }
fill:
    for (int l = 0; l < 5; l++) { out->E[l] = E[l]; out->numTermsInE[l] = nT[l]; out->numSaturated[l] = nS[l]; out->numRobust[l] = nR[l]; out->levelCutoffRepeat[l] = levelCutoffRepeat[l]; }
    for (int k = 0; k < 3; k++) out->flow[k] = flow[k];
    out->n_steps = nlog;
    return out->isCorrect;
#undef TRK_FAIL
}

/* DSOTracker::trackWithMotionModel, TR.h:238-383: the hypothesis loop around optimize.  hyp = n_hyp candidate refToNew poses
 * (reference->getCamera().to(testCamera) of Map::multiConstantVelocityMotionModel's list, built by the caller); every try starts
 * from the frame's initial exposure parameters; winner selection, the achieved-residual bookkeeping and the two early exits
 * (:300-309) are the reference's.  Returns haveOneGood; *winner = index of the adopted hypothesis (-1: none), *tries = hypotheses run. */
int orc_tracker_track_with_motion_model(orc_trk_problem* P, int n_hyp, const orc_se3* hyp, double init_a, double init_b, double last_coarse_rmse,
                                        int failure_mode, orc_se3* best_pose, double* best_a, double* best_b, orc_trk_result* best,
                                        double* achieved_out, int* winner, int* tries) {
    int haveOneGood = 0;
    orc_trk_result trackingResult, test;
    memset(&trackingResult, 0, sizeof trackingResult);
    trackingResult.tooManySaturated = 1;                                                  /* Residual(): isCorrect = false, tooManySaturated = true */
    double achievedRes = DBL_MAX;
    *winner = -1;
    int i = 0;
    for (; i < n_hyp; i++) {
        orc_se3 T = hyp[i];
        double a = init_a, b = init_b;
        P->have_last = trackingResult.isCorrect;                                          /* mLastResidual = trackingResult, :272 */
        for (int l = 0; l < 5; l++) P->last_rmse[l] = trackingResult.numTermsInE[l] > 0 ? trackingResult.E[l] / (double)trackingResult.numTermsInE[l] : 0.0;
        orc_tracker_optimize(P, &T, &a, &b, &test, NULL, 0);
        const double rm = test.numTermsInE[0] > 0 ? test.E[0] / (double)test.numTermsInE[0] : NAN;   /* rmse(): asserts numTermsInE[0] > 0 upstream */
        if (trackingResult.tooManySaturated == 1 && test.tooManySaturated == 0 && test.isCorrect && isfinite(rm)) {   /* :280-285 */
            haveOneGood = 1; *best_pose = T; *best_a = a; *best_b = b; trackingResult = test; *winner = i;
        }
        if (test.isCorrect && isfinite(rm) && !(rm >= achievedRes)) {                     /* :288-296 */
            if (trackingResult.tooManySaturated || !test.tooManySaturated) {
                haveOneGood = 1; *best_pose = T; *best_a = a; *best_b = b; trackingResult = test; *winner = i;
            }
        }
        if (haveOneGood) {                                                                /* :299-304 */
            if (test.numTermsInE[0] > 0 && rm < achievedRes) achievedRes = rm;
        }
        const float setting_reTrackThreshold = 1.5f;
        if (haveOneGood && achievedRes < last_coarse_rmse * (double)setting_reTrackThreshold) { i++; break; }   /* :306-309 */
        if (haveOneGood && i >= 50) { i++; break; }                                       /* :311-313 */
    }
    *tries = i;
    if (!haveOneGood && (failure_mode == 1 || failure_mode == 2) && n_hyp > 0) {          /* :322-352 (mode 2 additionally overrides the camera afterwards) */
        orc_se3 T = hyp[0];
        double a = init_a, b = init_b;
        P->have_last = trackingResult.isCorrect;
        for (int l = 0; l < 5; l++) P->last_rmse[l] = trackingResult.numTermsInE[l] > 0 ? trackingResult.E[l] / (double)trackingResult.numTermsInE[l] : 0.0;
        orc_tracker_optimize(P, &T, &a, &b, &trackingResult, NULL, 0);
        *best_pose = T; *best_a = a; *best_b = b; *winner = 0;
        haveOneGood = 1;
    }
    *best = trackingResult;
    *achieved_out = achievedRes;
    return haveOneGood;
}
