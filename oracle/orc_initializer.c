/* orc_initializer.c — CPU restatement of DSOInitializer::calcResAndGS
 * (src/cml/optimization/dso/DSOInitializer.cpp:451-750) with the tiered fp32 accumulators it uses
 * (Accumulator11 / Accumulator9, src/cml/optimization/dso/MatrixAccumulators.h:94-184,1006-1390).
 * TEST INFRASTRUCTURE ONLY (see cml_oracle.h).  Parity unpinned: the reference has no test or fixture for this path;
 * the restatement follows the statements line by line, including the two places where the reference's literal
 * behaviour differs from its comments: the "alpha energy" loop adds into E after E.finish() (so E.A is the photometric
 * energy only and E.num counts every point twice) and EAlpha stays empty (so alphaOpt depends on |t|^2 * npts only). */
#include <math.h>
#include <string.h>
#include "cml_oracle.h"

typedef struct { float S[45][4], S1k[45][4], S1m[45][4]; float n1, n1k; float H[81]; } acc9;
typedef struct { float S[4], S1k[4], S1m[4]; float n1, n1k; float A; size_t num; } acc11;

static void acc9_init(acc9* a) { memset(a, 0, sizeof *a); }
static void acc9_shift(acc9* a, int force) {                       /* ACC.h:1370-1390 */
    if (a->n1 > 1000 || force) {
        for (int k = 0; k < 45; k++) for (int l = 0; l < 4; l++) { a->S1k[k][l] = a->S[k][l] + a->S1k[k][l]; a->S[k][l] = 0; }
        a->n1k += a->n1; a->n1 = 0;
    }
    if (a->n1k > 1000 || force) {
        for (int k = 0; k < 45; k++) for (int l = 0; l < 4; l++) { a->S1m[k][l] = a->S1k[k][l] + a->S1m[k][l]; a->S1k[k][l] = 0; }
        a->n1k = 0;
    }
}
static void acc9_sse(acc9* a, const float J[9][4]) {              /* updateSSE, ACC.h:1048-1112 */
    int idx = 0;
    for (int r = 0; r < 9; r++)
        for (int c = r; c < 9; c++) { for (int l = 0; l < 4; l++) a->S[idx][l] += J[r][l] * J[c][l]; idx++; }
    a->n1++; acc9_shift(a, 0);
}
static void acc9_single(acc9* a, const float J[9]) {              /* updateSingle, ACC.h:1211-1284 */
    int idx = 0;
    for (int r = 0; r < 9; r++)
        for (int c = r; c < 9; c++) { a->S[idx][0] += J[c] * J[r]; idx++; }
    a->n1++; acc9_shift(a, 0);
}
static void acc9_single_weighted(acc9* a, const float Jin[9], float w) {   /* updateSingleWeighted, ACC.h:1286-1359 */
    float J[9];
    memcpy(J, Jin, sizeof J);
    int idx = 0;
    for (int r = 0; r < 9; r++) {
        a->S[idx][0] += J[r] * J[r] * w; idx++; J[r] *= w;
        for (int c = r + 1; c < 9; c++) { a->S[idx][0] += J[c] * J[r]; idx++; }
    }
    a->n1++; acc9_shift(a, 0);
}
static void acc9_finish(acc9* a) {                                /* ACC.h:1026-1045 */
    acc9_shift(a, 1);
    int idx = 0;
    for (int r = 0; r < 9; r++)
        for (int c = r; c < 9; c++) {
            float d = a->S1m[idx][0] + a->S1m[idx][1] + a->S1m[idx][2] + a->S1m[idx][3];
            a->H[r * 9 + c] = a->H[c * 9 + r] = d; idx++;
        }
}
static void acc11_init(acc11* a) { memset(a, 0, sizeof *a); }
static void acc11_shift(acc11* a, int force) {                    /* ACC.h:164-180 */
    if (a->n1 > 1000 || force) { for (int l = 0; l < 4; l++) { a->S1k[l] = a->S[l] + a->S1k[l]; a->S[l] = 0; } a->n1k += a->n1; a->n1 = 0; }
    if (a->n1k > 1000 || force) { for (int l = 0; l < 4; l++) { a->S1m[l] = a->S1k[l] + a->S1m[l]; a->S1k[l] = 0; } a->n1k = 0; }
}
static void acc11_single(acc11* a, float v) { a->S[0] += v; a->num++; a->n1++; acc11_shift(a, 0); }   /* ACC.h:121-127 */
static void acc11_finish(acc11* a) { acc11_shift(a, 1); a->A = a->S1m[0] + a->S1m[1] + a->S1m[2] + a->S1m[3]; }

/* DSOInitializer::calcResAndGS, DSOInitializer.cpp:451-750 */
void orc_init_calc_res_and_gs(const float* aos3, int wl, int hl, const cmlhip_init_params* P, int npts, cmlhip_init_point* pts,
                              float* H_out, float* b_out, float* H_out_sc, float* b_out_sc, float res[3]) {
    const float* RKi = P->RKi; const float* t = P->t;
    const float fxl = P->fx, fyl = P->fy, cxl = P->cx, cyl = P->cy;
    static acc9 A9, A9SC;
    acc11 E;
    acc9_init(&A9); acc11_init(&E);
    for (int pti = 0; pti < npts; pti++) {
        cmlhip_init_point* pt = &pts[pti];
        float tempPt[8][3], tempU[8], tempV[8], tempNewIdepth[8], hit[8][3];
        for (int idx = 0; idx < 8; idx++) {                                               /* :484-513 */
            const float* pp = pt->p_pattern[idx];
            for (int k = 0; k < 3; k++) {
                float v = 0.0f + (RKi[3 * k] * pp[0] + (RKi[3 * k + 1] * pp[1] + RKi[3 * k + 2] * pp[2]));   /* Eigen: e0 + (e1 + e2), pinned: orc_eig_matvec3f_noalias */
                tempPt[idx][k] = v + t[k] * pt->idepth_new;
            }
            tempU[idx] = tempPt[idx][0] / tempPt[idx][2];
            tempV[idx] = tempPt[idx][1] / tempPt[idx][2];
            const float Ku = fxl * tempU[idx] + cxl, Kv = fyl * tempV[idx] + cyl;
            tempNewIdepth[idx] = pt->idepth_new / tempPt[idx][2];
            hit[idx][0] = hit[idx][1] = hit[idx][2] = 0;
            if (Ku > 1 && Kv > 1 && Ku < wl - 2 && Kv < hl - 2 && tempNewIdepth[idx] > 0) orc_interpolate3(aos3, wl, Ku, Kv, hit[idx]);
            else pt->is_good = 0;
        }
        pt->maxstep = 1e10f;                                                                 /* :520 */
        if (!pt->is_good) {
            acc11_single(&E, pt->energy[0]);
            pt->energy_new[0] = pt->energy[0]; pt->energy_new[1] = pt->energy[1];
            pt->is_good_new = 0;
            continue;
        }
        float dp[9][8];                                                                      /* dp0..dp7, r */
        float dd[8];
        memset(dp, 0, sizeof dp); memset(dd, 0, sizeof dd);
        float* jb = pt->jb;
        for (int k = 0; k < 10; k++) jb[k] = 0;                                              /* :541 */
        int isGood = 1;
        float energy = 0;
        for (int idx = 0; idx < 8; idx++) {                                                  /* :546-613 */
            const float* hitColor = hit[idx];
            const float rlR = pt->color[idx];
            if (!isfinite(rlR) || !isfinite(hitColor[0]) || !isfinite(hitColor[1]) || !isfinite(hitColor[2])) { isGood = 0; break; }
            const float residual = hitColor[0] - P->aff_a * rlR - P->aff_b;
            float hw = fabsf(residual) < P->huber ? 1 : P->huber / fabsf(residual);
            energy += hw * residual * residual * (2 - hw);
            const float dxdd = (t[0] - t[2] * tempU[idx]) / tempPt[idx][2];
            const float dydd = (t[1] - t[2] * tempV[idx]) / tempPt[idx][2];
            if (hw < 1) hw = sqrtf(hw);
            const float dxInterp = hw * hitColor[1] * fxl, dyInterp = hw * hitColor[2] * fyl;
            const float u = tempU[idx], v = tempV[idx], nid = tempNewIdepth[idx];
            dp[0][idx] = nid * dxInterp;
            dp[1][idx] = nid * dyInterp;
            dp[2][idx] = -nid * (u * dxInterp + v * dyInterp);
            dp[3][idx] = -u * v * dxInterp - (1 + v * v) * dyInterp;
            dp[4][idx] = (1 + u * u) * dxInterp + u * v * dyInterp;
            dp[5][idx] = -v * dxInterp + u * dyInterp;
            dp[6][idx] = -hw * P->aff_a * rlR;
            dp[7][idx] = -hw * 1;
            dd[idx] = dxInterp * dxdd + dyInterp * dydd;
            dp[8][idx] = hw * residual;
            const float mx = dxdd * fxl, my = dydd * fyl;
            const float maxstep = 1.0f / sqrtf(mx * mx + my * my);
            if (maxstep < pt->maxstep) pt->maxstep = maxstep;
            for (int k = 0; k < 9; k++) jb[k] += dp[k][idx] * dd[idx];                        /* :600-608 (jb[8] = r*dd) */
            jb[9] += dd[idx] * dd[idx];
        }
        if (!isGood || energy > pt->outlier_th * 20) {                                      /* :616-623 */
            acc11_single(&E, pt->energy[0]);
            pt->is_good_new = 0;
            pt->energy_new[0] = pt->energy[0]; pt->energy_new[1] = pt->energy[1];
            continue;
        }
        acc11_single(&E, energy);
        pt->is_good_new = 1;
        pt->energy_new[0] = energy;
        for (int i = 0; i + 3 < 8; i += 4) {                                                 /* :633-645 */
            float J[9][4];
            for (int k = 0; k < 9; k++) for (int l = 0; l < 4; l++) J[k][l] = dp[k][i + l];
            acc9_sse(&A9, J);
        }
        (void)acc9_single;                                                                   /* :648-653: 8 % 4 == 0, no tail */
    }
    acc11_finish(&E);
    acc9_finish(&A9);
    for (int i = 0; i < npts; i++) {                                                         /* :665-679 */
        cmlhip_init_point* pt = &pts[i];
        if (!pt->is_good_new) acc11_single(&E, pt->energy[1]);
        else {
            pt->energy_new[1] = (pt->idepth_new - 1) * (pt->idepth_new - 1);
            acc11_single(&E, pt->energy_new[1]);
        }
    }
    const float EAlphaA = 0;                                                                 /* EAlpha is never fed */
    float alphaEnergy = (float)((double)P->alpha_w * ((double)EAlphaA + P->t_sqnorm * (double)npts));   /* :681 */
    float alphaOpt;
    if (alphaEnergy > P->alpha_k * npts) { alphaOpt = 0; alphaEnergy = P->alpha_k * npts; }  /* :690-699 */
    else alphaOpt = P->alpha_w;
    acc9_init(&A9SC);
    for (int i = 0; i < npts; i++) {                                                         /* :702-724 */
        cmlhip_init_point* pt = &pts[i];
        if (!pt->is_good_new) continue;
        float* jb = pt->jb;
        pt->last_hessian_new = jb[9];
        jb[8] += alphaOpt * (pt->idepth_new - 1);
        jb[9] += alphaOpt;
        if (alphaOpt == 0) {
            jb[8] += P->coupling_weight * (pt->idepth_new - pt->iR);
            jb[9] += P->coupling_weight;
        }
        jb[9] = 1 / (1 + jb[9]);
        acc9_single_weighted(&A9SC, jb, jb[9]);
    }
    acc9_finish(&A9SC);
    for (int r = 0; r < 8; r++) {                                                            /* :729-732 */
        for (int c = 0; c < 8; c++) { H_out[r * 8 + c] = A9.H[r * 9 + c]; H_out_sc[r * 8 + c] = A9SC.H[r * 9 + c]; }
        b_out[r] = A9.H[r * 9 + 8]; b_out_sc[r] = A9SC.H[r * 9 + 8];
    }
    for (int k = 0; k < 3; k++) {                                                            /* :736-742 */
        H_out[k * 8 + k] += alphaOpt * npts;
        b_out[k] += P->tlog[k] * alphaOpt * npts;
    }
    res[0] = E.A; res[1] = alphaEnergy; res[2] = (float)E.num;
}
