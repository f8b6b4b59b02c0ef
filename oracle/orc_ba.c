/*
 * orc_ba.c — oracle: sliding-window photometric bundle adjustment.
 * TEST INFRASTRUCTURE ONLY (see cml_oracle.h).  Restates BA.cpp =
 * src/cml/optimization/dso/DSOBundleAdjustment.cpp, including the reference's mixed
 * float/double arithmetic (member variables of DSOBundleAdjustmentLinearizationContext
 * are float, BA.cpp:20-58; Parameter::f() is double, src/cml/base/Parameter.h:16,40).
 * Build with -ffp-contract=off so the statement order below is the arithmetic order.
 */
#include "cml_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* offsets into the 74-float DSORawResidualJacobian record (DSOResidual.h:44-68) */
enum { O_RES = 0, O_XI0 = 8, O_XI1 = 14, O_C0 = 20, O_C1 = 24, O_DD = 28, O_JI0 = 30, O_JI1 = 38,
       O_JAB0 = 46, O_JAB1 = 54, O_JI2 = 62, O_JABJI = 66, O_JAB2 = 70 };

static const int STAR8[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}}; /* types.h:1381-1393 */

static void* zalloc(size_t n) { return calloc(n ? n : 1, 1); }

orc_ba_window* orc_ba_create(const cmlhip_ba_params* prm, int N, const cmlhip_ba_frame* frames,
                             const float* const* images, int P, const cmlhip_ba_point* points,
                             int R, const cmlhip_ba_residual* res) {
    orc_ba_window* w = (orc_ba_window*)zalloc(sizeof *w);
    w->prm = *prm; w->N = N; w->P = P; w->R = R;
    for (int i = 0; i < N; i++) {
        w->image[i] = images[i];
        w->frame_energy_th[i] = frames[i].frame_energy_th;
        w->b0[i] = frames[i].b0;
    }
    w->pairs = (cmlhip_ba_pair*)zalloc(sizeof(cmlhip_ba_pair) * N * N);
    w->points = (cmlhip_ba_point*)zalloc(sizeof(cmlhip_ba_point) * P);
    memcpy(w->points, points, sizeof(cmlhip_ba_point) * P);
    w->idepth_backup = (float*)zalloc(4 * P);
    w->Hdd_accAF = (float*)zalloc(4 * P); w->bd_accAF = (float*)zalloc(4 * P); w->Hcd_accAF = (float*)zalloc(16 * P);
    w->Hdd_accLF = (float*)zalloc(4 * P); w->bd_accLF = (float*)zalloc(4 * P); w->Hcd_accLF = (float*)zalloc(16 * P);
    w->HdiF = (float*)zalloc(4 * P); w->bdSumF = (float*)zalloc(4 * P); w->step = (double*)zalloc(8 * P);
    w->r_point = (int*)zalloc(4 * R); w->r_target = (int*)zalloc(4 * R);
    w->r_state = (int*)zalloc(4 * R); w->r_new_state = (int*)zalloc(4 * R);
    w->r_lin = (unsigned char*)zalloc(R); w->r_good = (unsigned char*)zalloc(R);
    w->r_energy = (float*)zalloc(4 * R); w->r_new_energy = (float*)zalloc(4 * R); w->r_new_energy_wo = (float*)zalloc(4 * R);
    w->r_center = (float*)zalloc(12 * R);
    w->rJ = (float*)zalloc(4 * 74 * (size_t)R); w->efsJ = (float*)zalloc(4 * 74 * (size_t)R);
    w->JpJdF = (float*)zalloc(32 * R); w->res_toZeroF = (float*)zalloc(32 * R);
    w->pair_of = (int*)zalloc(4 * R);
    w->by_point_off = (int*)zalloc(4 * (P + 1)); w->by_point = (int*)zalloc(4 * R);
    w->by_pair_off = (int*)zalloc(4 * (N * N + 1)); w->by_pair = (int*)zalloc(4 * R);
    w->accA = (float*)zalloc(4 * 169 * N * N); w->accL = (float*)zalloc(4 * 169 * N * N);
    w->accA_num = (int*)zalloc(4 * N * N); w->accL_num = (int*)zalloc(4 * N * N);
    for (int r = 0; r < R; r++) {
        w->r_point[r] = res[r].point; w->r_target[r] = res[r].target;
        w->r_state[r] = res[r].state; w->r_lin[r] = (unsigned char)res[r].is_linearized;
        w->r_new_state[r] = CMLHIP_RES_OUTLIER;            /* DSOResidual::resetOOB, DSOResidual.h:83-88 */
        w->pair_of[r] = points[res[r].point].host + res[r].target * N;   /* htIDX, BA.cpp:1677 */
        w->by_point_off[res[r].point + 1]++;
        w->by_pair_off[w->pair_of[r] + 1]++;
    }
    for (int p = 0; p < P; p++) w->by_point_off[p + 1] += w->by_point_off[p];
    for (int q = 0; q < N * N; q++) w->by_pair_off[q + 1] += w->by_pair_off[q];
    int* c1 = (int*)zalloc(4 * P); int* c2 = (int*)zalloc(4 * N * N);
    for (int r = 0; r < R; r++) {
        int p = res[r].point, q = w->pair_of[r];
        w->by_point[w->by_point_off[p] + c1[p]++] = r;
        w->by_pair[w->by_pair_off[q] + c2[q]++] = r;
    }
    free(c1); free(c2);
    return w;
}

void orc_ba_destroy(orc_ba_window* w) {
    if (!w) return;
    free(w->pairs); free(w->points); free(w->idepth_backup);
    free(w->Hdd_accAF); free(w->bd_accAF); free(w->Hcd_accAF); free(w->Hdd_accLF); free(w->bd_accLF); free(w->Hcd_accLF);
    free(w->HdiF); free(w->bdSumF); free(w->step);
    free(w->r_point); free(w->r_target); free(w->r_state); free(w->r_new_state); free(w->r_lin); free(w->r_good);
    free(w->r_energy); free(w->r_new_energy); free(w->r_new_energy_wo); free(w->r_center);
    free(w->rJ); free(w->efsJ); free(w->JpJdF); free(w->res_toZeroF);
    free(w->pair_of); free(w->by_point_off); free(w->by_point); free(w->by_pair_off); free(w->by_pair);
    free(w->accA); free(w->accL); free(w->accA_num); free(w->accL_num);
    free(w);
}

void orc_ba_set_pairs(orc_ba_window* w, const cmlhip_ba_pair* pairs) {
    memcpy(w->pairs, pairs, sizeof(cmlhip_ba_pair) * w->N * w->N);
}

/* ------------------------------------------------------------------ linearize, BA.cpp:62-316 */
double orc_ba_linearize_one(orc_ba_window* w, int r) {
    const cmlhip_ba_params* P = &w->prm;
    const cmlhip_ba_point* pt = &w->points[w->r_point[r]];
    const int host = pt->host, target = w->r_target[r];
    const cmlhip_ba_pair* pc = &w->pairs[host * w->N + target];
    float* rJ = w->rJ + 74 * (size_t)r;
    const float* image = w->image[target];
    const double fxi = 1.0 / P->fx, fyi = 1.0 / P->fy;  /* PinholeUndistorter::mFinv, InternalCalibration.h:42-47 */

    w->r_new_energy_wo[r] = -1;                         /* :66 */
    if (w->r_state[r] == CMLHIP_RES_OOB) return w->r_energy[r];   /* :68-72 */

    float JIdxJIdx_00 = 0, JIdxJIdx_11 = 0, JIdxJIdx_10 = 0;
    float JabJIdx_00 = 0, JabJIdx_01 = 0, JabJIdx_10 = 0, JabJIdx_11 = 0;
    float JabJab_00 = 0, JabJab_01 = 0, JabJab_11 = 0;
    float wJI2_sum = 0, energyLeft = 0;

    const double cxd = (double)pt->x, cyd = (double)pt->y;      /* Corner float -> DistortedVector2d, types.h:1177-1182 */
    const double idepth = pt->idepth;
    const double* Rm = pc->R; const double* t = pc->t;

    /* centre, :102-131 */
    double rx = (cxd - P->cx) * fxi, ry = (cyd - P->cy) * fyi;   /* undistort, InternalCalibration.h:79-82 */
    double px = (Rm[0] * rx + Rm[1] * ry + Rm[2] * 1.0) + t[0] * idepth;
    double py = (Rm[3] * rx + Rm[4] * ry + Rm[5] * 1.0) + t[1] * idepth;
    double pz = (Rm[6] * rx + Rm[7] * ry + Rm[8] * 1.0) + t[2] * idepth;
    double Kud = (px / pz) * P->fx + P->cx, Kvd = (py / pz) * P->fy + P->cy;   /* distort(hnormalized) */
    float drescale = (float)(1.0 / pz);
    if (!(Kud >= 2 && Kvd >= 2 && Kud < P->w - 2 && Kvd < P->h - 2)) {        /* :115-118 */
        w->r_new_state[r] = CMLHIP_RES_OOB;
        return w->r_energy[r];
    }
    float new_idepth = (float)(drescale * idepth);
    float u = (float)px, v = (float)py;          /* :121-122 — the UN-normalised x,y of projectedcurp, literal */
    float Ku = (float)Kud, Kv = (float)Kvd;
    double KliP0 = rx, KliP1 = ry;
    float fx = (float)P->fx, fy = (float)P->fy;  /* :127-129, float members */
    w->r_center[3 * r] = Ku; w->r_center[3 * r + 1] = Kv; w->r_center[3 * r + 2] = new_idepth;   /* :131 */

    const double* R0 = pc->R0; const double* t0 = pc->t0;
    float d_d_x = (float)(drescale * (t0[0] - t0[2] * u) * fx);   /* :141-142 */
    float d_d_y = (float)(drescale * (t0[1] - t0[2] * v) * fy);
    double dCx[4], dCy[4];
    dCx[2] = drescale * (R0[6] * u - R0[0]);                      /* :145-153 */
    dCx[3] = (fx * drescale) * (R0[7] * u - R0[1]) / fy;
    dCx[0] = KliP0 * dCx[2];
    dCx[1] = KliP1 * dCx[3];
    dCy[2] = (fy * drescale) * (R0[6] * v - R0[3]) / fx;
    dCy[3] = drescale * (R0[7] * v - R0[4]);
    dCy[0] = KliP0 * dCy[2];
    dCy[1] = KliP1 * dCy[3];
    dCx[0] = (dCx[0] + u) * P->scale_f;                           /* :155-163 */
    dCx[1] *= P->scale_f;
    dCx[2] = (dCx[2] + 1) * P->scale_c;
    dCx[3] *= P->scale_c;
    dCy[0] *= P->scale_f;
    dCy[1] = (dCy[1] + v) * P->scale_f;
    dCy[2] *= P->scale_c;
    dCy[3] = (dCy[3] + 1) * P->scale_c;
    /* :166-178, all-float expressions stored to double then cast back (exact) */
    rJ[O_XI0 + 0] = new_idepth * fx;
    rJ[O_XI0 + 1] = 0;
    rJ[O_XI0 + 2] = -new_idepth * u * fx;
    rJ[O_XI0 + 3] = -u * v * fx;
    rJ[O_XI0 + 4] = (1 + u * u) * fx;
    rJ[O_XI0 + 5] = -v * fx;
    rJ[O_XI1 + 0] = 0;
    rJ[O_XI1 + 1] = new_idepth * fy;
    rJ[O_XI1 + 2] = -new_idepth * v * fy;
    rJ[O_XI1 + 3] = -(1 + v * v) * fy;
    rJ[O_XI1 + 4] = u * v * fy;
    rJ[O_XI1 + 5] = u * fy;
    for (int i = 0; i < 4; i++) { rJ[O_C0 + i] = (float)dCx[i]; rJ[O_C1 + i] = (float)dCy[i]; }
    rJ[O_DD] = d_d_x; rJ[O_DD + 1] = d_d_y;

    const float b0 = w->b0[host];                                  /* :243 */
    for (int idx = 0; idx < 8; idx++) {                            /* :193-282 */
        double sx = cxd + STAR8[idx][0], sy = cyd + STAR8[idx][1];
        double qx = (sx - P->cx) * fxi, qy = (sy - P->cy) * fyi;
        double ppx = (Rm[0] * qx + Rm[1] * qy + Rm[2] * 1.0) + t[0] * idepth;
        double ppy = (Rm[3] * qx + Rm[4] * qy + Rm[5] * 1.0) + t[1] * idepth;
        double ppz = (Rm[6] * qx + Rm[7] * qy + Rm[8] * 1.0) + t[2] * idepth;
        double kx = (ppx / ppz) * P->fx + P->cx, ky = (ppy / ppz) * P->fy + P->cy;
        if (!(kx >= 2 && ky >= 2 && kx < P->w - 2 && ky < P->h - 2)) {       /* :209-212 */
            w->r_new_state[r] = CMLHIP_RES_OOB;
            return w->r_energy[r];
        }
        float refColor = pt->colors[idx];
        float tap[3];
        orc_interpolate3(image, P->w, (float)kx, (float)ky, tap);           /* :218 */
        if (!(isfinite(tap[0]) && isfinite(tap[1]) && isfinite(tap[2]))) {  /* :220-223: setState, not setNewState */
            w->r_state[r] = CMLHIP_RES_OOB;
            return w->r_energy[r];
        }
        float curColor = tap[0], gx = tap[1], gy = tap[2];
        float refRealColor = (float)(pc->aff_a * (double)refColor + pc->aff_b);   /* :229, ExposureTransition Exposure.h:30-32 */
        float residual = curColor - refRealColor;
        float hw = fabs((double)residual) < (double)P->huber ? 1.0f : (float)((double)P->huber / (double)fabsf(residual));  /* :233 */
        float wgt = sqrtf((float)((double)P->outlier_th_sum / ((double)P->outlier_th_sum + (double)(gx * gx + gy * gy)))); /* :234 */
        wgt = (float)(0.5f * ((double)wgt + (double)pt->weights[idx]));                                                   /* :235 */
        energyLeft = (float)((double)energyLeft + (double)(wgt * wgt * hw * residual * residual) * (2.0 - (double)hw));   /* :237 */
        if (hw < 1) hw = sqrtf(hw);
        hw = hw * wgt;
        double h1 = (double)(gx * hw), h2 = (double)(gy * hw);       /* hitColor is a double Vector3, :49,:246-247 */
        float drdA = curColor - b0;
        rJ[O_RES + idx] = residual * hw;
        rJ[O_JI0 + idx] = (float)h1;
        rJ[O_JI1 + idx] = (float)h2;
        rJ[O_JAB0 + idx] = drdA * hw;
        rJ[O_JAB1 + idx] = hw;
        JIdxJIdx_00 = (float)(JIdxJIdx_00 + h1 * h1);
        JIdxJIdx_11 = (float)(JIdxJIdx_11 + h2 * h2);
        JIdxJIdx_10 = (float)(JIdxJIdx_10 + h1 * h2);
        JabJIdx_00 = (float)(JabJIdx_00 + (double)(drdA * hw) * h1);
        JabJIdx_01 = (float)(JabJIdx_01 + (double)(drdA * hw) * h2);
        JabJIdx_10 = (float)(JabJIdx_10 + (double)hw * h1);
        JabJIdx_11 = (float)(JabJIdx_11 + (double)hw * h2);
        JabJab_00 += drdA * drdA * hw * hw;
        JabJab_01 += drdA * hw * hw;
        JabJab_11 += hw * hw;
        wJI2_sum = (float)(wJI2_sum + (double)(hw * hw) * (h1 * h1 + h2 * h2));   /* :271 */
        if (!P->optimize_a) rJ[O_JAB0 + idx] = 0;                  /* :273-278 (sums above already taken) */
        if (!P->optimize_b) rJ[O_JAB1 + idx] = 0;
    }
    rJ[O_JI2 + 0] = JIdxJIdx_00; rJ[O_JI2 + 1] = JIdxJIdx_10; rJ[O_JI2 + 2] = JIdxJIdx_10; rJ[O_JI2 + 3] = JIdxJIdx_11;
    /* JabJIdx(0,0),(1,0),(0,1),(1,1) column-major */
    rJ[O_JABJI + 0] = JabJIdx_00; rJ[O_JABJI + 1] = JabJIdx_10; rJ[O_JABJI + 2] = JabJIdx_01; rJ[O_JABJI + 3] = JabJIdx_11;
    rJ[O_JAB2 + 0] = JabJab_00; rJ[O_JAB2 + 1] = JabJab_01; rJ[O_JAB2 + 2] = JabJab_01; rJ[O_JAB2 + 3] = JabJab_11;

    if (!isfinite(energyLeft)) {                                    /* :297-300 */
        w->r_new_state[r] = CMLHIP_RES_OOB;
        return w->r_energy[r];
    }
    w->r_new_energy_wo[r] = energyLeft;
    float th = w->frame_energy_th[host] > w->frame_energy_th[target] ? w->frame_energy_th[host] : w->frame_energy_th[target];
    if (energyLeft > th || wJI2_sum < 2) {                          /* :303-307 */
        energyLeft = th;
        w->r_new_state[r] = CMLHIP_RES_OUTLIER;
    } else {
        w->r_new_state[r] = CMLHIP_RES_IN;
    }
    w->r_new_energy[r] = energyLeft;
    return energyLeft;
}

static int cmp_float(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* BA.cpp:1551-1565 (loop), :1608-1610, :2419-2464 (setNewFrameEnergyTH) */
void orc_ba_linearize_all(orc_ba_window* w, cmlhip_ba_lin_result* out) {
    double stats = 0;
    /* The reference's loop is serial (BA.cpp:1551-1565).  Built with -fopenmp (make fast_omp: bench.py's all-cores CPU baseline,
       SURVEY §8d) the residuals are spread over the host cores: each one touches only its own slots, the energy sum is a reduction. */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : stats)
#endif
    for (int r = 0; r < w->R; r++) {
        if (w->r_lin[r]) continue;          /* mActiveResiduals holds the non-linearized residuals, BA.cpp:766-779 */
        stats += orc_ba_linearize_one(w, r);
    }
    float* v = (float*)malloc(sizeof(float) * (w->R ? w->R : 1));
    int n = 0;
    for (int r = 0; r < w->R; r++)
        if (!w->r_lin[r] && w->r_new_energy_wo[r] >= 0 && w->r_target[r] == w->N - 1) v[n++] = w->r_new_energy_wo[r];
    double th;
    if (n == 0) {
        th = 12 * 12 * 8;
    } else {
        int nth = (int)(0.7f * (float)n);                /* :2448 */
        qsort(v, n, sizeof(float), cmp_float);           /* nth_element: the value at sorted position nth */
        float nthElement = sqrtf(v[nth]);
        th = (double)(nthElement * 1.5f);                /* :2458 */
        th = (double)(26.0f * 0.5f) + th * (double)(1 - 0.5f);
        th = th * th;
        th *= (double)(1.0f * 1.0f);
    }
    free(v);
    w->frame_energy_th[w->N - 1] = (float)th;
    if (out) {
        out->energy = stats;
        out->new_frame_energy_th = (float)th;
        out->n_in = out->n_oob = out->n_outlier = 0;
        for (int r = 0; r < w->R; r++) {
            if (w->r_lin[r]) continue;
            if (w->r_new_state[r] == CMLHIP_RES_IN) out->n_in++;
            else if (w->r_new_state[r] == CMLHIP_RES_OOB) out->n_oob++;
            else out->n_outlier++;
        }
    }
}

/* BA.cpp:2051-2093 */
static void apply_one(orc_ba_window* w, int r, int copy) {
    {
        if (copy) {
            if (w->r_state[r] == CMLHIP_RES_OOB) return;         /* return: can never go back from OOB */
            if (w->r_new_state[r] == CMLHIP_RES_IN) {
                w->r_good[r] = 1;
                float* a = w->rJ + 74 * (size_t)r; float* b = w->efsJ + 74 * (size_t)r;
                for (int i = 0; i < 74; i++) { float t = a[i]; a[i] = b[i]; b[i] = t; }   /* std::swap(rJ, efsJ) */
                const float* J = b;
                /* JI_JI_Jd = JIdx2 * Jpdd (column-major 2x2) */
                float g0 = J[O_JI2 + 0] * J[O_DD] + J[O_JI2 + 2] * J[O_DD + 1];
                float g1 = J[O_JI2 + 1] * J[O_DD] + J[O_JI2 + 3] * J[O_DD + 1];
                float* o = w->JpJdF + 8 * r;
                for (int i = 0; i < 6; i++) o[i] = J[O_XI0 + i] * g0 + J[O_XI1 + i] * g1;
                o[6] = J[O_JABJI + 0] * J[O_DD] + J[O_JABJI + 2] * J[O_DD + 1];
                o[7] = J[O_JABJI + 1] * J[O_DD] + J[O_JABJI + 3] * J[O_DD + 1];
            } else {
                w->r_good[r] = 0;
            }
        }
        w->r_state[r] = w->r_new_state[r];
        w->r_energy[r] = w->r_new_energy[r];
    }
}
void orc_ba_apply(orc_ba_window* w, int copy) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int r = 0; r < w->R; r++) {
        if (w->r_lin[r]) continue;
        apply_one(w, r, copy);
    }
}

/* ------------------------------------------------------------------ tiered accumulators, ACC.h */
typedef struct { float A[64], A1k[64], A1m[64]; float numIn1, numIn1k, numIn1m; int n; } tier_acc;   /* AccumulatorXX / AccumulatorX, ACC.h:33-92,185-250 */
static void tier_init(tier_acc* a, int n) { memset(a, 0, sizeof *a); a->n = n; }
static void tier_shift(tier_acc* a, int force) {
    if (a->numIn1 > 1000 || force) {
        for (int i = 0; i < a->n; i++) { a->A1k[i] += a->A[i]; a->A[i] = 0; }
        a->numIn1k += a->numIn1; a->numIn1 = 0;
    }
    if (a->numIn1k > 1000 || force) {
        for (int i = 0; i < a->n; i++) { a->A1m[i] += a->A1k[i]; a->A1k[i] = 0; }
        a->numIn1m += a->numIn1k; a->numIn1k = 0;
    }
}
/* A += w * L * R^T, rows x cols, stored row-major in A */
static void tier_update_outer(tier_acc* a, const float* L, int rows, const float* Rv, int cols, float wgt) {
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) a->A[i * cols + j] += (wgt * L[i]) * Rv[j];
    a->numIn1++;
    tier_shift(a, 0);
}
static void tier_update_vec(tier_acc* a, const float* L, int rows, float wgt) {
    for (int i = 0; i < rows; i++) a->A[i] += wgt * L[i];
    a->numIn1++;
    tier_shift(a, 0);
}
static int tier_finish(tier_acc* a) { tier_shift(a, 1); return (int)(a->numIn1 + a->numIn1k + a->numIn1m); }

typedef struct {                       /* AccumulatorApprox, ACC.h:613-998 */
    float D[55], D1k[55], D1m[55], TR[30], TR1k[30], TR1m[30], BR[6], BR1k[6], BR1m[6];
    float numIn1, numIn1k, numIn1m; int num;
} approx_acc;
static void approx_shift(approx_acc* a, int force) {   /* ACC.h:960-996 */
    if (a->numIn1 > 1000 || force) {
        for (int i = 0; i < 55; i++) { a->D1k[i] += a->D[i]; a->D[i] = 0; }
        for (int i = 0; i < 30; i++) { a->TR1k[i] += a->TR[i]; a->TR[i] = 0; }
        for (int i = 0; i < 6; i++) { a->BR1k[i] += a->BR[i]; a->BR[i] = 0; }
        a->numIn1k += a->numIn1; a->numIn1 = 0;
    }
    if (a->numIn1k > 1000 || force) {
        for (int i = 0; i < 55; i++) { a->D1m[i] += a->D1k[i]; a->D1k[i] = 0; }
        for (int i = 0; i < 30; i++) { a->TR1m[i] += a->TR1k[i]; a->TR1k[i] = 0; }
        for (int i = 0; i < 6; i++) { a->BR1m[i] += a->BR1k[i]; a->BR1k[i] = 0; }
        a->numIn1m += a->numIn1k; a->numIn1k = 0;
    }
}
/* ACC.h:776-858: x = [x4 x6], y = [y4 y6]; Data[idx(r<=c)] += a x_c x_r + c y_c y_r + b (x_c y_r + y_c x_r) */
static void approx_update(approx_acc* A, const float* x, const float* y, float a, float b, float c) {
    int idx = 0;
    for (int r = 0; r < 10; r++)
        for (int cc = r; cc < 10; cc++) {
            A->D[idx] += a * x[cc] * x[r] + c * y[cc] * y[r] + b * (x[cc] * y[r] + y[cc] * x[r]);
            idx++;
        }
    A->num++; A->numIn1++;
    approx_shift(A, 0);
}
/* ACC.h:861-916 */
static void approx_update_tr(approx_acc* A, const float* x, const float* y, float TR00, float TR10, float TR01,
                             float TR11, float TR02, float TR12) {
    for (int i = 0; i < 10; i++) {
        A->TR[3 * i + 0] += x[i] * TR00 + y[i] * TR10;
        A->TR[3 * i + 1] += x[i] * TR01 + y[i] * TR11;
        A->TR[3 * i + 2] += x[i] * TR02 + y[i] * TR12;
    }
}
/* ACC.h:918-932 */
static void approx_update_br(approx_acc* A, float a00, float a01, float a02, float a11, float a12, float a22) {
    A->BR[0] += a00; A->BR[1] += a01; A->BR[2] += a02; A->BR[3] += a11; A->BR[4] += a12; A->BR[5] += a22;
}
/* ACC.h:639-673 -> 13x13 row-major */
static void approx_finish(approx_acc* A, float* H) {
    memset(H, 0, sizeof(float) * 169);
    approx_shift(A, 1);
    int idx = 0;
    for (int r = 0; r < 10; r++)
        for (int c = r; c < 10; c++) { H[r * 13 + c] = H[c * 13 + r] = A->D1m[idx]; idx++; }
    idx = 0;
    for (int r = 0; r < 10; r++)
        for (int c = 0; c < 3; c++) { H[r * 13 + c + 10] = H[(c + 10) * 13 + r] = A->TR1m[idx]; idx++; }
    H[10 * 13 + 10] = A->BR1m[0];
    H[10 * 13 + 11] = H[11 * 13 + 10] = A->BR1m[1];
    H[10 * 13 + 12] = H[12 * 13 + 10] = A->BR1m[2];
    H[11 * 13 + 11] = A->BR1m[3];
    H[11 * 13 + 12] = H[12 * 13 + 11] = A->BR1m[4];
    H[12 * 13 + 12] = A->BR1m[5];
}

/* addToHessianTop, BA.cpp:1648-1779 (modes ACTIVE / LINEARIZED) */
static void add_to_hessian_top(orc_ba_window* w, int p, int mode, approx_acc* acc, const cmlhip_ba_accum_in* in) {
    const cmlhip_ba_point* pt = &w->points[p];
    float dc[4];
    for (int i = 0; i < 4; i++) dc[i] = (float)in->cdelta[i];
    float dd = (float)(pt->idepth - (double)pt->idepth_zero);        /* deltaF, BA.cpp:1183 */
    float bd_acc = 0, Hdd_acc = 0, Hcd_acc[4] = {0, 0, 0, 0};
    for (int k = w->by_point_off[p]; k < w->by_point_off[p + 1]; k++) {
        int r = w->by_point[k];
        if (mode == CMLHIP_MODE_ACTIVE) { if (w->r_lin[r] || !w->r_good[r]) continue; }
        if (mode == CMLHIP_MODE_LINEARIZED) { if (!w->r_lin[r] || !w->r_good[r]) continue; }
        if (mode == CMLHIP_MODE_MARGINALIZED) { if (!w->r_good[r]) continue; }   /* :1670-1674 (asserts isLinearized) */
        const float* J = w->efsJ + 74 * (size_t)r;
        int ht = w->pair_of[r];
        const float* dp = in->adHTdeltaF + 8 * ht;
        double resApprox[8];
        if (mode == CMLHIP_MODE_ACTIVE) {
            for (int i = 0; i < 8; i++) resApprox[i] = (double)J[O_RES + i];
        } else if (mode == CMLHIP_MODE_MARGINALIZED) {
            for (int i = 0; i < 8; i++) resApprox[i] = (double)w->res_toZeroF[8 * r + i];   /* :1689-1692 */
        } else {
            /* BA.cpp:1699-1713.  The reference stores the 8 float results through a float* into a
             * double[8] (:1712) — undefined contents; this restates the evident intent (float rtz
             * widened to double).  Unreachable in the default flow: linearized residuals only exist
             * between tryMarginalize and marginalizePointsF, never inside solveSystem. */
            /* rJ.Jpdxi[k].dot(dp.head<6>()) + rJ.Jpdc[k].dot(dc) + rJ.Jpdd[k]*dd, Eigen's evaluation order (orc_eig_jp_delta) */
            float Jp_delta_x = orc_eig_jp_delta(J + O_XI0, dp, J + O_C0, in->cdelta, J[O_DD], dd, 0);
            float Jp_delta_y = orc_eig_jp_delta(J + O_XI1, dp, J + O_C1, in->cdelta, J[O_DD + 1], dd, 0);
            for (int i = 0; i < 8; i++) {
                float rtz = w->res_toZeroF[8 * r + i];
                rtz = rtz + J[O_JI0 + i] * Jp_delta_x;
                rtz = rtz + J[O_JI1 + i] * Jp_delta_y;
                rtz = rtz + J[O_JAB0 + i] * dp[6];
                rtz = rtz + J[O_JAB1 + i] * dp[7];
                resApprox[i] = (double)rtz;
            }
        }
        double JI_r0 = 0, JI_r1 = 0, Jab_r0 = 0, Jab_r1 = 0;          /* Vector2 = double, :1719-1729 */
        float rr = 0;
        for (int i = 0; i < 8; i++) {
            JI_r0 += resApprox[i] * (double)J[O_JI0 + i];
            JI_r1 += resApprox[i] * (double)J[O_JI1 + i];
            Jab_r0 += resApprox[i] * (double)J[O_JAB0 + i];
            Jab_r1 += resApprox[i] * (double)J[O_JAB1 + i];
            rr = (float)((double)rr + resApprox[i] * resApprox[i]);
        }
        float x[10], y[10];
        for (int i = 0; i < 4; i++) { x[i] = J[O_C0 + i]; y[i] = J[O_C1 + i]; }
        for (int i = 0; i < 6; i++) { x[4 + i] = J[O_XI0 + i]; y[4 + i] = J[O_XI1 + i]; }
        approx_acc* A = &acc[ht];
        approx_update(A, x, y, J[O_JI2 + 0], J[O_JI2 + 2], J[O_JI2 + 3]);                                   /* :1731-1734 */
        approx_update_br(A, J[O_JAB2 + 0], J[O_JAB2 + 2], (float)Jab_r0, J[O_JAB2 + 3], (float)Jab_r1, rr);   /* :1736-1738 */
        approx_update_tr(A, x, y, J[O_JABJI + 0], J[O_JABJI + 2], J[O_JABJI + 1], J[O_JABJI + 3],
                         (float)JI_r0, (float)JI_r1);                                                       /* :1740-1745 */
        float g0 = J[O_JI2 + 0] * J[O_DD] + J[O_JI2 + 2] * J[O_DD + 1];    /* Ji2_Jpdd, :1747 */
        float g1 = J[O_JI2 + 1] * J[O_DD] + J[O_JI2 + 3] * J[O_DD + 1];
        bd_acc = (float)((double)bd_acc + (JI_r0 * (double)J[O_DD] + JI_r1 * (double)J[O_DD + 1]));
        Hdd_acc += g0 * J[O_DD] + g1 * J[O_DD + 1];
        for (int i = 0; i < 4; i++) Hcd_acc[i] += J[O_C0 + i] * g0 + J[O_C1 + i] * g1;
    }
    if (mode == CMLHIP_MODE_ACTIVE) {
        w->Hdd_accAF[p] = Hdd_acc; w->bd_accAF[p] = bd_acc; memcpy(w->Hcd_accAF + 4 * p, Hcd_acc, 16);
    } else {
        w->Hdd_accLF[p] = Hdd_acc; w->bd_accLF[p] = bd_acc; memcpy(w->Hcd_accLF + 4 * p, Hcd_acc, 16);
    }
    if (mode == CMLHIP_MODE_MARGINALIZED) {                  /* :1769-1775 */
        w->Hdd_accAF[p] = 0; w->bd_accAF[p] = 0; memset(w->Hcd_accAF + 4 * p, 0, 16);
    }
}

static void mat8_abt(const double* A, const double* B, const double* C, double* out) {
    /* out(8x8) += A * B * C^T */
    double T[64];
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            double s = 0;
            for (int k = 0; k < 8; k++) s += A[i * 8 + k] * B[k * 8 + j];
            T[i * 8 + j] = s;
        }
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            double s = 0;
            for (int k = 0; k < 8; k++) s += T[i * 8 + k] * C[j * 8 + k];
            out[i * 8 + j] = s;
        }
}

/* stitchDoubleTop, BA.cpp:1781-1878 */
static void stitch_top(const orc_ba_window* w, const float* accH, const int* accNum, const cmlhip_ba_accum_in* in,
                       int usePrior, double* fH, double* fb) {
    const int N = w->N, n = 8 * N + 4;
    memset(fH, 0, sizeof(double) * n * n);
    memset(fb, 0, sizeof(double) * n);
#define FH(i, j) fH[(size_t)(i) * n + (j)]
    for (int h = 0; h < N; h++)
        for (int t = 0; t < N; t++) {
            int hIdx = 4 + h * 8, tIdx = 4 + t * 8, aidx = h + N * t;
            if (accNum[aidx] == 0) continue;
            double aH[169];
            for (int i = 0; i < 169; i++) aH[i] = (double)accH[169 * aidx + i];
            const double* AH = in->adHost + 64 * aidx; const double* AT = in->adTarget + 64 * aidx;
            double B[64], blk[64];
            for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) B[i * 8 + j] = aH[(4 + i) * 13 + 4 + j];
            mat8_abt(AH, B, AH, blk);
            for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) FH(hIdx + i, hIdx + j) += blk[i * 8 + j];
            mat8_abt(AT, B, AT, blk);
            for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) FH(tIdx + i, tIdx + j) += blk[i * 8 + j];
            mat8_abt(AH, B, AT, blk);
            for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) FH(hIdx + i, tIdx + j) += blk[i * 8 + j];
            for (int i = 0; i < 8; i++)
                for (int j = 0; j < 4; j++) {
                    double s1 = 0, s2 = 0;
                    for (int k = 0; k < 8; k++) { s1 += AH[i * 8 + k] * aH[(4 + k) * 13 + j]; s2 += AT[i * 8 + k] * aH[(4 + k) * 13 + j]; }
                    FH(hIdx + i, j) += s1; FH(tIdx + i, j) += s2;
                }
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) FH(i, j) += aH[i * 13 + j];
            for (int i = 0; i < 8; i++) {
                double s1 = 0, s2 = 0;
                for (int k = 0; k < 8; k++) { s1 += AH[i * 8 + k] * aH[(4 + k) * 13 + 12]; s2 += AT[i * 8 + k] * aH[(4 + k) * 13 + 12]; }
                fb[hIdx + i] += s1; fb[tIdx + i] += s2;
            }
            for (int i = 0; i < 4; i++) fb[i] += aH[i * 13 + 12];
        }
    if (usePrior) {
        for (int i = 0; i < 4; i++) { FH(i, i) += in->cprior[i]; fb[i] += in->cprior[i] * in->cdelta[i]; }
        for (int h = 0; h < N; h++)
            for (int i = 0; i < 8; i++) {
                FH(4 + 8 * h + i, 4 + 8 * h + i) += in->prior[8 * h + i];
                fb[4 + 8 * h + i] += in->prior[8 * h + i] * in->delta_prior[8 * h + i];
            }
    }
    for (int h = 0; h < N; h++) {
        int hIdx = 4 + h * 8;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) FH(i, hIdx + j) = FH(hIdx + j, i);
        for (int t = h + 1; t < N; t++) {
            int tIdx = 4 + t * 8;
            for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) FH(hIdx + i, tIdx + j) += FH(tIdx + j, hIdx + i);
            for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) FH(tIdx + i, hIdx + j) = FH(hIdx + j, tIdx + i);
        }
    }
#undef FH
}

/* addToHessianSC over the points with sel[p] != 0 (NULL: all) + stitchDoubleSC, BA.cpp:1880-2043 */
static void schur_points(orc_ba_window* w, const cmlhip_ba_accum_in* in, const unsigned char* sel, int shift_prior, double* tH, double* tb) {
    const int N = w->N, n = 8 * N + 4, NN = N * N;
    tier_acc* accD = (tier_acc*)zalloc(sizeof(tier_acc) * NN * N);
    tier_acc* accE = (tier_acc*)zalloc(sizeof(tier_acc) * NN);
    tier_acc* accEB = (tier_acc*)zalloc(sizeof(tier_acc) * NN);
    tier_acc accHcc, accbc;
    for (int i = 0; i < NN * N; i++) tier_init(&accD[i], 64);
    for (int i = 0; i < NN; i++) { tier_init(&accE[i], 32); tier_init(&accEB[i], 8); }
    tier_init(&accHcc, 16); tier_init(&accbc, 4);
#ifdef _OPENMP
    /* TIMING BUILD ONLY (see orc_ba_accumulate): one accumulator set per thread over a static split of the points, reduced below */
    tier_acc* const accD0 = accD; tier_acc* const accE0 = accE; tier_acc* const accEB0 = accEB;
    tier_acc* const accHcc0 = &accHcc; tier_acc* const accbc0 = &accbc;
    const int T_ = omp_get_max_threads();
    const size_t per = (size_t)NN * N + 2 * (size_t)NN + 2;
    tier_acc* pool = (tier_acc*)zalloc(sizeof(tier_acc) * per * (size_t)T_);
#pragma omp parallel
    {
    tier_acc* base = pool + per * (size_t)omp_get_thread_num();
    tier_acc* accD = base; tier_acc* accE = base + (size_t)NN * N; tier_acc* accEB = accE + NN;
    tier_acc* accHccP = accEB + NN; tier_acc* accbcP = accHccP + 1;
    for (int i = 0; i < NN * N; i++) tier_init(&accD[i], 64);
    for (int i = 0; i < NN; i++) { tier_init(&accE[i], 32); tier_init(&accEB[i], 8); }
    tier_init(accHccP, 16); tier_init(accbcP, 4);
#define accHcc (*accHccP)
#define accbc (*accbcP)
#pragma omp for schedule(static)
#endif
    for (int p = 0; p < w->P; p++) {
        if (sel && !sel[p]) continue;
        const cmlhip_ba_point* pt = &w->points[p];
        int ngood = 0;
        for (int k = w->by_point_off[p]; k < w->by_point_off[p + 1]; k++) if (w->r_good[w->by_point[k]]) ngood++;
        if (ngood == 0) { w->HdiF[p] = 0; w->bdSumF[p] = 0; continue; }
        float H = w->Hdd_accAF[p] + w->Hdd_accLF[p] + pt->prior;
        if (H < 1e-10) H = 1e-10;
        w->HdiF[p] = (float)(1.0 / H);
        w->bdSumF[p] = w->bd_accAF[p] + w->bd_accLF[p];
        float deltaF = (float)(pt->idepth - (double)pt->idepth_zero);
        if (shift_prior) w->bdSumF[p] += pt->prior * deltaF;     /* shiftPriorToZero, :1904 */
        float Hcd[4];
        for (int i = 0; i < 4; i++) Hcd[i] = w->Hcd_accAF[4 * p + i] + w->Hcd_accLF[4 * p + i];
        tier_update_outer(&accHcc, Hcd, 4, Hcd, 4, w->HdiF[p]);
        tier_update_vec(&accbc, Hcd, 4, w->bdSumF[p] * w->HdiF[p]);
        for (int k1 = w->by_point_off[p]; k1 < w->by_point_off[p + 1]; k1++) {
            int r1 = w->by_point[k1];
            if (!w->r_good[r1]) continue;
            int r1ht = pt->host + w->r_target[r1] * N;
            for (int k2 = w->by_point_off[p]; k2 < w->by_point_off[p + 1]; k2++) {
                int r2 = w->by_point[k2];
                if (!w->r_good[r2]) continue;
                tier_update_outer(&accD[r1ht + w->r_target[r2] * NN], w->JpJdF + 8 * r1, 8, w->JpJdF + 8 * r2, 8, w->HdiF[p]);
            }
            tier_update_outer(&accE[r1ht], w->JpJdF + 8 * r1, 8, Hcd, 4, w->HdiF[p]);
            tier_update_vec(&accEB[r1ht], w->JpJdF + 8 * r1, 8, w->HdiF[p] * w->bdSumF[p]);
        }
    }
#ifdef _OPENMP
#undef accHcc
#undef accbc
    }   /* omp parallel */
    /* reduce the per-thread sets into the first-level slots of the shared ones (finish() folds them): slot by slot over the threads,
       the thread sets of one slot added in thread order */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < per; i++) {
        for (int t = 0; t < T_; t++) {
            tier_acc* base = pool + per * (size_t)t;
            tier_acc* src = &base[i];
            tier_acc* dst = i < (size_t)NN * N ? &accD0[i] : i < (size_t)NN * N + NN ? &accE0[i - (size_t)NN * N]
                          : i < (size_t)NN * N + 2 * (size_t)NN ? &accEB0[i - (size_t)NN * N - NN] : i == per - 2 ? accHcc0 : accbc0;
            const float cnt = src->numIn1 + src->numIn1k + src->numIn1m;
            if (cnt == 0) continue;
            tier_shift(src, 1);
            for (int k = 0; k < src->n; k++) dst->A1m[k] += src->A1m[k];
            dst->numIn1m += cnt;
        }
    }
    free(pool);
#endif
    /* stitchDoubleSC, BA.cpp:1939-2043 */
    memset(tH, 0, sizeof(double) * n * n); memset(tb, 0, sizeof(double) * n);
#ifdef _OPENMP
    /* TIMING BUILD ONLY: the N^2 (i, j) blocks of the stitch over the threads, each into its own copy of H / b, summed afterwards */
    double* const tH_shared = tH; double* const tb_shared = tb;
    const int T2_ = omp_get_max_threads();
    double* tHpool = (double*)zalloc(sizeof(double) * ((size_t)n * n + n) * (size_t)T2_);
#pragma omp parallel
    {
    double* tH = tHpool + ((size_t)n * n + n) * (size_t)omp_get_thread_num();
    double* tb = tH + (size_t)n * n;
#endif
#define TH(i, j) tH[(size_t)(i) * n + (j)]
#ifdef _OPENMP
#pragma omp for collapse(2) schedule(static)
#endif
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            int iIdx = 4 + i * 8, jIdx = 4 + j * 8, ij = i + N * j;
            tier_finish(&accE[ij]); tier_finish(&accEB[ij]);
            const double* AH = in->adHost + 64 * ij; const double* AT = in->adTarget + 64 * ij;
            for (int a = 0; a < 8; a++) {
                for (int c = 0; c < 4; c++) {
                    double s1 = 0, s2 = 0;
                    for (int k = 0; k < 8; k++) { s1 += AH[a * 8 + k] * (double)accE[ij].A1m[k * 4 + c]; s2 += AT[a * 8 + k] * (double)accE[ij].A1m[k * 4 + c]; }
                    TH(iIdx + a, c) += s1; TH(jIdx + a, c) += s2;
                }
                double s1 = 0, s2 = 0;
                for (int k = 0; k < 8; k++) { s1 += AH[a * 8 + k] * (double)accEB[ij].A1m[k]; s2 += AT[a * 8 + k] * (double)accEB[ij].A1m[k]; }
                tb[iIdx + a] += s1; tb[jIdx + a] += s2;
            }
            for (int k = 0; k < N; k++) {
                int kIdx = 4 + k * 8, ijk = ij + k * NN, ik = i + N * k;
                if (tier_finish(&accD[ijk]) == 0) continue;
                double D[64], blk[64];
                for (int a = 0; a < 64; a++) D[a] = (double)accD[ijk].A1m[a];
                const double* AHk = in->adHost + 64 * ik; const double* ATk = in->adTarget + 64 * ik;
                mat8_abt(AH, D, AHk, blk);
                for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) TH(iIdx + a, iIdx + b) += blk[a * 8 + b];
                mat8_abt(AT, D, ATk, blk);
                for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) TH(jIdx + a, kIdx + b) += blk[a * 8 + b];
                mat8_abt(AT, D, AHk, blk);
                for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) TH(jIdx + a, iIdx + b) += blk[a * 8 + b];
                mat8_abt(AH, D, ATk, blk);
                for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) TH(iIdx + a, kIdx + b) += blk[a * 8 + b];
            }
        }
#ifdef _OPENMP
    }   /* omp parallel */
    for (int t = 0; t < T2_; t++) {
        const double* src = tHpool + ((size_t)n * n + n) * (size_t)t;
        for (size_t k = 0; k < (size_t)n * n; k++) tH_shared[k] += src[k];
        for (int k = 0; k < n; k++) tb_shared[k] += src[(size_t)n * n + k];
    }
    free(tHpool);
#endif
    tier_finish(&accHcc); tier_finish(&accbc);
    for (int a = 0; a < 4; a++) { for (int b = 0; b < 4; b++) TH(a, b) = (double)accHcc.A1m[a * 4 + b]; tb[a] = (double)accbc.A1m[a]; }
    for (int h = 0; h < N; h++)
        for (int a = 0; a < 4; a++) for (int b = 0; b < 8; b++) TH(a, 4 + 8 * h + b) = TH(4 + 8 * h + b, a);
#undef TH
    free(accD); free(accE); free(accEB);
}


void orc_ba_accumulate(orc_ba_window* w, const cmlhip_ba_accum_in* in, double* HA, double* bA, double* HL,
                       double* bL, double* Hsc, double* bsc) {
    const int N = w->N, n = 8 * N + 4, NN = N * N;
    approx_acc* acc = (approx_acc*)zalloc(sizeof(approx_acc) * NN);
    double* tH = (double*)zalloc(sizeof(double) * n * n); double* tb = (double*)zalloc(sizeof(double) * n);
    /* ACTIVE, BA.cpp:1368-1371 */
#ifdef _OPENMP
    /* TIMING BUILD ONLY (make fast_omp: bench.py's all-cores CPU baseline; the checker build is serial and literal).  The reference
       loop is serial; here the points are spread over the cores with one accumulator set per thread, summed afterwards — the order of
       the fp32 sums differs from the serial pass, which is why no parity test loads this build. */
    {
        const int T = omp_get_max_threads();
        approx_acc* accT = (approx_acc*)zalloc(sizeof(approx_acc) * NN * (size_t)T);
#pragma omp parallel
        {
            approx_acc* mine = accT + (size_t)omp_get_thread_num() * NN;
#pragma omp for schedule(static)
            for (int p = 0; p < w->P; p++) add_to_hessian_top(w, p, CMLHIP_MODE_ACTIVE, mine, in);
        }
#pragma omp parallel for schedule(static)
        for (int q = 0; q < NN; q++) {
            float Hq[169];
            memset(w->accA + 169 * q, 0, sizeof(float) * 169); w->accA_num[q] = 0;
            for (int t = 0; t < T; t++) {
                approx_acc* a = &accT[(size_t)t * NN + q];
                if (a->num == 0) continue;
                approx_finish(a, Hq);
                for (int i = 0; i < 169; i++) w->accA[169 * q + i] += Hq[i];
                w->accA_num[q] += a->num;
            }
        }
        free(accT);
    }
#else
    for (int p = 0; p < w->P; p++) add_to_hessian_top(w, p, CMLHIP_MODE_ACTIVE, acc, in);
    for (int q = 0; q < NN; q++) { approx_finish(&acc[q], w->accA + 169 * q); w->accA_num[q] = acc[q].num; }
#endif
    stitch_top(w, w->accA, w->accA_num, in, 0, tH, tb);
    if (HA) memcpy(HA, tH, sizeof(double) * n * n);
    if (bA) memcpy(bA, tb, sizeof(double) * n);
    /* LINEARIZED, BA.cpp:1375-1378 */
    memset(acc, 0, sizeof(approx_acc) * NN);
    int any_lin = 1;
#ifdef _OPENMP
    any_lin = 0;                                               /* TIMING BUILD ONLY: a window without LINEARIZED residuals skips the second walk over the points */
    for (int r = 0; r < w->R && !any_lin; r++) any_lin = w->r_lin[r] != 0;
#endif
    for (int p = 0; p < w->P && any_lin; p++) add_to_hessian_top(w, p, CMLHIP_MODE_LINEARIZED, acc, in);
    for (int q = 0; q < NN; q++) { approx_finish(&acc[q], w->accL + 169 * q); w->accL_num[q] = acc[q].num; }
    stitch_top(w, w->accL, w->accL_num, in, 1, tH, tb);
    if (HL) memcpy(HL, tH, sizeof(double) * n * n);
    if (bL) memcpy(bL, tb, sizeof(double) * n);
    free(acc);

    schur_points(w, in, NULL, 1, tH, tb);
    if (Hsc) memcpy(Hsc, tH, sizeof(double) * n * n);
    if (bsc) memcpy(bsc, tb, sizeof(double) * n);
    free(tH); free(tb);
}

/* solveLevenbergMarquardt, BA.cpp:1284-1320 */
int orc_ba_solve(const orc_ba_window* w, double lambda, const double* HA, const double* bA, const double* HL,
                 const double* bL, const double* HM, const double* bM, const double* Hsc, const double* bsc,
                 int optcal, double* x) {
    const int n = 8 * w->N + 4;
    double* H = (double*)zalloc(sizeof(double) * n * n); double* b = (double*)zalloc(sizeof(double) * n);
    double* S = (double*)zalloc(sizeof(double) * n);
    for (int i = 0; i < n * n; i++) H[i] = (HL[i] + (HM ? HM[i] : 0.0)) + HA[i];
    for (int i = 0; i < n; i++) b[i] = ((bL[i] + (bM ? bM[i] : 0.0)) + bA[i]) - bsc[i];
    for (int i = 0; i < n; i++) H[(size_t)i * n + i] *= (1 + lambda);
    double f = 1.0 / (1 + lambda);                            /* 1.0f/(1+lambda), lambda double => double, :1309 */
    for (int i = 0; i < n * n; i++) H[i] -= Hsc[i] * f;
    for (int i = 0; i < n; i++) S[i] = 1.0 / sqrt(H[(size_t)i * n + i] + 10.0);
    int off = optcal ? 0 : 4, m = n - off;
    double* Hs = (double*)zalloc(sizeof(double) * m * m); double* bs = (double*)zalloc(sizeof(double) * m);
    double* xs = (double*)zalloc(sizeof(double) * m);
    for (int i = 0; i < m; i++) {
        for (int j = 0; j < m; j++) Hs[(size_t)i * m + j] = S[off + i] * H[(size_t)(off + i) * n + off + j] * S[off + j];
        bs[i] = S[off + i] * b[off + i];
    }
    int rc = orc_ldlt_solve(Hs, bs, m, xs);
    for (int i = 0; i < n; i++) x[i] = 0;
    for (int i = 0; i < m; i++) x[off + i] = S[off + i] * xs[i];
    free(H); free(b); free(S); free(Hs); free(bs); free(xs);
    return rc;
}

/* resubstitution, BA.cpp:1427-1487 */
int orc_ba_backsub(orc_ba_window* w, const cmlhip_ba_accum_in* in, const double* x) {
    const int N = w->N;
    double* xAd = (double*)zalloc(sizeof(double) * 8 * N * N);
    double cstep[4];
    for (int i = 0; i < 4; i++) cstep[i] = -x[i];
    for (int h = 0; h < N; h++)
        for (int t = 0; t < N; t++) {
            const double* AH = in->adHost + 64 * (h + N * t); const double* AT = in->adTarget + 64 * (h + N * t);
            for (int j = 0; j < 8; j++) {
                double s = 0;
                for (int i = 0; i < 8; i++) s += x[4 + 8 * h + i] * AH[i * 8 + j];
                double s2 = 0;
                for (int i = 0; i < 8; i++) s2 += x[4 + 8 * t + i] * AT[i * 8 + j];
                xAd[8 * (N * h + t) + j] = s + s2;
            }
        }
    int bad = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : bad)
#endif
    for (int p = 0; p < w->P; p++) {
        int ngood = 0;
        for (int k = w->by_point_off[p]; k < w->by_point_off[p + 1]; k++) if (w->r_good[w->by_point[k]]) ngood++;
        if (ngood == 0) { w->step[p] = 0; continue; }
        double b = (double)w->bdSumF[p];
        b -= orc_eig_calib_dot(cstep, w->Hcd_accAF + 4 * p, w->Hcd_accLF + 4 * p);     /* :1470, Eigen's order */
        int host = w->points[p].host;
        for (int k = w->by_point_off[p]; k < w->by_point_off[p + 1]; k++) {
            int r = w->by_point[k];
            if (!w->r_good[r]) continue;
            const double* xa = xAd + 8 * (host * N + w->r_target[r]);
            b -= orc_eig_row8_dot_cast(xa, w->JpJdF + 8 * r);                            /* :1478, Eigen's order */
        }
        w->step[p] = -b * (double)w->HdiF[p];
        if (!isfinite(w->step[p])) bad++;
    }
    free(xAd);
    return bad ? CMLHIP_ERR_NONFINITE : 0;
}

void orc_ba_backup_points(orc_ba_window* w) {           /* BA.cpp:919-922 */
    for (int p = 0; p < w->P; p++) w->idepth_backup[p] = (float)w->points[p].idepth;
}

void orc_ba_restore_points(orc_ba_window* w) {          /* loadSateBackup, BA.cpp:938-942 */
    for (int p = 0; p < w->P; p++) { w->points[p].idepth = (double)w->idepth_backup[p]; w->points[p].idepth_zero = w->idepth_backup[p]; }
}

void orc_ba_step_points(orc_ba_window* w, float sums[3]) {   /* BA.cpp:976-994 */
    float sumID = 0, sumNID = 0, numID = 0;
    for (int p = 0; p < w->P; p++) {
        double nid = (double)w->idepth_backup[p] + w->step[p];
        if (isfinite(nid) && nid > 0) w->points[p].idepth = nid;
        else continue;
        sumID = (float)((double)sumID + w->step[p] * w->step[p]);
        sumNID = (float)((double)sumNID + fabs((double)w->idepth_backup[p]));
        numID++;
        w->points[p].idepth_zero = (float)w->points[p].idepth;
    }
    if (sums) { sums[0] = sumID; sums[1] = sumNID; sums[2] = numID; }
}

/* ------------------------------------------------------------------ frame algebra (host side of the reference) */
static void frame_update_pre(orc_frame* f) {           /* PRE_worldToCam = exp(w2c_leftEps) * evalPT, DSOFrame.h:119-120 */
    orc_se3 e;
    orc_se3_exp(f->state_scaled, &e);
    orc_se3_mul(&e, &f->w2c_eval, &f->PRE_w2c);
    orc_se3_inv(&f->PRE_w2c, &f->PRE_c2w);
}
void orc_frame_set_state(orc_frame* f, const double st[10], const orc_scales* s) {   /* DSOFrame.h:110-124 */
    double tmp[10];
    memcpy(tmp, st, sizeof tmp);
    memcpy(f->state, tmp, sizeof tmp);
    for (int i = 0; i < 3; i++) { f->state_scaled[i] = s->trans * tmp[i]; f->state_scaled[3 + i] = s->rot * tmp[3 + i]; }
    f->state_scaled[6] = s->a * tmp[6]; f->state_scaled[7] = s->b * tmp[7];
    f->state_scaled[8] = s->a * tmp[8]; f->state_scaled[9] = s->b * tmp[9];
    frame_update_pre(f);
}
void orc_frame_set_state_scaled(orc_frame* f, const double ss[10], const orc_scales* s) {   /* DSOFrame.h:126-142 */
    double tmp[10];
    memcpy(tmp, ss, sizeof tmp);
    memcpy(f->state_scaled, tmp, sizeof tmp);
    for (int i = 0; i < 3; i++) { f->state[i] = tmp[i] / s->trans; f->state[3 + i] = tmp[3 + i] / s->rot; }
    f->state[6] = tmp[6] / s->a; f->state[7] = tmp[7] / s->b;
    f->state[8] = tmp[8] / s->a; f->state[9] = tmp[9] / s->b;
    frame_update_pre(f);
}
void orc_frame_set_state_zero(orc_frame* f, const double sz[10], const orc_scales* s) {     /* DSOFrame.h:154-186 */
    double tmp[10];
    memcpy(tmp, sz, sizeof tmp);
    memcpy(f->state_zero, tmp, sizeof tmp);
    orc_se3 Ti; orc_se3_inv(&f->w2c_eval, &Ti);
    for (int i = 0; i < 6; i++) {
        double eps[6] = {0, 0, 0, 0, 0, 0}, lp[6], lm[6];
        orc_se3 Ep, Em, A, B;
        eps[i] = 1e-3; orc_se3_exp(eps, &Ep);
        eps[i] = -1e-3; orc_se3_exp(eps, &Em);
        orc_se3_mul(&f->w2c_eval, &Ep, &A); orc_se3_mul(&A, &Ti, &A); orc_se3_log(&A, lp);
        orc_se3_mul(&f->w2c_eval, &Em, &B); orc_se3_mul(&B, &Ti, &B); orc_se3_log(&B, lm);
        for (int k = 0; k < 6; k++) f->ns_pose[i * 6 + k] = (lp[k] - lm[k]) / (2e-3);
    }
    {
        orc_se3 Pp = f->w2c_eval, Pm = f->w2c_eval;
        double lp[6], lm[6];
        for (int k = 0; k < 3; k++) { Pp.t[k] *= 1.00001; Pm.t[k] /= 1.00001; }
        orc_se3_mul(&Pp, &Ti, &Pp); orc_se3_mul(&Pm, &Ti, &Pm);
        orc_se3_log(&Pp, lp); orc_se3_log(&Pm, lm);
        for (int k = 0; k < 6; k++) f->ns_scale[k] = (lp[k] - lm[k]) / (2e-3);
    }
    memset(f->ns_affine, 0, sizeof f->ns_affine);
    f->ns_affine[0] = 1;                                                  /* col 0 = (1,0,..) */
    f->ns_affine[4 + 1] = (double)expf((float)(f->state_zero[6] * s->a)) * f->ab_exposure;   /* col 1 = (0, exp(a0)*t) */
}
void orc_frame_set_evalpt_scaled(orc_frame* f, const orc_se3* w2c, double aff_a, double aff_b, const orc_scales* s) {
    double init[10] = {0, 0, 0, 0, 0, 0, aff_a, aff_b, 0, 0};             /* DSOFrame.h:99-108 */
    f->w2c_eval = *w2c;
    orc_frame_set_state_scaled(f, init, s);
    orc_frame_set_state_zero(f, f->state, s);
}
/* DSOFramePrecomputed::precompute, DSOFrame.h:259-273.  trialRefToTarget =
 * cameraOf(host.PRE_w2c).to(cameraOf(target.PRE_w2c)) = target * host^-1 (Camera.h:289-299). */
void orc_frame_precompute(const orc_frame* host, const orc_frame* target, cmlhip_ba_pair* out) {
    orc_se3 ll, hi;
    orc_se3_mul(&target->PRE_w2c, &host->PRE_c2w, &ll);
    orc_se3_matrix(&ll, out->R);
    memcpy(out->t, ll.t, sizeof ll.t);
    orc_se3_inv(&host->w2c_eval, &hi);
    orc_se3_mul(&target->w2c_eval, &hi, &ll);
    orc_se3_matrix(&ll, out->R0);
    memcpy(out->t0, ll.t, sizeof ll.t);
    /* aff_g2l() = Exposure(ab_exposure, state_scaled[6], state_scaled[7]), DSOFrame.h:189-191 */
    orc_exposure_to(host->state_scaled[6], host->state_scaled[7], host->ab_exposure,
                    target->state_scaled[6], target->state_scaled[7], target->ab_exposure, &out->aff_a, &out->aff_b);
}

/* computeAdjoints, BA.cpp:1062-1097 */
void orc_ba_compute_adjoints(const orc_frame* fr, int N, const orc_scales* s, double* adHost, double* adTarget) {
    for (int h = 0; h < N; h++)
        for (int t = 0; t < N; t++) {
            orc_se3 hi, ht;
            orc_se3_inv(&fr[h].w2c_eval, &hi);
            orc_se3_mul(&fr[t].w2c_eval, &hi, &ht);
            double Adj[36], la, lb;
            orc_se3_adj(&ht, Adj);
            orc_exposure_to(fr[h].state_zero[6] * s->a, fr[h].state_zero[7] * s->b, fr[h].ab_exposure,
                            fr[t].state_zero[6] * s->a, fr[t].state_zero[7] * s->b, fr[t].ab_exposure, &la, &lb);
            double* AH = adHost + 64 * (h + t * N); double* AT = adTarget + 64 * (h + t * N);
            memset(AH, 0, 64 * sizeof(double)); memset(AT, 0, 64 * sizeof(double));
            for (int i = 0; i < 6; i++) {
                for (int j = 0; j < 6; j++) AH[i * 8 + j] = -Adj[j * 6 + i];
                AT[i * 8 + i] = 1;
            }
            AT[6 * 8 + 6] = -la; AH[6 * 8 + 6] = la; AT[7 * 8 + 7] = -1; AH[7 * 8 + 7] = la;
            const double rs[8] = {s->trans, s->trans, s->trans, s->rot, s->rot, s->rot, s->a, s->b};
            for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) { AH[i * 8 + j] *= rs[i]; AT[i * 8 + j] *= rs[i]; }
        }
}

/* computeDelta (frame part), BA.cpp:1103-1177 */
void orc_ba_compute_delta(orc_frame* fr, int N, const double* adHost, const double* adTarget, int optA, int optB,
                          float* adHTdeltaF) {
    for (int h = 0; h < N; h++)
        for (int t = 0; t < N; t++) {
            int idx = h + t * N;
            for (int j = 0; j < 8; j++) {
                double s = 0, s2 = 0;
                for (int i = 0; i < 8; i++) {
                    s += (fr[h].state[i] - fr[h].state_zero[i]) * adHost[64 * idx + i * 8 + j];
                    s2 += (fr[t].state[i] - fr[t].state_zero[i]) * adTarget[64 * idx + i * 8 + j];
                }
                adHTdeltaF[8 * idx + j] = (float)(s + s2);
            }
        }
    const float rotPrior = 1e11f, transPrior = 1e10f, affBPrior = 1e14f, affAPrior = 1e14f;
    float modeA = 1e12f, modeB = 1e8f;
    if (!optA) modeA = -1;
    if (!optB) modeB = -1;
    for (int f = 0; f < N; f++) {
        double* p = fr[f].prior;
        memset(p, 0, 8 * sizeof(double));
        if (fr[f].keyid == 0) {
            p[0] = p[1] = p[2] = transPrior; p[3] = p[4] = p[5] = rotPrior; p[6] = affAPrior; p[7] = affBPrior;
        } else {
            p[6] = modeA < 0 ? affAPrior : modeA;
            p[7] = modeB < 0 ? affBPrior : modeB;
        }
        for (int i = 0; i < 8; i++) {
            fr[f].delta[i] = fr[f].state[i] - fr[f].state_zero[i];
            fr[f].delta_prior[i] = fr[f].state[i] - fr[f].prior_zero[i];
        }
    }
}

/* computeNullspaces, BA.cpp:2365-2417: the 7 vectors orthogonalize() uses (6 pose + scale) */
void orc_ba_nullspaces(const orc_frame* fr, int N, const orc_scales* s, double* out) {
    const int n = 8 * N + 4;
    memset(out, 0, sizeof(double) * n * 7);
    for (int i = 0; i < 6; i++)
        for (int f = 0; f < N; f++)
            for (int k = 0; k < 6; k++) {
                double v = fr[f].ns_pose[i * 6 + k];
                v *= (k < 3) ? 1.0 / s->trans : 1.0 / s->rot;
                out[i * n + 4 + 8 * f + k] = v;
            }
    for (int f = 0; f < N; f++)
        for (int k = 0; k < 6; k++) {
            double v = fr[f].ns_scale[k];
            v *= (k < 3) ? 1.0 / s->trans : 1.0 / s->rot;
            out[6 * n + 4 + 8 * f + k] = v;
        }
}

/* ------------------------------------------------------------------ marginalisation */
/* fixLinearization, BA.cpp:2210-2238: res_toZeroF = resF - [JI*Jp Ja]*delta (float, statement order of the SSE code) */
void orc_ba_fix_linearization(orc_ba_window* w, int r, const cmlhip_ba_accum_in* in) {
    const cmlhip_ba_point* pt = &w->points[w->r_point[r]];
    const float* J = w->efsJ + 74 * (size_t)r;
    const float* dp = in->adHTdeltaF + 8 * w->pair_of[r];
    const float deltaF = (float)(pt->idepth - (double)pt->idepth_zero);
    /* J.Jpdxi[k].dot(dp.head<6>()) + J.Jpdc[k].dot(mCDeltaF.cast<float>()) + J.Jpdd[k] * deltaF: the cast stays inside the dot */
    const float Jp_delta_x = orc_eig_jp_delta(J + O_XI0, dp, J + O_C0, in->cdelta, J[O_DD], deltaF, 1);
    const float Jp_delta_y = orc_eig_jp_delta(J + O_XI1, dp, J + O_C1, in->cdelta, J[O_DD + 1], deltaF, 1);
    for (int i = 0; i < 8; i++) {
        float rtz = J[O_RES + i];
        rtz = rtz - J[O_JI0 + i] * Jp_delta_x;
        rtz = rtz - J[O_JI1 + i] * Jp_delta_y;
        rtz = rtz - J[O_JAB0 + i] * dp[6];
        rtz = rtz - J[O_JAB1 + i] * dp[7];
        w->res_toZeroF[8 * r + i] = rtz;
    }
    w->r_lin[r] = 1;
}

/* the residual loop of tryMarginalize for the points that will be marginalised, BA.cpp:2291-2304:
 * resetOOB, linearize, isLinearized = false, applyRes(true), fixLinearization of the good ones.  Returns #good. */
int orc_ba_relinearize_points(orc_ba_window* w, int n, const int* pts, const cmlhip_ba_accum_in* in) {
    int ngood = 0;
    for (int k = 0; k < n; k++) {
        const int p = pts[k];
        for (int q = w->by_point_off[p]; q < w->by_point_off[p + 1]; q++) {
            const int r = w->by_point[q];
            w->r_new_energy[r] = w->r_energy[r] = 0;                /* resetOOB, DSOResidual.h:81-86 */
            w->r_new_state[r] = CMLHIP_RES_OUTLIER;
            w->r_state[r] = CMLHIP_RES_IN;
            orc_ba_linearize_one(w, r);
            w->r_lin[r] = 0;
            apply_one(w, r, 1);
            if (w->r_good[r]) { orc_ba_fix_linearization(w, r, in); ngood++; }
        }
    }
    return ngood;
}

/* marginalizePointsF without the bookkeeping, BA.cpp:2466-2513: M, Mb = stitchDoubleTop of the MARGINALIZED-mode
 * accumulation of the listed points (usePrior = false), Msc, Mbsc = their Schur complement (shiftPriorToZero = false).
 * The caller adds 0.25 * (M - Msc), 0.25 * (Mb - Mbsc) to the prior (:2502-2507). */
void orc_ba_marginalize_points(orc_ba_window* w, int n, const int* pts, const cmlhip_ba_accum_in* in,
                               double* M, double* Mb, double* Msc, double* Mbsc) {
    const int N = w->N, nn = 8 * N + 4, NN = N * N;
    approx_acc* acc = (approx_acc*)zalloc(sizeof(approx_acc) * NN);
    unsigned char* sel = (unsigned char*)zalloc(w->P > 0 ? w->P : 1);
    for (int k = 0; k < n; k++) {
        sel[pts[k]] = 1;
        add_to_hessian_top(w, pts[k], CMLHIP_MODE_MARGINALIZED, acc, in);
    }
    for (int q = 0; q < NN; q++) { approx_finish(&acc[q], w->accA + 169 * q); w->accA_num[q] = acc[q].num; }
    stitch_top(w, w->accA, w->accA_num, in, 0, M, Mb);
    schur_points(w, in, sel, 0, Msc, Mbsc);
    free(acc); free(sel);
}

/* marginalizeFrame, the algebra of BA.cpp:483-558: HM, bM are (8N+4)^2 / (8N+4) row-major on entry, (8N-4)^2 / (8N-4) on exit */
void orc_ba_marginalize_frame(double* HM, double* bM, int N, int frame, const double prior[8], const double delta_prior[8]) {
    const int odim = 8 * N + 4, ndim = odim - 8;
    double* H = (double*)zalloc(sizeof(double) * odim * odim);
    double* b = (double*)zalloc(sizeof(double) * odim);
    int* perm = (int*)zalloc(sizeof(int) * odim);
    /* move the frame's 8 rows/cols to the end, order of the rest unchanged (:489-508) */
    const int io = 8 * frame + 4;
    int k = 0;
    for (int i = 0; i < odim; i++) if (i < io || i >= io + 8) perm[k++] = i;
    for (int i = 0; i < 8; i++) perm[k++] = io + i;
    for (int i = 0; i < odim; i++) {
        b[i] = bM[perm[i]];
        for (int j = 0; j < odim; j++) H[(size_t)i * odim + j] = HM[(size_t)perm[i] * odim + perm[j]];
    }
    for (int i = 0; i < 8; i++) {                                      /* :511-513 */
        H[(size_t)(ndim + i) * odim + ndim + i] += prior[i];
        b[ndim + i] += prior[i] * delta_prior[i];
    }
    double* SVec = (double*)zalloc(sizeof(double) * odim);
    for (int i = 0; i < odim; i++) SVec[i] = sqrt(fabs(H[(size_t)i * odim + i]) + 10.0);     /* :520-521 */
    for (int i = 0; i < odim; i++) {
        for (int j = 0; j < odim; j++) H[(size_t)i * odim + j] = (1.0 / SVec[i]) * H[(size_t)i * odim + j] * (1.0 / SVec[j]);
        b[i] = (1.0 / SVec[i]) * b[i];
    }
    double hpi[64], hinv[64];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) hpi[i * 8 + j] = H[(size_t)(ndim + i) * odim + ndim + j];
    for (int i = 0; i < 64; i++) hpi[i] = 0.5f * (hpi[i] + hpi[i]);                          /* :533 (sic: hpi + hpi) */
    orc_inverse(hpi, 8, hinv);
    for (int i = 0; i < 64; i++) hinv[i] = 0.5f * (hinv[i] + hinv[i]);
    /* bli = bottomLeft^T * hpi (ndim x 8); top-left -= bli * bottomLeft; b.head -= bli * b.tail (:538-541) */
    double* bli = (double*)zalloc(sizeof(double) * ndim * 8);
    for (int i = 0; i < ndim; i++)
        for (int j = 0; j < 8; j++) {
            double s = 0;
            for (int q = 0; q < 8; q++) s += H[(size_t)(ndim + q) * odim + i] * hinv[q * 8 + j];
            bli[i * 8 + j] = s;
        }
    for (int i = 0; i < ndim; i++) {
        for (int j = 0; j < ndim; j++) {
            double s = 0;
            for (int q = 0; q < 8; q++) s += bli[i * 8 + q] * H[(size_t)(ndim + q) * odim + j];
            H[(size_t)i * odim + j] -= s;
        }
        double s = 0;
        for (int q = 0; q < 8; q++) s += bli[i * 8 + q] * b[ndim + q];
        b[i] -= s;
    }
    for (int i = 0; i < odim; i++) {                                   /* unscale, :544-545 */
        for (int j = 0; j < odim; j++) H[(size_t)i * odim + j] = SVec[i] * H[(size_t)i * odim + j] * SVec[j];
        b[i] = SVec[i] * b[i];
    }
    for (int i = 0; i < ndim; i++) {                                   /* :548-549 */
        for (int j = 0; j < ndim; j++) HM[(size_t)i * ndim + j] = 0.5 * (H[(size_t)i * odim + j] + H[(size_t)j * odim + i]);
        bM[i] = b[i];
    }
    free(H); free(b); free(perm); free(SVec); free(bli);
}

/* calcMEnergy, BA.cpp:2095-2117 (the forceAccept early-out is the caller's) */
double orc_ba_calc_m_energy(const double* HM, const double* bM, int n, const double* delta) {
    double e = 0;
    for (int i = 0; i < n; i++) {
        double s = 2 * bM[i];
        for (int j = 0; j < n; j++) s += HM[(size_t)i * n + j] * delta[j];
        e += delta[i] * s;
    }
    return fabs(e);
}

/* calcLEnergy, BA.cpp:2119-2208: prior terms + sum over the LINEARIZED good residuals of (2 res_toZero + J delta) . J delta
 * (fp32, accumulated here in double: Accumulator11 is a tiered float sum) + per-point prior term */
double orc_ba_calc_l_energy(const orc_ba_window* w, const cmlhip_ba_accum_in* in, int* num_out) {
    double F = 0;
    for (int i = 0; i < 8 * w->N; i++) F += in->delta_prior[i] * in->prior[i] * in->delta_prior[i];
    for (int i = 0; i < 4; i++) F += in->cdelta[i] * 5e9 * in->cdelta[i];
    double E = 0;
    int num = 0;
    for (int p = 0; p < w->P; p++) {
        const cmlhip_ba_point* pt = &w->points[p];
        const float dd = (float)(pt->idepth - (double)pt->idepth_zero);
        for (int q = w->by_point_off[p]; q < w->by_point_off[p + 1]; q++) {
            const int r = w->by_point[q];
            if (!w->r_lin[r] || !w->r_good[r]) continue;
            num++;
            const float* J = w->efsJ + 74 * (size_t)r;
            const float* dp = in->adHTdeltaF + 8 * w->pair_of[r];
            const float Jpx = orc_eig_jp_delta(J + O_XI0, dp, J + O_C0, in->cdelta, J[O_DD], dd, 0);      /* :2166-2172, Vector4f dc */
            const float Jpy = orc_eig_jp_delta(J + O_XI1, dp, J + O_C1, in->cdelta, J[O_DD + 1], dd, 0);
            for (int i = 0; i < 8; i++) {
                float Jdelta = J[O_JI0 + i] * Jpx;
                Jdelta = Jdelta + J[O_JI1 + i] * Jpy;
                Jdelta = Jdelta + J[O_JAB0 + i] * dp[6];
                Jdelta = Jdelta + J[O_JAB1 + i] * dp[7];
                float r0 = w->res_toZeroF[8 * r + i];
                r0 = r0 + r0;
                r0 = r0 + Jdelta;
                E += (double)(Jdelta * r0);
            }
        }
        E += (double)(dd * dd * pt->prior);
    }
    if (num_out) *num_out = num;
    return E + F;
}
