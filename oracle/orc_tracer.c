/* orc_tracer.c — CPU restatement of DSOTracer::trace and optimizeImmaturePoint / linearizeResidual
 * (src/cml/optimization/dso/DSOTracer.cpp:280-494,585-823).  TEST INFRASTRUCTURE ONLY (see cml_oracle.h).
 * scalar_t is double (types.h:362-368); the float places of the reference (`float ptx`, Vector2f interpolation,
 * float& Hdd/bd, float energies of optimizeImmaturePoint) are kept float.  Parity unpinned: the reference has no
 * test or fixture for this path; the restatement follows the statements line by line. */
#include <math.h>
#include <string.h>
#include "cml_oracle.h"

static const int star8[16] = {0, -2, -1, -1, 1, -1, -2, 0, 0, 0, 2, 0, -1, 1, 0, 2};   /* types.h:1381-1393 */

static int inside(double x, double y, int w, int h, double pad) {                      /* Frame.h:136-138 */
    return x >= pad && y >= pad && x < (double)w - pad && y < (double)h - pad;
}

/* Array2D<T>::interpolate, image/Array2D.h:242-262, on channel c of the AoS3 gradient image (channel 0 = gray) */
static float bil(const float* aos3, int w, float x, float y, int c) {
    const int ix = (int)x, iy = (int)y;
    const float dx = x - (float)ix, dy = y - (float)iy, dxdy = dx * dy;
    const int i1 = iy * w + ix, i2 = i1 + w;
    return aos3[3 * (size_t)(i1 + 0) + c] * (1 - dx - dy + dxdy) + aos3[3 * (size_t)(i1 + 1) + c] * (dx - dxdy)
         + aos3[3 * (size_t)(i2 + 0) + c] * (dy - dxdy) + aos3[3 * (size_t)(i2 + 1) + c] * dxdy;
}

/* DSOTracer::trace, DSOTracer.cpp:585-823.  Returns the status it returns. */
int orc_trace_point(const float* aos3, int w, int h, const cmlhip_trace_pair* pr_, const cmlhip_tracer_params* P,
                    cmlhip_immature_point* s) {
    const double* M = pr_->KRKi; const double* Kt = pr_->Kt;
    if (s->last_status == CMLHIP_IPS_OOB) return CMLHIP_IPS_OOB;                           /* :601-604 */
    const double cx = (double)s->x, cy = (double)s->y;
    /* Vector3 pr = KRKi * Vector3(x, y, 1), :608 — Eigen's order for a double 3x3 * 3-vector: the SSE2 packet rows 0-1
     * accumulate column by column, the scalar row 2 reduces as e0 + (e1 + e2) (pinned: orc_eig_matvec3d) */
    const double pr[3] = {(M[0] * cx + M[1] * cy) + M[2] * 1.0, (M[3] * cx + M[4] * cy) + M[5] * 1.0, M[6] * cx + (M[7] * cy + M[8] * 1.0)};
    const double maxPixSearch = (double)(w + h) * P->max_pix_search;                       /* :611 */
    const double ptpMin[3] = {pr[0] + Kt[0] * s->idepth_min, pr[1] + Kt[1] * s->idepth_min, pr[2] + Kt[2] * s->idepth_min};
    double minx = ptpMin[0] / ptpMin[2], miny = ptpMin[1] / ptpMin[2];
#define SET_OOB() do { s->last_uv[0] = -1; s->last_uv[1] = -1; s->last_pixel_interval = 0; s->last_status = CMLHIP_IPS_OOB; return CMLHIP_IPS_OOB; } while (0)
    if (!inside(minx, miny, w, h, 4)) SET_OOB();                                            /* :620-626 */
    double maxx, maxy, pixelInterval;
    if (isfinite(s->idepth_max)) {
        const double q[3] = {pr[0] + Kt[0] * s->idepth_max, pr[1] + Kt[1] * s->idepth_max, pr[2] + Kt[2] * s->idepth_max};
        maxx = q[0] / q[2]; maxy = q[1] / q[2];
        if (!inside(maxx, maxy, w, h, 5)) SET_OOB();
        pixelInterval = sqrt((maxx - minx) * (maxx - minx) + (maxy - miny) * (maxy - miny));
        if (pixelInterval < P->max_slack_interval) {                                       /* :646-652 */
            s->last_uv[0] = (maxx + minx) / 2.0; s->last_uv[1] = (maxy + miny) / 2.0;
            s->last_pixel_interval = pixelInterval; s->last_status = CMLHIP_IPS_SKIPPED;
            return CMLHIP_IPS_SKIPPED;
        }
    } else {
        pixelInterval = maxPixSearch;
        const double q[3] = {pr[0] + Kt[0] * 0.01, pr[1] + Kt[1] * 0.01, pr[2] + Kt[2] * 0.01};
        maxx = q[0] / q[2]; maxy = q[1] / q[2];
        const double dirx = maxx - minx, diry = maxy - miny;
        const double inv = 1.0 / sqrt(dirx * dirx + diry * diry);
        maxx = minx + pixelInterval * dirx * inv; maxy = miny + pixelInterval * diry * inv;
        if (!inside(maxx, maxy, w, h, 5)) SET_OOB();
    }
    if (!(s->idepth_min < 0 || (ptpMin[2] > 0.75 && ptpMin[2] < 1.5))) SET_OOB();            /* :682-688 */
    double dx = P->trace_step_size * (maxx - minx), dy = P->trace_step_size * (maxy - miny);
    const double* G = s->gradH;
    const double a = dx * (G[0] * dx + G[1] * dy) + dy * (G[2] * dx + G[3] * dy);
    const double b = dy * (G[0] * dy + G[1] * (-dx)) + (-dx) * (G[2] * dy + G[3] * (-dx));
    double errorInPixel = (double)0.2f + (double)0.2f * (a + b) / a;                       /* :697 */
    if (errorInPixel * P->min_improvement_factor > pixelInterval && isfinite(s->idepth_max)) {
        s->last_uv[0] = (maxx + minx) / 2.0; s->last_uv[1] = (maxy + miny) / 2.0;
        s->last_pixel_interval = pixelInterval; s->last_status = CMLHIP_IPS_BADCONDITION;
        return CMLHIP_IPS_BADCONDITION;
    }
    if (errorInPixel > 10) errorInPixel = 10;
    dx /= pixelInterval; dy /= pixelInterval;
    if (pixelInterval > maxPixSearch) { maxx += maxPixSearch * dx; maxy += maxPixSearch * dy; pixelInterval = maxPixSearch; }
    int numSteps = (int)((double)1.9999f + pixelInterval / P->trace_step_size);
    const double randShift = minx * 1000 - floor(minx * 1000);
    float ptx = (float)(minx - randShift * dx), pty = (float)(miny - randShift * dy);
    double rot[16];
    for (int i = 0; i < 8; i++) {
        rot[2 * i] = M[0] * (double)star8[2 * i] + M[1] * (double)star8[2 * i + 1];
        rot[2 * i + 1] = M[3] * (double)star8[2 * i] + M[4] * (double)star8[2 * i + 1];
    }
    if (!isfinite(dx) || !isfinite(dy)) { s->last_pixel_interval = 0; s->last_uv[0] = -1; s->last_uv[1] = -1; s->last_status = CMLHIP_IPS_OOB; return CMLHIP_IPS_OOB; }
    double errors[100];
    double bestU = 0, bestV = 0, bestEnergy = 1e10;
    int bestIdx = -1;
    if (numSteps >= 100) numSteps = 99;
    for (int i = 0; i < numSteps; i++) {
        double energy = 0;
        for (int idx = 0; idx < 8; idx++) {
            const double px = (double)ptx + rot[2 * idx], py = (double)pty + rot[2 * idx + 1];
            if (!inside(px, py, w, h, 3)) { energy += 1e5; continue; }
            const double hit = (double)bil(aos3, w, (float)px, (float)py, 0);
            const double ref = (double)s->gray[idx];
            const double residual = hit - (pr_->aff_a * ref + pr_->aff_b);
            const double hw = fabs(residual) < P->huber_th ? 1 : P->huber_th / fabs(residual);
            energy += hw * residual * residual * (2 - hw);
        }
        errors[i] = energy;
        if (energy < bestEnergy) { bestU = ptx; bestV = pty; bestEnergy = energy; bestIdx = i; }
        ptx = (float)((double)ptx + dx); pty = (float)((double)pty + dy);
    }
    double secondBest = 1e10;
    for (int i = 0; i < numSteps; i++)
        if (((double)i < (double)bestIdx - P->min_trace_test_radius || (double)i > (double)bestIdx + P->min_trace_test_radius) && errors[i] < secondBest)
            secondBest = errors[i];
    const double newQuality = secondBest / bestEnergy;
    if (newQuality < s->quality || numSteps > 10) s->quality = newQuality;
    if (bestEnergy >= s->energy_th * P->extra_slack_on_th) {                                /* :777-789 */
        s->last_pixel_interval = 0; s->last_uv[0] = -1; s->last_uv[1] = -1;
        if (s->last_status == CMLHIP_IPS_OUTLIER) { s->last_status = CMLHIP_IPS_OOB; return CMLHIP_IPS_OOB; }
        s->last_status = CMLHIP_IPS_OUTLIER;
        return CMLHIP_IPS_OUTLIER;
    }
    if (dx * dx > dy * dy) {
        s->idepth_min = (pr[2] * (bestU - errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
        s->idepth_max = (pr[2] * (bestU + errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
    } else {
        s->idepth_min = (pr[2] * (bestV - errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
        s->idepth_max = (pr[2] * (bestV + errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
    }
    if (s->idepth_min > s->idepth_max) { const double t = s->idepth_min; s->idepth_min = s->idepth_max; s->idepth_max = t; }
    s->last_pixel_interval = 2 * errorInPixel;
    s->last_uv[0] = bestU; s->last_uv[1] = bestV;
    s->last_status = CMLHIP_IPS_GOOD;
    return CMLHIP_IPS_GOOD;
#undef SET_OOB
}

typedef struct { int state, new_state; double energy, new_energy; } tmp_res;       /* ImmaturePointTemporaryResidual */

/* DSOTracer::linearizeResidual, DSOTracer.cpp:406-494 */
static double lin_res(const float* aos3, int w, int h, const double K[4], const cmlhip_activation_pair* ht, const cmlhip_tracer_params* P,
                      const cmlhip_immature_point* pt, float slack, tmp_res* r, float* Hdd, float* bd, float idepth) {
    if (r->state == CMLHIP_RES_OOB) { r->new_state = CMLHIP_RES_OOB; return r->energy; }
    float energyLeft = 0;
    for (int idx = 0; idx < 8; idx++) {
        const double ux = ((double)pt->x + (double)star8[2 * idx] - K[2]) * (1.0 / K[0]);      /* undistort: PinholeUndistorter, InternalCalibration.h:42-47 */
        const double uy = ((double)pt->y + (double)star8[2 * idx + 1] - K[3]) * (1.0 / K[1]);
        const double p[3] = {(ht->R[0] * ux + ht->R[1] * uy + ht->R[2] * 1.0) + ht->t[0] * (double)idepth,
                             (ht->R[3] * ux + ht->R[4] * uy + ht->R[5] * 1.0) + ht->t[1] * (double)idepth,
                             (ht->R[6] * ux + ht->R[7] * uy + ht->R[8] * 1.0) + ht->t[2] * (double)idepth};
        const double upx = p[0] / p[2], upy = p[1] / p[2];
        const double projx = upx * K[0] + K[2], projy = upy * K[1] + K[3];
        const double drescale = 1.0 / p[2];
        if (!inside(projx, projy, w, h, 1) || drescale <= 0) { r->new_state = CMLHIP_RES_OOB; return r->energy; }   /* :436-440 */
        const float g0 = bil(aos3, w, (float)projx, (float)projy, 0), g1 = bil(aos3, w, (float)projx, (float)projy, 1),
                    g2 = bil(aos3, w, (float)projx, (float)projy, 2);
        const double hitColor = (double)g0;
        const float* gt = pt->dpatch + 3 * idx;
        const double groundtruth = ht->aff_a * (double)gt[0] + ht->aff_b;
        const double residual = hitColor - groundtruth;
        double hw = fabs(residual) < P->huber_th ? 1 : P->huber_th / fabs(residual);
        const float sq = gt[1] * gt[1] + gt[2] * gt[2];                                        /* Vector3f.tail<2>().squaredNorm(): float */
        const double weight = sqrt(P->outlier_th_sum_component / (P->outlier_th_sum_component + (double)sq));
        energyLeft = (float)((double)energyLeft + weight * weight * hw * residual * residual * (2 - hw));
        const double dxInterp = (double)g1 * K[0], dyInterp = (double)g2 * K[1];
        const double d_idepth = dxInterp * drescale * (ht->t[0] - ht->t[2] * upx) + dyInterp * drescale * (ht->t[1] - ht->t[2] * upy);
        hw *= weight * weight;
        *Hdd = (float)((double)*Hdd + (hw * d_idepth) * d_idepth);
        *bd = (float)((double)*bd + (hw * residual) * d_idepth);
    }
    if ((double)energyLeft > pt->energy_th * (double)slack) {
        energyLeft = (float)(pt->energy_th * (double)slack);
        r->new_state = CMLHIP_RES_OUTLIER;
    } else r->new_state = CMLHIP_RES_IN;
    r->new_energy = energyLeft;
    return energyLeft;
}

/* DSOTracer::optimizeImmaturePoint, DSOTracer.cpp:280-404 (map bookkeeping left to the caller).  images[t] = level-0 AoS3 gradient
 * image of frame t; pairs[h*N+t]; res_state[t] = final state_state (-1 for the host).  Returns 1 / 0 / -1. */
int orc_optimize_immature_point(int N, const float* const* images, int w, int h, const double K[4], const cmlhip_activation_pair* pairs,
                                const cmlhip_tracer_params* P, int min_obs, const cmlhip_immature_point* pt, float* idepth_out, int* res_state) {
    tmp_res res[CMLHIP_MAX_FRAMES];
    int tgt[CMLHIP_MAX_FRAMES];
    int nres = 0;
    for (int t = 0; t < N; t++) {
        res_state[t] = -1;
        if (t == pt->host) continue;
        res[nres].new_energy = res[nres].energy = 0; res[nres].new_state = CMLHIP_RES_OUTLIER; res[nres].state = CMLHIP_RES_IN;
        tgt[nres] = t; nres++;
    }
    float lastEnergy = 0, lastHdd = 0, lastbd = 0;
    float currentIdepth = (float)((pt->idepth_max + pt->idepth_min) * (double)0.5f);
    for (int i = 0; i < nres; i++) {
        lastEnergy = (float)((double)lastEnergy + lin_res(images[tgt[i]], w, h, K, &pairs[pt->host * N + tgt[i]], P, pt, 1000, &res[i], &lastHdd, &lastbd, currentIdepth));
        res[i].state = res[i].new_state; res[i].energy = res[i].new_energy;
    }
    if (!isfinite(lastEnergy) || (double)lastHdd < P->min_idepth_h_act) return 0;
    float lambda = 0.1f;
    for (int it = 0; it < P->gn_its_on_activation; it++) {
        float H = lastHdd;
        H = H * (1 + lambda);
        const float step = (float)((1.0 / (double)H) * (double)lastbd);
        const float newIdepth = currentIdepth - step;
        float newHdd = 0, newbd = 0, newEnergy = 0;
        for (int i = 0; i < nres; i++)
            newEnergy = (float)((double)newEnergy + lin_res(images[tgt[i]], w, h, K, &pairs[pt->host * N + tgt[i]], P, pt, 1, &res[i], &newHdd, &newbd, newIdepth));
        if (!isfinite(lastEnergy) || (double)newHdd < P->min_idepth_h_act) return 0;
        if (newEnergy < lastEnergy) {
            currentIdepth = newIdepth; lastHdd = newHdd; lastbd = newbd; lastEnergy = newEnergy;
            for (int i = 0; i < nres; i++) { res[i].state = res[i].new_state; res[i].energy = res[i].new_energy; }
            lambda *= 0.5;
        } else lambda *= 5;
        if (fabsf(step) < 0.0001 * currentIdepth) break;
    }
    if (!isfinite(currentIdepth)) return -1;
    if (currentIdepth <= 0) return -1;
    int numGood = 0;
    for (int i = 0; i < nres; i++) if (res[i].state == CMLHIP_RES_IN) numGood++;
    if (numGood < min_obs) return -1;
    if (!isfinite(pt->energy_th)) return -1;
    *idepth_out = currentIdepth;
    for (int i = 0; i < nres; i++) res_state[tgt[i]] = res[i].state;
    return 1;
}
