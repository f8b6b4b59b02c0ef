// tracer.hip — immature points on the device.
// Replaces the per-point work of DSOTracer::traceNewCoarse -> trace (DSOTracer.cpp:585-823: epipolar segment, discrete
// search of up to 99 steps x 8 pattern pixels, quality, new inverse-depth interval) and of activatePoints ->
// optimizeImmaturePoint / linearizeResidual (DSOTracer.cpp:280-494: per-point Gauss-Newton over the window's frames).
//
// Mapping (gfx950, wave64): ONE WAVE PER POINT.
//   trace:    the epipolar preamble is wave-uniform fp64; lane = search step (two rounds cover the 99 steps), each lane
//             replays the reference's `float ptx += dx` sequence to its step, sums its 8 bilinear gray taps in pattern order,
//             then (energy, step) goes through a lexicographic wave minimum — the first best step, like the sequential loop.
//   optimize: lane = (residual slot, pattern pixel); per-pixel terms cross an LDS tile and are combined in the reference's
//             order (residual by residual, pixel by pixel, float accumulators, early return on the first out-of-bounds pixel).
// scalar_t is double in the reference; its float places are kept float; FP contraction is off: results are bit-identical
// to the CPU statement order.
#include "cmlhip_internal.h"
#include "trace_pairs.h"
#include <atomic>
#include <cstring>

#pragma clang fp contract(off)

__constant__ int c_tr_star8[16] = {0, -2, -1, -1, 1, -1, -2, 0, 0, 0, 2, 0, -1, 1, 0, 2};   // types.h:1381-1393

template <bool HALF>
__device__ __forceinline__ float4 tr_texel(const void* img, size_t i) {
    if (HALF) {
        uint2 v = reinterpret_cast<const uint2*>(img)[i];
        __half2 a = *reinterpret_cast<__half2*>(&v.x), b = *reinterpret_cast<__half2*>(&v.y);
        return make_float4(__low2float(a), __high2float(a), __low2float(b), 0.f);
    }
    return reinterpret_cast<const float4*>(img)[i];
}
// Array2D<T>::interpolate, image/Array2D.h:242-262 (all three channels of the texel)
template <bool HALF>
__device__ __forceinline__ void tr_bil(const void* img, int w, float x, float y, float& c0, float& c1, float& c2) {
    const int ix = (int)x, iy = (int)y;
    const float dx = x - (float)ix, dy = y - (float)iy, dxdy = dx * dy;
    const size_t i1 = (size_t)iy * w + ix;
    const float4 a = tr_texel<HALF>(img, i1), b = tr_texel<HALF>(img, i1 + 1), c = tr_texel<HALF>(img, i1 + w), d = tr_texel<HALF>(img, i1 + w + 1);
    const float w00 = 1 - dx - dy + dxdy, w01 = dx - dxdy, w10 = dy - dxdy, w11 = dxdy;
    c0 = a.x * w00 + b.x * w01 + c.x * w10 + d.x * w11;
    c1 = a.y * w00 + b.y * w01 + c.y * w10 + d.y * w11;
    c2 = a.z * w00 + b.z * w01 + c.z * w10 + d.z * w11;
}
__device__ __forceinline__ bool tr_inside(double x, double y, int w, int h, double pad) {       // Frame.h:136-138
    return x >= pad && y >= pad && x < (double)w - pad && y < (double)h - pad;
}
__device__ __forceinline__ double tr_bcast(double v, int lane) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __shfl(u.i[0], lane); u.i[1] = __shfl(u.i[1], lane);
    return u.d;
}

// what trace() writes of a point (DSOTracer.cpp:585-823: lastTraceUV / lastTracePixelInterval / lastTraceStatus, iDepthMin / iDepthMax, quality)
// the immature set changes size at every keyframe: its buffers grow with headroom (an exact-size buffer is freed and allocated again — a stream
// synchronisation and two driver calls, ~0.3 ms — whenever the set is a few points larger than ever before)
static int tr_ensure(cmlhip_ctx* c, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes) return CMLHIP_OK;
    return cml_ensure(c, b, std::max(bytes + bytes / 2, (size_t)64 * 1024));
}
typedef cmlhip_immature_state TraceJournal;     // (include/cmlhip.h: the same seven fields)
struct TraceArgs {
    const void* img; int w, h, n;
    const cmlhip_trace_pair* pairs;
    cmlhip_tracer_params P;
    cmlhip_immature_point* pts;
    int skip_host; int* counts;                // resident mode: host index of the traced frame, status histogram (null otherwise)
    TraceJournal* journal;                     // speculative trace: the fields trace() may write, saved per point before it runs (null otherwise)
};

template <bool HALF>
__global__ __launch_bounds__(256) void k_trace_points(TraceArgs A);

// resident mode wrapper: one more wave-uniform test in front, the histogram behind
template <bool HALF>
__device__ __forceinline__ void trace_one(const TraceArgs& A, int pi);

// (round 6, measured and not kept: the publishing step folded into this launch — every wave counts itself done, the last one copies histogram and pairs
//  to the host block and stores the ticket.  13.4 + 4.1 us as two launches became 21.6 us as one: the last wave's system-scope release has the launch's
//  dirty point records to write back first, which the kernel boundary otherwise does while the publishing launch is being dispatched.)
template <bool HALF>
__global__ __launch_bounds__(256) void k_trace_points(TraceArgs A) {
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int pi = blockIdx.x * 4 + wv;
    if (pi >= A.n) return;                                                         // wave-uniform
    const int host = A.pts[pi].host;
    if (host < 0) return;                                                          // not in the window
    if (A.journal && l == 0) {
        const cmlhip_immature_point& q = A.pts[pi];
        TraceJournal j;
        j.idepth_min = q.idepth_min; j.idepth_max = q.idepth_max; j.quality = q.quality; j.last_uv[0] = q.last_uv[0]; j.last_uv[1] = q.last_uv[1];
        j.last_pixel_interval = q.last_pixel_interval; j.last_status = q.last_status; j.pad = 0;
        A.journal[pi] = j;
    }
    if (host != A.skip_host) trace_one<HALF>(A, pi);
    if (A.counts && l == 0) {
        __threadfence_block();
        atomicAdd(A.counts + A.pts[pi].last_status, 1);
    }
}

template <bool HALF>
__device__ __forceinline__ void trace_one(const TraceArgs& A, const int pi) {
    __shared__ double s_err[4][128];
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    cmlhip_immature_point* s = A.pts + pi;
    const cmlhip_trace_pair* pr_ = A.pairs + s->host;
    const cmlhip_tracer_params& P = A.P;
    const int w = A.w, h = A.h;
    const int status_in = s->last_status;
    if (status_in == CMLHIP_IPS_OOB) return;                                       // DSOTracer.cpp:601-604
    double M[9], Kt[3];
#pragma unroll
    for (int k = 0; k < 9; k++) M[k] = pr_->KRKi[k];
#pragma unroll
    for (int k = 0; k < 3; k++) Kt[k] = pr_->Kt[k];
    const double aff_a = pr_->aff_a, aff_b = pr_->aff_b;
    const double idmin = s->idepth_min, idmax = s->idepth_max;
    const double cx = (double)s->x, cy = (double)s->y;
    // Eigen's order for a double 3x3 * 3-vector: packet rows 0-1 (e0 + e1) + e2, scalar row 2 e0 + (e1 + e2)
    const double pr0 = (M[0] * cx + M[1] * cy) + M[2] * 1.0, pr1 = (M[3] * cx + M[4] * cy) + M[5] * 1.0, pr2 = M[6] * cx + (M[7] * cy + M[8] * 1.0);
    const double maxPixSearch = (double)(w + h) * P.max_pix_search;                 // :611
    const double pm0 = pr0 + Kt[0] * idmin, pm1 = pr1 + Kt[1] * idmin, pm2 = pr2 + Kt[2] * idmin;
    const double minx = pm0 / pm2, miny = pm1 / pm2;
    // every early exit writes {lastTraceUV, lastTracePixelInterval, lastTraceStatus}
#define TR_EXIT(u0, u1, itv, st) do { if (l == 0) { s->last_uv[0] = (u0); s->last_uv[1] = (u1); s->last_pixel_interval = (itv); s->last_status = (st); } return; } while (0)
    if (!tr_inside(minx, miny, w, h, 4)) TR_EXIT(-1.0, -1.0, 0.0, CMLHIP_IPS_OOB);      // :620-626
    double maxx, maxy, pixelInterval;
    const bool finite_max = isfinite(idmax);
    if (finite_max) {
        const double q0 = pr0 + Kt[0] * idmax, q1 = pr1 + Kt[1] * idmax, q2 = pr2 + Kt[2] * idmax;
        maxx = q0 / q2; maxy = q1 / q2;
        if (!tr_inside(maxx, maxy, w, h, 5)) TR_EXIT(-1.0, -1.0, 0.0, CMLHIP_IPS_OOB);
        pixelInterval = sqrt((maxx - minx) * (maxx - minx) + (maxy - miny) * (maxy - miny));
        if (pixelInterval < P.max_slack_interval) TR_EXIT((maxx + minx) / 2.0, (maxy + miny) / 2.0, pixelInterval, CMLHIP_IPS_SKIPPED);   // :646-652
    } else {
        pixelInterval = maxPixSearch;
        const double q0 = pr0 + Kt[0] * 0.01, q1 = pr1 + Kt[1] * 0.01, q2 = pr2 + Kt[2] * 0.01;
        maxx = q0 / q2; maxy = q1 / q2;
        const double dirx = maxx - minx, diry = maxy - miny;
        const double inv = 1.0 / sqrt(dirx * dirx + diry * diry);
        maxx = minx + pixelInterval * dirx * inv; maxy = miny + pixelInterval * diry * inv;
        if (!tr_inside(maxx, maxy, w, h, 5)) TR_EXIT(-1.0, -1.0, 0.0, CMLHIP_IPS_OOB);
    }
    if (!(idmin < 0 || (pm2 > 0.75 && pm2 < 1.5))) TR_EXIT(-1.0, -1.0, 0.0, CMLHIP_IPS_OOB);   // :682-688
    double dx = P.trace_step_size * (maxx - minx), dy = P.trace_step_size * (maxy - miny);
    const double G0 = s->gradH[0], G1 = s->gradH[1], G2 = s->gradH[2], G3 = s->gradH[3];
    const double a = dx * (G0 * dx + G1 * dy) + dy * (G2 * dx + G3 * dy);
    const double b = dy * (G0 * dy + G1 * (-dx)) + (-dx) * (G2 * dy + G3 * (-dx));
    double errorInPixel = (double)0.2f + (double)0.2f * (a + b) / a;                // :697
    if (errorInPixel * P.min_improvement_factor > pixelInterval && finite_max)
        TR_EXIT((maxx + minx) / 2.0, (maxy + miny) / 2.0, pixelInterval, CMLHIP_IPS_BADCONDITION);
    if (errorInPixel > 10) errorInPixel = 10;
    dx /= pixelInterval; dy /= pixelInterval;
    if (pixelInterval > maxPixSearch) { maxx += maxPixSearch * dx; maxy += maxPixSearch * dy; pixelInterval = maxPixSearch; }
    int numSteps = (int)((double)1.9999f + pixelInterval / P.trace_step_size);
    const double randShift = minx * 1000 - floor(minx * 1000);
    const float ptx0 = (float)(minx - randShift * dx), pty0 = (float)(miny - randShift * dy);
    if (!isfinite(dx) || !isfinite(dy)) TR_EXIT(-1.0, -1.0, 0.0, CMLHIP_IPS_OOB);
    if (numSteps >= 100) numSteps = 99;
    double rot[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        rot[2 * i] = M[0] * (double)c_tr_star8[2 * i] + M[1] * (double)c_tr_star8[2 * i + 1];
        rot[2 * i + 1] = M[3] * (double)c_tr_star8[2 * i] + M[4] * (double)c_tr_star8[2 * i + 1];
    }
    float gray[8];
#pragma unroll
    for (int i = 0; i < 8; i++) gray[i] = s->gray[i];
    // ---- discrete search: lane = step (two rounds)
    double bestE = 1e10; int bestI = 0x7fffffff; float bestU = 0.f, bestV = 0.f;
    for (int round = 0; round < 2; round++) {
        const int i = l + 64 * round;
        double energy = 1e300;
        float px = ptx0, py = pty0;
        if (i < numSteps) {
            for (int j = 0; j < i; j++) { px = (float)((double)px + dx); py = (float)((double)py + dy); }   // `ptx += dx`, :757-758
            energy = 0;
#pragma unroll
            for (int idx = 0; idx < 8; idx++) {
                const double qx = (double)px + rot[2 * idx], qy = (double)py + rot[2 * idx + 1];
                if (!tr_inside(qx, qy, w, h, 3)) { energy += 1e5; continue; }
                float c0, c1, c2;
                tr_bil<HALF>(A.img, w, (float)qx, (float)qy, c0, c1, c2);
                const double residual = (double)c0 - (aff_a * (double)gray[idx] + aff_b);
                const double hw = fabs(residual) < P.huber_th ? 1 : P.huber_th / fabs(residual);
                energy += hw * residual * residual * (2 - hw);
            }
            s_err[wv][i] = energy;
        }
        if (i < numSteps && energy < 1e10 && (energy < bestE || (energy == bestE && i < bestI))) { bestE = energy; bestI = i; bestU = px; bestV = py; }
    }
    // lexicographic wave minimum of (energy, step): the first step with the smallest energy, as `if (energy < bestEnergy)` finds it
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double oe = tr_bcast(bestE, l ^ o);
        const int oi = __shfl(bestI, l ^ o);
        const float ou = __shfl(bestU, l ^ o), ov = __shfl(bestV, l ^ o);
        if (oe < bestE || (oe == bestE && oi < bestI)) { bestE = oe; bestI = oi; bestU = ou; bestV = ov; }
    }
    const int bestIdx = bestI == 0x7fffffff ? -1 : bestI;
    double secondBest = 1e10;
    for (int round = 0; round < 2; round++) {
        const int i = l + 64 * round;
        if (i < numSteps) {
            const double e = s_err[wv][i];
            if (((double)i < (double)bestIdx - P.min_trace_test_radius || (double)i > (double)bestIdx + P.min_trace_test_radius) && e < secondBest) secondBest = e;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double oe = tr_bcast(secondBest, l ^ o); if (oe < secondBest) secondBest = oe; }
    if (l != 0) return;
    const double bestEnergy = bestIdx < 0 ? 1e10 : bestE;
    const double newQuality = secondBest / bestEnergy;
    if (newQuality < s->quality || numSteps > 10) s->quality = newQuality;
    if (bestEnergy >= s->energy_th * P.extra_slack_on_th) {                          // :777-789
        s->last_pixel_interval = 0; s->last_uv[0] = -1; s->last_uv[1] = -1;
        s->last_status = status_in == CMLHIP_IPS_OUTLIER ? CMLHIP_IPS_OOB : CMLHIP_IPS_OUTLIER;
        return;
    }
    const double bU = bestIdx < 0 ? 0.0 : (double)bestU, bV = bestIdx < 0 ? 0.0 : (double)bestV;
    double nmin, nmax;
    if (dx * dx > dy * dy) {
        nmin = (pr2 * (bU - errorInPixel * dx) - pr0) / (Kt[0] - Kt[2] * (bU - errorInPixel * dx));
        nmax = (pr2 * (bU + errorInPixel * dx) - pr0) / (Kt[0] - Kt[2] * (bU + errorInPixel * dx));
    } else {
        nmin = (pr2 * (bV - errorInPixel * dy) - pr1) / (Kt[1] - Kt[2] * (bV - errorInPixel * dy));
        nmax = (pr2 * (bV + errorInPixel * dy) - pr1) / (Kt[1] - Kt[2] * (bV + errorInPixel * dy));
    }
    if (nmin > nmax) { const double t = nmin; nmin = nmax; nmax = t; }
    s->idepth_min = nmin; s->idepth_max = nmax;
    s->last_pixel_interval = 2 * errorInPixel;
    s->last_uv[0] = bU; s->last_uv[1] = bV;
    s->last_status = CMLHIP_IPS_GOOD;
#undef TR_EXIT
}

// ------------------------------------------------------------------------------------------------ optimizeImmaturePoint
struct OptArgs {
    const void* img[CMLHIP_MAX_FRAMES];
    int N, w, h, n, min_obs;
    double K[4];
    const cmlhip_activation_pair* pairs;       // host * N + target
    cmlhip_tracer_params P;
    const cmlhip_immature_point* pts;
    const int* slots;                          // null: point i is pts[i]; else pts[slots[i]] (the device-resident set)
    int* result; float* idepth; int* res_state;
};

struct OptPix { double e, hdd, bd; int oob; int pad; };

// one evaluation of all residuals at `idepth` (linearizeResidual x nres, DSOTracer.cpp:406-494), combined in the reference's order
template <bool HALF>
__device__ __forceinline__ float opt_eval(const OptArgs& A, const cmlhip_immature_point* pt, OptPix* s_pix, int* s_state, int* s_new_state,
                                          double* s_energy, double* s_new_energy, const int* s_tgt, int nres, float slack, float idepth,
                                          float& Hdd, float& bd) {
    const int l = threadIdx.x & 63, slot = l >> 3, idx = l & 7;
    float total = 0.f;
    for (int base = 0; base < nres; base += 8) {
        const int ri = base + slot;
        if (ri < nres && s_state[ri] != CMLHIP_RES_OOB) {
            const int t = s_tgt[ri];
            const cmlhip_activation_pair* ht = A.pairs + pt->host * A.N + t;
            const double ux = ((double)pt->x + (double)c_tr_star8[2 * idx] - A.K[2]) * (1.0 / A.K[0]);
            const double uy = ((double)pt->y + (double)c_tr_star8[2 * idx + 1] - A.K[3]) * (1.0 / A.K[1]);
            const double p0 = (ht->R[0] * ux + ht->R[1] * uy + ht->R[2] * 1.0) + ht->t[0] * (double)idepth;
            const double p1 = (ht->R[3] * ux + ht->R[4] * uy + ht->R[5] * 1.0) + ht->t[1] * (double)idepth;
            const double p2 = (ht->R[6] * ux + ht->R[7] * uy + ht->R[8] * 1.0) + ht->t[2] * (double)idepth;
            const double upx = p0 / p2, upy = p1 / p2;
            const double projx = upx * A.K[0] + A.K[2], projy = upy * A.K[1] + A.K[3];
            const double drescale = 1.0 / p2;
            OptPix o; o.e = 0; o.hdd = 0; o.bd = 0; o.pad = 0;
            o.oob = (!tr_inside(projx, projy, A.w, A.h, 1) || drescale <= 0) ? 1 : 0;                    // :436-440
            if (!o.oob) {
                float g0, g1, g2;
                tr_bil<HALF>(A.img[t], A.w, (float)projx, (float)projy, g0, g1, g2);
                const float* gt = pt->dpatch + 3 * idx;
                const double residual = (double)g0 - (ht->aff_a * (double)gt[0] + ht->aff_b);
                double hw = fabs(residual) < A.P.huber_th ? 1 : A.P.huber_th / fabs(residual);
                const float sq = gt[1] * gt[1] + gt[2] * gt[2];
                const double weight = sqrt(A.P.outlier_th_sum_component / (A.P.outlier_th_sum_component + (double)sq));
                o.e = weight * weight * hw * residual * residual * (2 - hw);
                const double dxI = (double)g1 * A.K[0], dyI = (double)g2 * A.K[1];
                const double d_idepth = dxI * drescale * (ht->t[0] - ht->t[2] * upx) + dyI * drescale * (ht->t[1] - ht->t[2] * upy);
                hw *= weight * weight;
                o.hdd = (hw * d_idepth) * d_idepth;
                o.bd = (hw * residual) * d_idepth;
            }
            s_pix[slot * 8 + idx] = o;
        }
        // (one wave: LDS accesses are ordered) — sequential combination, identical in every lane
        for (int k = 0; k < 8 && base + k < nres; k++) {
            const int r = base + k;
            double ret;
            if (s_state[r] == CMLHIP_RES_OOB) { if (l == 0) s_new_state[r] = CMLHIP_RES_OOB; ret = s_energy[r]; }
            else {
                float energyLeft = 0.f;
                bool oob = false;
                for (int j = 0; j < 8; j++) {
                    const OptPix o = s_pix[k * 8 + j];
                    if (o.oob) { oob = true; break; }
                    energyLeft = (float)((double)energyLeft + o.e);
                    Hdd = (float)((double)Hdd + o.hdd);
                    bd = (float)((double)bd + o.bd);
                }
                if (oob) { if (l == 0) s_new_state[r] = CMLHIP_RES_OOB; ret = s_energy[r]; }
                else {
                    int ns;
                    if ((double)energyLeft > pt->energy_th * (double)slack) { energyLeft = (float)(pt->energy_th * (double)slack); ns = CMLHIP_RES_OUTLIER; }
                    else ns = CMLHIP_RES_IN;
                    if (l == 0) { s_new_state[r] = ns; s_new_energy[r] = (double)energyLeft; }
                    ret = (double)energyLeft;
                }
            }
            total = (float)((double)total + ret);
        }
    }
    return total;
}

template <bool HALF>
__global__ __launch_bounds__(256) void k_optimize_immature(OptArgs A) {
    __shared__ OptPix s_pix[4][64];
    __shared__ int s_state[4][CMLHIP_MAX_FRAMES], s_new_state[4][CMLHIP_MAX_FRAMES], s_tgt[4][CMLHIP_MAX_FRAMES];
    __shared__ double s_energy[4][CMLHIP_MAX_FRAMES], s_new_energy[4][CMLHIP_MAX_FRAMES];
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int pi = blockIdx.x * 4 + wv;
    if (pi >= A.n) return;
    const cmlhip_immature_point* pt = A.pts + (A.slots ? A.slots[pi] : pi);      // (resident form: the point's slot in the device's set)
    int nres = 0;
    for (int t = 0; t < A.N; t++) {
        if (l == 0) A.res_state[(size_t)pi * A.N + t] = -1;
        if (t == pt->host) continue;
        if (l == 0) { s_tgt[wv][nres] = t; s_state[wv][nres] = CMLHIP_RES_IN; s_new_state[wv][nres] = CMLHIP_RES_OUTLIER; s_energy[wv][nres] = 0; s_new_energy[wv][nres] = 0; }
        nres++;
    }
#define EVAL(slack, id, H, B) opt_eval<HALF>(A, pt, s_pix[wv], s_state[wv], s_new_state[wv], s_energy[wv], s_new_energy[wv], s_tgt[wv], nres, slack, id, H, B)
#define COMMIT() do { if (l == 0) for (int i_ = 0; i_ < nres; i_++) { s_state[wv][i_] = s_new_state[wv][i_]; s_energy[wv][i_] = s_new_energy[wv][i_]; } } while (0)
    float lastHdd = 0, lastbd = 0;
    float currentIdepth = (float)((pt->idepth_max + pt->idepth_min) * (double)0.5f);
    // first pass: the reference commits each residual right after its own linearisation (:321-326) — equivalent to committing after
    // the pass, because a residual's evaluation reads only its own state
    float lastEnergy = EVAL(1000.f, currentIdepth, lastHdd, lastbd);
    COMMIT();
    int result = 1;
    if (!isfinite(lastEnergy) || (double)lastHdd < A.P.min_idepth_h_act) result = 0;
    if (result == 1) {
        float lambda = 0.1f;
        for (int it = 0; it < A.P.gn_its_on_activation; it++) {
            float H = lastHdd;
            H = H * (1 + lambda);
            const float step = (float)((1.0 / (double)H) * (double)lastbd);
            const float newIdepth = currentIdepth - step;
            float newHdd = 0, newbd = 0;
            const float newEnergy = EVAL(1.f, newIdepth, newHdd, newbd);
            if (!isfinite(lastEnergy) || (double)newHdd < A.P.min_idepth_h_act) { result = 0; break; }
            if (newEnergy < lastEnergy) {
                currentIdepth = newIdepth; lastHdd = newHdd; lastbd = newbd; lastEnergy = newEnergy;
                COMMIT();
                lambda *= 0.5;
            } else lambda *= 5;
            if ((double)fabsf(step) < 0.0001 * (double)currentIdepth) break;
        }
    }
    if (result == 1) {
        if (!isfinite(currentIdepth) || currentIdepth <= 0) result = -1;
        else {
            int numGood = 0;
            for (int i = 0; i < nres; i++) numGood += s_state[wv][i] == CMLHIP_RES_IN;
            if (numGood < A.min_obs || !isfinite(pt->energy_th)) result = -1;
        }
    }
    if (l == 0) {
        A.result[pi] = result;
        A.idepth[pi] = result == 1 ? currentIdepth : 0.f;
        if (result == 1) for (int i = 0; i < nres; i++) A.res_state[(size_t)pi * A.N + s_tgt[wv][i]] = s_state[wv][i];
    }
#undef EVAL
#undef COMMIT
}

// ------------------------------------------------------------------------------------------------ API
extern "C" {

int cmlhip_trace_points(cmlhip_ctx* c, uint64_t image_id, const cmlhip_tracer_params* prm, int n_hosts, const cmlhip_trace_pair* pairs,
                        int n, cmlhip_immature_point* points) { CML_DEV(c);
    if (!c || !prm || n < 0 || n_hosts < 1 || !pairs || (n > 0 && !points)) return CMLHIP_ERR_INVALID;
    const Pyramid* py = cml_find_pyr(c, image_id);
    CML_REQUIRE(c, py && py->lv[0].grad, CMLHIP_ERR_NOT_FOUND, "traced image not in the pyramid cache");
    if (n == 0) return CMLHIP_OK;
    for (int i = 0; i < n; i++) if (points[i].host < 0 || points[i].host >= n_hosts) { c->err = "immature point host out of range"; return CMLHIP_ERR_INVALID; }
    int rc;
    if ((rc = cml_ensure(c, c->tr_points, sizeof(cmlhip_immature_point) * (size_t)n))) return rc;
    if ((rc = cml_ensure(c, c->tr_pairs, sizeof(cmlhip_trace_pair) * (size_t)n_hosts))) return rc;
    if ((rc = cml_h2d(c, c->tr_points.p, points, sizeof(cmlhip_immature_point) * (size_t)n))) return rc;
    if ((rc = cml_h2d(c, c->tr_pairs.p, pairs, sizeof(cmlhip_trace_pair) * (size_t)n_hosts))) return rc;
    c->tr_req_consumed = false;                              // (pairs a tracker launch left in tr_pairs are overwritten)
    TraceArgs A;
    A.img = py->lv[0].grad; A.w = py->lv[0].w; A.h = py->lv[0].h; A.n = n;
    A.pairs = c->tr_pairs.as<cmlhip_trace_pair>(); A.P = *prm; A.pts = c->tr_points.as<cmlhip_immature_point>();
    A.skip_host = -2; A.counts = nullptr; A.journal = nullptr;
    if (c->lim.texel_format == CMLHIP_TEXEL_F16) k_trace_points<true><<<cml_div_up(n, 4), 256, 0, c->stream>>>(A);
    else k_trace_points<false><<<cml_div_up(n, 4), 256, 0, c->stream>>>(A);
    CML_CHECK(c, hipGetLastError());
    return cml_d2h(c, points, c->tr_points.p, sizeof(cmlhip_immature_point) * (size_t)n);
}

int cmlhip_tracer_set_points(cmlhip_ctx* c, int n, const cmlhip_immature_point* points) { CML_DEV(c);
    if (!c || n < 0 || (n > 0 && !points)) return CMLHIP_ERR_INVALID;
    int rc;
    if ((rc = tr_ensure(c, c->tr_resident, sizeof(cmlhip_immature_point) * (size_t)std::max(n, 1)))) return rc;
    if (n > 0 && (rc = cml_h2d(c, c->tr_resident.p, points, sizeof(cmlhip_immature_point) * (size_t)n))) return rc;
    c->tr_resident_n = n;
    return CMLHIP_OK;
}

// the resident set edited where it lies: kept points move to the front (slot i <- old slot keep[i], host index hosts[i]), new points follow
__global__ void k_tracer_rebuild(const cmlhip_immature_point* __restrict__ old_, cmlhip_immature_point* __restrict__ new_, const int* __restrict__ keep,
                                 const int* __restrict__ hosts, int n_keep) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), w = threadIdx.x & 63;
    if (i >= n_keep) return;
    const unsigned* src = reinterpret_cast<const unsigned*>(old_ + keep[i]);
    unsigned* dst = reinterpret_cast<unsigned*>(new_ + i);
    constexpr int NW = sizeof(cmlhip_immature_point) / 4;
    if (w < NW) dst[w] = src[w];
    if (w == 0) new_[i].host = hosts[i];
}
__global__ void k_tracer_pack_state(const cmlhip_immature_point* __restrict__ pts, cmlhip_immature_state* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cmlhip_immature_point& q = pts[i];
    cmlhip_immature_state j;
    j.idepth_min = q.idepth_min; j.idepth_max = q.idepth_max; j.quality = q.quality; j.last_uv[0] = q.last_uv[0]; j.last_uv[1] = q.last_uv[1];
    j.last_pixel_interval = q.last_pixel_interval; j.last_status = q.last_status; j.pad = 0;
    out[i] = j;
}
int cmlhip_tracer_edit_points(cmlhip_ctx* c, int n_keep, const int* keep, const int* hosts, int n_new, const cmlhip_immature_point* new_points) { CML_DEV(c);
    if (!c || n_keep < 0 || n_new < 0 || (n_keep > 0 && (!keep || !hosts)) || (n_new > 0 && !new_points) || n_keep > c->tr_resident_n) return CMLHIP_ERR_INVALID;
    for (int i = 0; i < n_keep; i++) if (keep[i] < 0 || keep[i] >= c->tr_resident_n) { c->err = "cmlhip_tracer_edit_points: slot out of range"; return CMLHIP_ERR_INVALID; }
    CML_REQUIRE(c, !c->tr_spec_pending, CMLHIP_ERR_INVALID, "cmlhip_tracer_edit_points: a speculative trace is in flight");
    const int n = n_keep + n_new;
    int rc;
    if ((rc = tr_ensure(c, c->tr_resident2, sizeof(cmlhip_immature_point) * (size_t)std::max(n, 1)))) return rc;
    if (n_keep > 0 && (rc = tr_ensure(c, c->tr_edit, 8 * (size_t)n_keep))) return rc;
    // the kept slots, their host indices and the new records: ONE packed upload (the new records land behind the slots the rebuild kernel fills)
    cml_h2d_batch_begin(c);
    rc = CMLHIP_OK;
    if (n_keep > 0) {
        rc = cml_h2d(c, c->tr_edit.p, keep, 4 * (size_t)n_keep);
        if (!rc) rc = cml_h2d(c, c->tr_edit.as<char>() + 4 * (size_t)n_keep, hosts, 4 * (size_t)n_keep);
    }
    if (!rc && n_new > 0) rc = cml_h2d(c, c->tr_resident2.as<cmlhip_immature_point>() + n_keep, new_points, sizeof(cmlhip_immature_point) * (size_t)n_new);
    { const int rf = cml_h2d_batch_flush(c); if (rc || rf) return rc ? rc : rf; }
    if (n_keep > 0) {
        k_tracer_rebuild<<<cml_div_up(n_keep, 4), 256, 0, c->stream>>>(c->tr_resident.as<cmlhip_immature_point>(), c->tr_resident2.as<cmlhip_immature_point>(),
                                                                      c->tr_edit.as<int>(), c->tr_edit.as<int>() + n_keep, n_keep);
        CML_CHECK(c, hipGetLastError());
    }
    std::swap(c->tr_resident, c->tr_resident2);
    c->tr_resident_n = n;
    return CMLHIP_OK;
}
int cmlhip_tracer_get_state(cmlhip_ctx* c, int n, cmlhip_immature_state* out) { CML_DEV(c);
    if (!c || n < 0 || n > c->tr_resident_n || (n > 0 && !out)) return CMLHIP_ERR_INVALID;
    if (n == 0) return CMLHIP_OK;
    int rc;
    if ((rc = tr_ensure(c, c->tr_state, sizeof(cmlhip_immature_state) * (size_t)n))) return rc;
    k_tracer_pack_state<<<cml_div_up(n, 256), 256, 0, c->stream>>>(c->tr_resident.as<cmlhip_immature_point>(), c->tr_state.as<cmlhip_immature_state>(), n);
    CML_CHECK(c, hipGetLastError());
    return cml_d2h(c, out, c->tr_state.p, sizeof(cmlhip_immature_state) * (size_t)n);
}

int cmlhip_tracer_get_points(cmlhip_ctx* c, int n, cmlhip_immature_point* points) { CML_DEV(c);
    if (!c || n < 0 || n > c->tr_resident_n || (n > 0 && !points)) return CMLHIP_ERR_INVALID;
    return n ? cml_d2h(c, points, c->tr_resident.p, sizeof(cmlhip_immature_point) * (size_t)n) : CMLHIP_OK;
}

int cmlhip_tracer_trace_resident(cmlhip_ctx* c, uint64_t image_id, const cmlhip_tracer_params* prm, int n_hosts, const cmlhip_trace_pair* pairs,
                                 int skip_host, int counts[6]) { CML_DEV(c);
    if (!c || !prm || n_hosts < 1 || !pairs || !counts) return CMLHIP_ERR_INVALID;
    const Pyramid* py = cml_find_pyr(c, image_id);
    CML_REQUIRE(c, py && py->lv[0].grad, CMLHIP_ERR_NOT_FOUND, "traced image not in the pyramid cache");
    const int n = c->tr_resident_n;
    for (int k = 0; k < 6; k++) counts[k] = 0;
    if (n == 0) return CMLHIP_OK;
    int rc;
    if ((rc = cml_ensure(c, c->tr_pairs, sizeof(cmlhip_trace_pair) * (size_t)n_hosts))) return rc;
    if ((rc = cml_ensure(c, c->tr_out, 64))) return rc;
    if ((rc = cml_h2d(c, c->tr_pairs.p, pairs, sizeof(cmlhip_trace_pair) * (size_t)n_hosts))) return rc;
    c->tr_req_consumed = false;                              // (pairs a tracker launch left in tr_pairs are overwritten)
    CML_CHECK(c, hipMemsetAsync(c->tr_out.p, 0, 24, c->stream));
    TraceArgs A;
    A.img = py->lv[0].grad; A.w = py->lv[0].w; A.h = py->lv[0].h; A.n = n;
    A.pairs = c->tr_pairs.as<cmlhip_trace_pair>(); A.P = *prm; A.pts = c->tr_resident.as<cmlhip_immature_point>();
    A.skip_host = skip_host; A.counts = c->tr_out.as<int>(); A.journal = nullptr;
    if (c->lim.texel_format == CMLHIP_TEXEL_F16) k_trace_points<true><<<cml_div_up(n, 4), 256, 0, c->stream>>>(A);
    else k_trace_points<false><<<cml_div_up(n, 4), 256, 0, c->stream>>>(A);
    CML_CHECK(c, hipGetLastError());
    return cml_d2h(c, counts, c->tr_out.p, 24);
}

// ---- traceNewCoarse of the resident set behind a tracker batch that is still in flight (one enqueue, one wait per tracked frame).
// The pose traced against is the FIRST hypothesis' result — the try trackWithMotionModel's loop ends on whenever it is good (DSOTracker.h:306-309);
// the caller replays the selection on the batch's results after the wait and either keeps the trace or rolls it back (the journal) and traces
// again with the pose it did select.  Pairs as DSOTracer.cpp:606-608 forms them: frame = refToNew o reference, host -> frame = frame o host^-1,
// K R K^-1, K t, and the exposure transfer with exposure times 1 (Exposure.h:119-123).
}  // extern "C" (device code of the tracked trace)
struct TraceTracked {                          // kernel arguments of the tracked trace: the window's poses travel with the launch (windows of up to TR_INLINE_HOSTS frames)
    const double* pose0;                       // {R[9], t[3], a, b} of the batch's first result (device; written by k_tracker_optimize)
    TrackedReq W;
};
// the trace with the pairs formed IN the launch: a wave derives the pair of its point's host from the first result (a few hundred wave-uniform
// operations against a launch of its own for all of them), parks it in LDS and traces against it
template <bool HALF>
__global__ __launch_bounds__(256) void k_trace_points_tracked(TraceArgs A, TraceTracked T) {
    __shared__ cmlhip_trace_pair s_pair[4];
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int pi = blockIdx.x * 4 + wv;
    if (pi >= A.n) return;                                                         // wave-uniform
    const int host = A.pts[pi].host;
    if (host < 0 || host >= T.W.n_hosts) return;                                   // not in the window
    if (l == 0) {
        const cmlhip_immature_point& q = A.pts[pi];
        TraceJournal j;
        j.idepth_min = q.idepth_min; j.idepth_max = q.idepth_max; j.quality = q.quality; j.last_uv[0] = q.last_uv[0]; j.last_uv[1] = q.last_uv[1];
        j.last_pixel_interval = q.last_pixel_interval; j.last_status = q.last_status; j.pad = 0;
        A.journal[pi] = j;
    }
    if (host != A.skip_host) {
        if (l == 0) tp_pair(T.pose0, T.W.ref, T.W.K, T.W.hosts[host], s_pair[wv]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                     // (one wave writes and reads its slot: in-order LDS, compiler ordering)
        __builtin_amdgcn_wave_barrier();
        TraceArgs B = A;
        B.pairs = &s_pair[wv] - host;                                              // trace_one indexes the pairs by the point's host
        trace_one<HALF>(B, pi);
    }
    if (l == 0) {
        __threadfence_block();
        atomicAdd(A.counts + A.pts[pi].last_status, 1);
    }
}
// windows of more than TR_INLINE_HOSTS frames: the pairs by a launch of their own (hosts from the mapped block)
struct TracePoseArgs { const double* pose0; cmlhip_frame_pose ref; double K[4]; int n_hosts; const cmlhip_frame_pose* hosts; cmlhip_trace_pair* pairs; };
__global__ void k_trace_pairs_from_tracker(TracePoseArgs A) {
    const int h = threadIdx.x;
    if (h >= A.n_hosts) return;
    cmlhip_trace_pair P;
    tp_pair(A.pose0, A.ref, A.K, A.hosts[h], P);
    A.pairs[h] = P;
}
__global__ void k_trace_rollback(cmlhip_immature_point* pts, const TraceJournal* journal, int n, int skip_host) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || pts[i].host < 0 || pts[i].host == skip_host) return;
    const TraceJournal j = journal[i];
    cmlhip_immature_point& q = pts[i];
    q.idepth_min = j.idepth_min; q.idepth_max = j.idepth_max; q.quality = j.quality; q.last_uv[0] = j.last_uv[0]; q.last_uv[1] = j.last_uv[1];
    q.last_pixel_interval = j.last_pixel_interval; q.last_status = j.last_status;
}
// the chain's last launch: the status histogram and the pairs the trace used into mapped host memory (the pairs formed again by the same function: same
// bits), the histogram cleared for the next frame, then — behind a system-scope fence — the completion ticket the host's wait spins on
struct TracePublish { const double* pose0; cmlhip_frame_pose ref; double K[4]; int n_hosts; const cmlhip_frame_pose* hosts; const cmlhip_trace_pair* pairs; int* counts; int* out_counts;
                      cmlhip_trace_pair* out_pairs; volatile unsigned* ticket_word; unsigned ticket; };
__global__ void k_trace_publish(TracePublish A) {
    const int t = threadIdx.x;
    if (t < 6) { A.out_counts[t] = A.counts[t]; A.counts[t] = 0; }
    if (A.pairs) {                                           // the pairs the tracker launch left (its tail formed them): copied
        const int nw = A.n_hosts * (int)(sizeof(cmlhip_trace_pair) / 8);
        for (int w = t; w < nw; w += blockDim.x) reinterpret_cast<double*>(A.out_pairs)[w] = reinterpret_cast<const double*>(A.pairs)[w];
    } else if (t < A.n_hosts) {
        cmlhip_trace_pair P;
        tp_pair(A.pose0, A.ref, A.K, A.hosts[t], P);
        A.out_pairs[t] = P;
    }
    __threadfence_system();
    __syncthreads();
    if (t == 0) { __threadfence_system(); *A.ticket_word = A.ticket; }
}
extern "C" {

// The window of the NEXT tracked trace, handed over BEFORE the tracker batch is enqueued: the batch's launch then carries it and the workgroup that ends the
// first hypothesis forms the pairs host -> frame right behind its result (one wave, a lane per host) — the trace behind it reads them like any caller's pairs
// instead of every wave deriving its own (14.7 -> 10.5 us for the trace launch, and the publishing launch copies instead of forming them again).
int cmlhip_tracer_tracked_prepare(cmlhip_ctx* c, int n_hosts, const cmlhip_frame_pose* hosts, const cmlhip_frame_pose* reference, const double K[4]) {
    if (!c || n_hosts < 1 || n_hosts > CMLHIP_MAX_FRAMES || !hosts || !reference || !K) return CMLHIP_ERR_INVALID;
    c->tr_req_valid = false; c->tr_req_consumed = false;
    if (n_hosts > TR_INLINE_HOSTS) return CMLHIP_OK;        // (wider windows: the pairs by a launch of their own, as before)
    TrackedReq& W = c->tr_req;
    memset(&W, 0, sizeof W);
    W.ref = *reference; for (int k = 0; k < 4; k++) W.K[k] = K[k];
    W.n_hosts = n_hosts;
    memcpy(W.hosts, hosts, sizeof(cmlhip_frame_pose) * (size_t)n_hosts);
    c->tr_req_valid = true;
    return CMLHIP_OK;
}

int cmlhip_tracer_trace_resident_tracked_async(cmlhip_ctx* c, uint64_t image_id, const cmlhip_tracer_params* prm, int n_hosts,
                                               const cmlhip_frame_pose* hosts, const cmlhip_frame_pose* reference, const double K[4], int skip_host) { CML_DEV(c);
    if (!c || !prm || n_hosts < 1 || n_hosts > CMLHIP_MAX_FRAMES || !hosts || !reference || !K) return CMLHIP_ERR_INVALID;
    CML_REQUIRE(c, cml_tracker_pending_result_dev(c, 0) && c->trk_pose0.p, CMLHIP_ERR_INVALID,
                "cmlhip_tracer_trace_resident_tracked_async: no tracker batch in flight (cmlhip_tracker_optimize_batch_async first)");
    CML_REQUIRE(c, !c->tr_spec_pending, CMLHIP_ERR_INVALID, "cmlhip_tracer_trace_resident_tracked_async: the previous speculative trace was not finished");
    const Pyramid* py = cml_find_pyr(c, image_id);
    CML_REQUIRE(c, py && py->lv[0].grad, CMLHIP_ERR_NOT_FOUND, "traced image not in the pyramid cache");
    const int n = c->tr_resident_n;
    int rc;
    const unsigned gen0 = c->tr_counts.gen;
    if ((rc = cml_ensure(c, c->tr_pairs, sizeof(cmlhip_trace_pair) * (size_t)CMLHIP_MAX_FRAMES))) return rc;
    if ((rc = cml_ensure(c, c->tr_counts, 64))) return rc;      // (a buffer of its own: tr_out also carries the activation results)
    if ((rc = tr_ensure(c, c->tr_journal, sizeof(TraceJournal) * (size_t)std::max(n, 1)))) return rc;
    if (!c->tr_host) {
        CML_CHECK(c, hipHostMalloc(&c->tr_host, 64 + (sizeof(cmlhip_trace_pair) + sizeof(cmlhip_frame_pose)) * CMLHIP_MAX_FRAMES, hipHostMallocMapped | hipHostMallocCoherent));
        CML_CHECK(c, hipHostGetDevicePointer(&c->tr_host_dev, c->tr_host, 0));
    }
    // the histogram is cleared by the publishing kernel of every frame; once per allocation here
    if (c->tr_counts.gen != gen0) CML_CHECK(c, hipMemsetAsync(c->tr_counts.p, 0, 24, c->stream));
    // the window's poses: in the kernel arguments (up to TR_INLINE_HOSTS frames) and in the mapped block (the publishing kernel, wider windows)
    char* const hosts_h = static_cast<char*>(c->tr_host) + 64 + sizeof(cmlhip_trace_pair) * CMLHIP_MAX_FRAMES;
    memcpy(hosts_h, hosts, sizeof(cmlhip_frame_pose) * (size_t)n_hosts);
    std::atomic_thread_fence(std::memory_order_release);
    const cmlhip_frame_pose* hosts_dev = reinterpret_cast<const cmlhip_frame_pose*>(static_cast<char*>(c->tr_host_dev) + 64 + sizeof(cmlhip_trace_pair) * CMLHIP_MAX_FRAMES);
    const double* pose0 = c->trk_pose0.as<double>();
    TraceArgs A;
    A.img = py->lv[0].grad; A.w = py->lv[0].w; A.h = py->lv[0].h; A.n = n;
    A.pairs = c->tr_pairs.as<cmlhip_trace_pair>(); A.P = *prm; A.pts = c->tr_resident.as<cmlhip_immature_point>();
    A.skip_host = skip_host; A.counts = c->tr_counts.as<int>(); A.journal = c->tr_journal.as<TraceJournal>();
    const bool half = c->lim.texel_format == CMLHIP_TEXEL_F16;
    // did the tracker launch carry this very window (cmlhip_tracer_tracked_prepare ahead of it)?  Then its tail leaves the pairs in tr_pairs.
    bool from_tracker = c->tr_req_consumed && c->tr_req.n_hosts == n_hosts && memcmp(c->tr_req.hosts, hosts, sizeof(cmlhip_frame_pose) * (size_t)n_hosts) == 0 &&
                        memcmp(&c->tr_req.ref, reference, sizeof(cmlhip_frame_pose)) == 0 && memcmp(c->tr_req.K, K, 4 * sizeof(double)) == 0;
    c->tr_req_consumed = false;
    if (n > 0 && from_tracker) {
        if (half) k_trace_points<true><<<cml_div_up(n, 4), 256, 0, c->stream>>>(A);
        else k_trace_points<false><<<cml_div_up(n, 4), 256, 0, c->stream>>>(A);
    } else if (n > 0 && n_hosts <= TR_INLINE_HOSTS) {
        TraceTracked T;
        memset(&T, 0, sizeof T);
        T.pose0 = pose0; T.W.ref = *reference; for (int k = 0; k < 4; k++) T.W.K[k] = K[k];
        T.W.n_hosts = n_hosts;
        memcpy(T.W.hosts, hosts, sizeof(cmlhip_frame_pose) * (size_t)n_hosts);
        if (half) k_trace_points_tracked<true><<<cml_div_up(n, 4), 256, 0, c->stream>>>(A, T);
        else k_trace_points_tracked<false><<<cml_div_up(n, 4), 256, 0, c->stream>>>(A, T);
    } else if (n > 0) {
        TracePoseArgs PA;
        PA.pose0 = pose0; PA.ref = *reference; for (int k = 0; k < 4; k++) PA.K[k] = K[k];
        PA.n_hosts = n_hosts; PA.hosts = hosts_dev; PA.pairs = c->tr_pairs.as<cmlhip_trace_pair>();
        k_trace_pairs_from_tracker<<<1, 64, 0, c->stream>>>(PA);
        if (half) k_trace_points<true><<<cml_div_up(n, 4), 256, 0, c->stream>>>(A);
        else k_trace_points<false><<<cml_div_up(n, 4), 256, 0, c->stream>>>(A);
    }
    TracePublish PB;
    PB.pose0 = pose0; PB.ref = *reference; for (int k = 0; k < 4; k++) PB.K[k] = K[k];
    PB.n_hosts = n_hosts; PB.hosts = hosts_dev; PB.pairs = (from_tracker || n_hosts > TR_INLINE_HOSTS) ? c->tr_pairs.as<cmlhip_trace_pair>() : nullptr; PB.counts = c->tr_counts.as<int>(); PB.out_counts = static_cast<int*>(c->tr_host_dev);
    PB.out_pairs = reinterpret_cast<cmlhip_trace_pair*>(static_cast<char*>(c->tr_host_dev) + 64);
    if ((rc = cml_done_embed(c, &PB.ticket, &PB.ticket_word))) return rc;      // the ticket the tracker's wait (cmlhip_tracker_optimize_wait) then waits for: one host wait for the frame
    k_trace_publish<<<1, 64, 0, c->stream>>>(PB);
    CML_CHECK(c, hipGetLastError());
    c->tr_spec_pending = true; c->tr_spec_hosts = n_hosts; c->tr_spec_skip = skip_host;
    return CMLHIP_OK;
}

int cmlhip_tracer_trace_resident_finish(cmlhip_ctx* c, int keep, int counts[6], cmlhip_trace_pair* pairs_out) { CML_DEV(c);
    if (!c) return CMLHIP_ERR_INVALID;
    CML_REQUIRE(c, c->tr_spec_pending, CMLHIP_ERR_INVALID, "cmlhip_tracer_trace_resident_finish: no speculative trace in flight");
    int rc;
    if (c->done_pending && (rc = cml_done_wait(c))) return rc;       // (already waited for by cmlhip_tracker_optimize_wait in the usual order)
    c->tr_spec_pending = false;
    if (keep) {
        if (counts) memcpy(counts, c->tr_host, 24);
        if (pairs_out) memcpy(pairs_out, static_cast<char*>(c->tr_host) + 64, sizeof(cmlhip_trace_pair) * (size_t)c->tr_spec_hosts);
        return CMLHIP_OK;
    }
    const int n = c->tr_resident_n;                          // the caller selected another try: every traced point gets back what it held before
    if (n > 0) {
        k_trace_rollback<<<cml_div_up(n, 256), 256, 0, c->stream>>>(c->tr_resident.as<cmlhip_immature_point>(), c->tr_journal.as<TraceJournal>(), n, c->tr_spec_skip);
        CML_CHECK(c, hipGetLastError());
    }
    return CMLHIP_OK;
}

static int optimize_immature_common(cmlhip_ctx* c, int N, const uint64_t* image_ids, const double K[4], const cmlhip_activation_pair* pairs,
                                    const cmlhip_tracer_params* prm, int min_obs, int n, const cmlhip_immature_point* points, const int* slots, int* result,
                                    float* idepth, int* res_state) {
    if (n == 0) return CMLHIP_OK;
    OptArgs A;
    memset(&A, 0, sizeof A);
    for (int t = 0; t < N; t++) {
        const Pyramid* py = cml_find_pyr(c, image_ids[t]);
        CML_REQUIRE(c, py && py->lv[0].grad, CMLHIP_ERR_NOT_FOUND, "window image not in the pyramid cache");
        if (t == 0) { A.w = py->lv[0].w; A.h = py->lv[0].h; }
        CML_REQUIRE(c, py->lv[0].w == A.w && py->lv[0].h == A.h, CMLHIP_ERR_INVALID, "window images differ in size");
        A.img[t] = py->lv[0].grad;
    }
    int rc;
    const size_t out_bytes = (size_t)n * (8 + 4 * (size_t)N);
    if ((rc = cml_ensure(c, c->tr_pairs, sizeof(cmlhip_activation_pair) * (size_t)N * N))) return rc;
    if ((rc = tr_ensure(c, c->tr_out, out_bytes))) return rc;
    if (points) {
        for (int i = 0; i < n; i++) if (points[i].host < 0 || points[i].host >= N) { c->err = "immature point host out of range"; return CMLHIP_ERR_INVALID; }
        if ((rc = tr_ensure(c, c->tr_points, sizeof(cmlhip_immature_point) * (size_t)n))) return rc;
        A.pts = c->tr_points.as<cmlhip_immature_point>(); A.slots = nullptr;
    } else {
        for (int i = 0; i < n; i++) if (slots[i] < 0 || slots[i] >= c->tr_resident_n) { c->err = "immature point slot out of range"; return CMLHIP_ERR_INVALID; }
        if ((rc = tr_ensure(c, c->tr_edit, 4 * (size_t)n))) return rc;
        A.pts = c->tr_resident.as<cmlhip_immature_point>(); A.slots = c->tr_edit.as<int>();
    }
    cml_h2d_batch_begin(c);                                  // the candidates (records or slots) and the pairs: one packed upload
    rc = points ? cml_h2d(c, c->tr_points.p, points, sizeof(cmlhip_immature_point) * (size_t)n) : cml_h2d(c, c->tr_edit.p, slots, 4 * (size_t)n);
    if (!rc) rc = cml_h2d(c, c->tr_pairs.p, pairs, sizeof(cmlhip_activation_pair) * (size_t)N * N);
    { const int rf = cml_h2d_batch_flush(c); if (rc || rf) return rc ? rc : rf; }
    c->tr_req_consumed = false;                              // (pairs a tracker launch left in tr_pairs are overwritten)
    A.N = N; A.n = n; A.min_obs = min_obs;
    for (int k = 0; k < 4; k++) A.K[k] = K[k];
    A.pairs = c->tr_pairs.as<cmlhip_activation_pair>(); A.P = *prm;
    A.result = c->tr_out.as<int>(); A.idepth = reinterpret_cast<float*>(A.result + n); A.res_state = A.result + 2 * (size_t)n;
    if (c->lim.texel_format == CMLHIP_TEXEL_F16) k_optimize_immature<true><<<cml_div_up(n, 4), 256, 0, c->stream>>>(A);
    else k_optimize_immature<false><<<cml_div_up(n, 4), 256, 0, c->stream>>>(A);
    CML_CHECK(c, hipGetLastError());
    cml_d2h_batch_begin(c);
    cml_d2h(c, result, A.result, 4 * (size_t)n);
    cml_d2h(c, idepth, A.idepth, 4 * (size_t)n);
    cml_d2h(c, res_state, A.res_state, 4 * (size_t)n * N);
    return cml_d2h_batch_flush(c);
}
int cmlhip_optimize_immature_points(cmlhip_ctx* c, int N, const uint64_t* image_ids, const double K[4], const cmlhip_activation_pair* pairs,
                                    const cmlhip_tracer_params* prm, int min_obs, int n, const cmlhip_immature_point* points, int* result,
                                    float* idepth, int* res_state) { CML_DEV(c);
    if (!c || N < 2 || N > CMLHIP_MAX_FRAMES || !image_ids || !K || !pairs || !prm || n < 0 || (n > 0 && (!points || !result || !idepth || !res_state)))
        return CMLHIP_ERR_INVALID;
    return optimize_immature_common(c, N, image_ids, K, pairs, prm, min_obs, n, points, nullptr, result, idepth, res_state);
}
// the same for points of the device-resident set, named by their slots: nothing of the 232-byte records travels (the candidates' host indices are the
// ones the set was last edited with — cmlhip_tracer_edit_points / _set_points — and must refer to the frame list `image_ids`)
int cmlhip_optimize_immature_points_resident(cmlhip_ctx* c, int N, const uint64_t* image_ids, const double K[4], const cmlhip_activation_pair* pairs,
                                             const cmlhip_tracer_params* prm, int min_obs, int n, const int* slots, int* result, float* idepth, int* res_state) { CML_DEV(c);
    if (!c || N < 2 || N > CMLHIP_MAX_FRAMES || !image_ids || !K || !pairs || !prm || n < 0 || (n > 0 && (!slots || !result || !idepth || !res_state)))
        return CMLHIP_ERR_INVALID;
    CML_REQUIRE(c, !c->tr_spec_pending, CMLHIP_ERR_INVALID, "cmlhip_optimize_immature_points_resident: a speculative trace is in flight");
    return optimize_immature_common(c, N, image_ids, K, pairs, prm, min_obs, n, nullptr, slots, result, idepth, res_state);
}

}  // extern "C"
