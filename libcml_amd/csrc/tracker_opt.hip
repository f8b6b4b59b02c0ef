// tracker_opt.hip — DSOTracker::optimize (TR.cpp:15-246) resident on the device, batched over motion hypotheses.
//
// cmlhip_tracker_eval is one launch + one host round trip per Levenberg-Marquardt trial: 25-40 synchronous calls per optimize,
// and DSOTracker::trackWithMotionModel (DSOTracker.h:238-383) runs optimize once per motion hypothesis, one after the other.
// Here ONE workgroup runs the whole coarse-to-fine loop of one hypothesis — residual / Hessian evaluation by all 512 lanes (the
// per-point arithmetic and the matrix-core reduction of k_tracker_eval), the 8x8 pivoted LDL^T (Eigen semantics), SE(3)
// exponential, accept / reject and the lambda schedule on lane 0 in fp64 — and the grid is the hypothesis list: the hypotheses the
// reference tries in sequence are evaluated speculatively side by side, one CU each, in one launch and one readback.  The only
// coupling between the reference's tries — the abort of a try whose level rmse exceeds 1.5 x the best try's so far (TR.cpp:183-189)
// — only shortens a try, so it is applied afterwards by the caller from the per-pass rmse values this kernel records
// (cml_amd::DSOTracker::trackWithMotionModelBatched replays DSOTracker.h:262-313 on the results).
#include "cmlhip_internal.h"
#include <chrono>
#include "../host/se3.h"
#include <cstdlib>
#include <atomic>

#pragma clang fp contract(off)

using cml_amd::SE3;

#define TO_THREADS 512
#define TO_WAVES (TO_THREADS / 64)
#define TO_LD 17
#define TO_NRED 56
#define TO_PARTS 8                   // parts of a level with more than split_min reference points (fixed: the sums must not depend on G)
typedef float to_float4 __attribute__((ext_vector_type(4)));

struct TrkOptArgs {
    const void* img[5]; int w[5], h[5]; const float* uvic[5]; int n[5];
    int levels, opt_a, opt_b, n_hyp;
    double K0[4], ref_a, ref_b, ref_t, new_t, init_a, init_b, sat_th;
    float huber, cutoff_base, scale_rot, scale_trans, scale_a, scale_b;
    const cmlhip_tracker_hypothesis* hyp;
    cmlhip_tracker_opt_result* out;                    // n_hyp x G results (every workgroup of a hypothesis runs the whole loop; the host keeps the first)
    cmlhip_tracker_opt_result* out_host;               // n_hyp results in mapped host memory: written by the first workgroup of each hypothesis, no copy back
    // round 3: G workgroups per hypothesis.  A level with more than split_min reference points is evaluated in G parts, the 56 sums are
    // exchanged through memory (device-scope stores + a ticket per workgroup) and added in workgroup order by EVERY workgroup — all of
    // them then run the identical Levenberg-Marquardt algebra on identical numbers, no second exchange; smaller levels are evaluated
    // whole by every workgroup (no exchange at all)
    int G, split_min;
    float* xch;                                        // [n_hyp][2 parities][TO_PARTS][64] 8-byte words {sum | launch number, exchange number}
    int epoch;                                         // launch number of this context (16 bits used)
    int* tick;                                         // [n_hyp][G]
    int* late;                                         // mapped host word: set when an exchange gave up waiting (a workgroup of the hypothesis never became resident)
    // cmlhip_tracker_set_early_exit: the reference tries its hypotheses one after the other and leaves the loop behind the first good one
    // (DSOTracker.h:306-309).  early_flag (device word, holds the launch number once raised) is raised by hypothesis 0 when it ends correct
    // with E/n of level 0 below early_rmse; the other hypotheses see it at their next exchange and give up (n_steps = -1).
    int* early_flag; float early_rmse;
    double* pose0;                                     // device copy {R[9], t[3], a, b} of the FIRST hypothesis' result: read by the trace enqueued behind the batch (tracer.hip)
    // cmlhip_tracer_tracked_prepare: the window of the trace enqueued behind this batch — the workgroup that ends hypothesis 0 forms the pairs
    // host -> frame right behind its result (a lane per host), the trace reads them like any caller's pairs
    cmlhip_trace_pair* tr_pairs;                       // null: no request
    TrackedReq req;
};

// per-evaluation constants exactly as TR.cpp:260-278,426-429 forms them (float), shared by the workgroup
struct ToEval {
    float RKi[9], Ki[9], t[3], fxl, fyl, cxl, cyl, a0, a1, fxh, fyh, b0, a_h, maxEnergy;
    double huber_d, cutoff_d, cutoff_base_d;
    int level, n, w, h;
    const void* img; const float* uvic;
};
// Levenberg-Marquardt state of the hypothesis (lane 0 owns it)
struct ToState {
    double cur_q[4], cur_t[3], nw_q[4], nw_t[3];       // poses as plain numbers (__shared__ objects cannot have initialisers)
    double a, b, na, nb, lambda, H[64], bv[8], Hn[64], bn[8], levelCutoffRepeat[5];
    float E[5], E_new[5], flow[3], flow_new[3];
    int nT[5], nS[5], nR[5], nT_new[5], nS_new[5], nR_new[5], iterations[5];
    int ctrl[2], n_steps, n_pass, haveRepeated;
    long long t_eval, t_alg, t_mark;                   // wall_clock64 ticks (10 ns): evaluations / lane-0 algebra
#ifdef TO_PROFILE
    long long t_ldlt, t_pose, t_fin, t_p[4], t_e[4];
#endif
};

// The reference list and the image are DEVICE memory, but their pointers reach the evaluation through a struct in LDS: generic pointers, every access a
// FLAT one (it waits for the LDS counter as well and takes the slower address path).  The loads say "global" themselves.
typedef unsigned to_uint2 __attribute__((ext_vector_type(2)));
#if __HIP_DEVICE_COMPILE__
#define TO_GLOBAL __attribute__((address_space(1)))
#else
#define TO_GLOBAL
#endif
template <bool HALF>
__device__ __forceinline__ float4 to_texel(const void* img, size_t i) {
    if (HALF) {
        const to_uint2 v = ((const TO_GLOBAL to_uint2*)img)[i];
        unsigned vx = v.x, vy = v.y;
        __half2 a = *reinterpret_cast<__half2*>(&vx), b = *reinterpret_cast<__half2*>(&vy);
        return make_float4(__low2float(a), __high2float(a), __low2float(b), 0.f);
    }
    const to_float4 q = ((const TO_GLOBAL to_float4*)img)[i];
    return make_float4(q.x, q.y, q.z, q.w);
}

// Eigen compute_inverse_size3 (cofactors * 1/det) in float, as Matrix33f::inverse() at TR.cpp:261
__device__ void to_inv3f(const float m[9], float o[9]) {
#define MM(i, j) m[(i) * 3 + (j)]
#define COF(i, j) (MM(((i) + 1) % 3, ((j) + 1) % 3) * MM(((i) + 2) % 3, ((j) + 2) % 3) - MM(((i) + 1) % 3, ((j) + 2) % 3) * MM(((i) + 2) % 3, ((j) + 1) % 3))
    const float c0 = COF(0, 0), c1 = COF(1, 0), c2 = COF(2, 0);
    const float det = c0 * MM(0, 0) + (c1 * MM(1, 0) + c2 * MM(2, 0));
    const float invdet = 1.0f / det;
    o[0] = c0 * invdet; o[1] = c1 * invdet; o[2] = c2 * invdet;
    o[3] = COF(0, 1) * invdet; o[4] = COF(1, 1) * invdet; o[5] = COF(2, 1) * invdet;
    o[6] = COF(0, 2) * invdet; o[7] = COF(1, 2) * invdet; o[8] = COF(2, 2) * invdet;
#undef COF
#undef MM
}

// x = A.ldlt().solve(b), Eigen 3.4.0 semantics (Cholesky/LDLT.h:300-396,560-600), n <= 8 — the host mirror's ldltSolveSmall —
// by the 64 lanes of ONE wave, in registers: lane (l & 7) holds row l & 7 of the (symmetric, mirrored from the lower
// triangle Eigen reads) damped system.  Every operation of the scalar algorithm keeps its operands and its order — the left-looking update
// of column k is one lane per row, the sums over j < k run in j order, `/= akk` is the IEEE division — but nothing goes through
// LDS: the pivot search, row k of L and the pivots travel by v_readlane (k is a compile-time lane, the pivot index a scalar),
// the symmetric transposition is a two-lane row exchange plus an in-lane column exchange.  The substitutions run on wave-uniform
// copies (all lanes the same numbers).  17 us -> 2 us per Levenberg-Marquardt trial against lane 0 alone on LDS scratchpads.
// Hs: 8x8 row-major (LDS), rows/cols taken through map[] (the 7x7 / stitched / 6x6 variants of TR.cpp:96-119), diagonal * (1 + lambda).
__device__ __forceinline__ double to_rl(double v, int lane) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}
__device__ bool to_ldlt_solve_wave(const double* Hs, const double* bs, const double lambda, const int n, const int map6, double (&xs)[8]) {
    const int i = threadIdx.x & 7;
    auto mp = [&](int a) { return (a == 6) ? map6 : a; };
    double row[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int r = max(i, j), c = min(i, j);
        double v = (r < n) ? Hs[mp(r) * 8 + mp(c)] : 0.0;
        if (r == c) v *= (1 + lambda);
        row[j] = (r < n) ? v : 0.0;
    }
    int tr[8]; double dd[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { tr[k] = k; dd[k] = 0.0; }
    bool stop = false;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k >= n || stop) continue;                        // wave-uniform
        int big = k; double best = fabs(to_rl(row[k], k));
#pragma unroll
        for (int ii = k + 1; ii < 8; ii++)
            if (ii < n) { const double v = fabs(to_rl(row[ii], ii)); if (v > best) { best = v; big = ii; } }
        tr[k] = big;
        if (big != k) {                                      // symmetric transposition k <-> big
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const double a = to_rl(row[j], k), bq = to_rl(row[j], big);
                row[j] = (i == k) ? bq : ((i == big) ? a : row[j]);
            }
#pragma unroll
            for (int j = k + 1; j < 8; j++)
                if (j == big) { const double t = row[k]; row[k] = row[j]; row[j] = t; }
        }
        double akk = to_rl(row[k], k);
        if (k > 0) {
            double temp[8], s = 0;
#pragma unroll
            for (int j = 0; j < k; j++) { const double lkj = to_rl(row[j], k); temp[j] = dd[j] * lkj; s += lkj * temp[j]; }
            akk -= s;
            double s2 = 0;
#pragma unroll
            for (int j = 0; j < k; j++) s2 += row[j] * temp[j];
            if (i > k) row[k] -= s2;
        }
        if (i == k) row[k] = akk;
        if (k == 0 && !(fabs(akk) > 0.0)) {
#pragma unroll
            for (int j = 0; j < 8; j++) tr[j] = j;
            stop = true;
        } else if (fabs(akk) > 0.0) {
            if (i > k) row[k] /= akk;
        }
        dd[k] = akk;
    }
    if (stop) {                                              // Eigen leaves the matrix untouched: D is its diagonal, L what sits below it
#pragma unroll
        for (int k = 0; k < 8; k++) dd[k] = to_rl(row[k], k);
    }
    // substitutions on wave-uniform copies: x = P^T L^-T D^-1 L^-1 P b (LDLT.h:560-600)
#pragma unroll
    for (int k = 0; k < 8; k++) xs[k] = (k < n) ? -bs[mp(k)] : 0.0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (k < n && tr[k] != k) {
#pragma unroll
            for (int j = k + 1; j < 8; j++) if (j == tr[k]) { const double t = xs[k]; xs[k] = xs[j]; xs[j] = t; }
        }
    double Lm[8][8];
#pragma unroll
    for (int a = 1; a < 8; a++)
#pragma unroll
        for (int j = 0; j < a; j++) Lm[a][j] = to_rl(row[j], a);
#pragma unroll
    for (int a = 0; a < 8; a++) if (a < n) { double t = xs[a];
#pragma unroll
        for (int j = 0; j < a; j++) t -= Lm[a][j] * xs[j];
        xs[a] = t; }
#pragma unroll
    for (int a = 0; a < 8; a++) if (a < n) xs[a] = (fabs(dd[a]) > 2.2250738585072014e-308) ? xs[a] / dd[a] : 0.0;
#pragma unroll
    for (int a = 7; a >= 0; a--) if (a < n) { double t = xs[a];
#pragma unroll
        for (int j = a + 1; j < 8; j++) if (j < n) t -= Lm[j][a] * xs[j];
        xs[a] = t; }
#pragma unroll
    for (int k = 7; k >= 0; k--)
        if (k < n && tr[k] != k) {
#pragma unroll
            for (int j = k + 1; j < 8; j++) if (j == tr[k]) { const double t = xs[k]; xs[k] = xs[j]; xs[j] = t; }
        }
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 8; a++) if (a < n && !isfinite(xs[a])) ok = false;
    return ok;
}
// The same solve without the per-step pivot search and transposition.  Eigen's unblocked LDL^T is left-looking: when step k looks for
// its pivot, the diagonal below it still holds the ORIGINAL entries (only column k and (k,k) are ever updated, LDLT.h:330-380), so the
// pivot sequence is the order of the original |diagonal| — known before the first step.  When the active diagonal entries are pairwise
// different (no tie for the search's first-maximum rule to break, no NaN) the sequence is their descending order: the wave ranks them
// with one ballot, loads the matrix already permuted (P A P^T, pure data movement), and factorises without searches or exchanges — the
// arithmetic of every step is the one the pivoting form performs on the same numbers, so L, D and x are bit-identical.  Ties / NaN /
// an all-zero diagonal: `handled` stays false and the caller runs the pivoting form.  s_x: 64 doubles of LDS scratch.
__device__ bool to_ldlt_solve_wave_sorted(const double* Hs, const double* bs, const double lambda, const int n, const int map6, double (&xs)[8],
                                          double* s_x /* 64 doubles */, bool& handled) {
    const int l = threadIdx.x & 63, i = l & 7, a = l >> 3;
    auto mp = [&](int q) { return (q == 6) ? map6 : q; };
    handled = false;
    const double da = (a < n) ? fabs(Hs[mp(a) * 8 + mp(a)] * (1 + lambda)) : 0.0;
    const double di = (i < n) ? fabs(Hs[mp(i) * 8 + mp(i)] * (1 + lambda)) : 0.0;
    const bool act = a < n && i < n;
    const unsigned long long gt = __ballot(act && da > di);                               // bit a * 8 + i: |d[a]| > |d[i]|
    const unsigned long long bad = __ballot(act && a != i && !(da > di) && !(di > da));   // a tie or a NaN
    const unsigned long long zero = __ballot(act && a == i && !(da > 0.0));               // (the all-zero matrix is a tie already unless n == 1)
    if (bad || zero) return false;
    // pos[q] = number of active entries larger than d[q] = the step that picks q;  idx[k] = the entry step k picks.  The inverse goes through eight ints of
    // the scratchpad (lane q stores q at pos[q], every lane reads the eight back) — as 64 compare / select pairs on wave-uniform copies it was a tenth of
    // this solve's instructions, and a lone wave pays an issue slot for each
    int* s_pick = reinterpret_cast<int*>(s_x);
    if (a == 0 && i < n) s_pick[__popcll((gt >> i) & 0x0101010101010101ull)] = i;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // one wave, in-order LDS: compiler ordering only
    __builtin_amdgcn_wave_barrier();
    int idx_u[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { const int v = s_pick[k]; idx_u[k] = (k < n) ? v : k; }
    const int idx_k = (i < n) ? s_pick[i] : i;
    __builtin_amdgcn_wave_barrier();                        // (the scratchpad is written again further down)
    double row[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int r = max(idx_k, idx_u[j]), c = min(idx_k, idx_u[j]);
        const bool in = i < n && j < n;
        double v = in ? Hs[mp(r) * 8 + mp(c)] : 0.0;
        if (i == j) v *= (1 + lambda);
        row[j] = in ? v : 0.0;
    }
    double dd[8];
#pragma unroll
    for (int k = 0; k < 8; k++) dd[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k >= n) continue;                                // wave-uniform
        if (k > 0) {
            double temp[8];
#pragma unroll
            for (int j = 0; j < k; j++) { const double lkj = to_rl(row[j], k); temp[j] = dd[j] * lkj; }
            double s2 = 0;
#pragma unroll
            for (int j = 0; j < k; j++) s2 += row[j] * temp[j];
            if (i >= k) row[k] -= s2;                         // lane k: A_kk - sum_j L_kj temp_j — its row[j] IS L_kj: the pivot's own sum, same operands in the same order
        }
        const double akk = to_rl(row[k], k);
        if (fabs(akk) > 0.0) {
            if (i > k) row[k] /= akk;
        }
        dd[k] = akk;
    }
    // substitutions: x = P^T L^-T D^-1 L^-1 P b (LDLT.h:560-600); P b is b in pick order.  Round 6: lane i carries unknown i (rows of L are already
    // lane-resident) instead of every lane replaying all eight on wave-uniform copies — the 56 broadcasts that built those copies, and seven of the
    // eight IEEE divisions, were half of this solve's 3 us.  Every value keeps its operations and their order: the forward sum of row i runs over
    // j ascending (multiply, then subtract), the division is lane i's own, the backward sum over j ascending from i + 1.
    double t = (i < n) ? -bs[mp(idx_k)] : 0.0;
#pragma unroll
    for (int j = 0; j < 7; j++) {
        if (j + 1 >= n) break;                               // wave-uniform
        const double xj = to_rl(t, j);                       // y_j is final: lane j has subtracted every term below j
        const double p = row[j] * xj;
        if (i > j && i < n) t = t - p;
    }
    double piv = 0.0;                                        // D_ii: the pivot this lane's row was divided through at step i
#pragma unroll
    for (int k = 0; k < 8; k++) if (i == k) piv = dd[k];
    t = (i < n) ? ((fabs(piv) > 2.2250738585072014e-308) ? t / piv : 0.0) : 0.0;
    // column i of L for the backward sum: L_ji sits in lane j — through the 8 x 8 scratch (one replica writes)
    if (l < 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) s_x[i * 8 + j] = row[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // one wave, in-order LDS: compiler ordering only
    __builtin_amdgcn_wave_barrier();
    double Lt[8];
#pragma unroll
    for (int j = 1; j < 8; j++) Lt[j] = s_x[j * 8 + i];     // L_ji (used for j > i only)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 8; k++) xs[k] = 0.0;
#pragma unroll
    for (int q = 7; q >= 0; q--) {
        if (q >= n) continue;                                // wave-uniform
        double u = t;                                        // (lane q's chain is the one that counts: x_q = z_q - sum_{j > q} L_jq x_j, j ascending)
#pragma unroll
        for (int j = q + 1; j < 8; j++) if (j < n) u = u - Lt[j] * xs[j];
        xs[q] = to_rl(u, q);
    }
    // P^T: entry k of the permuted solution belongs to unknown idx[k] — through LDS (a register array cannot be indexed by idx)
    double mine = 0.0;
#pragma unroll
    for (int k = 0; k < 8; k++) if (i == k) mine = xs[k];
    if (l < 8) s_x[idx_k] = mine;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // one wave, in-order LDS: compiler ordering only
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 8; k++) xs[k] = s_x[k];
    __builtin_amdgcn_wave_barrier();
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 8; q++) if (q < n && !isfinite(xs[q])) ok = false;
    handled = true;
    return ok;
}
// hessian.inverse() (TR.cpp:243) on ONE WAVE: Gauss-Jordan with partial pivoting, lane (r, c) = l >> 3, l & 7 carries A[r][c] and
// Ai[r][c] in registers.  Per step: the pivot search on wave-uniform copies of column k (first largest |A[i][k]|, i >= k), the row
// exchange as one cross-lane move, the IEEE divisions of row k, and every other row's update with its own multiplier — the arithmetic
// of each entry is that of the scalar loop (same operations on the same numbers in the same order); only the lanes differ.  On lane 0
// alone, on LDS scratchpads, this inverse was 36 of the kernel's 245 us (behind the last in-kernel clock, so neither "evaluation" nor
// "algebra" showed it).
__device__ __forceinline__ double to_shfl_d(double v, int lane) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_ds_bpermute(lane << 2, u.i[0]);
    u.i[1] = __builtin_amdgcn_ds_bpermute(lane << 2, u.i[1]);
    return u.d;
}
__device__ __forceinline__ double to_inverse8_wave(const double* Ain /* 64, LDS */, const int l) {
    const int r = l >> 3, c = l & 7;
    double a = Ain[l], ai = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int p = k;
        double best = fabs(to_rl(a, k * 8 + k));
#pragma unroll
        for (int i = k + 1; i < 8; i++) {
            const double v = fabs(to_rl(a, i * 8 + k));
            if (v > best) { best = v; p = i; }
        }
        p = __builtin_amdgcn_readfirstlane(p);
        if (p != k) {                                      // wave-uniform
            const int sr = (r == k) ? p : ((r == p) ? k : r);
            a = to_shfl_d(a, sr * 8 + c); ai = to_shfl_d(ai, sr * 8 + c);
        }
        const double d = to_rl(a, k * 8 + k);
        const double qa = a / d, qi = ai / d;
        if (r == k) { a = qa; ai = qi; }
        const double f = to_shfl_d(a, r * 8 + k);
        const double pk = to_shfl_d(a, k * 8 + c), pik = to_shfl_d(ai, k * 8 + c);
        if (r != k && f != 0) { a -= f * pk; ai -= f * pik; }
    }
    return ai;
}

// SE3::exp (host/se3.h, Sophus se3.hpp:224-257) for LANES 0-3 of a wave running the same chain on the same numbers.  A lone lane pays a wave's issue
// slot per instruction, so work of the same SHAPE on different operands goes to different lanes of one instruction stream: the half-angle and the
// full-angle sincos are ONE call (even lanes theta / 2, odd lanes theta), sqrt(theta^2) and the sqrt of the increment's norm the caller needs are ONE
// square root, the three IEEE divisions (sin(theta / 2) / theta, (1 - cos) / theta^2, (theta - sin) / theta^3) ONE — each lane hands its result to the
// others by v_readlane.  Same operations on the same operands per value; 580 -> ~400 instructions on the chain (18.4 -> 13 us over a 15-trial frame).
__device__ __forceinline__ SE3 to_se3_exp_lanes(const double xi[6], const double nrm2, double& nrm, const int lane) {
    const double eps = 1e-10;
    const double* om = xi + 3;
    const double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    const double rt = sqrt((lane & 1) ? nrm2 : th2);
    nrm = to_rl(rt, 1);
    SE3 T;
    double O[9], O2[9], V[9];
    SE3::hat(om, O);
    SE3::mm(O, O, O2);
    if (th2 < eps * eps) {                                    // (wave-uniform: the lanes hold the same increment) — theta = 0 in se3.h
        const double p4 = th2 * th2;
        const double imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * p4, real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * p4;
        T.q[0] = real; T.q[1] = imag * om[0]; T.q[2] = imag * om[1]; T.q[3] = imag * om[2];
        T.matrix(V);
    } else {
        const double theta = to_rl(rt, 0);
        double s_, c_;
        sincos((lane & 1) ? theta : 0.5 * theta, &s_, &c_);
        const double sh_ = to_rl(s_, 0), ch_ = to_rl(c_, 0), st_ = to_rl(s_, 1), ct_ = to_rl(c_, 1);
        const double t2 = theta * theta, t3 = t2 * theta;
        const double num = lane == 0 ? sh_ : (lane == 1 ? (1.0 - ct_) : (theta - st_)), den = lane == 0 ? theta : (lane == 1 ? t2 : t3);
        const double qt = num / den;
        const double imag = to_rl(qt, 0), a = to_rl(qt, 1), b = to_rl(qt, 2);
        T.q[0] = ch_; T.q[1] = imag * om[0]; T.q[2] = imag * om[1]; T.q[3] = imag * om[2];
        if (theta < eps) T.matrix(V);                         // (se3.h tests theta here, theta^2 above)
        else for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * O[i] + b * O2[i];
    }
    SE3::mv(V, xi, T.t);
    return T;
}

// lane 0: the constants of one evaluation (TR.cpp:260-278, 426-429; InternalCalibration.h:116-127; Exposure.h:119-123)
__device__ __forceinline__ SE3 to_pose(const double q[4], const double t[3]) {
    SE3 T;
    for (int i = 0; i < 4; i++) T.q[i] = q[i];
    for (int i = 0; i < 3; i++) T.t[i] = t[i];
    return T;
}
__device__ __forceinline__ void to_store(const SE3& T, double q[4], double t[3]) {
    for (int i = 0; i < 4; i++) q[i] = T.q[i];
    for (int i = 0; i < 3; i++) t[i] = T.t[i];
}
__device__ void to_prepare(const TrkOptArgs& A, ToEval& ev, int level, const SE3& T, double a, double b, double cutoff_mult, const bool same_level = false) {
    double R[9];
    T.matrix(R);
    const double affA = exp(a - A.ref_a) * A.new_t / A.ref_t, affB = b - affA * A.ref_b;      // reference->getExposure().to(exposure)
    float Rf[9];
    if (!same_level) {                                      // the level's pinhole and its inverse: once per level, not once per trial (same values)
        const double d = (double)(1 << level);
        const double K[4] = {A.K0[0] / d, A.K0[1] / d, (A.K0[2] + 0.5) / d - 0.5, (A.K0[3] + 0.5) / d - 0.5};
        float Kf[9] = {(float)K[0], 0, (float)K[2], 0, (float)K[1], (float)K[3], 0, 0, 1};
        to_inv3f(Kf, ev.Ki);
        ev.fxl = Kf[0]; ev.fyl = Kf[4]; ev.cxl = Kf[2]; ev.cyl = Kf[5];
        ev.fxh = (float)K[0]; ev.fyh = (float)K[1];
    }
    for (int i = 0; i < 9; i++) Rf[i] = (float)R[i];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) ev.RKi[i * 3 + j] = Rf[i * 3] * ev.Ki[j] + (Rf[i * 3 + 1] * ev.Ki[3 + j] + Rf[i * 3 + 2] * ev.Ki[6 + j]);   // Matrix33f product, Eigen order
    for (int i = 0; i < 3; i++) ev.t[i] = (float)T.t[i];
    ev.a0 = (float)affA; ev.a1 = (float)affB;
    ev.b0 = (float)A.ref_b; ev.a_h = (float)affA;
    const float cutoff = (float)((double)A.cutoff_base * cutoff_mult);                      // mCutoffThreshold.f() * levelCutoffRepeat[level]
    ev.huber_d = (double)A.huber; ev.cutoff_d = (double)cutoff; ev.cutoff_base_d = (double)A.cutoff_base;
    ev.maxEnergy = (float)(2.0f * ev.huber_d * ev.cutoff_d - ev.huber_d * ev.huber_d);
    ev.level = level; ev.n = A.n[level]; ev.w = A.w[level]; ev.h = A.h[level]; ev.img = A.img[level]; ev.uvic = A.uvic[level];
}

// all lanes: computeResidual + computeHessian over the level's list (the per-point arithmetic and the reduction layout of
// k_tracker_eval, tracker.hip); leaves the 56 sums in s_red
// the sums of this workgroup's part -> the sums of the level, in every workgroup of the hypothesis (see TrkOptArgs::G)
__device__ __forceinline__ float to_exchange(float* __restrict__ xch, int* __restrict__ tick, const int g, const int G, const int seq, const float* __restrict__ s_part, const int epoch, int* __restrict__ late_flag,
                                             const int give_up_mine, int* __restrict__ s_abort) {
    // Every sum of every PART travels as ONE self-validating device-scope word {value | seq << 32} (past the non-coherent caches; valid on
    // its own): the writers neither wait for acknowledgements nor publish a ticket, the readers poll the eight words of their sum directly
    // and add them in PART order (round 4: the parts are a property of the level — eight, whatever G is — so the sums of a hypothesis do
    // not depend on the batch it travels in; a workgroup owns the parts s = g, g + G, ...).
    // (First form: device-scope stores, a release fence — an L2 write-back —, barrier, ticket; readers polled the tickets, fenced (acquire:
    // an L2 invalidation) and then fetched the sums: two fences and a dependent trip more per exchange.)
    // A workgroup can only write the sums of exchange seq + 2 — the next use of this parity's slots — after it has read every
    // part's words of seq + 1, which a workgroup still reading seq has not written yet: the slots are never overwritten under a reader.
    // Only wave 0 takes part (the 56 sums are its lanes' values) and nothing goes through LDS but its own part sums: no workgroup barrier in the exchange.
    (void)tick;
    const int tid = threadIdx.x;
    unsigned long long* base = reinterpret_cast<unsigned long long*>(xch) + (size_t)(seq & 1) * TO_PARTS * 64;
    const unsigned tagv = ((unsigned)epoch << 16) | ((unsigned)seq & 0xffffu);        // launch number | exchange number: words of an earlier call never match
    const unsigned long long tag = (unsigned long long)tagv << 32;
    if (tid < TO_NRED)
        for (int sp = g, k = 0; sp < TO_PARTS; sp += G, k++)
            __hip_atomic_store(base + (size_t)sp * 64 + tid, tag | (unsigned)__float_as_int(s_part[k * 64 + tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // word 56 of part 0: the control word of the hypothesis — written by the workgroup that owns part 0 (g == 0) with the exchange, read by
    // every workgroup with the sums: "give up" (early exit) reaches all G workgroups at the same exchange, so that none is left waiting
    if (tid == TO_NRED && g == 0) __hip_atomic_store(base + TO_NRED, tag | (unsigned)(give_up_mine ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float v = 0.f;
    bool late = false;
    if (tid == TO_NRED) {
        int spins = 0;
        while (true) {
            const unsigned long long w = __hip_atomic_load(base + TO_NRED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 32) == tagv) { if ((unsigned)w & 1u) *s_abort = 1; break; }
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { late = true; break; }
        }
    }
    if (tid < TO_NRED) {
        // the eight words of a sum are requested TOGETHER and polled as a set: one round trip per poll round — polled one after the
        // other each word was a dependent device-scope load of its own, eight round trips even when everything had arrived
        int spins = 0;
        while (true) {
            unsigned long long w[TO_PARTS];
            bool all = true;
#pragma unroll
            for (int q = 0; q < TO_PARTS; q++) w[q] = __hip_atomic_load(base + (size_t)q * 64 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < TO_PARTS; q++) all = all && (unsigned)(w[q] >> 32) == tagv;
            if (all) {
#pragma unroll
                for (int q = 0; q < TO_PARTS; q++) v += __int_as_float((int)(unsigned)w[q]);       // part order, from zero
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { late = true; break; }           // never spin forever: the level then fails (no terms)
        }
    }
    if (__ballot(late)) {                                              // reported to the host as CMLHIP_ERR_TIMEOUT: a scheduling problem, not a tracking failure
        if (late) __hip_atomic_store(late_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return 0.f;
    }
    return v;
}

template <bool HALF>
__device__ void to_eval(const ToEval& E, float (*s_a)[64][TO_LD], float (*s_b)[64][TO_LD], float (*s_tile)[256], float* s_red, float* s_part,
                        const int g, const int G, const int split_min, float* xch, int* tick, int& seq, const int epoch, int* late_flag,
                        const int* early_flag, int* s_abort, long long* te = nullptr) {
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
#ifdef TO_PROFILE
    long long te_last = wall_clock64();
#define TE_STAMP(k) do { if (te && tid == 0) { const long long n_ = wall_clock64(); te[k] += n_ - te_last; te_last = n_; } } while (0)
#else
#define TE_STAMP(k) do { } while (0)
#endif
    // The parts of a level are fixed by its SIZE: eight for a level with more than split_min reference points (part s = the chunks s, s + 8,
    // s + 16, ... of TO_THREADS points), one otherwise — NOT by G.  A workgroup evaluates the parts s = g, g + G, ... (G in {1, 2, 4, 8});
    // the level's sums are the part sums added in part order by every workgroup.  So a hypothesis gives the same bits alone (G = 8),
    // among 50 (G = 4) or among 200 (G = 1): ADVICE round 3.  A one-part level is evaluated whole by every workgroup, without exchange.
    const int NP = E.n > split_min ? TO_PARTS : 1;
    // early exit: the word is requested here, ahead of the evaluation, by the one lane that will pass it on (hypotheses > 0 only: early_flag is null for hypothesis 0)
    int give_up = 0;
    if (early_flag && g == 0 && tid == TO_NRED) give_up = __hip_atomic_load(early_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
    int src = -1;                                                       // wave 0: which tile entry this lane sums
    if (tid < 45) {
        int k = tid, r = 0;
        while (k >= 9 - r) { k -= 9 - r; r++; }
        src = r * 16 + (r + k);
    } else if (tid <= 50) src = 9 * 16 + 10 + (tid - 45);              // E sT sRT sN numTerms numSaturated
    else if (tid == 51) src = 10 * 16 + 9;                              // numRobust
    else if (tid == 52) src = 11 * 16 + 9;                              // numWarped
    int kown = 0;
    for (int sp = (NP > 1 ? g : 0); sp < NP; sp += (NP > 1 ? G : 1), kown++) {
        to_float4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int base = sp * TO_THREADS; base < E.n; base += NP * TO_THREADS) {
            const int i = base + tid;
            float va[16], vb[16];
#pragma unroll
            for (int k = 0; k < 16; k++) { va[k] = 0.f; vb[k] = 0.f; }
            if (i < E.n) {
                va[9] = 1.f; vb[9] = 1.f;
                const to_float4 q = ((const TO_GLOBAL to_float4*)E.uvic)[i];
                const float x = q.x, y = q.y, id = q.z, refColor = q.w;
                if (isfinite(refColor)) {                                           // TR.cpp:301-303
                    float pt[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) pt[k] = (E.RKi[k * 3] * x + (E.RKi[k * 3 + 1] * y + E.RKi[k * 3 + 2] * 1.0f)) + E.t[k] * id;
                    const float u = pt[0] / pt[2], vv = pt[1] / pt[2];
                    const float Ku = E.fxl * u + E.cxl, Kv = E.fyl * vv + E.cyl;
                    const float new_idepth = id / pt[2];
                    if (E.level == 0 && (i % 32) == 0) {                           // flow statistic, TR.cpp:313-344
                        float a[3], b[3], c[3];
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            const float kp = E.Ki[k * 3] * x + (E.Ki[k * 3 + 1] * y + E.Ki[k * 3 + 2] * 1.0f);
                            a[k] = kp + E.t[k] * id; b[k] = kp - E.t[k] * id;
                            c[k] = (E.RKi[k * 3] * x + (E.RKi[k * 3 + 1] * y + E.RKi[k * 3 + 2] * 1.0f)) - E.t[k] * id;
                        }
                        const float KuT = E.fxl * (a[0] / a[2]) + E.cxl, KvT = E.fyl * (a[1] / a[2]) + E.cyl;
                        const float KuT2 = E.fxl * (b[0] / b[2]) + E.cxl, KvT2 = E.fyl * (b[1] / b[2]) + E.cyl;
                        const float Ku3 = E.fxl * (c[0] / c[2]) + E.cxl, Kv3 = E.fyl * (c[1] / c[2]) + E.cyl;
                        float sT = 0, sRT = 0;
                        sT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
                        sT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
                        sRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
                        sRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
                        vb[11] = sT; vb[12] = sRT; vb[13] = 2.f;
                    }
                    if (Ku > 2 && Kv > 2 && Ku < E.w - 3 && Kv < E.h - 3 && new_idepth > 0) {     // TR.cpp:346
                        const int ix = (int)Ku, iy = (int)Kv;
                        const float dx = Ku - (float)ix, dy = Kv - (float)iy, dxdy = dx * dy;
                        const float w00 = 1 - dx - dy + dxdy, w01 = dx - dxdy, w10 = dy - dxdy, w11 = dxdy;
                        const size_t i1 = (size_t)iy * E.w + ix;
                        const float4 ta = to_texel<HALF>(E.img, i1), tb = to_texel<HALF>(E.img, i1 + 1);
                        const float4 tc = to_texel<HALF>(E.img, i1 + E.w), td = to_texel<HALF>(E.img, i1 + E.w + 1);
                        const float h0 = ta.x * w00 + tb.x * w01 + tc.x * w10 + td.x * w11;
                        const float h1 = ta.y * w00 + tb.y * w01 + tc.y * w10 + td.y * w11;
                        const float h2 = ta.z * w00 + tb.z * w01 + tc.z * w10 + td.z * w11;
                        if (isfinite(h0) && isfinite(h1) && isfinite(h2)) {
                            const float residual = h0 - (float)(E.a0 * refColor + E.a1);
                            const float hw = fabs((double)residual) < E.huber_d ? 1.0f : (float)(E.huber_d / fabs((double)residual));
                            if (fabs((double)residual) > E.cutoff_d) {
                                vb[10] = E.maxEnergy; vb[14] = 1.f; vb[15] = 1.f;                  // E, numTerms, numSaturated
                            } else {
                                vb[10] = hw * residual * residual * (2 - hw); vb[14] = 1.f; va[11] = 1.f;   // E, numTerms, numWarped
                                const float ddx = h1 * E.fxh, ddy = h2 * E.fyh;                      // computeHessian lanes, TR.cpp:443-470
                                vb[0] = new_idepth * ddx;
                                vb[1] = new_idepth * ddy;
                                vb[2] = 0.0f - (new_idepth * (u * ddx + vv * ddy));
                                vb[3] = 0.0f - ((u * vv * ddx) + ddy * (1.0f + vv * vv));
                                vb[4] = (u * vv * ddy) + (ddx * (1.0f + u * u));
                                vb[5] = u * ddy - vv * ddx;
                                vb[6] = E.a_h * (E.b0 - refColor);
                                vb[7] = -1.0f;
                                vb[8] = residual;
#pragma unroll
                                for (int r = 0; r < 9; r++) va[r] = vb[r] * hw;
                            }
                            if (fabs((double)residual) <= E.cutoff_base_d) va[10] = 1.f;           // numRobust
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 16; k++) { s_a[wv][l][k] = va[k]; s_b[wv][l][k] = vb[k]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                  // wave-private tiles, in-order LDS: compiler ordering only
            __builtin_amdgcn_wave_barrier();
            const int e = l & 15, kq = l >> 4;
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const float av = s_a[wv][4 * m + kq][e], bv = s_b[wv][4 * m + kq][e];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        TE_STAMP(0);                                                    // setup + this wave's points + matrix-core sums
        if (kown) __syncthreads();                                      // wave 0 has read the tiles of the part before
#pragma unroll
        for (int rg = 0; rg < 4; rg++) s_tile[wv][(4 * (l >> 4) + rg) * 16 + (l & 15)] = acc[rg];
        __syncthreads();
        TE_STAMP(1);                                                    // tiles out, every wave arrived
        // from here on only wave 0 works on the sums: it adds the waves' tiles (wave order) and keeps the part's 56 sums
        if (tid < 64) {
            float v = 0.f;
            if (src >= 0) for (int w = 0; w < TO_WAVES; w++) v += s_tile[w][src];
            s_part[kown * 64 + tid] = v;
        }
    }
    TE_STAMP(2);                                                        // wave 0: the waves' tiles added
    const bool xchg = NP > 1 && G > 1;
    if (xchg) seq++;                                                    // (every thread keeps the exchange number)
    if (tid < 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // s_part: written and read by this wave only
        __builtin_amdgcn_wave_barrier();
        float v;
        if (xchg) v = to_exchange(xch, tick, g, G, seq, s_part, epoch, late_flag, give_up, s_abort);
        else {
            if (G == 1 && give_up) *s_abort = 1;                      // one workgroup per hypothesis: nobody to agree with
                                                          // every part is this workgroup's own: the same sum, part by part, from zero
            v = 0.f;
            for (int k = 0; k < kown; k++) v += s_part[k * 64 + tid];
        }
        if (tid < TO_NRED) s_red[tid] = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // read back by other lanes of this wave only: compiler ordering
        __builtin_amdgcn_wave_barrier();
    }
    TE_STAMP(3);                                                        // exchange between the hypothesis' workgroups / own parts added
}

// wave 0: the level's sums -> Residual slots (lane 0) and the scaled 8x8 system, one entry per lane (TR.cpp:405-414, 472-490);
// s holds the upper triangle of the 9x9 accumulator in row order (entry (r, c >= r) at 9r - r(r-1)/2 + c - r), then the counters
__device__ __forceinline__ void to_finish(const TrkOptArgs& A, const float* s, float& E, int& nT, int& nS, int& nR, float flow[3], double* H, double* b) {
    const int l = threadIdx.x;                             // < 64
    // (the sums as values first: E, nT ... are references into the same LDS the sums live in — read one by one between the stores, every read was a
    //  round trip of its own behind the store before it)
    const float s45 = s[45], s46 = s[46], s47 = s[47], s48 = s[48], s49 = s[49], s50 = s[50], s51 = s[51], s52 = s[52];
    if (l == 0) {
        E = s45; nT = (int)s49; nS = (int)s50; nR = (int)s51;
        flow[0] = s46 / (s48 + 0.1f); flow[1] = 0; flow[2] = s47 / (s48 + 0.1f);
    }
    const int numWarped = (int)s52;
    int npad = numWarped;
    while (npad % 4 != 0) npad++;
    // the scale of unknown k = 0 .. 7 (TR.cpp:472-490), selected per lane from four VALUES: through a lambda indexing the argument struct the selection
    // became a per-lane LOAD from the kernel-argument segment — three dependent trips of ~0.3 us on the critical path of every trial
    const float s_rot = A.scale_rot, s_trans = A.scale_trans, s_a = A.scale_a, s_b = A.scale_b;
    auto h9 = [&](int r, int c) { const int lo = min(r, c), hi = max(r, c); return s[9 * lo - (lo * (lo - 1)) / 2 + (hi - lo)]; };
    const int r = l >> 3, cc = l & 7;
    const float sc_c = cc < 3 ? s_rot : (cc < 6 ? s_trans : (cc == 6 ? s_a : s_b));
    const float sc_r = r < 3 ? s_rot : (r < 6 ? s_trans : (r == 6 ? s_a : s_b));
    H[r * 8 + cc] = ((double)h9(r, cc) / (double)npad) * (double)sc_c * (double)sc_r;
    if (l < 8) b[l] = ((double)h9(l, 8) / (double)npad) * (double)sc_c;              // (l < 8: cc == l)
}

// lane 0's decision to every lane: read between two barriers, so that lane 0 may overwrite it right away
// (two slots used in turn: the writer of hand-over k + 2 has passed the barrier of hand-over k + 1, which every reader of k reached
//  after its read — one barrier per hand-over instead of two)
__device__ __forceinline__ int to_ctrl(const ToState& S, int& cseq) {
    __syncthreads();
    const int c = S.ctrl[cseq & 1];
    cseq++;
    return c;
}

// lane 0 books the time since the last mark as algebra, runs the evaluation, books it as evaluation
#ifdef TO_PROFILE
#define TO_TE S.t_e
#else
#define TO_TE nullptr
#endif
#define TO_TIMED_EVAL() do { if (tid == 0) { const long long t_ = wall_clock64(); S.t_alg += t_ - S.t_mark; S.t_mark = t_; } \
        to_eval<HALF>(ev, s_a, s_b, s_tile, s_red, s_part, g, A.G, A.split_min, xch, tick, seq, A.epoch, A.late, hyp > 0 ? A.early_flag : nullptr, &s_abort, TO_TE); \
        if (tid == 0) { const long long t_ = wall_clock64(); S.t_eval += t_ - S.t_mark; S.t_mark = t_; } } while (0)

enum { TO_CONTINUE = 0, TO_FAIL = 1, TO_REPEAT_SAT = 2, TO_ITERATE = 3, TO_LEVEL_DONE = 4 };

template <bool HALF>
__global__ __launch_bounds__(TO_THREADS) void k_tracker_optimize(TrkOptArgs A) {
    __shared__ float s_a[TO_WAVES][64][TO_LD], s_b[TO_WAVES][64][TO_LD];
    __shared__ float s_tile[TO_WAVES][256];
    __shared__ float s_red[64];
    __shared__ float s_part[TO_PARTS * 64];
    __shared__ ToEval ev;
    __shared__ ToState S;
    __shared__ int s_abort;                                  // raised by an exchange that carried "give up" (early exit)
    // the trial log and the level passes collect HERE and leave with the result at the end.  Written where they happen they were two byte stores per
    // trial into the result — for the first workgroup of a hypothesis that is MAPPED HOST memory — right in front of a workgroup barrier, whose release
    // waits for every outstanding store: a PCIe acknowledgement (≈ 1 us) on the critical path of every Levenberg-Marquardt trial, and every other
    // workgroup of the hypothesis waits for this one at the next exchange
    __shared__ unsigned char s_step_level[CMLHIP_TRACKER_MAX_STEPS], s_step_accept[CMLHIP_TRACKER_MAX_STEPS];
    __shared__ int s_pass_level[8];
    __shared__ double s_pass_rmse[8];
    __shared__ double s_wA[64], s_wD[64];                          // scratchpads of the 8 x 8 solve (a dynamically indexed local array would live in scratch memory)
    const int tid = threadIdx.x, hyp = blockIdx.x / A.G, g = blockIdx.x % A.G;
    cmlhip_tracker_opt_result* out = (g == 0 && A.out_host) ? A.out_host + hyp : A.out + blockIdx.x;    // (each workgroup of the hypothesis writes its own copy: they are identical; the first one's goes straight to the host)
    float* xch = A.xch + (size_t)hyp * 2 * 2 * TO_PARTS * 64;     // (8-byte words, [2 parities][TO_PARTS][64]: see to_exchange)
    int* tick = A.tick + (size_t)hyp * A.G;
    int seq = 0, cseq = 0;
    const int maxIterations[5] = {10, 20, 50, 50, 50};                               // TR.cpp:23
    const int maxLevel = min(A.levels - 1, 4);
    if (tid == 0) {
        to_store(SE3::fromRt(A.hyp[hyp].R, A.hyp[hyp].t), S.cur_q, S.cur_t);
        S.a = A.init_a; S.b = A.init_b;
        for (int l = 0; l < 5; l++) { S.E[l] = S.E_new[l] = 0; S.nT[l] = S.nS[l] = S.nR[l] = S.nT_new[l] = S.nS_new[l] = S.nR_new[l] = 0; S.levelCutoffRepeat[l] = 0; S.iterations[l] = 0; }
        for (int k = 0; k < 3; k++) S.flow[k] = S.flow_new[k] = 0;
        S.n_steps = 0; S.n_pass = 0; S.haveRepeated = 0; S.ctrl[0] = S.ctrl[1] = TO_CONTINUE; s_abort = 0;
        S.t_eval = 0; S.t_alg = 0; S.t_mark = wall_clock64();
#ifdef TO_PROFILE
        S.t_ldlt = S.t_pose = S.t_fin = 0; S.t_p[0] = S.t_p[1] = S.t_p[2] = S.t_p[3] = 0; S.t_e[0] = S.t_e[1] = S.t_e[2] = S.t_e[3] = 0;
#endif
    }
    __syncthreads();
    bool failed = false;
    for (int level = maxLevel; level >= 0 && !failed; level--) {
        // ---- initial evaluation of the level (+ the saturation repeat, TR.cpp:61-81)
        if (tid == 0) { S.levelCutoffRepeat[level] = 1; to_prepare(A, ev, level, to_pose(S.cur_q, S.cur_t), S.a, S.b, 1.0); }
        __syncthreads();
        while (true) {
            TO_TIMED_EVAL();
            if (tid < 64) to_finish(A, s_red, S.E[level], S.nT[level], S.nS[level], S.nR[level], S.flow, S.H, S.bv);
            if (tid == 0) {
                int c = TO_ITERATE;
                if (s_abort) c = TO_FAIL;                                                                    // early exit: another hypothesis ended the search
                else if (S.nT[level] < 20) c = TO_FAIL;                                                           // :65-69
                else if ((S.nS[level] / (double)S.nT[level]) > 0.6 && S.levelCutoffRepeat[level] < 50) {     // :71-75
                    S.levelCutoffRepeat[level] *= 2;
                    to_prepare(A, ev, level, to_pose(S.cur_q, S.cur_t), S.a, S.b, S.levelCutoffRepeat[level], true);
                    c = TO_REPEAT_SAT;
                } else if (S.nT[level] - S.nS[level] < 10) c = TO_FAIL;                                      // :77-81
                S.ctrl[cseq & 1] = c; S.lambda = 0.01;
            }
            const int c0 = to_ctrl(S, cseq);
            if (c0 != TO_REPEAT_SAT) { if (c0 == TO_FAIL) failed = true; break; }
        }
        if (failed) break;
        // ---- Levenberg-Marquardt trials, TR.cpp:91-181
        for (int iteration = 0; iteration < maxIterations[level]; iteration++) {
            if (tid < 64) {                                                                                 // wave 0: the damped solve, :96-119
                // which system: 8x8 (a and b), top-left 7x7 (a only), column/row 6 <- 7 stitched 7x7 (b only: HlStitch, bStitch[6] = b[7]),
                // top-left 6x6 (neither); the increment's lanes the variant does not solve stay 0
                const int nsolve = (A.opt_a && A.opt_b) ? 8 : ((A.opt_a || A.opt_b) ? 7 : 6);
                const int map6 = (!A.opt_a && A.opt_b) ? 7 : 6;
                double xs[8];
#ifdef TO_PROFILE
                const long long tp0 = wall_clock64();
#endif
                bool sorted_ok = false;
                bool ok = to_ldlt_solve_wave_sorted(S.H, S.bv, S.lambda, nsolve, map6, xs, s_wA, sorted_ok);
                if (!sorted_ok) ok = to_ldlt_solve_wave(S.H, S.bv, S.lambda, nsolve, map6, xs);      // (wave-uniform: ties / NaN on the diagonal)
#ifdef TO_PROFILE
                const long long tp1 = wall_clock64();
                if (tid == 0) S.t_ldlt += tp1 - tp0;
#endif
              if (tid < 4) {                                                                                // lanes 0-3 run the same chain (identical stores); see to_se3_exp_lanes
                S.iterations[level] = iteration + 1;
                // the increment in registers (statically indexed: the loops below are unrolled) — the LDS scratchpads it used to live on cost a
                // dependent round trip per read-modify-write
                double inc[8], incS[8];
#pragma unroll
                for (int i = 0; i < 6; i++) inc[i] = xs[i];
                inc[6] = A.opt_a ? xs[6] : 0.0;                                                             // the lanes the variant does not solve stay 0
                inc[7] = (A.opt_a && A.opt_b) ? xs[7] : ((!A.opt_a && A.opt_b) ? xs[6] : 0.0);
                if (!ok) S.ctrl[cseq & 1] = TO_FAIL;                                                                  // :121-138
                else {
                    double extrapFac = 1;
                    if (S.lambda < 0.001) extrapFac = sqrt(sqrt(0.001 / S.lambda));                         // :140-142
                    double nrm = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) { inc[i] *= extrapFac; incS[i] = inc[i]; nrm += inc[i] * inc[i]; }
#pragma unroll
                    for (int i = 0; i < 3; i++) { incS[i] *= (double)A.scale_rot; incS[3 + i] *= (double)A.scale_trans; }   // the literal lane / scale pairing, :144-148
                    incS[6] *= (double)A.scale_a; incS[7] *= (double)A.scale_b;
#ifdef TO_PROFILE
                    const long long q0 = wall_clock64(); S.t_p[0] += q0 - tp1;
                    double nrm_rt;
                    const SE3 ex_ = to_se3_exp_lanes(incS, nrm, nrm_rt, tid);
                    const long long q1 = wall_clock64(); S.t_p[1] += q1 - q0;
                    const SE3 nw = ex_ * to_pose(S.cur_q, S.cur_t);
#else
                    double nrm_rt;
                    const SE3 nw = to_se3_exp_lanes(incS, nrm, nrm_rt, tid) * to_pose(S.cur_q, S.cur_t);   // :155-157
#endif
                    to_store(nw, S.nw_q, S.nw_t);
                    S.na = S.a + incS[6]; S.nb = S.b + incS[7];                                             // :159
                    S.Hn[0] = nrm_rt;                                                                       // |increment| = sqrt(nrm), parked for the exit test below
#ifdef TO_PROFILE
                    const long long q2 = wall_clock64(); S.t_p[2] += q2 - q1;
#endif
                    to_prepare(A, ev, level, nw, S.na, S.nb, S.levelCutoffRepeat[level], true);
#ifdef TO_PROFILE
                    S.t_p[3] += wall_clock64() - q2;
#endif
                    S.ctrl[cseq & 1] = TO_ITERATE;
                }
#ifdef TO_PROFILE
                S.t_pose += wall_clock64() - tp1;
#endif
              }
            }
            if (to_ctrl(S, cseq) == TO_FAIL) { failed = true; break; }
            TO_TIMED_EVAL();
#ifdef TO_PROFILE
            const long long tf0 = wall_clock64();
#endif
            if (tid < 64) {
                const double incnorm = S.Hn[0];                                                             // (lane 0's own note, read before its slot is rewritten)
                to_finish(A, s_red, S.E_new[level], S.nT_new[level], S.nS_new[level], S.nR_new[level], S.flow_new, S.Hn, S.bn);
                const bool accept = ((double)s_red[45] / (double)(int)s_red[49]) < (S.E[level] / (double)S.nT[level]);   // E_new / n_new < E / n, :163
                if (accept) {                                                                               // every lane moves the entries it wrote
                    S.H[tid] = S.Hn[tid];
                    if (tid < 8) S.bv[tid] = S.bn[tid];
                    // oldResidual = newResidual (whole struct, :167) and the trial's pose: a lane per entry (lane 0 alone paid ~35 loads and as many
                    // stores, serialised in groups); nothing reads these fields before the hand-over below
                    if (tid < 5) { S.E[tid] = S.E_new[tid]; S.nT[tid] = S.nT_new[tid]; S.nS[tid] = S.nS_new[tid]; S.nR[tid] = S.nR_new[tid]; }
                    if (tid < 4) S.cur_q[tid] = S.nw_q[tid];
                    if (tid < 3) { S.flow[tid] = S.flow_new[tid]; S.cur_t[tid] = S.nw_t[tid]; }
                }
                if (tid == 0) {
                    if (S.n_steps < CMLHIP_TRACKER_MAX_STEPS) { s_step_level[S.n_steps] = (unsigned char)level; s_step_accept[S.n_steps] = accept ? 1 : 0; }
                    S.n_steps++;
                    if (accept) {
                        S.a = S.na; S.b = S.nb;                                                             // (the arrays: a lane per entry, above)
                        S.lambda *= 0.5;
                    } else {
                        S.lambda *= 4;
                    }
                    S.ctrl[cseq & 1] = s_abort ? TO_FAIL : ((incnorm < 1e-3) ? TO_LEVEL_DONE : TO_ITERATE);           // :176-179 (TO_FAIL: early exit)
                }
            }
#ifdef TO_PROFILE
            if (tid == 0) S.t_fin += wall_clock64() - tf0;
#endif
            { const int c1 = to_ctrl(S, cseq); if (c1 == TO_FAIL) { failed = true; break; } if (c1 == TO_LEVEL_DONE) break; }
        }
        if (failed) break;
        if (tid == 0) {
            // the rmse of the pass: what TR.cpp:183-189 compares with 1.5 x the previous correct try's (applied by the caller)
            if (S.n_pass < 8) { s_pass_level[S.n_pass] = level; s_pass_rmse[S.n_pass] = S.E[level] / (double)S.nT[level]; }
            S.n_pass++;
            S.ctrl[cseq & 1] = (S.levelCutoffRepeat[level] > 1 && !S.haveRepeated) ? 1 : 0;                 // :192-195
            if (S.ctrl[cseq & 1]) S.haveRepeated = 1;
        }
        if (to_ctrl(S, cseq)) level++;
    }
    if (tid == 0) {
        double R[9];
        to_pose(S.cur_q, S.cur_t).matrix(R);
        for (int i = 0; i < 9; i++) out->R[i] = R[i];
        for (int i = 0; i < 3; i++) out->t[i] = S.cur_t[i];
        out->a = S.a; out->b = S.b;
        if (hyp == 0 && g == 0 && A.pose0) {
            for (int i = 0; i < 9; i++) A.pose0[i] = R[i];
            for (int i = 0; i < 3; i++) A.pose0[9 + i] = S.cur_t[i];
            A.pose0[12] = S.a; A.pose0[13] = S.b;
        }
        for (int l = 0; l < 5; l++) {
            out->E[l] = S.E[l]; out->numTermsInE[l] = S.nT[l]; out->numSaturated[l] = S.nS[l]; out->numRobust[l] = S.nR[l];
            out->levelCutoffRepeat[l] = S.levelCutoffRepeat[l]; out->iterations[l] = S.iterations[l];
        }
        for (int k = 0; k < 3; k++) out->flow[k] = S.flow[k];
        for (int k = 0; k < S.n_steps && k < CMLHIP_TRACKER_MAX_STEPS; k++) { out->step_level[k] = s_step_level[k]; out->step_accept[k] = s_step_accept[k]; }
        for (int k = 0; k < S.n_pass && k < 8; k++) { out->pass_level[k] = s_pass_level[k]; out->pass_rmse[k] = s_pass_rmse[k]; }
        out->n_steps = s_abort ? -1 : S.n_steps; out->n_pass = S.n_pass < 8 ? S.n_pass : 8;      // -1: given up (early exit), nothing else of the result is meaningful
        out->eval_us = 0.01 * (double)S.t_eval; out->algebra_us = 0.01 * (double)(S.t_alg + (wall_clock64() - S.t_mark));
        for (int k = 0; k < 6; k++) out->covariance[k] = 999999;
#ifdef TO_PROFILE
        out->pass_rmse[7] = 0.01 * (double)S.t_ldlt; out->pass_rmse[6] = 0.01 * (double)S.t_pose; out->pass_rmse[5] = 0.01 * (double)S.t_fin;
#endif
        out->relAff[0] = out->relAff[1] = 0;
        if (failed) {
            out->isCorrect = 0; out->tooManySaturated = 1;
        } else {
            const double relA = exp(S.a - A.ref_a) * A.new_t / A.ref_t, relB = S.b - relA * A.ref_b;        // :203
            bool haveGoodLight = true;
            if (A.opt_a) { if (fabs(S.a) > 1.2) haveGoodLight = false; }
            else if (fabs(logf((float)relA)) > 1.5) haveGoodLight = false;
            if (A.opt_b) { if (fabs(S.b) > 200) haveGoodLight = false; }
            else if (fabs((float)relB) > 200) haveGoodLight = false;
            bool haveGoodPoints = true;
            if ((double)S.nS[0] / (double)S.nT[0] > A.sat_th) haveGoodPoints = false;                       // :231-235
            out->isCorrect = haveGoodLight ? 1 : 0;
            out->tooManySaturated = haveGoodPoints ? 1 : 0;                                                 // sic, :240
            // early exit: the first try ends the search when it is adopted and its E/n of level 0 is below the bar (DSOTracker.h:288-309 on an empty history)
            if (A.early_flag && hyp == 0 && g == 0 && haveGoodLight && S.nT[0] > 0) {      // (adoption of the first try: isCorrect and a finite E/n — the history's tooManySaturated starts true)
                const double rm = (double)S.E[0] / (double)S.nT[0];
                if (isfinite(rm) && rm < (double)A.early_rmse) __hip_atomic_store(A.early_flag, A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            out->relAff[0] = relA; out->relAff[1] = relB;
        }
#ifdef TO_PROFILE
        out->relAff[0] = 0.01 * (double)S.t_p[0]; out->relAff[1] = 0.01 * (double)S.t_p[1]; out->flow[0] = (float)(0.01 * (double)S.t_p[2]); out->flow[1] = (float)(0.01 * (double)S.t_p[3]);
#endif
    }
    if (A.tr_pairs && hyp == 0 && g == 0) {                                                                 // (uniform per workgroup)
        __threadfence_block();
        __syncthreads();                                                                                    // lane 0's pose0 stores, visible to the workgroup
        if (tid < A.req.n_hosts) {
            cmlhip_trace_pair P;
            tp_pair(A.pose0, A.req.ref, A.req.K, A.req.hosts[tid], P);
            A.tr_pairs[tid] = P;
        }
    }
    if (!failed && tid < 64) {                                                                              // :243, wave 0 (S.H is lane 0's no more: nothing writes it from here on)
        const double hi = to_inverse8_wave(S.H, tid);
        if ((tid >> 3) == (tid & 7) && (tid >> 3) < 6) out->covariance[tid >> 3] = hi;
    }
#ifdef TO_PROFILE
    __syncthreads();
    if (tid == 0) for (int k = 0; k < 4; k++) out->covariance[k] = 0.01 * (double)S.t_e[k];      // (profile build: the evaluation's phases ride in the covariance slots)
#endif
}

extern "C" int cmlhip_tracker_set_early_exit(cmlhip_ctx* c, double rmse_bar) {
    if (!c || !(rmse_bar == rmse_bar)) return CMLHIP_ERR_INVALID;
    c->trk_early_rmse = rmse_bar > 0.0 ? rmse_bar : 0.0;
    return CMLHIP_OK;
}

// ---- completion tickets: a one-thread kernel behind the work enqueued so far stores the ticket's number in mapped, coherent host memory and the
// host spins on that word — the wait of a frame's whole chain (tracker batch, speculative trace) without the stream-synchronise path
// (CMLHIP_NO_POLL=1 restores hipStreamSynchronize for an A/B).  The word only ever grows; a wait that sees nothing for 2 s falls back to the stream.
__global__ void k_done_ticket(volatile unsigned* word, unsigned ticket) {
    __threadfence_system();
    *word = ticket;
}
int cml_done_enqueue(cmlhip_ctx* c) {
    if (!c->done_word) {
        CML_CHECK(c, hipHostMalloc(&c->done_word, 64, hipHostMallocMapped | hipHostMallocCoherent));
        *static_cast<volatile unsigned*>(c->done_word) = 0;
        CML_CHECK(c, hipHostGetDevicePointer(&c->done_word_dev, c->done_word, 0));
    }
    c->done_ticket += 1;
    k_done_ticket<<<1, 1, 0, c->stream>>>(static_cast<volatile unsigned*>(c->done_word_dev), c->done_ticket);
    CML_CHECK(c, hipGetLastError());
    c->done_pending = true;
    return CMLHIP_OK;
}
// a ticket written by a kernel of the caller's own (its last act, behind a system-scope fence): the number to write and where
int cml_done_embed(cmlhip_ctx* c, unsigned* ticket, volatile unsigned** word_dev) {
    if (!c->done_word) {
        CML_CHECK(c, hipHostMalloc(&c->done_word, 64, hipHostMallocMapped | hipHostMallocCoherent));
        *static_cast<volatile unsigned*>(c->done_word) = 0;
        CML_CHECK(c, hipHostGetDevicePointer(&c->done_word_dev, c->done_word, 0));
    }
    c->done_ticket += 1;
    *ticket = c->done_ticket; *word_dev = static_cast<volatile unsigned*>(c->done_word_dev);
    c->done_pending = true;
    return CMLHIP_OK;
}
int cml_done_wait(cmlhip_ctx* c) {
    static const char* e_np = getenv("CMLHIP_NO_POLL");
    if (!c->done_pending || e_np) {
        CML_CHECK(c, hipStreamSynchronize(c->stream));
        c->done_pending = false;
        return CMLHIP_OK;
    }
    volatile unsigned* w = static_cast<volatile unsigned*>(c->done_word);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while ((int)(*w - c->done_ticket) < 0) {
        if ((++spins & 0x3ffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {      // (a failed launch never writes: the stream reports it)
            CML_CHECK(c, hipStreamSynchronize(c->stream));
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    c->done_pending = false;
    return CMLHIP_OK;
}

extern "C" int cmlhip_tracker_optimize_batch_async(cmlhip_ctx* c, uint64_t image_id, int levels, const double K0[4], const double ref_exposure[3],
                                                   const double init_exposure[3], const cmlhip_tracker_params* prm, int optimize_a, int optimize_b,
                                                   double saturated_ratio_th, int n_hyp, const cmlhip_tracker_hypothesis* hyp) { CML_DEV(c);
    if (!c || !K0 || !ref_exposure || !init_exposure || !prm || n_hyp < 0 || (n_hyp > 0 && !hyp) || levels < 1) return CMLHIP_ERR_INVALID;
    if (c->trk_pending_n) { c->err = "cmlhip_tracker_optimize_batch_async: the previous batch has not been waited for (cmlhip_tracker_optimize_wait)"; return CMLHIP_ERR_INVALID; }
    if (n_hyp == 0) return CMLHIP_OK;
    const Pyramid* py = cml_find_pyr(c, image_id);
    CML_REQUIRE(c, py && py->levels >= 1, CMLHIP_ERR_NOT_FOUND, "tracker image not in the pyramid cache");
    TrkOptArgs A;
    memset(&A, 0, sizeof A);
    A.levels = std::min(levels, std::min(py->levels, 5));
    for (int l = 0; l < A.levels; l++) {
        CML_REQUIRE(c, py->lv[l].grad, CMLHIP_ERR_NOT_FOUND, "tracker level not in the pyramid cache");
        A.img[l] = py->lv[l].grad; A.w[l] = py->lv[l].w; A.h[l] = py->lv[l].h;
        A.uvic[l] = c->trk_ref[l].as<float>(); A.n[l] = c->trk_n[l];
    }
    for (int k = 0; k < 4; k++) A.K0[k] = K0[k];
    A.ref_a = ref_exposure[0]; A.ref_b = ref_exposure[1]; A.ref_t = ref_exposure[2];
    A.init_a = init_exposure[0]; A.init_b = init_exposure[1]; A.new_t = init_exposure[2];
    A.huber = prm->huber; A.cutoff_base = prm->cutoff_base; A.scale_rot = prm->scale_rot; A.scale_trans = prm->scale_trans;
    A.scale_a = prm->scale_a; A.scale_b = prm->scale_b;
    A.opt_a = optimize_a; A.opt_b = optimize_b; A.sat_th = saturated_ratio_th; A.n_hyp = n_hyp;
    int rc;
    // workgroups per hypothesis: as many as keep every workgroup of the launch resident at once (they wait for one another), at most 8.
    // The capacity is asked of THIS device (CUs x occupancy of the kernel: a partitioned gfx950 — CPX / DPX — reports fewer CUs), not assumed.
    const bool half = c->lim.texel_format == CMLHIP_TEXEL_F16;
    if (c->trk_capacity[half] == 0) {
        hipDeviceProp_t prop;
        CML_CHECK(c, hipGetDeviceProperties(&prop, c->lim.device_id));
        int per_cu = 0;
        if (half) CML_CHECK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tracker_optimize<true>, TO_THREADS, 0));
        else CML_CHECK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tracker_optimize<false>, TO_THREADS, 0));
        c->trk_capacity[half] = std::max(1, prop.multiProcessorCount * std::max(per_cu, 1));
    }
    const int capacity = std::max(1, c->trk_capacity[half] / std::max(1, c->device_share));      // (cmlhip_set_device_share: other contexts launch beside this one)
    static const char* e_g = getenv("CMLHIP_TRACKER_G");                  // development: force G (still clamped to what fits)
    int G = e_g ? atoi(e_g) : std::min(TO_PARTS, capacity / n_hyp);
    if (G < 1) G = 1;
    if ((long long)n_hyp * G > capacity) G = std::max(1, capacity / n_hyp);          // G = 1: no workgroup waits for another, any launch size is fine
    G = std::min(G, TO_PARTS);
    G = G >= 8 ? 8 : (G >= 4 ? 4 : (G >= 2 ? 2 : 1));                                // a divisor of TO_PARTS: a workgroup owns whole parts
    static const char* e_sp = getenv("CMLHIP_TRACKER_SPLIT");         // development: the level size above which a level is evaluated in parts
    A.G = G; A.split_min = e_sp ? atoi(e_sp) : TO_THREADS;          // (measured, one hypothesis, 15 trials: 0 / 512 / 1024 / 2560 / 5120 -> 0.225 / 0.224 / 0.233 / 0.280 / 0.313 ms: a level of more than one chunk is worth its exchange)
    if ((rc = cml_ensure(c, c->trk_opt_out, sizeof(cmlhip_tracker_opt_result) * (size_t)n_hyp * G))) return rc;
    const size_t xch_bytes = sizeof(unsigned long long) * 2 * 64 * (size_t)TO_PARTS * n_hyp, tick_bytes = ((sizeof(int) * (size_t)G * n_hyp + 63) / 64) * 64;
    if ((rc = cml_ensure(c, c->trk_xch, xch_bytes + tick_bytes + 64))) return rc;
    // hypotheses in, results out through ONE mapped, coherent host block: the kernel reads the 96 bytes of its hypothesis and the first
    // workgroup of each hypothesis writes its result there — no staged upload before the launch and no copy back behind it (each was a
    // copy command of its own on the stream: 321 -> 299 us per call for one hypothesis, 421 -> 389 for fifty)
    const size_t hyp_bytes = ((sizeof(cmlhip_tracker_hypothesis) * (size_t)n_hyp + 255) / 256) * 256, res_bytes = ((sizeof(cmlhip_tracker_opt_result) * (size_t)n_hyp + 255) / 256) * 256;
    if (c->trk_opt_host_bytes < hyp_bytes + res_bytes + 256) {
        if (c->trk_opt_host) { CML_CHECK(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->trk_opt_host); c->trk_opt_host = nullptr; c->trk_opt_host_bytes = 0; }
        const size_t want = std::max<size_t>(2 * (hyp_bytes + res_bytes) + 256, 64 * 1024);
        CML_CHECK(c, hipHostMalloc(&c->trk_opt_host, want, hipHostMallocMapped | hipHostMallocCoherent));
        c->trk_opt_host_bytes = want;
        CML_CHECK(c, hipHostGetDevicePointer(&c->trk_opt_host_dev, c->trk_opt_host, 0));
    }
    char* const hb = static_cast<char*>(c->trk_opt_host);
    memcpy(hb, hyp, sizeof(cmlhip_tracker_hypothesis) * (size_t)n_hyp);
    memset(hb + hyp_bytes, 0, res_bytes + sizeof(int));
    void* const dptr = c->trk_opt_host_dev;
    A.hyp = reinterpret_cast<const cmlhip_tracker_hypothesis*>(dptr);
    A.out = c->trk_opt_out.as<cmlhip_tracker_opt_result>();
    A.out_host = reinterpret_cast<cmlhip_tracker_opt_result*>(static_cast<char*>(dptr) + hyp_bytes);
    A.xch = c->trk_xch.as<float>(); A.tick = reinterpret_cast<int*>(c->trk_xch.as<char>() + xch_bytes);
    A.early_rmse = (float)c->trk_early_rmse;
    // the early-exit word has a buffer of its own (a fixed address that only ever holds launch numbers: inside trk_xch its offset moved with (n_hyp, G)
    // and could fall on a stale exchange word of an earlier, larger launch).  It is raised by storing THIS launch's number, so it needs no clearing per
    // launch — once per allocation and when the 16-bit launch number wraps, with the exchange buffer below
    A.early_flag = nullptr;
    if (c->trk_early_rmse > 0.0 && n_hyp > 1) {
        const unsigned gen0 = c->trk_early.gen;
        if ((rc = cml_ensure(c, c->trk_early, 64))) return rc;
        if (c->trk_early.gen != gen0 || c->trk_epoch >= 0xfffe) CML_CHECK(c, hipMemsetAsync(c->trk_early.p, 0, 64, c->stream));
        A.early_flag = c->trk_early.as<int>();
    }
    if ((rc = cml_ensure(c, c->trk_pose0, 128))) return rc;
    A.pose0 = c->trk_pose0.as<double>();
    A.tr_pairs = nullptr;
    if (c->tr_req_valid) {                                 // one shot: the request is this launch's
        if ((rc = cml_ensure(c, c->tr_pairs, sizeof(cmlhip_trace_pair) * CMLHIP_MAX_FRAMES))) return rc;
        A.tr_pairs = c->tr_pairs.as<cmlhip_trace_pair>(); A.req = c->tr_req;
        c->tr_req_valid = false; c->tr_req_consumed = true;
    } else { c->tr_req_consumed = false; A.req.n_hosts = 0; }
    A.late = reinterpret_cast<int*>(static_cast<char*>(dptr) + hyp_bytes + res_bytes);
    // (no per-call clearing: the words carry the launch number; a fresh or moved buffer is cleared once)
    // cleared once per ALLOCATION (DevBuf::gen, not the address: a free + malloc may hand the address back) and whenever the 16-bit launch
    // number wraps, so that a stale word can never carry a matching tag
    if (c->trk_xch.gen != c->trk_xch_gen || c->trk_epoch >= 0xfffe) {
        CML_CHECK(c, hipMemsetAsync(c->trk_xch.p, 0, c->trk_xch.bytes, c->stream)); c->trk_xch_gen = c->trk_xch.gen; c->trk_epoch = 0;
        // the launch number restarts: the early-exit word (which holds launch numbers) restarts with it, armed launch or not — a word raised
        // at number E before the restart must not be read as "hypothesis 0 passed" when the restarted counter reaches E again
        if (c->trk_early.p) CML_CHECK(c, hipMemsetAsync(c->trk_early.p, 0, 64, c->stream));
    }
    c->trk_epoch += 1;                                     // 1 .. 0xfffe: never the zero of a cleared buffer
    A.epoch = c->trk_epoch;
    std::atomic_thread_fence(std::memory_order_release);
    if (half) CML_LAUNCH_EV(c, k_tracker_optimize<true>, n_hyp * G, TO_THREADS, 0, A);
    else CML_LAUNCH_EV(c, k_tracker_optimize<false>, n_hyp * G, TO_THREADS, 0, A);
    CML_CHECK(c, hipGetLastError());
    c->trk_pending_n = n_hyp; c->trk_pending_res_off = hyp_bytes; c->trk_pending_late_off = hyp_bytes + res_bytes;
    c->done_pending = false;                                 // (a ticket enqueued before this launch does not cover it)
    return CMLHIP_OK;
}

// the device address of the pending batch's result `i` (mapped host memory: written by the first workgroup of the hypothesis) — for kernels enqueued
// behind the batch on the same stream (tracer.hip: the speculative trace reads the first hypothesis' pose from it)
const cmlhip_tracker_opt_result* cml_tracker_pending_result_dev(cmlhip_ctx* c, int i) {
    if (!c->trk_pending_n || i < 0 || i >= c->trk_pending_n) return nullptr;
    return reinterpret_cast<const cmlhip_tracker_opt_result*>(static_cast<char*>(c->trk_opt_host_dev) + c->trk_pending_res_off) + i;
}

extern "C" int cmlhip_tracker_optimize_wait(cmlhip_ctx* c, cmlhip_tracker_opt_result* out) { CML_DEV(c);
    if (!c) return CMLHIP_ERR_INVALID;
    const int n_hyp = c->trk_pending_n;
    if (n_hyp == 0) return CMLHIP_OK;
    if (!out) return CMLHIP_ERR_INVALID;
    c->trk_pending_n = 0;
    int rc;
    if (!c->done_pending && (rc = cml_done_enqueue(c))) return rc;      // nothing was enqueued behind the batch with a ticket of its own: one behind the batch
    if ((rc = cml_done_wait(c))) return rc;
    const char* const hb = static_cast<const char*>(c->trk_opt_host);
    memcpy(out, hb + c->trk_pending_res_off, sizeof(cmlhip_tracker_opt_result) * (size_t)n_hyp);
    if (*reinterpret_cast<const volatile int*>(hb + c->trk_pending_late_off))
        CML_REQUIRE(c, false, CMLHIP_ERR_TIMEOUT, "tracker optimize: a workgroup gave up waiting for the partial sums of its hypothesis (the launch was not co-resident); results are void");
    return CMLHIP_OK;
}

extern "C" int cmlhip_tracker_optimize_batch(cmlhip_ctx* c, uint64_t image_id, int levels, const double K0[4], const double ref_exposure[3],
                                             const double init_exposure[3], const cmlhip_tracker_params* prm, int optimize_a, int optimize_b,
                                             double saturated_ratio_th, int n_hyp, const cmlhip_tracker_hypothesis* hyp,
                                             cmlhip_tracker_opt_result* out) {
    if (n_hyp > 0 && !out) return CMLHIP_ERR_INVALID;
    const int rc = cmlhip_tracker_optimize_batch_async(c, image_id, levels, K0, ref_exposure, init_exposure, prm, optimize_a, optimize_b, saturated_ratio_th, n_hyp, hyp);
    if (rc || n_hyp == 0) return rc;
    return cmlhip_tracker_optimize_wait(c, out);
}
