// ba_frames.h — the frame half of doStepFromBackup on the device (one workgroup of the back-substitution launch).
// Replaces, for the device-resident iterations, DSOFrame::setStep / doStepFromBackup / setState (DSOFrame.h:82-84,
// 110-124,205-214), the N^2 DSOFramePrecomputed (DSOFrame.h:248-291) and the frame part of computeDelta
// (BA.cpp:1103-1194).  The SE(3) algebra is the host mirror's (host/se3.h, Sophus 1.1.0 semantics), compiled for gfx950.
#pragma once
#include "cmlhip_internal.h"
#pragma clang fp contract(off)          // the host mirror runs the same algebra without fused multiply-adds
#include "../host/se3.h"

struct FrameStepArgs {
    cmlhip_ba_frame_state* fs;     // N
    cmlhip_ba_pair* pairs;         // N*N, host*N + target (R0, t0 stay: evaluation point)
    double* pre_w2c;               // N x 7 (q, t) of PRE_worldToCam, for readback
    const double* adH; const double* adT;
    float* adHTd;                  // N*N x 8, index host + target*N
    double* dprior;                // 8N: state - prior_zero
    double sc[4];                  // scale translation / rotation / a / b
    int N, on;
    int adhtd_done;                // wide windows: adHTd of the stepped state was already written by k_ba_xad
    float* frame_sums;             // optional: sumA sumB sumT sumR of doStepFromBackup (BA.cpp:957-972) for the convergence test
    // marginalisation prior inside the resident loop (cmlhip_ba_set_resident_prior): after the step, the right-hand side the NEXT
    // iteration's system takes, bM_top = mMarginalizedB + mMarginalizedHessian * getFramesDelta() (BA.cpp:1389-1401), n = 8N+4
    const double* HM; const double* bM_raw; double* bM_top; int n;
};

// bM_top[r] = bM_raw[r] + sum_j HM[r][j] d[j], d = (0,0,0,0, delta of frame 0, delta of frame 1, ...), summed in column order like the
// host mirror's solveSystem: one row per thread
__device__ __forceinline__ void frame_prior_rhs(const FrameStepArgs& F, const double (*delta)[8]) {
    for (int r = threadIdx.x; r < F.n; r += blockDim.x) {
        const double* row = F.HM + (size_t)r * F.n;
        double s = F.bM_raw[r];
        for (int j = 4; j < F.n; j++) s += row[j] * delta[(j - 4) >> 3][(j - 4) & 7];
        F.bM_top[r] = s;
    }
}

// What the frame step reads that does NOT depend on x — requested by frame_step_prefetch AHEAD of the wait for x when the block rides in the
// solve launch (in-kernel stamps, round 3: this block, not the point blocks, ended the launch — 6.8 against 4.0 us behind the solve
// workgroup's last stamp: the frame states were a trip behind x, and each of 64 lanes then pulled the 128 adjoint entries of its pair,
// 8192 scattered 8-byte loads, for the 128 multiply-adds of computeDelta).  With up to two (pair, column) entries of adHTd per thread
// the adjoint columns are 16 values per entry, held in registers across the wait; the sums keep their order (same bits).
struct FrameStepPre {
    double ahc[2][8], atc[2][8];
    double s_old[8], s_zero[8], p_zero[8], evq[4], evt[3], ab_exposure;
    bool fix_pose, dist;
};
__device__ __forceinline__ void frame_step_prefetch(const FrameStepArgs& F, FrameStepPre& P) {
    const int tid = threadIdx.x, N = F.N, ne = N * N * 8, bd = blockDim.x;
    P.dist = !F.adhtd_done && ne <= 2 * bd;
    if (P.dist) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = min(tid + u * bd, ne - 1);
            const int j = e & 7, q = e >> 3, h = q / N, t = q % N, idx = h + t * N;
            const double* AH = F.adH + 64 * (size_t)idx; const double* AT = F.adT + 64 * (size_t)idx;
#pragma unroll
            for (int i = 0; i < 8; i++) { P.ahc[u][i] = AH[i * 8 + j]; P.atc[u][i] = AT[i * 8 + j]; }
        }
    }
    const cmlhip_ba_frame_state& S = F.fs[min(tid, N - 1)];
    // Everything the step reads, in ONE round trip: written field by field against the record in memory, every load waits behind the
    // store before it (same type, possibly aliased — the compiler must keep the order): eight dependent round trips.
#pragma unroll
    for (int k = 0; k < 8; k++) { P.s_old[k] = S.state[k]; P.s_zero[k] = S.state_zero[k]; P.p_zero[k] = S.prior_zero[k]; }
#pragma unroll
    for (int k = 0; k < 4; k++) P.evq[k] = S.eval_q[k];
#pragma unroll
    for (int k = 0; k < 3; k++) P.evt[k] = S.eval_t[k];
    P.fix_pose = S.fix_pose != 0;
    P.ab_exposure = S.ab_exposure;
}
// fsdbg: optional development stamps (x in LDS = entry, states stepped, barrier passed, pair records stored)
#define FS_STAMP(k) do { if (fsdbg && threadIdx.x == 0) fsdbg[k] = wall_clock64(); } while (0)
__device__ __forceinline__ void frame_step_block(const FrameStepArgs& F, const double* __restrict__ x, const FrameStepPre& P, long long* fsdbg = nullptr) {
    using cml_amd::SE3;
    FS_STAMP(0);
    using cml_amd::Exposure;
    __shared__ double s_w2c[CMLHIP_MAX_FRAMES][7], s_c2w[CMLHIP_MAX_FRAMES][7], s_aff[CMLHIP_MAX_FRAMES][3], s_delta[CMLHIP_MAX_FRAMES][8], s_step[CMLHIP_MAX_FRAMES][8];
    const int tid = threadIdx.x, N = F.N;
    const bool dist = P.dist;
    if (tid < N) {
        cmlhip_ba_frame_state& S = F.fs[tid];
        double step[8], st[8];
        const double* s_old = P.s_old; const double* s_zero = P.s_zero; const double* p_zero = P.p_zero; const double* evq = P.evq; const double* evt = P.evt;
        const bool fix_pose = P.fix_pose;
        const double ab_exposure = P.ab_exposure;
        bool fin = true;
#pragma unroll
        for (int k = 0; k < 8; k++) { step[k] = -x[4 + 8 * tid + k]; fin = fin && isfinite(step[k]); }     // BA.cpp:1433-1441
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (!fin) step[k] = 0.0;                                   // setStep, DSOFrame.h:205-214
            if (fix_pose && k < 6) step[k] = 0.0;                      // BA.cpp:957-960
            s_step[tid][k] = step[k];
            st[k] = s_old[k] + step[k];                                // state_backup == state: every step is accepted here
            s_delta[tid][k] = st[k] - s_zero[k];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { S.state[k] = st[k]; F.dprior[8 * tid + k] = st[k] - p_zero[k]; }
        const double ss[6] = {F.sc[0] * st[0], F.sc[0] * st[1], F.sc[0] * st[2], F.sc[1] * st[3], F.sc[1] * st[4], F.sc[1] * st[5]};
        SE3 ev;
#pragma unroll
        for (int k = 0; k < 4; k++) ev.q[k] = evq[k];
#pragma unroll
        for (int k = 0; k < 3; k++) ev.t[k] = evt[k];
        const SE3 W = SE3::exp(ss) * ev;                               // PRE_worldToCam, DSOFrame.h:119
        const SE3 Ci = W.inverse();
#pragma unroll
        for (int k = 0; k < 4; k++) { s_w2c[tid][k] = W.q[k]; s_c2w[tid][k] = Ci.q[k]; F.pre_w2c[7 * tid + k] = W.q[k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { s_w2c[tid][4 + k] = W.t[k]; s_c2w[tid][4 + k] = Ci.t[k]; F.pre_w2c[7 * tid + 4 + k] = W.t[k]; }
        s_aff[tid][0] = ab_exposure; s_aff[tid][1] = F.sc[2] * st[6]; s_aff[tid][2] = F.sc[3] * st[7];    // aff_g2l
        FS_STAMP(1);
    }
    __syncthreads();
    FS_STAMP(2);
    if (F.frame_sums && tid == 0) {                                    // fp32 sums in frame order, as the host loop forms them
        float sumA = 0, sumB = 0, sumT = 0, sumR = 0;
        for (int f = 0; f < N; f++) {
            const double* st = s_step[f];
            sumA += (float)(st[6] * st[6]);
            sumB += (float)(st[7] * st[7]);
            sumT += (float)(st[0] * st[0] + st[1] * st[1] + st[2] * st[2]);
            sumR += (float)(st[3] * st[3] + st[4] * st[4] + st[5] * st[5]);
        }
        F.frame_sums[0] = sumA; F.frame_sums[1] = sumB; F.frame_sums[2] = sumT; F.frame_sums[3] = sumR;
    }
    if (F.HM) frame_prior_rhs(F, s_delta);
    if (dist) {
        // computeDelta, BA.cpp:1120-1135, one (pair, column) entry per thread and pass from the adjoint columns requested ahead: the
        // sums of the per-pair form below, term for term
        const int ne = N * N * 8, bd = blockDim.x;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = tid + u * bd;
            if (e < ne) {
                const int j = e & 7, q = e >> 3, h = q / N, t = q % N, idx = h + t * N;
                double sH = 0, sT = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) { sH += s_delta[h][i] * P.ahc[u][i]; sT += s_delta[t][i] * P.atc[u][i]; }
                F.adHTd[8 * (size_t)idx + j] = (float)(sH + sT);
            }
        }
    }
    for (int q = tid; q < N * N; q += blockDim.x) {
        const int h = q / N, t = q % N;
        SE3 Wt, Ch;
#pragma unroll
        for (int k = 0; k < 4; k++) { Wt.q[k] = s_w2c[t][k]; Ch.q[k] = s_c2w[h][k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { Wt.t[k] = s_w2c[t][4 + k]; Ch.t[k] = s_c2w[h][4 + k]; }
        // computeDelta, BA.cpp:1120-1135 — FIRST, with all 128 adjoint entries of the pair requested together and the eight results held
        // back: written per column between the stores of the pair record, every column's loads waited behind the stores before them
        // (a chain of eight round trips on the one workgroup the launch waits for).  Same sums in the same order.
        const int idx = h + t * N;
        double sH[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sT[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (!F.adhtd_done && !dist) {
            const double* AH = F.adH + 64 * (size_t)idx; const double* AT = F.adT + 64 * (size_t)idx;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const double dh = s_delta[h][i], dt = s_delta[t][i];
#pragma unroll
                for (int j = 0; j < 8; j++) { sH[j] += dh * AH[i * 8 + j]; sT[j] += dt * AT[i * 8 + j]; }
            }
        }
        const SE3 ll = Wt * Ch;                                        // DSOFrame.h:259-273
        cmlhip_ba_pair& P = F.pairs[q];
        double R[9];
        ll.matrix(R);
        double a, b;
        Exposure(s_aff[h][0], s_aff[h][1], s_aff[h][2]).to(Exposure(s_aff[t][0], s_aff[t][1], s_aff[t][2]), a, b);
#pragma unroll
        for (int k = 0; k < 9; k++) P.R[k] = R[k];
#pragma unroll
        for (int k = 0; k < 3; k++) P.t[k] = ll.t[k];
        P.aff_a = a; P.aff_b = b;
        if (F.adhtd_done || dist) continue;
#pragma unroll
        for (int j = 0; j < 8; j++) F.adHTd[8 * (size_t)idx + j] = (float)(sH[j] + sT[j]);
    }
    FS_STAMP(3);
}

// state + step - state_zero of frame f, entry k, exactly as frame_step_block forms it (setStep's non-finite and fix_pose rules included)
__device__ __forceinline__ double frame_stepped_delta(const cmlhip_ba_frame_state* fs, const double* __restrict__ x, int f, int k) {
    const cmlhip_ba_frame_state& S = fs[f];
    bool fin = true;
#pragma unroll
    for (int i = 0; i < 8; i++) fin = fin && isfinite(-x[4 + 8 * f + i]);
    double step = -x[4 + 8 * f + k];
    if (!fin) step = 0.0;
    if (S.fix_pose && k < 6) step = 0.0;
    const double st = S.state[k] + step;
    return st - S.state_zero[k];
}
#pragma clang fp contract(fast)
