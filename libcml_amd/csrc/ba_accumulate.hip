// ba_accumulate.hip — Hessian accumulation, point Schur complement, dense solve and back-substitution of the
// sliding-window BA.  Replaces addToHessianTop + AccumulatorApprox (BA.cpp:1648-1779, ACC.h:613-998),
// stitchDoubleTop (BA.cpp:1781-1878), addToHessianSC + stitchDoubleSC (BA.cpp:1880-2043),
// solveLevenbergMarquardt's factorisation (BA.cpp:1284-1320) and the resubstitution loop (BA.cpp:1427-1487).
//
// Design (MI355X-first, not a translation).  One Gauss-Newton iteration is FOUR dependent launches after the
// residual kernel, each of them horizontally fused so that no single-workgroup stage sits alone on the chip:
//  K3 k_ba_acc      blocks [0,N^2): one workgroup per (host,target) pair — 4 lanes per residual, each wave owns a
//                   quarter of the 91 unique entries of the 13x13 block, wave-shuffle reduction, then the fp64
//                   adjoint sandwiches (AH B AH^T, ...) straight from LDS.
//                   blocks [N^2,..): 8 lanes per point — Hdd/bd/Hcd, HdiF, and the point's coupling row
//                   g_p = dH/d(idepth) in FRAME coordinates (AH/AT applied per residual).
//  K4 k_ba_system   one workgroup per 16x16 tile of the (8N+4)^2 system: H_sc = G^T diag(HdiF) [G | bdSum] as an
//                   fp64 SYRK on the matrix cores (v_mfma_f64_16x16x4_f64, waves split the point range, fixed-order
//                   LDS reduction), plus the tile of H_A / H_L (pair blocks + priors) and of the final LM system.
//                   The reference's N^3 bucket accumulators and O(N^3) stitch (BA.cpp:1939-2043) do not exist here.
//  K5 k_ba_solve    workgroup 0: Jacobi scaling + blocked LDL^T of the trailing 8N block in LDS (fp64, block-packed
//                   lower storage, MFMA trailing updates); workgroup 1: energy sum + exact 70th-percentile
//                   threshold of the newest frame (radix select) for the NEXT residual pass.
//  K6 k_ba_backsub  xAd in LDS, per-point step, point update (doStepFromBackup), fixed-order partial sums.
#include "cmlhip_internal.h"
#include "ba_common.h"
#include "reproj_dev.h"
#include "ba_finish.h"
#include "ba_frames.h"

typedef double double4_ __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// wave64 sum on the DPP path (no LDS crossbar): row_shr 1/2/4/8 inside each 16-lane row, then row_bcast:15 and
// row_bcast:31 carry the row totals forward.  The TOTAL IS VALID IN LANE 63 ONLY.
__device__ __forceinline__ float wave_sum_dpp63(float v) {
#define DPP_ADD(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, false))
    DPP_ADD(0x111, 0xf);      // row_shr:1
    DPP_ADD(0x112, 0xf);      // row_shr:2
    DPP_ADD(0x114, 0xf);      // row_shr:4
    DPP_ADD(0x118, 0xf);      // row_shr:8
    DPP_ADD(0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    DPP_ADD(0x143, 0xc);      // row_bcast:31 into rows 2 and 3
#undef DPP_ADD
    return v;
}

__device__ __forceinline__ double wave_sum_dpp63_d(double v) {      // the same path for a double (two 32-bit DPP moves per step, one fp64 add)
#define DPP_ADD_D(ctrl, rmask) do { \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xf, false); \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xf, false); \
        v += __hiloint2double(hi_, lo_); } while (0)
    DPP_ADD_D(0x111, 0xf); DPP_ADD_D(0x112, 0xf); DPP_ADD_D(0x114, 0xf); DPP_ADD_D(0x118, 0xf); DPP_ADD_D(0x142, 0xa); DPP_ADD_D(0x143, 0xc);
#undef DPP_ADD_D
    return v;
}

__device__ __forceinline__ float sum8(float v) {          // sum over 8 consecutive lanes (xor butterflies stay inside the group)
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    return v;
}
__device__ __forceinline__ double sum8d(double v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    return v;
}

struct AccArgs {
    const double* adH; const double* adT; const float* adHTd; const double* cdelta;
    float* acc_out; int* num_out; double* pair_blocks;
    double* G; double* Wt; int ldg; int do_backup;
    int tile;                                               // residual slots per tile of this window (16 or 64)
    const float* part; const int* tile_off;                 // CML_MODE_ACTIVE_TILES: wave tiles of the resident residual kernel, tiles of pair q = [tile_off[q], tile_off[q+1])
};
#define CML_MODE_ACTIVE_TILES 3        // ACTIVE pair blocks summed from the 16x16 tiles k_ba_lin_rs left (no records read)

// ------------------------------------------------------------------------------------------------ K3
#define PAIR_TRIP 256          // residuals of a pair staged per trip (4 groups x 64 lanes)
#define PAIR_REC 41            // staged floats per record (40 used) + 1: odd stride, conflict-free lane-per-record reads
typedef float float4_ __attribute__((ext_vector_type(4)));

// One workgroup per (host,target) pair.  The 13x13 block of AccumulatorApprox (ACC.h:776-932) is a sum of small outer
// products per residual, which is one v_mfma_f32_16x16x4_f32 (IEEE fp32) per residual:
//   k = 0:  A = x (10)        B = [a x + b y (10) | JabJIdx(0,0) JabJIdx(0,1) JI^T r(0)]          x = [dC0 | dXi0], y = [dC1 | dXi1]
//   k = 1:  A = y (10)        B = [b x + c y (10) | JabJIdx(1,0) JabJIdx(1,1) JI^T r(1)]          [a b; b c] = JIdx2
//   k = 2:  A = e_10          B = [0 (10) | Jab2(0,0) Jab2(0,1) Jab^T r(0) Jab2(1,1) Jab^T r(1) r^T r]   (updateBotRight)
// so D[i][j], i,j < 10 is the top-left block, D[i][10..12] the top-right one and D[10][10..15] the six bottom-right sums.
// The reduction over the residuals is the K dimension of the matrix core: no per-lane accumulators, no wave reductions.
// The 38 floats of a record that are needed ([8,28) and [60,80) of the 80) are staged in LDS with coalesced 16-B loads
// (10 per record), PAIR_TRIP residuals per trip; wave w takes residuals w, w+16, ... of the trip and every lane reads its
// operand elements from the staged record (distinct banks or broadcast).  The 16 wave tiles are added in wave order.
// LINEARIZED mode (rare) computes res_toZero + J*delta per residual (BA.cpp:1699-1729) and stages the same 38 floats.
// NW = 16: the 1024-thread workgroup of every mode.  NW = 4 (resident loop only, mode == CML_MODE_ACTIVE_TILES): the same sums by a
// 256-thread workgroup — wave w carries the running sums of the virtual waves w, w + 4, w + 8, w + 12 of the 16-wave form, and the 16
// sums are added in the same order: bit-identical, at a quarter of the threads (k_ba_acc_rs).
template <int NW = 16>
__device__ __forceinline__ void acc_pair_block(const BAArgs& A, const AccArgs& X, const int q, const int mode, unsigned char* arena) {
    const bool TILES = mode == CML_MODE_ACTIVE_TILES || NW != 16;
    const bool LIN = mode != CMLHIP_MODE_ACTIVE && !TILES;  // LINEARIZED and MARGINALIZED walk the plain pair list
    float (*s_rec)[PAIR_REC] = reinterpret_cast<float (*)[PAIR_REC]>(arena);
    float (*s_tile)[256] = reinterpret_cast<float (*)[256]>(arena + (NW == 16 ? sizeof(float) * PAIR_TRIP * PAIR_REC : 0));
    __shared__ int s_cnt;
    __shared__ double s_H[13][13];
    __shared__ double s_AH[64], s_AT[64], s_T1[64], s_T2[64];
    const int tid = threadIdx.x, wave = tid >> 6, ln = tid & 63;
    const int beg = A.by_pair_off[q], end = A.by_pair_off[q + 1];
    const double ahv = X.adH[64 * (size_t)q + ln], atv = X.adT[64 * (size_t)q + ln];      // adjoints for the stitch: in flight under the loop
    const int tb = TILES ? X.tile_off[q] : 0, te = TILES ? X.tile_off[q + 1] : 0;           // (requested here: behind the barrier it is a round trip of its own)
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (A.dbg && tid == 0 && q == 1) A.dbg[16] = wall_clock64();
    // ---- this lane's operand elements: i = j = lane & 15, k = lane >> 4; staged offsets of x_i, y_i and the two multipliers
    const int e = ln & 15, kq = ln >> 4;
    const int ox = e < 4 ? 12 + e : e - 4, oy = e < 4 ? 16 + e : 2 + e;       // x = [O_C0 | O_XI0], y = [O_C1 | O_XI1] (e < 10)
    int o1 = 0, o2 = 0, o3 = 0, o4 = 0;
    bool prod = false, field = false;
    if (kq < 2 && e < 10) { prod = true; o1 = ox; o2 = oy; o3 = kq == 0 ? 22 : 24; o4 = kq == 0 ? 24 : 25; }   // (a,b) / (b,c)
    else if (kq < 2 && e < 13) { field = true; o1 = e < 12 ? 26 + (e - 10) + 2 * kq : 34 + kq; }               // JabJIdx(k, .), JI^T r(k)
    else if (kq == 2 && e >= 10) { field = true; o1 = e == 10 ? 30 : e == 11 ? 32 : e == 12 ? 36 : e == 13 ? 33 : e == 14 ? 37 : 38; }
    const float a_const = (kq == 2 && e == 10) ? 1.f : 0.f;
    float4_ acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NW != 16) {
        // four virtual waves per wave: all their tiles requested together, each virtual wave's sum taken in its own order
        const float4* P4 = reinterpret_cast<const float4*>(X.part);
        const int tpt = PAIR_TRIP / X.tile;
#pragma unroll
        for (int u = 0; u < 16 / NW; u++) {
            const int vw = wave + NW * u;
            float4_ va = {0.f, 0.f, 0.f, 0.f};
            for (int t = tb + vw; vw < tpt && t < te; t += tpt) {
                const float4 v = P4[(size_t)t * 64 + ln];
                va[0] += v.x; va[1] += v.y; va[2] += v.z; va[3] += v.w;
            }
#pragma unroll
            for (int rg = 0; rg < 4; rg++) s_tile[vw][(4 * kq + rg) * 16 + e] = va[rg];
        }
    } else if (TILES) {
        // the residual kernel of the resident loop already reduced its residuals on the matrix cores: add the pair's wave tiles
        // (same D layout, lane for lane); wave w < 256 / tile takes tiles w, w + 256 / tile, ... (the assignment of the record path below) and the wave sums are added in wave order below
        const float4* P4 = reinterpret_cast<const float4*>(X.part);
        const int tpt = PAIR_TRIP / X.tile;                  // tiles per trip = waves that own a tile
        for (int t = tb + wave; wave < tpt && t < te; t += tpt) {
            const float4 v = P4[(size_t)t * 64 + ln];
            acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        }
    }
    const int span = TILES ? 0 : (LIN ? end - beg : A.pair_stride);            // ACTIVE mode walks the fixed-stride list: no offset round trip
    for (int t0 = 0; t0 < span; t0 += PAIR_TRIP) {
        const int trip = beg + t0;
        const int ntrip = min(PAIR_TRIP, span - t0);
        if (!LIN) {
            // two memory round trips per trip: the efsJ codes applyRes keeps per pair slot (2r+sel, -1 = not good / not
            // ACTIVE, BA.cpp:1662-1669), then 16 B of the record per thread (every load unconditional, addresses clamped)
            const int* codes = A.pair_code + (size_t)q * A.pair_stride + t0;
            int code[3], rec[3], part[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int idx = tid + 1024 * u;
                rec[u] = idx / 10; part[u] = idx % 10;
                code[u] = codes[min(rec[u], ntrip - 1)];
            }
            float4 v[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int cu = max(code[u], 0);
                const float4* J4 = reinterpret_cast<const float4*>(((cu & 1) ? A.rj1 : A.rj0) + (size_t)(cu >> 1) * RJ_STRIDE);   // efsJ
                v[u] = J4[part[u] < 5 ? 2 + part[u] : 10 + part[u]];
            }
#pragma unroll
            for (int u = 0; u < 3; u++) {
                if (rec[u] < ntrip) {
                    float* d = &s_rec[rec[u]][4 * part[u]];
                    const bool ok = code[u] >= 0;                     // a slot that is not summed is staged as zeros: the MFMA loop is branch-free
                    d[0] = ok ? v[u].x : 0.f; d[1] = ok ? v[u].y : 0.f; d[2] = ok ? v[u].z : 0.f; d[3] = ok ? v[u].w : 0.f;
                    if (part[u] == 0 && ok) atomicAdd(&s_cnt, 1);
                }
            }
        } else if (tid < ntrip) {
            const int r = A.by_pair[trip + tid];
            // LINEARIZED: the window's linearized good residuals (BA.cpp:1666-1669).  MARGINALIZED: every good residual of
            // the points being marginalised (:1670-1674), residual vector = res_toZero as it is (:1689-1692)
            const bool ok = mode == CMLHIP_MODE_MARGINALIZED ? (A.r_good[r] && A.pt_mask[A.r_point[r]]) : (A.r_lin[r] && A.r_good[r]);
            if (!ok) { for (int j = 0; j < 40; j++) s_rec[tid][j] = 0.f; }
            if (ok) {
                atomicAdd(&s_cnt, 1);
                const float* J = (A.r_sel[r] ? A.rj1 : A.rj0) + (size_t)r * RJ_STRIDE;    // efsJ
                float* S = s_rec[tid];
                for (int j = 0; j < 20; j++) S[j] = J[8 + j];
                for (int j = 0; j < 14; j++) S[20 + j] = J[60 + j];
                // BA.cpp:1699-1729 (res_toZero + J*delta; see oracle note on the reference's float*/double[8] store)
                const float* dp = X.adHTd + 8 * q;
                const int p = A.r_point[r];
                const float dd = (float)(A.pt_idepth[p] - (double)A.pt_idepth_zero[p]);
                const float Jpx = cml_jp_delta(J + O_XI0, dp, J + O_C0, X.cdelta, J[O_DD], dd, false);
                const float Jpy = cml_jp_delta(J + O_XI1, dp, J + O_C1, X.cdelta, J[O_DD + 1], dd, false);
                double s0 = 0, s1 = 0, s2 = 0, s3 = 0; float srr = 0;
                for (int j = 0; j < 8; j++) {
                    float rtz = A.r_rtz[8 * (size_t)r + j];
                    if (mode == CMLHIP_MODE_LINEARIZED) {
                        rtz = rtz + J[O_JI0 + j] * Jpx; rtz = rtz + J[O_JI1 + j] * Jpy;
                        rtz = rtz + J[O_JAB0 + j] * dp[6]; rtz = rtz + J[O_JAB1 + j] * dp[7];
                    }
                    const double ra = (double)rtz;
                    s0 += ra * (double)J[O_JI0 + j]; s1 += ra * (double)J[O_JI1 + j];
                    s2 += ra * (double)J[O_JAB0 + j]; s3 += ra * (double)J[O_JAB1 + j];
                    srr = (float)((double)srr + ra * ra);
                }
                S[34] = (float)s0; S[35] = (float)s1; S[36] = (float)s2; S[37] = (float)s3; S[38] = srr;
            }
        }
        __syncthreads();
        {
            // Summation order = the one of the resident residual kernel (ba_linearize_rs.hip), so that a loop driven from the host
            // (records) and the device-resident loop (tiles) produce the same bits: a TILE is X.tile (16 or 64) consecutive slots of the pair
            // list, accumulated from zero in slot order; wave w < 256 / tile owns tile w of the trip and adds it to its running sum.
#pragma clang fp contract(off)
            float4_ tacc = {0.f, 0.f, 0.f, 0.f};
            const int lbeg = X.tile * wave, lend = min(lbeg + X.tile, ntrip);
#pragma unroll 4
            for (int li = lbeg; li < lend; li++) {
                const float* S = s_rec[li];
                const float v1 = S[o1], v2 = S[o2], m1 = S[o3], m2 = S[o4];
                const float av = prod ? (kq == 0 ? v1 : v2) : a_const;
                const float bv = prod ? m1 * v1 + m2 * v2 : (field ? v1 : 0.f);
                tacc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, tacc, 0, 0, 0);
            }
            acc[0] += tacc[0]; acc[1] += tacc[1]; acc[2] += tacc[2]; acc[3] += tacc[3];
        }
        if (t0 + PAIR_TRIP < span) __syncthreads();               // the staging tile is reused
    }
    if (A.dbg && tid == 0 && q == 1) A.dbg[17] = wall_clock64();
    // D: col = lane & 15, row = 4 * (lane >> 4) + reg
    if constexpr (NW == 16) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) s_tile[wave][(4 * kq + rg) * 16 + e] = acc[rg];
    }
    if (tid < 64) { s_AH[tid] = ahv; s_AT[tid] = atv; }
    __syncthreads();
    if (tid < 91) {
        // canonical 91-entry order: 55 upper-tri row-major of the 10x10, 30 top-right (row-major 10x3), 6 bottom-right;
        // symmetric 13x13 (AccumulatorApprox::finish, ACC.h:639-673)
        int rr_, cc, src;
        if (tid < 55) {
            int k = tid; rr_ = 0;
            while (k >= 10 - rr_) { k -= 10 - rr_; rr_++; }
            cc = rr_ + k; src = rr_ * 16 + cc;
        } else if (tid < 85) {
            rr_ = (tid - 55) / 3; cc = 10 + (tid - 55) % 3; src = rr_ * 16 + cc;
        } else {
            const int ee = tid - 85;
            rr_ = ee < 3 ? 10 : (ee < 5 ? 11 : 12);
            cc = ee < 3 ? 10 + ee : (ee < 5 ? 11 + (ee - 3) : 12);
            src = 10 * 16 + 10 + ee;
        }
        float v = s_tile[0][src];
#pragma unroll
        for (int w = 1; w < 16; w++) v += s_tile[w][src];
        X.acc_out[(size_t)q * ACC_STRIDE + tid] = v;
        s_H[rr_][cc] = (double)v; s_H[cc][rr_] = (double)v;
    }
    if (tid == 0) X.num_out[q] = s_cnt;
    __syncthreads();
    if (A.dbg && tid == 0 && q == 1) A.dbg[18] = wall_clock64();
    // ---- stitchDoubleTop per-pair products, BA.cpp:1827-1843 (fp64)
    double* pb = X.pair_blocks + (size_t)q * PB_STRIDE;
    if (tid < 64) {
        const int a = tid >> 3, b = tid & 7;
        double t1 = 0, t2 = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { t1 += s_AH[a * 8 + k] * s_H[4 + k][4 + b]; t2 += s_AT[a * 8 + k] * s_H[4 + k][4 + b]; }
        s_T1[tid] = t1; s_T2[tid] = t2;
    }
    __syncthreads();
    if (tid < 64) {
        const int a = tid >> 3, b = tid & 7;
        double hh = 0, tt = 0, ht = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            hh += s_T1[a * 8 + k] * s_AH[b * 8 + k];
            tt += s_T2[a * 8 + k] * s_AT[b * 8 + k];
            ht += s_T1[a * 8 + k] * s_AT[b * 8 + k];
        }
        pb[PB_HH + tid] = hh; pb[PB_TT + tid] = tt; pb[PB_HT + tid] = ht;
    } else if (tid < 96) {
        const int e = tid - 64, a = e >> 2, b = e & 3;
        double hc = 0, tc = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { hc += s_AH[a * 8 + k] * s_H[4 + k][b]; tc += s_AT[a * 8 + k] * s_H[4 + k][b]; }
        pb[PB_HC + e] = hc; pb[PB_TC + e] = tc;
    } else if (tid < 104) {
        const int a = tid - 96;
        double bh = 0, bt = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { bh += s_AH[a * 8 + k] * s_H[4 + k][12]; bt += s_AT[a * 8 + k] * s_H[4 + k][12]; }
        pb[PB_BH + a] = bh; pb[PB_BT + a] = bt;
    } else if (tid < 120) {
        const int e = tid - 104;
        pb[PB_CC + e] = s_H[e >> 2][e & 3];
    } else if (tid < 124) {
        pb[PB_BC + tid - 120] = s_H[tid - 120][12];
    }
    if (A.dbg && tid == 0 && q == 1) A.dbg[19] = wall_clock64();
}

// One WAVE per point: lane = (slot s = lane>>3, component a = lane&7); slot s walks the point's s-th residual (a point
// has at most N-1 of them, so one pass for N <= 9), component a owns output a of the two 8x8 adjoint products.  The
// dependent chain by_point -> r -> {good, sel, target} -> record is walked once per point, the 1 KB of AH/AT a residual
// needs is spread over 8 lanes x 128 B, and 16 points per workgroup put the rows on >= P/16 CUs.
// Per point: Hdd/bd/Hcd sums (BA.cpp:1747-1750), HdiF, bdSum (BA.cpp:1895-1905); coupling row in FRAME coordinates:
// g_p[0:4] = Hcd, g_p[4+8h+i] = sum_r (AH_ht JpJdF_r)_i, g_p[4+8t+i] = (AT_ht JpJdF_r)_i, G[p][n] = bdSum.
// The row is assembled in LDS and leaves the CU as one coalesced store.
#define PT_PER_BLOCK 16
#define LDG_MAX (((8 * CMLHIP_MAX_FRAMES + 4) + 1 + 15) / 16 * 16)
__device__ __forceinline__ float sum_slots(float v) {     // sum over the 8 slots (xor butterflies over lane bits 3..5)
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ double sum_slots_d(double v) {
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
}
template <int PPB = PT_PER_BLOCK>
__device__ __forceinline__ void point_rows_block(const BAArgs& A, const AccArgs& X, const int blk, unsigned char* arena) {
    double (*s_row)[LDG_MAX] = reinterpret_cast<double (*)[LDG_MAX]>(arena);
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, s = l >> 3, a = l & 7;
    const int p = blk * PPB + wv;
    if (p >= A.P) return;                                    // wave-uniform
    double* srow = s_row[wv];
    for (int c = l; c < X.ldg; c += 64) srow[c] = 0.0;
    const int host = A.pt_host[p];
    // (requested with the first round trip: read where they are used — behind the shuffles, inside `if (ngood > 0)` — they are a third one)
    float* pa = A.pt_acc + (size_t)p * PT_ACC_STRIDE;
    const float bdLv = pa[7];                              // written by the LINEARIZED pre-pass (0 when none)
    const float prior_p = A.pt_prior[p];
    const double idepth_p = A.pt_idepth[p];
    const float idepth_zero_p = A.pt_idepth_zero[p];
    // component a of a slot computes one of the per-residual scalars: a<4 Hcd[a], a=4 Hdd, a=5 bd, a=6 good count
    float accA = 0.f, accL = 0.f;
    double hostacc = 0.0;
    // Round trips: {efsJ code kept by applyRes, static target} of EVERY slot of the point first (up to 4 passes of 8 slots: one trip
    // for all of them), then per pass the record fields — every load unconditional on clamped indices, so that the next pass's loads
    // are in flight under the arithmetic of the current one (one pass for N <= 9).
    // (round 6: the slot's residual comes from the STATIC table point_res and its isActiveAndIsGoodNEW flag rides in the second trip beside the row —
    //  applyRes no longer scatters a per-slot code: one gathered 4-byte store per residual less in the residual kernels, same number of trips here)
    int ress[4], tgls[4];
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
        const int slot = p * A.pt_stride + min(8 * ps + s, A.pt_stride - 1);
        const bool live = 8 * ps + s < A.pt_stride;
        const int rs_ = A.point_res[slot], tg = A.point_tgt[slot];
        ress[ps] = live ? rs_ : -1; tgls[ps] = live ? tg : 0;
    }
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
        if (8 * ps >= A.pt_stride) break;                     // wave-uniform
        const int tgl = tgls[ps];
        const int r = max(ress[ps], 0);
        const bool good = ress[ps] >= 0 && A.r_good[r] != 0;
        const bool lin = good && (tgl & 256);
        const float* PS = A.r_jpjdf + PS_STRIDE * (size_t)r;          // one 64-B line per residual: JpJdF + the residual's terms of Hcd, Hdd, bd (kept by applyRes)
        const int t = min(max(tgl, 0) & 255, A.N - 1);        // (clamped: empty slots load pair 0..N-1 and are masked)
        const int q = host + t * A.N;
        const float4 v0 = *reinterpret_cast<const float4*>(PS), v1 = *reinterpret_cast<const float4*>(PS + 4);
        const double4_* AH = reinterpret_cast<const double4_*>(X.adH + 64 * (size_t)q + 8 * a);
        const double4_* AT = reinterpret_cast<const double4_*>(X.adT + 64 * (size_t)q + 8 * a);
        const double4_ h0 = AH[0], h1 = AH[1], t0 = AT[0], t1 = AT[1];
        float val = PS[8 + (a < 6 ? a : 0)];                  // Hcd[a] (a < 4), Hdd (a == 4), bd (a == 5)
        if (a == 5 && lin) val = 0.f;                         // bdL: k_ba_point_bdL
        if (a == 6) val = 1.f;
        if (a == 7) val = 0.f;
        if (!good) val = 0.f;
        double sh = 0.0, st = 0.0;
        const double v[8] = {(double)v0.x, (double)v0.y, (double)v0.z, (double)v0.w, (double)v1.x, (double)v1.y, (double)v1.z, (double)v1.w};
#pragma unroll
        for (int j = 0; j < 4; j++) { sh += h0[j] * v[j]; st += t0[j] * v[j]; }
#pragma unroll
        for (int j = 0; j < 4; j++) { sh += h1[j] * v[4 + j]; st += t1[j] * v[4 + j]; }
        if (good) srow[4 + 8 * t + a] = st;                   // one residual per (point, target): plain store
        else sh = 0.0;
        // every lane takes part in every shuffle; LINEARIZED residuals (rare) are routed to the L sums by masking
        accA += sum_slots(lin ? 0.f : val);
        accL += sum_slots(lin ? val : 0.f);
        hostacc += sum_slots_d(sh);
    }
    // gather the point's scalars (valid in every lane): lanes 0..6 of slot 0 hold items 0..6
    const float HcdA_a = __shfl(accA, a & 3), HcdL_a = __shfl(accL, a & 3);
    const float HddA = __shfl(accA, 4), HddL = __shfl(accL, 4), bdA = __shfl(accA, 5);
    const int ngood = (int)(__shfl(accA, 6) + __shfl(accL, 6));
    float HdiF = 0.f, bdSum = 0.f;
    if (ngood > 0) {
        float H = HddA + HddL + prior_p;
        if (H < 1e-10) H = 1e-10;
        HdiF = (float)(1.0 / H);
        bdSum = bdA + bdLv;
        const float deltaF = (float)(idepth_p - (double)idepth_zero_p);
        bdSum += prior_p * deltaF;                         // shiftPriorToZero, :1904
        if (s == 0) {
            srow[4 + 8 * host + a] = hostacc;
            if (a < 4) srow[a] = (double)(HcdA_a + HcdL_a);
            if (a == 4) srow[A.n] = (double)bdSum;
        }
    }
    if (s == 0) {
        // pt_acc: HddA bdA HcdA[4] HddL bdL HcdL[4] HdiF bdSum
        if (a < 4) { pa[2 + a] = HcdA_a; pa[8 + a] = HcdL_a; }
        else if (a == 4) { pa[0] = HddA; pa[6] = HddL; }
        else if (a == 5) { pa[1] = bdA; pa[12] = HdiF; }
        else if (a == 6) { pa[13] = bdSum; X.Wt[p] = (double)HdiF; }
        else if (X.do_backup) A.pt_backup[p] = (float)idepth_p;                // backupState, BA.cpp:919-922
    }
    double* row = X.G + (size_t)p * X.ldg;                 // LDS accesses of one wave are ordered: no barrier needed
    for (int c = l; c < X.ldg; c += 64) row[c] = srow[c];
}

// mode: ACTIVE = pair blocks + point rows; LINEARIZED / MARGINALIZED = pair blocks only (rare paths, own point kernels)
__device__ __forceinline__ void k_ba_acc_body(const BAArgs& A, const AccArgs& X, const int mode, const int bx_, const int gx_) {
    const int NN = A.N * A.N;
    DBG_BLK(A.dbg, 1, 0);
    if (A.ctl && A.ctl->stop) {                            // converged: the loop of BA::run has left (BA.cpp:879)
        // the residual kernel must not test the flag its own launch publishes (blocks scheduled after the publishing one would
        // skip residuals of the pass that must still complete): it tests `stop_lin`, which is raised HERE, one launch later
        if (bx_ == 0 && threadIdx.x == 0) A.ctl->stop_lin = 1;
        return;
    }
    // the point-row workgroups (16 points per 1024-thread block) take about twice as long as the pair workgroups at a wide window: they
    // are handed out FIRST, the short pair workgroups fill the tail of the launch
    const int npt = (int)gx_ - NN;
    __shared__ __attribute__((aligned(16))) unsigned char s_arena[sizeof(float) * PAIR_TRIP * PAIR_REC + sizeof(float) * 16 * 256];
    static_assert(sizeof(s_arena) >= sizeof(double) * PT_PER_BLOCK * LDG_MAX, "arena");
    if ((int)bx_ < npt) point_rows_block(A, X, bx_, s_arena);
    else acc_pair_block(A, X, bx_ - npt, mode, s_arena);
    DBG_BLK_END(A.dbg, 1);
}
// Resident loop (mode CML_MODE_ACTIVE_TILES): the same two kinds of workgroups at 256 threads — 4 points, or one pair summed by 4 waves
// (acc_pair_block<4>).  Neither kind needs 16 waves there (a point row is one wave's work; the pair's tiles are 16 KB), and a
// 1024-thread workgroup at 94 VGPRs is alone on its CU: 900 of them at 20 keyframes x 8000 points were 3.5 rounds, eight batched
// config-B windows six.  20 KB of LDS instead of 62.
#define PT_PER_BLOCK_RS 4
__device__ __forceinline__ void k_ba_acc_rs_body(const BAArgs& A, const AccArgs& X, const int bx_, const int gx_) {
    const int NN = A.N * A.N;
    DBG_BLK(A.dbg, 1, 0);
    if (A.ctl && A.ctl->stop) {
        if (bx_ == 0 && threadIdx.x == 0) A.ctl->stop_lin = 1;          // (see k_ba_acc_body)
        return;
    }
    __shared__ __attribute__((aligned(16))) unsigned char s_arena[sizeof(float) * 16 * 256];
    static_assert(sizeof(s_arena) >= sizeof(double) * PT_PER_BLOCK_RS * LDG_MAX, "arena");
    // (the pair workgroups — two dependent trips, two barriers, the fp64 stitch — are the longer kind here, at config B (4.2 against
    //  3.1 us) as at config E: handed out FIRST.  Same-box A/B at config B: iteration 46.2 -> 45.4 us; E unchanged.  Their tile ranges
    //  handed over in the kernel arguments — one dependent trip less — measured without effect once they start first.)
    if ((int)bx_ < NN) acc_pair_block<4>(A, X, bx_, CML_MODE_ACTIVE_TILES, s_arena);
    else point_rows_block<PT_PER_BLOCK_RS>(A, X, bx_ - NN, s_arena);
    DBG_BLK_END(A.dbg, 1);
}
__global__ __launch_bounds__(256) void k_ba_acc_rs(BAArgs A, AccArgs X) { k_ba_acc_rs_body(A, X, blockIdx.x, gridDim.x); }
#ifdef CML_ACC_WPE8
#define CML_ACC_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))      /* development: two 1024-thread workgroups per CU at 64 VGPRs (192 B of scratch) */
#else
#define CML_ACC_ATTR
#endif
__global__ __launch_bounds__(1024) CML_ACC_ATTR void k_ba_acc(BAArgs A, AccArgs X, int mode) { k_ba_acc_body(A, X, mode, blockIdx.x, gridDim.x); }


// LINEARIZED-mode bd (BA.cpp:1699-1750), rare path: one thread per point
__global__ void k_ba_point_bdL(BAArgs A, const float* __restrict__ adHTd, const double* __restrict__ cdelta) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.P) return;
    const int host = A.pt_host[p];
    const float dd = (float)(A.pt_idepth[p] - (double)A.pt_idepth_zero[p]);
    float bd = 0;
    for (int kk = A.by_point_off[p]; kk < A.by_point_off[p + 1]; kk++) {
        const int r = A.by_point[kk];
        if (!A.r_lin[r] || !A.r_good[r]) continue;
        const float* J = (A.r_sel[r] ? A.rj1 : A.rj0) + (size_t)r * RJ_STRIDE;
        const float* dp = adHTd + 8 * (host + A.r_target[r] * A.N);
        const float Jpx = cml_jp_delta(J + O_XI0, dp, J + O_C0, cdelta, J[O_DD], dd, false);
        const float Jpy = cml_jp_delta(J + O_XI1, dp, J + O_C1, cdelta, J[O_DD + 1], dd, false);
        double s0 = 0, s1 = 0;
        for (int j = 0; j < 8; j++) {
            float rtz = A.r_rtz[8 * (size_t)r + j];
            rtz = rtz + J[O_JI0 + j] * Jpx; rtz = rtz + J[O_JI1 + j] * Jpy;
            rtz = rtz + J[O_JAB0 + j] * dp[6]; rtz = rtz + J[O_JAB1 + j] * dp[7];
            s0 += (double)rtz * (double)J[O_JI0 + j]; s1 += (double)rtz * (double)J[O_JI1 + j];
        }
        bd = (float)((double)bd + (s0 * (double)J[O_DD] + s1 * (double)J[O_DD + 1]));
    }
    A.pt_acc[(size_t)p * PT_ACC_STRIDE + 7] = bd;
}

// marginalizePointsF, point side (BA.cpp:2490-2493): for the points being marginalised, addToHessianTop(MARGINALIZED)'s
// Hdd/bd/Hcd (stored as the L sums, the A sums zeroed, :1763-1775) with res_toZero as the residual vector, addToHessianSC's
// HdiF and bdSum WITHOUT the prior shift (shiftPriorToZero = false), and the coupling row in frame coordinates.  Every
// other point gets weight 0.  Once per keyframe: one thread per point.
__global__ void k_ba_point_rows_marg(BAArgs A, AccArgs X) {
    // 8 lanes per point (round 4: one thread per point before, 39 us per keyframe for ~400 selected points): lane a owns row a of the two
    // adjoint products of every residual; the pattern sums of a residual are formed by every lane of the group (same numbers), the
    // per-point outputs are written by lane 0.  Same expressions in the same order per entry.
    const int gt = blockIdx.x * blockDim.x + threadIdx.x, p = gt >> 3, a = gt & 7;
    if (p >= A.P) return;
    double* row = X.G + (size_t)p * X.ldg;                 // (zeroed by the launcher: a thread striding through its own row is the worst store pattern there is)
    float* pa = A.pt_acc + (size_t)p * PT_ACC_STRIDE;
    if (a == 0) { for (int k = 0; k < 14; k++) pa[k] = 0.f; X.Wt[p] = 0.0; }
    if (!A.pt_mask[p]) return;
    const int host = A.pt_host[p];
    float Hdd = 0, bd = 0, Hcd[4] = {0, 0, 0, 0};
    int ngood = 0;
    double acc_h = 0.0;                                     // row[4 + 8 * host + a], accumulated over the point's residuals in list order
    for (int kk = A.by_point_off[p]; kk < A.by_point_off[p + 1]; kk++) {
        const int r = A.by_point[kk];
        if (!A.r_good[r]) continue;
        ngood++;
        const float* J = (A.r_sel[r] ? A.rj1 : A.rj0) + (size_t)r * RJ_STRIDE;
        double s0 = 0, s1 = 0;
        for (int j = 0; j < 8; j++) {
            const double ra = (double)A.r_rtz[8 * (size_t)r + j];
            s0 += ra * (double)J[O_JI0 + j]; s1 += ra * (double)J[O_JI1 + j];
        }
        const float g0 = J[O_JI2 + 0] * J[O_DD] + J[O_JI2 + 2] * J[O_DD + 1];
        const float g1 = J[O_JI2 + 1] * J[O_DD] + J[O_JI2 + 3] * J[O_DD + 1];
        bd = (float)((double)bd + (s0 * (double)J[O_DD] + s1 * (double)J[O_DD + 1]));
        Hdd += g0 * J[O_DD] + g1 * J[O_DD + 1];
        for (int j = 0; j < 4; j++) Hcd[j] += J[O_C0 + j] * g0 + J[O_C1 + j] * g1;
        const int t = A.r_target[r], q = host + t * A.N;
        const float* v = A.r_jpjdf + PS_STRIDE * (size_t)r;
        const double* AH = X.adH + 64 * (size_t)q; const double* AT = X.adT + 64 * (size_t)q;
        double sh = 0, st = 0;
        for (int j = 0; j < 8; j++) { sh += AH[a * 8 + j] * (double)v[j]; st += AT[a * 8 + j] * (double)v[j]; }
        acc_h += sh;
        row[4 + 8 * t + a] = st;
    }
    row[4 + 8 * host + a] = acc_h;
    if (a != 0) return;
    pa[6] = Hdd; pa[7] = bd; pa[8] = Hcd[0]; pa[9] = Hcd[1]; pa[10] = Hcd[2]; pa[11] = Hcd[3];
    if (ngood == 0) return;
    float H = Hdd + A.pt_prior[p];
    if (H < 1e-10) H = 1e-10;
    const float HdiF = (float)(1.0 / H);
    pa[12] = HdiF; pa[13] = bd;
    for (int j = 0; j < 4; j++) row[j] = (double)Hcd[j];
    row[A.n] = (double)bd;
    X.Wt[p] = (double)HdiF;
}

// ------------------------------------------------------------------------------------------------ K4
struct SysArgs {
    const int* stop;                                        // early-exit flag of the resident loop (may be null)
    int N, n, ldg, ntile, P, use_lin_blocks, nsl, nsyrk;    // nsl point slices per tile, nsyrk = ntiles*nsl SYRK workgroups (0: row blocks only)
    int nt2;                                                // wide systems (k_ba_system<true>): SYRK workgroups own 2 x 2 SUPER-tiles, nt2 = ceil(ntile / 2), nsyrk = nt2 (nt2 + 1) / 2 * nsl
    double* part;                                           // SYRK partial tiles [tile][slice][256]
    const double* G; const double* Wt;
    const double* pbA; const double* pbL;                   // per-pair stitched blocks (ACTIVE / LINEARIZED)
    const double* cdelta; const double* cprior; const double* prior; const double* dprior;
    const double* HM; const double* bM;                     // may be null
    double lambda;
    double* HA; double* bA; double* HL; double* bL; double* Hsc; double* bsc;
    long long* dbg;
    double* Hb; double* bb;                                 // lambda-independent part of the final LM system: (HL+HM)+HA, (bL+bM)+bA
};

// Per-frame sums of the stitched pair blocks (the accumulation order of stitchDoubleTop, BA.cpp:1827-1843, per frame):
//   D_a = sum_t HH[a,t] + sum_h TT[h,a]   (8x8, diagonal block of frame a)
//   C_a = sum_t HC[a,t] + sum_h TC[h,a]   (8x4, frame-calibration coupling)      B_a = sum_t bH[a,t] + sum_h bT[h,a]
// Each tile workgroup builds the few it needs cooperatively in LDS (one summed value per task, loads pipelined).
#define FS_STRIDE 104     // 64 (D) + 32 (C) + 8 (B)
__device__ __forceinline__ double frame_sum(const double* pb, int N, int a, int v) {
    const int offH = v < 64 ? PB_HH + v : (v < 96 ? PB_HC + (v - 64) : PB_BH + (v - 96));
    const int offT = v < 64 ? PB_TT + v : (v < 96 ? PB_TC + (v - 64) : PB_BT + (v - 96));
    double s = 0;
    for (int base = 0; base < N; base += 8) {                // 16 loads in flight per trip (clamped, masked), added in pair order
        double h[8], t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = min(base + u, N - 1);
            h[u] = pb[(size_t)(a + k * N) * PB_STRIDE + offH];
            t[u] = pb[(size_t)(k + a * N) * PB_STRIDE + offT];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { const double mk = base + u < N ? 1.0 : 0.0; h[u] *= mk; t[u] *= mk; }
#pragma unroll
        for (int u = 0; u < 8; u++) s += h[u];
#pragma unroll
        for (int u = 0; u < 8; u++) s += t[u];
    }
    return s;
}

// k_ba_system, 256 threads per workgroup, two kinds of workgroup, no communication between them:
//   [0, ntiles*nsl)  SYRK: the 16x16 tile (ti <= tj) of H_sc = G^T diag(HdiF) [G | bdSum] over slice sl of the points, on the
//                    matrix cores (A[i][k] = G[p][16 ti + i], B[k][j] = w[p] G[p][16 tj + j]; D: col = lane&15,
//                    row = (lane>>4) + 4*reg) -> part[(tile*nsl + sl)*256 + e].  The slices are added in slice order by
//                    the consumer (k_ba_solve / k_ba_schur_out), so no workgroup pulls more than P/nsl rows through its CU
//                    and nothing depends on arrival order.
//   then N+1         row blocks of H_A, H_L (+ priors) and Hb = (H_L + H_M) + H_A (BA.cpp:1299), the lambda-independent part of
//                    the final system: workgroup 0 = calibration rows, workgroup 1+a = the 8 rows of frame a.
#define SYS_NW 8
// SYRK trip: 4 * SYS_U points per wave (3 * SYS_U loads in flight per lane, SYS_U MFMAs on two accumulators).  Small trips on
// purpose: at 60 VGPRs three 512-thread workgroups share a CU (6 waves per SIMD), so the 549 workgroups of a 20-frame window are
// resident at once and the loads of one wave hide under the matrix instructions of the others.  Measured (config E / B, us per
// iteration): SYS_U 16 at 1 workgroup per CU 162.6 / 52.0 (three dispatch rounds at E), 12 at 2: 159.4 / 54.1, 8 at 2: 156.8 / 52.9,
// 6 at 3: 157.4 / 52.0, 4 at 3: 155.5 / 51.7.
#define SYS_U 4
#define SYS_WPE 6
__device__ __forceinline__ int sys_tile_index(int ti, int tj, int ntile) { return ti * ntile - (ti * (ti - 1)) / 2 + (tj - ti); }

// SUPER (wide systems, ntile >= 8): a SYRK workgroup owns a 2 x 2 block of tiles — the two 16-column panels on either side are
// loaded once for four matrix products on four independent accumulators: half the reads of G per product (at a 20-frame window the 66
// plain tiles re-read 135 MB of G from beyond the L2s: the launch was bound by that, not by the matrix cores) and four chains in
// flight; operands of the next trip are requested before the current trip's products.  Partials land in the same per-tile slots.
template <bool SUPER>
__device__ __forceinline__ void k_ba_system_body(const SysArgs& S, const int bx_) {
    __shared__ double s_part[SYS_NW][256];
    __shared__ double s_f[2][FS_STRIDE];          // [ACTIVE | LINEARIZED] D (64) C (32) B (8) of this frame, or CC (16) bC (4)
    DBG_BLK(S.dbg, 2, 0);
    if (S.stop && *S.stop) return;
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, NT = 64 * SYS_NW;
    const int N = S.N, n = S.n;
    if (SUPER && (int)bx_ < S.nsyrk) {
        const int st = bx_ / S.nsl, sl = bx_ % S.nsl;
        int Ti = 0, rem = st;
        while (rem >= S.nt2 - Ti) { rem -= S.nt2 - Ti; Ti++; }
        const int Tj = Ti + rem;
        const int ti0 = 2 * Ti, ti1 = min(2 * Ti + 1, S.ntile - 1), tj0 = 2 * Tj, tj1 = min(2 * Tj + 1, S.ntile - 1);   // (clamped panels are loaded and dropped)
        const int kk = l >> 4, c = l & 15;
        const int per_s = ((S.P + S.nsl - 1) / S.nsl + 3) & ~3;
        const int s_beg = sl * per_s, s_end = min(S.P, s_beg + per_s);
        const int per = ((max(s_end - s_beg, 0) + SYS_NW - 1) / SYS_NW + 3) & ~3;
        const int p_beg = s_beg + wv * per, p_end = min(s_end, p_beg + per);
        double4_ acc[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};   // (ti0,tj0) (ti0,tj1) (ti1,tj0) (ti1,tj1)
        constexpr int U = 4;
        double a0[2][U], a1[2][U], b0[2][U], b1[2][U], w[2][U];
        auto request = [&](int sp, int buf) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int p = min(sp + 4 * u + kk, max(p_end - 1, 0));
                const double* row = S.G + (size_t)p * S.ldg;
                a0[buf][u] = row[16 * ti0 + c]; a1[buf][u] = row[16 * ti1 + c];
                b0[buf][u] = row[16 * tj0 + c]; b1[buf][u] = row[16 * tj1 + c];
                w[buf][u] = S.Wt[p];
            }
        };
        auto products = [&](int sp, int buf) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double mk = (sp + 4 * u + kk < p_end) ? 1.0 : 0.0;
                const double x0 = a0[buf][u] * mk, x1 = a1[buf][u] * mk;
                const double y0 = (w[buf][u] * b0[buf][u]) * mk, y1 = (w[buf][u] * b1[buf][u]) * mk;
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y1, acc[3], 0, 0, 0);
            }
        };
        if (p_beg < p_end) request(p_beg, 0);
        for (int sp = p_beg; sp < p_end; sp += 8 * U) {               // two trips per turn: static buffer indices
            if (sp + 4 * U < p_end) request(sp + 4 * U, 1);
            products(sp, 0);
            if (sp + 4 * U < p_end) {
                if (sp + 8 * U < p_end) request(sp + 8 * U, 0);
                products(sp + 4 * U, 1);
            }
        }
        // the four tiles, one after the other through the per-wave LDS rows; slots of tiles outside the upper triangle / the matrix are skipped
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ti = (q & 2) ? 2 * Ti + 1 : 2 * Ti, tj = (q & 1) ? 2 * Tj + 1 : 2 * Tj;
            const bool live = ti < S.ntile && tj < S.ntile && ti <= tj;          // wave-uniform
            if (live) {
#pragma unroll
                for (int rg = 0; rg < 4; rg++) s_part[wv][(kk + 4 * rg) * 16 + c] = acc[q][rg];
            }
            __syncthreads();
            if (live && tid < 256) {
                double sum = s_part[0][tid];
#pragma unroll
                for (int wq = 1; wq < SYS_NW; wq++) sum += s_part[wq][tid];
                S.part[((size_t)sys_tile_index(ti, tj, S.ntile) * S.nsl + sl) * 256 + tid] = sum;
            }
            __syncthreads();
        }
        DBG_BLK_END(S.dbg, 2);
        return;
    }
    if (!SUPER && (int)bx_ < S.nsyrk) {
        const int tile = bx_ / S.nsl, sl = bx_ % S.nsl;
        int ti = 0, rem = tile;
        while (rem >= S.ntile - ti) { rem -= S.ntile - ti; ti++; }
        const int tj = ti + rem;
        // every wave owns a contiguous point range, loads issued ahead of the MFMA chain
        const int kk = l >> 4, c = l & 15;
        const int per_s = ((S.P + S.nsl - 1) / S.nsl + 3) & ~3;             // points per slice, multiple of 4
        const int s_beg = sl * per_s, s_end = min(S.P, s_beg + per_s);
        const int per = ((max(s_end - s_beg, 0) + SYS_NW - 1) / SYS_NW + 3) & ~3;
        const int p_beg = s_beg + wv * per, p_end = min(s_end, p_beg + per);
        double4_ acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
        for (int sp = p_beg; sp < p_end; sp += 4 * SYS_U) {               // SYS_U MFMAs per trip on two independent accumulators
            // all 3 * SYS_U loads of the trip are issued before the first use (clamped rows, masks multiplied in afterwards)
            double a[SYS_U], b[SYS_U], w[SYS_U];
#pragma unroll
            for (int u = 0; u < SYS_U; u++) {
                const int p = min(sp + 4 * u + kk, p_end - 1);
                const double* row = S.G + (size_t)p * S.ldg;
                a[u] = row[16 * ti + c];
                b[u] = row[16 * tj + c];
                w[u] = S.Wt[p];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // one wait for the whole batch (measured: a staggered wait chain is ~1 us slower)
            if (S.dbg && tid == 0 && bx_ == 33) S.dbg[32] = wall_clock64();
#pragma unroll
            for (int u = 0; u < SYS_U; u++) {
                const double mk = (sp + 4 * u + kk < p_end) ? 1.0 : 0.0;
                a[u] *= mk;
                b[u] = (w[u] * b[u]) * mk;
            }
#pragma unroll
            for (int u = 0; u < SYS_U; u += 2) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u + 1], b[u + 1], acc2, 0, 0, 0);
            }
        }
#pragma unroll
        for (int rg = 0; rg < 4; rg++) acc[rg] += acc2[rg];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) s_part[wv][(kk + 4 * rg) * 16 + c] = acc[rg];
        if (S.dbg && tid == 0 && bx_ == 33) S.dbg[33] = wall_clock64();
        __syncthreads();
        if (S.dbg && tid == 0 && bx_ == 33) S.dbg[34] = wall_clock64();
        if (tid < 256) {
            double sum = s_part[0][tid];
#pragma unroll
            for (int w = 1; w < SYS_NW; w++) sum += s_part[w][tid];
            S.part[(size_t)bx_ * 256 + tid] = sum;
        }
        DBG_BLK_END(S.dbg, 2);
        return;
    }
    const int a = (int)bx_ - S.nsyrk - 1;                             // -1: calibration rows
    const int nmat = S.use_lin_blocks ? 2 : 1;
    if (a < 0) {
        // CC (16) + bC (4) sums over the N^2 pairs: 8 groups of pairs per entry, 8 loads in flight per trip, then a fixed-order add
        __shared__ double s_cpart[2][8][20];
        const int NN = N * N, per = (NN + 7) / 8;
        for (int task = tid; task < nmat * 160; task += NT) {
            const int v = task % 20, grp = (task / 20) % 8, mat = task / 160;
            const double* pb = mat == 0 ? S.pbA : S.pbL;
            const int off = v < 16 ? PB_CC + v : PB_BC + (v - 16);
            const int q0 = grp * per, q1 = min(NN, q0 + per);
            double sum = 0;
            for (int base = q0; base < q1; base += 8) {
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t[u] = pb[(size_t)min(base + u, q1 - 1) * PB_STRIDE + off];
#pragma unroll
                for (int u = 0; u < 8; u++) sum += t[u] * (base + u < q1 ? 1.0 : 0.0);
            }
            s_cpart[mat][grp][v] = sum;
        }
        __syncthreads();
        for (int task = tid; task < nmat * 20; task += NT) {
            const int v = task % 20, mat = task / 20;
            double sum = 0;
#pragma unroll
            for (int g8 = 0; g8 < 8; g8++) sum += s_cpart[mat][g8][v];
            s_f[mat][v] = sum;
        }
        __syncthreads();
        for (int e = tid; e < 4 * 5; e += NT) {
            const int r = e / 5, cq = e % 5;
            if (cq == 4) {                                                                  // right-hand sides, BA.cpp:1300
                const double ba = s_f[0][16 + r], bl = (S.use_lin_blocks ? s_f[1][16 + r] : 0.0) + S.cprior[r] * S.cdelta[r];
                S.bA[r] = ba; S.bL[r] = bl;
                S.bb[r] = (bl + (S.bM ? S.bM[r] : 0.0)) + ba;
            } else {
                const int lo = min(r, cq), hi = max(r, cq);                                 // lower half mirrors the upper one
                const double ha = s_f[0][lo * 4 + hi], hl = (S.use_lin_blocks ? s_f[1][lo * 4 + hi] : 0.0) + (r == cq ? S.cprior[r] : 0.0);
                S.HA[(size_t)r * n + cq] = ha; S.HL[(size_t)r * n + cq] = hl;
                S.Hb[(size_t)r * n + cq] = (hl + (S.HM ? S.HM[(size_t)r * n + cq] : 0.0)) + ha;
            }
        }
        DBG_BLK_END(S.dbg, 2);
        return;
    }
    for (int task = tid; task < nmat * FS_STRIDE; task += NT) {
        const int v = task % FS_STRIDE, mat = task / FS_STRIDE;
        s_f[mat][v] = frame_sum(mat == 0 ? S.pbA : S.pbL, N, a, v);
    }
    __syncthreads();
    for (int e = tid; e < 8 * (n + 1); e += NT) {
        const int i = e / (n + 1), cc = e % (n + 1), r = 4 + 8 * a + i;
        if (cc == n) {
            const double ba = s_f[0][96 + i], bl = (S.use_lin_blocks ? s_f[1][96 + i] : 0.0) + S.prior[8 * a + i] * S.dprior[8 * a + i];
            S.bA[r] = ba; S.bL[r] = bl;
            S.bb[r] = (bl + (S.bM ? S.bM[r] : 0.0)) + ba;
            continue;
        }
        double ha, hl;
        if (cc < 4) {                                                                       // frame-calibration coupling and its mirror (:1869)
            ha = s_f[0][64 + i * 4 + cc]; hl = S.use_lin_blocks ? s_f[1][64 + i * 4 + cc] : 0.0;
            S.HA[(size_t)cc * n + r] = ha; S.HL[(size_t)cc * n + r] = hl;
            S.Hb[(size_t)cc * n + r] = (hl + (S.HM ? S.HM[(size_t)cc * n + r] : 0.0)) + ha;
        } else {
            const int b = (cc - 4) >> 3, j = (cc - 4) & 7;
            if (a == b) {
                const int lo = min(i, j), hi = max(i, j);                                   // lower half mirrors the upper one
                ha = s_f[0][lo * 8 + hi]; hl = (S.use_lin_blocks ? s_f[1][lo * 8 + hi] : 0.0) + (i == j ? S.prior[8 * a + i] : 0.0);
            } else {                                                                        // symmetrisation of :1871-1875 (the sum commutes: (r,c) == (c,r) bitwise)
                const int lo = min(a, b), hi = max(a, b), ii = a < b ? i : j, jj = a < b ? j : i;
                const size_t q0 = (size_t)(lo + hi * N) * PB_STRIDE + PB_HT + ii * 8 + jj, q1 = (size_t)(hi + lo * N) * PB_STRIDE + PB_HT + jj * 8 + ii;
                ha = S.pbA[q0] + S.pbA[q1];
                hl = S.use_lin_blocks ? S.pbL[q0] + S.pbL[q1] : 0.0;
            }
        }
        S.HA[(size_t)r * n + cc] = ha; S.HL[(size_t)r * n + cc] = hl;
        S.Hb[(size_t)r * n + cc] = (hl + (S.HM ? S.HM[(size_t)r * n + cc] : 0.0)) + ha;
    }
    DBG_BLK_END(S.dbg, 2);
}
template <bool SUPER>
__global__ __launch_bounds__(64 * SYS_NW) __attribute__((amdgpu_waves_per_eu(SUPER ? 3 : SYS_WPE, SUPER ? 3 : SYS_WPE))) void k_ba_system(SysArgs S) { k_ba_system_body<SUPER>(S, blockIdx.x); }


// H_sc / b_sc for the host (statistics, tests): slices added in slice order.  Not on the iteration path.
__global__ __launch_bounds__(256) void k_ba_schur_out(SysArgs S) {
    const int tile = blockIdx.x, e = threadIdx.x, n = S.n;
    int ti = 0, rem = tile;
    while (rem >= S.ntile - ti) { rem -= S.ntile - ti; ti++; }
    const int tj = ti + rem;
    double hsc = 0;
    for (int k = 0; k < S.nsl; k++) hsc += S.part[((size_t)tile * S.nsl + k) * 256 + e];
    const int r = 16 * ti + (e >> 4), cc = 16 * tj + (e & 15);
    if (r >= n || cc > n) return;
    if (cc == n) { S.bsc[r] = hsc; return; }
    if (ti == tj && cc < r) return;
    S.Hsc[(size_t)r * n + cc] = hsc;
    if (r != cc) S.Hsc[(size_t)cc * n + r] = hsc;
}

// ------------------------------------------------------------------------------------------------ K5
// Workgroup 0: x = S LDLT(S Hf S)^-1 S bf on rows/cols [off, n) (BA.cpp:1312-1320).  Block-packed lower storage in LDS:
// block (I,J), J <= I, 16x16 doubles row-major at ((I(I+1)/2)+J)*256.  Right-looking, per block column K:
//   (1) wave 0 factors the diagonal block (16 sequential rank-1 steps, lanes own 4 entries each),
//   (2) one thread per row below solves L_IK = A_IK L_KK^-T D_K^-1 and keeps W_IK = L_IK D_K,
//   (3) trailing tiles A_IJ -= W_IK L_JK^T on the matrix cores (4 x v_mfma_f64_16x16x4_f64 per tile, tiles round-robin on waves).
// Workgroup 1 (blockIdx.x == 1): energy/census sums of the last residual pass + setNewFrameEnergyTH (see ba_linearize.hip).
#define SOLVE_THREADS 512
#define ORTHO_K 2                    // 64-column chunks of the nullspace basis prefetched per lane (covers 8N+4 <= 128)
#define BLD 17                      // leading dimension of a 16x16 LDS block (+1 double: conflict-free column access)
#define BSZ (16 * BLD)

__device__ __forceinline__ int blk_off(int I, int J) { return ((I * (I + 1)) / 2 + J) * BSZ; }
// broadcast of one lane's double (lane is wave-uniform): two v_readlane_b32, no LDS round trip
// 1/d: v_rcp_f64 seed + two Newton steps (error ~ eps); the IEEE division expansion is ~3x longer on the pivot chain
__device__ __forceinline__ double fast_rcp(double d) {
    double x = __builtin_amdgcn_rcp(d);
    x = x * (2.0 - d * x);
    x = x * (2.0 - d * x);
    return x;
}
// d^-1/2: v_rsq_f64 seed + two Newton steps (error ~ eps) — the IEEE sqrt + division expansions are ~60 instructions per call
__device__ __forceinline__ double fast_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = y * (1.5 - (0.5 * d) * (y * y));
    y = y * (1.5 - (0.5 * d) * (y * y));
    return y;
}
__device__ __forceinline__ double rl(double v, int lane) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}

struct SolveSys {            // the final LM system, assembled while it is loaded (BA.cpp:1299-1312)
    const double* Hb; const double* bb; const double* part; int nsl, ntile; double lambda;
    double* image;          // wide windows: the scaled system in the LDS layout, written by k_ba_assemble (null: assembled while loaded)
};
// H_sc(gi, gj), gj <= gi or the rhs column gi == n: slices added in slice order
template <int NSL>
__device__ __forceinline__ double schur_entry(const SolveSys& Y, int grow, int gcol) {     // grow <= gcol (upper tile storage)
    const int ti = grow >> 4, tj = gcol >> 4;
    const double* q = Y.part + ((size_t)sys_tile_index(ti, tj, Y.ntile) * Y.nsl) * 256 + (grow & 15) * 16 + (gcol & 15);
    double ps[NSL], sum = 0;
#pragma unroll
    for (int k = 0; k < NSL; k++) ps[k] = q[(size_t)k * 256];        // all loads in flight
#pragma unroll
    for (int k = 0; k < NSL; k++) sum += ps[k];
    return sum;
}

// SOLVE_IPT (lower block, element) items of the final system for one thread: item it0 + u*SOLVE_THREADS, u < SOLVE_IPT.
// Hf = Hb (diagonal * (1+lambda)) - H_sc / (1+lambda)  (:1306-1309); identity on the padding keeps the padded system SPD.
// Every load is unconditional (clamped address, mask multiplied in): a select would let the compiler sink each load
// behind its own branch + s_waitcnt, i.e. one memory round trip per load instead of one per thread.
#define SOLVE_IPT 8
template <int NSL, int IPT>
__device__ __forceinline__ void solve_load_items(const SolveSys& Y, int n, int off, int m, int items, int it0,
                                                 double (&val)[IPT], int (&dst)[IPT], int (&ij)[IPT]) {
    const double il = 1.0 / (1 + Y.lambda);
    double hb[IPT], ps[IPT][NSL];
    bool diag[IPT];
#pragma unroll
    for (int u = 0; u < IPT; u++) {
        const int it = it0 + u * SOLVE_THREADS;
        int I = 0, rem = it >> 8;
        while (rem >= I + 1) { rem -= I + 1; I++; }
        const int J = rem, i = 16 * I + (it & 15), j = 16 * J + ((it >> 4) & 15);      // i fastest: the slices are stored (j, i) row-major
        const bool live = it < items && j <= i;
        const bool real = live && i < m;                      // (j <= i < m)
        dst[u] = live ? blk_off(I, J) + (i & 15) * BLD + (j & 15) : -1;
        ij[u] = (i << 16) | j;
        diag[u] = real && i == j;
        const int gr = off + j, gc = off + i;                 // upper-tile storage of the Schur slices: row <= col
        const double* q = real ? Y.part + ((size_t)sys_tile_index(gr >> 4, gc >> 4, Y.ntile) * Y.nsl) * 256 + (gr & 15) * 16 + (gc & 15) : Y.part;
        const double hv = Y.Hb[real ? (size_t)gr * n + gc : 0];       // Hb is bitwise symmetric (k_ba_system mirrors it): (gr, gc) keeps the lanes on one row
        hb[u] = real ? hv : (i == j ? 1.0 : 0.0);
#pragma unroll
        for (int k = 0; k < NSL; k++) ps[u][k] = q[(size_t)k * 256] * (real ? 1.0 : 0.0);
    }
#pragma unroll
    for (int u = 0; u < IPT; u++) {
        double hsc = 0;
#pragma unroll
        for (int k = 0; k < NSL; k++) hsc += ps[u][k];
        double v = hb[u];
        if (diag[u]) v *= (1 + Y.lambda);
        val[u] = v - hsc * il;
    }
}

// the usual window sizes: one global round trip for the whole system, assembled while it is loaded; the values stay in registers
// across the Jacobi scaling SVecI = (diag + 10)^-1/2 (:1312).  IPT items per thread (5 covers 8N + 4 <= 64, 8 covers <= 80).
template <int NSL, int IPT>
__device__ __forceinline__ void solve_load_direct(const BAArgs& A, const SolveSys& Y, int n, int off, int m, int mp, int items, int tid,
                                              double* L, double* Sv, double* y) {
    double val[IPT]; int dst[IPT], ij[IPT];
    const int yi = SOLVE_THREADS - 1 - tid;                  // rhs on the last waves (fewest live matrix items), issued first
    const double yraw = (Y.bb[off + min(yi, m - 1)] - schur_entry<NSL>(Y, off + min(yi, m - 1), n)) * (yi < m ? 1.0 : 0.0);
    solve_load_items<NSL, IPT>(Y, n, off, m, items, tid, val, dst, ij);
    if (A.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); DBG_T(A, 54); }
#pragma unroll
    for (int u = 0; u < IPT; u++) {
        const int i = ij[u] >> 16, j = ij[u] & 0xffff;
        if (dst[u] >= 0 && i == j) Sv[i] = (i < m) ? fast_rsqrt(val[u] + 10.0) : 0.0;      // SVecI, :1312
    }
    __syncthreads();
    DBG_T(A, 55);
#pragma unroll
    for (int u = 0; u < IPT; u++) {
        const int i = ij[u] >> 16, j = ij[u] & 0xffff;
        if (dst[u] >= 0) L[dst[u]] = (i < m) ? (Sv[i] * val[u]) * Sv[j] : val[u];
    }
    if (yi < mp) y[yi] = (yi < m) ? Sv[yi] * yraw : 0.0;
}

// The same loader with 16-BYTE loads: a thread's item is a PAIR of vertically adjacent elements (i, i+1) of a lower block — adjacent
// addresses in Hb and in every Schur slice (i is the fast index of both; off = 4 and n = 8N + 4 are even, so a pair never straddles a
// tile and is 16-byte aligned).  Half the load instructions through the ONE CU that pulls the whole system (the load count, not the
// latency, is what that CU feels).  Same expressions per element as solve_load_items.
template <int NSL, int IPT2>
__device__ __forceinline__ void solve_load_direct2(const BAArgs& A, const SolveSys& Y, int n, int off, int m, int mp, int items2, int tid,
                                                   double* L, double* Sv, double* y) {
    const double il = 1.0 / (1 + Y.lambda);
    const int yi = SOLVE_THREADS - 1 - tid;                  // rhs on the last waves (fewest live matrix items), issued first
    const double yraw = (Y.bb[off + min(yi, m - 1)] - schur_entry<NSL>(Y, off + min(yi, m - 1), n)) * (yi < m ? 1.0 : 0.0);
    double2 hb[IPT2], ps[IPT2][NSL];
    int dst[IPT2], ij[IPT2];
    bool real0[IPT2], real1[IPT2], live0[IPT2], live1[IPT2];
#pragma unroll
    for (int u = 0; u < IPT2; u++) {
        const int it = tid + u * SOLVE_THREADS;
        int I = 0, rem = it >> 7;
        while (rem >= I + 1) { rem -= I + 1; I++; }
        const int J = rem, i = 16 * I + 2 * (it & 7), j = 16 * J + ((it >> 3) & 15);
        const bool in = it < items2;
        live0[u] = in && j <= i; live1[u] = in && j <= i + 1;
        real0[u] = live0[u] && i < m; real1[u] = live1[u] && i + 1 < m;
        const bool any = real0[u] || real1[u];
        dst[u] = blk_off(I, J) + (i & 15) * BLD + (j & 15);
        ij[u] = (i << 16) | j;
        const int gr = off + j, gc = off + i;                 // upper-tile storage of the Schur slices: row <= col
        const double* q = any ? Y.part + ((size_t)sys_tile_index(gr >> 4, gc >> 4, Y.ntile) * Y.nsl) * 256 + (gr & 15) * 16 + (gc & 15) : Y.part;
        hb[u] = *reinterpret_cast<const double2*>(Y.Hb + (any ? (size_t)gr * n + gc : 0));
#pragma unroll
        for (int k = 0; k < NSL; k++) ps[u][k] = *reinterpret_cast<const double2*>(q + (size_t)k * 256);
    }
    if (A.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); DBG_T(A, 54); }
    double v0[IPT2], v1[IPT2];
#pragma unroll
    for (int u = 0; u < IPT2; u++) {
        const int i = ij[u] >> 16, j = ij[u] & 0xffff;
        double h0 = 0, h1 = 0;
#pragma unroll
        for (int k = 0; k < NSL; k++) { h0 += ps[u][k].x * (real0[u] ? 1.0 : 0.0); h1 += ps[u][k].y * (real1[u] ? 1.0 : 0.0); }
        double a0 = real0[u] ? hb[u].x : (i == j ? 1.0 : 0.0), a1 = real1[u] ? hb[u].y : (i + 1 == j ? 1.0 : 0.0);
        if (real0[u] && i == j) a0 *= (1 + Y.lambda);
        if (real1[u] && i + 1 == j) a1 *= (1 + Y.lambda);
        v0[u] = a0 - h0 * il; v1[u] = a1 - h1 * il;
        if (live0[u] && i == j) Sv[i] = (i < m) ? fast_rsqrt(v0[u] + 10.0) : 0.0;                 // SVecI, :1312
        if (live1[u] && i + 1 == j) Sv[i + 1] = (i + 1 < m) ? fast_rsqrt(v1[u] + 10.0) : 0.0;
    }
    __syncthreads();
    DBG_T(A, 55);
#pragma unroll
    for (int u = 0; u < IPT2; u++) {
        const int i = ij[u] >> 16, j = ij[u] & 0xffff;
        if (live0[u]) L[dst[u]] = (i < m) ? (Sv[i] * v0[u]) * Sv[j] : v0[u];
        if (live1[u]) L[dst[u] + BLD] = (i + 1 < m) ? (Sv[i + 1] * v1[u]) * Sv[j] : v1[u];
    }
    if (yi < mp) y[yi] = (yi < m) ? Sv[yi] * yraw : 0.0;
}

// Wide windows (more than SOLVE_IPT items per solve thread): the final system is assembled and Jacobi-scaled by one workgroup per
// lower 16x16 block — all CUs pull the (1 + nsl) sources, the solve workgroup then copies ONE image, already in its LDS layout
// (blocks | SVecI | scaled rhs).  Same expressions, in the same order, as solve_load_items + the scaling in k_ba_solve.
template <int NSL>
__device__ __forceinline__ double final_entry(const SolveSys& Y, int n, int off, int i, int j, double il) {     // j <= i < m
    const int gr = off + j, gc = off + i;
    double v = Y.Hb[(size_t)gr * n + gc];
    const double hsc = schur_entry<NSL>(Y, gr, gc);
    if (i == j) v *= (1 + Y.lambda);
    return v - hsc * il;
}
template <int NSL>
__global__ __launch_bounds__(256) void k_ba_assemble(int n, int off, SolveSys Y, const int* __restrict__ stop) {
    if (stop && *stop) return;
    const int m = n - off, nb = (m + 15) / 16, mp = nb * 16, e = threadIdx.x;
    int I = 0, rem = blockIdx.x;
    while (rem >= I + 1) { rem -= I + 1; I++; }
    const int J = rem, i = 16 * I + (e & 15), j = 16 * J + (e >> 4);             // i fastest: the slices are stored (j, i) row-major
    const double il = 1.0 / (1 + Y.lambda);
    const bool real = i < m && j <= i;
    const int ic = min(i, m - 1), jc = min(j, ic);
    const double val = final_entry<NSL>(Y, n, off, ic, jc, il);
    const double dii = final_entry<NSL>(Y, n, off, ic, ic, il), djj = final_entry<NSL>(Y, n, off, jc, jc, il);
    const double si = fast_rsqrt(dii + 10.0), sj = fast_rsqrt(djj + 10.0);      // SVecI, BA.cpp:1312
    Y.image[blk_off(I, J) + (i & 15) * BLD + (j & 15)] = real ? (si * val) * sj : (i == j ? 1.0 : 0.0);
    if (I == J && e < 16) {                                                       // the diagonal blocks also leave SVecI and the scaled rhs
        double* sv = Y.image + (size_t)(nb * (nb + 1) / 2) * BSZ;
        const double yraw = Y.bb[off + ic] - schur_entry<NSL>(Y, off + ic, n);
        sv[i] = i < m ? si : 0.0;
        sv[mp + i] = i < m ? si * yraw : 0.0;
    }
}

// K6 inside the K5 launch (round 3): the back-substitution's point workgroups and its frame-step workgroup ride in the solve launch as
// further blocks, request everything that does not depend on x while the factorisation runs and continue when the solve workgroup
// publishes x — one launch, its gap and the head of K6 less per iteration.  g_pts = 0: not merged.
struct BacksubCall {
    const double* adH; const double* adT; float* step_partial; FrameStepArgs F;
    int g_pts;                 // point blocks of 512 threads (two virtual 256-thread blocks each)
    int* xticket; int ticket;  // published by the solve workgroup behind x
    unsigned long long* xpub;  // x as the waiting workgroups read it: two self-validating words per entry, {low half | ticket << 32}, {high half | ticket << 32}
};
__device__ __forceinline__ bool wait_and_fetch_x(const int* xticket, int ticket, const unsigned long long* xpub, int n, double* __restrict__ s_x, int nthreads);
template <int NT, bool INLAUNCH>
__device__ __forceinline__ void k_ba_backsub_body(const BAArgs& A, const double* __restrict__ adH, const double* __restrict__ adT,
                                                  const double* __restrict__ x_in, LinSummary* __restrict__ sum, float* __restrict__ step_partial,
                                                  int do_step, const FrameStepArgs& F, const double* __restrict__ xad, const int bx_, const int gx_,
                                                  double* __restrict__ s_xAd, float* __restrict__ s_redf, const int* xticket, const int ticket,
                                                  const unsigned long long* xpub);


// Factorisation with LOOK-AHEAD for wide systems (nb >= SOLVE_LOOKAHEAD_NB block columns).  The plain loop below
// runs elimination(K) | barrier | all trailing tiles of step K | barrier: at 20 keyframes 45 tiles on 8 waves are six rounds per step
// with the elimination wave(s) waiting behind them (12.7 us of the 48).  Here step K first updates the tiles of block column K + 1 only;
// then the waves that carry rows of column K + 1 eliminate it WHILE the other waves — and the eliminating ones when they are done —
// draw the remaining tiles of step K from a counter in LDS.  The trailing operand W_IK = L_IK D_K is formed from the factor and D
// when it is read (the plain loop keeps the pre-scaling values in a second panel, for which there is no room beside a double
// buffer at 160 unknowns): same factorisation, last-bit differences against the plain loop, same 1e-7 bar against the oracle.
#define SOLVE_LOOKAHEAD_NB 6
// Column updates behind a pivot, row[j] -= cid * broadcast(row[k], lane j) for j = k+2 .. 15: the broadcasts of FOUR columns first
// (an empty asm that takes the four values as scalar operands keeps them live together, in four distinct scalar pairs), then their four
// multiply-adds.  Left alone the compiler pairs every v_readlane pair with its v_fma_f64 on ONE reused scalar pair: tools/microbench7.hip
// measures 23.7 cycles per column update that way, 21.1 in groups of four, 17.1 with all fourteen broadcasts first (which the kernel's
// scalar registers do not allow); in the kernel the elimination of a block column drops from 2.08 to 1.96 us whatever the group size
// (2 .. 6) — it is issue-bound: two readlanes at ~5 cycles and an fp64 multiply-add at ~7 per column, ~65 cycles of chain per pivot.
#define SOLVE_COLUMN_UPDATES(row, cid, k) do { \
        double bq_[16]; \
        _Pragma("unroll") for (int j_ = (k) + 2; j_ < 16; j_++) bq_[j_] = rl(row[(k)], j_); \
        _Pragma("unroll") for (int j0_ = (k) + 2; j0_ < 16; j0_ += 4) { \
            if (16 - j0_ >= 4) asm volatile("" : "+s"(bq_[j0_]), "+s"(bq_[j0_ + 1]), "+s"(bq_[j0_ + 2]), "+s"(bq_[j0_ + 3])); \
            else if (16 - j0_ == 3) asm volatile("" : "+s"(bq_[j0_]), "+s"(bq_[j0_ + 1]), "+s"(bq_[j0_ + 2])); \
            else if (16 - j0_ == 2) asm volatile("" : "+s"(bq_[j0_]), "+s"(bq_[j0_ + 1])); \
        } \
        _Pragma("unroll") for (int j_ = (k) + 2; j_ < 16; j_++) row[j_] -= (cid) * bq_[j_]; } while (0)
__device__ __forceinline__ void solve_eliminate_column_la(double* __restrict__ L, double* __restrict__ y, double* __restrict__ dvec,
                                                          double* __restrict__ dinv, const int K, const int nb, const int wv, const int l) {
    const int nbelow = (nb - K - 1) * 16;
    const int gi = (l < 16) ? 16 * K + l : 16 * (K + 1) + wv * 48 + (l - 16);
    const bool prow = l >= 16 && (wv * 48 + (l - 16)) < nbelow;
    const bool have = l < 16 || prow;
    double* Rrow = L + blk_off(have ? gi >> 4 : K, K) + (gi & 15) * BLD;
    double row[16];
#pragma unroll
    for (int j = 0; j < 16; j++) row[j] = have ? Rrow[j] : 0.0;
    double yv = have ? y[gi] : 0.0, mydk = 0.0;
    double dk, cid;                                         // (the pivot chain of the plain loop, statement for statement)
#define SOLVE_PIVOT_CHAIN_LA(K_, DK_, CID_) do { \
        union { double d; int i[2]; } ud_; \
        ud_.d = row[K_]; \
        const int dlo_ = __builtin_amdgcn_readlane(ud_.i[0], K_), dhi_ = __builtin_amdgcn_readlane(ud_.i[1], K_); \
        const bool tiny_ = (dhi_ & 0x7ff00000) == 0; \
        ud_.i[0] = dlo_; ud_.i[1] = dhi_; \
        DK_ = ud_.d; \
        const double x0_ = __builtin_amdgcn_rcp(ud_.d); \
        const double e0_ = __builtin_fma(-ud_.d, x0_, 1.0), p_ = row[K_] * x0_; \
        const double t_ = __builtin_fma(p_, e0_, p_), e2_ = e0_ * e0_; \
        const double c_ = __builtin_fma(t_, e2_, t_); \
        CID_ = tiny_ ? 0.0 : c_; } while (0)
    SOLVE_PIVOT_CHAIN_LA(0, dk, cid);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const double zk = rl(yv, k);
        double dk_next = 0.0, cid_next = 0.0;
        if (k + 1 < 16) {
            row[k + 1] -= cid * rl(row[k], k + 1);
            SOLVE_PIVOT_CHAIN_LA(k + 1, dk_next, cid_next);
        }
        SOLVE_COLUMN_UPDATES(row, cid, k);
        row[k] = cid;
        if (l > k) yv -= cid * zk;
        if (l == k) mydk = dk;
        dk = dk_next; cid = cid_next;
    }
#undef SOLVE_PIVOT_CHAIN_LA
    const double mydi = fabs(mydk) > 2.2250738585072014e-308 ? fast_rcp(mydk) : 0.0;
    if (have && (l >= 16 || wv == 0)) {
#pragma unroll
        for (int j = 0; j < 16; j++) Rrow[j] = row[j];
        y[gi] = yv;
    }
    if (wv == 0 && l < 16) { dvec[16 * K + l] = mydk; dinv[16 * K + l] = mydi; }
}
// A_IJ -= (L_IK D_K) L_JK^T on the matrix cores, one tile per call
__device__ __forceinline__ void solve_update_tile_la(double* __restrict__ L, const double* __restrict__ dvec, const int K, const int I, const int J, const int l) {
    double* C = L + blk_off(I, J);
    const double* Li = L + blk_off(I, K);
    const double* Lj = L + blk_off(J, K);
    const int kq = l >> 4, cidx = l & 15;
    double4_ acc;
#pragma unroll
    for (int rg = 0; rg < 4; rg++) acc[rg] = C[(kq + 4 * rg) * BLD + cidx];
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
        const double av = -(Li[cidx * BLD + 4 * s4 + kq] * dvec[16 * K + 4 * s4 + kq]);
        const double bv = Lj[cidx * BLD + 4 * s4 + kq];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rg = 0; rg < 4; rg++) C[(kq + 4 * rg) * BLD + cidx] = acc[rg];
}
__device__ __forceinline__ void solve_factor_lookahead(double* __restrict__ L, int* __restrict__ s_q, double* __restrict__ y, double* __restrict__ dvec,
                                                       double* __restrict__ dinv, const int nb, const int tid) {
    const int wv = tid >> 6, l = tid & 63, NWV = SOLVE_THREADS / 64;
    if (wv * 48 < (nb - 1) * 16 || wv == 0) solve_eliminate_column_la(L, y, dvec, dinv, 0, nb, wv, l);
    __syncthreads();
    for (int K = 0; K + 1 < nb; K++) {
        // (a) the tiles of block column K + 1: (I, K + 1), I = K + 1 .. nb - 1
        for (int I = K + 1 + wv; I < nb; I += NWV) solve_update_tile_la(L, dvec, K, I, K + 1, l);
        if (tid == 0) *s_q = 0;
        __syncthreads();
        // (b) column K + 1 is eliminated by the waves that carry its rows; every wave then draws the remaining tiles (I, J), K + 2 <= J <= I, of step K
        const int nbelow1 = (nb - K - 2) * 16;
        if (wv * 48 < nbelow1 || wv == 0) solve_eliminate_column_la(L, y, dvec, dinv, K + 1, nb, wv, l);
        const int nt = nb - K - 2, ntiles = nt * (nt + 1) / 2;
        for (;;) {
            int tix = 0;
            if (l == 0) tix = atomicAdd(s_q, 1);
            tix = __builtin_amdgcn_readfirstlane(tix);
            if (tix >= ntiles) break;
            int a = 0, rem = tix;
            while (rem >= a + 1) { rem -= a + 1; a++; }
            solve_update_tile_la(L, dvec, K, K + 2 + a, K + 2 + rem, l);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void publish_x_entry(const BacksubCall& BC, const int i, const double v) {
    const unsigned long long tk = (unsigned long long)(unsigned)BC.ticket << 32;
    __hip_atomic_store(BC.xpub + 2 * i, tk | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(BC.xpub + 2 * i + 1, tk | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int NSL, bool WIDE_OK = true, bool HYBRID = false, bool MERGE = false>
__device__ __forceinline__ void k_ba_solve_body(const BAArgs& A, int n, int off, const SolveSys& Y, double* __restrict__ x, int* __restrict__ flag,
                                                const int* newframe_res, int n_newframe, const double* lin_partial,
                                                int n_partial, LinSummary* lin_out, FrameDev* frames_rw, int do_finish,
                                                const double* __restrict__ nullU, const double* __restrict__ indirect_x, const ReprojArgs& RP, const BacksubCall& BC, const int bx_) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x;
    DBG_BLK(A.dbg, 3, 0);
    if (A.ctl && A.ctl->stop) return;
    if (bx_ == 1) {
        if (do_finish) lin_finish_block(A, newframe_res, n_newframe, lin_partial, n_partial, lin_out, frames_rw,
                                        reinterpret_cast<unsigned*>(sm), sm + 2048);
        DBG_BLK_END(A.dbg, 3);
        return;
    }
    if (bx_ >= 2) {
        // hybrid ORB term (config C): the per-frame workgroups of addIndirectToProblem run BESIDE the factorisation, in its launch — they
        // depend only on the frame states of the previous iteration; workgroup 0 picks their solutions up at its tail (tickets below)
        static_assert(RP_THREADS == SOLVE_THREADS, "the frame workgroups of the hybrid term run in the solve launch");
        const int nrp = HYBRID ? RP.N : 0;
        if constexpr (HYBRID) { if (bx_ - 2 < nrp) { reproj_frame_block(RP, bx_ - 2, sm); return; } }      // (its own instantiation: the term's local arrays give the kernel a scratch frame)
        if constexpr (MERGE) {
            const int b6 = bx_ - 2 - nrp;                     // block of the back-substitution: [frame step] [point blocks ...]
            if (BC.g_pts > 0 && b6 >= 0) {
                if (BC.F.on && b6 == 0) {
                    FrameStepPre FP;
                    frame_step_prefetch(BC.F, FP);                  // (ahead of the wait: see FrameStepPre)
                    if (wait_and_fetch_x(BC.xticket, BC.ticket, BC.xpub, n, sm, SOLVE_THREADS)) frame_step_block(BC.F, sm, FP, A.dbg ? A.dbg + 104 : nullptr);
                    else if (tid == 0) atomicAdd(&lin_out->nonfinite, 1);
                    return;
                }
                const int pb = b6 - (BC.F.on ? 1 : 0);
                if (pb < BC.g_pts) {
                    float* red = reinterpret_cast<float*>(sm + A.N * A.N * 8 + ((n + 1) & ~1));
                    k_ba_backsub_body<SOLVE_THREADS, true>(A, BC.adH, BC.adT, x, lin_out, BC.step_partial, 1, BC.F, nullptr, pb, BC.g_pts, sm, red, BC.xticket, BC.ticket, BC.xpub);
                }
            }
        }
        return;
    }
    const int m = n - off, nb = (m + 15) / 16, mp = nb * 16;
    DBG_T(A, 48);
    double u_dot[ORTHO_K], u_upd[7];                            // nullspace basis entries this thread will need at the very end
#pragma unroll
    for (int k = 0; k < ORTHO_K; k++) u_dot[k] = 0.0;
#pragma unroll
    for (int e = 0; e < 7; e++) u_upd[e] = 0.0;
    if (nullU) {
        const int w7 = min(tid >> 6, 6), ln = tid & 63;
#pragma unroll
        for (int k = 0; k < ORTHO_K; k++) { const int i = ln + 64 * k; u_dot[k] = nullU[(size_t)w7 * n + min(i, n - 1)] * (i < n ? 1.0 : 0.0); }
#pragma unroll
        for (int e = 0; e < 7; e++) u_upd[e] = nullU[(size_t)e * n + min(tid, n - 1)];
    }
    if (tid == 0) lin_out->nonfinite = 0;            // consumed by the back-substitution launch that follows
    double* L = sm;                                  // nb(nb+1)/2 blocks
    double* Wk = L + (size_t)(nb * (nb + 1) / 2) * BSZ;   // panel W_IK: nb blocks
    double* Sv = Wk + (size_t)nb * BSZ;              // mp
    double* y = Sv + mp;                             // mp
    double* dvec = y + mp;                           // mp (D)
    double* dinv = dvec + mp;                        // mp (1/D)
    // one global round trip for the whole system, then the Jacobi scaling SVecI = (diag + 10)^-1/2 (:1312)
    const int items = (nb * (nb + 1) / 2) * 256;                 // (lower block, element) pairs
    if (items <= 6 * SOLVE_THREADS) solve_load_direct2<NSL, 3>(A, Y, n, off, m, mp, items / 2, tid, L, Sv, y);
    else if (items <= SOLVE_IPT * SOLVE_THREADS) solve_load_direct2<NSL, SOLVE_IPT / 2>(A, Y, n, off, m, mp, items / 2, tid, L, Sv, y);
    else if constexpr (WIDE_OK) {          // (the batched instantiation refuses wide windows on the host: without this path it has no scratch frame)
        // wide windows: k_ba_assemble left the scaled system in this layout — one flat copy, 16 bytes per lane, every load in flight at once
        const int nd2 = ((nb * (nb + 1) / 2) * BSZ) / 2;                              // BSZ is even
        const double2* src = reinterpret_cast<const double2*>(Y.image);
        double2* dst2 = reinterpret_cast<double2*>(L);
        for (int base = 0; base < nd2; base += 16 * SOLVE_THREADS) {
            double2 v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = src[min(base + u * SOLVE_THREADS + tid, nd2 - 1)];
            // (no "loads landed" stamp here: behind an asm with a memory clobber the compiler keeps v[] in a scratch frame — 16 scratch
            //  round trips per thread on this path, and a kernel with a scratch frame pays for it at every dispatch, small windows too)
#pragma unroll
            for (int u = 0; u < 16; u++) if (base + u * SOLVE_THREADS + tid < nd2) dst2[base + u * SOLVE_THREADS + tid] = v[u];
        }
        if (tid < mp) { Sv[tid] = Y.image[2 * (size_t)nd2 + tid]; y[tid] = Y.image[2 * (size_t)nd2 + mp + tid]; }
    }
    __syncthreads();
    DBG_T(A, 49);
    const int wv = tid >> 6, l = tid & 63, NWV = SOLVE_THREADS / 64;
    bool factored = false;
    if constexpr (WIDE_OK && !HYBRID && !MERGE) {
        if (nb >= SOLVE_LOOKAHEAD_NB && !A.no_lookahead) { solve_factor_lookahead(L, reinterpret_cast<int*>(Wk), y, dvec, dinv, nb, tid); factored = true; }
    }
    for (int K = 0; K < nb && !factored; K++) {
        DBG_T(A, 64 + 2 * K);
        // (1) block column K, ONE ROW PER LANE in registers: lanes 0-15 carry the diagonal block, lanes 16-63 up to 48 rows
        //     of the panel below it (further waves repeat the diagonal rows and take the next 48 panel rows).  The 16
        //     elimination steps of the diagonal block — pivots and column entries travel by v_readlane from lanes 0-15, no
        //     LDS latency on the chain — update the panel rows in the same instructions, and the right-hand side rides
        //     along as a 17th column (forward substitution L z = y).  W_IK = L_IK D is the pre-scaling value of each entry.
        const int nbelow = (nb - K - 1) * 16;
        const int gi = (l < 16) ? 16 * K + l : 16 * (K + 1) + wv * 48 + (l - 16);
        const bool prow = l >= 16 && (wv * 48 + (l - 16)) < nbelow;                 // a live panel row
        const bool active = wv * 48 < nbelow || wv == 0;                             // wave-uniform
        if (active) {
            const bool have = l < 16 || prow;
            double* Rrow = L + blk_off(have ? gi >> 4 : K, K) + (gi & 15) * BLD;
            double row[16], w[16];
#pragma unroll
            for (int j = 0; j < 16; j++) row[j] = have ? Rrow[j] : 0.0;
            double yv = have ? y[gi] : 0.0, mydk = 0.0;
            // The pivot chain, per step: readlane (d_k) -> v_rcp_f64 -> {e = 1 - d x0, p = a_ik x0} -> {t = p + p e, e^2} ->
            // l_ik = t + t e^2 -> the update of column k+1 -> readlane.  x0 (1 + e)(1 + e^2) is two Newton steps on the seed, folded
            // into the product with a_ik: three dependent operations after the seed instead of five plus a multiply.  The loop is
            // written software-pipelined — column k+1 first, then the NEXT pivot's multipliers, then the other columns — so that
            // the chain does not wait behind the 14 independent updates.
            // A zero pivot of a positive SEMI-definite system (frames without any good residual and without a pose prior) has a
            // zero column below it and is skipped — Eigen's pivoted LDLT stops at a zero corner (LDLT.h:300-396) and pseudo-inverts
            // D (:580-587), D^-1 below maps it to x = 0: the exponent test runs on the uniform (scalar) copy of d_k BESIDE the chain and
            // one select zeroes the multipliers at its end (tools/microbench6.hip: with the guard between the readlanes and the
            // reciprocal a bare step is 126 cycles, without 86).
            double dk, cid;
#define SOLVE_PIVOT_CHAIN(K_, DK_, CID_) do { \
                union { double d; int i[2]; } ud_; \
                ud_.d = row[K_]; \
                const int dlo_ = __builtin_amdgcn_readlane(ud_.i[0], K_), dhi_ = __builtin_amdgcn_readlane(ud_.i[1], K_); \
                const bool tiny_ = (dhi_ & 0x7ff00000) == 0;        /* scalar side, BESIDE the chain (round 3: it sat on it, +40 cycles per pivot) */ \
                ud_.i[0] = dlo_; ud_.i[1] = dhi_; \
                DK_ = ud_.d; \
                const double x0_ = __builtin_amdgcn_rcp(ud_.d); \
                const double e0_ = __builtin_fma(-ud_.d, x0_, 1.0), p_ = row[K_] * x0_; \
                const double t_ = __builtin_fma(p_, e0_, p_), e2_ = e0_ * e0_; \
                const double c_ = __builtin_fma(t_, e2_, t_); \
                CID_ = tiny_ ? 0.0 : c_; } while (0)
            SOLVE_PIVOT_CHAIN(0, dk, cid);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const double zk = rl(yv, k);
                double dk_next = 0.0, cid_next = 0.0;
                w[k] = row[k];
                if (k + 1 < 16) {
                    row[k + 1] -= cid * rl(row[k], k + 1);
                    SOLVE_PIVOT_CHAIN(k + 1, dk_next, cid_next);
                }
                SOLVE_COLUMN_UPDATES(row, cid, k);                                 // lanes above the pivot compute unused values
                row[k] = cid;                  // every lane: the entries on and above the diagonal of the factor are never read (D lives in dvec)
                if (l > k) yv -= cid * zk;
                if (l == k) mydk = dk;
                dk = dk_next; cid = cid_next;
            }
#undef SOLVE_PIVOT_CHAIN
            const double mydi = fabs(mydk) > 2.2250738585072014e-308 ? fast_rcp(mydk) : 0.0;
            if (have && (l >= 16 || wv == 0)) {
#pragma unroll
                for (int j = 0; j < 16; j++) Rrow[j] = row[j];
                y[gi] = yv;
            }
            if (prow) {
                double* Wrow = Wk + (size_t)((gi >> 4) - K - 1) * BSZ + (gi & 15) * BLD;
#pragma unroll
                for (int j = 0; j < 16; j++) Wrow[j] = w[j];
            }
            if (wv == 0 && l < 16) { dvec[16 * K + l] = mydk; dinv[16 * K + l] = mydi; }
        }
        __syncthreads();
        DBG_T(A, 65 + 2 * K);
        // (2) trailing update on the matrix cores: A_IJ -= W_IK L_JK^T, K < J <= I
        const int nt = nb - K - 1, ntiles = nt * (nt + 1) / 2;
        for (int tix = wv; tix < ntiles; tix += NWV) {
            int a = 0, rem = tix;
            while (rem >= a + 1) { rem -= a + 1; a++; }       // tile (a, rem), rem <= a
            const int I = K + 1 + a, J = K + 1 + rem;
            double* C = L + blk_off(I, J);
            const double* W = Wk + (size_t)a * BSZ;
            const double* Lj = L + blk_off(J, K);
            const int kq = l >> 4, cidx = l & 15;
            double4_ acc;
#pragma unroll
            for (int rg = 0; rg < 4; rg++) acc[rg] = C[(kq + 4 * rg) * BLD + cidx];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const double av = -W[cidx * BLD + 4 * s + kq];           // A[i = l&15][k]
                const double bv = Lj[cidx * BLD + 4 * s + kq];           // B[k][j = l&15] = L_JK[j][k]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int rg = 0; rg < 4; rg++) C[(kq + 4 * rg) * BLD + cidx] = acc[rg];
        }
        __syncthreads();
    }
    DBG_T(A, 50);
    DBG_T(A, 51);
    if (mp <= 64) {
        // up to 64 unknowns: D^-1 and L^T x = z on ONE wave, a row per lane, no barrier inside — step k broadcasts x_k (readlane) and
        // every row above it subtracts L(k, row) x_k; the 16 factor entries of a block row are fetched ahead of its 16 steps.
        // (Measured: 2.4 us for 64 steps against 2.7 us blocked over the workgroup; broadcasting inside the block with DPP moves
        // instead — 3.2 us — is slower, the dependent fp64 multiply-add itself is the step.)
        if (wv == 0) {
            const bool in = l < mp;
            const double d = in ? dvec[l] : 0.0;
            double yv = (in && fabs(d) > 2.2250738585072014e-308) ? y[l] * dinv[l] : 0.0;
            for (int kb = nb - 1; kb >= 0; kb--) {
                const double* Lrow = L + blk_off(kb, min(l >> 4, kb)) + (l & 15);
                double col[16];
#pragma unroll
                for (int kk = 0; kk < 16; kk++) { const double c_ = Lrow[kk * BLD]; col[kk] = (l < 16 * kb + kk) ? c_ : 0.0; }   // unconditional loads (valid addresses), select afterwards: no branch per load
#pragma unroll
                for (int kk = 15; kk >= 0; kk--) yv -= col[kk] * rl(yv, 16 * kb + kk);
            }
            if (in) y[l] = yv;
        }
        __syncthreads();
    } else {
        for (int i = tid; i < mp; i += SOLVE_THREADS) {
            const double d = dvec[i];
            y[i] = (fabs(d) > 2.2250738585072014e-308) ? y[i] * dinv[i] : 0.0;
        }
        __syncthreads();
        for (int K = nb - 1; K >= 0; K--) {
            if (wv == 0) {                                       // L_KK^T x = z: COLUMN per lane in registers
                const double* D = L + blk_off(K, K);
                double col[16];
#pragma unroll
                for (int k = 0; k < 16; k++) col[k] = (l < 16) ? D[k * BLD + l] : 0.0;
                double yv = (l < 16) ? y[16 * K + l] : 0.0;
#pragma unroll
                for (int k = 15; k >= 0; k--) {
                    const double xk = rl(yv, k);
                    if (l < k) yv -= col[k] * xk;
                }
                if (l < 16) y[16 * K + l] = yv;
            }
            __syncthreads();
            for (int rr = tid; rr < K * 16; rr += SOLVE_THREADS) {      // rows of blocks J < K: y_J -= L_KJ^T x_K
                const int J = rr >> 4, j = rr & 15;
                const double* Lb = L + blk_off(K, J);
                double s = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) s += Lb[i * BLD + j] * y[16 * K + i];
                y[16 * J + j] -= s;
            }
            __syncthreads();
        }
    }
    DBG_T(A, 52);
    int bad = 0;
    double* xs = Wk;                                    // n: the solution in the caller's scaling
    double* dots = Wk + mp + 16;                        // 7
    for (int i = tid; i < n; i += SOLVE_THREADS) xs[i] = (i < off) ? 0.0 : Sv[i - off] * y[i - off];
    __syncthreads();
    if (indirect_x) {
        // hybrid ORB term, addIndirectToProblem (BA.cpp:2702-2727): unless the indirect solution has a non-finite entry, the pose part
        // of x becomes x * directRatio + indirectX * indirectRatio with the literal constants numIndirectPoint = 1, numDirectPoint = 0
        int* ind_flag = reinterpret_cast<int*>(dots + 8);     // (dynamic LDS: the kernel may take the whole 160 KB, a static variable —
        if (tid == 0) *ind_flag = 0;                          //  or __syncthreads_or's hidden one — would make that request invalid)
        __syncthreads();
        const bool same_launch = RP.ready != nullptr;        // produced by workgroups 2.. of THIS launch: wait for every frame's ticket
        if (same_launch) {
            if (tid < A.N) {
                int spins = 0;
                while (__hip_atomic_load(RP.ready + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != RP.ticket) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1 << 22)) { *ind_flag = 2; break; }      // never spin forever: reported as a failed solve
                }
            }                                            // (the solutions are then read with device-scope loads: no acquire fence)
            __syncthreads();
        }
        int mybad = 0;
        for (int i = tid; i < 6 * A.N; i += SOLVE_THREADS) {
            const double v = same_launch ? __hip_atomic_load(indirect_x + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : indirect_x[i];
            mybad |= !isfinite(v);
        }
        if (mybad && *ind_flag == 0) *ind_flag = 1;
        __syncthreads();
        const int ind_bad = *ind_flag;
        if (ind_bad == 2) bad = 1;
        if (!ind_bad) {
            const double indirectRatio = 1.0 / (1.0 + 0.0), directRatio = 1.0 - indirectRatio;
            for (int i = tid; i < 6 * A.N; i += SOLVE_THREADS) {
                const int f = i / 6, k = i % 6;
                const double xi = same_launch ? __hip_atomic_load(indirect_x + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : indirect_x[i];
                xs[4 + 8 * f + k] = xs[4 + 8 * f + k] * directRatio + xi * indirectRatio;
            }
        }
        __syncthreads();
    }
    if (nullU) {
        // orthogonalize (BA.cpp:1196-1261): x -= U^T (U x), U = orthonormal basis of the kept gauge directions (7 x n).
        // The basis entries were fetched at kernel start (registers), the products only wait for x.
        if (wv < 7) {
            double sdot = 0;
#pragma unroll
            for (int k = 0; k < ORTHO_K; k++) { const int i = l + 64 * k; sdot += u_dot[k] * xs[min(i, n - 1)]; }
            for (int i = l + 64 * ORTHO_K; i < n; i += 64) sdot += nullU[(size_t)wv * n + i] * xs[i];
            // (DPP path, total in lane 63: six xor-shuffles of a double are twelve LDS-crossbar round trips on the critical path of the iteration)
            sdot = wave_sum_dpp63_d(sdot);
            if (l == 63) dots[wv] = sdot;
        }
        __syncthreads();
        for (int i = tid; i < n; i += SOLVE_THREADS) {
            double v = xs[i];
            if (i == tid) {
#pragma unroll
                for (int e = 0; e < 7; e++) v -= u_upd[e] * dots[e];
            } else {
                for (int e = 0; e < 7; e++) v -= nullU[(size_t)e * n + i] * dots[e];
            }
            x[i] = v;
            if (MERGE && BC.g_pts > 0) publish_x_entry(BC, i, v);
            bad |= !isfinite(v);
        }
    } else {
        for (int i = tid; i < n; i += SOLVE_THREADS) {
            const double v = xs[i];
            x[i] = v;
            if (MERGE && BC.g_pts > 0) publish_x_entry(BC, i, v);
            bad |= !isfinite(v);
        }
    }
    // x is consumed by the back-substitution blocks of THIS launch.  Every entry travels as two device-scope words that carry the
    // launch's ticket beside their half of the value (they go past the non-coherent caches; a word is valid on its own), so the
    // ticket word can follow WITHOUT the writers waiting for their acknowledgements and without the workgroup meeting: a reader that
    // sees the ticket ahead of an entry simply reads that entry again.  (Round 3, first form: device-scope stores of x, s_waitcnt
    // vmcnt(0), barrier, ticket — an exposed store round trip on the critical path of the iteration.)
    if (MERGE && BC.g_pts > 0 && tid == (n - 1) % SOLVE_THREADS) __hip_atomic_store(BC.xticket, BC.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) *flag = 0;
    __syncthreads();
    if (bad) atomicOr(flag, 1);
    DBG_T(A, 53);
    DBG_BLK_END(A.dbg, 3);
}
template <int NSL, bool HYBRID, bool MERGE>
__global__ __launch_bounds__(SOLVE_THREADS) void k_ba_solve(BAArgs A, int n, int off, SolveSys Y, double* __restrict__ x, int* __restrict__ flag,
                                                            const int* newframe_res, int n_newframe, const double* lin_partial,
                                                            int n_partial, LinSummary* lin_out, FrameDev* frames_rw, int do_finish,
                                                            const double* __restrict__ nullU, const double* __restrict__ indirect_x, ReprojArgs RP, BacksubCall BC) {
    k_ba_solve_body<NSL, !MERGE, HYBRID, MERGE>(A, n, off, Y, x, flag, newframe_res, n_newframe, lin_partial, n_partial, lin_out, frames_rw, do_finish, nullU, indirect_x, RP, BC, blockIdx.x);
}


// Windows wider than the LDS-resident factorisation takes (8N (+4) > 160, i.e. N = 21 ... 32): the same system — assembled and Jacobi-
// scaled by k_ba_assemble in the block-packed layout — factorised IN GLOBAL MEMORY by one workgroup, column by column (right-looking
// LDL^T, the right-hand side riding along; same zero-pivot rule, same D^-1 and back-substitution, same tail as k_ba_solve).  A
// correctness path, not a fast one (the reference's windows hold 6-7 keyframes; this one is ~1 ms at N = 32): it exists so that
// CMLHIP_MAX_FRAMES is a limit of the whole iteration and not only of its first half.
#define SG_THREADS 1024
#define SG_MAX 272
__global__ __launch_bounds__(SG_THREADS) void k_ba_solve_global(BAArgs A, int n, int off, double* __restrict__ image, double* __restrict__ x, int* __restrict__ flag,
                                                               LinSummary* lin_out, const double* __restrict__ nullU, const double* __restrict__ indirect_x) {
    __shared__ double s_w[SG_MAX], s_l[SG_MAX], s_y[SG_MAX], s_d[SG_MAX], s_sv[SG_MAX], s_x[SG_MAX], s_dot[8];
    __shared__ double s_dk, s_yk;
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    if (A.ctl && A.ctl->stop) return;
    const int m = n - off, nb = (m + 15) / 16, mp = nb * 16;
    double* L = image;
    const double* tail = image + (size_t)(nb * (nb + 1) / 2) * BSZ;                // SVecI | scaled rhs
    auto at = [&](int i, int j) -> double& { return L[blk_off(i >> 4, j >> 4) + (i & 15) * BLD + (j & 15)]; };    // j <= i
    if (tid == 0) { lin_out->nonfinite = 0; s_bad = 0; }
    for (int i = tid; i < mp; i += SG_THREADS) { s_sv[i] = tail[i]; s_y[i] = tail[mp + i]; }
    __syncthreads();
    for (int k = 0; k < mp; k++) {
        if (tid == 0) { s_dk = at(k, k); s_yk = s_y[k]; }
        __syncthreads();
        const double d = s_dk, yk = s_yk;
        const bool tiny = !(fabs(d) > 2.2250738585072014e-308);                  // zero pivot of a positive semi-definite system: column skipped (see k_ba_solve)
        for (int i = k + 1 + tid; i < mp; i += SG_THREADS) {
            const double w = at(i, k), l = tiny ? 0.0 : w / d;
            s_w[i] = w; s_l[i] = l;
            at(i, k) = l;
            s_y[i] -= l * yk;
        }
        if (tid == 0) s_d[k] = d;
        __syncthreads();
        // trailing update A_ij -= w_i l_j, k < j <= i: 32 rows per pass, a row's columns over 32 lanes
        for (int i = k + 1 + (tid >> 5); i < mp; i += SG_THREADS / 32) {
            const double wi = s_w[i];
            for (int j = k + 1 + (tid & 31); j <= i; j += 32) at(i, j) -= wi * s_l[j];
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int i = tid; i < mp; i += SG_THREADS) { const double d = s_d[i]; s_y[i] = (fabs(d) > 2.2250738585072014e-308) ? s_y[i] / d : 0.0; }
    __syncthreads();
    for (int k = mp - 1; k >= 0; k--) {                                           // L^T x = z
        const double xk = s_y[k];
        for (int i = tid; i < k; i += SG_THREADS) s_y[i] -= at(k, i) * xk;
        __syncthreads();
    }
    for (int i = tid; i < n; i += SG_THREADS) s_x[i] = (i < off) ? 0.0 : s_sv[i - off] * s_y[i - off];
    __syncthreads();
    if (indirect_x) {                                                             // hybrid ORB term, the literal weighting of BA.cpp:2714-2727
        int mybad = 0;
        for (int i = tid; i < 6 * A.N; i += SG_THREADS) mybad |= !isfinite(indirect_x[i]);
        if (mybad) s_bad = 1;
        __syncthreads();
        if (!s_bad) {
            const double indirectRatio = 1.0 / (1.0 + 0.0), directRatio = 1.0 - indirectRatio;
            for (int i = tid; i < 6 * A.N; i += SG_THREADS) { const int f = i / 6, kk = i % 6; s_x[4 + 8 * f + kk] = s_x[4 + 8 * f + kk] * directRatio + indirect_x[i] * indirectRatio; }
        }
        __syncthreads();
    }
    if (nullU) {                                                                  // orthogonalize, BA.cpp:1196-1261
        const int wv = tid >> 6, l = tid & 63;
        if (wv < 7) {
            double sdot = 0;
            for (int i = l; i < n; i += 64) sdot += nullU[(size_t)wv * n + i] * s_x[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o);
            if (l == 0) s_dot[wv] = sdot;
        }
        __syncthreads();
        for (int i = tid; i < n; i += SG_THREADS) { double v = s_x[i]; for (int e = 0; e < 7; e++) v -= nullU[(size_t)e * n + i] * s_dot[e]; s_x[i] = v; }
        __syncthreads();
    }
    int bad = 0;
    for (int i = tid; i < n; i += SG_THREADS) { const double v = s_x[i]; x[i] = v; bad |= !isfinite(v); }
    if (tid == 0) *flag = 0;
    __syncthreads();
    if (bad) atomicOr(flag, 1);
}

// ------------------------------------------------------------------------------------------------ K6
// xAd[(host*N + target)*8 + j] = x_host . adHost(:, j) + x_target . adTarget(:, j)   (BA.cpp:1447)
__device__ __forceinline__ double xad_entry(const double* __restrict__ adH, const double* __restrict__ adT, const double* __restrict__ x, int N, int e) {
    const int j = e & 7, ht = e >> 3, h = ht / N, t = ht % N;
    const double* AH = adH + 64 * (size_t)(h + N * t); const double* AT = adT + 64 * (size_t)(h + N * t);
    double s = 0, s2 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { s += x[4 + 8 * h + i] * AH[i * 8 + j]; s2 += x[4 + 8 * t + i] * AT[i * 8 + j]; }
    return s + s2;
}
// Wide windows: every back-substitution workgroup needs the whole N*N*8 table (1 KB of adjoints behind each pair) — built once here
// instead of once per workgroup (251 workgroups x 400 KB at 20 frames).
__global__ __launch_bounds__(256) void k_ba_xad(const double* __restrict__ adH, const double* __restrict__ adT, const double* __restrict__ x,
                                                int N, double* __restrict__ xad, const int* __restrict__ stop, FrameStepArgs F) {
    if (stop && *stop) return;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * N * 8) return;
    xad[e] = xad_entry(adH, adT, x, N, e);
    if (F.on) {
        // the same adjoint columns, against the STEPPED delta: adHTd of computeDelta (BA.cpp:1120-1135) for the state the frame
        // workgroup of the back-substitution is about to write (this launch only reads the old state)
        const int j = e & 7, ht = e >> 3, h = ht / N, t = ht % N, idx = h + N * t;
        const double* AH = adH + 64 * (size_t)idx; const double* AT = adT + 64 * (size_t)idx;
        double s = 0, s2 = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { s += frame_stepped_delta(F.fs, x, h, i) * AH[i * 8 + j]; s2 += frame_stepped_delta(F.fs, x, t, i) * AT[i * 8 + j]; }
        F.adHTd[8 * (size_t)idx + j] = (float)(s + s2);
    }
}

// back-substitution (BA.cpp:1427-1487) + optional point update (doStepFromBackup, BA.cpp:976-994)
// wait until the solve workgroup of THIS launch has published x, then copy x into LDS.  One lane polls the ticket word (relaxed
// device-scope loads: an acquire per poll would invalidate caches chip-wide at every turn); the entries are then read as the
// self-validating device-scope words the solve workgroup wrote (publish_x_entry) — a word that does not carry the ticket yet is read
// again, so the producer does not have to order its ticket behind its entries.  false: gave up (never spin forever).
__device__ __forceinline__ bool wait_and_fetch_x(const int* xticket, int ticket, const unsigned long long* xpub, int n, double* __restrict__ s_x, int nthreads) {
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        int spins = 0, ok = 1;
        while (__hip_atomic_load(xticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ticket) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 22)) { ok = 0; break; }
        }
        s_ok = ok;
    }
    __syncthreads();
    bool mine_ok = true;
    for (int e = threadIdx.x; e < n; e += nthreads) {
        unsigned long long a, b;
        int spins = 0;
        for (;;) {
            a = __hip_atomic_load(xpub + 2 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b = __hip_atomic_load(xpub + 2 * e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(a >> 32) == (unsigned)ticket && (unsigned)(b >> 32) == (unsigned)ticket) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 20)) { mine_ok = false; break; }
        }
        s_x[e] = __hiloint2double((int)(unsigned)b, (int)(unsigned)a);
    }
    if (!mine_ok) s_ok = 0;
    __syncthreads();
    return s_ok != 0;
}

// NT = 256: the standalone launch.  NT = 512, INLAUNCH: the point workgroups ride in the SOLVE launch (blocks of 512 threads = two
// virtual 256-thread blocks: same points per virtual block, same partial sums, bit-identical results), request everything that does
// not depend on x while the factorisation runs, and continue when the solve workgroup publishes x.
// development stamps (cmlhip_debug_timestamps, slots 96..100): the second point block of the merged launch — wait begun, x in LDS, table built,
// steps stored, partials stored (tools/probe_phases.py)
#define BS_STAMP(k) do { if (INLAUNCH && A.dbg && bx_ == 1 && threadIdx.x == 0) A.dbg[96 + (k)] = wall_clock64(); } while (0)
template <int NT, bool INLAUNCH>
__device__ __forceinline__ void k_ba_backsub_body(const BAArgs& A, const double* __restrict__ adH, const double* __restrict__ adT,
                                                  const double* __restrict__ x_in, LinSummary* __restrict__ sum, float* __restrict__ step_partial,
                                                  int do_step, const FrameStepArgs& F, const double* __restrict__ xad, const int bx_, const int gx_,
                                                  double* __restrict__ s_xAd /* N*N*8 (+ n with INLAUNCH) */, float* __restrict__ s_redf /* [NT/256][3][4] */,
                                                  const int* xticket, const int ticket, const unsigned long long* xpub) {
    const int N = A.N;
    DBG_BLK(A.dbg, 4, 0);
    if (A.ctl && A.ctl->stop) return;
    const double* __restrict__ x = x_in;
    if (!INLAUNCH && F.on && bx_ == gx_ - 1) {               // last workgroup: the frames' half of doStepFromBackup
        FrameStepPre FP;
        frame_step_prefetch(F, FP);
        frame_step_block(F, x, FP);
        DBG_BLK_END(A.dbg, 4);
        return;
    }
    const int half = NT == 512 ? (int)(threadIdx.x >> 8) : 0, t256 = threadIdx.x & 255, vbx = NT == 512 ? 2 * bx_ + half : bx_;
    // 8 lanes per point, one residual per lane per pass (a point has at most N-1 residuals): the chain by_point -> r ->
    // {good, target, JpJdF} is walked once per point, every load unconditional (clamped), masks multiplied in.
    const int gid = vbx * 256 + t256;
    const int p = gid >> 3, i = gid & 7;
    const bool pv = p < A.P;
    const int pp = pv ? p : 0;
    // Everything that does not depend on the x.adjoint table is requested BEFORE the table is built (the barrier below would pin these
    // loads behind it: one more dependent round trip on a launch that is nothing but round trips): the point's sums, its host, the
    // calibration step, every pass's codes / targets / residual slots, the backed-up inverse depth, and — behind the codes — the JpJdF.
    const float* pa = A.pt_acc + (size_t)pp * PT_ACC_STRIDE;
    const int host = A.pt_host[pp];
    const float pa12 = pa[12], pa13 = pa[13], hcd = (i < 4) ? pa[2 + (i & 3)] + 0.f : 0.f, hcl = (i < 4) ? pa[8 + (i & 3)] : 0.f;
    // (all passes' codes / targets / residual slots: up to 4 passes of 8 slots, CMLHIP_MAX_FRAMES = 32; clamped slots are masked)
    int tgls[4], ress[4];
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
        const bool lv = 8 * ps + i < A.pt_stride;
        const int slot = pp * A.pt_stride + min(8 * ps + i, A.pt_stride - 1);
        const int tg = A.point_tgt[slot], rs = A.point_res[slot];
        tgls[ps] = lv ? tg : 0; ress[ps] = lv ? rs : -1;
    }
    const float backup_p = A.pt_backup[pp];
    // second round trip: the JpJdF of every pass and the residual's isActiveAndIsGoodNEW flag, requested together (unconditional, clamped) — also ahead of the table's barrier
    float4 v0s[4], v1s[4];
    int codes[4];                                            // >= 0: the slot holds a good residual (what the per-slot code of applyRes said before round 6)
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
        const int r = max(ress[ps], 0);
        v0s[ps] = *reinterpret_cast<const float4*>(A.r_jpjdf + PS_STRIDE * (size_t)r); v1s[ps] = *reinterpret_cast<const float4*>(A.r_jpjdf + PS_STRIDE * (size_t)r + 4);
        codes[ps] = (ress[ps] >= 0 && A.r_good[r] != 0) ? 2 * r : -1;
    }
    // third: the adjoint columns behind this thread's entries of the x.adjoint table (static over the iteration) — with the table built
    // after the wait they were a dependent trip of their own behind the arrival of x
    const int ne = N * N * 8;
    const bool pre_ad = INLAUNCH && !xad && ne <= 2 * NT;
    double ahc[2][8], atc[2][8];
    if (pre_ad) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = min((int)threadIdx.x + u * NT, ne - 1);
            const int j = e & 7, ht = e >> 3, h = ht / N, t = ht % N;
            const double* AH = adH + 64 * (size_t)(h + N * t); const double* AT = adT + 64 * (size_t)(h + N * t);
#pragma unroll
            for (int k = 0; k < 8; k++) { ahc[u][k] = AH[k * 8 + j]; atc[u][k] = AT[k * 8 + j]; }
        }
    }
    if (INLAUNCH) {
        double* s_x = s_xAd + N * N * 8;
        BS_STAMP(0);
        if (!wait_and_fetch_x(xticket, ticket, xpub, A.n, s_x, NT)) { if (threadIdx.x == 0) atomicAdd(&sum->nonfinite, 1); return; }
        x = s_x;
        BS_STAMP(1);
    }
    const double xc = x[i & 3];
    if (xad) {                                               // wide windows: the table was built once by k_ba_xad
        for (int e = threadIdx.x; e < ne; e += NT) s_xAd[e] = xad[e];
    } else if (pre_ad) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = (int)threadIdx.x + u * NT;
            if (e < ne) {
                const int ht = e >> 3, h = ht / N, t = ht % N;
                double s = 0, s2 = 0;                        // (xad_entry's sums, operand for operand)
#pragma unroll
                for (int k = 0; k < 8; k++) { s += x[4 + 8 * h + k] * ahc[u][k]; s2 += x[4 + 8 * t + k] * atc[u][k]; }
                s_xAd[e] = s + s2;
            }
        }
    } else {
        for (int e = threadIdx.x; e < ne; e += NT) s_xAd[e] = xad_entry(adH, adT, x, N, e);
    }
    if (!INLAUNCH && bx_ == 0 && threadIdx.x == 0) sum->nonfinite = 0;      // (in the solve launch its workgroup 0 reset the counter at its start)
    __syncthreads();
    BS_STAMP(2);
    float sumID = 0, sumNID = 0, numID = 0;
    {
        int ngood = 0;
        // scalar_t b = bdSumF; b -= mCalibStep.dot(Hcd_A + Hcd_L); then b -= xAd * JpJdF for every good residual IN LIST ORDER
        // (BA.cpp:1469-1479).  Both dots reduce by halves in Eigen (pinned on the reference's vendored Eigen, tests/golden) — the
        // 8-lane butterfly below and the bracketing of `d` are exactly that order — and the subtractions are replayed one
        // residual at a time, so the point step is the reference's bit for bit.
        double bb = (double)pa13 - sum8d(i < 4 ? (-xc) * ((double)hcd + (double)hcl) : 0.0);
#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            if (8 * ps >= A.pt_stride) break;                 // uniform
            const int code = codes[ps], tgl = tgls[ps];       // efsJ code kept by applyRes, static target
            const bool good = pv && code >= 0;
            const float4 v0 = v0s[ps], v1 = v1s[ps];
            const double* xa = s_xAd + 8 * (host * N + (max(tgl, 0) & 255));
            double d = ((xa[0] * (double)v0.x + xa[1] * (double)v0.y) + (xa[2] * (double)v0.z + xa[3] * (double)v0.w))
                     + ((xa[4] * (double)v1.x + xa[5] * (double)v1.y) + (xa[6] * (double)v1.z + xa[7] * (double)v1.w));
            d = good ? d : 0.0;
            ngood += (int)sum8(good ? 1.f : 0.f);
#pragma unroll
            for (int k = 0; k < 8; k++) bb -= __shfl(d, (threadIdx.x & 56) + k);      // b - 0.0 is exact for the slots that are not good
        }
        double st = 0.0;
        if (ngood > 0) {
            st = -bb * (double)pa12;
            if (pv && i == 0 && !isfinite(st)) atomicAdd(&sum->nonfinite, 1);
        }
        double nid_w = 0.0;
        int nid_ok = 0;
        if (pv && i == 0) {
            A.pt_step[p] = st;
            if (do_step) {
                const double nid = (double)backup_p + st;
                if (isfinite(nid) && nid > 0) {
                    A.pt_idepth[p] = nid;
                    sumID = (float)(st * st); sumNID = (float)fabs((double)backup_p); numID = 1.f;
                    A.pt_idepth_zero[p] = (float)nid;
                    nid_w = nid; nid_ok = 1;
                }
            }
        }
        if (do_step) {
            // the residual kernel of the resident loop reads the inverse depth per RESIDUAL (no hop through the point index): the
            // 8 lanes of the point refresh the copies of its residuals
            const int src = threadIdx.x & 56;                 // (blockDim = 256: the 8-lane group never straddles a wave)
            nid_w = __shfl(nid_w, src); nid_ok = __shfl(nid_ok, src);
#pragma unroll
            for (int ps = 0; ps < 4; ps++) if (pv && nid_ok && ress[ps] >= 0) A.r_idepth[ress[ps]] = nid_w;
        }
    }
    BS_STAMP(3);
    if (do_step) {                                           // fixed-order block partials; the host adds the few blocks
        sumID = wave_sum_dpp63(sumID); sumNID = wave_sum_dpp63(sumNID); numID = wave_sum_dpp63(numID);      // (totals in lane 63)
        float* red = s_redf + 12 * half;                      // [3][4] of this virtual block
        if ((threadIdx.x & 63) == 63) { red[0 * 4 + (t256 >> 6)] = sumID; red[1 * 4 + (t256 >> 6)] = sumNID; red[2 * 4 + (t256 >> 6)] = numID; }
        __syncthreads();
        if (t256 < 3 && vbx < A.n_step_blocks) {
            const int k = t256;
            step_partial[4 * vbx + k] = ((red[k * 4 + 0] + red[k * 4 + 1]) + red[k * 4 + 2]) + red[k * 4 + 3];
        }
    }
    BS_STAMP(4);
    DBG_BLK_END(A.dbg, 4);
}
__global__ __launch_bounds__(256) void k_ba_backsub(BAArgs A, const double* __restrict__ adH, const double* __restrict__ adT,
                                                    const double* __restrict__ x, LinSummary* __restrict__ sum, float* __restrict__ step_partial,
                                                    int do_step, FrameStepArgs F, const double* __restrict__ xad) {
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];     // N*N*8, index (host*N + target)*8 + j  (:1447)
    __shared__ float s_red[12];
    k_ba_backsub_body<256, false>(A, adH, adT, x, sum, step_partial, do_step, F, xad, blockIdx.x, gridDim.x, s_dyn, s_red, nullptr, 0, nullptr);
}


__global__ void k_ba_backup_points(BAArgs A) {          // BA.cpp:919-922
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < A.P) A.pt_backup[p] = (float)A.pt_idepth[p];
}
__global__ void k_ba_restore_points(BAArgs A) {         // loadSateBackup, BA.cpp:938-942
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < A.P) { A.pt_idepth[p] = (double)A.pt_backup[p]; A.pt_idepth_zero[p] = A.pt_backup[p]; }
}
// doStepFromBackup, point part (BA.cpp:976-994), standalone form
__global__ __launch_bounds__(256) void k_ba_step_points(BAArgs A, float* __restrict__ step_partial) {
    __shared__ float s_red[3][4];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    float sumID = 0, sumNID = 0, numID = 0;
    if (p < A.P) {
        const double st = A.pt_step[p];
        const double nid = (double)A.pt_backup[p] + st;
        if (isfinite(nid) && nid > 0) {
            A.pt_idepth[p] = nid;
            sumID = (float)(st * st); sumNID = (float)fabs((double)A.pt_backup[p]); numID = 1.f;
            A.pt_idepth_zero[p] = (float)nid;
        }
    }
    sumID = wave_sum(sumID); sumNID = wave_sum(sumNID); numID = wave_sum(numID);
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = sumID; s_red[1][threadIdx.x >> 6] = sumNID; s_red[2][threadIdx.x >> 6] = numID; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        step_partial[4 * blockIdx.x + k] = ((s_red[k][0] + s_red[k][1]) + s_red[k][2]) + s_red[k][3];
    }
}

// ------------------------------------------------------------------------------------------------ launchers
static inline int ldg_of(int n) { return ((n + 1 + 15) / 16) * 16; }
static size_t solve_lds_bytes(int m) {
    const int nb = (m + 15) / 16, mp = nb * 16;
    size_t d = (size_t)(nb * (nb + 1) / 2) * BSZ + (size_t)nb * BSZ + 4 * (size_t)mp;
    if (d < 2048 + 1024) d = 2048 + 1024;                     // lin_finish_block scratch: 2048 u32-pairs + 1024 doubles
    return d * sizeof(double);
}

// the marginalisation prior of the resident loop rides in the frame step (frame_prior_rhs, ba_frames.h)
static inline void fill_frame_prior(cmlhip_ctx* c, FrameStepArgs& F, int n) {
    F.n = n;
    if (c->resident_prior) { F.HM = c->HM.as<double>(); F.bM_raw = c->bM_raw.as<double>(); F.bM_top = c->bM.as<double>(); }
    else { F.HM = nullptr; F.bM_raw = nullptr; F.bM_top = nullptr; }
}

// arguments of K3 / K4 for the ACTIVE pass of a window as the context stands (shared by the solo launcher and the batched iteration)
static void fill_acc_args(cmlhip_ctx* c, const BAArgs& A, bool do_backup, AccArgs& X) {
    const int n = A.n;
    X.adH = c->adH.as<double>(); X.adT = c->adT.as<double>(); X.adHTd = c->adHTd.as<float>(); X.cdelta = c->vec_small.as<double>();
    X.acc_out = c->acc_pair[0].as<float>(); X.num_out = c->acc_num[0].as<int>(); X.pair_blocks = c->pair_blocks.as<double>();
    X.ldg = ldg_of(n); X.G = c->G.as<double>(); X.Wt = X.G + (size_t)A.P * X.ldg; X.do_backup = do_backup ? 1 : 0;
    X.part = c->rs_part.as<float>(); X.tile_off = c->rs_tile_off.as<int>(); X.tile = c->rs_tile;
}
static bool fill_sys_args(cmlhip_ctx* c, const BAArgs& A, const AccArgs& X, double lambda, bool have_hm, bool system_only, bool marg, SysArgs& S) {
    const int N = A.N, n = A.n, NN = N * N;
    const double* vs = c->vec_small.as<double>();        // cdelta[4] cprior[4] prior[8N] dprior[8N]
    double* pbL = c->pair_blocks.as<double>() + (size_t)PB_STRIDE * NN;
    S.N = N; S.n = n; S.ldg = X.ldg; S.ntile = X.ldg / 16; S.P = A.P; S.use_lin_blocks = (c->n_lin > 0 && !marg) ? 1 : 0;
    S.G = X.G; S.Wt = X.Wt; S.pbA = c->pair_blocks.as<double>(); S.pbL = pbL; S.dbg = A.dbg; S.stop = A.ctl ? &A.ctl->stop : nullptr;
    S.cdelta = vs; S.cprior = vs + 4; S.prior = vs + 8; S.dprior = vs + 8 + 8 * N;
    S.HM = have_hm ? c->HM.as<double>() : nullptr; S.bM = have_hm ? c->bM.as<double>() : nullptr;
    S.lambda = lambda;
    S.HA = c->HA.as<double>(); S.bA = c->bA.as<double>(); S.HL = c->HL.as<double>(); S.bL = c->bL.as<double>();
    S.Hsc = c->Hsc.as<double>(); S.bsc = c->bsc.as<double>(); S.Hb = c->Hf.as<double>(); S.bb = c->bf.as<double>();
    const int ntiles = S.ntile * (S.ntile + 1) / 2;
    S.nsl = cml_sys_slices(A.P);
    S.nsyrk = system_only ? 0 : ntiles * S.nsl;             // the Schur slices only change with the residuals
    static const char* e_super = getenv("CMLHIP_SYS_SUPER");  // development: 0 / 1 forces the plain / the super-tile SYRK
    const bool super = e_super ? atoi(e_super) != 0 : S.ntile >= 8;
    S.nt2 = (S.ntile + 1) / 2;
    if (super && !system_only) S.nsyrk = S.nt2 * (S.nt2 + 1) / 2 * S.nsl;
    S.part = c->syrk_part.as<double>();
    return super;
}

int cml_launch_accumulate(cmlhip_ctx* c, const BAArgs& A, double lambda, bool have_hm, bool do_backup, bool system_only, bool marg) {
    const int N = A.N, NN = N * N;
    const double* vs = c->vec_small.as<double>();
    AccArgs X;
    fill_acc_args(c, A, do_backup, X);
    double* pbL = c->pair_blocks.as<double>() + (size_t)PB_STRIDE * NN;
    if (marg) {                                          // marginalizePointsF: MARGINALIZED-mode blocks of the selected points only
        k_ba_acc<<<NN, 1024, 0, c->stream>>>(A, X, CMLHIP_MODE_MARGINALIZED);
        if (A.P > 0) {
            (void)hipMemsetAsync(X.G, 0, sizeof(double) * (size_t)A.P * X.ldg, c->stream);
            k_ba_point_rows_marg<<<cml_div_up(A.P * 8, 256), 256, 0, c->stream>>>(A, X);
        }
    } else if (!system_only) {
        if (c->n_lin > 0) {                              // rare path: LINEARIZED residuals present
            AccArgs XL = X;
            XL.acc_out = c->acc_pair[1].as<float>(); XL.num_out = c->acc_num[1].as<int>(); XL.pair_blocks = pbL;
            k_ba_acc<<<NN, 1024, 0, c->stream>>>(A, XL, 1);
            if (A.P > 0) k_ba_point_bdL<<<cml_div_up(A.P, 256), 256, 0, c->stream>>>(A, c->adHTd.as<float>(), vs);
        }
        static const bool acc16 = getenv("CMLHIP_ACC_1024") != nullptr;          // development: the 1024-thread workgroups in the resident loop too
        if (c->efs_in_partials && !acc16) CML_LAUNCH_EV(c, k_ba_acc_rs, NN + cml_div_up(A.P, PT_PER_BLOCK_RS), 256, 0, A, X);
        else CML_LAUNCH_EV(c, k_ba_acc, NN + cml_div_up(A.P, PT_PER_BLOCK), 1024, 0, A, X, c->efs_in_partials ? CML_MODE_ACTIVE_TILES : 0);
    }
    SysArgs S;
    const bool super = fill_sys_args(c, A, X, lambda, have_hm, system_only, marg, S);
    c->sys_lambda = lambda;
    if (super) k_ba_system<true><<<S.nsyrk + N + 1, 64 * SYS_NW, 0, c->stream>>>(S);
    else k_ba_system<false><<<S.nsyrk + N + 1, 64 * SYS_NW, 0, c->stream>>>(S);
    return CMLHIP_OK;
}

int cml_launch_schur_out(cmlhip_ctx* c, const BAArgs& A) {
    SysArgs S = {};
    S.n = A.n; S.ntile = ldg_of(A.n) / 16; S.nsl = cml_sys_slices(A.P); S.part = c->syrk_part.as<double>();
    S.Hsc = c->Hsc.as<double>(); S.bsc = c->bsc.as<double>();
    k_ba_schur_out<<<S.ntile * (S.ntile + 1) / 2, 256, 0, c->stream>>>(S);
    return CMLHIP_OK;
}

int cml_launch_solve(cmlhip_ctx* c, const BAArgs& A, int optcal, bool with_lin_finish, bool ortho, const double* indirect_x, const ReprojArgs* rp, bool merge_backsub) {
    const int n = A.n, off = optcal ? 0 : 4, m = n - off;
    const size_t sh = solve_lds_bytes(m);
    int* flag = reinterpret_cast<int*>(c->scal.as<char>() + 256);
    if (sh > 160 * 1024) {                                   // the LDS-resident factorisation takes 8N (+4 with the calibration block) <= 160: wider windows factorise in global memory
        CML_REQUIRE(c, ((m + 15) / 16) * 16 <= SG_MAX, CMLHIP_ERR_INVALID, "window too wide for the solver");
        CML_REQUIRE(c, !rp, CMLHIP_ERR_INVALID, "hybrid term inside the solve launch: windows of up to 20 frames");
        SolveSys Yg;
        Yg.Hb = c->Hf.as<double>(); Yg.bb = c->bf.as<double>(); Yg.part = c->syrk_part.as<double>();
        Yg.nsl = cml_sys_slices(A.P); Yg.ntile = ldg_of(n) / 16; Yg.lambda = c->sys_lambda;
        const int nbg = (m + 15) / 16, nblkg = nbg * (nbg + 1) / 2;
        if (int rc = cml_ensure(c, c->solve_image, 8 * ((size_t)nblkg * BSZ + 2 * (size_t)nbg * 16))) return rc;
        Yg.image = c->solve_image.as<double>();
        const int* stopg = A.ctl ? &A.ctl->stop : nullptr;
        switch (Yg.nsl) {
            case 1: k_ba_assemble<1><<<nblkg, 256, 0, c->stream>>>(n, off, Yg, stopg); break;
            case 2: k_ba_assemble<2><<<nblkg, 256, 0, c->stream>>>(n, off, Yg, stopg); break;
            case 4: k_ba_assemble<4><<<nblkg, 256, 0, c->stream>>>(n, off, Yg, stopg); break;
            default: k_ba_assemble<8><<<nblkg, 256, 0, c->stream>>>(n, off, Yg, stopg); break;
        }
        if (c->ext_stop_if_merged) c->ext_stop_if_merged = nullptr;
        k_ba_solve_global<<<1, SG_THREADS, 0, c->stream>>>(A, n, off, Yg.image, c->xvec.as<double>(), flag, c->scal.as<LinSummary>(),
                                                          ortho ? c->null_basis.as<double>() : nullptr, indirect_x);
        c->backsub_merged = false;
        if (with_lin_finish) return cml_launch_lin_finish(c, A);
        return CMLHIP_OK;
    }
    SolveSys Y;
    Y.Hb = c->Hf.as<double>(); Y.bb = c->bf.as<double>(); Y.part = c->syrk_part.as<double>();
    Y.nsl = cml_sys_slices(A.P); Y.ntile = ldg_of(n) / 16; Y.lambda = c->sys_lambda;
    const int nb = (m + 15) / 16, nblk = nb * (nb + 1) / 2;
    const bool wide = nblk * 256 > SOLVE_IPT * SOLVE_THREADS;
    Y.image = nullptr;
    if (wide) {
        if (int rc = cml_ensure(c, c->solve_image, 8 * ((size_t)nblk * BSZ + 2 * (size_t)nb * 16))) return rc;
        Y.image = c->solve_image.as<double>();
    }
    const int* stopp = A.ctl ? &A.ctl->stop : nullptr;
    ReprojArgs RPv;
    memset(&RPv, 0, sizeof RPv);
    if (rp) RPv = *rp;                                          // hybrid term: its per-frame workgroups ride in this launch (blocks 2 .. 2 + N)
    // K6 in this launch (see BacksubCall): small windows of the resident loop only
    BacksubCall BC;
    memset(&BC, 0, sizeof BC);
    const bool merge = merge_backsub && !wide && with_lin_finish && !(A.N >= 12 && A.P >= 2048) && A.P > 0;
    if (merge) {
        BC.adH = c->adH.as<double>(); BC.adT = c->adT.as<double>(); BC.step_partial = c->step_partial.as<float>();
        FrameStepArgs& F = BC.F;
        F.on = c->resident_on ? 1 : 0;
        if (F.on) {
            F.fs = c->frame_state.as<cmlhip_ba_frame_state>(); F.pairs = c->pairs.as<cmlhip_ba_pair>(); F.pre_w2c = c->pre_w2c.as<double>();
            F.adH = c->adH.as<double>(); F.adT = c->adT.as<double>(); F.adHTd = c->adHTd.as<float>();
            F.dprior = c->vec_small.as<double>() + 8 + 8 * A.N;
            for (int i = 0; i < 4; i++) F.sc[i] = c->res_scales[i];
            F.N = A.N; F.frame_sums = A.ctl ? A.ctl->frame_sums : nullptr;
            fill_frame_prior(c, F, A.n);
        }
        BC.g_pts = cml_div_up(A.P * 8, 512);
        if (int rc = cml_ensure(c, c->x_ticket, 64 + 16 * (8 * CMLHIP_MAX_FRAMES + 4))) return rc;
        if (!c->x_ticket_zeroed) { CML_CHECK(c, hipMemsetAsync(c->x_ticket.p, 0, c->x_ticket.bytes, c->stream)); c->x_ticket_zeroed = true; }
        BC.xticket = c->x_ticket.as<int>(); BC.ticket = ++c->x_ticket_seq;
        BC.xpub = reinterpret_cast<unsigned long long*>(c->x_ticket.as<char>() + 64);
    }
    if (merge && c->ext_stop_if_merged) c->ext_stop = c->ext_stop_if_merged;
    c->ext_stop_if_merged = nullptr;
    const int grid = (rp || merge) ? 2 + (rp ? rp->N : 0) + (merge ? BC.F.on + BC.g_pts : 0) : (with_lin_finish ? 2 : 1);
    size_t shm = sh;
    if (merge) shm = std::max(shm, (size_t)(A.N * A.N * 8 + ((n + 1) & ~1) + 16) * sizeof(double));
#define LAUNCH_SOLVE_H(NSL, HYB, MRG, BIT) do { \
        if (!(c->attr_done & (1u << (BIT)))) { (void)hipFuncSetAttribute((const void*)k_ba_solve<NSL, HYB, MRG>, hipFuncAttributeMaxDynamicSharedMemorySize, (MRG) ? 96 * 1024 : 160 * 1024); c->attr_done |= 1u << (BIT); } \
        CML_LAUNCH_EV(c, (k_ba_solve<NSL, HYB, MRG>), grid, SOLVE_THREADS, shm, A, n, off, Y, c->xvec.as<double>(), flag, \
            (const int*)c->newframe_res.as<int>(), c->n_newframe, (const double*)c->lin_partial.as<double>(), c->lin_partial_n, c->scal.as<LinSummary>(), \
            c->frames.as<FrameDev>(), with_lin_finish ? 1 : 0, (const double*)(ortho ? c->null_basis.as<double>() : nullptr), indirect_x, RPv, BC); } while (0)
#define LAUNCH_SOLVE(NSL, B0) do { \
        if (wide) k_ba_assemble<NSL><<<nblk, 256, 0, c->stream>>>(n, off, Y, stopp); \
        if (rp && merge) LAUNCH_SOLVE_H(NSL, true, true, (B0) + 3); else if (rp) LAUNCH_SOLVE_H(NSL, true, false, (B0) + 2); \
        else if (merge) LAUNCH_SOLVE_H(NSL, false, true, (B0) + 1); else LAUNCH_SOLVE_H(NSL, false, false, (B0)); } while (0)
    c->backsub_merged = merge;
    switch (Y.nsl) {
        case 1: LAUNCH_SOLVE(1, 0); break;
        case 2: LAUNCH_SOLVE(2, 4); break;
        case 4: LAUNCH_SOLVE(4, 8); break;
        default: LAUNCH_SOLVE(8, 12); break;
    }
#undef LAUNCH_SOLVE
#undef LAUNCH_SOLVE_H
    return CMLHIP_OK;
}

int cml_launch_backsub(cmlhip_ctx* c, const BAArgs& A, bool do_step) {
    const size_t sh = (size_t)A.N * A.N * 8 * sizeof(double);
    FrameStepArgs F = {};
    F.on = (do_step && c->resident_on) ? 1 : 0;
    if (F.on) {
        F.fs = c->frame_state.as<cmlhip_ba_frame_state>(); F.pairs = c->pairs.as<cmlhip_ba_pair>(); F.pre_w2c = c->pre_w2c.as<double>();
        F.adH = c->adH.as<double>(); F.adT = c->adT.as<double>(); F.adHTd = c->adHTd.as<float>();
        F.dprior = c->vec_small.as<double>() + 8 + 8 * A.N;
        for (int i = 0; i < 4; i++) F.sc[i] = c->res_scales[i];
        F.N = A.N; F.frame_sums = A.ctl ? A.ctl->frame_sums : nullptr;
        fill_frame_prior(c, F, A.n);
    }
    if (cml_div_up(A.P * 8, 256) + F.on == 0) return CMLHIP_OK;     // no points and no frame step: nothing to launch (a zero grid is a HIP error)
    const double* xad = nullptr;
    if (A.N >= 12 && A.P >= 2048) {                                  // many workgroups x a large table: build it once
        if (int rc = cml_ensure(c, c->xad, 8 * (size_t)A.N * A.N * 8)) return rc;
        k_ba_xad<<<cml_div_up(A.N * A.N * 8, 256), 256, 0, c->stream>>>(c->adH.as<double>(), c->adT.as<double>(), c->xvec.as<double>(), A.N,
                                                                       c->xad.as<double>(), A.ctl ? &A.ctl->stop : nullptr, F);
        xad = c->xad.as<double>();
        F.adhtd_done = F.on;
    }
    CML_LAUNCH_EV(c, k_ba_backsub, cml_div_up(A.P * 8, 256) + F.on, 256, sh, A, (const double*)c->adH.as<double>(), (const double*)c->adT.as<double>(),
                  (const double*)c->xvec.as<double>(), c->scal.as<LinSummary>(), c->step_partial.as<float>(), do_step ? 1 : 0, F, xad);
    return CMLHIP_OK;
}
int cml_launch_backup_points(cmlhip_ctx* c, const BAArgs& A) {
    if (A.P > 0) k_ba_backup_points<<<cml_div_up(A.P, 256), 256, 0, c->stream>>>(A);
    return CMLHIP_OK;
}
int cml_launch_restore_points(cmlhip_ctx* c, const BAArgs& A) {
    if (A.P > 0) k_ba_restore_points<<<cml_div_up(A.P, 256), 256, 0, c->stream>>>(A);
    return CMLHIP_OK;
}
int cml_launch_step_points(cmlhip_ctx* c, const BAArgs& A) {
    if (A.P > 0) k_ba_step_points<<<cml_div_up(A.P, 256), 256, 0, c->stream>>>(A, c->step_partial.as<float>());
    return CMLHIP_OK;
}

// ================================================================================================ several windows per launch
// cmlhip_ba_iteration_batch: ONE resident Gauss-Newton iteration of S independent windows (sequence shards mapped to one GPU) in the
// five launches a single window takes — gridDim.y = window.  A small latency-bound window occupies well under a third of the chip
// and S host threads launching S streams are host-bound from S = 4 on (tools/probe_concurrent.py), so throughput mode batches.
// Every kernel body is the one the solo path runs (k_*_body), on the window's own buffers, with the arguments it would get solo: a
// window's result is bit-identical to its solo run.  The per-window argument blocks live in device memory (read through uniform
// addresses: scalar loads, like the kernel-argument segment they replace) and are re-uploaded only when they change.
struct BatchWin {
    BAArgs A; AccArgs X; SysArgs S; SolveSys Y; FrameStepArgs F;
    int acc_mode, g_acc, g_sys, g_back;
    int n, off, n_newframe, n_partial;
    double* x; int* flag; const int* newframe_res; const double* lin_partial; LinSummary* lin_out; FrameDev* frames_rw; const double* nullU;
    const double* adH; const double* adT; float* step_partial;
};
__global__ __launch_bounds__(1024) void k_ba_acc_batch(const BatchWin* __restrict__ W) {
    const BatchWin& w = *(const BatchWin*)(const BatchWin __attribute__((address_space(4)))*)(W + blockIdx.y);     // constant address space: scalar loads
    if ((int)blockIdx.x >= w.g_acc) return;
    k_ba_acc_body(w.A, w.X, w.acc_mode, blockIdx.x, w.g_acc);
}
__global__ __launch_bounds__(256) void k_ba_acc_rs_batch(const BatchWin* __restrict__ W) {          // every window's Jacobians in tile form
    const BatchWin& w = *(const BatchWin*)(const BatchWin __attribute__((address_space(4)))*)(W + blockIdx.y);
    if ((int)blockIdx.x >= w.g_acc) return;
    k_ba_acc_rs_body(w.A, w.X, blockIdx.x, w.g_acc);
}
__global__ __launch_bounds__(64 * SYS_NW) __attribute__((amdgpu_waves_per_eu(SYS_WPE, SYS_WPE))) void k_ba_system_batch(const BatchWin* __restrict__ W) {
    const BatchWin& w = *(const BatchWin*)(const BatchWin __attribute__((address_space(4)))*)(W + blockIdx.y);     // constant address space: scalar loads
    if ((int)blockIdx.x >= w.g_sys) return;
    k_ba_system_body<false>(w.S, blockIdx.x);
}
template <int NSL>
__global__ __launch_bounds__(SOLVE_THREADS) void k_ba_solve_batch(const BatchWin* __restrict__ W) {
    const BatchWin& w = *(const BatchWin*)(const BatchWin __attribute__((address_space(4)))*)(W + blockIdx.y);     // constant address space: scalar loads
    ReprojArgs RP0;
    RP0.N = 0; RP0.ready = nullptr; RP0.ticket = 0;         // the hybrid term is not batched
    BacksubCall BC0;
    BC0.g_pts = 0; BC0.xticket = nullptr; BC0.ticket = 0; BC0.xpub = nullptr; BC0.F.on = 0;
    k_ba_solve_body<NSL, false>(w.A, w.n, w.off, w.Y, w.x, w.flag, w.newframe_res, w.n_newframe, w.lin_partial, w.n_partial, w.lin_out, w.frames_rw, 1, w.nullU,
                         nullptr, RP0, BC0, blockIdx.x);
}
__global__ __launch_bounds__(256) void k_ba_backsub_batch(const BatchWin* __restrict__ W) {
    const BatchWin& w = *(const BatchWin*)(const BatchWin __attribute__((address_space(4)))*)(W + blockIdx.y);     // constant address space: scalar loads
    if ((int)blockIdx.x >= w.g_back) return;
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    __shared__ float s_red[12];
    k_ba_backsub_body<256, false>(w.A, w.adH, w.adT, w.x, w.lin_out, w.step_partial, 1, w.F, nullptr, blockIdx.x, w.g_back, s_dyn, s_red, nullptr, 0, nullptr);
}

int cml_iteration_batch(cmlhip_ctx* const* ctxs, int S, double lambda) {
    cmlhip_ctx* c0 = ctxs[0];
    // The launches below all go to ctxs[0]'s stream, but a window's upload / cmlhip_ba_set_resident_state are asynchronous on ITS context's
    // stream (h2d scatter kernel, closing memset): order that work ahead of the batch.  A stream that is already idle (every round but the
    // first) costs one query and no device-side operation.
    for (int k = 1; k < S; k++) {
        cmlhip_ctx* c = ctxs[k];
        if (hipStreamQuery(c->stream) == hipSuccess) continue;
        if (!c->batch_ev) CML_CHECK(c0, hipEventCreateWithFlags(&c->batch_ev, hipEventDisableTiming));
        CML_CHECK(c0, hipEventRecord(c->batch_ev, c->stream));
        CML_CHECK(c0, hipStreamWaitEvent(c0->stream, c->batch_ev, 0));
    }
    (void)hipGetLastError();                                // (hipStreamQuery reports hipErrorNotReady through the sticky error too)
    std::vector<BatchWin> H((size_t)S);
    std::vector<unsigned char> Hrs;
    int g_acc = 0, g_sys = 0, g_back = 0, nsl = 0, rs_blocks = 0;
    bool all_tiles = true;
    size_t solve_lds = 0, back_lds = 0;
    for (int k = 0; k < S; k++) {
        cmlhip_ctx* c = ctxs[k];
        BatchWin& w = H[k];
        memset(&w, 0, sizeof w);
        if (c->arith_relaxed != ctxs[0]->arith_relaxed) { ctxs[0]->err = "cmlhip_ba_iteration_batch: the windows of a batch must share one arithmetic mode (cmlhip_ba_set_arithmetic)"; return CMLHIP_ERR_INVALID; }
        BAArgs& A = w.A;
        cml_make_ba_args(c, A);
        A.fuse_apply = 1;                                    // the step is always accepted here (forceAccept, BA.h:265)
        A.ctl = nullptr;                                     // throughput mode: every enqueued iteration runs (no early exit)
        A.dbg = nullptr;
        fill_acc_args(c, A, true, w.X);
        w.acc_mode = c->efs_in_partials ? CML_MODE_ACTIVE_TILES : 0;
        w.g_acc = A.N * A.N + cml_div_up(A.P, PT_PER_BLOCK);            // (re-sized below when every window takes the 256-thread form)
        all_tiles = all_tiles && c->efs_in_partials;
        const bool super = fill_sys_args(c, A, w.X, lambda, c->resident_prior, false, false, w.S);
        if (super) { c0->err = "cmlhip_ba_iteration_batch: a window this wide fills the chip on its own (use cmlhip_ba_iteration_async)"; return CMLHIP_ERR_INVALID; }
        w.g_sys = w.S.nsyrk + A.N + 1;
        // K5, as cml_launch_solve
        const int n = A.n, off = 4, m = n - off;
        const size_t sh = solve_lds_bytes(m);
        const int nb = (m + 15) / 16, nblk = nb * (nb + 1) / 2;
        if (sh > 160 * 1024 || nblk * 256 > SOLVE_IPT * SOLVE_THREADS) { c0->err = "cmlhip_ba_iteration_batch: window too wide for the batched solve"; return CMLHIP_ERR_INVALID; }
        w.n = n; w.off = off;
        w.Y.Hb = c->Hf.as<double>(); w.Y.bb = c->bf.as<double>(); w.Y.part = c->syrk_part.as<double>();
        w.Y.nsl = cml_sys_slices(A.P); w.Y.ntile = ldg_of(n) / 16; w.Y.lambda = lambda; w.Y.image = nullptr;
        if (k == 0) nsl = w.Y.nsl;
        if (w.Y.nsl != nsl) { c0->err = "cmlhip_ba_iteration_batch: the windows of a batch must fall into one point-slice class (P <= 512 / 1024 / 2048 / more)"; return CMLHIP_ERR_INVALID; }
        w.x = c->xvec.as<double>(); w.flag = reinterpret_cast<int*>(c->scal.as<char>() + 256);
        w.newframe_res = c->newframe_res.as<int>(); w.n_newframe = c->n_newframe; w.lin_partial = c->lin_partial.as<double>(); w.n_partial = c->lin_partial_n;
        w.lin_out = c->scal.as<LinSummary>(); w.frames_rw = c->frames.as<FrameDev>();
        w.nullU = (c->resident_on && c->have_null && c->resident_iter >= 2) ? c->null_basis.as<double>() : nullptr;
        // K6, as cml_launch_backsub(do_step = true)
        if (A.N >= 12 && A.P >= 2048) { c0->err = "cmlhip_ba_iteration_batch: window too wide for the batched back-substitution"; return CMLHIP_ERR_INVALID; }
        FrameStepArgs& F = w.F;
        F.on = c->resident_on ? 1 : 0;
        if (F.on) {
            F.fs = c->frame_state.as<cmlhip_ba_frame_state>(); F.pairs = c->pairs.as<cmlhip_ba_pair>(); F.pre_w2c = c->pre_w2c.as<double>();
            F.adH = c->adH.as<double>(); F.adT = c->adT.as<double>(); F.adHTd = c->adHTd.as<float>();
            F.dprior = c->vec_small.as<double>() + 8 + 8 * A.N;
            for (int i = 0; i < 4; i++) F.sc[i] = c->res_scales[i];
            F.N = A.N; F.frame_sums = nullptr;
            fill_frame_prior(c, F, A.n);
        }
        w.g_back = cml_div_up(A.P * 8, 256) + F.on;
        w.adH = c->adH.as<double>(); w.adT = c->adT.as<double>(); w.step_partial = c->step_partial.as<float>();
        g_acc = std::max(g_acc, w.g_acc); g_sys = std::max(g_sys, w.g_sys); g_back = std::max(g_back, w.g_back);
        solve_lds = std::max(solve_lds, sh); back_lds = std::max(back_lds, (size_t)A.N * A.N * 8 * sizeof(double));
        if (c->r_idepth_dirty) { cml_refresh_r_idepth(c, A, c0->stream); c->r_idepth_dirty = false; }
        int blocks = 0;
        if (int rc = cml_fill_rs4_batch(c, A, Hrs, blocks)) { c0->err = c->err; return rc; }
        if (c->rs_tile != c0->rs_tile) { c0->err = "cmlhip_ba_iteration_batch: the windows of a batch must be uploaded in one residual-kernel regime (cmlhip_ba_set_window_regime)"; return CMLHIP_ERR_INVALID; }
        rs_blocks = std::max(rs_blocks, blocks);
        c->sys_lambda = lambda;
    }
    if (g_back == 0 || g_acc == 0) { c0->err = "cmlhip_ba_iteration_batch: empty windows"; return CMLHIP_ERR_INVALID; }
    // every window's Jacobians in tile form (every round but the first of a loop that starts from records): K3 at 256 threads (k_ba_acc_rs_body)
    static const bool acc16 = getenv("CMLHIP_ACC_1024") != nullptr;
    const bool acc_rs = all_tiles && !acc16;
    if (acc_rs) {
        g_acc = 0;
        for (int k = 0; k < S; k++) { H[k].g_acc = H[k].A.N * H[k].A.N + cml_div_up(H[k].A.P, PT_PER_BLOCK_RS); g_acc = std::max(g_acc, H[k].g_acc); }
    }
    // argument blocks -> device (only when something changed: the first iterations of a loop, a new lambda, a new window)
    const size_t bytes_main = sizeof(BatchWin) * (size_t)S;
    int rc;
    if ((rc = cml_ensure(c0, c0->batch_main, bytes_main))) return rc;
    if ((rc = cml_ensure(c0, c0->batch_rs, Hrs.size()))) return rc;
    if (c0->batch_main_host.size() != bytes_main || memcmp(c0->batch_main_host.data(), H.data(), bytes_main) != 0) {
        c0->batch_main_host.assign(reinterpret_cast<unsigned char*>(H.data()), reinterpret_cast<unsigned char*>(H.data()) + bytes_main);
        CML_CHECK(c0, hipMemcpyAsync(c0->batch_main.p, c0->batch_main_host.data(), bytes_main, hipMemcpyHostToDevice, c0->stream));
    }
    if (c0->batch_rs_host != Hrs) {
        c0->batch_rs_host = Hrs;
        CML_CHECK(c0, hipMemcpyAsync(c0->batch_rs.p, c0->batch_rs_host.data(), Hrs.size(), hipMemcpyHostToDevice, c0->stream));
    }
    const BatchWin* W = c0->batch_main.as<BatchWin>();
    if (acc_rs) k_ba_acc_rs_batch<<<dim3(g_acc, S), 256, 0, c0->stream>>>(W);          // K3
    else k_ba_acc_batch<<<dim3(g_acc, S), 1024, 0, c0->stream>>>(W);
    k_ba_system_batch<<<dim3(g_sys, S), 64 * SYS_NW, 0, c0->stream>>>(W);              // K4
#define LAUNCH_SOLVE_B(NSL) do { \
        if (!(c0->attr_done_batch & (1u << NSL))) { (void)hipFuncSetAttribute((const void*)k_ba_solve_batch<NSL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); c0->attr_done_batch |= 1u << NSL; } \
        k_ba_solve_batch<NSL><<<dim3(2, S), SOLVE_THREADS, solve_lds, c0->stream>>>(W); } while (0)
    switch (nsl) {                                                                     // K5: S solve workgroups side by side (+ S threshold workgroups)
        case 1: LAUNCH_SOLVE_B(1); break;
        case 2: LAUNCH_SOLVE_B(2); break;
        case 4: LAUNCH_SOLVE_B(4); break;
        default: LAUNCH_SOLVE_B(8); break;
    }
#undef LAUNCH_SOLVE_B
    k_ba_backsub_batch<<<dim3(g_back, S), 256, back_lds, c0->stream>>>(W);             // K6
    if ((rc = (c0->rs_tile == 16 ? cml_launch_linearize_rs4_batch : cml_launch_linearize_rs_batch)(c0, c0->batch_rs.p, S, rs_blocks))) return rc;   // K1
    CML_CHECK(c0, hipGetLastError());
    for (int k = 0; k < S; k++) {
        cmlhip_ctx* c = ctxs[k];
        c->resident_iter++;
        c->efs_in_partials = true; c->lin_partial_n = c->n_tiles;
        c->lin_finish_pending = true;
        c->last_lambda = lambda;
    }
    return CMLHIP_OK;
}
