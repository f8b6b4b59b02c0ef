// ba_accumulate.hip — Hessian accumulation, point Schur complement, dense solve and back-substitution of the
// sliding-window BA.  Replaces addToHessianTop + AccumulatorApprox (BA.cpp:1648-1779, ACC.h:613-998),
// stitchDoubleTop (BA.cpp:1781-1878), addToHessianSC + stitchDoubleSC (BA.cpp:1880-2043),
// solveLevenbergMarquardt's factorisation (BA.cpp:1284-1320) and the resubstitution loop (BA.cpp:1427-1487).
//
// Design (MI355X-first, not a translation):
//  * top: residuals are grouped by (host,target) on upload; one workgroup per pair keeps the 91 unique entries
//    of the 13x13 block in registers per lane, reduces them with wave shuffles + one LDS hop, and immediately
//    applies the fp64 adjoint sandwiches (AH B AH^T, ...) so the 13x13 never leaves the CU in fp32 only.
//  * Schur: the reference buckets rank-1 8x8 tiles into N^3 accumulators and then does an O(N^3) stitch.  Here
//    each point's coupling row g_p = dH/d(idepth) is formed directly in frame coordinates (AH/AT applied per
//    residual, 8 lanes per point), and H_sc = G^T diag(HdiF) [G | bdSum] is one fp64 SYRK on the matrix cores
//    (v_mfma_f64_16x16x4_f64), K-split over point chunks with a fixed-order reduction.  Same sums, no N^3 pass.
//  * solve: Jacobi-scaled LDL^T of the trailing 8N block in LDS (fp64, packed lower storage), one workgroup.
#include "cmlhip_internal.h"
#include "ba_common.h"

typedef double double4_ __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------------ top
// grid = N*N workgroups (pair q = host + target*N, BA.cpp:1677), 256 threads.
template <bool LIN>
__global__ __launch_bounds__(256) void k_ba_acc_top(BAArgs A, const double* __restrict__ adH, const double* __restrict__ adT,
                                                    const float* __restrict__ adHTd, const double* __restrict__ cdelta,
                                                    float* __restrict__ acc_out, int* __restrict__ num_out,
                                                    double* __restrict__ pair_blocks) {
    __shared__ float s_red[4][ACC_STRIDE];
    __shared__ int s_cnt[4];
    __shared__ double s_H[13][13];
    __shared__ double s_AH[64], s_AT[64], s_T1[64], s_T2[64];
    const int q = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    float acc[91];
#pragma unroll
    for (int i = 0; i < 91; i++) acc[i] = 0.f;
    int cnt = 0;
    const int beg = A.by_pair_off[q], end = A.by_pair_off[q + 1];
    for (int i = beg + tid; i < end; i += 256) {
        const int r = A.by_pair[i];
        const bool lin = A.r_lin[r] != 0;
        if (LIN ? (!lin || !A.r_good[r]) : (lin || !A.r_good[r])) continue;      // BA.cpp:1662-1669
        const float* J = (A.r_sel[r] ? A.rj1 : A.rj0) + (size_t)r * RJ_STRIDE;    // efsJ
        float x[10], y[10];
#pragma unroll
        for (int j = 0; j < 4; j++) { x[j] = J[O_C0 + j]; y[j] = J[O_C1 + j]; }
#pragma unroll
        for (int j = 0; j < 6; j++) { x[4 + j] = J[O_XI0 + j]; y[4 + j] = J[O_XI1 + j]; }
        const float a = J[O_JI2 + 0], b = J[O_JI2 + 2], c = J[O_JI2 + 3];
        float JIr0, JIr1, Jabr0, Jabr1, rr;
        if (!LIN) {
            JIr0 = J[O_X_JIR]; JIr1 = J[O_X_JIR + 1]; Jabr0 = J[O_X_JABR]; Jabr1 = J[O_X_JABR + 1]; rr = J[O_X_RR];
        } else {
            // BA.cpp:1699-1729 (res_toZero + J*delta; see oracle note on the reference's float*/double[8] store)
            const float* dp = adHTd + 8 * q;
            const int p = A.r_point[r];
            const float dd = (float)(A.pt_idepth[p] - (double)A.pt_idepth_zero[p]);
            float jdx = 0, jdy = 0, cx = 0, cy = 0;
            for (int j = 0; j < 6; j++) { jdx += J[O_XI0 + j] * dp[j]; jdy += J[O_XI1 + j] * dp[j]; }
            for (int j = 0; j < 4; j++) { cx += J[O_C0 + j] * (float)cdelta[j]; cy += J[O_C1 + j] * (float)cdelta[j]; }
            const float Jpx = jdx + cx + J[O_DD] * dd, Jpy = jdy + cy + J[O_DD + 1] * dd;
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0; float srr = 0;
            for (int j = 0; j < 8; j++) {
                float rtz = A.r_rtz[8 * (size_t)r + j];
                rtz = rtz + J[O_JI0 + j] * Jpx; rtz = rtz + J[O_JI1 + j] * Jpy;
                rtz = rtz + J[O_JAB0 + j] * dp[6]; rtz = rtz + J[O_JAB1 + j] * dp[7];
                const double ra = (double)rtz;
                s0 += ra * (double)J[O_JI0 + j]; s1 += ra * (double)J[O_JI1 + j];
                s2 += ra * (double)J[O_JAB0 + j]; s3 += ra * (double)J[O_JAB1 + j];
                srr = (float)((double)srr + ra * ra);
            }
            JIr0 = (float)s0; JIr1 = (float)s1; Jabr0 = (float)s2; Jabr1 = (float)s3; rr = srr;
        }
        // AccumulatorApprox::update, ACC.h:776-858: 10x10 upper triangle of [x y][a b; b c][x y]^T
        int idx = 0;
#pragma unroll
        for (int rr_ = 0; rr_ < 10; rr_++)
#pragma unroll
            for (int cc = rr_; cc < 10; cc++) {
                acc[idx] += a * x[cc] * x[rr_] + c * y[cc] * y[rr_] + b * (x[cc] * y[rr_] + y[cc] * x[rr_]);
                idx++;
            }
        // updateTopRight, ACC.h:861-916 (TR00,TR10 = JabJIdx(0,0),(0,1); TR01,TR11 = (1,0),(1,1); TR02,TR12 = JI^T r)
        const float TR00 = J[O_JABJI + 0], TR10 = J[O_JABJI + 2], TR01 = J[O_JABJI + 1], TR11 = J[O_JABJI + 3];
#pragma unroll
        for (int j = 0; j < 10; j++) {
            acc[55 + 3 * j + 0] += x[j] * TR00 + y[j] * TR10;
            acc[55 + 3 * j + 1] += x[j] * TR01 + y[j] * TR11;
            acc[55 + 3 * j + 2] += x[j] * JIr0 + y[j] * JIr1;
        }
        // updateBotRight, ACC.h:918-932
        acc[85] += J[O_JAB2 + 0]; acc[86] += J[O_JAB2 + 2]; acc[87] += Jabr0;
        acc[88] += J[O_JAB2 + 3]; acc[89] += Jabr1; acc[90] += rr;
        cnt++;
    }
#pragma unroll
    for (int i = 0; i < 91; i++) {
        const float v = wave_sum(acc[i]);
        if (ln == 0) s_red[wv][i] = v;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (ln == 0) s_cnt[wv] = cnt;
    if (tid < 64) { s_AH[tid] = adH[64 * (size_t)q + tid]; s_AT[tid] = adT[64 * (size_t)q + tid]; }
    __syncthreads();
    if (tid < 91) {
        const float v = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
        acc_out[(size_t)q * ACC_STRIDE + tid] = v;
        // scatter into the symmetric 13x13 (AccumulatorApprox::finish, ACC.h:639-673)
        int rr_, cc;
        if (tid < 55) {
            int k = tid; rr_ = 0;
            while (k >= 10 - rr_) { k -= 10 - rr_; rr_++; }
            cc = rr_ + k;
        } else if (tid < 85) {
            rr_ = (tid - 55) / 3; cc = 10 + (tid - 55) % 3;
        } else {
            const int m[6][2] = {{10, 10}, {10, 11}, {10, 12}, {11, 11}, {11, 12}, {12, 12}};
            rr_ = m[tid - 85][0]; cc = m[tid - 85][1];
        }
        s_H[rr_][cc] = (double)v; s_H[cc][rr_] = (double)v;
    }
    if (tid == 0) num_out[q] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __syncthreads();
    // ---- stitchDoubleTop per-pair products, BA.cpp:1827-1843 (fp64)
    double* pb = pair_blocks + (size_t)q * PB_STRIDE;
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
}

// assemble (8N+4)^2 from the per-pair blocks, incl. priors and the symmetrisation of BA.cpp:1857-1876.
// one thread per output element (+ n threads for b).  use_blocks = 0 gives the prior-only matrix.
__global__ void k_ba_assemble_top(int N, const double* __restrict__ pb, int use_blocks, int use_prior,
                                  const double* __restrict__ cdelta, const double* __restrict__ cprior,
                                  const double* __restrict__ prior, const double* __restrict__ dprior,
                                  double* __restrict__ H, double* __restrict__ bvec) {
    const int n = 8 * N + 4;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * n + n) return;
    if (e >= n * n) {                                   // b
        const int I = e - n * n;
        double s = 0;
        if (I < 4) {
            if (use_blocks) for (int q = 0; q < N * N; q++) s += pb[(size_t)q * PB_STRIDE + PB_BC + I];
            if (use_prior) s += cprior[I] * cdelta[I];
        } else {
            const int a = (I - 4) >> 3, i = (I - 4) & 7;
            if (use_blocks) {
                for (int t = 0; t < N; t++) s += pb[(size_t)(a + t * N) * PB_STRIDE + PB_BH + i];
                for (int h = 0; h < N; h++) s += pb[(size_t)(h + a * N) * PB_STRIDE + PB_BT + i];
            }
            if (use_prior) s += prior[8 * a + i] * dprior[8 * a + i];
        }
        bvec[I] = s;
        return;
    }
    int I = e / n, Jc = e % n;
    double s = 0;
    if (I < 4 && Jc < 4) {
        if (use_blocks) for (int q = 0; q < N * N; q++) s += pb[(size_t)q * PB_STRIDE + PB_CC + I * 4 + Jc];
        if (use_prior && I == Jc) s += cprior[I];
    } else if (I < 4 || Jc < 4) {
        const int F = I < 4 ? Jc : I, C = I < 4 ? I : Jc;       // frame row, calib column (mirrored, :1869)
        const int a = (F - 4) >> 3, i = (F - 4) & 7;
        if (use_blocks) {
            for (int t = 0; t < N; t++) s += pb[(size_t)(a + t * N) * PB_STRIDE + PB_HC + i * 4 + C];
            for (int h = 0; h < N; h++) s += pb[(size_t)(h + a * N) * PB_STRIDE + PB_TC + i * 4 + C];
        }
    } else {
        const int a = (I - 4) >> 3, i = (I - 4) & 7, b = (Jc - 4) >> 3, j = (Jc - 4) & 7;
        if (a == b) {
            if (use_blocks) {
                for (int t = 0; t < N; t++) s += pb[(size_t)(a + t * N) * PB_STRIDE + PB_HH + i * 8 + j];
                for (int h = 0; h < N; h++) s += pb[(size_t)(h + a * N) * PB_STRIDE + PB_TT + i * 8 + j];
            }
            if (use_prior && i == j) s += prior[8 * a + i];
        } else if (use_blocks) {
            s = pb[(size_t)(a + b * N) * PB_STRIDE + PB_HT + i * 8 + j] + pb[(size_t)(b + a * N) * PB_STRIDE + PB_HT + j * 8 + i];
        }
    }
    H[(size_t)I * n + Jc] = s;
}

// ------------------------------------------------------------------------------------------------ Schur rows
// 8 lanes per point.  Per point: Hdd/bd/Hcd sums (BA.cpp:1747-1750), HdiF, bdSum (BA.cpp:1895-1905); row
// g_p[0:4] = Hcd, g_p[4+8h+i] = sum_r (AH_ht JpJdF_r)_i, g_p[4+8t+i] = (AT_ht JpJdF_r)_i, G[p][n] = bdSum.
__global__ __launch_bounds__(256) void k_ba_point_schur(BAArgs A, const double* __restrict__ adH, const double* __restrict__ adT,
                                                        double* __restrict__ G, double* __restrict__ Wt, int ldg) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int p = gid >> 3, i = gid & 7;
    if (p >= A.P) return;
    const int host = A.pt_host[p];
    const int beg = A.by_point_off[p], end = A.by_point_off[p + 1];
    double* row = G + (size_t)p * ldg;
    // zero-fill: every lane clears exactly the columns it may write below, so store order is per-lane program order
    for (int f = 0; f < A.N; f++) row[4 + 8 * f + i] = 0.0;
    if (i < 4) row[i] = 0.0;
    if (i == 4) row[A.n] = 0.0;
    if (i == 5) for (int cix = A.n + 1; cix < ldg; cix++) row[cix] = 0.0;
    float HddA = 0, bdA = 0, HcdA[4] = {0, 0, 0, 0}, HddL = 0, bdL = 0, HcdL[4] = {0, 0, 0, 0};
    int ngood = 0;
    double hostacc = 0;
    for (int kk = beg; kk < end; kk++) {
        const int r = A.by_point[kk];
        if (!A.r_good[r]) continue;
        ngood++;
        const float* J = (A.r_sel[r] ? A.rj1 : A.rj0) + (size_t)r * RJ_STRIDE;
        const int t = A.r_target[r];
        const int q = host + t * A.N;
        // per-point scalars: ACTIVE residuals use the linearize-time JI^T r; LINEARIZED ones are folded by the host path
        const float g0 = J[O_JI2 + 0] * J[O_DD] + J[O_JI2 + 2] * J[O_DD + 1];
        const float g1 = J[O_JI2 + 1] * J[O_DD] + J[O_JI2 + 3] * J[O_DD + 1];
        if (!A.r_lin[r]) {
            bdA = (float)((double)bdA + ((double)J[O_X_JIR] * (double)J[O_DD] + (double)J[O_X_JIR + 1] * (double)J[O_DD + 1]));
            HddA += g0 * J[O_DD] + g1 * J[O_DD + 1];
#pragma unroll
            for (int j = 0; j < 4; j++) HcdA[j] += J[O_C0 + j] * g0 + J[O_C1 + j] * g1;
        } else {
            // BA.cpp:1699-1729 in LINEARIZED mode: JI^T r with r = res_toZero + J*delta is recomputed here
            HddL += g0 * J[O_DD] + g1 * J[O_DD + 1];
#pragma unroll
            for (int j = 0; j < 4; j++) HcdL[j] += J[O_C0 + j] * g0 + J[O_C1 + j] * g1;
            // bdL needs adHTdeltaF/cdelta; it is accumulated by k_ba_point_bdL (rare path)
        }
        const float* v = A.r_jpjdf + 8 * (size_t)r;
        const double* AH = adH + 64 * (size_t)q + 8 * i;
        double ah = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) ah += AH[j] * (double)v[j];
        hostacc += ah;
        // AT is diagonal by construction (BA.cpp:1078-1092) but is applied as a full row for generality
        const double* AT = adT + 64 * (size_t)q + 8 * i;
        double at = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) at += AT[j] * (double)v[j];
        row[4 + 8 * t + i] = at;
    }
    float* pa = A.pt_acc + (size_t)p * PT_ACC_STRIDE;
    float HdiF = 0.f, bdSum = 0.f;
    if (ngood > 0) {
        const float bdLv = pa[7];                          // written by the LINEARIZED pre-pass (0 when none)
        float H = HddA + HddL + A.pt_prior[p];
        if (H < 1e-10) H = 1e-10;
        HdiF = (float)(1.0 / H);
        bdSum = bdA + bdLv;
        const float deltaF = (float)(A.pt_idepth[p] - (double)A.pt_idepth_zero[p]);
        bdSum += A.pt_prior[p] * deltaF;                   // shiftPriorToZero, :1904
        row[4 + 8 * host + i] = hostacc;
        if (i < 4) row[i] = (double)(HcdA[i] + HcdL[i]);
        if (i == 4) row[A.n] = (double)bdSum;
    }
    if (i == 0) {
        pa[0] = HddA; pa[1] = bdA; pa[2] = HcdA[0]; pa[3] = HcdA[1]; pa[4] = HcdA[2]; pa[5] = HcdA[3];
        pa[6] = HddL; pa[8] = HcdL[0]; pa[9] = HcdL[1]; pa[10] = HcdL[2]; pa[11] = HcdL[3];
        pa[12] = HdiF; pa[13] = bdSum;
        Wt[p] = (double)HdiF;
    }
}

// H_aug = G^T diag(w) [G | bdSum] : one wave per (tile_i <= tile_j, point chunk); fp64 matrix cores.
// A[i][k] = G[p0+k][i0+i], B[k][j] = w[p0+k] G[p0+k][j0+j]; D: col = lane&15, row = (lane>>4) + 4*reg.
#define SYRK_CHUNK 64
__global__ __launch_bounds__(64) void k_ba_schur_syrk(const double* __restrict__ G, const double* __restrict__ Wt, int P,
                                                      int ldg, int ntile, double* __restrict__ part) {
    const int tile = blockIdx.x, chunk = blockIdx.y, l = threadIdx.x;
    // unrank the upper-triangular tile index
    int ti = 0, rem = tile;
    while (rem >= ntile - ti) { rem -= ntile - ti; ti++; }
    const int tj = ti + rem;
    const int p0 = chunk * SYRK_CHUNK;
    double4_ acc = {0.0, 0.0, 0.0, 0.0};
    const int kk = l >> 4, c = l & 15;
#pragma unroll 4
    for (int s = 0; s < SYRK_CHUNK; s += 4) {
        const int p = p0 + s + kk;
        double a = 0.0, b = 0.0;
        if (p < P) {
            const double* row = G + (size_t)p * ldg;
            a = row[16 * ti + c];
            b = Wt[p] * row[16 * tj + c];
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    double* o = part + ((size_t)chunk * gridDim.x + tile) * 256;
#pragma unroll
    for (int rg = 0; rg < 4; rg++) o[(kk + 4 * rg) * 16 + c] = acc[rg];
}

// fixed-order reduction of the chunk partials into H_sc (mirrored) and b_sc
__global__ void k_ba_schur_finish(const double* __restrict__ part, int nchunk, int ntile, int ntiles_ut, int n,
                                  double* __restrict__ Hsc, double* __restrict__ bsc) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * (n + 1)) return;
    const int I = e / (n + 1), Jc = e % (n + 1);
    int r = I, c = Jc;
    if (Jc < n && Jc < I) { r = Jc; c = I; }           // lower triangle: read the mirrored element
    const int ti = r >> 4, tj = c >> 4;
    // rank of (ti,tj), ti <= tj
    int tile = 0;
    for (int k = 0; k < ti; k++) tile += ntile - k;
    tile += tj - ti;
    const int off = (r & 15) * 16 + (c & 15);
    double s = 0;
    for (int ch = 0; ch < nchunk; ch++) s += part[((size_t)ch * ntiles_ut + tile) * 256 + off];
    if (Jc == n) bsc[I] = s;
    else Hsc[(size_t)I * n + Jc] = s;
}

// ------------------------------------------------------------------------------------------------ solve
// One workgroup.  H = HL + HM + HA, diag*(1+lambda), - Hsc/(1+lambda); S = 1/sqrt(diag+10); LDL^T (no pivoting:
// the scaled matrix is SPD with unit-order diagonal) of rows/cols [off, n) in packed lower LDS storage.
__global__ __launch_bounds__(256) void k_ba_solve(int n, int off, double lambda, const double* __restrict__ HA,
                                                  const double* __restrict__ bA, const double* __restrict__ HL,
                                                  const double* __restrict__ bL, const double* __restrict__ HM,
                                                  const double* __restrict__ bM, const double* __restrict__ Hsc,
                                                  const double* __restrict__ bsc, double* __restrict__ x, int* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int m = n - off, tid = threadIdx.x;
    double* L = sm;                              // m(m+1)/2
    double* S = L + (size_t)m * (m + 1) / 2;     // m
    double* y = S + m;                           // m
    const double f = 1.0 / (1 + lambda);
#define LT(i, j) L[(size_t)(i) * ((i) + 1) / 2 + (j)]
    for (int i = tid; i < m; i += 256) {
        const size_t d = (size_t)(off + i) * n + off + i;
        double h = (HL[d] + (HM ? HM[d] : 0.0)) + HA[d];
        h *= (1 + lambda);
        h -= Hsc[d] * f;
        S[i] = 1.0 / sqrt(h + 10.0);
    }
    __syncthreads();
    for (int e = tid; e < m * m; e += 256) {
        const int i = e / m, j = e % m;
        if (j > i) continue;
        const size_t d = (size_t)(off + i) * n + off + j;
        double h = (HL[d] + (HM ? HM[d] : 0.0)) + HA[d];
        if (i == j) h *= (1 + lambda);
        h -= Hsc[d] * f;
        LT(i, j) = S[i] * h * S[j];
    }
    for (int i = tid; i < m; i += 256) {
        const int I = off + i;
        y[i] = S[i] * (((bL[I] + (bM ? bM[I] : 0.0)) + bA[I]) - bsc[I]);
    }
    __syncthreads();
    // right-looking LDL^T: after step k column k holds l_ik, LT(k,k) holds d_k
    const int tx = tid & 15, ty = tid >> 4;
    for (int k = 0; k < m; k++) {
        const double d = LT(k, k);
        const double dinv = 1.0 / d;
        for (int i = k + 1 + ty; i < m; i += 16) {
            const double ci = LT(i, k);
            for (int j = k + 1 + tx; j <= i; j += 16) LT(i, j) -= ci * LT(j, k) * dinv;
        }
        __syncthreads();
        for (int i = k + 1 + tid; i < m; i += 256) LT(i, k) *= dinv;
        __syncthreads();
    }
    // forward substitution L z = y
    for (int k = 0; k < m; k++) {
        const double yk = y[k];
        for (int i = k + 1 + tid; i < m; i += 256) y[i] -= LT(i, k) * yk;
        __syncthreads();
    }
    for (int i = tid; i < m; i += 256) {
        const double d = LT(i, i);
        y[i] = (fabs(d) > 2.2250738585072014e-308) ? y[i] / d : 0.0;     // Eigen LDLT.h:580-587 pseudo-inverse of D
    }
    __syncthreads();
    // back substitution L^T x = z
    for (int k = m - 1; k >= 0; k--) {
        const double xk = y[k];
        for (int i = tid; i < k; i += 256) y[i] -= LT(k, i) * xk;
        __syncthreads();
    }
    int bad = 0;
    for (int i = tid; i < n; i += 256) {
        const double v = (i < off) ? 0.0 : S[i - off] * y[i - off];
        x[i] = v;
        bad |= !isfinite(v);
    }
    if (bad) atomicOr(flag, 1);
#undef LT
}

// ------------------------------------------------------------------------------------------------ back-substitution
__global__ __launch_bounds__(256) void k_ba_backsub(BAArgs A, const double* __restrict__ adH, const double* __restrict__ adT,
                                                    const double* __restrict__ x, LinSummary* __restrict__ sum) {
    extern __shared__ __attribute__((aligned(16))) double s_xAd[];     // N*N*8, index (host*N + target)*8 + j  (:1447)
    const int N = A.N;
    for (int e = threadIdx.x; e < N * N * 8; e += blockDim.x) {
        const int j = e & 7, ht = e >> 3, h = ht / N, t = ht % N;
        const double* AH = adH + 64 * (size_t)(h + N * t); const double* AT = adT + 64 * (size_t)(h + N * t);
        double s = 0, s2 = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { s += x[4 + 8 * h + i] * AH[i * 8 + j]; s2 += x[4 + 8 * t + i] * AT[i * 8 + j]; }
        s_xAd[e] = s + s2;
    }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.P) return;
    const float* pa = A.pt_acc + (size_t)p * PT_ACC_STRIDE;
    const int beg = A.by_point_off[p], end = A.by_point_off[p + 1];
    int ngood = 0;
    for (int kk = beg; kk < end; kk++) ngood += A.r_good[A.by_point[kk]] != 0;
    if (ngood == 0) { A.pt_step[p] = 0.0; return; }
    double b = (double)pa[13];
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) s += (-x[i]) * ((double)pa[2 + i] + (double)pa[8 + i]);     // mCalibStep . (Hcd_accAF + Hcd_accLF)
    b -= s;
    const int host = A.pt_host[p];
    for (int kk = beg; kk < end; kk++) {
        const int r = A.by_point[kk];
        if (!A.r_good[r]) continue;
        const double* xa = s_xAd + 8 * (host * N + A.r_target[r]);
        const float* v = A.r_jpjdf + 8 * (size_t)r;
        double d = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d += xa[i] * (double)v[i];
        b -= d;
    }
    const double st = -b * (double)pa[12];
    A.pt_step[p] = st;
    if (!isfinite(st)) atomicAdd(&sum->nonfinite, 1);
}

__global__ void k_ba_backup_points(BAArgs A) {          // BA.cpp:919-922
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < A.P) A.pt_backup[p] = (float)A.pt_idepth[p];
}

__global__ void k_ba_restore_points(BAArgs A) {         // loadSateBackup, BA.cpp:938-942
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < A.P) { A.pt_idepth[p] = (double)A.pt_backup[p]; A.pt_idepth_zero[p] = A.pt_backup[p]; }
}

// doStepFromBackup, point part (BA.cpp:976-994); sums are tree-reduced (float)
__global__ __launch_bounds__(1024) void k_ba_step_points(BAArgs A, LinSummary* __restrict__ sum) {
    __shared__ float s[3][16];
    float sumID = 0, sumNID = 0, numID = 0;
    for (int p = threadIdx.x; p < A.P; p += 1024) {
        const double st = A.pt_step[p];
        const double nid = (double)A.pt_backup[p] + st;
        if (isfinite(nid) && nid > 0) {
            A.pt_idepth[p] = nid;
            sumID += (float)(st * st);
            sumNID += (float)fabs((double)A.pt_backup[p]);
            numID += 1.f;
            A.pt_idepth_zero[p] = (float)nid;
        }
    }
    sumID = wave_sum(sumID); sumNID = wave_sum(sumNID); numID = wave_sum(numID);
    if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = sumID; s[1][threadIdx.x >> 6] = sumNID; s[2][threadIdx.x >> 6] = numID; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0, b = 0, c = 0;
        for (int i = 0; i < 16; i++) { a += s[0][i]; b += s[1][i]; c += s[2][i]; }
        sum->sums[0] = a; sum->sums[1] = b; sum->sums[2] = c;
    }
}

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
        float jdx = 0, jdy = 0, cx = 0, cy = 0;
        for (int j = 0; j < 6; j++) { jdx += J[O_XI0 + j] * dp[j]; jdy += J[O_XI1 + j] * dp[j]; }
        for (int j = 0; j < 4; j++) { cx += J[O_C0 + j] * (float)cdelta[j]; cy += J[O_C1 + j] * (float)cdelta[j]; }
        const float Jpx = jdx + cx + J[O_DD] * dd, Jpy = jdy + cy + J[O_DD + 1] * dd;
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

// ------------------------------------------------------------------------------------------------ launchers
static inline int ldg_of(int n) { return ((n + 1 + 15) / 16) * 16; }

int cml_launch_accumulate(cmlhip_ctx* c, const BAArgs& A) {
    const int N = A.N, n = A.n, NN = N * N;
    const double* vs = c->vec_small.as<double>();        // cdelta[4] cprior[4] prior[8N] dprior[8N]
    const double* cdelta = vs; const double* cprior = vs + 4; const double* prior = vs + 8; const double* dprior = vs + 8 + 8 * N;
    const int asm_threads = n * n + n;
    // ACTIVE
    k_ba_acc_top<false><<<NN, 256, 0, c->stream>>>(A, c->adH.as<double>(), c->adT.as<double>(), c->adHTd.as<float>(), cdelta,
                                                   c->acc_pair[0].as<float>(), c->acc_num[0].as<int>(), c->pair_blocks.as<double>());
    k_ba_assemble_top<<<cml_div_up(asm_threads, 256), 256, 0, c->stream>>>(N, c->pair_blocks.as<double>(), 1, 0, cdelta, cprior, prior,
                                                                            dprior, c->HA.as<double>(), c->bA.as<double>());
    // LINEARIZED (prior-only when the window holds no linearized residual)
    if (c->n_lin > 0) {
        k_ba_acc_top<true><<<NN, 256, 0, c->stream>>>(A, c->adH.as<double>(), c->adT.as<double>(), c->adHTd.as<float>(), cdelta,
                                                      c->acc_pair[1].as<float>(), c->acc_num[1].as<int>(), c->pair_blocks.as<double>());
        k_ba_point_bdL<<<cml_div_up(A.P, 256), 256, 0, c->stream>>>(A, c->adHTd.as<float>(), cdelta);
    }
    k_ba_assemble_top<<<cml_div_up(asm_threads, 256), 256, 0, c->stream>>>(N, c->pair_blocks.as<double>(), c->n_lin > 0 ? 1 : 0, 1, cdelta,
                                                                            cprior, prior, dprior, c->HL.as<double>(), c->bL.as<double>());
    // Schur
    const int ldg = ldg_of(n), ntile = ldg / 16, ntiles_ut = ntile * (ntile + 1) / 2;
    const int nchunk = cml_div_up(A.P, SYRK_CHUNK);
    double* Wt = c->G.as<double>() + (size_t)A.P * ldg;
    k_ba_point_schur<<<cml_div_up(A.P * 8, 256), 256, 0, c->stream>>>(A, c->adH.as<double>(), c->adT.as<double>(), c->G.as<double>(), Wt, ldg);
    k_ba_schur_syrk<<<dim3(ntiles_ut, nchunk), 64, 0, c->stream>>>(c->G.as<double>(), Wt, A.P, ldg, ntile, c->syrk_part.as<double>());
    k_ba_schur_finish<<<cml_div_up(n * (n + 1), 256), 256, 0, c->stream>>>(c->syrk_part.as<double>(), nchunk, ntile, ntiles_ut, n,
                                                                            c->Hsc.as<double>(), c->bsc.as<double>());
    return CMLHIP_OK;
}

int cml_launch_solve(cmlhip_ctx* c, const BAArgs& A, double lambda, bool have_hm, int optcal) {
    const int n = A.n, off = optcal ? 0 : 4, m = n - off;
    const size_t sh = ((size_t)m * (m + 1) / 2 + 2 * (size_t)m) * sizeof(double);
    int* flag = reinterpret_cast<int*>(c->scal.as<char>() + 256);
    hipMemsetAsync(flag, 0, sizeof(int), c->stream);
    if (sh > 64 * 1024) hipFuncSetAttribute((const void*)k_ba_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    k_ba_solve<<<1, 256, sh, c->stream>>>(n, off, lambda, c->HA.as<double>(), c->bA.as<double>(), c->HL.as<double>(), c->bL.as<double>(),
                                          have_hm ? c->HM.as<double>() : nullptr, have_hm ? c->bM.as<double>() : nullptr,
                                          c->Hsc.as<double>(), c->bsc.as<double>(), c->xvec.as<double>(), flag);
    return CMLHIP_OK;
}

int cml_launch_backsub(cmlhip_ctx* c, const BAArgs& A) {
    LinSummary* S = c->scal.as<LinSummary>();
    hipMemsetAsync(&S->nonfinite, 0, sizeof(int), c->stream);
    const size_t sh = (size_t)A.N * A.N * 8 * sizeof(double);
    k_ba_backsub<<<cml_div_up(A.P, 256), 256, sh, c->stream>>>(A, c->adH.as<double>(), c->adT.as<double>(), c->xvec.as<double>(), S);
    return CMLHIP_OK;
}
int cml_launch_backup_points(cmlhip_ctx* c, const BAArgs& A) {
    k_ba_backup_points<<<cml_div_up(A.P, 256), 256, 0, c->stream>>>(A);
    return CMLHIP_OK;
}
int cml_launch_restore_points(cmlhip_ctx* c, const BAArgs& A) {
    k_ba_restore_points<<<cml_div_up(A.P, 256), 256, 0, c->stream>>>(A);
    return CMLHIP_OK;
}
int cml_launch_step_points(cmlhip_ctx* c, const BAArgs& A) {
    k_ba_step_points<<<1, 1024, 0, c->stream>>>(A, c->scal.as<LinSummary>());
    return CMLHIP_OK;
}
