// ba_finish.h — the single-workgroup tail of a residual pass: fp64 energy sum + state census from the per-block
// partials of k_ba_linearize (BA.cpp:1565,1608), and setNewFrameEnergyTH (BA.cpp:2419-2464) as an EXACT radix select
// (4 x 8-bit passes over the float bit patterns; energies are >= 0, so unsigned order == float order).
// Included by ba_linearize.hip (standalone kernel) and ba_accumulate.hip (second workgroup of the solve launch).
#pragma once
#include "ba_common.h"

// s_u32: >= 264 words, s_f64: >= blockDim.x doubles (LDS)
__device__ __forceinline__ void lin_finish_block(const BAArgs& A, const int* newframe_res, int n_newframe, const double* lin_partial,
                                                 int n_partial, LinSummary* out, FrameDev* frames_rw, unsigned* s_u32, double* s_f64) {
    const int tid = threadIdx.x, nt = blockDim.x;
    unsigned* s_hist = s_u32;                 // 256
    unsigned* s_misc = s_u32 + 256;           // [0] prefix [1] k [2] nvalid
    // ---- fixed-order sums of the per-block partials {energy, n_in, n_oob, n_outlier}
    double e = 0, c0 = 0, c1 = 0, c2 = 0;
    for (int b = tid; b < n_partial; b += nt) {
        e += lin_partial[4 * (size_t)b]; c0 += lin_partial[4 * (size_t)b + 1];
        c1 += lin_partial[4 * (size_t)b + 2]; c2 += lin_partial[4 * (size_t)b + 3];
    }
    // (round 4: the four sums through ONE exchange — a fixed butterfly inside each wave, then the waves' sums added in wave order by every
    //  thread — instead of four ten-level trees of workgroup barriers; the order is fixed by the block size alone)
    double tot[4];
    double vals[4] = {e, c0, c1, c2};
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vals[k] += __shfl_xor(vals[k], o);
    }
    const int nw = nt >> 6;
    if ((tid & 63) == 0) { for (int k = 0; k < 4; k++) s_f64[4 * (tid >> 6) + k] = vals[k]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) { double t_ = 0; for (int w = 0; w < nw; w++) t_ += s_f64[4 * w + k]; tot[k] = t_; }
    __syncthreads();
    // ---- number of valid candidates (residuals into the newest frame with NewEnergyWithOutlier >= 0)
    if (tid == 0) s_misc[2] = 0;
    __syncthreads();
    // the candidates of this thread, fetched ONCE (bit patterns; 0xFFFFFFFF = not a candidate: a negative or NaN energy, a LINEARIZED residual) —
    // the four passes of the select below then run from registers; candidates beyond LF_CACHE per thread are re-read as before
    constexpr int LF_CACHE = 4;
    unsigned cand[LF_CACHE];
    unsigned nvalid = 0;
#pragma unroll
    for (int q = 0; q < LF_CACHE; q++) {
        const int i = tid + q * nt;
        unsigned b = 0xFFFFFFFFu;
        if (i < n_newframe) {
            const int r = newframe_res[i];
            const float v = A.r_new_energy_wo[r];
            if (!A.r_lin[r] && v >= 0.f) b = __float_as_uint(v);
        }
        cand[q] = b;
        nvalid += b != 0xFFFFFFFFu;
    }
    for (int i = tid + LF_CACHE * nt; i < n_newframe; i += nt) {
        const int r = newframe_res[i];
        nvalid += (!A.r_lin[r] && A.r_new_energy_wo[r] >= 0.f);
    }
    for (int o = 32; o > 0; o >>= 1) nvalid += __shfl_down(nvalid, o);
    if ((tid & 63) == 0 && nvalid) atomicAdd(&s_misc[2], nvalid);
    __syncthreads();
    const unsigned n = s_misc[2];
    float th;
    if (n == 0) {
        th = 12 * 12 * 8;                                  // BA.cpp:2432-2436
    } else {
        if (tid == 0) { s_misc[0] = 0; s_misc[1] = (unsigned)(int)(0.7f * (float)n); }   // nthIdx, :2448
        __syncthreads();
        for (int pass = 3; pass >= 0; pass--) {
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = s_misc[0];
            const unsigned hmask = pass == 3 ? 0u : (0xFFFFFFFFu << (8 * (pass + 1)));
#pragma unroll
            for (int q = 0; q < LF_CACHE; q++) {
                const unsigned b = cand[q];
                if (b != 0xFFFFFFFFu && (b & hmask) == prefix) atomicAdd(&s_hist[(b >> (8 * pass)) & 0xFFu], 1u);
            }
            for (int i = tid + LF_CACHE * nt; i < n_newframe; i += nt) {
                const int r = newframe_res[i];
                const float v = A.r_new_energy_wo[r];
                if (A.r_lin[r] || !(v >= 0.f)) continue;
                const unsigned b = __float_as_uint(v);
                if ((b & hmask) == prefix) atomicAdd(&s_hist[(b >> (8 * pass)) & 0xFFu], 1u);
            }
            __syncthreads();
            if (tid < 64) {                                 // one wave scans the 256 buckets: 4 per lane + shuffle prefix
                const unsigned h0 = s_hist[4 * tid], h1 = s_hist[4 * tid + 1], h2 = s_hist[4 * tid + 2], h3 = s_hist[4 * tid + 3];
                const unsigned s = h0 + h1 + h2 + h3;
                unsigned incl = s;
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned v = __shfl_up(incl, o);
                    if (tid >= o) incl += v;
                }
                const unsigned excl = incl - s, kk = s_misc[1];
                if (kk >= excl && kk < incl) {              // exactly one lane
                    unsigned acc = excl;
                    int d = 4 * tid;
                    if (kk >= acc + h0) { acc += h0; d++; if (kk >= acc + h1) { acc += h1; d++; if (kk >= acc + h2) { acc += h2; d++; } } }
                    s_misc[1] = kk - acc;
                    s_misc[0] = prefix | ((unsigned)d << (8 * pass));
                }
            }
            __syncthreads();
        }
        const float nthElement = sqrtf(__uint_as_float(s_misc[0]));      // :2455
        double t = (double)(nthElement * 1.5f);                          // :2458
        t = (double)(26.0f * 0.5f) + t * (double)(1 - 0.5f);
        t = t * t;
        t *= (double)(1.0f * 1.0f);
        th = (float)t;
    }
    if (tid == 0) {
        if (A.ctl) { const int it_ = A.ctl->iters_done - 1; if (it_ >= 0 && it_ < 40) A.ctl->energy[it_] = tot[0]; else if (it_ < 0) A.ctl->energy0 = tot[0]; }   // statEnergyP of the resident loop
        out->energy = tot[0]; out->n_in = (int)tot[1]; out->n_oob = (int)tot[2]; out->n_outlier = (int)tot[3];
        out->new_frame_energy_th = th;
        frames_rw[A.N - 1].frame_energy_th = th;                         // takes effect from the next residual pass
    }
}
