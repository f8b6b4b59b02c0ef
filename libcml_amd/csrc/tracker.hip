// tracker.hip — coarse photometric tracker on the device.
// Replaces DSOTracker::computeResidual + computeHessian (TR.cpp:248-492, Accumulator9 ACC.h:1006-1211) and the
// image-sized part of makeCoarseDepthL0 (TR.cpp:550-719).
//
// tracker_eval is ONE fused launch per LM iteration: warp + bilinear gather + Huber/cutoff + the 45 unique entries
// of the weighted 9x9 JJ^T, reduced with wave shuffles, one LDS hop, one fp32 partial row per workgroup, and a
// last-block finish (agent-scope release/acquire, MI355X per-XCD L2s) — so the host gets E, counts, flow and the
// scaled 8x8 system from a single readback.  The reference's compaction into the warped buffer is unnecessary
// here (the Hessian is accumulated where the residual is computed); the per-point warped record is still
// written, uncompacted, for parity tests and for callers that want it.
#include "cmlhip_internal.h"
#include <atomic>
#include <chrono>
#include <cstdlib>

#pragma clang fp contract(off)

struct TrkArgs {
    const void* img; int w, h, level, n, want_h, half;
    const float* uvic;
    float RKi[9], Ki[9], t[3], fxl, fyl, cxl, cyl, a0, a1;
    float fxh, fyh, b0, a_h;                 // computeHessian constants (TR.cpp:426-429)
    float maxEnergy; double huber_d, cutoff_d, cutoff_base_d;
    float* warped;                           // n x 8 floats {idepth,u,v,dx,dy,residual,weight,refcolor}
    unsigned char* flag;                     // n: 1 = written to the warped buffer
    float* partial;                          // gridDim x TRK_NRED (device buffer, or the mapped host buffer)
    unsigned* done; unsigned seq;            // host-visible per-workgroup flags (null: the caller copies `partial` back)
    int trips;                               // 256-point trips per workgroup (1 for the usual list sizes: more workgroups, one round trip each)
};
#define TRK_NRED 56   // 45 (H upper) + E + sT + sRT + sN + numTerms + numSat + numRobust + numWarped + pad(3)

template <bool HALF>
__device__ __forceinline__ float4 trk_texel(const void* img, size_t i) {
    if (HALF) {
        uint2 v = reinterpret_cast<const uint2*>(img)[i];
        __half2 a = *reinterpret_cast<__half2*>(&v.x), b = *reinterpret_cast<__half2*>(&v.y);
        return make_float4(__low2float(a), __high2float(a), __low2float(b), 0.f);
    }
    return reinterpret_cast<const float4*>(img)[i];
}

// Reduction layout.  Everything the host needs is a sum over the points of an outer product a_k b_k^T with
//   a = [hw*J (9) | 1 | numRobust numWarped 0 0 0 0]      b = [J (9) | 1 | E sT sRT sN numTerms numSat]
// so D = sum_k a_k b_k^T holds the 9x9 system in D[0..8][0..8] (entry (r,c) = sum (J_r hw) J_c, TR.cpp:443-470 /
// Accumulator9), the b-side scalars in row 9 (a_9 = 1) and the a-side scalars in column 9 (b_9 = 1).  D accumulates on the
// matrix cores (v_mfma_f32_16x16x4_f32, IEEE fp32, 4 points per instruction): no per-lane accumulators, no shuffles.
// Each lane computes one point, the a/b vectors cross to the MFMA operand layout through a per-wave LDS tile, a
// workgroup covers 256 x trips points, writes the 56 sums it owns, and the (synchronous) caller adds the few
// workgroup rows in block order on the host — no atomics, no inter-workgroup fences.
typedef float trk_float4 __attribute__((ext_vector_type(4)));
#define TRK_LD 17
#define TRK_HOST_BLOCKS 256

template <bool HALF>
__global__ __launch_bounds__(256) void k_tracker_eval(TrkArgs A) {
    __shared__ float s_a[4][64][TRK_LD], s_b[4][64][TRK_LD];
    __shared__ float s_tile[4][256];
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    trk_float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int trip = 0; trip < A.trips; trip++) {
        const int i = (blockIdx.x * A.trips + trip) * 256 + tid;
        if ((blockIdx.x * A.trips + trip) * 256 >= A.n) break;               // workgroup-uniform
        float va[16], vb[16];
#pragma unroll
        for (int k = 0; k < 16; k++) { va[k] = 0.f; vb[k] = 0.f; }
        if (i < A.n) {
            va[9] = 1.f; vb[9] = 1.f;
            const float4 q = reinterpret_cast<const float4*>(A.uvic)[i];
            const float x = q.x, y = q.y, id = q.z, refColor = q.w;
            bool wrote = false;
            if (isfinite(refColor)) {                                           // TR.cpp:301-303
                float pt[3];
#pragma unroll
                for (int k = 0; k < 3; k++) pt[k] = (A.RKi[k * 3] * x + (A.RKi[k * 3 + 1] * y + A.RKi[k * 3 + 2] * 1.0f)) + A.t[k] * id;   // Eigen's order for a float 3x3 * 3-vector: e0 + (e1 + e2)
                const float u = pt[0] / pt[2], vv = pt[1] / pt[2];
                const float Ku = A.fxl * u + A.cxl, Kv = A.fyl * vv + A.cyl;
                const float new_idepth = id / pt[2];
                if (A.level == 0 && (i % 32) == 0) {                           // flow statistic, TR.cpp:313-344
                    float a[3], b[3], c[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const float kp = A.Ki[k * 3] * x + (A.Ki[k * 3 + 1] * y + A.Ki[k * 3 + 2] * 1.0f);
                        a[k] = kp + A.t[k] * id; b[k] = kp - A.t[k] * id;
                        c[k] = (A.RKi[k * 3] * x + (A.RKi[k * 3 + 1] * y + A.RKi[k * 3 + 2] * 1.0f)) - A.t[k] * id;
                    }
                    const float KuT = A.fxl * (a[0] / a[2]) + A.cxl, KvT = A.fyl * (a[1] / a[2]) + A.cyl;
                    const float KuT2 = A.fxl * (b[0] / b[2]) + A.cxl, KvT2 = A.fyl * (b[1] / b[2]) + A.cyl;
                    const float Ku3 = A.fxl * (c[0] / c[2]) + A.cxl, Kv3 = A.fyl * (c[1] / c[2]) + A.cyl;
                    float sT = 0, sRT = 0;
                    sT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
                    sT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
                    sRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
                    sRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
                    vb[11] = sT; vb[12] = sRT; vb[13] = 2.f;
                }
                if (Ku > 2 && Kv > 2 && Ku < A.w - 3 && Kv < A.h - 3 && new_idepth > 0) {     // TR.cpp:346
                    const int ix = (int)Ku, iy = (int)Kv;
                    const float dx = Ku - (float)ix, dy = Kv - (float)iy, dxdy = dx * dy;
                    const float w00 = 1 - dx - dy + dxdy, w01 = dx - dxdy, w10 = dy - dxdy, w11 = dxdy;
                    const size_t i1 = (size_t)iy * A.w + ix;
                    const float4 ta = trk_texel<HALF>(A.img, i1), tb = trk_texel<HALF>(A.img, i1 + 1);
                    const float4 tc = trk_texel<HALF>(A.img, i1 + A.w), td = trk_texel<HALF>(A.img, i1 + A.w + 1);
                    const float h0 = ta.x * w00 + tb.x * w01 + tc.x * w10 + td.x * w11;
                    const float h1 = ta.y * w00 + tb.y * w01 + tc.y * w10 + td.y * w11;
                    const float h2 = ta.z * w00 + tb.z * w01 + tc.z * w10 + td.z * w11;
                    if (isfinite(h0) && isfinite(h1) && isfinite(h2)) {
                        const float residual = h0 - (float)(A.a0 * refColor + A.a1);
                        const float hw = fabs((double)residual) < A.huber_d ? 1.0f : (float)(A.huber_d / fabs((double)residual));
                        if (fabs((double)residual) > A.cutoff_d) {
                            vb[10] = A.maxEnergy; vb[14] = 1.f; vb[15] = 1.f;                  // E, numTerms, numSaturated
                        } else {
                            vb[10] = hw * residual * residual * (2 - hw); vb[14] = 1.f; va[11] = 1.f;   // E, numTerms, numWarped
                            wrote = true;
                            float* W = A.warped + 8 * (size_t)i;
                            W[0] = new_idepth; W[1] = u; W[2] = vv; W[3] = h1; W[4] = h2; W[5] = residual; W[6] = hw; W[7] = refColor;
                            if (A.want_h) {                                       // computeHessian lanes, TR.cpp:443-470
                                const float ddx = h1 * A.fxh, ddy = h2 * A.fyh;
                                vb[0] = new_idepth * ddx;
                                vb[1] = new_idepth * ddy;
                                vb[2] = 0.0f - (new_idepth * (u * ddx + vv * ddy));
                                vb[3] = 0.0f - ((u * vv * ddx) + ddy * (1.0f + vv * vv));
                                vb[4] = (u * vv * ddy) + (ddx * (1.0f + u * u));
                                vb[5] = u * ddy - vv * ddx;
                                vb[6] = A.a_h * (A.b0 - refColor);
                                vb[7] = -1.0f;
                                vb[8] = residual;
#pragma unroll
                                for (int r = 0; r < 9; r++) va[r] = vb[r] * hw;
                            }
                        }
                        if (fabs((double)residual) <= A.cutoff_base_d) va[10] = 1.f;           // numRobust
                    }
                }
            }
            A.flag[i] = wrote ? 1 : 0;
        }
        // ---- this wave's 64 points to the MFMA operand layout (A[i][k] = a_k[i], B[k][j] = b_k[j], 4 points per step)
#pragma unroll
        for (int k = 0; k < 16; k++) { s_a[wv][l][k] = va[k]; s_b[wv][l][k] = vb[k]; }
        // (LDS accesses of one wave are ordered: no barrier between the stores above and the loads below)
        const int e = l & 15, kq = l >> 4;
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const float av = s_a[wv][4 * m + kq][e], bv = s_b[wv][4 * m + kq][e];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
    }
    // D: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int rg = 0; rg < 4; rg++) s_tile[wv][(4 * (l >> 4) + rg) * 16 + (l & 15)] = acc[rg];
    __syncthreads();
    if (tid < TRK_NRED) {
        int src = -1;
        if (tid < 45) {
            int k = tid, r = 0;
            while (k >= 9 - r) { k -= 9 - r; r++; }
            src = r * 16 + (r + k);
        } else if (tid <= 50) src = 9 * 16 + 10 + (tid - 45);          // E sT sRT sN numTerms numSaturated
        else if (tid == 51) src = 10 * 16 + 9;                          // numRobust
        else if (tid == 52) src = 11 * 16 + 9;                          // numWarped
        float v = 0.f;
        if (src >= 0) v = ((s_tile[0][src] + s_tile[1][src]) + s_tile[2][src]) + s_tile[3][src];
        A.partial[(size_t)blockIdx.x * TRK_NRED + tid] = v;
    }
    if (A.done) {                                          // publish to the polling host: rows first, then the flag (system scope)
        __threadfence_system();
        __syncthreads();
        if (tid == 0) { __hip_atomic_store(A.done + blockIdx.x, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}

// ------------------------------------------------------------------------------------------------ makeCoarseDepthL0
// `idepth(u + w0 v) += new_idepth * weight; weightSum += weight` is a sequential float accumulation over the points in the
// reference, and two points do land on one pixel now and then: float atomics in arrival order would differ from it in the last
// bit (and from run to run).  Three small launches keep the list order exactly: the lowest point index of every pixel
// (atomicMin on an int map), the plain store of those first points, and one thread that applies the few later points of
// shared pixels in index order.
struct CdPoint { int u, v; double x; float weight; bool valid; };
__device__ __forceinline__ CdPoint cd_point(const double* __restrict__ pts, int i, int w0, int h0) {     // TR.cpp:538-548
    CdPoint p;
    const double Ku = pts[4 * (size_t)i], Kv = pts[4 * (size_t)i + 1], nid = pts[4 * (size_t)i + 2];
    p.weight = (float)pts[4 * (size_t)i + 3];
    p.u = (int)(Ku + 0.5); p.v = (int)(Kv + 0.5);
    p.valid = !(p.u < 0 || p.u >= w0 || p.v < 0 || p.v >= h0);
    p.x = nid * (double)p.weight;
    return p;
}
__global__ void k_cd_owner(const double* __restrict__ pts, int n, int w0, int h0, int* owner) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CdPoint p = cd_point(pts, i, w0, h0);
    if (p.valid) atomicMin(&owner[p.u + w0 * p.v], i);
}
__global__ void k_cd_splat(const double* __restrict__ pts, int n, int w0, int h0, const int* __restrict__ owner, float* idepth, float* wsum,
                           int* late, int* n_late) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CdPoint p = cd_point(pts, i, w0, h0);
    if (!p.valid) return;
    const int pix = p.u + w0 * p.v;
    if (owner[pix] == i) { idepth[pix] = (float)((double)0.0f + p.x); wsum[pix] = 0.0f + p.weight; }     // the maps start at zero
    else late[atomicAdd(n_late, 1)] = i;
}
// The later points of shared pixels, applied in point-index order per pixel (the reference's sequential `+=`, TR.cpp:550-553).  Up to
// CD_LATE_MAX of them: ranked by index in LDS (a comparison per pair, no sort loop), then every pixel's chain walked by the thread of
// its FIRST late point — chains of different pixels are independent.  (Round 4: the one-thread insertion sort + walk over global memory
// this replaces took 84 us on average, up to 160 us, per keyframe of the sequence test.)  Beyond CD_LATE_MAX: the one-thread form.
#define CD_LATE_MAX 2048
__global__ __launch_bounds__(1024) void k_cd_late(const double* __restrict__ pts, int w0, int h0, float* idepth, float* wsum, int* late, const int* n_late) {
    if (blockIdx.x != 0) return;
    const int m = *n_late, tid = threadIdx.x;
    if (m > CD_LATE_MAX) {
        if (tid != 0) return;
        for (int a = 1; a < m; a++) {
            const int key = late[a];
            int b = a - 1;
            while (b >= 0 && late[b] > key) { late[b + 1] = late[b]; b--; }
            late[b + 1] = key;
        }
        for (int a = 0; a < m; a++) {
            const CdPoint p = cd_point(pts, late[a], w0, h0);
            const int pix = p.u + w0 * p.v;
            idepth[pix] = (float)((double)idepth[pix] + p.x);
            wsum[pix] += p.weight;
        }
        return;
    }
    __shared__ int s_raw[CD_LATE_MAX], s_idx[CD_LATE_MAX], s_pix[CD_LATE_MAX];
    for (int a = tid; a < m; a += blockDim.x) s_raw[a] = late[a];
    __syncthreads();
    for (int a = tid; a < m; a += blockDim.x) {                       // rank = number of late points with a smaller index (indices are distinct)
        const int key = s_raw[a];
        int rank = 0;
        for (int b = 0; b < m; b++) rank += s_raw[b] < key;
        const CdPoint p = cd_point(pts, key, w0, h0);
        s_idx[rank] = key; s_pix[rank] = p.u + w0 * p.v;
    }
    __syncthreads();
    for (int a = tid; a < m; a += blockDim.x) {
        const int pix = s_pix[a];
        bool head = true;
        for (int b = 0; b < a && head; b++) head = s_pix[b] != pix;
        if (!head) continue;
        float id = idepth[pix], ws = wsum[pix];
        for (int c = a; c < m; c++) {
            if (s_pix[c] != pix) continue;
            const CdPoint p = cd_point(pts, s_idx[c], w0, h0);
            id = (float)((double)id + p.x);
            ws += p.weight;
        }
        idepth[pix] = id; wsum[pix] = ws;
    }
}
__global__ void k_cd_down(const float* __restrict__ idm, const float* __restrict__ wm, int wm1, int wl, int hl,
                          float* __restrict__ idl, float* __restrict__ wsl, float* __restrict__ wbak) {             // TR.cpp:571-584
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= wl || y >= hl) return;
    const int b = 2 * x + 2 * y * wm1;
    idl[x + y * wl] = ((idm[b] + idm[b + 1]) + idm[b + wm1]) + idm[b + wm1 + 1];
    const float ws = ((wm[b] + wm[b + 1]) + wm[b + wm1]) + wm[b + wm1 + 1];
    wsl[x + y * wl] = ws; wbak[x + y * wl] = ws;                       // backupWeightSum of this level rides along
}
// per-level pointers of the passes that run over all levels in one launch (blockIdx.y = level)
struct CdLevels {
    float* idepth[8]; float* wsum[8]; float* wbak[8]; const float* gray[8]; float* uvic[8];
    int w[8], h[8], cnt_off[8], nb[8];
    int* counts; int* totals;
};
// dilation, TR.cpp:589-665: reads only cells with weightSumBak > 0, writes only cells with weightSumBak <= 0
__global__ void k_cd_dilate(CdLevels L) {
    const int lv = blockIdx.y, wl = L.w[lv], hl = L.h[lv], diag = lv < 2 ? 1 : 0;
    float* idepth = L.idepth[lv]; float* wsum = L.wsum[lv]; const float* __restrict__ wbak = L.wbak[lv];
    const int i = blockIdx.x * blockDim.x + threadIdx.x + wl;
    const int wh = wl * hl - wl, size = wl * hl;
    if (i >= wh) return;
    if (wbak[i] > 0) return;
    int d[4];
    if (diag) { d[0] = 1 + wl; d[1] = -1 - wl; d[2] = wl - 1; d[3] = -wl + 1; }
    else { d[0] = 1; d[1] = -1; d[2] = wl; d[3] = -wl; }
    float sum = 0, num = 0, numn = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = i + d[k];
        if (j >= 0 && j < size && wbak[j] > 0) { sum += idepth[j]; num += wbak[j]; numn++; }
    }
    if (numn > 0) { idepth[i] = sum / numn; wsum[i] = num / numn; }
}
// normalise + ordered compaction (raster order of TR.cpp:688-716), 3 passes: count / scan / scatter
__device__ __forceinline__ bool cd_valid(const float* idepth, const float* wsum, const float* gray, int wl, int hl, int i, float& id, float& col) {
    const int x = i % wl, y = i / wl;
    if (x < 2 || y < 2 || x >= wl - 2 || y >= hl - 2) return false;
    if (!(wsum[i] > 0)) return false;
    id = idepth[i] / wsum[i];
    col = gray[i];
    return isfinite(col) && (id > 0);
}
__global__ __launch_bounds__(1024) void k_cd_count(CdLevels L) {
    __shared__ int s[16];
    const int lv = blockIdx.y;
    if ((int)blockIdx.x >= L.nb[lv]) return;
    const float* idepth = L.idepth[lv]; const float* wsum = L.wsum[lv]; const float* gray = L.gray[lv];
    const int wl = L.w[lv], hl = L.h[lv];
    int* counts = L.counts + L.cnt_off[lv];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    float id, col;
    const bool ok = (i < wl * hl) && cd_valid(idepth, wsum, gray, wl, hl, i, id, col);
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < 16; k++) t += s[k]; counts[blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void k_cd_scan(CdLevels L) {      // exclusive scan of the per-block counts, one workgroup per level
    int* counts = L.counts + L.cnt_off[blockIdx.x]; const int nb = L.nb[blockIdx.x]; int* total = L.totals + blockIdx.x;
    __shared__ int s[1024];
    __shared__ int carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + t;
        const int v = i < nb ? counts[i] : 0;
        s[t] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {                         // Hillis-Steele inclusive scan
            const int a = t >= o ? s[t - o] : 0;
            __syncthreads();
            s[t] += a;
            __syncthreads();
        }
        if (i < nb) counts[i] = carry + s[t] - v;
        __syncthreads();
        if (t == 1023) carry += s[1023];
        __syncthreads();
    }
    if (t == 0) *total = carry;
}
__global__ __launch_bounds__(1024) void k_cd_scatter(CdLevels L) {
    __shared__ int s[16];
    const int lv = blockIdx.y;
    if ((int)blockIdx.x >= L.nb[lv]) return;
    const float* idepth = L.idepth[lv]; const float* wsum = L.wsum[lv]; const float* gray = L.gray[lv];
    const int wl = L.w[lv], hl = L.h[lv];
    const int* offs = L.counts + L.cnt_off[lv]; float* uvic = L.uvic[lv];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    float id = 0, col = 0;
    const bool ok = (i < wl * hl) && cd_valid(idepth, wsum, gray, wl, hl, i, id, col);
    const unsigned long long m = __ballot(ok);
    const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (ln == 0) s[wv] = __popcll(m);
    __syncthreads();
    int base = offs[blockIdx.x];
    for (int k = 0; k < wv; k++) base += s[k];
    if (ok) {
        const int pos = base + __popcll(m & ((1ull << ln) - 1ull));
        uvic[4 * (size_t)pos] = (float)(i % wl); uvic[4 * (size_t)pos + 1] = (float)(i / wl);
        uvic[4 * (size_t)pos + 2] = id; uvic[4 * (size_t)pos + 3] = col;
    }
}

// Eigen compute_inverse_size3 (cofactors * 1/det) in float, as Matrix33f::inverse() at TR.cpp:261
static void inv3f(const float m[9], float o[9]) {
#define MM(i, j) m[(i) * 3 + (j)]
#define COF(i, j) (MM(((i) + 1) % 3, ((j) + 1) % 3) * MM(((i) + 2) % 3, ((j) + 2) % 3) - MM(((i) + 1) % 3, ((j) + 2) % 3) * MM(((i) + 2) % 3, ((j) + 1) % 3))
    const float c0 = COF(0, 0), c1 = COF(1, 0), c2 = COF(2, 0);
    const float det = c0 * MM(0, 0) + (c1 * MM(1, 0) + c2 * MM(2, 0));     // Eigen's redux of three terms: e0 + (e1 + e2)
    const float invdet = 1.0f / det;
    o[0] = c0 * invdet; o[1] = c1 * invdet; o[2] = c2 * invdet;
    o[3] = COF(0, 1) * invdet; o[4] = COF(1, 1) * invdet; o[5] = COF(2, 1) * invdet;
    o[6] = COF(0, 2) * invdet; o[7] = COF(1, 2) * invdet; o[8] = COF(2, 2) * invdet;
#undef COF
#undef MM
}

extern "C" {

int cmlhip_tracker_set_reference(cmlhip_ctx* c, int level, const float* uvic, int n) { CML_DEV(c);
    if (!c || level < 0 || level >= 8 || n < 0 || (n > 0 && !uvic)) return CMLHIP_ERR_INVALID;
    CML_REQUIRE(c, n <= c->lim.max_tracker_points, CMLHIP_ERR_INVALID, "tracker list exceeds max_tracker_points");
    int rc = cml_ensure(c, c->trk_ref[level], 16 * (size_t)(n ? n : 1));
    if (rc) return rc;
    if ((rc = cml_h2d(c, c->trk_ref[level].p, uvic, 16 * (size_t)n))) return rc;
    c->trk_n[level] = n;
    return CMLHIP_OK;
}

int cmlhip_tracker_get_reference(cmlhip_ctx* c, int level, float* out, int* n_out) { CML_DEV(c);
    if (!c || level < 0 || level >= 8) return CMLHIP_ERR_INVALID;
    if (n_out) *n_out = c->trk_n[level];
    if (out && c->trk_n[level] > 0) return cml_d2h(c, out, c->trk_ref[level].p, 16 * (size_t)c->trk_n[level]);
    return CMLHIP_OK;
}

int cmlhip_tracker_eval(cmlhip_ctx* c, uint64_t image_id, int level, const double R[9], const double t[3], const double K[4],
                        const double aff[2], double b0, const cmlhip_tracker_params* prm, int want_hessian,
                        cmlhip_tracker_result* out) { CML_DEV(c);
    if (!c || !R || !t || !K || !aff || !prm || !out || level < 0 || level >= 8) return CMLHIP_ERR_INVALID;
    const Pyramid* py = cml_find_pyr(c, image_id);
    CML_REQUIRE(c, py && level < py->levels && py->lv[level].grad, CMLHIP_ERR_NOT_FOUND, "tracker image/level not in the pyramid cache");
    const int n = c->trk_n[level];
    TrkArgs A;
    memset(&A, 0, sizeof A);
    A.img = py->lv[level].grad; A.w = py->lv[level].w; A.h = py->lv[level].h; A.level = level; A.n = n; A.want_h = want_hessian;
    A.uvic = c->trk_ref[level].as<float>();
    // host-side constants exactly as TR.cpp:260-278,426-429 forms them (float)
    float Kf[9] = {(float)K[0], 0, (float)K[2], 0, (float)K[1], (float)K[3], 0, 0, 1}, Rf[9];
    inv3f(Kf, A.Ki);
    for (int i = 0; i < 9; i++) Rf[i] = (float)R[i];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) A.RKi[i * 3 + j] = Rf[i * 3] * A.Ki[j] + (Rf[i * 3 + 1] * A.Ki[3 + j] + Rf[i * 3 + 2] * A.Ki[6 + j]);   // Matrix33f product, Eigen order
    for (int i = 0; i < 3; i++) A.t[i] = (float)t[i];
    A.fxl = Kf[0]; A.fyl = Kf[4]; A.cxl = Kf[2]; A.cyl = Kf[5];
    A.a0 = (float)aff[0]; A.a1 = (float)aff[1];
    A.fxh = (float)K[0]; A.fyh = (float)K[1]; A.b0 = (float)b0; A.a_h = (float)aff[0];
    A.huber_d = (double)prm->huber; A.cutoff_d = (double)prm->cutoff; A.cutoff_base_d = (double)prm->cutoff_base;
    A.maxEnergy = (float)(2.0f * A.huber_d * A.cutoff_d - A.huber_d * A.huber_d);
    A.trips = n <= 65536 ? 1 : 4;
    const int blocks = cml_div_up(n > 0 ? n : 1, 256 * A.trips);
    int rc;
    if ((rc = cml_ensure(c, c->trk_warped, (size_t)(n ? n : 1) * 36))) return rc;
    if ((rc = cml_ensure(c, c->trk_partial, (size_t)blocks * TRK_NRED * 4))) return rc;
    if ((rc = cml_ensure(c, c->trk_out, 1024))) return rc;
    A.warped = c->trk_warped.as<float>();
    A.flag = reinterpret_cast<unsigned char*>(c->trk_warped.as<float>() + 8 * (size_t)(n ? n : 1));
    A.partial = c->trk_partial.as<float>();
    c->trk_last_n = n;
    // result path: up to TRK_HOST_BLOCKS workgroups write straight into a mapped host buffer that the caller polls
    const bool direct = blocks <= TRK_HOST_BLOCKS && !getenv("CMLHIP_TRACKER_NO_HOST_POLL");
    if (direct && !c->trk_host) {
        CML_CHECK(c, hipHostMalloc(reinterpret_cast<void**>(&c->trk_host), TRK_HOST_BLOCKS * (TRK_NRED + 1) * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
        memset(c->trk_host, 0, TRK_HOST_BLOCKS * (TRK_NRED + 1) * sizeof(float));
    }
    if (direct) {
        void* dptr = nullptr;
        CML_CHECK(c, hipHostGetDevicePointer(&dptr, c->trk_host, 0));
        A.partial = static_cast<float*>(dptr);
        A.done = reinterpret_cast<unsigned*>(static_cast<float*>(dptr) + TRK_HOST_BLOCKS * TRK_NRED);
        A.seq = ++c->trk_seq;
        if (A.seq == 0) A.seq = ++c->trk_seq;
    }
    if (c->lim.texel_format == CMLHIP_TEXEL_F16) CML_LAUNCH_EV(c, k_tracker_eval<true>, blocks, 256, 0, A);
    else CML_LAUNCH_EV(c, k_tracker_eval<false>, blocks, 256, 0, A);
    CML_CHECK(c, hipGetLastError());
    // the workgroup rows are added here, in block order, fp64 (what the last-block pass of a fused finish would do)
    std::vector<float> part((size_t)blocks * TRK_NRED);
    if (direct) {
        volatile unsigned* flags = reinterpret_cast<volatile unsigned*>(c->trk_host + TRK_HOST_BLOCKS * TRK_NRED);
        const auto t0 = std::chrono::steady_clock::now();
        bool ok = true;
        for (int b = 0; b < blocks && ok; b++) {
            unsigned spins = 0;
            while (flags[b] != A.seq) {
                if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) { ok = false; break; }
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!ok) CML_CHECK(c, hipStreamSynchronize(c->stream));      // never spin forever: fall back to the stream
        memcpy(part.data(), c->trk_host, part.size() * sizeof(float));
    } else if ((rc = cml_d2h(c, part.data(), c->trk_partial.p, part.size() * sizeof(float)))) return rc;
    float s[TRK_NRED];
    for (int k = 0; k < TRK_NRED; k++) {
        double acc = 0;
        for (int b = 0; b < blocks; b++) acc += (double)part[(size_t)b * TRK_NRED + k];
        s[k] = (float)acc;
    }
    memset(out, 0, sizeof *out);
    out->E = s[45]; out->numTermsInE = (int)s[49]; out->numSaturated = (int)s[50]; out->numRobust = (int)s[51]; out->numWarped = (int)s[52];
    out->flow[0] = s[46] / (s[48] + 0.1f); out->flow[1] = 0; out->flow[2] = s[47] / (s[48] + 0.1f);       // TR.cpp:412-414
    if (want_hessian) {
        int idx = 0;
        for (int r = 0; r < 9; r++) for (int cc = r; cc < 9; cc++) { out->H9[r * 9 + cc] = out->H9[cc * 9 + r] = s[idx]; idx++; }
        int npad = out->numWarped;
        while (npad % 4 != 0) npad++;                                                                       // TR.cpp:391-403,436
        const double sc[8] = {prm->scale_rot, prm->scale_rot, prm->scale_rot, prm->scale_trans, prm->scale_trans, prm->scale_trans,
                              prm->scale_a, prm->scale_b};                                                  // TR.cpp:477-488 (literal lane/scale pairing)
        for (int r = 0; r < 8; r++) {
            for (int cc = 0; cc < 8; cc++) out->H[r * 8 + cc] = ((double)out->H9[r * 9 + cc] / (double)npad) * sc[cc] * sc[r];
            out->b[r] = ((double)out->H9[r * 9 + 8] / (double)npad) * sc[r];
        }
        for (int k = 0; k < 64; k++) if (!std::isfinite(out->H[k])) return CMLHIP_ERR_NONFINITE;
    }
    return CMLHIP_OK;
}

int cmlhip_tracker_get_warped(cmlhip_ctx* c, float* out, int capacity, int* n_out) { CML_DEV(c);
    if (!c || !out || capacity < 0) return CMLHIP_ERR_INVALID;
    const int n = c->trk_last_n;
    if (n_out) *n_out = 0;
    if (n == 0) return CMLHIP_OK;
    std::vector<float> w(8 * (size_t)n);
    std::vector<unsigned char> f(n);
    int rc;
    if ((rc = cml_d2h(c, w.data(), c->trk_warped.p, 32 * (size_t)n))) return rc;
    if ((rc = cml_d2h(c, f.data(), c->trk_warped.as<float>() + 8 * (size_t)n, (size_t)n))) return rc;
    int m = 0;                                  // compaction in reference-list order (TR.cpp:372-380)
    for (int i = 0; i < n; i++) {
        if (!f[i]) continue;
        if (m < capacity) for (int k = 0; k < 8; k++) out[(size_t)k * capacity + m] = w[8 * (size_t)i + k];
        m++;
    }
    if (n_out) *n_out = m;
    return CMLHIP_OK;
}

int cmlhip_tracker_make_coarse_depth(cmlhip_ctx* c, uint64_t ref_image_id, int levels, const double* pts, int n, int* n_out) { CML_DEV(c);
    if (!c || levels < 1 || levels > 8 || n < 0 || (n > 0 && !pts) || !n_out) return CMLHIP_ERR_INVALID;
    const Pyramid* py = cml_find_pyr(c, ref_image_id);
    CML_REQUIRE(c, py && py->levels >= levels && py->lv[0].gray, CMLHIP_ERR_NOT_FOUND, "reference pyramid (with gray levels) not cached");
    int rc;
    CdLevels L{};
    int maxblocks = 1, cnt_total_off = 0, maxdil = 1;
    for (int l = 0; l < levels; l++) {
        const size_t sz = (size_t)py->lv[l].w * py->lv[l].h;
        if ((rc = cml_ensure(c, c->cd_idepth[l], 4 * sz))) return rc;
        if ((rc = cml_ensure(c, c->cd_wsum[l], 4 * sz))) return rc;
        if ((rc = cml_ensure(c, c->cd_wbak[l], 4 * sz))) return rc;
        if ((rc = cml_ensure(c, c->trk_ref[l], 16 * sz))) return rc;
        const int nb = cml_div_up((int)sz, 1024);
        L.idepth[l] = c->cd_idepth[l].as<float>(); L.wsum[l] = c->cd_wsum[l].as<float>(); L.wbak[l] = c->cd_wbak[l].as<float>();
        L.gray[l] = py->lv[l].gray; L.uvic[l] = c->trk_ref[l].as<float>();
        L.w[l] = py->lv[l].w; L.h[l] = py->lv[l].h; L.cnt_off[l] = cnt_total_off; L.nb[l] = nb;
        if (nb > maxblocks) maxblocks = nb;
        cnt_total_off += nb;
        maxdil = std::max(maxdil, cml_div_up(std::max((int)sz - 2 * py->lv[l].w, 1), 256));
    }
    if ((rc = cml_ensure(c, c->cd_cnt, 4 * (size_t)(cnt_total_off + 16)))) return rc;
    L.counts = c->cd_cnt.as<int>(); L.totals = L.counts + cnt_total_off;
    const size_t sz0 = (size_t)py->lv[0].w * py->lv[0].h;
    DevBuf& dpts = c->cd_pts;                                        // grow-only, kept across calls
    if (n > 0 && (rc = cml_ensure(c, dpts, 32 * (size_t)n + 4 * ((size_t)n + 4)))) return rc;
    // the two cleared level-0 maps, the points, the owner map's "nobody yet" and the cleared counter leave as ONE packed upload (a scatter kernel over the
    // pinned staging block) — they were four fills and a copy, a launch and a runtime call each
    cml_h2d_batch_begin(c);
    rc = cml_zero(c, L.idepth[0], 4 * sz0);                                  // only level 0 is splatted into; the others are written whole
    if (!rc) rc = cml_zero(c, L.wsum[0], 4 * sz0);
    if (!rc && n > 0) {
        rc = cml_h2d(c, dpts.p, pts, 32 * (size_t)n);
        if (!rc) rc = cml_fill_7f(c, c->cd_wbak[0].p, 4 * sz0);
        if (!rc) rc = cml_zero(c, dpts.as<char>() + 32 * (size_t)n + 4 * (size_t)n, 4);
    }
    { const int rf = cml_h2d_batch_flush(c); if (rc || rf) return rc ? rc : rf; }
    if (n > 0) {
        int* late = reinterpret_cast<int*>(dpts.as<char>() + 32 * (size_t)n);
        int* n_late = late + n;
        int* owner = c->cd_wbak[0].as<int>();                           // free until the weights are backed up into it
        k_cd_owner<<<cml_div_up(n, 256), 256, 0, c->stream>>>(dpts.as<double>(), n, py->lv[0].w, py->lv[0].h, owner);
        k_cd_splat<<<cml_div_up(n, 256), 256, 0, c->stream>>>(dpts.as<double>(), n, py->lv[0].w, py->lv[0].h, owner, L.idepth[0], L.wsum[0], late, n_late);
        k_cd_late<<<1, 1024, 0, c->stream>>>(dpts.as<double>(), py->lv[0].w, py->lv[0].h, L.idepth[0], L.wsum[0], late, n_late);
    }
    CML_CHECK(c, hipMemcpyAsync(L.wbak[0], L.wsum[0], 4 * sz0, hipMemcpyDeviceToDevice, c->stream));   // backupWeightSum, level 0
    for (int l = 1; l < levels; l++) {
        dim3 g(cml_div_up(py->lv[l].w, 256), py->lv[l].h);
        k_cd_down<<<g, 256, 0, c->stream>>>(L.idepth[l - 1], L.wsum[l - 1], py->lv[l - 1].w, py->lv[l].w, py->lv[l].h, L.idepth[l], L.wsum[l], L.wbak[l]);
    }
    // (the down-sampling reads the UN-dilated maps of the level above, TR.cpp:571-584 runs before :589-665, so every level is
    //  complete before any is dilated and the dilation, like the compaction, is one launch over all levels)
    k_cd_dilate<<<dim3(maxdil, levels), 256, 0, c->stream>>>(L);
    k_cd_count<<<dim3(maxblocks, levels), 1024, 0, c->stream>>>(L);
    k_cd_scan<<<levels, 1024, 0, c->stream>>>(L);
    k_cd_scatter<<<dim3(maxblocks, levels), 1024, 0, c->stream>>>(L);
    int tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((rc = cml_d2h(c, tot, L.totals, sizeof(int) * levels))) return rc;      // one readback for all levels
    for (int l = 0; l < levels; l++) { c->trk_n[l] = tot[l]; n_out[l] = tot[l]; }
    CML_CHECK(c, hipGetLastError());
    return CMLHIP_OK;
}

}  // extern "C"
