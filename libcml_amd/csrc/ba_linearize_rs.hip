// ba_linearize_rs.hip — residual / Jacobian kernel of the device-RESIDENT Gauss-Newton loop (cmlhip_ba_iteration_async).
// Same arithmetic as k_ba_linearize (ba_linearize.hip: DSOBundleAdjustmentLinearizationContext::linearize BA.cpp:62-316 with the
// fused applyRes BA.cpp:2051-2093, statement order kept, FP contraction off), re-mapped for throughput:
//
//   * the device residual order is (host,target)-pair-sorted (cmlhip_ba_upload_window), a WAVE owns up to 64 residuals of ONE pair:
//     the pair record (R, t, R0, t0, affine), both frame descriptors and the camera are wave-uniform and live in SGPRs (scalar
//     loads, scalar operands);
//   * ONE LANE PER RESIDUAL.  Measured on the way here (profiles/round2_*): with 8 or 4 lanes per residual only a quarter of the
//     issued lane-instructions is the per-pixel work, the rest is per-residual work replicated in every lane of the group, lane
//     selects, and the exchange of per-pixel operands through LDS; with one lane per residual the 8 pattern pixels are a plain
//     unrolled loop, the 19 pattern sums are register accumulators updated in pattern order (the reference's order by
//     construction), the geometric Jacobians are evaluated once, and there is no exchange, no ballot, no lane select at all;
//   * every per-residual input is addressed directly by the residual index (static copies of the point's pixel, colours and
//     weights are kept per residual, the copy of the inverse depth is refreshed by the point step of k_ba_backsub);
//   * the kernel's time at a 20-frame window follows the number of DISTINCT CACHE LINES it pulls from beyond the L2 (measured with the
//     development switches of cml_launch_linearize_rs: DESIGN.md section 7), so fp16 texels are gathered from a TILED copy of the level 0
//     (cml_tiled_level0: 4 x 4 texels + one overlap column per 128-byte line, 5.1 instead of 7.9 lines per residual), the two texels
//     of a bilinear row as one 16-byte load, unconditional on clamped positions;
//   * the pattern pixels run as a software pipeline (projection / loads / sums interleaved, three pixels in flight), the geometry is
//     evaluated while texels are in flight and parked in LDS: 135 VGPRs, three waves per SIMD, no scratch frame;
//   * nothing of the 74-float DSORawResidualJacobian is written to memory.  What the next iteration consumes leaves the kernel in
//     reduced form: the wave's contribution to the 13x13 AccumulatorApprox block of its pair (BA.cpp:1731-1745, ACC.h:776-932) as
//     ONE 16x16 fp32 tile accumulated on the matrix cores over the wave's residuals (v_mfma_f32_16x16x4_f32, one per residual,
//     the formulation of k_ba_acc, over the residuals that are IN), and per
//     residual 14 floats: JpJdF (BA.cpp:2066-2080) and the terms of Hdd / bd / Hcd (BA.cpp:1747-1750).  The full records are
//     re-materialised on demand by k_ba_linearize (cml_materialize_records).
#include "cmlhip_internal.h"
#include "ba_common.h"
#include <cstdlib>
#include <cstddef>
#include <type_traits>

#pragma clang fp contract(off)

typedef float float4_ __attribute__((ext_vector_type(4)));
typedef int rs_int8 __attribute__((ext_vector_type(8)));
static_assert(offsetof(cmlhip_ba_pair, R0) == 0x60 && offsetof(cmlhip_ba_pair, t0) == 0xa8, "cmlhip_ba_pair layout (explicit scalar loads in k_ba_lin_rs)");

template <bool HALF>
__device__ __forceinline__ float4 rs_load_texel(const void* img, size_t i) {
    if (HALF) {
        uint2 v = reinterpret_cast<const uint2*>(img)[i];
        __half2 a = *reinterpret_cast<__half2*>(&v.x), b = *reinterpret_cast<__half2*>(&v.y);
        return make_float4(__low2float(a), __high2float(a), __low2float(b), 0.f);
    }
    return reinterpret_cast<const float4*>(img)[i];
}

// fp64 division x / z as the compiler lowers it (v_rcp_f64, two Newton steps, quotient, remainder, one correction), WITHOUT the
// v_div_scale / v_div_fixup wrapping that only acts on operands at the ends of the exponent range or on non-finite ones: for every
// finite operand pair in the normal range the bits are those of the IEEE quotient; a zero, infinite or NaN denominator yields a
// non-finite result here as there (inf may become NaN: every consumer below only asks whether the value is inside the image).
// Splitting it lets the two projections of a pixel (x/z, y/z) share the reciprocal.
__device__ __forceinline__ double rs_rcp_refined(const double z) {
    double r = __builtin_amdgcn_rcp(z);
    double e = __builtin_fma(-z, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-z, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
__device__ __forceinline__ double rs_div(const double x, const double z, const double r) {
    const double q = x * r;
    const double rem = __builtin_fma(-z, q, x);
    return __builtin_fma(rem, r, q);
}

// matrix-core operand offsets into the staged record, per lane (e = lane & 15, kq = lane >> 4), the formulation of acc_pair_block
// (ba_accumulate.hip) made branch-free: A = S[a], B = S[o3] * S[o1] + S[o4] * S[o2], with a slot of ones (39) and a slot of zeros (20)
__constant__ unsigned c_rs_mfma_off[64] = {0x1816100Cu, 0x1816110Du, 0x1816120Eu, 0x1816130Fu, 0x18160600u, 0x18160701u, 0x18160802u, 0x18160903u, 0x18160A04u, 0x18160B05u, 0x1427141Au, 0x1427141Bu, 0x14271422u, 0x14141414u, 0x14141414u, 0x14141414u, 0x1918100Cu, 0x1918110Du, 0x1918120Eu, 0x1918130Fu, 0x19180600u, 0x19180701u, 0x19180802u, 0x19180903u, 0x19180A04u, 0x19180B05u, 0x1427141Cu, 0x1427141Du, 0x14271423u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x1427141Eu, 0x14271420u, 0x14271424u, 0x14271421u, 0x14271425u, 0x14271426u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u};
__constant__ unsigned char c_rs_mfma_a[64] = {12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 20, 20, 20, 20, 20, 20, 16, 17, 18, 19, 6, 7, 8, 9, 10, 11, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 39, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20};

#ifndef RS_DEPTH
#define RS_DEPTH 3           // pixels whose texels are in flight ahead of the sums (3 or 5)
#endif
#define RS_SSTRIDE 41        // floats per residual of the staged reduced record (odd: conflict-free lane-per-record accesses)

// star8 pattern offsets, types.h:1381-1393
#define RS_OX(k) ((k) == 0 ? 0 : (k) == 1 ? -1 : (k) == 2 ? 1 : (k) == 3 ? -2 : (k) == 4 ? 0 : (k) == 5 ? 2 : (k) == 6 ? -1 : 0)
#define RS_OY(k) ((k) == 0 ? -2 : (k) == 1 ? -1 : (k) == 2 ? -1 : (k) == 3 ? 0 : (k) == 4 ? 0 : (k) == 5 ? 0 : (k) == 6 ? 1 : 2)

// per-residual element of an array through a 32-bit byte offset from the (scalar) base: one offset VGPR serves every array of the
// same element size (global_load ... v_off, s[base:base+1]) instead of a 64-bit address pair per array
template <class T> __device__ __forceinline__ T& rs_at(T* base, unsigned byte_off) { return *reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(base)) + byte_off); }
template <class T> __device__ __forceinline__ const T& rs_at(const T* base, unsigned byte_off) { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off); }

// the two texels of a bilinear row (x, x+1) as ONE load: fp16 texels are 8 B, the pair is 16 B at 8-byte alignment (global loads
// only need dword alignment on gfx950); fp32 texels stay two 16-B loads
typedef unsigned rs_u4v __attribute__((ext_vector_type(4)));
typedef rs_u4v rs_u4v_a8 __attribute__((aligned(8)));
typedef float rs_f4v __attribute__((ext_vector_type(4)));
template <bool HALF, int LDM> struct RsRow;           // LDM (development): 0 = plain loads, 1 = nontemporal
template <int LDM> struct RsRow<true, LDM> {
    // fp16 texels come from the TILED level 0 (cml_tiled_level0, cmlhip_internal.h): texel (ix, row) and its right neighbour are 12
    // contiguous bytes at offset (ix & 3) * 6 of a 32-byte tile row — one 16-byte load from the dword below it, then a 0- or 16-bit
    // funnel shift (v_alignbit) lines the six halves up
    unsigned e0, e1, e2;
    rs_u4v raw; unsigned sh;
    __device__ __forceinline__ void load(const void* img, int tw, int ix, int row) {
        const unsigned o = (unsigned)(ix & 3) * 6u;
        const size_t byte = ((size_t)(row >> 2) * tw + (ix >> 2)) * 128u + (unsigned)(row & 3) * 32u + (o & ~3u);
        typedef rs_u4v rs_u4v_a4 __attribute__((aligned(4)));
        const rs_u4v_a4* p = reinterpret_cast<const rs_u4v_a4*>(reinterpret_cast<const char*>(img) + byte);
        raw = LDM ? __builtin_nontemporal_load(p) : *p;
        sh = (o & 2u) * 8u;
    }
    __device__ __forceinline__ void unpack() {
        e0 = __builtin_amdgcn_alignbit(raw.y, raw.x, sh); e1 = __builtin_amdgcn_alignbit(raw.z, raw.y, sh); e2 = __builtin_amdgcn_alignbit(raw.w, raw.z, sh);
    }
    static __device__ __forceinline__ float lo(unsigned u) { return __low2float(*reinterpret_cast<const __half2*>(&u)); }
    static __device__ __forceinline__ float hi(unsigned u) { return __high2float(*reinterpret_cast<const __half2*>(&u)); }
    __device__ __forceinline__ float I0() const { return lo(e0); }
    __device__ __forceinline__ float X0() const { return hi(e0); }
    __device__ __forceinline__ float Y0() const { return lo(e1); }
    __device__ __forceinline__ float I1() const { return hi(e1); }
    __device__ __forceinline__ float X1() const { return lo(e2); }
    __device__ __forceinline__ float Y1() const { return hi(e2); }
};
template <int LDM> struct RsRow<false, LDM> {
    rs_f4v a, b;
    __device__ __forceinline__ void load(const void* img, int w, int ix, int row) {
        const rs_f4v* p = reinterpret_cast<const rs_f4v*>(img) + ((size_t)row * w + ix);
        a = LDM ? __builtin_nontemporal_load(p) : p[0]; b = LDM ? __builtin_nontemporal_load(p + 1) : p[1];
    }
    __device__ __forceinline__ void unpack() {}
    __device__ __forceinline__ float I0() const { return a.x; }
    __device__ __forceinline__ float X0() const { return a.y; }
    __device__ __forceinline__ float Y0() const { return a.z; }
    __device__ __forceinline__ float I1() const { return b.x; }
    __device__ __forceinline__ float X1() const { return b.y; }
    __device__ __forceinline__ float Y1() const { return b.z; }
};

// slots of the staged row that hold something else until the sums are staged: the point's pattern colours / weights (read by the
// pixel loop from LDS instead of sixteen registers) and the depth Jacobian
#define RS_S_COL 22          // 22..29 colours, 30..37 weights: overwritten by the sums after the pixel loop
#define RS_S_D0 21
#define RS_S_D1 40

template <bool HALF, int WPE, int WPB, int LDM>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(WPE))) void k_ba_lin_rs(BAArgs A, RsArgs X) {
    // staged reduced record of the wave's residuals: the layout of k_ba_acc's s_rec (0..5 Jpdxi[0], 6..11 Jpdxi[1], 12..15 Jpdc[0], 16..19 Jpdc[1],
    // 20 zeros, 22..25 JIdx2, 26..29 JabJIdx, 30..33 Jab2, 34,35 JI^T r, 36,37 Jab^T r, 38 r^T r, 39 ones)
    // WPB independent waves per workgroup (no workgroup barrier anywhere): a workgroup is the unit the dispatcher hands out
    __shared__ float s_stg_all[WPB * (RS_TILE + 1) * RS_SSTRIDE];      // (+ one row of zeros per wave: the padding of the matrix-core loop)
    const int ln = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* s_stg = s_stg_all + wv * ((RS_TILE + 1) * RS_SSTRIDE);
    if (A.ctl && A.ctl->stop_lin) return;                  // converged in an earlier launch (raised by k_ba_acc), BA.cpp:879
    const int ti = blockIdx.x * WPB + wv;
    if (ti >= X.ntiles) return;
#ifdef CML_RS_STAMPS                                       // development build (CML_HIPCC_EXTRA=-DCML_RS_STAMPS): per-tile phase stamps, tools/probe_rs_tiles.py
    long long* const ts = (A.dbg && ti < CML_DEBUG_RS_TILES) ? A.dbg + CMLHIP_DEBUG_SLOTS + 8 * (size_t)ti : nullptr;
#define RS_STAMP(i) do { if (ts && ln == 0) ts[i] = wall_clock64(); } while (0)
#else
    long long* const ts = nullptr;
#define RS_STAMP(i) do { } while (0)
#endif
    // ---- wave-uniform data: tile -> pair record, frames (scalar loads, before any store)
    const int4 T = X.tiles[ti];                            // {first residual, count, host, target}
    const int first = T.x, cnt = T.y, host = T.z, target = T.w;
    const cmlhip_ba_pair* pc = &A.pairs[host * A.N + target];
    const FrameDev fh = A.frames[host], ft = A.frames[target];
    double R0_ = pc->R[0], R1_ = pc->R[1], R2_ = pc->R[2], R3_ = pc->R[3], R4_ = pc->R[4], R5_ = pc->R[5],
           R6_ = pc->R[6], R7_ = pc->R[7], R8_ = pc->R[8];
    double t0_ = pc->t[0], t1_ = pc->t[1], t2_ = pc->t[2];
    const double aff_a = pc->aff_a, aff_b = pc->aff_b;
    float th = fh.frame_energy_th > ft.frame_energy_th ? fh.frame_energy_th : ft.frame_energy_th;             // BA.cpp:297-300
    asm volatile("" : "+v"(th));                           // evaluated here: one register held over the kernel instead of two scalars and a late compare
    // each entry an opaque scalar: as a <4 x double> load the vectoriser shuffles two of them together, which the backend lowers
    // through a stack temporary — and a kernel with a scratch frame pays for it at every dispatch
#define RS_OPAQUE(x) asm volatile("" : "+s"(x))
    RS_OPAQUE(R0_); RS_OPAQUE(R1_); RS_OPAQUE(R2_); RS_OPAQUE(R3_); RS_OPAQUE(R4_); RS_OPAQUE(R5_); RS_OPAQUE(R6_); RS_OPAQUE(R7_); RS_OPAQUE(R8_);
    RS_OPAQUE(t0_); RS_OPAQUE(t1_); RS_OPAQUE(t2_);
#undef RS_OPAQUE
    {   // development (CMLHIP_RS_DBG bits 8..): stagger the waves of a SIMD by their slot, n x 0.43 us per slot
        const int stag = (X.dbg_flags >> 8) * (int)(__builtin_amdgcn_s_getreg(63492) & 15u);
        for (int i = 0; i < stag; i++) __builtin_amdgcn_s_sleep(16);
    }
    const long long ts_begin = ts ? wall_clock64() : 0;    // (taken and stored behind the scalar loads above: anything with a side effect before them turns them into vector loads)
    if (ts && ln == 0) { ts[0] = ts_begin; ts[7] = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(63508) << 32); }

    // ---- per-residual inputs, all addressed by the residual index
    const bool valid = ln < cnt;
    const int r = first + (valid ? ln : 0);
    const unsigned r1 = (unsigned)r, r4 = r1 * 4u;          // byte offsets of the residual in 1- and 4-byte arrays
    const int lin_ = rs_at(A.r_lin, r1), st_ = rs_at(A.r_state, r4);
    const double cxd = (double)rs_at(X.r_px, r4), cyd = (double)rs_at(X.r_py, r4);
    float* S = &s_stg[ln * RS_SSTRIDE];
    {
        const float4 colA = rs_at(reinterpret_cast<const float4*>(X.r_colors), r4 * 8u), colB = rs_at(reinterpret_cast<const float4*>(X.r_colors), r4 * 8u + 16u);
        const float4 wgtA = rs_at(reinterpret_cast<const float4*>(X.r_weights), r4 * 8u), wgtB = rs_at(reinterpret_cast<const float4*>(X.r_weights), r4 * 8u + 16u);
        S[RS_S_COL + 0] = colA.x; S[RS_S_COL + 1] = colA.y; S[RS_S_COL + 2] = colA.z; S[RS_S_COL + 3] = colA.w;
        S[RS_S_COL + 4] = colB.x; S[RS_S_COL + 5] = colB.y; S[RS_S_COL + 6] = colB.z; S[RS_S_COL + 7] = colB.w;
        S[RS_S_COL + 8] = wgtA.x; S[RS_S_COL + 9] = wgtA.y; S[RS_S_COL + 10] = wgtA.z; S[RS_S_COL + 11] = wgtA.w;
        S[RS_S_COL + 12] = wgtB.x; S[RS_S_COL + 13] = wgtB.y; S[RS_S_COL + 14] = wgtB.z; S[RS_S_COL + 15] = wgtB.w;
    }
    // the lane's matrix-core operand offsets (c_rs_mfma_*): requested HERE with the inputs — left to the compiler the two table loads
    // sink to the matrix-core loop at the end of the kernel, an exposed memory round trip
    unsigned mf_off = c_rs_mfma_off[ln];
    int mf_a = c_rs_mfma_a[ln];
    const double idepth = rs_at(X.r_idepth, r4 * 2u);                   // == pt_idepth[r_point[r]] (cml_launch_linearize_rs refreshes the copies when needed)
    const bool live = valid && !lin_;
    const int st = live ? st_ : CMLHIP_RES_OOB;
    const bool run = live && st != CMLHIP_RES_OOB;

    // ---- projection of the 8 pattern pixels, BA.cpp:193-212; the centre (BA.cpp:102-131) is pattern pixel 4, offset (0,0): the very
    //      same expressions on the very same operands
    const double tid0 = t0_ * idepth, tid1 = t1_ * idepth, tid2 = t2_ * idepth;
    asm volatile("" : "+v"(mf_off), "+v"(mf_a));            // (pins the table loads to the input round trip)
#ifdef CML_RS_STAMPS
    { double dep = tid0 + cxd + (double)st; asm volatile("" : "+v"(dep)); RS_STAMP(1); }        // every input of the lane has arrived
#endif
    float kxf[8], kyf[8];
    unsigned m_in = 0;
    double rx = 0, ry = 0, px = 0, py = 0, Kud = 0, Kvd = 0;
    float drescale = 0.f;
    // The kernel is a software pipeline over the pattern pixels (stages pinned by scheduling barriers): projection P(k), the two
    // 16-byte texel loads L(k) of its bilinear rows, and the photometric sums S(k) — always in pattern order — run interleaved,
    //     P4 P0 L0 P1 L1 [geometry] P2 L2 | S0 P3 L3 | S1 L4 | S2 P5 L5 | S3 P6 L6 | S4 P7 L7 | S5 S6 S7,
    // so that the sixteen scattered loads of a lane (64 distinct lines per instruction: ~64 cycles of the CU's address pipeline each)
    // are spread over the whole arithmetic instead of arriving from all twelve waves of the CU at once, and only three pixels'
    // texels are in flight per lane.  The centre (pattern pixel 4, offset (0,0)) goes first: every sampling mask needs it.
#define RS_PROJ(k) do { \
        const double sx = cxd + RS_OX(k), sy = cyd + RS_OY(k); \
        const double qx = (sx - A.cx) * A.fxi, qy = (sy - A.cy) * A.fyi; \
        const double ppx = (R0_ * qx + R1_ * qy + R2_ * 1.0) + tid0; \
        const double ppy = (R3_ * qx + R4_ * qy + R5_ * 1.0) + tid1; \
        const double ppz = (R6_ * qx + R7_ * qy + R8_ * 1.0) + tid2; \
        const double rz = rs_rcp_refined(ppz); \
        const double kx = rs_div(ppx, ppz, rz) * A.fx + A.cx, ky = rs_div(ppy, ppz, rz) * A.fy + A.cy; \
        if (kx >= 2 && ky >= 2 && kx < A.w - 2 && ky < A.h - 2) m_in |= 1u << (k); \
        kxf[k] = (float)kx; kyf[k] = (float)ky; \
        if ((k) == 4) { \
            rx = qx; ry = qy; px = ppx; py = ppy; Kud = kx; Kvd = ky; \
            drescale = (float)rs_div(1.0, ppz, rz);            /* (float)(1.0 / pz) */ \
        } \
        __builtin_amdgcn_sched_barrier(0); } while (0)
    // GradientImage::interpolate (Array2D.h:265-286): unconditional loads on clamped addresses (a lane that does not sample reads texel 0)
    RsRow<HALF, LDM> t0r[8], t1r[8];
#define RS_LOAD(k) do { \
        const bool smp_ = run && centre_in && ((m_in >> (k)) & 1u) && !(X.dbg_flags & 1); \
        int ix_ = smp_ ? (int)kxf[k] : 0, iy_ = smp_ ? (int)kyf[k] : 0; \
        if (X.dbg_flags & 4) { ix_ = (ln & 7) * 64; iy_ = (ln >> 3) * 4 + ((X.dbg_flags & 8) ? (ti & 31) * 32 : 0); }     /* development: a line per lane, always cached (8: per tile) */ \
        t0r[k].load(img0, ldw, ix_, iy_); t1r[k].load(img0, ldw, ix_, iy_ + 1); \
        __builtin_amdgcn_sched_barrier(0); } while (0)
    const void* const img0 = HALF ? ft.grad0t : ft.grad0;                         // fp16: the tiled copy
    const int ldw = HALF ? (A.w + CML_TILE_W - 1) / CML_TILE_W : A.w;             // tiles per tile row / texels per row
    RS_PROJ(4);
    const bool centre_in = (m_in >> 4) & 1u;
    RS_PROJ(0); RS_LOAD(0); RS_PROJ(1); RS_LOAD(1);
#if RS_DEPTH >= 5
    RS_PROJ(2); RS_LOAD(2);
#endif
    RS_STAMP(2);

    // ---- geometric Jacobians, BA.cpp:120-188 (the expression shapes of k_ba_linearize with its per-lane constants folded), evaluated
    //      while the texels are in flight and parked in the staged row: nothing of the geometry stays in registers over the pixel loop
    {
        const float new_idepth = (float)(drescale * idepth);
        if (run && centre_in) {                                  // setCenterProjectedTo, :131
            rs_at(A.r_center, r4 * 3u) = (float)Kud; rs_at(A.r_center, r4 * 3u + 4u) = (float)Kvd;
            rs_at(A.r_center, r4 * 3u + 8u) = new_idepth;
        }
        // evaluation-point pair (PRE_RTll_0 / PRE_tTll_0) for the calibration / depth Jacobians: explicit scalar loads HERE (the
        // compiler only scalarises loads it can prove unclobbered, i.e. before the first store of the kernel; holding these 18
        // SGPRs across the projection loop spilled scalars, and a kernel with a scratch frame pays for it at every dispatch)
        rs_int8 w0, w1, w2;
        asm volatile("s_load_dwordx8 %0, %3, 0x60\n\ts_load_dwordx8 %1, %3, 0x80\n\ts_load_dwordx8 %2, %3, 0xa0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(w0), "=&s"(w1), "=&s"(w2) : "s"(pc) : "memory");
#define RS_D(w, i) __hiloint2double((w)[2 * (i) + 1], (w)[2 * (i)])
        const double E0 = RS_D(w0, 0), E1 = RS_D(w0, 1), E3 = RS_D(w0, 3), E4 = RS_D(w1, 0), E6 = RS_D(w1, 2), E7 = RS_D(w1, 3);   // R0[0,1,3,4,6,7]
        const double et0 = RS_D(w2, 1), et1 = RS_D(w2, 2), et2 = RS_D(w2, 3);                                                     // t0[0..2]
#undef RS_D
        const float u = (float)px, v = (float)py;            // BA.cpp:121-122: un-normalised x,y, literal
        const float fxf = (float)A.fx, fyf = (float)A.fy;
        const double rfx = rs_rcp_refined((double)fxf), rfy = rs_rcp_refined((double)fyf);      // wave-uniform
        // Jpdxi (fp32, BA.cpp:133-147)
        const float xi0[6] = {new_idepth * fxf, 0.f, -new_idepth * u * fxf, -u * v * fxf, (1 + u * u) * fxf, -v * fxf};
        const float xi1[6] = {0.f, new_idepth * fyf, -new_idepth * v * fyf, -(1 + v * v) * fyf, u * v * fyf, u * fyf};
        // Jpdc (:150-176): q = (sfac * drescale) * (Ea * w - Eb) / s2, entry = ((m * q) + add) * scale; entries 0,2 / 1,3 / 4,6 / 5,7 share q
        const double q02 = (double)(1.f * drescale) * (E6 * u - E0);
        const double q13 = rs_div((double)(fxf * drescale) * (E7 * u - E1), (double)fyf, rfy);
        const double q46 = rs_div((double)(fyf * drescale) * (E6 * v - E3), (double)fxf, rfx);
        const double q57 = (double)(1.f * drescale) * (E7 * v - E4);
        const float c0[4] = {(float)(((rx * q02) + (double)u) * A.scale_f), (float)(((ry * q13) + -0.0) * A.scale_f),
                             (float)(((1.0 * q02) + 1.0) * A.scale_c), (float)(((1.0 * q13) + -0.0) * A.scale_c)};
        const float c1[4] = {(float)(((rx * q46) + -0.0) * A.scale_f), (float)(((ry * q57) + (double)v) * A.scale_f),
                             (float)(((1.0 * q46) + -0.0) * A.scale_c), (float)(((1.0 * q57) + 1.0) * A.scale_c)};
        // Jpdd (:178-182)
        const float d0 = (float)(drescale * (et0 - et2 * u) * fxf), d1 = (float)(drescale * (et1 - et2 * v) * fyf);
#pragma unroll
        for (int i = 0; i < 6; i++) { S[i] = xi0[i]; S[6 + i] = xi1[i]; }
#pragma unroll
        for (int i = 0; i < 4; i++) { S[12 + i] = c0[i]; S[16 + i] = c1[i]; }
        S[RS_S_D0] = d0; S[RS_S_D1] = d1;
        S[20] = 0.f; S[39] = 1.f;
    }
    RS_STAMP(3);
    __builtin_amdgcn_sched_barrier(0);
#if RS_DEPTH >= 5
    RS_PROJ(3); RS_LOAD(3); RS_LOAD(4);
#else
    RS_PROJ(2); RS_LOAD(2);
#endif
    // what the classification needs of the previous state: requested here, behind the texels, so that the round trip runs under the pixel loop
    const float pre_energy = rs_at(A.r_energy, r4);
    const float pre_new_energy = rs_at(A.r_new_energy, r4); // (read here: behind the stores of the classification it would wait for every one of them)
    const int pre_new_state = rs_at(A.r_new_state, r4), pre_ppos = rs_at(A.point_pos, r4);
    const unsigned char pre_sel = rs_at(A.r_sel, r1);

    // ---- photometric terms and pattern sums, pixel by pixel in pattern order (BA.cpp:214-271 and the ACTIVE-mode inner products of
    //      BA.cpp:1719-1729).  Form A: acc = (float)((double)acc + X*Y); form B: acc += rF*Y in fp64 (a masked column adds rF * 0);
    //      form C: acc += ((p*q)*r)*s in fp32 — the forms and operand conversions of k_ba_linearize.
    // RS_XFMA(x, y, z) = z + x*y where x and y are floats widened to double: the 48-bit product is exact in fp64, so the fused form
    // rounds once exactly where the reference's separate multiply and add round once — same bits, one instruction less
#define RS_XFMA(x, y, z) __builtin_fma((x), (y), (z))
    float J00 = 0, J10 = 0, J11 = 0, Q00 = 0, Q10 = 0, Q01 = 0, Q11 = 0, rr = 0, E = 0, wJI2 = 0;
    double JIr0 = 0, JIr1 = 0, Jabr0 = 0, Jabr1 = 0;
    float B00 = 0, B01 = 0, B11 = 0;
    unsigned m_nf = 0;
#define RS_SUM(k) do { \
        t0r[k].unpack(); t1r[k].unpack(); \
        const bool smp = run && centre_in && ((m_in >> (k)) & 1u); \
        const float x = kxf[(k)], y = kyf[(k)]; \
        const int ix = (int)x, iy = (int)y; \
        const float dx = x - (float)ix, dy = y - (float)iy; \
        const float dxdy = dx * dy; \
        const float tw00 = 1 - dx - dy + dxdy, tw01 = dx - dxdy, tw10 = dy - dxdy, tw11 = dxdy; \
        const float Iv = t0r[(k)].I0() * tw00 + t0r[(k)].I1() * tw01 + t1r[(k)].I0() * tw10 + t1r[(k)].I1() * tw11; \
        const float gxv = t0r[(k)].X0() * tw00 + t0r[(k)].X1() * tw01 + t1r[(k)].X0() * tw10 + t1r[(k)].X1() * tw11; \
        const float gyv = t0r[(k)].Y0() * tw00 + t0r[(k)].Y1() * tw01 + t1r[(k)].Y0() * tw10 + t1r[(k)].Y1() * tw11; \
        const float I = smp ? Iv : 0.f, gx = smp ? gxv : 0.f, gy = smp ? gyv : 0.f; \
        const bool finite = isfinite(I) && isfinite(gx) && isfinite(gy); \
        if (((m_in >> (k)) & 1u) && !finite) m_nf |= 1u << (k); \
        const float refColor = S[RS_S_COL + (k)]; \
        const float refRealColor = (float)(aff_a * (double)refColor + aff_b); \
        const float residual = I - refRealColor; \
        float hw = fabs((double)residual) < A.huber_d ? 1.0f : (float)(A.huber_d / (double)fabsf(residual)); \
        const double wden = A.oth_d + (double)(gx * gx + gy * gy); \
        float wgt = sqrtf((float)rs_div(A.oth_d, wden, rs_rcp_refined(wden))); \
        wgt = (float)(0.5f * ((double)wgt + (double)S[RS_S_COL + 8 + (k)])); \
        const float pf = wgt * wgt * hw * residual * residual;      /* energy term factor, :237 */ \
        const float hw0 = hw; \
        if (hw < 1) hw = sqrtf(hw); \
        hw = hw * wgt; \
        const float f1 = gx * hw, f2 = gy * hw;                     /* hitColor[1], hitColor[2] */ \
        const float drdA = I - fh.b0; \
        const float a_ = drdA * hw; \
        const float rF = residual * hw; \
        const double f1d = (double)f1, f2d = (double)f2, ad = (double)a_, hwd = (double)hw, rFd = (double)rF; \
        J00 = (float)RS_XFMA(f1d, f1d, (double)J00); J10 = (float)RS_XFMA(f1d, f2d, (double)J10); J11 = (float)RS_XFMA(f2d, f2d, (double)J11); \
        Q00 = (float)RS_XFMA(ad, f1d, (double)Q00); Q10 = (float)RS_XFMA(hwd, f1d, (double)Q10); \
        Q01 = (float)RS_XFMA(ad, f2d, (double)Q01); Q11 = (float)RS_XFMA(hwd, f2d, (double)Q11); \
        rr = (float)RS_XFMA(rFd, rFd, (double)rr); \
        E = (float)((double)E + (double)pf * (2.0 - (double)hw0));                            /* energyLeft, BA.cpp:237 */ \
        wJI2 = (float)((double)wJI2 + (double)(hw * hw) * RS_XFMA(f2d, f2d, f1d * f1d));           /* wJI2_sum, BA.cpp:257 */ \
        JIr0 = RS_XFMA(rFd, f1d, JIr0); JIr1 = RS_XFMA(rFd, f2d, JIr1); \
        Jabr0 = RS_XFMA(rFd, (A.opt_a ? ad : 0.0), Jabr0); Jabr1 = RS_XFMA(rFd, (A.opt_b ? hwd : 0.0), Jabr1);           /* BA.cpp:273-278: a zeroed column contributes rF * 0 */ \
        B00 += drdA * drdA * hw * hw; B01 += drdA * hw * hw * 1.f; B11 += hw * hw * 1.f * 1.f; \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#if RS_DEPTH >= 5
    RS_SUM(0); RS_PROJ(5); RS_LOAD(5);
    RS_SUM(1); RS_PROJ(6); RS_LOAD(6);
    RS_SUM(2); RS_PROJ(7); RS_LOAD(7);
    RS_SUM(3); RS_SUM(4); RS_SUM(5); RS_SUM(6); RS_SUM(7);
#else
    RS_SUM(0); RS_PROJ(3); RS_LOAD(3);
    RS_SUM(1); RS_LOAD(4);
    RS_SUM(2); RS_PROJ(5); RS_LOAD(5);
    RS_SUM(3); RS_PROJ(6); RS_LOAD(6);
    RS_SUM(4); RS_PROJ(7); RS_LOAD(7);
    RS_SUM(5); RS_SUM(6); RS_SUM(7);
#endif
#undef RS_SUM
#undef RS_LOAD
#undef RS_PROJ

#ifdef CML_RS_STAMPS
    { float dep = J00 + E + B11; asm volatile("" : "+v"(dep)); RS_STAMP(4); }
#endif
    // first failing pixel in pattern order decides between setNewState(OOB) (:209-212) and setState(OOB) (:220-223)
    const unsigned m_oob = ~m_in & 0xFFu;
    const unsigned m_bad = m_oob | m_nf;
    const int first_bad = m_bad ? __ffs((int)m_bad) - 1 : 8;
    const bool fail_new_oob = !centre_in || (m_bad && ((m_oob >> first_bad) & 1u));
    const bool fail_state_oob = centre_in && m_bad && !((m_oob >> first_bad) & 1u);

    // ---- classification, BA.cpp:66-72,115-118,297-314, and the fused applyRes(copyJacobians = true), BA.cpp:2051-2093
    // From here on the kernel arguments are read AGAIN from the argument segment (through a pointer the compiler cannot connect with
    // the first reads): otherwise the pointers of the stores below are held — or split, spilled and rematerialised, leaving a scratch
    // frame behind — across the pixel loop.
    typedef const __attribute__((address_space(4))) char* rs_karg_ptr;
    rs_karg_ptr kargs = (rs_karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kargs));
    const __attribute__((address_space(4))) BAArgs& B = *(const __attribute__((address_space(4))) BAArgs*)kargs;
    const __attribute__((address_space(4))) RsArgs& Y = *(const __attribute__((address_space(4))) RsArgs*)(kargs + ((sizeof(BAArgs) + alignof(RsArgs) - 1) / alignof(RsArgs)) * alignof(RsArgs));
    double ret_d = 0.0;
    int ns_cnt = -1, flip = 0;
    if (live) {
        float ret = pre_energy;
        float nwo = -1.f;
        int ns_final = pre_new_state;
        bool state_now_oob = (st == CMLHIP_RES_OOB), wrote_e = false;
        if (run) {
            if (fail_new_oob) {
                ns_final = CMLHIP_RES_OOB;
            } else if (fail_state_oob) {
                rs_at(B.r_state, r4) = CMLHIP_RES_OOB;
                state_now_oob = true;
            } else if (!isfinite(E)) {
                ns_final = CMLHIP_RES_OOB;
            } else {
                nwo = E;
                float e = E;
                ns_final = CMLHIP_RES_IN;
                if (E > th || wJI2 < 2) { e = th; ns_final = CMLHIP_RES_OUTLIER; }
                rs_at(B.r_new_energy, r4) = e;
                ret = e;
                wrote_e = true;
            }
            rs_at(B.r_new_state, r4) = ns_final;
        }
        rs_at(B.r_new_energy_wo, r4) = nwo;
        rs_at(B.r_ret_energy, r4) = ret;
        ret_d = (double)ret; ns_cnt = ns_final;
        int code = -1;
        if (!state_now_oob) {                                       // applyRes
            if (ns_final == CMLHIP_RES_IN) { rs_at(B.r_good, r1) = 1; flip = 1; code = 2 * r + pre_sel; }
            else rs_at(B.r_good, r1) = 0;
            rs_at(B.r_state, r4) = ns_final;
            rs_at(B.r_energy, r4) = wrote_e ? ret : pre_new_energy;      // state_energy = state_NewEnergy
            rs_at(B.point_code, (unsigned)pre_ppos * 4u) = code;                          // read by the point rows of k_ba_acc and by k_ba_backsub
        }
    }

    RS_STAMP(5);
    if (flip) {
        // ---- per residual: JpJdF (BA.cpp:2066-2080) and the terms of Hcd, Hdd, bd (BA.cpp:1747-1750); the geometry comes back from the staged row
        if (!(Y.dbg_flags & 2)) {
            const float d0 = S[RS_S_D0], d1 = S[RS_S_D1];
            const float g0 = J00 * d0 + J10 * d1;
            const float g1 = J10 * d0 + J11 * d1;
            float4* o = &rs_at(reinterpret_cast<float4*>(B.r_jpjdf), r4 * (unsigned)PS_STRIDE);
            o[0] = make_float4(S[0] * g0 + S[6] * g1, S[1] * g0 + S[7] * g1, S[2] * g0 + S[8] * g1, S[3] * g0 + S[9] * g1);
            o[1] = make_float4(S[4] * g0 + S[10] * g1, S[5] * g0 + S[11] * g1, Q00 * d0 + Q01 * d1, Q10 * d0 + Q11 * d1);
            o[2] = make_float4(S[12] * g0 + S[16] * g1, S[13] * g0 + S[17] * g1, S[14] * g0 + S[18] * g1, S[15] * g0 + S[19] * g1);
            o[3] = make_float4(d0 * g0 + d1 * g1, (float)((double)(float)JIr0 * (double)d0 + (double)(float)JIr1 * (double)d1), 0.f, 0.f);
        }
        // ---- staged operands of the matrix-core reduction
        S[22] = J00; S[23] = J10; S[24] = J10; S[25] = J11;
        S[26] = Q00; S[27] = Q10; S[28] = Q01; S[29] = Q11;
        S[30] = B00; S[31] = B01; S[32] = B01; S[33] = B11;
        S[34] = (float)JIr0; S[35] = (float)JIr1; S[36] = (float)Jabr0; S[37] = (float)Jabr1;
        S[38] = rr;
    }                                                        // (rows of residuals that are not IN are never read: the loop below walks the IN mask)
    if (ln < RS_SSTRIDE) s_stg[RS_TILE * RS_SSTRIDE + ln] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the workgroup is ONE wave and a wave's LDS operations execute in order:
    __builtin_amdgcn_wave_barrier();                        // only the compiler has to be kept from moving the reads up

    // ---- the wave's contribution to the 13x13 block of its pair: one v_mfma_f32_16x16x4_f32 per residual (see acc_pair_block)
    {
        const int oa = mf_a, o1 = mf_off & 255, o2 = (mf_off >> 8) & 255, o3 = (mf_off >> 16) & 255, o4 = mf_off >> 24;
        float4_ acc = {0.f, 0.f, 0.f, 0.f};
        // only the residuals that are IN contribute (every product of a staged-zero row is +0, and acc + 0 == acc): the loop walks the
        // set bits of the wave's IN mask in ascending order, eight rows per trip, padded with the row of zeros.  All forty operand reads
        // of a trip are issued before the first product (the scheduling barrier keeps them together: one LDS latency per trip, not
        // one per row).
        unsigned long long inm = __ballot(flip != 0);
        while (inm) {
            float av[8], b1[8], b2[8], b3[8], b4[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int li = inm ? __builtin_ctzll(inm) : RS_TILE;
                inm &= inm - 1;
                const float* SL = s_stg + li * RS_SSTRIDE;
                av[u] = SL[oa]; b1[u] = SL[o1]; b2[u] = SL[o2]; b3[u] = SL[o3]; b4[u] = SL[o4];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b3[u] * b1[u] + b4[u] * b2[u], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(Y.dbg_flags & 2)) reinterpret_cast<float4*>(Y.part)[(size_t)ti * 64 + ln] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }

    // ---- per-tile partials {energy, n_in, n_oob, n_outlier} (BA.cpp:1565): fixed butterfly order over the wave's residuals
    if (B.lin_partial) {
        double e = ret_d;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
        const int c0 = __popcll(__ballot(ns_cnt == CMLHIP_RES_IN)), c1 = __popcll(__ballot(ns_cnt == CMLHIP_RES_OOB)), c2 = __popcll(__ballot(ns_cnt == CMLHIP_RES_OUTLIER));
        if (ln == 0) {
            double* o = B.lin_partial + 4 * (size_t)ti;
            o[0] = e; o[1] = (double)c0; o[2] = (double)c1; o[3] = (double)c2;
        }
    }
    RS_STAMP(6);
    // ---- resident loop: the convergence test of doStepFromBackup (BA.cpp:996-1027) on the sums of the step that preceded this
    //      pass; `if (canbreak && it >= 1) break` (BA.cpp:879) becomes a sticky flag that every later kernel checks first
    ResidentCtl* ctl_end = B.ctl;
    asm volatile("" : "+s"(ctl_end));                        // (re-read from the kernel arguments here instead of a flag held in SGPRs over the whole kernel)
    if (ti == 0 && ln == 0 && ctl_end) {
        float sumID = 0, sumNID = 0, numID = 0;
        for (int b = 0; b < B.n_step_blocks; b++) { sumID += B.step_partial_ro[4 * b]; sumNID += B.step_partial_ro[4 * b + 1]; numID += B.step_partial_ro[4 * b + 2]; }
        float sumA = ctl_end->frame_sums[0], sumB_ = ctl_end->frame_sums[1], sumT = ctl_end->frame_sums[2], sumR = ctl_end->frame_sums[3];
        const float nf = (float)B.N;
        sumA /= nf; sumB_ /= nf; sumR /= nf; sumT /= nf; sumID /= numID; sumNID /= numID;
        const bool canbreak = sqrtf(sumA) < 0.0005 * B.th_opt && sqrtf(sumB_) < 0.00005 * B.th_opt && sqrtf(sumR) < 0.00005 * B.th_opt &&
                              sqrtf(sumT) * sumNID < 0.00005 * B.th_opt;
        ctl_end->iters_done = B.it_index + 1;
        if (canbreak && B.it_index >= 1) ctl_end->stop = 1;
    }
}

__global__ void k_ba_idepth_to_res(BAArgs A) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < A.R) A.r_idepth[r] = A.pt_idepth[A.r_point[r]];
}

void cml_refresh_r_idepth(cmlhip_ctx* c, const BAArgs& A, hipStream_t stream) {
    k_ba_idepth_to_res<<<cml_div_up(A.R, 256), 256, 0, stream>>>(A);
}
static void fill_rs_args(cmlhip_ctx* c, RsArgs& X) {
    X.tiles = c->rs_tiles.as<int4>(); X.ntiles = c->n_tiles;
    X.r_px = c->r_px.as<float>(); X.r_py = c->r_py.as<float>(); X.r_colors = c->r_colors.as<float>(); X.r_weights = c->r_weights.as<float>();
    X.part = c->rs_part.as<float>(); X.r_idepth = c->r_idepth.as<double>();
    { static const char* e = getenv("CMLHIP_RS_DBG"); X.dbg_flags = e ? atoi(e) : 0; }      // development: 1 = all texel taps at texel 0, 2 = no tile / reduced-record stores
    X.stop_lin = reinterpret_cast<const int*>(c->scal.as<char>() + CML_ZERO_WORD_OFFSET);
}
int cml_fill_rs4_batch(cmlhip_ctx* c, const BAArgs& A, std::vector<unsigned char>& blob, int& blocks) {
    if (!c->rs_ok || c->rs_tile != 16 || c->n_tiles == 0) {
        c->err = "cmlhip_ba_iteration_batch takes small windows only (4-lane residual kernel, R < 36 k): a larger window fills the chip on its own";
        return CMLHIP_ERR_INVALID;
    }
    BatchRs r;
    memset(&r, 0, sizeof r);
    r.A = A;
    fill_rs_args(c, r.X);
    r.blocks = blocks = cml_div_up(c->n_tiles, 4);
    const unsigned char* b = reinterpret_cast<const unsigned char*>(&r);
    blob.insert(blob.end(), b, b + sizeof r);
    return CMLHIP_OK;
}
int cml_launch_linearize_rs(cmlhip_ctx* c, const BAArgs& A) {
    if (c->n_tiles == 0) return CMLHIP_OK;
    if (c->r_idepth_dirty) {                               // pt_idepth was written outside the resident point step (upload, set_idepth, restore, standalone step)
        cml_refresh_r_idepth(c, A, c->stream);
        c->r_idepth_dirty = false;
    }
    RsArgs X;
    fill_rs_args(c, X);
    if (A.ctl) X.stop_lin = &A.ctl->stop_lin;
    if (c->rs_tile == 16) return cml_launch_linearize_rs4(c, A, X);                         // small window: 4 lanes per residual (ba_linearize_rs4.hip)
    // fp16 texels: 168 VGPRs, three waves per SIMD — every tile of a 20-frame window resident at once; fp32 texels hold twice the
    // registers in flight and stay at two
    static const char* e_mt = getenv("CMLHIP_RS_MAXTILES");    // development: launch only the first n tiles (timing experiments)
    const int nt = e_mt ? (atoi(e_mt) < c->n_tiles ? atoi(e_mt) : c->n_tiles) : c->n_tiles;
    static const char* e_wpb = getenv("CMLHIP_RS_WPB");        // development: waves per workgroup (1 or 4)
    const int wpb = e_wpb ? atoi(e_wpb) : 4;
    X.ntiles = nt;
    static const char* e_ldm = getenv("CMLHIP_RS_LDM");        // development: 1 = nontemporal texel loads
    const int ldm = e_ldm ? atoi(e_ldm) : 0;
    if (c->lim.texel_format == CMLHIP_TEXEL_F16) {
        if (ldm == 1 && wpb == 1) CML_LAUNCH_EV(c, (k_ba_lin_rs<true, 3, 1, 1>), nt, 64, 0, A, X);
        else if (ldm == 1) CML_LAUNCH_EV(c, (k_ba_lin_rs<true, 3, 4, 1>), cml_div_up(nt, 4), 256, 0, A, X);
        else if (wpb == 1) CML_LAUNCH_EV(c, (k_ba_lin_rs<true, 3, 1, 0>), nt, 64, 0, A, X);
        else CML_LAUNCH_EV(c, (k_ba_lin_rs<true, 3, 4, 0>), cml_div_up(nt, 4), 256, 0, A, X);
    } else CML_LAUNCH_EV(c, (k_ba_lin_rs<false, 2, 1, 0>), nt, 64, 0, A, X);
    return CMLHIP_OK;
}
