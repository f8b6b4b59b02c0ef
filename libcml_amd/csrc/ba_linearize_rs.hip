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
// RELAX arithmetic (cmlhip_ba_set_arithmetic): v_rcp_f64 + ONE Newton step (relative error ~2^-50 from the ~2^-26 seed)
__device__ __forceinline__ double rs_rcp_once(const double z) {
    const double r = __builtin_amdgcn_rcp(z);
    return __builtin_fma(r, __builtin_fma(-z, r, 1.0), r);
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

// the target frame of tile ti, re-read from the tile table at the kernel's tail (a scalar load; not a register held over the pixel loop)
__device__ __forceinline__ int rs_tile_target(const int4* tiles, int ti) {
    return *(const __attribute__((address_space(4))) int*)(reinterpret_cast<const int*>(tiles + ti) + 3);
}

template <bool HALF, int WPE, int WPB, int LDM, bool RELAX = false>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(WPE))) void k_ba_lin_rs(BAArgs A, RsArgs X) {
#define RS_BATCH 0
#include "ba_linearize_rs_body.inc"
#undef RS_BATCH
}
// several windows per launch (cmlhip_ba_iteration_batch, windows uploaded in the throughput regime: tiles of 64): gridDim.y = window, the
// body of the solo kernel on the window's own argument block
template <bool HALF, int WPE, int WPB, int LDM, bool RELAX = false>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(WPE))) void k_ba_lin_rs_batch(const BatchRs* __restrict__ W) {
    const BatchRs __attribute__((address_space(4)))* rs_window = (const BatchRs __attribute__((address_space(4)))*)(W + blockIdx.y);     // constant address space: scalar loads
    const BatchRs& rs_w = *(const BatchRs*)rs_window;
    if ((int)blockIdx.x >= rs_w.blocks) return;
    const BAArgs& A = rs_w.A;
    const RsArgs& X = rs_w.X;
#define RS_BATCH 1
#include "ba_linearize_rs_body.inc"
#undef RS_BATCH
}
int cml_launch_linearize_rs_batch(cmlhip_ctx* c0, const void* dev_records, int S, int max_blocks) {
    const BatchRs* W = static_cast<const BatchRs*>(dev_records);
    // (the arithmetic mode of a batch is its first context's: cmlhip_ba_iteration_batch refuses windows that differ)
    if (c0->lim.texel_format == CMLHIP_TEXEL_F16) {
        if (c0->arith_relaxed) k_ba_lin_rs_batch<true, 3, 4, 0, true><<<dim3(max_blocks, S), 256, 0, c0->stream>>>(W);
        else k_ba_lin_rs_batch<true, 3, 4, 0><<<dim3(max_blocks, S), 256, 0, c0->stream>>>(W);
    } else if (c0->arith_relaxed) k_ba_lin_rs_batch<false, 2, 1, 0, true><<<dim3(max_blocks, S), 64, 0, c0->stream>>>(W);
    else k_ba_lin_rs_batch<false, 2, 1, 0><<<dim3(max_blocks, S), 64, 0, c0->stream>>>(W);
    return CMLHIP_OK;
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
    { static const char* e = getenv("CMLHIP_RS_DBG"); X.dbg_flags = e ? (atoi(e) & 255) : 0; }      // development: 1 = all texel taps at texel 0, 2 = no tile / reduced-record stores, 4 / 8 cached taps, 16 / 32 / 64 single stores off
    { static const char* e = getenv("CMLHIP_RS_FULL"); if (c->rs_lean && !e) X.dbg_flags |= RS_LEAN_BIT; }      // (development: CMLHIP_RS_FULL forces the full outputs for an A/B)
    X.stop_lin = reinterpret_cast<const int*>(c->scal.as<char>() + CML_ZERO_WORD_OFFSET);
}
int cml_fill_rs4_batch(cmlhip_ctx* c, const BAArgs& A, std::vector<unsigned char>& blob, int& blocks) {
    if (!c->rs_ok || c->n_tiles == 0) {
        c->err = "cmlhip_ba_iteration_batch: the window has no resident residual tiles";
        return CMLHIP_ERR_INVALID;
    }
    BatchRs r;
    memset(&r, 0, sizeof r);
    r.A = A;
    fill_rs_args(c, r.X);
    // workgroups of the window: 4 tiles of 16 (4-lane kernel) or, in the throughput regime (tiles of 64), 4 waves (fp16 texels) / 1 wave (fp32)
    r.blocks = blocks = (c->rs_tile == 16 || c->lim.texel_format == CMLHIP_TEXEL_F16) ? cml_div_up(c->n_tiles, 4) : c->n_tiles;
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
    // waves that share a SIMD start a third of a wave's life apart (measured at config E, 2.6 waves per SIMD: 31.5 -> 29.6 us); windows of at most
    // one wave per SIMD are unaffected (a lone wave has slot 0)
    { static const char* e_st = getenv("CMLHIP_RS_STAGGER"); X.dbg_flags |= ((e_st ? atoi(e_st) : 8) & 255) << 8; }
    static const char* e_ldm = getenv("CMLHIP_RS_LDM");        // development: 1 = nontemporal texel loads
    const int ldm = e_ldm ? atoi(e_ldm) : 0;
    if (c->arith_relaxed) {                                 // CMLHIP_ARITH_RELAXED: the throughput-regime kernel only (small windows keep the exact 4-lane kernel)
        if (c->lim.texel_format == CMLHIP_TEXEL_F16) CML_LAUNCH_EV(c, (k_ba_lin_rs<true, 3, 4, 0, true>), cml_div_up(nt, 4), 256, 0, A, X);
        else CML_LAUNCH_EV(c, (k_ba_lin_rs<false, 2, 1, 0, true>), nt, 64, 0, A, X);
        return CMLHIP_OK;
    }
    if (c->lim.texel_format == CMLHIP_TEXEL_F16) {
        if (ldm == 1 && wpb == 1) CML_LAUNCH_EV(c, (k_ba_lin_rs<true, 3, 1, 1>), nt, 64, 0, A, X);
        else if (ldm == 1) CML_LAUNCH_EV(c, (k_ba_lin_rs<true, 3, 4, 1>), cml_div_up(nt, 4), 256, 0, A, X);
        else if (wpb == 1) CML_LAUNCH_EV(c, (k_ba_lin_rs<true, 3, 1, 0>), nt, 64, 0, A, X);
        else CML_LAUNCH_EV(c, (k_ba_lin_rs<true, 3, 4, 0>), cml_div_up(nt, 4), 256, 0, A, X);
    } else CML_LAUNCH_EV(c, (k_ba_lin_rs<false, 2, 1, 0>), nt, 64, 0, A, X);
    return CMLHIP_OK;
}
