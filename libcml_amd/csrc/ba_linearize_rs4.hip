// ba_linearize_rs4.hip — residual / Jacobian kernel of the device-RESIDENT Gauss-Newton loop for SMALL windows (latency regime).
// The lane-per-residual kernel of ba_linearize_rs.hip issues ~2600 instructions per wave: right when there are thousands of waves
// (config E), too long a dependent stream when a window has fewer waves than the chip has SIMDs (config B: 224).  Here a residual
// is spread over 4 lanes (16 residuals per wave, ~1200 instructions per wave): more replicated work, shorter critical path.
// Same arithmetic as k_ba_linearize (ba_linearize.hip: DSOBundleAdjustmentLinearizationContext::linearize BA.cpp:62-316 with the
// fused applyRes BA.cpp:2051-2093, statement order kept, FP contraction off), re-mapped for throughput:
//
//   * the device residual order is (host,target)-pair-sorted (cmlhip_ba_upload_window), a WAVE owns 16 residuals of ONE pair: the
//     pair record (R, t, R0, t0, affine), both frame descriptors and the camera are wave-uniform and live in SGPRs (scalar
//     loads, scalar operands) instead of 60 VGPRs per lane;
//   * 4 lanes per residual, two pattern pixels per lane (8 texel loads of a lane in flight together, unconditional on clamped
//     addresses); the 19 pattern sums are spread 5/5/5/4 over the quad and every one is still added in pattern order
//     (per-pixel operands exchanged through wave-private LDS rows: no workgroup barrier anywhere in the kernel);
//   * every per-residual input is addressed directly by the residual index (static copies of the point's pixel, colours and
//     weights are kept per residual), only the inverse depth goes through the point index;
//   * nothing of the 74-float DSORawResidualJacobian is written to memory.  What the next iteration consumes leaves the kernel in
//     reduced form: the wave's contribution to the 13x13 AccumulatorApprox block of its pair (BA.cpp:1731-1745, ACC.h:776-932) as
//     ONE 16x16 fp32 tile accumulated on the matrix cores over the wave's residuals (v_mfma_f32_16x16x4_f32, one per residual,
//     the formulation of k_ba_acc), and per residual 14 floats: JpJdF (BA.cpp:2066-2080) and the terms of Hdd / bd / Hcd
//     (BA.cpp:1747-1750).  The full records are re-materialised on demand by k_ba_linearize (cml_materialize_records).
#include "cmlhip_internal.h"
#include "ba_common.h"
#include <cstdlib>

#pragma clang fp contract(off)

typedef float float4_ __attribute__((ext_vector_type(4)));

template <bool HALF>
__device__ __forceinline__ float4 rs4_load_texel(const void* img, size_t i) {
    // (global address space spelled out: the image pointer comes out of a frame record, and a generic pointer makes these flat loads)
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    typedef const u2v __attribute__((address_space(1)))* g_u2;
    typedef const float4_ __attribute__((address_space(1)))* g_f4;
    if (HALF) {
        const u2v w = ((g_u2)img)[i];
        uint2 v = make_uint2(w.x, w.y);
        __half2 a = *reinterpret_cast<__half2*>(&v.x), b = *reinterpret_cast<__half2*>(&v.y);
        return make_float4(__low2float(a), __high2float(a), __low2float(b), 0.f);
    }
    // (three dwords: with the pad lane loaded too the register allocator hands that dead register to the next temporary and waits for
    //  the load to land first — an s_waitcnt right behind the texel loads)
    typedef float float3_ __attribute__((ext_vector_type(3)));
    const float3_ t = *reinterpret_cast<const float3_ __attribute__((address_space(1)))*>((g_f4)img + i);
    return make_float4(t.x, t.y, t.z, 0.f);
}

// value held by quad lane `L` (0..3) of this lane's group of 4: DPP quad_perm, no LDS
template <int L>
__device__ __forceinline__ int rs4_quad_bcast_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, L * 0x55, 0xf, 0xf, true);       // quad_perm:[L,L,L,L]
}
template <int L>
__device__ __forceinline__ float rs4_quad_bcast_f(float v) { return __int_as_float(rs4_quad_bcast_i<L>(__float_as_int(v))); }
template <int L>
__device__ __forceinline__ double rs4_quad_bcast_d(double v) {
    const int lo = rs4_quad_bcast_i<L>(__double2loint(v)), hi = rs4_quad_bcast_i<L>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// exact select by quad lane without control flow (a nested ?: over lane-varying conditions becomes a branch tree)
__device__ __forceinline__ float rs4_sel4(const int j, const float a, const float b, const float c, const float d) {
    const int m0 = -(int)(j == 0), m1 = -(int)(j == 1), m2 = -(int)(j == 2), m3 = -(int)(j == 3);
    return __int_as_float((__float_as_int(a) & m0) | (__float_as_int(b) & m1) | (__float_as_int(c) & m2) | (__float_as_int(d) & m3));
}
// matrix-core operand offsets into the staged record, per lane (e = lane & 15, kq = lane >> 4), the formulation of acc_pair_block
// (ba_accumulate.hip) made branch-free: A = S[a], B = S[o3] * S[o1] + S[o4] * S[o2], with a slot of ones (39) and a slot of zeros (20)
__constant__ unsigned c_rs4_mfma_off[64] = {0x1816100Cu, 0x1816110Du, 0x1816120Eu, 0x1816130Fu, 0x18160600u, 0x18160701u, 0x18160802u, 0x18160903u, 0x18160A04u, 0x18160B05u, 0x1427141Au, 0x1427141Bu, 0x14271422u, 0x14141414u, 0x14141414u, 0x14141414u, 0x1918100Cu, 0x1918110Du, 0x1918120Eu, 0x1918130Fu, 0x19180600u, 0x19180701u, 0x19180802u, 0x19180903u, 0x19180A04u, 0x19180B05u, 0x1427141Cu, 0x1427141Du, 0x14271423u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x1427141Eu, 0x14271420u, 0x14271424u, 0x14271421u, 0x14271425u, 0x14271426u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u, 0x14141414u};
__constant__ unsigned char c_rs4_mfma_a[64] = {12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 20, 20, 20, 20, 20, 20, 16, 17, 18, 19, 6, 7, 8, 9, 10, 11, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 39, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20};

// fp64 division x / z as the compiler lowers it (v_rcp_f64, two Newton steps, quotient, remainder, one correction), WITHOUT the
// v_div_scale / v_div_fixup wrapping that only acts on operands at the ends of the exponent range or on non-finite ones: for every
// finite operand pair in the normal range the bits are those of the IEEE quotient; a zero, infinite or NaN denominator yields a
// non-finite result here as there (inf may become NaN: every consumer below only asks whether the value is inside the image).
// Splitting it lets the two projections of a pixel (x/z, y/z) share the reciprocal.
__device__ __forceinline__ double rs4_rcp_refined(const double z) {
    double r = __builtin_amdgcn_rcp(z);
    double e = __builtin_fma(-z, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-z, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
// CMLHIP_ARITH_RELAXED (cmlhip_ba_set_arithmetic; see ba_linearize_rs_body.inc): v_rcp_f64 + one Newton step
__device__ __forceinline__ double rs4_rcp_once(const double z) {
    const double r = __builtin_amdgcn_rcp(z);
    return __builtin_fma(r, __builtin_fma(-z, r, 1.0), r);
}
__device__ __forceinline__ double rs4_div(const double x, const double z, const double r) {
    const double q = x * r;
    const double rem = __builtin_fma(-z, q, x);
    return __builtin_fma(rem, r, q);
}

#define RS_RES 16            // residuals per wave
#define RS_DSTRIDE 90        // doubles per residual in the fp64 operand rows (10 rows x 9): 180 dwords = 52 mod 64, 8 residuals on distinct bank pairs
#define RS_FSTRIDE 25        // floats per residual in the fp32 operand rows (3 rows x 8)
#define RS_SSTRIDE 45        // floats per residual of the staged reduced record (odd: conflict-free lane-per-record reads)

// one entry of Jpdc (BA.cpp:150-176), k = 0..3 -> Jpdc[0][k] (LO), k = 4..7 -> Jpdc[1][k-4]; same expression shape as k_ba_linearize
template <bool LO>
__device__ __forceinline__ float rs4_jpdc(const int k, const double E0, const double E1, const double E3, const double E4, const double E6,
                                         const double E7, const float u, const float v, const float fxf, const float fyf, const float drescale,
                                         const double rx, const double ry, const double scale_f, const double scale_c, const double rfx, const double rfy) {
    const bool odd = k & 1;
    const double Ea = odd ? E7 : E6, Eb = LO ? (odd ? E1 : E0) : (odd ? E4 : E3);
    const float wq = LO ? u : v;
    const float sfac = LO ? (odd ? fxf : 1.f) : (odd ? 1.f : fyf), s2 = LO ? (odd ? fyf : 1.f) : (odd ? 1.f : fxf);
    const double rs2 = LO ? (odd ? rfy : 1.0) : (odd ? 1.0 : rfx);             // refined reciprocal of s2 (exactly 1 for s2 = 1: the division is then exact)
    const double q = rs4_div((sfac * drescale) * (Ea * wq - Eb), (double)s2, rs2);
    const double m = (k & 2) ? 1.0 : (odd ? ry : rx);
    const double add = k == 0 ? (double)u : (k == 5 ? (double)v : ((k == 2 || k == 7) ? 1.0 : -0.0));
    const double scl = (k & 2) ? scale_c : scale_f;
    return (float)(((m * q) + add) * scl);
}

// MINB: workgroups per CU the register budget is sized for — 1: no register bound (196 VGPRs, two waves per SIMD), no scratch frame;
// 3: 168 VGPRs, the 13-23 registers beyond that spill to a scratch frame — measured slower at every window size (see the launcher)
template <bool HALF, int MINB, bool RELAX = false>
__device__ __forceinline__ void k_ba_lin_rs4_body(const BAArgs& A, const RsArgs& X, const int ti, const int4 T, const bool dead = false) {
    __shared__ double s_shd[4][RS_RES * RS_DSTRIDE];                                   // [wave][residual * RS_DSTRIDE + quantity * 9 + pixel]
    __shared__ float s_shf[4][RS_RES * RS_FSTRIDE];                                    // [wave][residual * RS_FSTRIDE + quantity * 8 + pixel]
    // the staged reduced record (layout of k_ba_acc's s_rec + Jpdd at 40,41) reuses the wave's fp64 rows once the sums are taken (a wave's
    // LDS operations execute in order): 52.5 KB per workgroup, three workgroups per CU
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63, g = ln >> 2, j = ln & 3;
    const int stop_lin = *X.stop_lin;                      // converged in an earlier launch (raised by k_ba_acc), BA.cpp:879: requested with the
                                                           // pair record and the inputs, tested behind them (a test up here is a trip of its own)
    // ---- wave-uniform data: tile {first residual, count, host, target} (from the caller) -> pair record, frames (scalar loads)
    const int first = T.x, cnt = T.y, host = T.z, target = T.w;
    const cmlhip_ba_pair* pc = &A.pairs[host * A.N + target];
    const FrameDev fh = A.frames[host], ft = A.frames[target];
    const double R0_ = pc->R[0], R1_ = pc->R[1], R2_ = pc->R[2], R3_ = pc->R[3], R4_ = pc->R[4], R5_ = pc->R[5],
                 R6_ = pc->R[6], R7_ = pc->R[7], R8_ = pc->R[8];
    const double t0_ = pc->t[0], t1_ = pc->t[1], t2_ = pc->t[2];
    const double aff_a = pc->aff_a, aff_b = pc->aff_b;
    // evaluation-point pair (PRE_RTll_0 / PRE_tTll_0) for the calibration / depth Jacobians: scalar loads with the rest, before any store
    const double E0 = pc->R0[0], E1 = pc->R0[1], E3 = pc->R0[3], E4 = pc->R0[4], E6 = pc->R0[6], E7 = pc->R0[7];
    const double et0 = pc->t0[0], et1 = pc->t0[1], et2 = pc->t0[2];

    // ---- per-residual inputs, all addressed by the residual index (the 4 lanes of a quad read the same addresses: broadcast)
    const bool valid = g < cnt;
    const int r = first + (valid ? g : 0);
    const int lin_ = A.r_lin[r], st_ = A.r_state[r];
    const float pre_energy = A.r_energy[r];
    const float pre_new_energy = A.r_new_energy[r];        // (read with the inputs: behind the stores of the classification it would wait for every one of them)
    const int pre_new_state = A.r_new_state[r];
    const double cxd = (double)X.r_px[r], cyd = (double)X.r_py[r];
    const float2 col2 = reinterpret_cast<const float2*>(X.r_colors)[4 * (size_t)r + j];     // colours / weights of pattern pixels 2j, 2j+1
    const float2 wgt2 = reinterpret_cast<const float2*>(X.r_weights)[4 * (size_t)r + j];
    // the lane's matrix-core operand offsets (c_rs4_mfma_*): requested HERE with the inputs — left to the compiler the two table loads
    // sink to the matrix-core loop at the end of the kernel, an exposed memory round trip
    unsigned mf_off = c_rs4_mfma_off[ln];
    int mf_a = c_rs4_mfma_a[ln];
    const double idepth = X.r_idepth[r];                   // == pt_idepth[r_point[r]] (the launcher refreshes the copies when needed)
    const bool live = valid && !lin_;
    const int st = live ? st_ : CMLHIP_RES_OOB;
    const bool run = live && st != CMLHIP_RES_OOB;
    // (pins the table loads, the stop word and the target's image pointer to the input round trip: left to the compiler the pointer is
    //  requested in the middle of the projections and waited for ahead of the first texel load)
    int stop_pin = stop_lin;
    unsigned long long g0 = (unsigned long long)ft.grad0;
    unsigned g0lo = __builtin_amdgcn_readfirstlane((unsigned)g0), g0hi = __builtin_amdgcn_readfirstlane((unsigned)(g0 >> 32));
    asm volatile("" : "+v"(mf_off), "+v"(mf_a), "+v"(stop_pin), "+s"(g0lo), "+s"(g0hi));
    const void* const grad0 = (const void*)(((unsigned long long)g0hi << 32) | g0lo);
    __builtin_amdgcn_sched_barrier(0);                     // (an asm statement alone is moved down past the projections by the scheduler)
#ifdef CML_RS_STAMPS                                       // development build (CML_HIPCC_EXTRA=-DCML_RS_STAMPS): per-tile phase stamps, tools/probe_rs_tiles.py
    long long* const ts = (A.dbg && ti < CML_DEBUG_RS_TILES) ? A.dbg + CMLHIP_DEBUG_SLOTS + 8 * (size_t)ti : nullptr;
#define RS4_STAMP(i) do { if (ts && ln == 0) ts[i] = wall_clock64(); } while (0)
    if (ts && ln == 0) { ts[0] = wall_clock64(); ts[7] = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(63508) << 32); }
    { double dep = idepth + cxd + (double)st + (double)col2.x + (double)wgt2.x; asm volatile("" : "+v"(dep)); RS4_STAMP(1); }
#else
#define RS4_STAMP(i) do { } while (0)
#endif

    // ---- the lane's two pattern pixels, BA.cpp:193-212 (star8 offsets + 2 packed by nibble, types.h:1381-1393)
    double qx[2], qy[2], ppx[2], ppy[2], ppz[2], kx[2], ky[2], rz[2];
    bool pix_in[2];
    const double rc0 = R2_ + t0_ * idepth, rc1 = R5_ + t1_ * idepth, rc2 = R8_ + t2_ * idepth;      // RELAX only
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int pk = 2 * j + u;
        const int ox = (int)((0x21420312u >> (4 * pk)) & 15u) - 2, oy = (int)((0x43222110u >> (4 * pk)) & 15u) - 2;
        const double sx = cxd + ox, sy = cyd + oy;
        qx[u] = (sx - A.cx) * A.fxi; qy[u] = (sy - A.cy) * A.fyi;
        if (RELAX) {                                           // fused multiply-adds, one Newton step, a plain product per quotient (still fp64)
            ppx[u] = __builtin_fma(R0_, qx[u], __builtin_fma(R1_, qy[u], rc0)); ppy[u] = __builtin_fma(R3_, qx[u], __builtin_fma(R4_, qy[u], rc1));
            ppz[u] = __builtin_fma(R6_, qx[u], __builtin_fma(R7_, qy[u], rc2));
            rz[u] = rs4_rcp_once(ppz[u]);
            kx[u] = __builtin_fma(ppx[u] * rz[u], A.fx, A.cx); ky[u] = __builtin_fma(ppy[u] * rz[u], A.fy, A.cy);
        } else {
            ppx[u] = (R0_ * qx[u] + R1_ * qy[u] + R2_ * 1.0) + t0_ * idepth;
            ppy[u] = (R3_ * qx[u] + R4_ * qy[u] + R5_ * 1.0) + t1_ * idepth;
            ppz[u] = (R6_ * qx[u] + R7_ * qy[u] + R8_ * 1.0) + t2_ * idepth;
            rz[u] = rs4_rcp_refined(ppz[u]);
            kx[u] = rs4_div(ppx[u], ppz[u], rz[u]) * A.fx + A.cx; ky[u] = rs4_div(ppy[u], ppz[u], rz[u]) * A.fy + A.cy;
        }
        pix_in[u] = (kx[u] >= 2 && ky[u] >= 2 && kx[u] < A.w - 2 && ky[u] < A.h - 2);
    }
    // ---- centre projection, BA.cpp:102-131: pattern pixel 4 is the offset (0,0) = first pixel of quad lane 2: the very same
    //      expressions on the very same operands, taken from there
    const double rx = rs4_quad_bcast_d<2>(qx[0]), ry = rs4_quad_bcast_d<2>(qy[0]);
    const double px = rs4_quad_bcast_d<2>(ppx[0]), py = rs4_quad_bcast_d<2>(ppy[0]), pz = rs4_quad_bcast_d<2>(ppz[0]);
    const double Kud = rs4_quad_bcast_d<2>(kx[0]), Kvd = rs4_quad_bcast_d<2>(ky[0]);
    const float drescale = rs4_quad_bcast_f<2>(RELAX ? (float)rz[0] : (float)rs4_div(1.0, ppz[0], rz[0]));     // (float)(1.0 / pz) of the centre pixel
    const bool centre_in = (Kud >= 2 && Kvd >= 2 && Kud < A.w - 2 && Kvd < A.h - 2);

    // ---- GradientImage::interpolate (Array2D.h:265-286) of both pixels: eight unconditional loads on clamped addresses
    bool sample[2];
    float tw00[2], tw01[2], tw10[2], tw11[2];
    float4 ta[2], tb[2], tc[2], td[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        sample[u] = run && centre_in && pix_in[u];
        const float x = (float)kx[u], y = (float)ky[u];
        const int ix = (int)x, iy = (int)y;
        const float dx = x - (float)ix, dy = y - (float)iy;
        const float dxdy = dx * dy;
        tw00[u] = 1 - dx - dy + dxdy; tw01[u] = dx - dxdy; tw10[u] = dy - dxdy; tw11[u] = dxdy;
        const size_t i1 = (sample[u] && !(X.dbg_flags & 1)) ? (size_t)iy * A.w + ix : (size_t)0;
        ta[u] = rs4_load_texel<HALF>(grad0, i1); tb[u] = rs4_load_texel<HALF>(grad0, i1 + 1);
        tc[u] = rs4_load_texel<HALF>(grad0, i1 + A.w); td[u] = rs4_load_texel<HALF>(grad0, i1 + A.w + 1);
    }
    // ---- geometric Jacobians, BA.cpp:120-188, evaluated HERE — behind the issue of the texel loads, ahead of their first use: they
    //      depend on the projection only, and at small windows a wave is alone on its SIMD, so whatever runs under the texel round
    //      trip is free (the values wait in eight registers for the staging below).  Every lane evaluates two entries of each group
    //      (k = j and k = j + 4), same expression shapes as k_ba_linearize.
    RS4_STAMP(2);
    __builtin_amdgcn_sched_barrier(0);
    const float new_idepth = (float)(drescale * idepth);
    const float u = (float)px, v = (float)py;            // BA.cpp:121-122: un-normalised x,y, literal
    const float fxf = (float)A.fx, fyf = (float)A.fy;
    const double rfx = rs4_rcp_refined((double)fxf), rfy = rs4_rcp_refined((double)fyf);      // wave-uniform
    // Jpdxi[0][k], Jpdxi[1][k], k = j (0..3) and k = j + 4 (4, 5 for j < 2)          (fp32, BA.cpp:133-147)
    const float xa0 = rs4_sel4(j, new_idepth * fxf, 0.f, -new_idepth * u * fxf, -u * v * fxf);
    const float xa1 = rs4_sel4(j, 0.f, new_idepth * fyf, -new_idepth * v * fyf, -(1 + v * v) * fyf);
    const float xb0 = rs4_sel4(j, (1 + u * u) * fxf, -v * fxf, 0.f, 0.f);
    const float xb1 = rs4_sel4(j, u * v * fyf, u * fyf, 0.f, 0.f);
    // Jpdc[0][j], Jpdc[1][j]                                                         (:150-176)
    const float c0 = rs4_jpdc<true>(j, E0, E1, E3, E4, E6, E7, u, v, fxf, fyf, drescale, rx, ry, A.scale_f, A.scale_c, rfx, rfy);
    const float c1 = rs4_jpdc<false>(j + 4, E0, E1, E3, E4, E6, E7, u, v, fxf, fyf, drescale, rx, ry, A.scale_f, A.scale_c, rfx, rfy);
    // Jpdd[j], j < 2                                                                 (:178-182)
    const bool odd = j & 1;
    double dd = drescale * ((odd ? et1 : et0) - et2 * (odd ? v : u)) * (odd ? fyf : fxf);
    // (pinned: the values are first used in the staging, beyond the exit branch below, and the compiler's code sinking moves the whole
    //  evaluation down there — 130 instructions on the wave's critical path instead of under the texel round trip; the scheduling
    //  barriers alone do not hold it, they act after the sinking)
    float xa0p = xa0, xa1p = xa1, xb0p = xb0, xb1p = xb1, c0p = c0, c1p = c1;
#ifndef RS4_NO_GEOM_PIN
    asm volatile("" : "+v"(xa0p), "+v"(xa1p), "+v"(xb0p), "+v"(xb1p), "+v"(c0p), "+v"(c1p), "+v"(dd));
#endif
    __builtin_amdgcn_sched_barrier(0);
    RS4_STAMP(3);
    float I[2], gx[2], gy[2];
    bool finite[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const float Iv = ta[u].x * tw00[u] + tb[u].x * tw01[u] + tc[u].x * tw10[u] + td[u].x * tw11[u];
        const float gxv = ta[u].y * tw00[u] + tb[u].y * tw01[u] + tc[u].y * tw10[u] + td[u].y * tw11[u];
        const float gyv = ta[u].z * tw00[u] + tb[u].z * tw01[u] + tc[u].z * tw10[u] + td[u].z * tw11[u];
        I[u] = sample[u] ? Iv : 0.f; gx[u] = sample[u] ? gxv : 0.f; gy[u] = sample[u] ? gyv : 0.f;
        finite[u] = isfinite(I[u]) && isfinite(gx[u]) && isfinite(gy[u]);
    }

    // (wave-uniform; nothing has been stored yet.  Tested HERE, behind the first use of the texels: wherever the branch stands the compiler
    //  keeps the loads of whatever is first used beyond it on its far side — ahead of the projections that was the pair record, ahead
    //  of the interpolation the image pointer and the texels themselves: a dependent trip more each.  A wave that leaves has requested
    //  texel 0 only: none of its lanes is `live`.)
    if (__builtin_amdgcn_readfirstlane(stop_pin) || dead) return;
    // first failing pixel in pattern order decides between setNewState(OOB) (:209-212) and setState(OOB) (:220-223)
    const int shift = ln & ~3;
    unsigned m_oob, m_nf;
    {
        const unsigned o0 = (unsigned)(__ballot(!pix_in[0]) >> shift) & 0xFu, o1 = (unsigned)(__ballot(!pix_in[1]) >> shift) & 0xFu;
        const unsigned n0 = (unsigned)(__ballot(pix_in[0] && !finite[0]) >> shift) & 0xFu, n1 = (unsigned)(__ballot(pix_in[1] && !finite[1]) >> shift) & 0xFu;
        // bit (2 * lane + u) of the 8-bit pattern mask
#define RS_SPREAD(x) (((x) & 1u) | (((x) & 2u) << 1) | (((x) & 4u) << 2) | (((x) & 8u) << 3))
        m_oob = RS_SPREAD(o0) | (RS_SPREAD(o1) << 1);
        m_nf = RS_SPREAD(n0) | (RS_SPREAD(n1) << 1);
#undef RS_SPREAD
    }
    const unsigned m_bad = m_oob | m_nf;
    const int first_bad = m_bad ? __ffs((int)m_bad) - 1 : 8;
    const bool fail_new_oob = !centre_in || (m_bad && ((m_oob >> first_bad) & 1u));
    const bool fail_state_oob = centre_in && m_bad && !((m_oob >> first_bad) & 1u);

    // ---- photometric terms of the two pixels, BA.cpp:214-255; every operand of the pattern sums is converted once by its lane
    double* D = &s_shd[wv][g * RS_DSTRIDE];
    float* F = &s_shf[wv][g * RS_FSTRIDE];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int pk = 2 * j + u;
        const float refColor = u == 0 ? col2.x : col2.y;
        float hw, drdA;
        if (RELAX) {                                           // photometric terms in fp32 (v_rcp_f32 / v_sqrt_f32), operand rows as floats; the colour term stays fp64: I and it cancel
            const float huber_f = (float)A.huber_d, oth_f = (float)A.oth_d;
            const float residual = I[u] - (float)__builtin_fma(aff_a, (double)refColor, aff_b);
            const float ares = fabsf(residual);
            hw = ares < huber_f ? 1.0f : huber_f * __builtin_amdgcn_rcpf(ares);
            float wgt = __builtin_amdgcn_sqrtf(oth_f * __builtin_amdgcn_rcpf(oth_f + __builtin_fmaf(gx[u], gx[u], gy[u] * gy[u])));
            wgt = 0.5f * (wgt + (u == 0 ? wgt2.x : wgt2.y));
            const float pf = wgt * wgt * hw * residual * residual;
            const float hw0 = hw;
            if (hw < 1) hw = __builtin_amdgcn_sqrtf(hw);
            hw = hw * wgt;
            const float f1 = gx[u] * hw, f2 = gy[u] * hw;
            drdA = I[u] - fh.b0;
            float* DF = reinterpret_cast<float*>(D);
            DF[0 * 9 + pk] = f1; DF[1 * 9 + pk] = f2; DF[2 * 9 + pk] = drdA * hw; DF[3 * 9 + pk] = hw; DF[4 * 9 + pk] = residual * hw;
            DF[5 * 9 + pk] = pf; DF[6 * 9 + pk] = 2.0f - hw0;
            DF[7 * 9 + pk] = hw * hw; DF[8 * 9 + pk] = __builtin_fmaf(f2, f2, f1 * f1);
            DF[9 * 9 + pk] = 0.f;
        } else {
            const float refRealColor = (float)(aff_a * (double)refColor + aff_b);
            const float residual = I[u] - refRealColor;
            hw = fabs((double)residual) < A.huber_d ? 1.0f : (float)(A.huber_d / (double)fabsf(residual));
            const double wden = A.oth_d + (double)(gx[u] * gx[u] + gy[u] * gy[u]);
            float wgt = sqrtf((float)rs4_div(A.oth_d, wden, rs4_rcp_refined(wden)));
            wgt = (float)(0.5f * ((double)wgt + (double)(u == 0 ? wgt2.x : wgt2.y)));
            const float pf = wgt * wgt * hw * residual * residual;      // energy term factor, :237
            const float hw0 = hw;
            if (hw < 1) hw = sqrtf(hw);
            hw = hw * wgt;
            const float f1 = gx[u] * hw, f2 = gy[u] * hw;               // hitColor[1], hitColor[2]
            drdA = I[u] - fh.b0;
            const float a_ = drdA * hw;
            const float rF = residual * hw;
            const double f1d = (double)f1, f2d = (double)f2;
            D[0 * 9 + pk] = f1d; D[1 * 9 + pk] = f2d; D[2 * 9 + pk] = (double)a_; D[3 * 9 + pk] = (double)hw; D[4 * 9 + pk] = (double)rF;
            D[5 * 9 + pk] = (double)pf; D[6 * 9 + pk] = 2.0 - (double)hw0;                          // energy term, BA.cpp:237
            D[7 * 9 + pk] = (double)(hw * hw); D[8 * 9 + pk] = f1d * f1d + f2d * f2d;              // wJI2_sum, BA.cpp:257
            D[9 * 9 + pk] = 0.0;
        }
        F[pk] = drdA; F[8 + pk] = hw; F[16 + pk] = 1.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // rows are wave-private and a wave's LDS operations execute in order:
    __builtin_amdgcn_wave_barrier();                        // only the compiler has to be kept from moving the reads up

    // ---- pattern-order sums, BA.cpp:237,257-271 and the ACTIVE-mode inner products of BA.cpp:1719-1729; three arithmetic forms
    //   A  acc = (float)((double)acc + X*Y)   slot 0: lane j: J00 J10 J11 Q00   slot 1: Q10 Q01 Q11 r^T r   slot 2: energy, wJI2_sum, -, -
    //   B  acc += rF*Y (fp64)                 lane j: JI^T r (2), Jab^T r (2)   (a masked column reads the row of zeros: BA.cpp:273-278)
    //   C  acc += ((p*q)*r)*s (fp32)          lane j: B00 B01 B11 -
    // fp64 rows: 0 F1, 1 F2, 2 a, 3 hw, 4 rF, 5 pf, 6 2-hw0, 7 hw*hw, 8 F1^2+F2^2, 9 zeros.   fp32 rows: 0 drdA, 1 hw, 2 ones
    const int ax0 = (0x2100 >> (4 * j)) & 15, ay0 = (0x0110 >> (4 * j)) & 15;
    const int ax1 = (0x4323 >> (4 * j)) & 15, ay1 = (0x4110 >> (4 * j)) & 15;
    const int ax2 = (0x9975 >> (4 * j)) & 15, ay2 = (0x9986 >> (4 * j)) & 15;
    const int by = ((j == 2 && !A.opt_a) || (j == 3 && !A.opt_b)) ? 9 : j;
    const int cp = j < 2 ? 0 : (j == 2 ? 1 : 2), cq = j == 0 ? 0 : (j < 3 ? 1 : 2);
    const int cr = j < 2 ? 1 : 2, cs = j == 0 ? 1 : 2;
    float sumA0 = 0, sumA1 = 0, sumA2 = 0, sumC = 0, sumBf = 0;
    double sumB = 0;
    const float* DFr = reinterpret_cast<const float*>(D);
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        if (RELAX) {                                           // fp32 fused multiply-adds where the exact mode widens, adds in fp64 and rounds back per term
            sumA0 = __builtin_fmaf(DFr[ax0 * 9 + jj], DFr[ay0 * 9 + jj], sumA0);
            sumA1 = __builtin_fmaf(DFr[ax1 * 9 + jj], DFr[ay1 * 9 + jj], sumA1);
            sumA2 = __builtin_fmaf(DFr[ax2 * 9 + jj], DFr[ay2 * 9 + jj], sumA2);
            sumBf = __builtin_fmaf(DFr[4 * 9 + jj], DFr[by * 9 + jj], sumBf);
        } else {
            sumA0 = (float)((double)sumA0 + D[ax0 * 9 + jj] * D[ay0 * 9 + jj]);
            sumA1 = (float)((double)sumA1 + D[ax1 * 9 + jj] * D[ay1 * 9 + jj]);
            sumA2 = (float)((double)sumA2 + D[ax2 * 9 + jj] * D[ay2 * 9 + jj]);
            sumB += D[4 * 9 + jj] * D[by * 9 + jj];
        }
        sumC += F[cp * 8 + jj] * F[cq * 8 + jj] * F[cr * 8 + jj] * F[cs * 8 + jj];
    }
    RS4_STAMP(4);
    const float E = rs4_quad_bcast_f<0>(sumA2);                // quad lane 0: the energy; lane 1: wJI2_sum
    const float wJI2 = rs4_quad_bcast_f<1>(sumA2);

    // ---- classification, BA.cpp:66-72,115-118,297-314, and the fused applyRes(copyJacobians = true), BA.cpp:2051-2093 — quad lane 0
    double ret_d = 0.0;
    int ns_cnt = -1, flip = 0;
    if (live && j == 0) {
        float ret = pre_energy;
        float nwo = -1.f;
        int ns_final = pre_new_state;
        bool state_now_oob = (st == CMLHIP_RES_OOB), wrote_e = false;
        if (run) {
            if (centre_in && !(X.dbg_flags & RS_LEAN_BIT)) {                             // setCenterProjectedTo, :131 (lean outputs: nothing on the product path reads it)
                A.r_center[3 * (size_t)r] = (float)Kud; A.r_center[3 * (size_t)r + 1] = (float)Kvd;
                A.r_center[3 * (size_t)r + 2] = new_idepth;
            }
            if (fail_new_oob) {
                ns_final = CMLHIP_RES_OOB;
            } else if (fail_state_oob) {
                A.r_state[r] = CMLHIP_RES_OOB;
                state_now_oob = true;
            } else if (!isfinite(E)) {
                ns_final = CMLHIP_RES_OOB;
            } else {
                nwo = E;
                const float th = fh.frame_energy_th > ft.frame_energy_th ? fh.frame_energy_th : ft.frame_energy_th;
                float e = E;
                ns_final = CMLHIP_RES_IN;
                if (E > th || wJI2 < 2) { e = th; ns_final = CMLHIP_RES_OUTLIER; }
                A.r_new_energy[r] = e;
                ret = e;
                wrote_e = true;
            }
            A.r_new_state[r] = ns_final;
        }
        if (!(X.dbg_flags & RS_LEAN_BIT)) { A.r_new_energy_wo[r] = nwo; A.r_ret_energy[r] = ret; }     // lean outputs: NewEnergyWithOutlier only where setNewFrameEnergyTH reads it
        else if (T.w == A.N - 1) A.r_new_energy_wo[r] = nwo;
        ret_d = (double)ret; ns_cnt = ns_final;
        if (!state_now_oob) {                                       // applyRes
            if (ns_final == CMLHIP_RES_IN) { A.r_good[r] = 1; flip = 1; }
            else A.r_good[r] = 0;
            A.r_state[r] = ns_final;
            A.r_energy[r] = wrote_e ? ret : pre_new_energy;      // state_energy = state_NewEnergy
            // (no per-slot code any more: the point rows of k_ba_acc and k_ba_backsub read r_good through the static slot table)
        }
    }
    flip = rs4_quad_bcast_i<0>(flip);
    RS4_STAMP(5);

    // ---- staging of the geometric Jacobians (evaluated above, under the texel round trip) and of the sums; a residual that is not IN
    //      stages zeros (the matrix-core loop below is branch-free)
    float* const stg = reinterpret_cast<float*>(&s_shd[wv][0]);
    float* S = &stg[g * RS_SSTRIDE];
    {
        // Every lane issues the SAME stores with per-lane addresses (a lane with nothing to contribute to a group writes a spare slot,
        // 42..44): a lane-divergent `if` around an LDS access costs a branch each.
        const bool lo2 = j < 2;
        S[j] = flip ? xa0p : 0.f; S[6 + j] = flip ? xa1p : 0.f;
        S[lo2 ? 4 + j : 42] = flip ? xb0p : 0.f; S[lo2 ? 10 + j : 43] = flip ? xb1p : 0.f;
        S[12 + j] = flip ? c0p : 0.f; S[16 + j] = flip ? c1p : 0.f;
        S[lo2 ? 40 + j : 44] = flip ? (float)dd : 0.f;
        // the sums of this lane (staged layout of k_ba_acc: 22..25 JIdx2, 26..29 JabJIdx, 30..33 Jab2, 34,35 JI^T r, 36,37 Jab^T r, 38 r^T r)
        //   lane 0: J00 -> 22, Q10 -> 27, B00 -> 30     lane 1: J10 -> 23, 24, Q01 -> 28, B01 -> 31, 32
        //   lane 2: J11 -> 25, Q11 -> 29, B11 -> 33     lane 3: Q00 -> 26, r^T r -> 38
        const float a0 = flip ? sumA0 : 0.f, a1 = flip ? sumA1 : 0.f, sb = flip ? (RELAX ? sumBf : (float)sumB) : 0.f, sc = flip ? sumC : 0.f;
        const int oa0 = (0x1A191716 >> (8 * j)) & 255, oa1 = (0x261D1C1B >> (8 * j)) & 255, osc = (0x2A211F1E >> (8 * j)) & 255;
        S[oa0] = a0; S[j == 1 ? 24 : 43] = a0;
        S[oa1] = a1;
        S[osc] = sc; S[j == 1 ? 32 : 44] = sc;
        S[34 + j] = sb;
        S[(0x2B271514 >> (8 * j)) & 255] = j == 2 ? 1.f : 0.f;          // slots of zeros (20, 21) and of ones (39) for the matrix-core operands
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    // ---- per residual: JpJdF (BA.cpp:2066-2080) and the terms of Hcd, Hdd, bd (BA.cpp:1747-1750): 4 floats per quad lane, each
    //      P*s + Q*t with per-lane LDS offsets (all loads unconditional):
    //        lane 0: JpJdF[0..3] = Jpdxi[0][i] g0 + Jpdxi[1][i] g1        lane 1: JpJdF[4,5] likewise, JpJdF[6,7] = JabJIdx(.,0) d0 + JabJIdx(.,1) d1
    //        lane 2: Hcd[i] = Jpdc[0][i] g0 + Jpdc[1][i] g1               lane 3: Hdd = d0 g0 + d1 g1, bd = JI^T r . Jpdd (fp64 as BA.cpp:1748), 0, 0
    {
        const float d0 = S[40], d1 = S[41];
        const float g0 = S[22] * d0 + S[24] * d1;
        const float g1 = S[23] * d0 + S[25] * d1;
        const int pa = (0x280C0400 >> (8 * j)) & 255, pb = (0x29100A06 >> (8 * j)) & 255;          // {0, 4, 12, 40}, {6, 10, 16, 41}
        const int pa2 = j == 1 ? 26 : pa + 2, pb2 = j == 1 ? 28 : pb + 2;             // lane 1: JabJIdx (0,0),(1,0) | (0,1),(1,1)
        const float P0 = S[pa], P1 = S[pa + 1], P2 = S[pa2], P3 = S[pa2 + 1];
        const float Q0 = S[pb], Q1 = S[pb + 1], Q2 = S[pb2], Q3 = S[pb2 + 1];
        const float s2 = j == 1 ? d0 : g0, t2 = j == 1 ? d1 : g1;
        const float bd = RELAX ? __builtin_fmaf(S[34], d0, S[35] * d1) : (float)((double)S[34] * (double)d0 + (double)S[35] * (double)d1);
        float4 o;
        o.x = P0 * g0 + Q0 * g1;
        o.y = P1 * g0 + Q1 * g1;
        o.z = P2 * s2 + Q2 * t2;
        o.w = P3 * s2 + Q3 * t2;
        if (j == 3) { o.y = bd; o.z = 0.f; o.w = 0.f; }
        if (flip && !(X.dbg_flags & 2)) reinterpret_cast<float4*>(A.r_jpjdf + PS_STRIDE * (size_t)r)[j] = o;
    }

    // ---- the wave's contribution to the 13x13 block of its pair: one v_mfma_f32_16x16x4_f32 per residual (see acc_pair_block)
    {
        const int oa = mf_a, o1 = mf_off & 255, o2 = (mf_off >> 8) & 255, o3 = (mf_off >> 16) & 255, o4 = mf_off >> 24;
        float4_ acc = {0.f, 0.f, 0.f, 0.f};
        const float* SW = stg;
        // eight residuals per trip: all forty operand reads are issued before the first product (one LDS latency per trip, not one per row)
#pragma unroll
        for (int l0 = 0; l0 < RS_RES; l0 += 8) {
            float av[8], b1[8], b2[8], b3[8], b4[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float* SL = SW + (l0 + u) * RS_SSTRIDE;
                av[u] = SL[oa]; b1[u] = SL[o1]; b2[u] = SL[o2]; b3[u] = SL[o3]; b4[u] = SL[o4];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b3[u] * b1[u] + b4[u] * b2[u], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // a residual slot beyond the tile's count (or not IN) staged zeros: the ones slot then multiplies zero fields only
        if (!(X.dbg_flags & 2)) reinterpret_cast<float4*>(X.part)[(size_t)ti * 64 + ln] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }

    RS4_STAMP(6);
    // ---- per-tile partials {energy, n_in, n_oob, n_outlier} (BA.cpp:1565): fixed butterfly order over the 16 residuals
    if (A.lin_partial) {
        double e = ret_d;                                            // non-zero in quad lane 0 only
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
        const int c0 = __popcll(__ballot(ns_cnt == CMLHIP_RES_IN)), c1 = __popcll(__ballot(ns_cnt == CMLHIP_RES_OOB)), c2 = __popcll(__ballot(ns_cnt == CMLHIP_RES_OUTLIER));
        if (ln == 0) {
            double* o = A.lin_partial + 4 * (size_t)ti;
            o[0] = e; o[1] = (double)c0; o[2] = (double)c1; o[3] = (double)c2;
        }
    }
    // ---- resident loop: the convergence test of doStepFromBackup (BA.cpp:996-1027) on the sums of the step that preceded this
    //      pass; `if (canbreak && it >= 1) break` (BA.cpp:879) becomes a sticky flag that every later kernel checks first
    if (A.ctl && ti == 0 && ln == 0) {
        float sumID = 0, sumNID = 0, numID = 0;
        for (int b = 0; b < A.n_step_blocks; b++) { sumID += A.step_partial_ro[4 * b]; sumNID += A.step_partial_ro[4 * b + 1]; numID += A.step_partial_ro[4 * b + 2]; }
        float sumA = A.ctl->frame_sums[0], sumB_ = A.ctl->frame_sums[1], sumT = A.ctl->frame_sums[2], sumR = A.ctl->frame_sums[3];
        const float nf = (float)A.N;
        sumA /= nf; sumB_ /= nf; sumR /= nf; sumT /= nf; sumID /= numID; sumNID /= numID;
        const bool canbreak = sqrtf(sumA) < 0.0005 * A.th_opt && sqrtf(sumB_) < 0.00005 * A.th_opt && sqrtf(sumR) < 0.00005 * A.th_opt &&
                              sqrtf(sumT) * sumNID < 0.00005 * A.th_opt;
        A.ctl->iters_done = A.it_index + 1;
        if (canbreak && A.it_index >= 1) A.ctl->stop = 1;
    }
}
template <bool HALF, int MINB, bool RELAX = false>
__global__ __launch_bounds__(256, MINB) void k_ba_lin_rs4(BAArgs A, RsArgs X) {
    const int ti = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (ti >= X.ntiles) return;                            // wave-uniform; there is no workgroup barrier in the body
    k_ba_lin_rs4_body<HALF, MINB, RELAX>(A, X, ti, X.tiles[ti]);
}
// The same kernel launched over (tile group of a pair, pair): the pair's entry {first residual, residuals, first tile, host | target << 16}
// is indexed by blockIdx.y inside the KERNEL-ARGUMENT segment, so it arrives with the arguments — the solo kernel above reads its tile
// from a table in memory first, a dependent trip of its own ahead of the pair record and the inputs (4 trips: arguments, tile, pair +
// inputs, texels; here 3).  Waves beyond a pair's last tile return at once.  Windows with more pairs than the table holds take the
// 1-D launch.
#define RS4_PAIR_TAB 128
struct RsPairTab { int4 e[RS4_PAIR_TAB]; };
template <bool HALF, bool RELAX = false>
#ifndef RS4_2D_WPB
#define RS4_2D_WPB 4         // waves per workgroup of the 2-D launch
#endif
__global__ __launch_bounds__(64 * RS4_2D_WPB, 1) void k_ba_lin_rs4_2d(BAArgs A, RsArgs X, RsPairTab Q) {
    const int4 e = Q.e[blockIdx.y];
    const int lt = __builtin_amdgcn_readfirstlane(blockIdx.x * RS4_2D_WPB + (threadIdx.x >> 6));        // tile of the pair
    const int left = e.y - lt * RS_RES;
    // a wave beyond its pair's last tile leaves where the stop flag is tested — behind the loads of the input trip (clamped onto the
    // pair's first residuals), not up here: a branch ahead of them makes the entry, the control word and the arguments three waits
    const bool dead = left <= 0;
    k_ba_lin_rs4_body<HALF, 1, RELAX>(A, X, e.z + lt, make_int4(dead ? e.x : e.x + lt * RS_RES, left < RS_RES ? left : RS_RES, e.w & 0xffff, e.w >> 16), dead);
}


int cml_launch_linearize_rs4(cmlhip_ctx* c, const BAArgs& A, RsArgs X) {
    const int blocks = cml_div_up(c->n_tiles, 4);
    // Measured (tools/probe_rs_regime.sh, fp32 windows of 8 keyframes): the register-bounded instantiation loses at every size this
    // kernel is used for (R = 25 200: 15.2 against 11.2 us; R = 35 000: 18.4 against 15.3 us, k_ba_lin_rs 15.9) — it stays for the
    // development switch only.
    static const char* e_minb = getenv("CMLHIP_RS4_MINB");   // development: 3 forces the 168-VGPR instantiation
    const bool small = !(e_minb && atoi(e_minb) == 3);                      // at most two workgroups per CU: 196 VGPRs still leave two waves per SIMD
    const char* e_1d = getenv("CMLHIP_RS4_1D");             // development / tests: 1 = the 1-D launch (tile table in memory) whatever the window (read per launch)
    if (small && !(e_1d && atoi(e_1d)) && c->rs_pair_n > 0 && c->rs_pair_n <= RS4_PAIR_TAB) {
        RsPairTab Q;
        memset(&Q, 0, sizeof Q);
        memcpy(Q.e, c->h_rs_pair_tab.data(), 16 * (size_t)c->rs_pair_n);
        const dim3 grid(cml_div_up(c->rs_pair_max_tiles, RS4_2D_WPB), c->rs_pair_n);
        if (c->arith_relaxed) {
            if (c->lim.texel_format == CMLHIP_TEXEL_F16) CML_LAUNCH_EV(c, (k_ba_lin_rs4_2d<true, true>), grid, 64 * RS4_2D_WPB, 0, A, X, Q);
            else CML_LAUNCH_EV(c, (k_ba_lin_rs4_2d<false, true>), grid, 64 * RS4_2D_WPB, 0, A, X, Q);
        } else if (c->lim.texel_format == CMLHIP_TEXEL_F16) CML_LAUNCH_EV(c, (k_ba_lin_rs4_2d<true>), grid, 64 * RS4_2D_WPB, 0, A, X, Q);
        else CML_LAUNCH_EV(c, (k_ba_lin_rs4_2d<false>), grid, 64 * RS4_2D_WPB, 0, A, X, Q);
        return CMLHIP_OK;
    }
    if (c->arith_relaxed) {
        if (c->lim.texel_format == CMLHIP_TEXEL_F16) CML_LAUNCH_EV(c, (k_ba_lin_rs4<true, 1, true>), blocks, 256, 0, A, X);
        else CML_LAUNCH_EV(c, (k_ba_lin_rs4<false, 1, true>), blocks, 256, 0, A, X);
        return CMLHIP_OK;
    }
    if (c->lim.texel_format == CMLHIP_TEXEL_F16) {
        if (small) CML_LAUNCH_EV(c, (k_ba_lin_rs4<true, 1>), blocks, 256, 0, A, X);
        else CML_LAUNCH_EV(c, (k_ba_lin_rs4<true, 3>), blocks, 256, 0, A, X);
    } else {
        if (small) CML_LAUNCH_EV(c, (k_ba_lin_rs4<false, 1>), blocks, 256, 0, A, X);
        else CML_LAUNCH_EV(c, (k_ba_lin_rs4<false, 3>), blocks, 256, 0, A, X);
    }
    return CMLHIP_OK;
}

// several windows per launch (cmlhip_ba_iteration_batch): gridDim.y = window, the body and the arguments of the solo kernel
template <bool HALF, bool RELAX = false>
__global__ __launch_bounds__(256, 1) void k_ba_lin_rs4_batch(const BatchRs* __restrict__ W) {
    const BatchRs& w = *(const BatchRs*)(const BatchRs __attribute__((address_space(4)))*)(W + blockIdx.y);     // constant address space: scalar loads
    if ((int)blockIdx.x >= w.blocks) return;
    const int ti = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (ti >= w.X.ntiles) return;
    k_ba_lin_rs4_body<HALF, 1, RELAX>(w.A, w.X, ti, w.X.tiles[ti]);
}
int cml_launch_linearize_rs4_batch(cmlhip_ctx* c0, const void* dev_records, int S, int max_blocks) {
    const BatchRs* W = static_cast<const BatchRs*>(dev_records);
    if (c0->arith_relaxed) {                                 // (one arithmetic mode per batch: cml_iteration_batch checks)
        if (c0->lim.texel_format == CMLHIP_TEXEL_F16) k_ba_lin_rs4_batch<true, true><<<dim3(max_blocks, S), 256, 0, c0->stream>>>(W);
        else k_ba_lin_rs4_batch<false, true><<<dim3(max_blocks, S), 256, 0, c0->stream>>>(W);
    } else if (c0->lim.texel_format == CMLHIP_TEXEL_F16) k_ba_lin_rs4_batch<true><<<dim3(max_blocks, S), 256, 0, c0->stream>>>(W);
    else k_ba_lin_rs4_batch<false><<<dim3(max_blocks, S), 256, 0, c0->stream>>>(W);
    return CMLHIP_OK;
}
