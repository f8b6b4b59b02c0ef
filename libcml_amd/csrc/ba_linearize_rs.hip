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
//     weights are kept per residual, the copy of the inverse depth is refreshed by the point step of k_ba_backsub): after ONE round
//     trip for the inputs the 32 texel loads of a lane are issued together, unconditional on clamped addresses;
//   * nothing of the 74-float DSORawResidualJacobian is written to memory.  What the next iteration consumes leaves the kernel in
//     reduced form: the wave's contribution to the 13x13 AccumulatorApprox block of its pair (BA.cpp:1731-1745, ACC.h:776-932) as
//     ONE 16x16 fp32 tile accumulated on the matrix cores over the wave's residuals (v_mfma_f32_16x16x4_f32, one per residual,
//     the formulation of k_ba_acc; the only LDS use of the kernel is the transposition of the 40 operands per residual), and per
//     residual 14 floats: JpJdF (BA.cpp:2066-2080) and the terms of Hdd / bd / Hcd (BA.cpp:1747-1750).  The full records are
//     re-materialised on demand by k_ba_linearize (cml_materialize_records).
#include "cmlhip_internal.h"
#include "ba_common.h"
#include <cstdlib>
#include <cstddef>

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

#define RS_SSTRIDE 41        // floats per residual of the staged reduced record (odd: conflict-free lane-per-record accesses)

// star8 pattern offsets, types.h:1381-1393
#define RS_OX(k) ((k) == 0 ? 0 : (k) == 1 ? -1 : (k) == 2 ? 1 : (k) == 3 ? -2 : (k) == 4 ? 0 : (k) == 5 ? 2 : (k) == 6 ? -1 : 0)
#define RS_OY(k) ((k) == 0 ? -2 : (k) == 1 ? -1 : (k) == 2 ? -1 : (k) == 3 ? 0 : (k) == 4 ? 0 : (k) == 5 ? 0 : (k) == 6 ? 1 : 2)

template <bool HALF>
__global__ __launch_bounds__(64) void k_ba_lin_rs(BAArgs A, RsArgs X) {
    // staged reduced record of the wave's residuals: the layout of k_ba_acc's s_rec (0..5 Jpdxi[0], 6..11 Jpdxi[1], 12..15 Jpdc[0], 16..19 Jpdc[1],
    // 20 zeros, 22..25 JIdx2, 26..29 JabJIdx, 30..33 Jab2, 34,35 JI^T r, 36,37 Jab^T r, 38 r^T r, 39 ones)
    __shared__ float s_stg[RS_TILE * RS_SSTRIDE];
    const int ln = threadIdx.x;
    if (A.ctl && A.ctl->stop_lin) return;                  // converged in an earlier launch (raised by k_ba_acc), BA.cpp:879
    const int ti = blockIdx.x;
    // ---- wave-uniform data: tile -> pair record, frames (scalar loads, before any store)
    const int4 T = X.tiles[ti];                            // {first residual, count, host, target}
    const int first = T.x, cnt = T.y, host = T.z, target = T.w;
    const cmlhip_ba_pair* pc = &A.pairs[host * A.N + target];
    const FrameDev fh = A.frames[host], ft = A.frames[target];
    const double R0_ = pc->R[0], R1_ = pc->R[1], R2_ = pc->R[2], R3_ = pc->R[3], R4_ = pc->R[4], R5_ = pc->R[5],
                 R6_ = pc->R[6], R7_ = pc->R[7], R8_ = pc->R[8];
    const double t0_ = pc->t[0], t1_ = pc->t[1], t2_ = pc->t[2];
    const double aff_a = pc->aff_a, aff_b = pc->aff_b;

    // ---- per-residual inputs, all addressed by the residual index
    const bool valid = ln < cnt;
    const int r = first + (valid ? ln : 0);
    const int lin_ = A.r_lin[r], st_ = A.r_state[r];
    const float pre_energy = A.r_energy[r];
    const int pre_new_state = A.r_new_state[r], pre_ppos = A.point_pos[r];
    const unsigned char pre_sel = A.r_sel[r];
    const double cxd = (double)X.r_px[r], cyd = (double)X.r_py[r];
    const float4 colA = reinterpret_cast<const float4*>(X.r_colors)[2 * (size_t)r], colB = reinterpret_cast<const float4*>(X.r_colors)[2 * (size_t)r + 1];
    const float4 wgtA = reinterpret_cast<const float4*>(X.r_weights)[2 * (size_t)r], wgtB = reinterpret_cast<const float4*>(X.r_weights)[2 * (size_t)r + 1];
    const float colors[8] = {colA.x, colA.y, colA.z, colA.w, colB.x, colB.y, colB.z, colB.w};
    const float weights[8] = {wgtA.x, wgtA.y, wgtA.z, wgtA.w, wgtB.x, wgtB.y, wgtB.z, wgtB.w};
    const double idepth = X.r_idepth[r];                   // == pt_idepth[r_point[r]] (cml_launch_linearize_rs refreshes the copies when needed)
    const bool live = valid && !lin_;
    const int st = live ? st_ : CMLHIP_RES_OOB;
    const bool run = live && st != CMLHIP_RES_OOB;

    // ---- projection of the 8 pattern pixels, BA.cpp:193-212; the centre (BA.cpp:102-131) is pattern pixel 4, offset (0,0): the very
    //      same expressions on the very same operands
    const double tid0 = t0_ * idepth, tid1 = t1_ * idepth, tid2 = t2_ * idepth;
    float kxf[8], kyf[8];
    unsigned m_in = 0;
    double rx = 0, ry = 0, px = 0, py = 0, Kud = 0, Kvd = 0;
    float drescale = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const double sx = cxd + RS_OX(k), sy = cyd + RS_OY(k);
        const double qx = (sx - A.cx) * A.fxi, qy = (sy - A.cy) * A.fyi;
        const double ppx = (R0_ * qx + R1_ * qy + R2_ * 1.0) + tid0;
        const double ppy = (R3_ * qx + R4_ * qy + R5_ * 1.0) + tid1;
        const double ppz = (R6_ * qx + R7_ * qy + R8_ * 1.0) + tid2;
        const double rz = rs_rcp_refined(ppz);
        const double kx = rs_div(ppx, ppz, rz) * A.fx + A.cx, ky = rs_div(ppy, ppz, rz) * A.fy + A.cy;
        if (kx >= 2 && ky >= 2 && kx < A.w - 2 && ky < A.h - 2) m_in |= 1u << k;
        kxf[k] = (float)kx; kyf[k] = (float)ky;
        if (k == 4) {
            rx = qx; ry = qy; px = ppx; py = ppy; Kud = kx; Kvd = ky;
            drescale = (float)rs_div(1.0, ppz, rz);            // (float)(1.0 / pz)
        }
    }
    const bool centre_in = (m_in >> 4) & 1u;

    // ---- photometric terms and pattern sums, pixel by pixel in pattern order (BA.cpp:214-271 and the ACTIVE-mode inner products of
    //      BA.cpp:1719-1729).  Form A: acc = (float)((double)acc + X*Y); form B: acc += rF*Y in fp64 (a masked column adds rF * 0);
    //      form C: acc += ((p*q)*r)*s in fp32 — the forms and operand conversions of k_ba_linearize.
    float J00 = 0, J10 = 0, J11 = 0, Q00 = 0, Q10 = 0, Q01 = 0, Q11 = 0, rr = 0, E = 0, wJI2 = 0;
    double JIr0 = 0, JIr1 = 0, Jabr0 = 0, Jabr1 = 0;
    float B00 = 0, B01 = 0, B11 = 0;
    unsigned m_nf = 0;
    // GradientImage::interpolate (Array2D.h:265-286): the 32 texel loads of the lane are issued together, unconditional on clamped
    // addresses (ONE memory round trip for the whole pattern), then consumed in pattern order
    float4 ta[8], tb[8], tc[8], td[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const bool smp = run && centre_in && ((m_in >> k) & 1u);
        const int ix = (int)kxf[k], iy = (int)kyf[k];
        const size_t i1 = (smp && !(X.dbg_flags & 1)) ? (size_t)iy * A.w + ix : (size_t)0;
        ta[k] = rs_load_texel<HALF>(ft.grad0, i1); tb[k] = rs_load_texel<HALF>(ft.grad0, i1 + 1);
        tc[k] = rs_load_texel<HALF>(ft.grad0, i1 + A.w); td[k] = rs_load_texel<HALF>(ft.grad0, i1 + A.w + 1);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        {
            const bool smp = run && centre_in && ((m_in >> k) & 1u);
            const float x = kxf[k], y = kyf[k];
            const int ix = (int)x, iy = (int)y;
            const float dx = x - (float)ix, dy = y - (float)iy;
            const float dxdy = dx * dy;
            const float tw00 = 1 - dx - dy + dxdy, tw01 = dx - dxdy, tw10 = dy - dxdy, tw11 = dxdy;
            const float Iv = ta[k].x * tw00 + tb[k].x * tw01 + tc[k].x * tw10 + td[k].x * tw11;
            const float gxv = ta[k].y * tw00 + tb[k].y * tw01 + tc[k].y * tw10 + td[k].y * tw11;
            const float gyv = ta[k].z * tw00 + tb[k].z * tw01 + tc[k].z * tw10 + td[k].z * tw11;
            const float I = smp ? Iv : 0.f, gx = smp ? gxv : 0.f, gy = smp ? gyv : 0.f;
            const bool finite = isfinite(I) && isfinite(gx) && isfinite(gy);
            if (((m_in >> k) & 1u) && !finite) m_nf |= 1u << k;
            const float refColor = colors[k];
            const float refRealColor = (float)(aff_a * (double)refColor + aff_b);
            const float residual = I - refRealColor;
            float hw = fabs((double)residual) < A.huber_d ? 1.0f : (float)(A.huber_d / (double)fabsf(residual));
            const double wden = A.oth_d + (double)(gx * gx + gy * gy);
            float wgt = sqrtf((float)rs_div(A.oth_d, wden, rs_rcp_refined(wden)));
            wgt = (float)(0.5f * ((double)wgt + (double)weights[k]));
            const float pf = wgt * wgt * hw * residual * residual;      // energy term factor, :237
            const float hw0 = hw;
            if (hw < 1) hw = sqrtf(hw);
            hw = hw * wgt;
            const float f1 = gx * hw, f2 = gy * hw;                     // hitColor[1], hitColor[2]
            const float drdA = I - fh.b0;
            const float a_ = drdA * hw;
            const float rF = residual * hw;
            const double f1d = (double)f1, f2d = (double)f2, ad = (double)a_, hwd = (double)hw, rFd = (double)rF;
            J00 = (float)((double)J00 + f1d * f1d); J10 = (float)((double)J10 + f1d * f2d); J11 = (float)((double)J11 + f2d * f2d);
            Q00 = (float)((double)Q00 + ad * f1d); Q10 = (float)((double)Q10 + hwd * f1d);
            Q01 = (float)((double)Q01 + ad * f2d); Q11 = (float)((double)Q11 + hwd * f2d);
            rr = (float)((double)rr + rFd * rFd);
            E = (float)((double)E + (double)pf * (2.0 - (double)hw0));                            // energyLeft, BA.cpp:237
            wJI2 = (float)((double)wJI2 + (double)(hw * hw) * (f1d * f1d + f2d * f2d));           // wJI2_sum, BA.cpp:257
            JIr0 += rFd * f1d; JIr1 += rFd * f2d;
            Jabr0 += rFd * (A.opt_a ? ad : 0.0); Jabr1 += rFd * (A.opt_b ? hwd : 0.0);           // BA.cpp:273-278: a zeroed column contributes rF * 0
            B00 += drdA * drdA * hw * hw; B01 += drdA * hw * hw * 1.f; B11 += hw * hw * 1.f * 1.f;
        }
    }

    // first failing pixel in pattern order decides between setNewState(OOB) (:209-212) and setState(OOB) (:220-223)
    const unsigned m_oob = ~m_in & 0xFFu;
    const unsigned m_bad = m_oob | m_nf;
    const int first_bad = m_bad ? __ffs((int)m_bad) - 1 : 8;
    const bool fail_new_oob = !centre_in || (m_bad && ((m_oob >> first_bad) & 1u));
    const bool fail_state_oob = centre_in && m_bad && !((m_oob >> first_bad) & 1u);

    // ---- classification, BA.cpp:66-72,115-118,297-314, and the fused applyRes(copyJacobians = true), BA.cpp:2051-2093
    const float new_idepth = (float)(drescale * idepth);
    double ret_d = 0.0;
    int ns_cnt = -1, flip = 0;
    if (live) {
        float ret = pre_energy;
        float nwo = -1.f;
        int ns_final = pre_new_state;
        bool state_now_oob = (st == CMLHIP_RES_OOB), wrote_e = false;
        if (run) {
            if (centre_in) {                                        // setCenterProjectedTo, :131
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
        A.r_new_energy_wo[r] = nwo;
        A.r_ret_energy[r] = ret;
        ret_d = (double)ret; ns_cnt = ns_final;
        int code = -1;
        if (!state_now_oob) {                                       // applyRes
            if (ns_final == CMLHIP_RES_IN) { A.r_good[r] = 1; flip = 1; code = 2 * r + pre_sel; }
            else A.r_good[r] = 0;
            A.r_state[r] = ns_final;
            A.r_energy[r] = wrote_e ? ret : A.r_new_energy[r];      // state_energy = state_NewEnergy
            A.point_code[pre_ppos] = code;                          // read by the point rows of k_ba_acc and by k_ba_backsub
        }
    }

    // ---- geometric Jacobians, BA.cpp:120-188 (the expression shapes of k_ba_linearize with its per-lane constants folded)
    float* S = &s_stg[ln * RS_SSTRIDE];
    {
        // evaluation-point pair (PRE_RTll_0 / PRE_tTll_0) for the calibration / depth Jacobians: explicit scalar loads HERE (the
        // compiler only scalarises loads it can prove unclobbered, i.e. before the first store of the kernel; holding these 18
        // SGPRs across the pixel loop spilled scalars, and a kernel with a scratch frame pays for it at every dispatch)
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

        // ---- per residual: JpJdF (BA.cpp:2066-2080) and the terms of Hcd, Hdd, bd (BA.cpp:1747-1750)
        if (flip && !(X.dbg_flags & 2)) {
            const float g0 = J00 * d0 + J10 * d1;
            const float g1 = J10 * d0 + J11 * d1;
            float4* o = reinterpret_cast<float4*>(A.r_jpjdf + PS_STRIDE * (size_t)r);
            o[0] = make_float4(xi0[0] * g0 + xi1[0] * g1, xi0[1] * g0 + xi1[1] * g1, xi0[2] * g0 + xi1[2] * g1, xi0[3] * g0 + xi1[3] * g1);
            o[1] = make_float4(xi0[4] * g0 + xi1[4] * g1, xi0[5] * g0 + xi1[5] * g1, Q00 * d0 + Q01 * d1, Q10 * d0 + Q11 * d1);
            o[2] = make_float4(c0[0] * g0 + c1[0] * g1, c0[1] * g0 + c1[1] * g1, c0[2] * g0 + c1[2] * g1, c0[3] * g0 + c1[3] * g1);
            o[3] = make_float4(d0 * g0 + d1 * g1, (float)((double)(float)JIr0 * (double)d0 + (double)(float)JIr1 * (double)d1), 0.f, 0.f);
        }
        // ---- staged operands of the matrix-core reduction; a residual that is not IN (or a lane beyond the tile) stages zeros
#define RS_ST(i, val) S[i] = flip ? (val) : 0.f
#pragma unroll
        for (int i = 0; i < 6; i++) { RS_ST(i, xi0[i]); RS_ST(6 + i, xi1[i]); }
#pragma unroll
        for (int i = 0; i < 4; i++) { RS_ST(12 + i, c0[i]); RS_ST(16 + i, c1[i]); }
        S[20] = 0.f;
        RS_ST(22, J00); RS_ST(23, J10); RS_ST(24, J10); RS_ST(25, J11);
        RS_ST(26, Q00); RS_ST(27, Q10); RS_ST(28, Q01); RS_ST(29, Q11);
        RS_ST(30, B00); RS_ST(31, B01); RS_ST(32, B01); RS_ST(33, B11);
        RS_ST(34, (float)JIr0); RS_ST(35, (float)JIr1); RS_ST(36, (float)Jabr0); RS_ST(37, (float)Jabr1);
        RS_ST(38, rr);
        S[39] = 1.f;
#undef RS_ST
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the workgroup is ONE wave and a wave's LDS operations execute in order:
    __builtin_amdgcn_wave_barrier();                        // only the compiler has to be kept from moving the reads up

    // ---- the wave's contribution to the 13x13 block of its pair: one v_mfma_f32_16x16x4_f32 per residual (see acc_pair_block)
    {
        const unsigned off = c_rs_mfma_off[ln];
        const int oa = c_rs_mfma_a[ln], o1 = off & 255, o2 = (off >> 8) & 255, o3 = (off >> 16) & 255, o4 = off >> 24;
        float4_ acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int li = 0; li < RS_TILE; li++) {
            const float* SL = s_stg + li * RS_SSTRIDE;
            const float av = SL[oa];
            const float bv = SL[o3] * SL[o1] + SL[o4] * SL[o2];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
        if (!(X.dbg_flags & 2)) reinterpret_cast<float4*>(X.part)[(size_t)ti * 64 + ln] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }

    // ---- per-tile partials {energy, n_in, n_oob, n_outlier} (BA.cpp:1565): fixed butterfly order over the wave's residuals
    if (A.lin_partial) {
        double e = ret_d;
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

__global__ void k_ba_idepth_to_res(BAArgs A) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < A.R) A.r_idepth[r] = A.pt_idepth[A.r_point[r]];
}

int cml_launch_linearize_rs(cmlhip_ctx* c, const BAArgs& A) {
    if (c->n_tiles == 0) return CMLHIP_OK;
    if (c->r_idepth_dirty) {                               // pt_idepth was written outside the resident point step (upload, set_idepth, restore, standalone step)
        k_ba_idepth_to_res<<<cml_div_up(A.R, 256), 256, 0, c->stream>>>(A);
        c->r_idepth_dirty = false;
    }
    RsArgs X;
    X.tiles = c->rs_tiles.as<int4>(); X.ntiles = c->n_tiles;
    X.r_px = c->r_px.as<float>(); X.r_py = c->r_py.as<float>(); X.r_colors = c->r_colors.as<float>(); X.r_weights = c->r_weights.as<float>();
    X.part = c->rs_part.as<float>(); X.r_idepth = c->r_idepth.as<double>();
    { static const char* e = getenv("CMLHIP_RS_DBG"); X.dbg_flags = e ? atoi(e) : 0; }      // development: 1 = all texel taps at texel 0, 2 = no tile / reduced-record stores
    if (c->rs_tile == 16) return cml_launch_linearize_rs4(c, A, X);                         // small window: 4 lanes per residual (ba_linearize_rs4.hip)
    if (c->lim.texel_format == CMLHIP_TEXEL_F16) CML_LAUNCH_EV(c, k_ba_lin_rs<true>, c->n_tiles, 64, 0, A, X);
    else CML_LAUNCH_EV(c, k_ba_lin_rs<false>, c->n_tiles, 64, 0, A, X);
    return CMLHIP_OK;
}
