// ba_linearize.hip — photometric residual / Jacobian evaluation of the sliding-window BA.
// Replaces DSOBundleAdjustmentLinearizationContext::linearize (BA.cpp:62-316), the linearizeAll loop
// (BA.cpp:1551-1565), setNewFrameEnergyTH (BA.cpp:2419-2464) and applyRes (BA.cpp:2051-2093).
//
// Mapping (gfx950, wave64): one lane per pattern pixel, 8 lanes per point-residual, 8 residuals per
// wave, 32 per 256-thread workgroup.  Geometry in fp64 and photometrics in fp32 exactly where the
// reference mixes them (its context members are float, Parameter::f() is double), FP contraction off,
// and the 8-pixel inner products are summed in pattern order — so a record is bit-identical to the
// CPU statement order.  Per-pixel terms are exchanged through a 2.3 KB LDS tile (7 floats x 8 lanes
// per residual, ds_read_b128 broadcast reads); the finished 80-float records are staged in LDS and
// leave the CU as coalesced 16-B stores.
#include "cmlhip_internal.h"
#include "ba_common.h"
#include "ba_finish.h"

#pragma clang fp contract(off)

template <bool HALF>
__device__ __forceinline__ float4 load_texel(const void* img, size_t i) {
    if (HALF) {
        uint2 v = reinterpret_cast<const uint2*>(img)[i];
        __half2 a = *reinterpret_cast<__half2*>(&v.x), b = *reinterpret_cast<__half2*>(&v.y);
        return make_float4(__low2float(a), __high2float(a), __low2float(b), 0.f);
    }
    return reinterpret_cast<const float4*>(img)[i];
}

// GradientImage::interpolate, src/cml/image/Array2D.h:265-286
template <bool HALF>
__device__ __forceinline__ void tap3(const void* img, int w, float x, float y, float& I, float& gx, float& gy) {
    const int ix = (int)x, iy = (int)y;
    const float dx = x - (float)ix, dy = y - (float)iy;
    const float dxdy = dx * dy;
    const float w00 = 1 - dx - dy + dxdy, w01 = dx - dxdy, w10 = dy - dxdy, w11 = dxdy;
    const size_t i1 = (size_t)iy * w + ix;
    const float4 a = load_texel<HALF>(img, i1), b = load_texel<HALF>(img, i1 + 1);
    const float4 c = load_texel<HALF>(img, i1 + w), d = load_texel<HALF>(img, i1 + w + 1);
    I = a.x * w00 + b.x * w01 + c.x * w10 + d.x * w11;
    gx = a.y * w00 + b.y * w01 + c.y * w10 + d.y * w11;
    gy = a.z * w00 + b.z * w01 + c.z * w10 + d.z * w11;
}

// value held by the lane of pattern pixel 4 in this lane's group of 8 (ds_swizzle, bit-mask mode: lane' = (lane & 0x18) | 0x04)
__device__ __forceinline__ double bcast_pix4(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x98), hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x98);
    return __hiloint2double(hi, lo);
}

__constant__ int c_star8[16] = {0, -2, -1, -1, 1, -1, -2, 0, 0, 0, 2, 0, -1, 1, 0, 2};   // types.h:1381-1393

#define RES_PER_BLOCK 32
#define NROWD 10          // fp64 rows exchanged per pattern pixel: F1 F2 a hw rF | pf, 2-hw0 | hw*hw, F1^2+F2^2 | zeros

template <bool HALF>
__global__ __launch_bounds__(256) void k_ba_linearize(BAArgs A) {
    // LDS strides chosen against the bank maps of MI355X_MICROARCH.md (b64 reads: 32 lanes = 4 residuals over 64 banks; b32: over 32 banks):
    // residual stride = 16 dwords mod 64 for the fp64 rows (row stride 18 dwords), 25 dwords for the fp32 rows, 88 dwords for the records
    __shared__ double s_shd[RES_PER_BLOCK][104];                                       // [residual][quantity * 9 + pixel], fp64 operands of the sums
    __shared__ float s_shf[RES_PER_BLOCK][25];                                         // [residual][quantity * 8 + pixel], fp32 operands of the product form: drdA, hw, ones
    __shared__ __attribute__((aligned(16))) float s_rec[RES_PER_BLOCK][RJ_STRIDE + 8];
    __shared__ int s_write[RES_PER_BLOCK], s_ns[RES_PER_BLOCK], s_flip[RES_PER_BLOCK], s_app[RES_PER_BLOCK], s_sel[RES_PER_BLOCK], s_pos[RES_PER_BLOCK], s_ppos[RES_PER_BLOCK];
    __shared__ double s_ret[RES_PER_BLOCK];
    const int tid = threadIdx.x, g = tid >> 3, k = tid & 7;
    DBG_BLK(A.dbg, 0, 0);
    if (A.ctl && A.ctl->stop_lin) return;                  // converged in an EARLIER launch (raised by k_ba_acc): the loop of BA::run has left (BA.cpp:879)
    const int r = blockIdx.x * RES_PER_BLOCK + g;
    // ---- per-residual inputs (all 8 lanes of the group read the same addresses: broadcast).  Every load is unconditional
    //      on a clamped index — a `cond ? load : 0` costs its own branch and memory round trip — and what the tail of the
    //      kernel needs (energy, new state, Jacobian-buffer selector, pair-list slot) is fetched in this first round trip.
    const int rc = min(r, A.R - 1);
    const int lin_ = A.r_lin[rc], st_ = A.r_state[rc], p_ = A.r_point[rc], tg_ = A.r_target[rc], hs_ = A.r_host[rc];
    const float pre_energy = A.r_energy[rc];
    const int pre_new_state = A.r_new_state[rc], pre_pos = A.pair_pos[rc], pre_ppos = A.point_pos[rc];
    const unsigned char pre_sel = A.r_sel[rc];
    const bool live = (r < A.R) && !lin_ && (!A.pt_mask || A.pt_mask[p_]);      // pt_mask: tryMarginalize's point subset
    const int st = live ? st_ : CMLHIP_RES_OOB;
    const bool run = live && st != CMLHIP_RES_OOB;
    const int p = live ? p_ : 0;
    const int host = hs_, target = tg_;                       // (static per residual: pair / frame data load with the point data)
    const cmlhip_ba_pair* pc = &A.pairs[host * A.N + target];
    const FrameDev fh = A.frames[host], ft = A.frames[target];
    const double cxd = (double)A.pt_x[p], cyd = (double)A.pt_y[p];
    const double idepth = A.pt_idepth[p];
    const double R0_ = pc->R[0], R1_ = pc->R[1], R2_ = pc->R[2], R3_ = pc->R[3], R4_ = pc->R[4], R5_ = pc->R[5],
                 R6_ = pc->R[6], R7_ = pc->R[7], R8_ = pc->R[8];
    const double t0_ = pc->t[0], t1_ = pc->t[1], t2_ = pc->t[2];
    // evaluation-point pair (PRE_RTll_0 / PRE_tTll_0) for the calibration / depth Jacobians: fetched with the rest, not inside the lane branches
    const double E0 = pc->R0[0], E1 = pc->R0[1], E3 = pc->R0[3], E4 = pc->R0[4], E6 = pc->R0[6], E7 = pc->R0[7];
    const double et0 = pc->t0[0], et1 = pc->t0[1], et2 = pc->t0[2];

    // ---- this lane's pattern pixel, BA.cpp:193-212
    const double sx = cxd + c_star8[2 * k], sy = cyd + c_star8[2 * k + 1];
    const double qx = (sx - A.cx) * A.fxi, qy = (sy - A.cy) * A.fyi;
    const double ppx = (R0_ * qx + R1_ * qy + R2_ * 1.0) + t0_ * idepth;
    const double ppy = (R3_ * qx + R4_ * qy + R5_ * 1.0) + t1_ * idepth;
    const double ppz = (R6_ * qx + R7_ * qy + R8_ * 1.0) + t2_ * idepth;
    const double kx = (ppx / ppz) * A.fx + A.cx, ky = (ppy / ppz) * A.fy + A.cy;
    const bool pix_in = (kx >= 2 && ky >= 2 && kx < A.w - 2 && ky < A.h - 2);

    // ---- centre projection, BA.cpp:102-131: pattern pixel 4 is the offset (0,0), so its lane has already evaluated the very same
    //      expressions on the very same operands; the group takes them from there instead of issuing them again
    const double rx = bcast_pix4(qx), ry = bcast_pix4(qy);
    const double px = bcast_pix4(ppx), py = bcast_pix4(ppy), pz = bcast_pix4(ppz);
    const double Kud = bcast_pix4(kx), Kvd = bcast_pix4(ky);
    const float drescale = (float)(1.0 / pz);
    const bool centre_in = (Kud >= 2 && Kvd >= 2 && Kud < A.w - 2 && Kvd < A.h - 2);

    // BA.cpp:218.  The four texel loads are unconditional on a clamped position (a lane that does not sample reads texel (0,0) and
    // drops the result): inside a lane-divergent `if` the compiler gives every load its own branch and wait
    const bool sample = run && centre_in && pix_in;
    float Iv, gxv, gyv;
    tap3<HALF>(ft.grad0, A.w, sample ? (float)kx : 0.f, sample ? (float)ky : 0.f, Iv, gxv, gyv);
    const float I = sample ? Iv : 0.f, gx = sample ? gxv : 0.f, gy = sample ? gyv : 0.f;
    const bool finite = isfinite(I) && isfinite(gx) && isfinite(gy);

    // first failing pixel in pattern order decides between setNewState(OOB) (:209-212) and setState(OOB) (:220-223)
    const unsigned long long bal_oob = __ballot(!pix_in), bal_nf = __ballot(pix_in && !finite);
    const int shift = (tid & 63) & ~7;
    const unsigned m_oob = (unsigned)(bal_oob >> shift) & 0xFFu, m_nf = (unsigned)(bal_nf >> shift) & 0xFFu;
    const unsigned m_bad = m_oob | m_nf;
    const int first_bad = m_bad ? __ffs((int)m_bad) - 1 : 8;
    const bool fail_new_oob = !centre_in || (m_bad && ((m_oob >> first_bad) & 1u));
    const bool fail_state_oob = centre_in && m_bad && !((m_oob >> first_bad) & 1u);
    const bool ok = run && !fail_new_oob && !fail_state_oob;

    // ---- photometric terms of this pixel, BA.cpp:214-255
    const float refColor = A.pt_colors[8 * (size_t)p + k];
    const float refRealColor = (float)(pc->aff_a * (double)refColor + pc->aff_b);
    const float residual = I - refRealColor;
    float hw = fabs((double)residual) < A.huber_d ? 1.0f : (float)(A.huber_d / (double)fabsf(residual));
    float wgt = sqrtf((float)(A.oth_d / (A.oth_d + (double)(gx * gx + gy * gy))));
    wgt = (float)(0.5f * ((double)wgt + (double)A.pt_weights[8 * (size_t)p + k]));
    const float pf = wgt * wgt * hw * residual * residual;      // energy term factor, :237
    const float hw0 = hw;
    if (hw < 1) hw = sqrtf(hw);
    hw = hw * wgt;
    const float f1 = gx * hw, f2 = gy * hw;                     // hitColor[1], hitColor[2]
    const float drdA = I - fh.b0;
    const float a_ = drdA * hw;
    const float rF = residual * hw;

    // Every operand of the pattern sums is converted ONCE by the lane that owns the pixel (the casts and the per-pixel factors of the
    // energy and wJI2 terms are the reference's own sub-expressions, so the sums below see the same values bit for bit).
    {
        const double f1d = (double)f1, f2d = (double)f2;
        double* D = &s_shd[g][0];
        D[0 * 9 + k] = f1d; D[1 * 9 + k] = f2d; D[2 * 9 + k] = (double)a_; D[3 * 9 + k] = (double)hw; D[4 * 9 + k] = (double)rF;
        D[5 * 9 + k] = (double)pf; D[6 * 9 + k] = 2.0 - (double)hw0;                          // energy term, BA.cpp:237
        D[7 * 9 + k] = (double)(hw * hw); D[8 * 9 + k] = f1d * f1d + f2d * f2d;              // wJI2_sum, BA.cpp:257
        D[9 * 9 + k] = 0.0;
        s_shf[g][k] = drdA; s_shf[g][8 + k] = hw; s_shf[g][16 + k] = 1.f;
    }
    __syncthreads();

    // ---- pattern-order sums, BA.cpp:237,257-271 and the ACTIVE-mode inner products of BA.cpp:1719-1729.
    // The 19 sums over the 8 pattern pixels have three arithmetic forms; lane k evaluates TWO sums of form A and one of B and C
    // (same instructions in every lane, per-lane operand rows), so a wave issues 4 sums per pixel instead of 19:
    //   A  acc = (float)((double)acc + X*Y)     first:  k: J00 J10 J11 Q00 Q10 Q01 Q11 r^T r      second:  k=0 energy, k=1 wJI2_sum
    //   B  acc += rF*Y (fp64)                   k: JI^T r (2), Jab^T r (2)   (a masked column reads the row of zeros: BA.cpp:273-278)
    //   C  acc += ((p*q)*r)*s (fp32)            k: B00 B01 B11               (multiplications by the ones row are exact)
    // Statement order per sum is the reference's.
    const double* SD = &s_shd[g][0];
    const float* SF = &s_shf[g][0];
    const int ax = (0x43232100 >> (4 * k)) & 15, ay = (0x41100110 >> (4 * k)) & 15;     // fp64 rows: 0 F1, 1 F2, 2 a, 3 hw, 4 rF
    const int ex = k == 0 ? 5 : (k == 1 ? 7 : 9), ey = k == 0 ? 6 : (k == 1 ? 8 : 9);
    const int by = ((k == 2 && !A.opt_a) || (k == 3 && !A.opt_b) || k > 3) ? 9 : k;
    const int cp = k < 2 ? 0 : (k == 2 ? 1 : 2), cq = k == 0 ? 0 : (k < 3 ? 1 : 2);     // fp32 rows: 0 drdA, 1 hw, 2 ones
    const int cr = k < 2 ? 1 : 2, cs = k == 0 ? 1 : 2;
    float sumA = 0, sumE = 0, sumC = 0;
    double sumB = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        sumA = (float)((double)sumA + SD[ax * 9 + j] * SD[ay * 9 + j]);
        sumE = (float)((double)sumE + SD[ex * 9 + j] * SD[ey * 9 + j]);
        sumB += SD[4 * 9 + j] * SD[by * 9 + j];
        sumC += SF[cp * 8 + j] * SF[cq * 8 + j] * SF[cr * 8 + j] * SF[cs * 8 + j];
    }
    const float E = sumE;                                             // lane 0 of the group: the energy; its neighbour holds wJI2_sum
    const float wJI2 = __shfl(sumE, ((tid & 63) & ~7) + 1);

    // ---- geometric Jacobians, BA.cpp:120-188 (computed by every lane, each stores its share)
    const float new_idepth = (float)(drescale * idepth);
    const float u = (float)px, v = (float)py;            // BA.cpp:121-122: un-normalised x,y, literal
    const float fxf = (float)A.fx, fyf = (float)A.fy;
    float* rec = s_rec[g];
    // Same instructions in every lane, per-lane operands (four lane-divergent branches with two fp64 divisions cost ~1 us):
    //   lane k < 6   Jpdxi[0][k], Jpdxi[1][k]                                       (fp32, BA.cpp:133-147)
    //   lane k       one entry of Jpdc: ((m * q) + add) * scale with q = (sfac * drescale) * (Ea * w - Eb) / s2   (:150-176)
    //                lanes 0-3 -> Jpdc[0][k], lanes 4-7 -> Jpdc[1][k-4]; multiplications by 1 and the addition of -0.0 are exact
    //   lane k < 2   Jpdd[k] = drescale * (t0[k] - t0[2] * {u,v}) * {fx,fy}                                        (:178-182)
    {
        const float xi0 = k == 0 ? new_idepth * fxf : (k == 1 ? 0.f : (k == 2 ? -new_idepth * u * fxf : (k == 3 ? -u * v * fxf : (k == 4 ? (1 + u * u) * fxf : -v * fxf))));
        const float xi1 = k == 0 ? 0.f : (k == 1 ? new_idepth * fyf : (k == 2 ? -new_idepth * v * fyf : (k == 3 ? -(1 + v * v) * fyf : (k == 4 ? u * v * fyf : u * fyf))));
        if (k < 6) { rec[O_XI0 + k] = xi0; rec[O_XI1 + k] = xi1; }
        const bool odd = k & 1, lo = k < 4;
        const double Ea = odd ? E7 : E6, Eb = lo ? (odd ? E1 : E0) : (odd ? E4 : E3);
        const float wq = lo ? u : v;
        const float sfac = lo ? (odd ? fxf : 1.f) : (odd ? 1.f : fyf), s2 = lo ? (odd ? fyf : 1.f) : (odd ? 1.f : fxf);
        const double q = (sfac * drescale) * (Ea * wq - Eb) / s2;
        const double m = (k & 2) ? 1.0 : (odd ? ry : rx);
        const double add = k == 0 ? (double)u : (k == 5 ? (double)v : ((k == 2 || k == 7) ? 1.0 : -0.0));
        const double scl = (k & 2) ? A.scale_c : A.scale_f;
        rec[lo ? O_C0 + k : O_C1 + (k - 4)] = (float)(((m * q) + add) * scl);
        const double dd = drescale * ((odd ? et1 : et0) - et2 * (odd ? v : u)) * (odd ? fyf : fxf);
        if (k < 2) rec[O_DD + k] = (float)dd;
    }
    // the sums of this lane: form A -> JIdx2 (J00 J10 J10 J11), JabJIdx (Q00 Q10 Q01 Q11), r^T r; B -> JI^T r, Jab^T r; C -> Jab2
    rec[k < 1 ? O_JI2 : (k < 2 ? O_JI2 + 1 : (k < 3 ? O_JI2 + 3 : (k < 7 ? O_JABJI + (k - 3) : O_X_RR)))] = sumA;
    if (k == 1) rec[O_JI2 + 2] = sumA;
    if (k < 4) rec[O_X_JIR + k] = (float)sumB;               // O_X_JIR, O_X_JIR+1, O_X_JABR, O_X_JABR+1 are contiguous
    if (k < 3) rec[k == 0 ? O_JAB2 : (k == 1 ? O_JAB2 + 1 : O_JAB2 + 3)] = sumC;
    if (k == 1) rec[O_JAB2 + 2] = sumC;
    if (k == 7) rec[O_X_RR + 1] = 0.f;
    rec[O_RES + k] = rF;
    rec[O_JI0 + k] = f1;
    rec[O_JI1 + k] = f2;
    rec[O_JAB0 + k] = A.opt_a ? a_ : 0.f;
    rec[O_JAB1 + k] = A.opt_b ? hw : 0.f;

    // ---- classification, BA.cpp:66-72,115-118,297-314
    if (k == 0) { s_write[g] = run ? 1 : 0; s_ret[g] = 0.0; s_ns[g] = -1; s_flip[g] = 0; s_app[g] = 0; s_sel[g] = pre_sel; s_pos[g] = pre_pos; s_ppos[g] = pre_ppos; }
    if (A.records_only) {
        // cml_materialize_records: the resident kernel kept this pass's Jacobians in reduced form only; re-create the efsJ record of
        // every good residual from the unchanged inputs (same expressions, same bits), touch no state
        if (live && k == 0) {
            const int good = A.r_good[r];
            s_write[g] = good; s_flip[g] = good; s_app[g] = 1;
        }
    } else if (live && k == 0) {
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
        s_ret[g] = (double)ret; s_ns[g] = ns_final;
        if (A.fuse_apply && !state_now_oob) {                       // applyRes(copyJacobians = true), BA.cpp:2051-2093
            if (ns_final == CMLHIP_RES_IN) { A.r_good[r] = 1; s_flip[g] = 1; }
            else A.r_good[r] = 0;
            s_app[g] = 1;
            A.r_state[r] = ns_final;
            A.r_energy[r] = wrote_e ? ret : A.r_new_energy[r];      // state_energy = state_NewEnergy
        }
    }
    __syncthreads();

    // ---- fused applyRes: swap(rJ, efsJ) = flip of the buffer selector, JpJdF from the fresh record (BA.cpp:2064-2080)
    if (A.fuse_apply && s_flip[g]) {
        const float* Jn = s_rec[g];
        const float g0 = Jn[O_JI2 + 0] * Jn[O_DD] + Jn[O_JI2 + 2] * Jn[O_DD + 1];
        const float g1 = Jn[O_JI2 + 1] * Jn[O_DD] + Jn[O_JI2 + 3] * Jn[O_DD + 1];
        float v;
        if (k < 6) v = Jn[O_XI0 + k] * g0 + Jn[O_XI1 + k] * g1;
        else if (k == 6) v = Jn[O_JABJI + 0] * Jn[O_DD] + Jn[O_JABJI + 2] * Jn[O_DD + 1];
        else v = Jn[O_JABJI + 1] * Jn[O_DD] + Jn[O_JABJI + 3] * Jn[O_DD + 1];
        A.r_jpjdf[PS_STRIDE * (size_t)r + k] = v;
        // this residual's terms of the point sums Hcd, Hdd, bd (BA.cpp:1747-1750), read by the point rows of k_ba_acc
        float w;
        if (k < 4) w = Jn[O_C0 + k] * g0 + Jn[O_C1 + k] * g1;
        else if (k == 4) w = Jn[O_DD] * g0 + Jn[O_DD + 1] * g1;
        else if (k == 5) w = (float)((double)Jn[O_X_JIR] * (double)Jn[O_DD] + (double)Jn[O_X_JIR + 1] * (double)Jn[O_DD + 1]);
        else w = 0.f;
        A.r_jpjdf[PS_STRIDE * (size_t)r + 8 + k] = w;
    }

    // ---- coalesced copy-out of the finished records into the residual's rJ buffer (the one that is not efsJ)
    const int r0 = blockIdx.x * RES_PER_BLOCK;
    for (int i = tid; i < RES_PER_BLOCK * (RJ_STRIDE / 4); i += 256) {
        const int gg = i / (RJ_STRIDE / 4), q = i % (RJ_STRIDE / 4);
        const int rr_ = r0 + gg;
        if (rr_ >= A.R) break;
        if (!s_write[gg]) continue;
        float* dst = (s_sel[gg] ? A.rj0 : A.rj1) + (size_t)rr_ * RJ_STRIDE;
        reinterpret_cast<float4*>(dst)[q] = reinterpret_cast<const float4*>(s_rec[gg])[q];
    }
    if (tid < RES_PER_BLOCK && s_app[tid]) {                        // efsJ code of the pair list (read by the accumulate kernel)
        const int rr_ = r0 + tid;
        int code = -1;
        if (s_flip[tid]) { const unsigned char sl = (unsigned char)(s_sel[tid] ^ 1); A.r_sel[rr_] = sl; code = 2 * rr_ + sl; }
        A.pair_code[s_pos[tid]] = code;
        A.point_code[s_ppos[tid]] = code;
    }
    // ---- per-block partials {energy, n_in, n_oob, n_outlier} (BA.cpp:1565): fixed butterfly order over the 32 residuals
    if (tid < 64 && A.lin_partial) {
        double e = tid < RES_PER_BLOCK ? s_ret[tid & (RES_PER_BLOCK - 1)] : 0.0;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) e += __shfl_xor(e, o);
        const int ns = tid < RES_PER_BLOCK ? s_ns[tid & (RES_PER_BLOCK - 1)] : -1;
        const int c0 = __popcll(__ballot(ns == CMLHIP_RES_IN)), c1 = __popcll(__ballot(ns == CMLHIP_RES_OOB)), c2 = __popcll(__ballot(ns == CMLHIP_RES_OUTLIER));
        if (tid == 0) {
            double* o = A.lin_partial + 4 * (size_t)blockIdx.x;
            o[0] = e; o[1] = (double)c0; o[2] = (double)c1; o[3] = (double)c2;
        }
    }
    // ---- resident loop: the convergence test of doStepFromBackup (BA.cpp:996-1027) on the sums of the step that preceded this
    //      pass; `if (canbreak && it >= 1) break` (BA.cpp:879) becomes a sticky flag that every later kernel checks first
    if (A.ctl && blockIdx.x == 0 && tid == 0) {
        float sumID = 0, sumNID = 0, numID = 0;
        for (int b = 0; b < A.n_step_blocks; b++) { sumID += A.step_partial_ro[4 * b]; sumNID += A.step_partial_ro[4 * b + 1]; numID += A.step_partial_ro[4 * b + 2]; }
        float sumA = A.ctl->frame_sums[0], sumB = A.ctl->frame_sums[1], sumT = A.ctl->frame_sums[2], sumR = A.ctl->frame_sums[3];
        const float nf = (float)A.N;
        sumA /= nf; sumB /= nf; sumR /= nf; sumT /= nf; sumID /= numID; sumNID /= numID;
        const bool canbreak = sqrtf(sumA) < 0.0005 * A.th_opt && sqrtf(sumB) < 0.00005 * A.th_opt && sqrtf(sumR) < 0.00005 * A.th_opt &&
                              sqrtf(sumT) * sumNID < 0.00005 * A.th_opt;
        A.ctl->iters_done = A.it_index + 1;
        if (canbreak && A.it_index >= 1) A.ctl->stop = 1;
    }
    (void)ok;
    DBG_BLK_END(A.dbg, 0);
}

// ------------------------------------------------------------------------------------------------
// standalone tail of a residual pass (the iteration pipeline runs it as the second workgroup of the solve launch)
__global__ __launch_bounds__(1024) void k_ba_lin_finish(BAArgs A, const int* __restrict__ newframe_res, int n_newframe,
                                                        const double* __restrict__ lin_partial, int n_partial,
                                                        LinSummary* __restrict__ out, FrameDev* __restrict__ frames_rw) {
    __shared__ unsigned s_u32[264];
    __shared__ double s_f64[1024];
    lin_finish_block(A, newframe_res, n_newframe, lin_partial, n_partial, out, frames_rw, s_u32, s_f64);
}

// ------------------------------------------------------------------------------------------------
// applyRes, BA.cpp:2051-2093.  "swap(rJ, efsJ)" is a flip of the residual's buffer selector.
__global__ void k_ba_apply(BAArgs A, int copy) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.R || A.r_lin[r]) return;
    if (copy) {
        if (A.r_state[r] == CMLHIP_RES_OOB) return;
        if (A.r_new_state[r] == CMLHIP_RES_IN) {
            A.r_good[r] = 1;
            const unsigned char sel = A.r_sel[r] ^ 1;
            A.r_sel[r] = sel;
            A.pair_code[A.pair_pos[r]] = 2 * r + sel;
            A.point_code[A.point_pos[r]] = 2 * r + sel;
            const float* J = (sel ? A.rj1 : A.rj0) + (size_t)r * RJ_STRIDE;
            const float g0 = J[O_JI2 + 0] * J[O_DD] + J[O_JI2 + 2] * J[O_DD + 1];
            const float g1 = J[O_JI2 + 1] * J[O_DD] + J[O_JI2 + 3] * J[O_DD + 1];
            float* o = A.r_jpjdf + PS_STRIDE * (size_t)r;
#pragma unroll
            for (int i = 0; i < 6; i++) o[i] = J[O_XI0 + i] * g0 + J[O_XI1 + i] * g1;
            o[6] = J[O_JABJI + 0] * J[O_DD] + J[O_JABJI + 2] * J[O_DD + 1];
            o[7] = J[O_JABJI + 1] * J[O_DD] + J[O_JABJI + 3] * J[O_DD + 1];
#pragma unroll
            for (int i = 0; i < 4; i++) o[PS_HCD + i] = J[O_C0 + i] * g0 + J[O_C1 + i] * g1;       // terms of Hcd, Hdd, bd (BA.cpp:1747-1750)
            o[PS_HDD] = J[O_DD] * g0 + J[O_DD + 1] * g1;
            o[PS_BD] = (float)((double)J[O_X_JIR] * (double)J[O_DD] + (double)J[O_X_JIR + 1] * (double)J[O_DD + 1]);
            o[14] = 0.f; o[15] = 0.f;
        } else {
            A.r_good[r] = 0;
            A.pair_code[A.pair_pos[r]] = -1;
            A.point_code[A.point_pos[r]] = -1;
        }
    }
    A.r_state[r] = A.r_new_state[r];
    A.r_energy[r] = A.r_new_energy[r];
}

// ------------------------------------------------------------------------------------------------ marginalisation, residual side
// tryMarginalize's residual loop for the selected points (BA.cpp:2291-2304) = k_ba_marg_reset -> k_ba_linearize with the
// point mask and fused applyRes(true) -> k_ba_marg_fix.
__global__ void k_ba_marg_reset(BAArgs A) {                      // resetOOB (DSOResidual.h:81-86) + isLinearized = false
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.R || !A.pt_mask[A.r_point[r]] || A.r_dead[r]) return;        // (a residual the closing pass of the run removed stays removed)
    A.r_new_energy[r] = 0.f; A.r_energy[r] = 0.f;
    A.r_new_state[r] = CMLHIP_RES_OUTLIER; A.r_state[r] = CMLHIP_RES_IN;
    A.r_lin_rw[r] = 0;
    A.point_tgt_rw[A.point_pos[r]] &= 255;
}
// fixLinearization (BA.cpp:2210-2238) of the selected points' good residuals: res_toZero = resF - [JI*Jp Ja]*delta in the
// statement order of the reference's SSE code (fp contraction is off in this file), isLinearized = true
__global__ void k_ba_marg_fix(BAArgs A, const float* __restrict__ adHTd, const double* __restrict__ cdelta, int* counter) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.R) return;
    const int p = A.r_point[r];
    if (!A.pt_mask[p] || !A.r_good[r]) return;
    const float* J = (A.r_sel[r] ? A.rj1 : A.rj0) + (size_t)r * RJ_STRIDE;      // efsJ
    const float* dp = adHTd + 8 * (A.pt_host[p] + A.r_target[r] * A.N);
    const float deltaF = (float)(A.pt_idepth[p] - (double)A.pt_idepth_zero[p]);
    const float Jpx = cml_jp_delta(J + O_XI0, dp, J + O_C0, cdelta, J[O_DD], deltaF, true);       // fixLinearization: the cast stays inside the dot
    const float Jpy = cml_jp_delta(J + O_XI1, dp, J + O_C1, cdelta, J[O_DD + 1], deltaF, true);
    for (int i = 0; i < 8; i++) {
        float rtz = J[O_RES + i];
        rtz = rtz - J[O_JI0 + i] * Jpx;
        rtz = rtz - J[O_JI1 + i] * Jpy;
        rtz = rtz - J[O_JAB0 + i] * dp[6];
        rtz = rtz - J[O_JAB1 + i] * dp[7];
        A.r_rtz[8 * (size_t)r + i] = rtz;
    }
    A.r_lin_rw[r] = 1;
    A.pair_code[A.pair_pos[r]] = -1;                             // no longer in the ACTIVE sums
    A.point_tgt_rw[A.point_pos[r]] |= 256;
    atomicAdd(counter, 1);
}
// calcLEnergy's residual sum (BA.cpp:2149-2203): per point, over its LINEARIZED good residuals, (2 res_toZero + J delta).J delta,
// plus deltaF^2 priorF; one thread per point, fixed-order block partials (fp64; the reference's Accumulator11 is a tiered fp32 sum)
__global__ __launch_bounds__(256) void k_ba_lin_energy(BAArgs A, const float* __restrict__ adHTd, const double* __restrict__ cdelta,
                                                      double* __restrict__ partial, int* __restrict__ num) {
    __shared__ double s_e[256];
    __shared__ int s_n[256];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    double E = 0;
    int n = 0;
    if (p < A.P) {
        const float dd = (float)(A.pt_idepth[p] - (double)A.pt_idepth_zero[p]);
        for (int kk = A.by_point_off[p]; kk < A.by_point_off[p + 1]; kk++) {
            const int r = A.by_point[kk];
            if (!A.r_lin[r] || !A.r_good[r]) continue;
            n++;
            const float* J = (A.r_sel[r] ? A.rj1 : A.rj0) + (size_t)r * RJ_STRIDE;
            const float* dp = adHTd + 8 * (A.pt_host[p] + A.r_target[r] * A.N);
            const float Jpx = cml_jp_delta(J + O_XI0, dp, J + O_C0, cdelta, J[O_DD], dd, false);
            const float Jpy = cml_jp_delta(J + O_XI1, dp, J + O_C1, cdelta, J[O_DD + 1], dd, false);
            for (int i = 0; i < 8; i++) {
                float Jdelta = J[O_JI0 + i] * Jpx;
                Jdelta = Jdelta + J[O_JI1 + i] * Jpy;
                Jdelta = Jdelta + J[O_JAB0 + i] * dp[6];
                Jdelta = Jdelta + J[O_JAB1 + i] * dp[7];
                float r0 = A.r_rtz[8 * (size_t)r + i];
                r0 = r0 + r0;
                r0 = r0 + Jdelta;
                E += (double)(Jdelta * r0);
            }
        }
        E += (double)(dd * dd * A.pt_prior[p]);
    }
    s_e[threadIdx.x] = E; s_n[threadIdx.x] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        double e = 0; int c = 0;
        for (int i = 0; i < 256; i++) { e += s_e[i]; c += s_n[i]; }
        partial[blockIdx.x] = e; num[blockIdx.x] = c;
    }
}

int cml_launch_marg_reset(cmlhip_ctx* c, const BAArgs& A) {
    if (A.R > 0) k_ba_marg_reset<<<cml_div_up(A.R, 256), 256, 0, c->stream>>>(A);
    return CMLHIP_OK;
}
int cml_launch_marg_fix(cmlhip_ctx* c, const BAArgs& A, const float* adHTd, const double* cdelta, int* counter) {
    if (A.R > 0) k_ba_marg_fix<<<cml_div_up(A.R, 256), 256, 0, c->stream>>>(A, adHTd, cdelta, counter);
    return CMLHIP_OK;
}
int cml_launch_lin_energy(cmlhip_ctx* c, const BAArgs& A, const float* adHTd, const double* cdelta, double* partial, int* num) {
    if (A.P > 0) k_ba_lin_energy<<<cml_div_up(A.P, 256), 256, 0, c->stream>>>(A, adHTd, cdelta, partial, num);
    return CMLHIP_OK;
}

int cml_launch_linearize(cmlhip_ctx* c, const BAArgs& A) {
    const int blocks = cml_div_up(A.R, RES_PER_BLOCK);
    if (A.lin_partial) c->lin_partial_n = blocks;
    if (blocks == 0) return CMLHIP_OK;
    if (c->lim.texel_format == CMLHIP_TEXEL_F16) CML_LAUNCH_EV(c, k_ba_linearize<true>, blocks, 256, 0, A);
    else CML_LAUNCH_EV(c, k_ba_linearize<false>, blocks, 256, 0, A);
    return CMLHIP_OK;
}
int cml_launch_lin_finish(cmlhip_ctx* c, const BAArgs& A, size_t out_offset) {
    k_ba_lin_finish<<<1, 1024, 0, c->stream>>>(A, c->newframe_res.as<int>(), c->n_newframe, c->lin_partial.as<double>(), c->lin_partial_n,
                                                reinterpret_cast<LinSummary*>(c->scal.as<char>() + out_offset), c->frames.as<FrameDev>());
    return CMLHIP_OK;
}
int cml_launch_apply(cmlhip_ctx* c, const BAArgs& A, int copy) {
    if (A.R == 0) return CMLHIP_OK;
    k_ba_apply<<<cml_div_up(A.R, 256), 256, 0, c->stream>>>(A, copy);
    return CMLHIP_OK;
}

// The resident residual kernel (ba_linearize_rs.hip) keeps a pass's Jacobians in reduced form (pair tiles + 14 floats per residual).
// Before anything reads a 74-float record, or changes the state the records would be re-created from, the efsJ record of every good
// residual is re-created here: k_ba_linearize in records-only mode on the unchanged inputs (bit-identical by construction), selector
// flip and pair / point codes as applyRes would have left them.  No state, energy or classification is touched.
int cml_materialize_records(cmlhip_ctx* c) {
    if (!c->efs_in_partials) return CMLHIP_OK;
    c->efs_in_partials = false;
    if (!c->ba_uploaded || !c->ba_pairs_set || c->R == 0) return CMLHIP_OK;
    BAArgs A;
    cml_make_ba_args(c, A);
    A.records_only = 1; A.fuse_apply = 1; A.lin_partial = nullptr; A.ctl = nullptr;
    return cml_launch_linearize(c, A);
}
