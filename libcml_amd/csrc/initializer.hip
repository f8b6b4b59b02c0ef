// initializer.hip — the coarse initializer's per-evaluation work on the device.
// Replaces DSOInitializer::calcResAndGS (src/cml/optimization/dso/DSOInitializer.cpp:451-750; Accumulator9 / Accumulator11,
// MatrixAccumulators.h:94-184,1006-1390): projection of the 8 pattern pixels of every initializer point into the tracked
// frame, bilinear gather, Huber weights, the per-point JbBuffer rows, the 9x9 system and its idepth Schur complement.
//
// One launch.  A lane owns one point (its 8 residuals are summed in pattern order in fp32, exactly as the reference does, so
// every per-point output is bit-exact); the two 9x9 sums over the points run on the matrix cores (v_mfma_f32_16x16x4_f32 as
// a reduction over points: D = sum_k a_k b_k^T), one partial row per workgroup, added in block order by the (synchronous)
// caller.  All 32 texel loads of a lane are issued against clamped addresses before the first use: no load sits under a
// lane-divergent branch.  The reference's second and third point loops (alpha energy, Schur accumulation) need nothing
// from a global reduction — alphaOpt depends only on |t|^2 * npts because EAlpha is never fed (:665-699) — so they fold
// into the same launch.
#include "cmlhip_internal.h"
#include <vector>

#pragma clang fp contract(off)

typedef float ini_float4 __attribute__((ext_vector_type(4)));
#define INI_LD 17
#define INI_NRED 96    // 45 (Acc9 upper) + 45 (Acc9SC upper) + E + pad

struct InitArgs {
    const void* img; int w, h, n;
    cmlhip_init_params P;
    float alpha_opt;
    cmlhip_init_point* pts;
    float* partial;        // gridDim x INI_NRED
};

template <bool HALF>
__device__ __forceinline__ float4 ini_texel(const void* img, size_t i) {
    if (HALF) {
        uint2 v = reinterpret_cast<const uint2*>(img)[i];
        __half2 a = *reinterpret_cast<__half2*>(&v.x), b = *reinterpret_cast<__half2*>(&v.y);
        return make_float4(__low2float(a), __high2float(a), __low2float(b), 0.f);
    }
    return reinterpret_cast<const float4*>(img)[i];
}

template <bool HALF>
__global__ __launch_bounds__(256) void k_init_calc_res_and_gs(InitArgs A) {
    __shared__ float s_a[4][64][INI_LD], s_b[4][64][INI_LD];
    __shared__ float s_tile[2][4][256];
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const int i = blockIdx.x * 256 + tid;
    const bool live = i < A.n;
    const cmlhip_init_params& P = A.P;
    cmlhip_init_point* pt = A.pts + (live ? i : 0);

    float dp[9][8];                 // dp0..dp7, r of the 8 pattern residuals (zero where not computed)
    float jbv[9], jbw = 0.f;        // the JbBuffer row after the Schur loop (jbw = jb[9] = 1/(1+H_dd))
    float eterm = 0.f;
    bool good_new = false;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        jbv[k] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) dp[k][j] = 0.f;
    }
    if (live) {
        const float idn = pt->idepth_new, iR = pt->iR, oth = pt->outlier_th, e0 = pt->energy[0], e1 = pt->energy[1];
        bool good = pt->is_good != 0;
        float tz[8], tu[8], tv[8], tnid[8], w00[8], w01[8], w10[8], w11[8];
        size_t ti[8];
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {                                    // :484-513
            const float p0 = pt->p_pattern[idx][0], p1 = pt->p_pattern[idx][1], p2 = pt->p_pattern[idx][2];
            float q[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float v = 0.0f + (P.RKi[3 * k] * p0 + (P.RKi[3 * k + 1] * p1 + P.RKi[3 * k + 2] * p2));   // Eigen's order: e0 + (e1 + e2)
                q[k] = v + P.t[k] * idn;
            }
            tz[idx] = q[2];
            tu[idx] = q[0] / q[2];
            tv[idx] = q[1] / q[2];
            const float Ku = P.fx * tu[idx] + P.cx, Kv = P.fy * tv[idx] + P.cy;
            tnid[idx] = idn / q[2];
            const bool in = Ku > 1 && Kv > 1 && Ku < A.w - 2 && Kv < A.h - 2 && tnid[idx] > 0;
            if (!in) good = false;
            const float cu = in ? Ku : 1.5f, cv = in ? Kv : 1.5f;               // clamped tap: the value is unused when !in
            const int ix = (int)cu, iy = (int)cv;
            const float dx = cu - (float)ix, dy = cv - (float)iy, dxdy = dx * dy;
            w00[idx] = 1 - dx - dy + dxdy; w01[idx] = dx - dxdy; w10[idx] = dy - dxdy; w11[idx] = dxdy;
            ti[idx] = (size_t)iy * A.w + ix;
        }
        float4 ta[8], tb[8], tc[8], td[8];
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            ta[idx] = ini_texel<HALF>(A.img, ti[idx]); tb[idx] = ini_texel<HALF>(A.img, ti[idx] + 1);
            tc[idx] = ini_texel<HALF>(A.img, ti[idx] + A.w); td[idx] = ini_texel<HALF>(A.img, ti[idx] + A.w + 1);
        }
        float color[8];
#pragma unroll
        for (int idx = 0; idx < 8; idx++) color[idx] = pt->color[idx];

        float maxstep_pt = 1e10f;                                              // :520
        if (!good) {                                                           // :521-527
            eterm = e0;
            pt->is_good = 0;
            pt->energy_new[0] = e0; pt->energy_new[1] = e1;
            pt->is_good_new = 0;
            pt->maxstep = maxstep_pt;
        } else {
            float jb[10];
#pragma unroll
            for (int k = 0; k < 10; k++) jb[k] = 0.f;
            bool isGood = true;
            float energy = 0.f;
#pragma unroll
            for (int idx = 0; idx < 8; idx++) {                                // :546-613
                const float h0 = ta[idx].x * w00[idx] + tb[idx].x * w01[idx] + tc[idx].x * w10[idx] + td[idx].x * w11[idx];
                const float h1 = ta[idx].y * w00[idx] + tb[idx].y * w01[idx] + tc[idx].y * w10[idx] + td[idx].y * w11[idx];
                const float h2 = ta[idx].z * w00[idx] + tb[idx].z * w01[idx] + tc[idx].z * w10[idx] + td[idx].z * w11[idx];
                const float rlR = color[idx];
                if (isGood && !(isfinite(rlR) && isfinite(h0) && isfinite(h1) && isfinite(h2))) isGood = false;   // break
                if (isGood) {
                    const float residual = h0 - P.aff_a * rlR - P.aff_b;
                    float hw = fabsf(residual) < P.huber ? 1 : P.huber / fabsf(residual);
                    energy += hw * residual * residual * (2 - hw);
                    const float dxdd = (P.t[0] - P.t[2] * tu[idx]) / tz[idx];
                    const float dydd = (P.t[1] - P.t[2] * tv[idx]) / tz[idx];
                    if (hw < 1) hw = sqrtf(hw);
                    const float dxInterp = hw * h1 * P.fx, dyInterp = hw * h2 * P.fy;
                    const float u = tu[idx], v = tv[idx], nid = tnid[idx];
                    float d[9];
                    d[0] = nid * dxInterp;
                    d[1] = nid * dyInterp;
                    d[2] = -nid * (u * dxInterp + v * dyInterp);
                    d[3] = -u * v * dxInterp - (1 + v * v) * dyInterp;
                    d[4] = (1 + u * u) * dxInterp + u * v * dyInterp;
                    d[5] = -v * dxInterp + u * dyInterp;
                    d[6] = -hw * P.aff_a * rlR;
                    d[7] = -hw * 1;
                    const float ddv = dxInterp * dxdd + dyInterp * dydd;
                    d[8] = hw * residual;
                    const float mx = dxdd * P.fx, my = dydd * P.fy;
                    const float maxstep = 1.0f / sqrtf(mx * mx + my * my);
                    if (maxstep < maxstep_pt) maxstep_pt = maxstep;
#pragma unroll
                    for (int k = 0; k < 9; k++) { jb[k] += d[k] * ddv; dp[k][idx] = d[k]; }
                    jb[9] += ddv * ddv;
                }
            }
            pt->maxstep = maxstep_pt;
            if (!isGood || energy > oth * 20) {                                // :616-623
                eterm = e0;
                pt->is_good_new = 0;
                pt->energy_new[0] = e0; pt->energy_new[1] = e1;
            } else {
                good_new = true;
                eterm = energy;
                pt->is_good_new = 1;
                pt->energy_new[0] = energy;
                pt->energy_new[1] = (idn - 1) * (idn - 1);                     // :672
                pt->last_hessian_new = jb[9];                                  // :707
                jb[8] += A.alpha_opt * (idn - 1);
                jb[9] += A.alpha_opt;
                if (A.alpha_opt == 0) {
                    jb[8] += P.coupling_weight * (idn - iR);
                    jb[9] += P.coupling_weight;
                }
                jb[9] = 1 / (1 + jb[9]);
#pragma unroll
                for (int k = 0; k < 9; k++) jbv[k] = jb[k];
                jbw = jb[9];
            }
#pragma unroll
            for (int k = 0; k < 10; k++) pt->jb[k] = jb[k];
        }
    }
    // ---- Acc9: sum over inlier points and pattern pixels of J J^T, J = (dp0..dp7, r); E rides in row/col 9,10 of step 0
    ini_float4 acc = {0.f, 0.f, 0.f, 0.f}, accsc = {0.f, 0.f, 0.f, 0.f};
    const int e = l & 15, kq = l >> 4;
    const float gm = good_new ? 1.f : 0.f;
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
#pragma unroll
        for (int k = 0; k < 9; k++) s_a[wv][l][k] = dp[k][idx] * gm;
        s_a[wv][l][9] = (idx == 0 && live) ? 1.f : 0.f;
        s_a[wv][l][10] = idx == 0 ? eterm : 0.f;
#pragma unroll
        for (int k = 11; k < 16; k++) s_a[wv][l][k] = 0.f;
        // (LDS accesses of one wave are ordered: no barrier between the stores above and the loads below)
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const float av = s_a[wv][4 * m + kq][e];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, av, acc, 0, 0, 0);
        }
    }
    // ---- Acc9SC: sum over inlier points of w * jb jb^T (updateSingleWeighted, ACC.h:1286-1359)
#pragma unroll
    for (int k = 0; k < 9; k++) { s_a[wv][l][k] = jbv[k] * jbw; s_b[wv][l][k] = jbv[k]; }
#pragma unroll
    for (int k = 9; k < 16; k++) { s_a[wv][l][k] = 0.f; s_b[wv][l][k] = 0.f; }
#pragma unroll
    for (int m = 0; m < 16; m++) {
        const float av = s_a[wv][4 * m + kq][e], bv = s_b[wv][4 * m + kq][e];
        accsc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accsc, 0, 0, 0);
    }
    // D: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
        s_tile[0][wv][(4 * (l >> 4) + rg) * 16 + (l & 15)] = acc[rg];
        s_tile[1][wv][(4 * (l >> 4) + rg) * 16 + (l & 15)] = accsc[rg];
    }
    __syncthreads();
    if (tid < INI_NRED) {
        int which = 0, src = -1;
        if (tid < 90) {
            which = tid / 45;
            int k = tid % 45, r = 0;
            while (k >= 9 - r) { k -= 9 - r; r++; }
            src = r * 16 + (r + k);
        } else if (tid == 90) src = 9 * 16 + 10;                               // sum of the E terms (a_9 = 1, b_10 = E)
        float v = 0.f;
        if (src >= 0) v = ((s_tile[which][0][src] + s_tile[which][1][src]) + s_tile[which][2][src]) + s_tile[which][3][src];
        A.partial[(size_t)blockIdx.x * INI_NRED + tid] = v;
    }
}

extern "C" {

int cmlhip_initializer_calc_res_and_gs(cmlhip_ctx* c, uint64_t image_id, int level, const cmlhip_init_params* prm, int n,
                                       cmlhip_init_point* points, float* H_out, float* b_out, float* H_out_sc, float* b_out_sc,
                                       float res[3]) { CML_DEV(c);
    if (!c || !prm || n < 0 || (n > 0 && !points) || !H_out || !b_out || !H_out_sc || !b_out_sc || !res) return CMLHIP_ERR_INVALID;
    const Pyramid* py = cml_find_pyr(c, image_id);
    CML_REQUIRE(c, py && level >= 0 && level < py->levels && py->lv[level].grad, CMLHIP_ERR_NOT_FOUND, "tracked image / level not in the pyramid cache");
    // alphaEnergy / alphaOpt, DSOInitializer.cpp:681-699 (EAlpha.A == 0: the accumulator is never fed)
    float alphaEnergy = (float)((double)prm->alpha_w * ((double)0.0f + prm->t_sqnorm * (double)n));
    float alphaOpt;
    if (alphaEnergy > prm->alpha_k * n) { alphaOpt = 0; alphaEnergy = prm->alpha_k * n; }
    else alphaOpt = prm->alpha_w;
    double sums[INI_NRED] = {0};
    if (n > 0) {
        const int nb = cml_div_up(n, 256);
        int rc;
        if ((rc = cml_ensure(c, c->ini_points, sizeof(cmlhip_init_point) * (size_t)n))) return rc;
        if ((rc = cml_ensure(c, c->ini_partial, sizeof(float) * INI_NRED * (size_t)nb))) return rc;
        if ((rc = cml_h2d(c, c->ini_points.p, points, sizeof(cmlhip_init_point) * (size_t)n))) return rc;
        InitArgs A;
        A.img = py->lv[level].grad; A.w = py->lv[level].w; A.h = py->lv[level].h; A.n = n;
        A.P = *prm; A.alpha_opt = alphaOpt;
        A.pts = c->ini_points.as<cmlhip_init_point>(); A.partial = c->ini_partial.as<float>();
        if (c->lim.texel_format == CMLHIP_TEXEL_F16) k_init_calc_res_and_gs<true><<<nb, 256, 0, c->stream>>>(A);
        else k_init_calc_res_and_gs<false><<<nb, 256, 0, c->stream>>>(A);
        CML_CHECK(c, hipGetLastError());
        std::vector<float> part((size_t)nb * INI_NRED);
        cml_d2h_batch_begin(c);
        cml_d2h(c, part.data(), c->ini_partial.p, sizeof(float) * part.size());
        cml_d2h(c, points, c->ini_points.p, sizeof(cmlhip_init_point) * (size_t)n);
        if ((rc = cml_d2h_batch_flush(c))) return rc;
        for (int b = 0; b < nb; b++)
            for (int k = 0; k < INI_NRED; k++) sums[k] += (double)part[(size_t)b * INI_NRED + k];
    }
    float H9[2][81];
    for (int wch = 0; wch < 2; wch++) {
        int idx = 0;
        for (int r = 0; r < 9; r++)
            for (int cc = r; cc < 9; cc++) { H9[wch][r * 9 + cc] = H9[wch][cc * 9 + r] = (float)sums[45 * wch + idx]; idx++; }
    }
    for (int r = 0; r < 8; r++) {                                              // :729-732
        for (int cc = 0; cc < 8; cc++) { H_out[r * 8 + cc] = H9[0][r * 9 + cc]; H_out_sc[r * 8 + cc] = H9[1][r * 9 + cc]; }
        b_out[r] = H9[0][r * 9 + 8]; b_out_sc[r] = H9[1][r * 9 + 8];
    }
    for (int k = 0; k < 3; k++) {                                              // :736-742
        H_out[k * 8 + k] += alphaOpt * n;
        b_out[k] += prm->tlog[k] * alphaOpt * n;
    }
    res[0] = (float)sums[90]; res[1] = alphaEnergy; res[2] = (float)(2 * (size_t)n);   // E.num counts both point loops
    for (int k = 0; k < 64; k++) if (!std::isfinite(H_out[k]) || !std::isfinite(H_out_sc[k])) return CMLHIP_ERR_NONFINITE;
    return CMLHIP_OK;
}

}  // extern "C"
