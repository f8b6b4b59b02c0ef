// microbench6.hip — cycles of the dependent PIVOT CHAIN of the LDS-resident LDL^T (k_ba_solve), one wave alone on its SIMD (tools only).
//   hipcc --offload-arch=gfx950 -O2 tools/microbench6.hip -o tools/mb6.bin && tools/mb6.bin
// Variants of one step "broadcast d = x[lane k] -> 1/d (seed + folded Newton) -> multiplier -> update of the next column -> next broadcast":
//   0 as in k_ba_solve (v_readlane pair, zero-pivot guard on the scalar side, v_rcp_f64, 3 dependent fma, update)
//   1 the same without the zero-pivot guard (no SALU between the readlanes and the reciprocal)
//   2 readlane -> fma -> readlane only (the backward-substitution step)
//   3 pure VALU: rcp + 3 fma + fma, no cross-lane traffic
//   4 broadcast through ds_bpermute_b32 instead of v_readlane (stays in VGPRs)
//   5 broadcast through an LDS write + broadcast read
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int lane) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane); u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}
__device__ __forceinline__ double bperm(double v, int lane) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_ds_bpermute(lane * 4, u.i[0]); u.i[1] = __builtin_amdgcn_ds_bpermute(lane * 4, u.i[1]);
    return u.d;
}
template <int V> __global__ void k_chain(long long* out, double* sink, int n, double seed) {
    __shared__ double sh[64];
    const int l = threadIdx.x;
    double a = seed + 1e-3 * l, b = 0.25 + 1e-4 * l, c = 1.0 + 1e-5 * l;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (V == 0 || V == 1) {
                union { double d; int i[2]; } ud, us; ud.d = a;
                const int lo = __builtin_amdgcn_readlane(ud.i[0], k), hi = __builtin_amdgcn_readlane(ud.i[1], k);
                if (V == 0) { const bool tiny = (hi & 0x7ff00000) == 0; us.i[0] = tiny ? 0 : lo; us.i[1] = tiny ? 0x7e700000 : hi; }
                else { us.i[0] = lo; us.i[1] = hi; }
                const double x0 = __builtin_amdgcn_rcp(us.d);
                const double e0 = __builtin_fma(-us.d, x0, 1.0), p = a * x0;
                const double t = __builtin_fma(p, e0, p), e2 = e0 * e0;
                const double cid = __builtin_fma(t, e2, t);
                b = __builtin_fma(-cid, rl(a, (k + 1) & 15), b);      // update of the next column
                a = b + c;                                              // (the next pivot column depends on it)
            } else if (V == 2) {
                a = __builtin_fma(-b, rl(a, k), a);
            } else if (V == 3) {
                const double x0 = __builtin_amdgcn_rcp(a);
                const double e0 = __builtin_fma(-a, x0, 1.0), p = a * x0;
                const double t = __builtin_fma(p, e0, p), e2 = e0 * e0;
                const double cid = __builtin_fma(t, e2, t);
                b = __builtin_fma(-cid, c, b);
                a = b + c;
            } else if (V == 4) {
                const double d = bperm(a, k);
                const double x0 = __builtin_amdgcn_rcp(d);
                const double e0 = __builtin_fma(-d, x0, 1.0), p = a * x0;
                const double t = __builtin_fma(p, e0, p), e2 = e0 * e0;
                const double cid = __builtin_fma(t, e2, t);
                b = __builtin_fma(-cid, bperm(a, (k + 1) & 15), b);
                a = b + c;
            } else {
                sh[l] = a;
                const double d = sh[k];
                const double x0 = __builtin_amdgcn_rcp(d);
                const double e0 = __builtin_fma(-d, x0, 1.0), p = a * x0;
                const double t = __builtin_fma(p, e0, p), e2 = e0 * e0;
                const double cid = __builtin_fma(t, e2, t);
                b = __builtin_fma(-cid, sh[(k + 1) & 15], b);
                a = b + c;
            }
        }
    }
    const long long t1 = clock64();
    if (l == 0) { out[0] = t1 - t0; out[1] = wall_clock64() - w0; }
    sink[l] = a + b;
}
int main() {
    long long* d; double* s; hipMalloc(&d, 64); hipMalloc(&s, 4096);
    const int n = 2000;
    const char* names[6] = {"as in k_ba_solve (readlane, scalar guard, rcp, 3 fma, update)", "no zero-pivot guard", "readlane -> fma only (backward substitution step)",
                            "pure VALU (no cross-lane traffic)", "broadcast by ds_bpermute", "broadcast through LDS write + read"};
    for (int v = 0; v < 6; v++) {
        long long h = 0, hw[2] = {0, 0};
        for (int rep = 0; rep < 3; rep++) {
            switch (v) {
                case 0: hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(64), 0, 0, d, s, n, 3.0); break;
                case 1: hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, d, s, n, 3.0); break;
                case 2: hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(64), 0, 0, d, s, n, 3.0); break;
                case 3: hipLaunchKernelGGL(k_chain<3>, dim3(1), dim3(64), 0, 0, d, s, n, 3.0); break;
                case 4: hipLaunchKernelGGL(k_chain<4>, dim3(1), dim3(64), 0, 0, d, s, n, 3.0); break;
                default: hipLaunchKernelGGL(k_chain<5>, dim3(1), dim3(64), 0, 0, d, s, n, 3.0); break;
            }
            hipDeviceSynchronize();
        }
        hipMemcpy(hw, d, 16, hipMemcpyDeviceToHost); h = hw[0];
        printf("variant %d  %-66s %7.1f cycles per step  (shader clock during the run: %.0f MHz, %.1f ns per step)\n", v, names[v], (double)h / (16.0 * n), hw[0] / (hw[1] * 0.01), hw[1] * 10.0 / (16.0 * n));
    }
    return 0;
}
