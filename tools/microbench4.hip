// microbench4.hip — shader clock under load (tools only): every workgroup runs a chain of matrix (or vector fp64) instructions and
// reports the shader-clock cycles (s_memtime) per 10-ns tick of the constant 100-MHz counter (s_memrealtime), for 1 ... 1024 workgroups.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench4.hip -o tools/mb4.bin && tools/mb4.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int KIND> __global__ void k_load(long long* out, int n, double seed) {
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = seed + threadIdx.x * 1e-3, b = 1.0 - seed * 1e-3, x0 = a, x1 = b, x2 = a + b, x3 = a - b;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
        if (KIND == 0) {
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) { x0 = __builtin_fma(x0, 1.0000001, 0.5); x1 = __builtin_fma(x1, 1.0000001, 0.5); x2 = __builtin_fma(x2, 1.0000001, 0.5); x3 = __builtin_fma(x3, 1.0000001, 0.5); }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + x0 + x1 + x2 + x3 == 12345.678) out[0] = 1;
}
int main() {
    long long* d; hipMalloc(&d, 2 * 2048 * sizeof(long long));
    std::vector<long long> h(2 * 2048);
    for (int kind = 0; kind < 2; kind++) {
        for (int wgs : {1, 32, 168, 256, 512, 1024}) {
            for (int rep = 0; rep < 3; rep++) {
                if (kind == 0) hipLaunchKernelGGL(k_load<0>, dim3(wgs), dim3(512), 0, 0, d, 4000, 1.0);
                else hipLaunchKernelGGL(k_load<1>, dim3(wgs), dim3(512), 0, 0, d, 4000, 1.0);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), d, 2 * wgs * sizeof(long long), hipMemcpyDeviceToHost);
            std::vector<double> mhz, us;
            for (int i = 0; i < wgs; i++) { mhz.push_back(h[2 * i] / (h[2 * i + 1] * 0.01)); us.push_back(h[2 * i + 1] * 0.01); }
            std::sort(mhz.begin(), mhz.end()); std::sort(us.begin(), us.end());
            printf("%s  %4d workgroups x 512 threads: shader clock min %.0f med %.0f max %.0f MHz ; workgroup duration med %.1f us\n",
                   kind == 0 ? "v_mfma_f64_16x16x4 (4 chains)" : "v_fma_f64 (4 chains)         ", wgs, mhz.front(), mhz[wgs / 2], mhz.back(), us[wgs / 2]);
        }
    }
    return 0;
}
