// microbench5.hip — what a kernel's ARGUMENTS cost before its first dependent load (tools only).
// A chain kernel<<<1, 64>>>: read p[0], then q[p[0]], store — launched back to back in one stream (as the kernels of the BA iteration
// are); the dispatch's own duration (hipExtLaunchKernelGGL events) tells how long the arguments + two dependent trips take.
// Variants: pointers as plain leading arguments | behind a 600-byte by-value struct (the BAArgs shape), each compiled without and with
//   -mllvm -amdgpu-kernarg-preload-count=16  (the command processor writes the first 16 argument dwords into SGPRs at dispatch).
//   hipcc --offload-arch=gfx950 -O2 [-mllvm -amdgpu-kernarg-preload-count=16] tools/microbench5.hip -o tools/mb5[p].bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
struct Big { int a[150]; const int* p; const double* q; double* out; };
__global__ void k_front(const int* __restrict__ p, const double* __restrict__ q, double* __restrict__ out, Big b) {
    const int i = p[threadIdx.x & 1];
    out[threadIdx.x] = q[i + threadIdx.x] + b.a[7];
}
__global__ void k_struct(Big b) {
    const int i = b.p[threadIdx.x & 1];
    b.out[threadIdx.x] = b.q[i + threadIdx.x] + b.a[7];
}
__global__ void k_touch(double* out) { out[threadIdx.x + 64] = 1.0; }
int main() {
    int* p; double *q, *out;
    hipMalloc(&p, 64); hipMalloc(&q, 1 << 20); hipMalloc(&out, 4096);
    hipMemset(p, 0, 64); hipMemset(q, 0, 1 << 20);
    Big b = {}; b.p = p; b.q = q; b.out = out;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t s; hipStreamCreate(&s);
    for (int variant = 0; variant < 2; variant++) {
        std::vector<float> us;
        for (int it = 0; it < 300; it++) {
            hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, s, out);          // a predecessor in the stream, as in the iteration
            if (variant == 0) hipExtLaunchKernelGGL(k_front, dim3(1), dim3(64), 0, s, e0, e1, 0, (const int*)p, (const double*)q, out, b);
            else hipExtLaunchKernelGGL(k_struct, dim3(1), dim3(64), 0, s, e0, e1, 0, b);
            hipStreamSynchronize(s);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 50) us.push_back(ms * 1e3f);
        }
        std::sort(us.begin(), us.end());
        printf("%s: dispatch duration min %.2f med %.2f p90 %.2f us\n", variant == 0 ? "pointers as leading arguments " : "pointers inside a 600-B struct", us.front(), us[us.size() / 2], us[us.size() * 9 / 10]);
    }
    return 0;
}
