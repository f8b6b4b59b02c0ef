// microbench.hip — calibration of the launch / latency floor on the GPU box (tools only, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_ __attribute__((ext_vector_type(4)));
__global__ void k_empty() {}
__global__ void k_chase(const int* __restrict__ next, int steps, int* out) {
    int i = threadIdx.x;
    for (int s = 0; s < steps; s++) i = next[i];
    if (i == -1) *out = i;
}
__global__ void k_clock(long long* out, int spin) {
    long long t0 = wall_clock64(), c0 = clock64();
    double x = threadIdx.x;
    for (int i = 0; i < spin; i++) x = x * 1.0000001 + 0.5;
    long long t1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = c1 - c0; out[2] = (long long)x; }
}
__global__ void k_mfma64(double* out, int n) {
    double4_ acc = {0, 0, 0, 0};
    double a = threadIdx.x * 0.001, b = 1.0;
    for (int i = 0; i < n; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ void k_barriers(int n, int* out) {
    __shared__ int s[64];
    for (int i = 0; i < n; i++) { if (threadIdx.x < 64) s[threadIdx.x] = i; __syncthreads(); }
    if (threadIdx.x == 0) *out = s[3];
}
static float timeit(hipStream_t st, int reps, void (*f)(hipStream_t)) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) f(st);
    hipEventRecord(a, st);
    for (int i = 0; i < reps; i++) f(st);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / reps;
}
static int* d_next; static int* d_out; static double* d_dbl; static long long* d_ll;
int main() {
    hipStream_t st; hipStreamCreate(&st);
    const int NN = 1 << 22;
    std::vector<int> h(NN);
    for (int i = 0; i < NN; i++) h[i] = (int)(((long long)i * 1048573 + 12345) % NN);
    hipMalloc(&d_next, NN * 4); hipMemcpy(d_next, h.data(), NN * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_out, 64); hipMalloc(&d_dbl, 4096); hipMalloc(&d_ll, 64);
    printf("empty kernel back-to-back      : %.2f us\n", timeit(st, 500, [](hipStream_t s) { k_empty<<<1, 64, 0, s>>>(); }));
    printf("empty 256 blocks               : %.2f us\n", timeit(st, 500, [](hipStream_t s) { k_empty<<<256, 256, 0, s>>>(); }));
    float t1 = timeit(st, 200, [](hipStream_t s) { k_chase<<<1, 64, 0, s>>>(d_next, 10, d_out); });
    float t2 = timeit(st, 200, [](hipStream_t s) { k_chase<<<1, 64, 0, s>>>(d_next, 110, d_out); });
    printf("dependent global load (16MB tbl): %.3f us per hop (kernel(10)=%.2f, kernel(110)=%.2f)\n", (t2 - t1) / 100, t1, t2);
    float m1 = timeit(st, 200, [](hipStream_t s) { k_mfma64<<<1, 64, 0, s>>>(d_dbl, 100); });
    float m2 = timeit(st, 200, [](hipStream_t s) { k_mfma64<<<1, 64, 0, s>>>(d_dbl, 1100); });
    printf("dependent f64 mfma 16x16x4      : %.1f ns each\n", (m2 - m1));
    float b1 = timeit(st, 200, [](hipStream_t s) { k_barriers<<<1, 512, 0, s>>>(100, d_out); });
    float b2 = timeit(st, 200, [](hipStream_t s) { k_barriers<<<1, 512, 0, s>>>(1100, d_out); });
    printf("__syncthreads (512 thr) + lds wr: %.1f ns each\n", (b2 - b1));
    for (int rep = 0; rep < 3; rep++) {
        k_clock<<<1, 64, 0, st>>>(d_ll, 200000);
        long long r[3]; hipMemcpy(r, d_ll, 24, hipMemcpyDeviceToHost);
        printf("wall_clock ticks %lld  shader cycles %lld  -> shader clock = %.0f MHz if wall_clock is 100 MHz\n", r[0], r[1], 100.0 * r[1] / r[0]);
    }
    return 0;
}
