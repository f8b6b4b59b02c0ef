// microbench2.hip — first-touch (TLB) cost per kernel: independent vs dependent loads across separately allocated buffers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Ptrs { const int* p[16]; };
__global__ void k_indep(Ptrs P, int k, int* out) {
    int s = 0;
    for (int i = 0; i < k; i++) s += P.p[i][threadIdx.x];
    if (s == -1) *out = s;
}
__global__ void k_dep(Ptrs P, int k, int* out) {
    int idx = threadIdx.x;
    for (int i = 0; i < k; i++) idx = P.p[i][idx];       // buffers hold identity -> idx stays, but the chain is dependent
    if (idx == -1) *out = idx;
}
__global__ void k_store(int* p) { p[threadIdx.x] = 1; }
static float timeit(hipStream_t st, int reps, void (*f)(hipStream_t)) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 20; i++) f(st);
    (void)hipEventRecord(a, st);
    for (int i = 0; i < reps; i++) f(st);
    (void)hipEventRecord(b, st); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / reps;
}
static Ptrs g_sep, g_arena; static int* d_out; static int g_k;
int main() {
    hipStream_t st; (void)hipStreamCreate(&st);
    std::vector<int> id(1024);
    for (int i = 0; i < 1024; i++) id[i] = i;
    for (int i = 0; i < 16; i++) { int* p; (void)hipMalloc(&p, 4096); (void)hipMemcpy(p, id.data(), 4096, hipMemcpyHostToDevice); g_sep.p[i] = p; }
    int* arena; (void)hipMalloc(&arena, 16 * 4096);
    for (int i = 0; i < 16; i++) { (void)hipMemcpy(arena + i * 1024, id.data(), 4096, hipMemcpyHostToDevice); g_arena.p[i] = arena + i * 1024; }
    (void)hipMalloc(&d_out, 64);
    for (int k : {1, 2, 4, 8, 16}) {
        g_k = k;
        float a = timeit(st, 300, [](hipStream_t s) { k_indep<<<1, 64, 0, s>>>(g_sep, g_k, d_out); });
        float b = timeit(st, 300, [](hipStream_t s) { k_dep<<<1, 64, 0, s>>>(g_sep, g_k, d_out); });
        float c = timeit(st, 300, [](hipStream_t s) { k_indep<<<1, 64, 0, s>>>(g_arena, g_k, d_out); });
        float d = timeit(st, 300, [](hipStream_t s) { k_dep<<<1, 64, 0, s>>>(g_arena, g_k, d_out); });
        printf("k=%2d  separate buffers: indep %.2f us  dep %.2f us   | one arena: indep %.2f us  dep %.2f us\n", k, a, b, c, d);
    }
    printf("store-only kernel: %.2f us\n", timeit(st, 300, [](hipStream_t s) { k_store<<<1, 64, 0, s>>>(d_out); }));
    return 0;
}
