// microbench3.hip — VALU issue cost on gfx950 (tools only): cycles per wave64 instruction on one SIMD for the instruction kinds the
// residual kernel is made of, with 1, 2 and 3 waves per SIMD, as a dependent chain and as 8 independent chains.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench3.hip -o /tmp/mb3 && /tmp/mb3
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define OUTER 64

#define KERNEL(name, decl, body_dep, body_ind, sink)                                                        \
    __global__ void name##_dep(long long* out, double seed) {                                               \
        decl;                                                                                               \
        long long t0 = clock64();                                                                           \
        for (int o = 0; o < OUTER; o++) {                                                                   \
            _Pragma("unroll") for (int i = 0; i < REP; i++) { body_dep; }                                   \
        }                                                                                                   \
        long long t1 = clock64();                                                                           \
        if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;                       \
        if (sink == 12345.678) out[100] = 1;                                                                \
    }                                                                                                       \
    __global__ void name##_ind(long long* out, double seed) {                                               \
        decl;                                                                                               \
        long long t0 = clock64();                                                                           \
        for (int o = 0; o < OUTER; o++) {                                                                   \
            _Pragma("unroll") for (int i = 0; i < REP / 8; i++) { body_ind; }                               \
        }                                                                                                   \
        long long t1 = clock64();                                                                           \
        if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;                       \
        if (sink == 12345.678) out[100] = 1;                                                                \
    }

#define D8 double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, m = 1.0000001, c = 0.5
#define F8 float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, m = 1.0000001f, c = 0.5f
#define ASM1(op, x) asm volatile(op " %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c))
#define ASM8(op) ASM1(op, a0); ASM1(op, a1); ASM1(op, a2); ASM1(op, a3); ASM1(op, a4); ASM1(op, a5); ASM1(op, a6); ASM1(op, a7)
#define ASM2_1(op, x) asm volatile(op " %0, %0, %1" : "+v"(x) : "v"(m))
#define ASM2_8(op) ASM2_1(op, a0); ASM2_1(op, a1); ASM2_1(op, a2); ASM2_1(op, a3); ASM2_1(op, a4); ASM2_1(op, a5); ASM2_1(op, a6); ASM2_1(op, a7)
#define ASMU_1(op, x) asm volatile(op " %0, %0" : "+v"(x))
#define ASMU_8(op) ASMU_1(op, a0); ASMU_1(op, a1); ASMU_1(op, a2); ASMU_1(op, a3); ASMU_1(op, a4); ASMU_1(op, a5); ASMU_1(op, a6); ASMU_1(op, a7)

KERNEL(fma64, D8, ASM1("v_fma_f64", a0), ASM8("v_fma_f64"), (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(mul64, D8, ASM2_1("v_mul_f64", a0), ASM2_8("v_mul_f64"), (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(add64, D8, ASM2_1("v_add_f64", a0), ASM2_8("v_add_f64"), (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(rcp64, D8, ASMU_1("v_rcp_f64", a0), ASMU_8("v_rcp_f64"), (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(fma32, F8, ASM1("v_fma_f32", a0), ASM8("v_fma_f32"), (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(mul32, F8, ASM2_1("v_mul_f32", a0), ASM2_8("v_mul_f32"), (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(sqrt32, F8, ASMU_1("v_sqrt_f32", a0), ASMU_8("v_sqrt_f32"), (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(rcp32, F8, ASMU_1("v_rcp_f32", a0), ASMU_8("v_rcp_f32"), (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))

// conversions: f32 -> f64 -> f32 round trip (2 instructions per step)
#define CVT1(x) { double t_; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(t_) : "v"(x)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x) : "v"(t_)); }
#define CVT8 CVT1(a0) CVT1(a1) CVT1(a2) CVT1(a3) CVT1(a4) CVT1(a5) CVT1(a6) CVT1(a7)
KERNEL(cvt, F8, CVT1(a0), CVT8, (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
// the sum form of the residual kernel: acc = (float)((double)acc + x*y), x,y already double: cvt, fma(or mul+add), cvt
#define SUM1(x) { double t_; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(t_) : "v"(x)); asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p_) : "v"(md), "v"(cd)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(t_) : "v"(p_)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x) : "v"(t_)); }
#define SUM8 SUM1(a0) SUM1(a1) SUM1(a2) SUM1(a3) SUM1(a4) SUM1(a5) SUM1(a6) SUM1(a7)
KERNEL(sumform, F8; double md = seed; double cd = seed + 0.25; double p_, SUM1(a0), SUM8, (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
// packed fp32
typedef float float2_ __attribute__((ext_vector_type(2)));
#define PK1(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(pm), "v"(pc))
#define PKD float2_ a0; a0.x = (float)seed; a0.y = 1.f; float2_ a1 = a0 + 1.f; float2_ a2 = a0 + 2.f; float2_ a3 = a0 + 3.f; float2_ a4 = a0 + 4.f; float2_ a5 = a0 + 5.f; float2_ a6 = a0 + 6.f; float2_ a7 = a0 + 7.f; float2_ pm; pm.x = 1.0000001f; pm.y = 1.f; float2_ pc; pc.x = 0.5f; pc.y = 0.25f
#define PK8 PK1(a0); PK1(a1); PK1(a2); PK1(a3); PK1(a4); PK1(a5); PK1(a6); PK1(a7)
KERNEL(pkfma32, PKD, PK1(a0), PK8, (double)(a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y))
// LDS read (ds_read_b32), dependent address chain vs independent
__global__ void lds_ind(long long* out, double seed) {
    __shared__ float s[64 * 41];
    for (int i = threadIdx.x; i < 64 * 41; i += blockDim.x) s[i] = i;
    __syncthreads();
    const float* S = s + (threadIdx.x % 64) * 41;
    float acc = 0;
    long long t0 = clock64();
    for (int o = 0; o < OUTER; o++) {
#pragma unroll
        for (int i = 0; i < 32; i++) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)S), "n"(0)); acc += v; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    long long t1 = clock64();
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
    if (acc == 12345.678f) out[100] = 1;
}


// matrix instructions: one dependent chain / four independent chains per wave
typedef double mb_d4 __attribute__((ext_vector_type(4)));
typedef float mb_f4 __attribute__((ext_vector_type(4)));
template <int CH> __global__ void mfma64(long long* out, double seed) {
    mb_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = seed + threadIdx.x * 1e-3, b = 1.0 - seed * 1e-3;
    long long t0 = clock64();
    for (int o = 0; o < OUTER; o++) {
#pragma unroll
        for (int i = 0; i < REP / 4; i++) {
#pragma unroll
            for (int c = 0; c < 4; c++) acc[CH == 1 ? 0 : c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[CH == 1 ? 0 : c], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678) out[100] = 1;
}
template <int CH> __global__ void mfma32(long long* out, double seed) {
    mb_f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = (float)seed + threadIdx.x * 1e-3f, b = 1.0f - (float)seed * 1e-3f;
    long long t0 = clock64();
    for (int o = 0; o < OUTER; o++) {
#pragma unroll
        for (int i = 0; i < REP / 4; i++) {
#pragma unroll
            for (int c = 0; c < 4; c++) acc[CH == 1 ? 0 : c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[CH == 1 ? 0 : c], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) out[100] = 1;
}

template <class K> static void run(const char* name, K kd, K ki, int nper, long long* d_out, int ops_dep, int ops_ind) {
    long long h[16];
    printf("%-10s", name);
    for (int variant = 0; variant < 2; variant++) {
        for (int w = 1; w <= 3; w++) {
            K k = variant ? ki : kd;
            for (int it = 0; it < 3; it++) { hipLaunchKernelGGL(k, dim3(1), dim3(256 * w), 0, 0, d_out, 1.0); hipDeviceSynchronize(); }
            hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
            long long mx = 0; for (int i = 0; i < 4 * w; i++) if (h[i] > mx) mx = h[i];
            const double n = (double)(variant ? ops_ind : ops_dep) * w;    // instructions issued on one SIMD
            printf("  %s w=%d: %6.2f", variant ? "ind" : "dep", w, mx / n);
        }
    }
    printf("   (cycles per instruction per SIMD)\n");
}

int main() {
    long long* d_out; hipMalloc(&d_out, 4096);
    const int N = REP * OUTER;
#define RUN(name, mult) run(#name, name##_dep, name##_ind, 0, d_out, N * mult, N * mult)
    run("mfma_f64", mfma64<1>, mfma64<4>, 0, d_out, N, N);      // v_mfma_f64_16x16x4_f64: "dep" = one accumulator chain, "ind" = four chains
    run("mfma_f32", mfma32<1>, mfma32<4>, 0, d_out, N, N);      // v_mfma_f32_16x16x4_f32
    RUN(fma64, 1); RUN(mul64, 1); RUN(add64, 1); RUN(rcp64, 1); RUN(fma32, 1); RUN(mul32, 1); RUN(sqrt32, 1); RUN(rcp32, 1); RUN(cvt, 2); RUN(sumform, 4); RUN(pkfma32, 1);
    {
        long long h[16];
        printf("%-10s", "ds_read");
        for (int w = 1; w <= 3; w++) {
            for (int it = 0; it < 3; it++) { hipLaunchKernelGGL(lds_ind, dim3(1), dim3(256 * w), 0, 0, d_out, 1.0); hipDeviceSynchronize(); }
            hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
            long long mx = 0; for (int i = 0; i < 4 * w; i++) if (h[i] > mx) mx = h[i];
            printf("  w=%d: %6.2f", w, mx / (double)(32 * OUTER * w));
        }
        printf("   (cycles per ds_read_b32 + v_add per SIMD, stride-41 rows)\n");
    }
    return 0;
}
