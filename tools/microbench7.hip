// microbench7.hip — cost of the COLUMN UPDATES of the LDS-resident LDL^T (k_ba_solve): "row[j] -= cid * broadcast(row[k], lane j)" for the
// columns behind a pivot, one wave alone on its SIMD (tools only).
//   hipcc --offload-arch=gfx950 -O2 tools/microbench7.hip -o tools/mb7.bin && tools/mb7.bin
// The compiler pairs every v_readlane pair with its v_fma_f64 and reuses ONE scalar register pair for all of them (the kernel sits at
// the scalar-register limit); k_ba_solve measures 12 ns = 29 cycles per column update.  Variants, 14 independent updates per trip:
//   0 as the compiler emits it from the plain source (readlane, readlane, fma, ...)
//   1 the 14 broadcasts first (28 readlanes into 14 distinct scalar pairs, inline asm), then the 14 multiply-adds
//   2 groups of 4: 8 readlanes, 4 multiply-adds
//   3 the multiply-adds alone (operand already in a vector register): the floor
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int lane) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane); u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}
#define RLP(S, LANE) "v_readlane_b32 s" #S ", %[lo], " #LANE "\n\tv_readlane_b32 s" #S "+1, %[hi], " #LANE "\n\t"
template <int V> __global__ void k_cols(long long* out, double* sink, int n, double seed) {
    const int l = threadIdx.x;
    double row[16];
#pragma unroll
    for (int j = 0; j < 16; j++) row[j] = seed + 1e-3 * l + 1e-2 * j;
    double cid = 1e-7 * (1 + l);
    const long long t0 = clock64();
    for (int it = 0; it < n; it++) {
        if (V == 0) {
#pragma unroll
            for (int j = 2; j < 16; j++) row[j] -= cid * rl(row[0], j);
        } else if (V == 1) {
            union { double d; int i[2]; } u; u.d = row[0];
            asm volatile(
                "v_readlane_b32 s40, %[lo], 2\n\tv_readlane_b32 s41, %[hi], 2\n\t"
                "v_readlane_b32 s42, %[lo], 3\n\tv_readlane_b32 s43, %[hi], 3\n\t"
                "v_readlane_b32 s44, %[lo], 4\n\tv_readlane_b32 s45, %[hi], 4\n\t"
                "v_readlane_b32 s46, %[lo], 5\n\tv_readlane_b32 s47, %[hi], 5\n\t"
                "v_readlane_b32 s48, %[lo], 6\n\tv_readlane_b32 s49, %[hi], 6\n\t"
                "v_readlane_b32 s50, %[lo], 7\n\tv_readlane_b32 s51, %[hi], 7\n\t"
                "v_readlane_b32 s52, %[lo], 8\n\tv_readlane_b32 s53, %[hi], 8\n\t"
                "v_readlane_b32 s54, %[lo], 9\n\tv_readlane_b32 s55, %[hi], 9\n\t"
                "v_readlane_b32 s56, %[lo], 10\n\tv_readlane_b32 s57, %[hi], 10\n\t"
                "v_readlane_b32 s58, %[lo], 11\n\tv_readlane_b32 s59, %[hi], 11\n\t"
                "v_readlane_b32 s60, %[lo], 12\n\tv_readlane_b32 s61, %[hi], 12\n\t"
                "v_readlane_b32 s62, %[lo], 13\n\tv_readlane_b32 s63, %[hi], 13\n\t"
                "v_readlane_b32 s64, %[lo], 14\n\tv_readlane_b32 s65, %[hi], 14\n\t"
                "v_readlane_b32 s66, %[lo], 15\n\tv_readlane_b32 s67, %[hi], 15\n\t"
                "s_nop 1\n\t"
                "v_fma_f64 %[r2], -%[c], s[40:41], %[r2]\n\tv_fma_f64 %[r3], -%[c], s[42:43], %[r3]\n\t"
                "v_fma_f64 %[r4], -%[c], s[44:45], %[r4]\n\tv_fma_f64 %[r5], -%[c], s[46:47], %[r5]\n\t"
                "v_fma_f64 %[r6], -%[c], s[48:49], %[r6]\n\tv_fma_f64 %[r7], -%[c], s[50:51], %[r7]\n\t"
                "v_fma_f64 %[r8], -%[c], s[52:53], %[r8]\n\tv_fma_f64 %[r9], -%[c], s[54:55], %[r9]\n\t"
                "v_fma_f64 %[r10], -%[c], s[56:57], %[r10]\n\tv_fma_f64 %[r11], -%[c], s[58:59], %[r11]\n\t"
                "v_fma_f64 %[r12], -%[c], s[60:61], %[r12]\n\tv_fma_f64 %[r13], -%[c], s[62:63], %[r13]\n\t"
                "v_fma_f64 %[r14], -%[c], s[64:65], %[r14]\n\tv_fma_f64 %[r15], -%[c], s[66:67], %[r15]\n\t"
                : [r2] "+v"(row[2]), [r3] "+v"(row[3]), [r4] "+v"(row[4]), [r5] "+v"(row[5]), [r6] "+v"(row[6]), [r7] "+v"(row[7]), [r8] "+v"(row[8]),
                  [r9] "+v"(row[9]), [r10] "+v"(row[10]), [r11] "+v"(row[11]), [r12] "+v"(row[12]), [r13] "+v"(row[13]), [r14] "+v"(row[14]), [r15] "+v"(row[15])
                : [lo] "v"(u.i[0]), [hi] "v"(u.i[1]), [c] "v"(cid)
                : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59",
                  "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
        } else if (V == 2) {
            union { double d; int i[2]; } u; u.d = row[0];
#define GRP4(J0, J1, J2, J3, R0, R1, R2, R3) \
            asm volatile( \
                "v_readlane_b32 s40, %[lo], " #J0 "\n\tv_readlane_b32 s41, %[hi], " #J0 "\n\t" \
                "v_readlane_b32 s42, %[lo], " #J1 "\n\tv_readlane_b32 s43, %[hi], " #J1 "\n\t" \
                "v_readlane_b32 s44, %[lo], " #J2 "\n\tv_readlane_b32 s45, %[hi], " #J2 "\n\t" \
                "v_readlane_b32 s46, %[lo], " #J3 "\n\tv_readlane_b32 s47, %[hi], " #J3 "\n\t" \
                "s_nop 1\n\t" \
                "v_fma_f64 %[a], -%[c], s[40:41], %[a]\n\tv_fma_f64 %[b], -%[c], s[42:43], %[b]\n\t" \
                "v_fma_f64 %[d], -%[c], s[44:45], %[d]\n\tv_fma_f64 %[e], -%[c], s[46:47], %[e]\n\t" \
                : [a] "+v"(R0), [b] "+v"(R1), [d] "+v"(R2), [e] "+v"(R3) : [lo] "v"(u.i[0]), [hi] "v"(u.i[1]), [c] "v"(cid) \
                : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47")
            GRP4(2, 3, 4, 5, row[2], row[3], row[4], row[5]);
            GRP4(6, 7, 8, 9, row[6], row[7], row[8], row[9]);
            GRP4(10, 11, 12, 13, row[10], row[11], row[12], row[13]);
            GRP4(14, 15, 14, 15, row[14], row[15], row[1], row[0 + 1]);
        } else {
#pragma unroll
            for (int j = 2; j < 16; j++) row[j] -= cid * row[1];
        }
    }
    const long long t1 = clock64();
    if (l == 0) out[0] = t1 - t0;
    double s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) s += row[j];
    sink[l] = s;
}
int main() {
    long long* d; double* s; hipMalloc(&d, 64); hipMalloc(&s, 4096);
    const int n = 4000;
    const char* names[4] = {"compiler-paired readlane / fma", "14 broadcasts (distinct scalar pairs), then 14 fma", "groups of four", "fma alone (floor)"};
    for (int v = 0; v < 4; v++) {
        long long h = 0;
        for (int rep = 0; rep < 3; rep++) {
            if (v == 0) k_cols<0><<<1, 64>>>(d, s, n, 1.5); else if (v == 1) k_cols<1><<<1, 64>>>(d, s, n, 1.5);
            else if (v == 2) k_cols<2><<<1, 64>>>(d, s, n, 1.5); else k_cols<3><<<1, 64>>>(d, s, n, 1.5);
            hipDeviceSynchronize();
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        }
        printf("%-52s %7.1f clock64 ticks per column update (14 per trip)\n", names[v], (double)h / n / 14.0);
    }
    return 0;
}
