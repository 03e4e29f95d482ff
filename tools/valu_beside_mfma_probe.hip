// What a VALU instruction costs one wave while ANOTHER wave of the same SIMD streams v_mfma_f32_32x32x16_f16 back to back
// (the situation of k_gru_proj32's chain waves beside its G waves), and what the MFMA stream loses.
// Workgroup of 8 waves = 2 per SIMD (wave w and w + 4 share SIMD w % 4): waves 0-3 run the VALU body, waves 4-7 the MFMA
// stream (mode bit 0) or nothing; mode bit 1: the VALU waves at s_setprio 2.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_beside_mfma_probe.hip -o build/valu_beside_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define BODY_MUL(i) "v_mul_f32 %" #i ", %" #i ", %" #i "\n\t"
#define BODY_EXP(i) "v_exp_f32 %" #i ", %" #i "\n\t"
#define BODY_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n\t"
#define BODY_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %" #i ", vcc\n\t"
#define BODY_MIX(i) "v_fma_mixlo_f16 %" #i ", %" #i ", %" #i ", 0\n\t"
#define BODY_CVT(i) "v_cvt_pk_f16_f32 %" #i ", %" #i ", %" #i "\n\t"
#define BODY_FMA(i) "v_fma_f32 %" #i ", %" #i ", %" #i ", %" #i "\n\t"

#define PROBE(NAME, ASM)                                                                                          \
__global__ __launch_bounds__(512) void NAME(unsigned long long *out, float seed, int mode) {                      \
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                                            \
    if (wave < 4) {                                                                                               \
        if (mode & 2) __builtin_amdgcn_s_setprio(2);                                                              \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        __syncthreads();                                                                                          \
        unsigned long long t0 = __builtin_readcyclecounter();                                                     \
        for (int it = 0; it < 256; it++)                                                                          \
            asm volatile(".rept 8\n\t" ASM ".endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); \
        unsigned long long t1 = __builtin_readcyclecounter();                                                     \
        if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;                                                         \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[100] = 0;                                     \
    } else {                                                                                                      \
        f16x8 a, b;                                                                                               \
        for (int k = 0; k < 8; k++) { a[k] = (_Float16)(seed + k); b[k] = (_Float16)(seed - k); }                 \
        f32x16 c0 = {}, c1 = {};                                                                                  \
        __syncthreads();                                                                                          \
        unsigned long long t0 = __builtin_readcyclecounter();                                                     \
        if (mode & 1)                                                                                             \
            for (int it = 0; it < 600; it++) {   /* 2400 MFMAs x 32 cycles: longer than the VALU waves' loop */    \
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);                                    \
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);                                    \
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);                                    \
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);                                    \
            }                                                                                                     \
        unsigned long long t1 = __builtin_readcyclecounter();                                                     \
        if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;                                                         \
        if (c0[0] + c1[3] == 12345.f) out[100] = 0;                                                               \
    }                                                                                                             \
}
PROBE(k_mul, R8(BODY_MUL))
PROBE(k_fma, R8(BODY_FMA))
PROBE(k_exp, R8(BODY_EXP))
PROBE(k_rcp, R8(BODY_RCP))
PROBE(k_cnd, R8(BODY_CND))
PROBE(k_mix, R8(BODY_MIX))
PROBE(k_cvt, R8(BODY_CVT))
// packed f32 multiply on register pairs: needs 64-bit operands
__global__ __launch_bounds__(512) void k_pk(unsigned long long *out, float seed, int mode) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        if (mode & 2) __builtin_amdgcn_s_setprio(2);
        f32x2 a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
        __syncthreads();
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < 256; it++)
            asm volatile(".rept 8\n\tv_pk_mul_f32 %0, %0, %0\n\tv_pk_mul_f32 %1, %1, %1\n\tv_pk_mul_f32 %2, %2, %2\n\tv_pk_mul_f32 %3, %3, %3\n\t"
                         "v_pk_mul_f32 %4, %4, %4\n\tv_pk_mul_f32 %5, %5, %5\n\tv_pk_mul_f32 %6, %6, %6\n\tv_pk_mul_f32 %7, %7, %7\n\t.endr"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        unsigned long long t1 = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
        if (a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a6[0] + a7[1] == 12345.f) out[100] = 0;
    } else {
        f16x8 a, b;
        for (int k = 0; k < 8; k++) { a[k] = (_Float16)(seed + k); b[k] = (_Float16)(seed - k); }
        f32x16 c0 = {}, c1 = {};
        __syncthreads();
        unsigned long long t0 = __builtin_readcyclecounter();
        if (mode & 1)
            for (int it = 0; it < 600; it++) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            }
        unsigned long long t1 = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
        if (c0[0] + c1[3] == 12345.f) out[100] = 0;
    }
}
// the chain waves' logistic(acc) * h on 16 values exactly as hipcc emits it (8 v_pk_mul, 16 v_exp, 8 v_pk_add, 16 v_rcp, 8 v_pk_mul)
// beside: kind 0 nothing, 1 two independent accumulators + fixed A/B (as above), 2 ONE accumulator (dependent chain), 3 one accumulator and
// 6 rotating A operands + 2 B operands (the G waves' stream: weights from 48 registers), 4 as 3 with an LDS read per MFMA pair
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k_logistic(unsigned long long *out, float seed, int kind, int prio) {
    __shared__ f16x8 lds[64 * 8];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lds[threadIdx.x & 511] = (f16x8){};
    if (wave < 4) {
        if (prio) __builtin_amdgcn_s_setprio(2);
        f32x2 a[8], h[8];
        for (int i = 0; i < 8; i++) { a[i] = (f32x2){seed + i, seed - i}; h[i] = (f32x2){0.5f, 0.25f}; }
        const f32x2 sc = {-1.44f / 16384.f, -1.44f / 16384.f};
        __syncthreads();
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < 256; it++) {
            f32x2 t[8];
#pragma unroll
            for (int i = 0; i < 8; i++) t[i] = a[i] * sc;
#pragma unroll
            for (int i = 0; i < 8; i++) { t[i][0] = __builtin_amdgcn_exp2f(t[i][0]); t[i][1] = __builtin_amdgcn_exp2f(t[i][1]); }
#pragma unroll
            for (int i = 0; i < 8; i++) t[i] = t[i] + 1.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) { t[i][0] = __builtin_amdgcn_rcpf(t[i][0]); t[i][1] = __builtin_amdgcn_rcpf(t[i][1]); }
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = t[i] * h[i];
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("" : "+v"(a[i]));
        }
        unsigned long long t1 = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
        float s = 0; for (int i = 0; i < 8; i++) s += a[i][0] + a[i][1];
        if (s == 12345.f) out[100] = 0;
    } else {
        f16x8 a[6], b[2];
        for (int j = 0; j < 6; j++) for (int k = 0; k < 8; k++) a[j][k] = (_Float16)(seed + k + j);
        for (int j = 0; j < 2; j++) for (int k = 0; k < 8; k++) b[j][k] = (_Float16)(seed - k + j);
        f32x16 c0 = {}, c1 = {};
        __syncthreads();
        unsigned long long t0 = __builtin_readcyclecounter();
        if (kind == 1)
            for (int it = 0; it < 600; it++) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c1, 0, 0, 0);
            }
        else if (kind == 2)
            for (int it = 0; it < 600; it++) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c0, 0, 0, 0);
            }
        else if (kind == 3)
            for (int it = 0; it < 400; it++) {
#pragma unroll
                for (int j = 0; j < 6; j++) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[j & 1], c0, 0, 0, 0);
            }
        else if (kind == 4)
            for (int it = 0; it < 400; it++) {
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    if (!(j & 1)) b[(j >> 1) & 1] = lds[(threadIdx.x & 63) + 64 * (j + (it & 1))];
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[j & 1], c0, 0, 0, 0);
                }
            }
        unsigned long long t1 = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
        if (c0[0] + c1[3] == 12345.f) out[100] = 0;
    }
}
int main() {
    unsigned long long *out; hipMalloc(&out, 1024 * 8);
    unsigned long long h[8];
    const double n = 256.0 * 8 * 8;
    printf("cycles per VALU instruction of one wave (8 independent per group); MFMA wave: cycles per v_mfma_f32_32x32x16_f16 (2400 issued)\n");
    printf("%-8s %10s %22s %22s %26s\n", "", "alone", "beside MFMA stream", "... at s_setprio 2", "MFMA cycles each (beside)");
#define RUN(K) { double r[3]; double mf = 0; for (int mode = 0; mode < 3; mode++) { int m = mode == 0 ? 0 : mode == 1 ? 1 : 3; \
        hipLaunchKernelGGL(K, dim3(1), dim3(512), 0, 0, out, 1.0f, m); hipDeviceSynchronize(); hipMemcpy(h, out, 64, hipMemcpyDeviceToHost); \
        r[mode] = h[1] / n; if (mode == 1) mf = h[5] / 2400.0; } \
        printf("%-8s %10.2f %22.2f %22.2f %26.2f\n", #K, r[0], r[1], r[2], mf); }
    RUN(k_mul) RUN(k_mul) RUN(k_fma) RUN(k_pk) RUN(k_exp) RUN(k_rcp) RUN(k_cnd) RUN(k_mix) RUN(k_cvt)
    printf("\nlogistic(acc) * h on 16 values (56 VALU instructions: 24 packed f32, 32 transcendental), cycles per pass of one wave; MFMA wave: cycles per MFMA\n");
    for (int kind = 0; kind < 5; kind++) for (int prio = 0; prio < 3; prio += 2) {
        hipLaunchKernelGGL(k_logistic, dim3(1), dim3(512), 0, 0, out, 1.0f, kind, prio); hipDeviceSynchronize(); hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        static const char *nm[5] = {"alone", "2 accumulators, fixed A/B", "1 accumulator (dependent chain)", "1 accumulator, 6 A x 2 B rotating", "... + ds_read_b128 per pair"};
        printf("kind %d (%-34s) prio %d: %7.1f cycles per pass; MFMA %6.2f\n", kind, nm[kind], prio, h[1] / 256.0, kind ? h[5] / 2400.0 : 0.0);
    }
    return 0;
}
