// Issue rate of a few VALU instructions on gfx950: one wave per SIMD, N independent instructions in a loop,
// cycles per instruction from s_memtime.   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o build/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define STR(x) #x
#define XSTR(x) STR(x)
#define PROBE(NAME, ASM)                                                                                   \
__global__ void NAME(unsigned long long *out, float seed) {                                              \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;   \
    unsigned long long t0 = __builtin_readcyclecounter();                                                  \
    for (int it = 0; it < 256; it++) {                                                                     \
        asm volatile(".rept 8\n\t" ASM "\n\t.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); \
    }                                                                                                      \
    unsigned long long t1 = __builtin_readcyclecounter();                                                  \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                       \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[0] = 0;                                    \
}
// each ASM body: 8 independent instructions (one per register)
PROBE(k_add,  "v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3\n\tv_add_f32 %4, %4, %4\n\tv_add_f32 %5, %5, %5\n\tv_add_f32 %6, %6, %6\n\tv_add_f32 %7, %7, %7")
PROBE(k_exp,  "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7")
PROBE(k_cvt16, "v_cvt_f16_f32 %0, %0\n\tv_cvt_f16_f32 %1, %1\n\tv_cvt_f16_f32 %2, %2\n\tv_cvt_f16_f32 %3, %3\n\tv_cvt_f16_f32 %4, %4\n\tv_cvt_f16_f32 %5, %5\n\tv_cvt_f16_f32 %6, %6\n\tv_cvt_f16_f32 %7, %7")
PROBE(k_cvt32, "v_cvt_f32_f16 %0, %0\n\tv_cvt_f32_f16 %1, %1\n\tv_cvt_f32_f16 %2, %2\n\tv_cvt_f32_f16 %3, %3\n\tv_cvt_f32_f16 %4, %4\n\tv_cvt_f32_f16 %5, %5\n\tv_cvt_f32_f16 %6, %6\n\tv_cvt_f32_f16 %7, %7")
PROBE(k_cvtpk, "v_cvt_pk_f16_f32 %0, %0, %1\n\tv_cvt_pk_f16_f32 %1, %1, %2\n\tv_cvt_pk_f16_f32 %2, %2, %3\n\tv_cvt_pk_f16_f32 %3, %3, %4\n\tv_cvt_pk_f16_f32 %4, %4, %5\n\tv_cvt_pk_f16_f32 %5, %5, %6\n\tv_cvt_pk_f16_f32 %6, %6, %7\n\tv_cvt_pk_f16_f32 %7, %7, %0")
PROBE(k_mix,  "v_fma_mix_f32 %0, %0, %1, %2\n\tv_fma_mix_f32 %1, %1, %2, %3\n\tv_fma_mix_f32 %2, %2, %3, %4\n\tv_fma_mix_f32 %3, %3, %4, %5\n\tv_fma_mix_f32 %4, %4, %5, %6\n\tv_fma_mix_f32 %5, %5, %6, %7\n\tv_fma_mix_f32 %6, %6, %7, %0\n\tv_fma_mix_f32 %7, %7, %0, %1")
PROBE(k_sdwa, "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\tv_cndmask_b32_sdwa %1, %1, %2, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\tv_cndmask_b32_sdwa %2, %2, %3, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\tv_cndmask_b32_sdwa %3, %3, %4, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\tv_cndmask_b32_sdwa %4, %4, %5, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\tv_cndmask_b32_sdwa %5, %5, %6, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\tv_cndmask_b32_sdwa %6, %6, %7, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\tv_cndmask_b32_sdwa %7, %7, %0, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0")
PROBE(k_cmp,  "v_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 vcc, %1, %2\n\tv_cmp_lt_f32 vcc, %2, %3\n\tv_cmp_lt_f32 vcc, %3, %4\n\tv_cmp_lt_f32 vcc, %4, %5\n\tv_cmp_lt_f32 vcc, %5, %6\n\tv_cmp_lt_f32 vcc, %6, %7\n\tv_cmp_lt_f32 vcc, %7, %0")
PROBE(k_log,  "v_log_f32 %0, %0\n\tv_log_f32 %1, %1\n\tv_log_f32 %2, %2\n\tv_log_f32 %3, %3\n\tv_log_f32 %4, %4\n\tv_log_f32 %5, %5\n\tv_log_f32 %6, %6\n\tv_log_f32 %7, %7")
// dependent chains: every instruction needs the previous one's result
PROBE(k_dep_add, "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1")
PROBE(k_dep_exp, "v_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %0, %0")
PROBE(k_dep_cut, "v_mul_f32 %1, 0x42800000, %0\n\tv_cvt_pk_f16_f32 %2, %1, %1\n\tv_cvt_f32_f16 %3, %2\n\tv_sub_f32 %4, %1, %3\n\tv_cvt_pk_f16_f32 %5, %4, %4\n\tv_add_f32 %0, %0, %5\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %1")
PROBE(k_dep_cmp, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cmp_lt_f32 vcc, %0, %3\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cmp_lt_f32 vcc, %0, %4\n\tv_cndmask_b32 %0, %0, %4, vcc")
int main() {
    unsigned long long *out; hipMalloc(&out, 1024 * 8);
    unsigned long long h[4];
    const double n = 256.0 * 8 * 8;
#define RUN(K) hipLaunchKernelGGL(K, dim3(4), dim3(64), 0, 0, out, 1.0f); hipDeviceSynchronize(); hipMemcpy(h, out, 32, hipMemcpyDeviceToHost); printf("%-10s %.2f cycles per instruction (one wave alone on its SIMD)\n", #K, h[1] / n);
    RUN(k_add) RUN(k_add) RUN(k_exp) RUN(k_log) RUN(k_cvt16) RUN(k_cvt32) RUN(k_cvtpk) RUN(k_mix) RUN(k_sdwa) RUN(k_cmp) RUN(k_dep_add) RUN(k_dep_exp) RUN(k_dep_cut) RUN(k_dep_cmp)
    return 0;
}
