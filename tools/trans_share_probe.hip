// Is the transcendental rate of gfx950 a per-SIMD or a per-CU resource?  One workgroup of W waves (wave w -> SIMD w % 4), every wave runs the same
// stream of independent instructions; cycles per instruction seen by wave 0, for W = 1, 2, 4, 8.
//   hipcc --offload-arch=gfx950 -O3 tools/trans_share_probe.hip -o build/trans_share_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define PROBE(NAME, ASM)                                                                                   \
__global__ void NAME(unsigned long long *out, float seed) {                                              \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;   \
    __syncthreads();                                                                                       \
    unsigned long long t0 = __builtin_readcyclecounter();                                                  \
    for (int it = 0; it < 256; it++) {                                                                     \
        asm volatile(".rept 8\n\t" ASM "\n\t.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); \
    }                                                                                                      \
    unsigned long long t1 = __builtin_readcyclecounter();                                                  \
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                                          \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[0] = 0;                                    \
}
PROBE(k_add,  "v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3\n\tv_add_f32 %4, %4, %4\n\tv_add_f32 %5, %5, %5\n\tv_add_f32 %6, %6, %6\n\tv_add_f32 %7, %7, %7")
PROBE(k_fma,  "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %4\n\tv_fma_f32 %3, %3, %4, %5\n\tv_fma_f32 %4, %4, %5, %6\n\tv_fma_f32 %5, %5, %6, %7\n\tv_fma_f32 %6, %6, %7, %0\n\tv_fma_f32 %7, %7, %0, %1")
PROBE(k_exp,  "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7")
PROBE(k_rcp,  "v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\tv_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7")
PROBE(k_mixed, "v_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_rcp_f32 %2, %2\n\tv_add_f32 %3, %3, %3\n\tv_exp_f32 %4, %4\n\tv_add_f32 %5, %5, %5\n\tv_rcp_f32 %6, %6\n\tv_add_f32 %7, %7, %7")
int main() {
    unsigned long long *out; hipMalloc(&out, 1024 * 8);
    unsigned long long h[16];
    const double n = 256.0 * 8 * 8;
#define RUN(K, W) hipLaunchKernelGGL(K, dim3(1), dim3(64 * W), 0, 0, out, 1.0f); hipDeviceSynchronize(); hipMemcpy(h, out, 8 * W, hipMemcpyDeviceToHost); \
    { double mx = 0; for (int i = 0; i < W; i++) mx = h[i] > mx ? h[i] : mx; printf("%-8s %d waves in one workgroup: wave 0 %.2f, slowest %.2f cycles per instruction\n", #K, W, h[0] / n, mx / n); }
    RUN(k_add, 1) RUN(k_add, 1) RUN(k_add, 4) RUN(k_add, 8) RUN(k_add, 12) RUN(k_add, 16)
    RUN(k_fma, 1) RUN(k_fma, 8) RUN(k_fma, 16)
    RUN(k_exp, 1) RUN(k_exp, 2) RUN(k_exp, 4) RUN(k_exp, 8) RUN(k_exp, 12)
    RUN(k_rcp, 1) RUN(k_rcp, 4) RUN(k_rcp, 8)
    RUN(k_mixed, 1) RUN(k_mixed, 4) RUN(k_mixed, 8)
    return 0;
}
