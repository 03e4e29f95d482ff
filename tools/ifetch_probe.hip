// Is instruction FETCH what holds the big unrolled kernels (k_ff_viterbi_teams, k_gru_proj: 12 waves per CU, 20-40 KB of straight-line
// code per block loop) at ~11 cycles per instruction when a wave can issue one per 4.1?  tools/trans_share_probe.hip answered the
// issue-rate question with 8-instruction loops on ONE CU; here the body is long straight-line code (nothing is re-used from a wave's
// instruction buffer), every CU runs it (two CUs share an instruction cache), and the encoding size is varied.
//
// One workgroup per CU (100 KB of LDS each), W waves per workgroup, every wave runs BODY independent VALU instructions x ITER.
// Prints cycles per instruction as seen by the fastest and the slowest wave of workgroup 0 and the mean over all workgroups, plus
// the instruction bytes per clock that rate means per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/ifetch_probe.hip -o build/ifetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define G8_ADD "v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3\n\tv_add_f32 %4, %4, %4\n\tv_add_f32 %5, %5, %5\n\tv_add_f32 %6, %6, %6\n\tv_add_f32 %7, %7, %7\n\t"
#define G8_FMA "v_fma_f32 %0, %0, %0, %1\n\tv_fma_f32 %1, %1, %1, %2\n\tv_fma_f32 %2, %2, %2, %3\n\tv_fma_f32 %3, %3, %3, %4\n\tv_fma_f32 %4, %4, %4, %5\n\tv_fma_f32 %5, %5, %5, %6\n\tv_fma_f32 %6, %6, %6, %7\n\tv_fma_f32 %7, %7, %7, %0\n\t"
// the decoder's mix: add, compare, sdwa select, max (4 + 4 + 8 + 4 bytes... v_max_f32 e32 is 4)
#define G8_MIX "v_add_f32 %0, %0, %1\n\tv_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32_sdwa %3, %3, %4, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\tv_max_f32 %2, %2, %0\n\t" \
               "v_add_f32 %5, %5, %1\n\tv_cmp_lt_f32 vcc, %5, %6\n\tv_cndmask_b32_sdwa %3, %3, %4, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_0\n\tv_max_f32 %6, %6, %5\n\t"

#define KERNEL(NAME, GROUP, REPT)                                                                                           \
__global__ __launch_bounds__(1024) void NAME(unsigned long long *out, float seed, int iters) {                             \
    extern __shared__ float lds[];                                                                                          \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    if (seed == 123.f) lds[threadIdx.x] = seed;                                                                              \
    __syncthreads();                                                                                                         \
    const unsigned long long t0 = __builtin_readcyclecounter();                                                              \
    for (int it = 0; it < iters; it++)                                                                                       \
        asm volatile(".rept " #REPT "\n\t" GROUP ".endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc"); \
    const unsigned long long t1 = __builtin_readcyclecounter();                                                              \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                       \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[0] = 0;                                                      \
}
KERNEL(k_add_8k, G8_ADD, 256)        /* 2048 x 4 B =  8 KB */
KERNEL(k_fma_16k, G8_FMA, 256)       /* 2048 x 8 B = 16 KB */
KERNEL(k_fma_48k, G8_FMA, 768)       /* 6144 x 8 B = 48 KB */
KERNEL(k_mix_10k, G8_MIX, 256)       /* 2048 instructions, 5 B on average = 10 KB */
KERNEL(k_fma_loop, G8_FMA, 1)        /* the 8-instruction loop of the older probe */

typedef void (*kfn)(unsigned long long *, float, int);
int main() {
    unsigned long long *out;
    hipMalloc(&out, 256 * 16 * 8);
    struct { const char *name; kfn f; int body; double bytes; } ks[] = {
        {"v_add_f32 e32, 8 KB straight-line", k_add_8k, 2048, 4.0}, {"v_fma_f32, 16 KB straight-line", k_fma_16k, 2048, 8.0},
        {"v_fma_f32, 48 KB straight-line", k_fma_48k, 6144, 8.0}, {"add / cmp / sdwa-select / max, 10 KB", k_mix_10k, 2048, 5.0},
        {"v_fma_f32, 8-instruction loop", k_fma_loop, 8, 8.0}};
    for (auto &k : ks) {
        hipFuncSetAttribute((const void *)k.f, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        for (int ncu : {1, 256})
            for (int W : {4, 8, 12, 16}) {
                const int iters = k.body >= 2048 ? (k.body > 4096 ? 16 : 48) : 12288;
                std::vector<unsigned long long> h(256 * 16);
                for (int rep = 0; rep < 2; rep++) {
                    hipLaunchKernelGGL(k.f, dim3(ncu), dim3(64 * W), 100 * 1024, 0, out, 1.0f, iters);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h.data(), out, 256 * 16 * 8, hipMemcpyDeviceToHost);
                const double n = (double)k.body * iters;
                double lo = 1e30, hi = 0, sum = 0;
                for (int g = 0; g < ncu; g++) for (int w = 0; w < W; w++) { const double c = h[g * 16 + w] / n; sum += c; if (g == 0) { lo = std::min(lo, c); hi = std::max(hi, c); } }
                const double mean = sum / (ncu * W);
                printf("%-38s  %3d CU x %2d waves: wg 0 fastest %5.2f slowest %5.2f, mean over all %5.2f cycles per instruction = %5.1f instruction bytes per clock and CU\n",
                       k.name, ncu, W, lo, hi, mean, W * k.bytes / mean);
            }
    }
    return 0;
}
