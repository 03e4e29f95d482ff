// occupancy / residency probe for the GRU kernel (experiment tool, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../scrappie_amd/csrc/sh_kernels.h"

__global__ __launch_bounds__(384) void census(unsigned long long *t0, unsigned long long *t1, unsigned *hwid, int spin) {
    __shared__ float pad[3072];
    if (threadIdx.x == 0) {
        t0[blockIdx.x] = wall_clock64();
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        hwid[blockIdx.x] = (id & 0xffff) | (xcc << 16);
    }
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(100);
    __syncthreads();
    if (threadIdx.x == 0) t1[blockIdx.x] = wall_clock64() + (unsigned long long)pad[5];
}

int main() {
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_gru<6>, 384, 0);
    printf("k_gru<6> occupancy API: %d blocks/CU\n", nb);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void *)k_gru<6>);
    printf("k_gru<6>: numRegs %d sharedSizeBytes %zu localSizeBytes %zu maxThreadsPerBlock %d\n", fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes, fa.maxThreadsPerBlock);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, census, 384, 0);
    printf("census occupancy API: %d blocks/CU\n", nb);
    const int N = 512;
    unsigned long long *t0, *t1; unsigned *hw;
    hipMalloc(&t0, N * 8); hipMalloc(&t1, N * 8); hipMalloc(&hw, N * 4);
    hipLaunchKernelGGL(census, dim3(N), dim3(384), 0, 0, t0, t1, hw, 200);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h0(N), h1(N); std::vector<unsigned> hh(N);
    hipMemcpy(h0.data(), t0, N * 8, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), t1, N * 8, hipMemcpyDeviceToHost); hipMemcpy(hh.data(), hw, N * 4, hipMemcpyDeviceToHost);
    unsigned long long mn = ~0ull, mx = 0; for (int i = 0; i < N; i++) { mn = std::min(mn, h0[i]); mx = std::max(mx, h1[i]); }
    int late = 0; unsigned long long first_end = ~0ull; for (int i = 0; i < N; i++) first_end = std::min(first_end, h1[i]);
    for (int i = 0; i < N; i++) if (h0[i] >= first_end) late++;
    printf("census: %d blocks, span %.1f us (100MHz ticks), blocks that started after the first finished: %d\n", N, (mx - mn) / 100.0, late);
    // count distinct (xcc, se, cu) ids
    std::vector<unsigned> ids(hh); std::sort(ids.begin(), ids.end()); int distinct = std::unique(ids.begin(), ids.end()) - ids.begin();
    printf("distinct hw ids (xcc|hw_id low16): %d\n", distinct);
    return 0;
}
