// MFMA f32 16x16x4 issue-rate probe (experiment tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ void probe(float *out, unsigned long long *cyc, int iters) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int CHAINS>
void run(int waves) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
    int iters = 2000;
    hipLaunchKernelGGL(probe<CHAINS>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, cyc, 8 * waves, hipMemcpyDeviceToHost);
    printf("chains %d waves/CU %d: ", CHAINS, waves);
    for (int w = 0; w < waves; w++) printf("%.1f ", h[w] / (double)(iters * 6 * CHAINS));
    printf(" cycles per MFMA per wave\n");
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1>(1); run<2>(1); run<4>(1);
    run<2>(4); run<2>(6); run<2>(8); run<4>(8); run<2>(12);
    return 0;
}
