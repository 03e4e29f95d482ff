// Do f32 MFMAs and VALU work of two different waves of one SIMD overlap?  (experiment tool)
// Block of 8 waves: waves 0-3 (one per SIMD) run an MFMA loop, waves 4-7 (their SIMD partners)
// run a VALU loop of the chosen kind (0 none, 1 v_fma_f32, 2 v_exp_f32, 3 v_add_u32 (int), 4 v_pk_fma_f32).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void probe(float *out, unsigned long long *cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    unsigned long long t0, t1;
    float res = 0;
    if (wave < 4) {
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
        float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
            }
        }
        t1 = __builtin_readcyclecounter();
        res = acc0[0] + acc1[0];
    } else {
        float x0 = threadIdx.x * 0.5f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
        int k0 = threadIdx.x, k1 = k0 + 1, k2 = k0 + 2, k3 = k0 + 3;
        f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x1, x0}, p3 = {x3, x2};
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        if (KIND != 0)
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 32; r++) {
                if (KIND == 1) { x0 = __builtin_fmaf(x0, 0.999f, 0.5f); x1 = __builtin_fmaf(x1, 0.999f, 0.5f); x2 = __builtin_fmaf(x2, 0.999f, 0.5f); x3 = __builtin_fmaf(x3, 0.999f, 0.5f); }
                if (KIND == 2) { x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3); }
                if (KIND == 3) { k0 = (k0 ^ k1) + 7; k1 = (k1 ^ k2) + 5; k2 = (k2 ^ k3) + 3; k3 = (k3 ^ k0) + 1; }
                if (KIND == 4) { p0 = p0 * p1 + p2; p1 = p1 * p2 + p3; p2 = p2 * p3 + p0; p3 = p3 * p0 + p1; }
            }
        }
        t1 = __builtin_readcyclecounter();
        res = x0 + x1 + x2 + x3 + k0 + k1 + k2 + k3 + p0[0] + p1[1] + p2[0] + p3[1];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
template <int KIND>
void run(const char *name) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
    int iters = 2000;
    hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("partner %-12s: MFMA wave %.1f cycles per MFMA (16 per iter); partner wave %.1f cycles per VALU instr (128 per iter), partner total %.0f vs mfma total %.0f\n",
           name, h[0] / (double)(iters * 16), h[4] / (double)(iters * 128), (double)h[4], (double)h[0]);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("none"); run<1>("v_fma_f32"); run<2>("v_exp_f32"); run<3>("int xor/add"); run<4>("v_pk_fma_f32");
    return 0;
}
