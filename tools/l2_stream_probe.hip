// Can every CU re-read the same 400 KB (the S1 weight matrix as fp16 pieces) from L2 once per decoder block?
// 256+ workgroups x 512 threads; each wave reads its 1/8 of the buffer per iteration with 16-byte loads,
// DEPTH loads in flight.  Prints microseconds per iteration (= per block step of a fused S1 + decoder kernel)
// and the aggregate rate.    hipcc --offload-arch=gfx950 -O3 tools/l2_stream_probe.hip -o build/l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int DEPTH>
__global__ __launch_bounds__(512, 2) void k_stream(const u32x4 *w, int n16_per_wave, int iters, unsigned *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 *p = w + (size_t)wave * n16_per_wave;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        for (int j = 0; j < n16_per_wave / 64; j += DEPTH) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; d++) v[d] = __builtin_nontemporal_load(p + (size_t)(j + d) * 64 + lane) ;
#pragma unroll
            for (int d = 0; d < DEPTH; d++) acc ^= v[d];
        }
        asm volatile("" : "+v"(acc));
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}
template <int DEPTH>
__global__ __launch_bounds__(512, 2) void k_stream_plain(const u32x4 *w, int n16_per_wave, int iters, unsigned *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 *p = w + (size_t)wave * n16_per_wave;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        for (int j = 0; j < n16_per_wave / 64; j += DEPTH) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; d++) v[d] = p[(size_t)(j + d) * 64 + lane];
#pragma unroll
            for (int d = 0; d < DEPTH; d++) acc ^= v[d];
        }
        asm volatile("" : "+v"(acc) :: "memory");
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}
int main() {
    const size_t bytes = 64 * 3 * 2 * 1024;      // 64 m-tiles x 3 k steps x 2 pieces x 1 KiB = 393216
    const int n16_per_wave = (int)(bytes / 16 / 8);
    u32x4 *w; unsigned *out;
    hipMalloc(&w, bytes); hipMemset(w, 1, bytes); hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    for (int nwg : {256, 512}) {
        for (int variant = 0; variant < 4; variant++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(a);
                switch (variant) {
                case 0: hipLaunchKernelGGL(k_stream_plain<6>, dim3(nwg), dim3(512), 0, 0, w, n16_per_wave, iters, out); break;
                case 1: hipLaunchKernelGGL(k_stream_plain<12>, dim3(nwg), dim3(512), 0, 0, w, n16_per_wave, iters, out); break;
                case 2: hipLaunchKernelGGL(k_stream_plain<24>, dim3(nwg), dim3(512), 0, 0, w, n16_per_wave, iters, out); break;
                default: hipLaunchKernelGGL(k_stream<12>, dim3(nwg), dim3(512), 0, 0, w, n16_per_wave, iters, out); break;
                }
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep) printf("wg %d variant %d: %.2f us per 384 KiB pass per workgroup (%d per CU), aggregate %.1f TB/s\n", nwg, variant,
                                1e3 * ms / iters, nwg / 256, (double)bytes * iters * nwg / (ms * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
