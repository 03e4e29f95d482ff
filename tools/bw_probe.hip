// HBM bandwidth probe: write-only, read-only, copy (experiment tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_write(f32x4 *p, size_t n) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_write_nt(f32x4 *p, size_t n) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, p + i);
}
__global__ void k_read(const f32x4 *p, size_t n, float *out) {
    f32x4 a = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a[0] + a[1] + a[2] + a[3] == 12345.f) out[0] = 1;
}
__global__ void k_copy(const f32x4 *s, f32x4 *d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
// write pattern of the S1 kernel: each wave writes 1 KB runs, 4160 B... (tile rows) strided by mtiles*1KB
__global__ void k_write_tiles(f32x4 *p, size_t ntile) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t t = wave; t < ntile; t += nw) p[t * 64 + lane] = v;
}
int main() {
    const size_t bytes = 8ull << 30, n = bytes / 16;
    f32x4 *a, *b; float *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, double gb, auto fn) {
        fn(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 5; i++) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-28s %7.3f ms  %7.1f GB/s\n", name, ms, gb / (ms * 1e-3));
    };
    for (int grid : {1024, 4096, 16384}) {
        printf("grid %d x 256\n", grid);
        timeit("write", bytes / 1e9, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, a, n); });
        timeit("write nontemporal", bytes / 1e9, [&] { hipLaunchKernelGGL(k_write_nt, dim3(grid), dim3(256), 0, 0, a, n); });
        timeit("write 1KB tiles per wave", bytes / 1e9, [&] { hipLaunchKernelGGL(k_write_tiles, dim3(grid), dim3(256), 0, 0, a, n / 64); });
        timeit("read", bytes / 1e9, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, o); });
        timeit("copy (r+w bytes)", 2 * bytes / 1e9, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
    }
    timeit("hipMemsetAsync", bytes / 1e9, [&] { hipMemsetAsync(a, 0, bytes, 0); });
    return 0;
}
