// can the recurrent layers' gate inputs (18 KB per tile and block) travel from a projection kernel to a
// co-resident chain kernel through a ring in memory, W blocks deep, WITHOUT costing the 18 GB of HBM traffic per layer that
// made the two-kernel form of round 1 slower?
//
// Producer kernel: workgroup g (384 threads = the projection team's six waves) writes, per step, the gate inputs of its two
// tile slots (2 x 18 KB) into slot (step mod W) of its ring and publishes the step count.  Consumer kernel (same grid, same
// block, another stream): workgroup g waits for step s, reads the 2 x 18 KB, publishes how far it has read; the producer never
// runs more than W steps ahead.  977 steps = one layer of the bench workload (10 000 reads x 800 blocks on 512 tile slots):
// 9.2 GB written + 9.2 GB read.  A third stream meanwhile copies as the layer itself does (3 GB in, 3 GB out per "layer").
// Reported: time per layer-equivalent, achieved ring GB/s, where the two workgroups of a pair ran (XCC_ID), and -- under
// rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes -- what left the L2 (MI355X_MICROARCH.md: Infinity-Cache hits are
// counted too, so the counters separate "stayed in L2" from "went to the fabric", not MALL from HBM; the time against HBM's
// ~6 TB/s is what tells those apart).
//
// Two ways of making the data visible:
//   mode 0  agent-scope release / acquire fences as the compiler emits them (buffer_wbl2 sc1 / buffer_inv sc1: correct wherever
//           the two workgroups run)
//   mode 1  stores and loads with the sc1 bit (coherent at the L2) + workgroup-scope fences: correct only while producer g and
//           consumer g share an XCD (reported), which is what a same-grid co-resident pair gets from the round-robin dispatcher
// Every wait is bounded by wall time (2 s): a stuck pair raises `abort` and everybody leaves.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mall_ring_probe.hip -o build/mall_ring_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NTH 384
#define LANE_FLOATS 4608            /* 288 gate inputs x 16 reads */
#define SLOT_FLOATS (2 * LANE_FLOATS) /* two tile slots per workgroup */
#define PASSES (SLOT_FLOATS / 4 / NTH) /* 16-byte accesses per thread and step: 6 */

__device__ __forceinline__ unsigned long long wall() { return wall_clock64(); }
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }

__device__ __forceinline__ bool wait_ge(const unsigned *p, unsigned need, unsigned *abort_) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
    const unsigned long long t0 = wall();
    for (;;) {
        __builtin_amdgcn_s_sleep(2);
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
        if (__hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        if (wall() - t0 > 200000000ull) { __hip_atomic_store(abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    }
}

template <int MODE>
__global__ __launch_bounds__(NTH) void k_producer(float *ring, unsigned *prod, const unsigned *cons, unsigned *abort_, unsigned *where, int W, int steps) {
    const int g = blockIdx.x, tid = threadIdx.x;
    float *mine = ring + (size_t)g * W * SLOT_FLOATS;
    if (tid == 0) where[g] = xcc_id();
    __shared__ int ok;
    for (int s = 0; s < steps; s++) {
        if (tid == 0) ok = (s < W) ? 1 : (int)wait_ge(cons + g, (unsigned)(s - W + 1), abort_);
        __syncthreads();
        if (!ok) return;
        float *slot = mine + (size_t)(s % W) * SLOT_FLOATS;
        const float v = (float)(s + 1);
#pragma unroll
        for (int p = 0; p < PASSES; p++) {
            f32x4 x = {v, v + (float)p, (float)g, (float)tid};
            f32x4 *d = (f32x4 *)slot + p * NTH + tid;
            if (MODE == 0) *d = x;
            else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(d), "v"(x) : "memory");
        }
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(prod + g, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int MODE>
__global__ __launch_bounds__(NTH) void k_consumer(const float *ring, const unsigned *prod, unsigned *cons, unsigned *abort_, unsigned *where, float *sink,
                                                  unsigned *bad, int W, int steps) {
    const int g = blockIdx.x, tid = threadIdx.x;
    const float *mine = ring + (size_t)g * W * SLOT_FLOATS;
    if (tid == 0) where[g] = xcc_id();
    __shared__ int ok;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned nbad = 0;
    for (int s = 0; s < steps; s++) {
        if (tid == 0) ok = (int)wait_ge(prod + g, (unsigned)(s + 1), abort_);
        __syncthreads();
        if (!ok) return;
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const float *slot = mine + (size_t)(s % W) * SLOT_FLOATS;
        f32x4 x[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; p++) {
            const f32x4 *d = (const f32x4 *)slot + p * NTH + tid;
            if (MODE == 0) x[p] = *d;
            else asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x[p]) : "v"(d) : "memory");
        }
        if (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < PASSES; p++) { acc += x[p]; nbad += (x[p][0] != (float)(s + 1)); }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(cons + g, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (nbad) atomicAdd(bad, nbad);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = 1.f;
}

// the layer's own traffic beside it: read n float4, write n float4
__global__ __launch_bounds__(256) void k_copy(const f32x4 *in, f32x4 *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

int main(int argc, char **argv) {
    const int G = 256, steps = argc > 1 ? atoi(argv[1]) : 977;
    const size_t copy_bytes = (size_t)3072 << 20;            // 3 GB in + 3 GB out = one layer's activations
    hipStream_t sp, sc, sb;
    CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned *prod, *cons, *abort_, *where, *bad; float *sink;
    CK(hipMalloc(&prod, G * 4)); CK(hipMalloc(&cons, G * 4)); CK(hipMalloc(&abort_, 4)); CK(hipMalloc(&where, 2 * G * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&sink, 4));
    f32x4 *cin, *cout;
    CK(hipMalloc(&cin, copy_bytes)); CK(hipMalloc(&cout, copy_bytes)); CK(hipMemset(cin, 1, copy_bytes));
    hipEvent_t e0, e1, b0, b1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    const double ring_bytes = (double)G * SLOT_FLOATS * 4 * steps;     // one direction
    printf("# steps %d, %d workgroups x 2 tile slots x 18 KB: %.2f GB written + %.2f GB read per layer-equivalent\n", steps, G, ring_bytes / 1e9, ring_bytes / 1e9);
    printf("# mode 0 = agent-scope fences, 1 = sc1 accesses (L2-coherent, same-XCD pairs only); bg = a 3 GB -> 3 GB copy running beside it\n");
    // mall_ring_probe [steps [mode W bg]]: one configuration only (the rocprofv3 --pmc passes)
    const int only_mode = argc > 4 ? atoi(argv[2]) : -1, only_W = argc > 4 ? atoi(argv[3]) : -1, only_bg = argc > 4 ? atoi(argv[4]) : -1;
    for (int mode = 0; mode < 2; mode++)
        for (int W : {2, 4, 8, 16})
            for (int bg = 0; bg < 2; bg++) {
                if (only_mode >= 0 && (mode != only_mode || W != only_W || bg != only_bg)) continue;
                float *ring;
                const size_t rb = (size_t)G * W * SLOT_FLOATS * 4;
                CK(hipMalloc(&ring, rb)); CK(hipMemset(ring, 0, rb));
                float best = 1e30f, bgms = 0.f; unsigned hbad = 0, habort = 0; int same = 0;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipMemset(prod, 0, G * 4)); CK(hipMemset(cons, 0, G * 4)); CK(hipMemset(abort_, 0, 4)); CK(hipMemset(bad, 0, 4));
                    CK(hipDeviceSynchronize());
                    if (bg) { CK(hipEventRecord(b0, sb)); for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, sb, cin, cout, copy_bytes / 16); CK(hipEventRecord(b1, sb)); }
                    CK(hipEventRecord(e0, sc));
                    if (mode == 0) {
                        hipLaunchKernelGGL(k_consumer<0>, dim3(G), dim3(NTH), 0, sc, ring, prod, cons, abort_, where + G, sink, bad, W, steps);
                        hipLaunchKernelGGL(k_producer<0>, dim3(G), dim3(NTH), 0, sp, ring, prod, cons, abort_, where, W, steps);
                    } else {
                        hipLaunchKernelGGL(k_consumer<1>, dim3(G), dim3(NTH), 0, sc, ring, prod, cons, abort_, where + G, sink, bad, W, steps);
                        hipLaunchKernelGGL(k_producer<1>, dim3(G), dim3(NTH), 0, sp, ring, prod, cons, abort_, where, W, steps);
                    }
                    CK(hipEventRecord(e1, sc));
                    CK(hipDeviceSynchronize());
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (bg) CK(hipEventElapsedTime(&bgms, b0, b1));
                    CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&habort, abort_, 4, hipMemcpyDeviceToHost));
                    std::vector<unsigned> w(2 * G); CK(hipMemcpy(w.data(), where, 2 * G * 4, hipMemcpyDeviceToHost));
                    same = 0; for (int g = 0; g < G; g++) same += (w[g] == w[G + g]);
                    if (ms < best) best = ms;
                    if (habort) break;
                }
                printf("mode %d  W %2d (ring %6.1f MB)  bg %d :  %7.3f ms per layer-equivalent = %6.2f TB/s written + read%s   pairs on one XCD %d / %d   stale reads %u%s\n",
                       mode, W, rb / 1e6, bg, best, 2 * ring_bytes / (best * 1e-3) / 1e12,
                       bg ? ({ static char b[64]; snprintf(b, 64, "   [copy: %.2f ms for 12 GB = %.2f TB/s]", bgms, 4.0 * copy_bytes / (bgms * 1e-3) / 1e12); b; }) : "",
                       same, G, hbad, habort ? "   ABORTED (a wait timed out)" : "");
                fflush(stdout);
                CK(hipFree(ring));
                if (habort) { printf("a pair was not co-resident or deadlocked: stopping\n"); return 1; }
            }
    return 0;
}
