// Which launch pattern keeps a thread of the HIP runtime busy while the caller sleeps: process CPU time (all threads) over wall time for 2 s of
//   A  one stream, one ~25 ms kernel per step                         B  one stream, 50 kernels of ~0.5 ms per step
//   C  B + a second stream that waits for an event of the first and runs a kernel (hipStreamWaitEvent both ways)
//   D  B + 15 hipMemsetAsync per step                                 E  B + a timing event recorded per kernel (hipEventRecord on default events)
//   F  B + a kernel that writes pinned host memory                    G  B + hipMemcpyAsync D2H of 4 KB per step
// The caller waits with hipEventQuery + nanosleep, so what is left is the runtime's own.   hipcc --offload-arch=gfx950 -O2 tools/helper_thread_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <ctime>
__global__ void k_spin(long long cycles, int *out) { const long long t0 = clock64(); while (clock64() - t0 < cycles) {} if (out && threadIdx.x == 0) *out = 1; }
static double proc_cpu() { timespec t; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double wall() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void nap_until(hipEvent_t e) { timespec ts = {0, 50000}; while (hipEventQuery(e) == hipErrorNotReady) nanosleep(&ts, nullptr); }
int main() {
    hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t done, x1, x2, tev[64];
    hipEventCreateWithFlags(&done, hipEventDisableTiming); hipEventCreateWithFlags(&x1, hipEventDisableTiming); hipEventCreateWithFlags(&x2, hipEventDisableTiming);
    for (auto &e : tev) hipEventCreate(&e);
    int *d, *h; hipMalloc(&d, 1 << 20); hipHostMalloc(&h, 1 << 20, hipHostMallocDefault);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 1000, nullptr); hipStreamSynchronize(s);
    double t0 = wall(); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 5000000LL, nullptr); hipStreamSynchronize(s); const double per = (wall() - t0) / 5000000.0;
    const long long c25 = (long long)(0.025 / per), c05 = (long long)(0.0005 / per);
    const char *names[7] = {"A one 25 ms kernel per step", "B 50 kernels of 0.5 ms", "C B + second stream, events both ways", "D B + 15 hipMemsetAsync", "E B + a timing event per kernel", "F B + kernel writing pinned host memory", "G B + 4 KB hipMemcpyAsync D2H"};
    for (int mode = 0; mode < 7; mode++) {
        const double w0 = wall(), p0 = proc_cpu();
        int steps = 0;
        while (wall() - w0 < 2.0) {
            if (mode == 0) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, c25, nullptr);
            else for (int k = 0; k < 50; k++) {
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, c05, mode == 5 && k == 49 ? h : nullptr);
                if (mode == 4) hipEventRecord(tev[k], s);
                if (mode == 2 && k == 10) { hipEventRecord(x1, s); hipStreamWaitEvent(s2, x1, 0); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s2, c05, nullptr); hipEventRecord(x2, s2); }
                if (mode == 2 && k == 40) hipStreamWaitEvent(s, x2, 0);
            }
            if (mode == 3) for (int k = 0; k < 15; k++) hipMemsetAsync(d + 1024 * k, 0, 4096, s);
            if (mode == 6) hipMemcpyAsync(h, d, 4096, hipMemcpyDeviceToHost, s);
            hipEventRecord(done, s);
            nap_until(done);
            steps++;
        }
        const double w = wall() - w0;
        printf("%-42s %3d steps of %.1f ms: process %.2f CPUs\n", names[mode], steps, 1e3 * w / steps, (proc_cpu() - p0) / w);
    }
    return 0;
}
