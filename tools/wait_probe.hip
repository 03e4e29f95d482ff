// How a host thread waits for the GPU on this runtime: CPU time of the waiting thread (and of the whole process) while a ~50 ms kernel runs, for
//   hipEventSynchronize on a default event / on a hipEventBlockingSync event, hipStreamSynchronize, hipEventQuery + nanosleep polling,
//   and the same after hipSetDeviceFlags(hipDeviceScheduleBlockingSync) (argv[1] = "flag").     hipcc --offload-arch=gfx950 -O2 tools/wait_probe.hip -o wait_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <sys/resource.h>
__global__ void k_spin(long long cycles, int *out) { const long long t0 = clock64(); while (clock64() - t0 < cycles) {} if (out && threadIdx.x == 1024) *out = 1; }
static double thr_cpu() { timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double proc_cpu() { timespec t; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double wall() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "flag")) printf("hipSetDeviceFlags(hipDeviceScheduleBlockingSync): %s\n", hipGetErrorString(hipSetDeviceFlags(hipDeviceScheduleBlockingSync)));
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e_def, e_blk;
    hipEventCreateWithFlags(&e_def, hipEventDisableTiming);
    hipEventCreateWithFlags(&e_blk, hipEventDisableTiming | hipEventBlockingSync);
    const long long cyc = 100000000LL;      // 100 MHz timer: ~1 s?  calibrated below
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 1000, nullptr); hipStreamSynchronize(s);
    double t0 = wall(); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 5000000LL, nullptr); hipStreamSynchronize(s); const double per = (wall() - t0) / 5000000.0;
    const long long c50 = (long long)(0.05 / per);
    (void)cyc;
    for (int mode = 0; mode < 4; mode++) {
        double w = 0, tc = 0, pc = 0;
        for (int rep = 0; rep < 10; rep++) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, c50, nullptr);
            if (mode == 0) hipEventRecord(e_def, s);
            if (mode == 1 || mode == 3) hipEventRecord(e_blk, s);
            const double w0 = wall(), t0c = thr_cpu(), p0 = proc_cpu();
            if (mode == 0) hipEventSynchronize(e_def);
            else if (mode == 1) hipEventSynchronize(e_blk);
            else if (mode == 2) hipStreamSynchronize(s);
            else { timespec ts = {0, 200000}; while (hipEventQuery(e_blk) == hipErrorNotReady) nanosleep(&ts, nullptr); }
            w += wall() - w0; tc += thr_cpu() - t0c; pc += proc_cpu() - p0;
        }
        const char *names[4] = {"hipEventSynchronize, default event      ", "hipEventSynchronize, blocking-sync event", "hipStreamSynchronize                    ", "hipEventQuery + nanosleep(200 us)       "};
        printf("%s  waited %.1f ms per launch: waiting thread %.2f CPUs, whole process %.2f CPUs\n", names[mode], 1e2 * w, tc / w, pc / w);
    }
    return 0;
}
