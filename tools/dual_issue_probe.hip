// Which VALU instructions can a gfx950 SIMD issue for TWO waves at once?
// tools/trans_share_probe.hip (round 4) found "a SIMD sustains two streams of v_add_f32 / v_fma_f32 at 4.1 cycles each" and
// tools/ifetch_probe.hip (round 6) that a mix of add / compare / sdwa-select / max does NOT: two waves of that mix on one SIMD take
// turns (8.0 cycles per instruction for the younger).  This probe classifies instructions one by one.
//
// One workgroup on one CU; wave w runs on SIMD w % 4.  Part 1: waves 0-7 (two per SIMD) all run instruction X: cycles per instruction
// of the fastest and the slowest wave -- 4.0 / 4.0 = both waves of a SIMD issue at full rate (a "dual" instruction), 4.0 / 8.0 = they
// take turns.  Part 2: waves 0-3 run X, waves 4-7 run v_add_f32 (dual): does a single-rate instruction of one wave leave room for
// a dual-rate one of the other?
//   hipcc --offload-arch=gfx950 -O3 tools/dual_issue_probe.hip -o build/dual_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define R8(a) a a a a a a a a
#define BODY(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                                          \
__device__ __forceinline__ void NAME(float &a0, float &a1, float &a2, float &a3, float &a4, float &a5, float &a6, float &a7, int iters) { \
    for (int it = 0; it < iters; it++)                                                                                       \
        asm volatile(".rept 32\n\t" I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t" I4 "\n\t" I5 "\n\t" I6 "\n\t" I7 "\n\t.endr"    \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");           \
}
#define OP2(NAME, OP) BODY(NAME, OP " %0, %0, %1", OP " %1, %1, %2", OP " %2, %2, %3", OP " %3, %3, %4", OP " %4, %4, %5", OP " %5, %5, %6", OP " %6, %6, %7", OP " %7, %7, %0")
#define OP1(NAME, OP) BODY(NAME, OP " %0, %0", OP " %1, %1", OP " %2, %2", OP " %3, %3", OP " %4, %4", OP " %5, %5", OP " %6, %6", OP " %7, %7")
#define OP3(NAME, OP) BODY(NAME, OP " %0, %0, %1, %2", OP " %1, %1, %2, %3", OP " %2, %2, %3, %4", OP " %3, %3, %4, %5", OP " %4, %4, %5, %6", OP " %5, %5, %6, %7", OP " %6, %6, %7, %0", OP " %7, %7, %0, %1")
#define OPC(NAME, OP) BODY(NAME, OP " vcc, %0, %1", OP " vcc, %1, %2", OP " vcc, %2, %3", OP " vcc, %3, %4", OP " vcc, %4, %5", OP " vcc, %5, %6", OP " vcc, %6, %7", OP " vcc, %7, %0")
#define OPV(NAME, OP, TAIL) BODY(NAME, OP " %0, %0, %1" TAIL, OP " %1, %1, %2" TAIL, OP " %2, %2, %3" TAIL, OP " %3, %3, %4" TAIL, OP " %4, %4, %5" TAIL, OP " %5, %5, %6" TAIL, OP " %6, %6, %7" TAIL, OP " %7, %7, %0" TAIL)

OP2(b_add, "v_add_f32")
OP2(b_sub, "v_sub_f32")
OP2(b_mul, "v_mul_f32")
OP3(b_fma, "v_fma_f32")
OP2(b_max, "v_max_f32")
OP2(b_min, "v_min_f32")
OP3(b_max3, "v_max3_f32")
OP3(b_med3, "v_med3_f32")
OPC(b_cmp, "v_cmp_lt_f32")
OPV(b_cnd, "v_cndmask_b32", ", vcc")
OPV(b_sdwa, "v_cndmask_b32_sdwa", ", vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0")
OP1(b_mov, "v_mov_b32")
OP2(b_addu, "v_add_u32")
OP2(b_and, "v_and_b32")
OP2(b_or, "v_or_b32")
OP2(b_lshl, "v_lshlrev_b32")
OP3(b_bfi, "v_bfi_b32")
OP3(b_perm, "v_perm_b32")
OP3(b_lshlor, "v_lshl_or_b32")
OP3(b_add3, "v_add3_u32")
OP1(b_cvt16, "v_cvt_f16_f32")
OP1(b_cvt32, "v_cvt_f32_f16")
OP2(b_cvtpk, "v_cvt_pk_f16_f32")
OP1(b_exp, "v_exp_f32")
OP1(b_log, "v_log_f32")
OP1(b_rcp, "v_rcp_f32")
OP2(b_maxi, "v_max_i32")
OP2(b_pkadd16, "v_pk_add_f16")
OP2(b_pkmax16, "v_pk_max_f16")
OP3(b_fmamix, "v_fma_mix_f32")

typedef void (*body_fn)(float &, float &, float &, float &, float &, float &, float &, float &, int);

template <int WHICH>
__device__ __forceinline__ void run_body(float &a0, float &a1, float &a2, float &a3, float &a4, float &a5, float &a6, float &a7, int iters);
#define MAP(N, F) template <> __device__ __forceinline__ void run_body<N>(float &a0, float &a1, float &a2, float &a3, float &a4, float &a5, float &a6, float &a7, int iters) { F(a0, a1, a2, a3, a4, a5, a6, a7, iters); }
MAP(0, b_add) MAP(1, b_sub) MAP(2, b_mul) MAP(3, b_fma) MAP(4, b_max) MAP(5, b_min) MAP(6, b_max3) MAP(7, b_med3) MAP(8, b_cmp) MAP(9, b_cnd)
MAP(10, b_sdwa) MAP(11, b_mov) MAP(12, b_addu) MAP(13, b_and) MAP(14, b_or) MAP(15, b_lshl) MAP(16, b_bfi) MAP(17, b_perm) MAP(18, b_lshlor) MAP(19, b_add3)
MAP(20, b_cvt16) MAP(21, b_cvt32) MAP(22, b_cvtpk) MAP(23, b_exp) MAP(24, b_log) MAP(25, b_rcp) MAP(26, b_maxi) MAP(27, b_pkadd16) MAP(28, b_pkmax16) MAP(29, b_fmamix)
static const char *names[] = {"v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_max_f32", "v_min_f32", "v_max3_f32", "v_med3_f32", "v_cmp_lt_f32 (vcc)", "v_cndmask_b32",
                              "v_cndmask_b32_sdwa", "v_mov_b32", "v_add_u32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_bfi_b32", "v_perm_b32", "v_lshl_or_b32", "v_add3_u32",
                              "v_cvt_f16_f32", "v_cvt_f32_f16", "v_cvt_pk_f16_f32", "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_max_i32", "v_pk_add_f16", "v_pk_max_f16", "v_fma_mix_f32"};
#define NOPS 30

// PAIR = false: every wave runs WHICH.  PAIR = true: waves 0-3 run WHICH, waves 4-7 run v_add_f32.
template <int WHICH, bool PAIR>
__global__ __launch_bounds__(512) void k_probe(unsigned long long *out, float seed, int iters) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (PAIR && wave >= 4) run_body<0>(a0, a1, a2, a3, a4, a5, a6, a7, iters);
    else run_body<WHICH>(a0, a1, a2, a3, a4, a5, a6, a7, iters);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[0] = 0;
}

template <int WHICH>
static void one(unsigned long long *out) {
    const int iters = 64;
    const double n = 256.0 * iters;
    unsigned long long h[8];
    double r[3][2];
    for (int cfg = 0; cfg < 3; cfg++) {      /* 0: 4 waves (one per SIMD), 1: 8 waves same op, 2: 4 waves op + 4 waves v_add_f32 */
        for (int rep = 0; rep < 2; rep++) {
            if (cfg == 2) hipLaunchKernelGGL((k_probe<WHICH, true>), dim3(1), dim3(512), 0, 0, out, 1.0f, iters);
            else hipLaunchKernelGGL((k_probe<WHICH, false>), dim3(1), dim3(cfg == 0 ? 256 : 512), 0, 0, out, 1.0f, iters);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        if (cfg == 0) { r[0][0] = *std::min_element(h, h + 4) / n; r[0][1] = *std::max_element(h, h + 4) / n; }
        else if (cfg == 1) { r[1][0] = *std::min_element(h, h + 8) / n; r[1][1] = *std::max_element(h, h + 8) / n; }
        else { r[2][0] = *std::max_element(h, h + 4) / n; r[2][1] = *std::max_element(h + 4, h + 8) / n; }
    }
    printf("%-22s alone %5.2f | two waves per SIMD: fastest %5.2f slowest %5.2f  %-10s | beside v_add_f32: this %5.2f, the add wave %5.2f\n", names[WHICH], r[0][1], r[1][0], r[1][1],
           r[1][1] < 1.3 * r[0][1] ? "DUAL" : "take turns", r[2][0], r[2][1]);
    fflush(stdout);
}
template <int N> struct All { static void go(unsigned long long *o) { All<N - 1>::go(o); one<N - 1>(o); } };
template <> struct All<0> { static void go(unsigned long long *) {} };

int main() {
    unsigned long long *out;
    (void)hipMalloc(&out, 64 * 8);
    printf("# cycles per instruction (s_memtime), 8192 independent instructions per wave; wave w on SIMD w %% 4\n");
    All<NOPS>::go(out);
    return 0;
}
