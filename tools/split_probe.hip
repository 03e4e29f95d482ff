// Split-precision probe (experiment tool, not part of the product): can the fp32 contractions of the
// recurrence ([3S x S] . [S x 16 reads] per step, S = 96) run on the bf16 matrix pipe without leaving
// fp32 accuracy?  Each fp32 operand is cut into three bf16 pieces x = x1 + x2 + x3 (exact: 3 x 8 bits
// of mantissa, fp32's exponent range) and the product is the sum of the partial products a_i b_j with
// i + j <= 4 (six of the nine; the dropped ones are below 2^-26 of the product), accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16.  Prints, against a float64 reference:
//   * max / rms error of the exact-f32 MFMA (16x16x4), and of the 3-, 6- and 9-product bf16 splits;
//   * cycles per gate tile (16 units x 96 inputs x 16 reads) for the f32 MFMA and for the 6-product
//     split including the per-step split of the B operand.
// hipcc --offload-arch=gfx950 -O3 -o split_probe tools/split_probe.hip && ./split_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int S = 96;          // K
constexpr int MT = 18;         // m-tiles: 3 gates x 6 unit tiles

__device__ __forceinline__ u16 bf16_rn(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__device__ __forceinline__ float bf16_f(u16 h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3(float x, u16 &p1, u16 &p2, u16 &p3) {
    p1 = bf16_rn(x);
    const float r1 = x - bf16_f(p1);
    p2 = bf16_rn(r1);
    const float r2 = r1 - bf16_f(p2);
    p3 = bf16_rn(r2);
}
union V8 { bf16x8 v; u16 h[8]; };

// ---- accuracy: one wave per (m-tile, column block); A [MT*16][S] row-major, B [ncb][S][16] ----
__global__ void k_f32(const float *A, const float *B, float *C, int ncb) {
    const int lane = threadIdx.x, mt = blockIdx.x, cb = blockIdx.y;
    const int m = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < S; k0 += 4) {
        const float a = A[(mt * 16 + m) * S + k0 + q];
        const float b = B[((long long)cb * S + k0 + q) * 16 + m];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; i++) C[((long long)cb * MT * 16 + mt * 16 + 4 * q + i) * 16 + m] = acc[i];
}

template <int NPROD>   // 3, 6 or 9 partial products
__global__ void k_split(const float *A, const float *B, float *C, int ncb) {
    const int lane = threadIdx.x, mt = blockIdx.x, cb = blockIdx.y;
    const int m = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < S; k0 += 32) {
        V8 a[3], b[3];
        for (int j = 0; j < 8; j++) {
            split3(A[(mt * 16 + m) * S + k0 + 8 * q + j], a[0].h[j], a[1].h[j], a[2].h[j]);
            split3(B[((long long)cb * S + k0 + 8 * q + j) * 16 + m], b[0].h[j], b[1].h[j], b[2].h[j]);
        }
        // smallest terms first
        if (NPROD >= 9) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2].v, b[2].v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[2].v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2].v, b[1].v, acc, 0, 0, 0);
        }
        if (NPROD >= 6) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[2].v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2].v, b[0].v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[1].v, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[1].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[0].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[0].v, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; i++) C[((long long)cb * MT * 16 + mt * 16 + 4 * q + i) * 16 + m] = acc[i];
}

// ---- fp16 two-piece split: x = p1 + p2 / 2048, p1 = fp16(x), p2 = fp16((x - p1) * 2048) (the scaling keeps the
// second piece out of fp16's subnormal range); a.b = a1 b1 + (a1 b2 + a2 b1) / 2048 [+ a2 b2 / 2048^2], the
// cross terms in their own accumulator.  3 (or 4) MFMAs per 32-wide k step instead of 6.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union H8 { f16x8 v; _Float16 h[8]; };
__device__ __forceinline__ void split2h(float x, _Float16 &p1, _Float16 &p2) {
    p1 = (_Float16)x;
    p2 = (_Float16)((x - (float)p1) * 2048.0f);
}
template <int NPROD>   // 3 or 4
__global__ void k_split_h(const float *A, const float *B, float *C, int ncb) {
    const int lane = threadIdx.x, mt = blockIdx.x, cb = blockIdx.y;
    const int m = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accx = acc, accxx = acc;
    for (int k0 = 0; k0 < S; k0 += 32) {
        H8 a[2], b[2];
        for (int j = 0; j < 8; j++) {
            split2h(A[(mt * 16 + m) * S + k0 + 8 * q + j], a[0].h[j], a[1].h[j]);
            split2h(B[((long long)cb * S + k0 + 8 * q + j) * 16 + m], b[0].h[j], b[1].h[j]);
        }
        if (NPROD >= 4) accxx = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1].v, b[1].v, accxx, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0].v, b[1].v, accx, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1].v, b[0].v, accx, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0].v, b[0].v, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; i++) {
        float r = accx[i];
        if (NPROD >= 4) r += accxx[i] * (1.0f / 2048.0f);
        C[((long long)cb * MT * 16 + mt * 16 + 4 * q + i) * 16 + m] = acc[i] + r * (1.0f / 2048.0f);
    }
}

// ---- fp16 two-piece split, common scale, ONE accumulator: A' = 256 A, B' = 64 B, each cut into p1 = fp16(x'),
// p2 = fp16(x' - p1) (no relative scaling: the up-scaling keeps p2 out of fp16's subnormals); the accumulator takes the
// cross terms of all k steps first, then the main terms; result = acc / 2^14.
__device__ __forceinline__ void split2u(float x, _Float16 &p1, _Float16 &p2) {
    p1 = (_Float16)x;
    p2 = (_Float16)(x - (float)p1);
}
template <int ORDER>   // 0: per k step (x1, x2, m); 1: all cross terms first, then all main terms
__global__ void k_split_u(const float *A, const float *B, float *C, int ncb) {
    const int lane = threadIdx.x, mt = blockIdx.x, cb = blockIdx.y;
    const int m = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    H8 a[3][2], b[3][2];
    for (int ks = 0; ks < 3; ks++)
        for (int j = 0; j < 8; j++) {
            split2u(256.0f * A[(mt * 16 + m) * S + ks * 32 + 8 * q + j], a[ks][0].h[j], a[ks][1].h[j]);
            split2u(64.0f * B[((long long)cb * S + ks * 32 + 8 * q + j) * 16 + m], b[ks][0].h[j], b[ks][1].h[j]);
        }
    if (ORDER == 0) {
        for (int ks = 0; ks < 3; ks++) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks][0].v, b[ks][1].v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks][1].v, b[ks][0].v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks][0].v, b[ks][0].v, acc, 0, 0, 0);
        }
    } else {
        for (int ks = 0; ks < 3; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks][0].v, b[ks][1].v, acc, 0, 0, 0);
        for (int ks = 0; ks < 3; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks][1].v, b[ks][0].v, acc, 0, 0, 0);
        for (int ks = 0; ks < 3; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks][0].v, b[ks][0].v, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; i++) C[((long long)cb * MT * 16 + mt * 16 + 4 * q + i) * 16 + m] = acc[i] * (1.0f / 16384.0f);
}

// ---- rate: a wave owns 3 gate tiles (weights resident), per step: take B (fp32, 24 registers), [split], GEMM ----
template <bool SPLIT>
__global__ __launch_bounds__(768) void k_rate(const float *A, const float *B, float *out, unsigned long long *cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, q = lane >> 4;
    float bf[24];
    for (int r = 0; r < 24; r++) bf[r] = B[(r * 4 + q) * 16 + m];
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    unsigned long long t0, t1;
    if (!SPLIT) {
        float a[3][24];
        for (int g = 0; g < 3; g++) for (int r = 0; r < 24; r++) a[g][r] = A[((g * 6 + wave % 6) * 16 + m) * S + r * 4 + q];
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 24; r++) asm volatile("" : "+v"(bf[r]));
#pragma unroll
            for (int r = 0; r < 24; r++)
#pragma unroll
                for (int g = 0; g < 3; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][r], bf[r], acc[g], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
    } else {
        V8 a[3][3][3];     // [gate][k-step][piece]
        for (int g = 0; g < 3; g++) for (int ks = 0; ks < 3; ks++) for (int j = 0; j < 8; j++)
            split3(A[((g * 6 + wave % 6) * 16 + m) * S + ks * 32 + 8 * q + j], a[g][ks][0].h[j], a[g][ks][1].h[j], a[g][ks][2].h[j]);
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 24; r++) asm volatile("" : "+v"(bf[r]));
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
                V8 b[3];
#pragma unroll
                for (int j = 0; j < 8; j++) split3(bf[ks * 8 + j], b[0].h[j], b[1].h[j], b[2].h[j]);
#pragma unroll
                for (int g = 0; g < 3; g++) {
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][ks][0].v, b[2].v, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][ks][2].v, b[0].v, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][ks][1].v, b[1].v, acc[g], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 3; g++) {
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][ks][0].v, b[1].v, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][ks][1].v, b[0].v, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][ks][0].v, b[0].v, acc[g], 0, 0, 0);
                }
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    float s = 0;
    for (int g = 0; g < 3; g++) for (int i = 0; i < 4; i++) s += acc[g][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

static void report(const char *name, const std::vector<float> &C, const std::vector<double> &R) {
    double mx = 0, ss = 0, rs = 0;
    for (size_t i = 0; i < C.size(); i++) { const double d = C[i] - R[i]; mx = std::max(mx, std::fabs(d)); ss += d * d; rs += R[i] * R[i]; }
    printf("%-28s max |err| %.3e   rms err %.3e   (rms of the result %.3f)\n", name, mx, std::sqrt(ss / C.size()), std::sqrt(rs / C.size()));
}

int main() {
    const int ncb = 512;
    std::vector<float> A((size_t)MT * 16 * S), B((size_t)ncb * S * 16);
    srand(1);
    const float wr = std::sqrt(3.0f / S);
    for (auto &x : A) x = wr * (2.0f * rand() / RAND_MAX - 1.0f);
    for (auto &x : B) x = 2.0f * rand() / RAND_MAX - 1.0f;          // recurrent state: tanh/blend output in (-1, 1)
    std::vector<double> R((size_t)ncb * MT * 16 * 16);
    for (int cb = 0; cb < ncb; cb++) for (int r = 0; r < MT * 16; r++) for (int n = 0; n < 16; n++) {
        double s = 0;
        for (int k = 0; k < S; k++) s += (double)A[(size_t)r * S + k] * (double)B[((size_t)cb * S + k) * 16 + n];
        R[((size_t)cb * MT * 16 + r) * 16 + n] = s;
    }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, R.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> C(R.size());
    auto run = [&](const char *name, auto kern) {
        hipMemset(dC, 0, R.size() * 4);
        hipLaunchKernelGGL(kern, dim3(MT, ncb), dim3(64), 0, 0, dA, dB, dC, ncb);
        hipDeviceSynchronize();
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        report(name, C, R);
    };
    run("f32 MFMA 16x16x4", k_f32);
    run("bf16 split, 3 products", k_split<3>);
    run("bf16 split, 6 products", k_split<6>);
    run("bf16 split, 9 products", k_split<9>);
    run("fp16 2-piece split, 3 products", k_split_h<3>);
    run("fp16 2-piece split, 4 products", k_split_h<4>);
    run("fp16 common scale, 1 acc, per k", k_split_u<0>);
    run("fp16 common scale, 1 acc, cross 1st", k_split_u<1>);
    for (float sc : {1e-2f, 1e-4f, 10.0f}) {      // operand magnitude: fp16's exponent range
        std::vector<float> B2(B);
        for (auto &x : B2) x *= sc;
        hipMemcpy(dB, B2.data(), B2.size() * 4, hipMemcpyHostToDevice);
        for (int cb = 0; cb < ncb; cb++) for (int r = 0; r < MT * 16; r++) for (int n = 0; n < 16; n++) {
            double s = 0;
            for (int k = 0; k < S; k++) s += (double)A[(size_t)r * S + k] * (double)B2[((size_t)cb * S + k) * 16 + n];
            R[((size_t)cb * MT * 16 + r) * 16 + n] = s;
        }
        printf("-- B scaled by %g:\n", sc);
        run("  f32 MFMA 16x16x4", k_f32);
        run("  bf16 split, 6 products", k_split<6>);
        run("  fp16 2-piece split, 3 products", k_split_h<3>);
        run("  fp16 2-piece split, 4 products", k_split_h<4>);
        run("  fp16 common scale, 1 acc, per k", k_split_u<0>);
        run("  fp16 common scale, 1 acc, cross 1st", k_split_u<1>);
    }
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    {   // a float fmaf chain on the host in natural k order, for scale
        for (int cb = 0; cb < ncb; cb++) for (int r = 0; r < MT * 16; r++) for (int n = 0; n < 16; n++) {
            float s = 0;
            for (int k = 0; k < S; k++) s = fmaf(A[(size_t)r * S + k], B[((size_t)cb * S + k) * 16 + n], s);
            C[((size_t)cb * MT * 16 + r) * 16 + n] = s;
        }
        report("host fmaf chain (f32)", C, R);
    }
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 1 << 16);
    const int iters = 2000;
    for (int split = 0; split < 2; split++) {
        for (int waves : {4, 12}) {
            if (split) hipLaunchKernelGGL(k_rate<true>, dim3(256), dim3(64 * waves), 0, 0, dA, dB, out, cyc, iters);
            else hipLaunchKernelGGL(k_rate<false>, dim3(256), dim3(64 * waves), 0, 0, dA, dB, out, cyc, iters);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(waves);
            hipMemcpy(h.data(), cyc, 8 * waves, hipMemcpyDeviceToHost);
            double mxc = 0;
            for (auto v : h) mxc = std::max(mxc, (double)v);
            printf("%s, %2d waves per CU: %.0f cycles per step of a wave (3 gate tiles, 16 reads), slowest wave of workgroup 0\n",
                   split ? "bf16 6-product split (B split every step)" : "f32 MFMA                                  ", waves, mxc / iters);
        }
    }
    return 0;
}
