// Microbenchmark: an fp32 GEMM step on the bf16 matrix pipe.  x = x1 + x2 + x3 with three bf16 parts (8 bits each: the split of an
// fp32 value by truncation is EXACT), w likewise (done once, off line); the six products x1w1, x1w2, x2w1, x1w3, x2w2, x3w1 carry
// everything above 2^-24 |x w| and accumulate in fp32 inside v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate, and a pipe of its own:
// the fp32 MFMA shares the vector ALU's datapath).
//   part A  accuracy against a float64 product: fp32 MFMA chain, 6-product split (one / two accumulators), 3-product split
//   part B  time per 8 k-values of a 32x32 tile: 8 fp32 MFMAs  vs  (split of 8 activations + 6 bf16 MFMAs)  vs  6 bf16 MFMAs alone,
//           each also with extra independent v_fma_f32 work beside it (does the vector ALU run under the bf16 MFMAs?)
// hipcc --offload-arch=gfx950 -O3 tools/micro/split_bf16_mfma.hip -o /tmp/split && /tmp/split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float trunc16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_hi(float a, float b) {      // (bf16 bits of a) | (bf16 bits of b) << 16, by truncation
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// 8 fp32 values of a lane -> three bf16x8 fragments (hi, mid, lo); x = hi + mid + lo exactly
__device__ __forceinline__ void split3(const float (&v)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = v[2 * j], b = v[2 * j + 1];
        h[j] = pack_hi(a, b);
        const float ra = a - trunc16(a), rb = b - trunc16(b);
        m[j] = pack_hi(ra, rb);
        const float sa = ra - trunc16(ra), sb = rb - trunc16(rb);
        l[j] = pack_hi(sa, sb);
    }
}
#define BF(x) __builtin_bit_cast(bf16x8, x)
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a), BF(b), c, 0, 0, 0)

// ---------------------------------------------------------------------------------------------------------------------------
// part A: C[32][32] = W[32][K] X[K][32], one wave.  mode 0 fp32 MFMA; 1 six products one accumulator (small terms first per
// 16-deep step); 2 six products, a1b1 in one accumulator and the five corrections in another; 3 three products (2-way split)
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void acc_kernel(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ C, int K, int mode) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc, cor;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f, cor[i] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W[r * K + k + h], X[(k + h) * 32 + r], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            float wv[8], xv[8];
            for (int j = 0; j < 8; ++j) wv[j] = W[r * K + k + 8 * h + j], xv[j] = X[(k + 8 * h + j) * 32 + r];
            u32x4 w1, w2, w3, x1, x2, x3;
            split3(wv, w1, w2, w3);
            split3(xv, x1, x2, x3);
            if (mode == 1) {
                acc = MF(w3, x1, acc); acc = MF(w2, x2, acc); acc = MF(w1, x3, acc);
                acc = MF(w2, x1, acc); acc = MF(w1, x2, acc); acc = MF(w1, x1, acc);
            } else if (mode == 2) {
                cor = MF(w3, x1, cor); cor = MF(w2, x2, cor); cor = MF(w1, x3, cor);
                cor = MF(w2, x1, cor); cor = MF(w1, x2, cor); acc = MF(w1, x1, acc);
            } else {
                acc = MF(w2, x1, acc); acc = MF(w1, x2, acc); acc = MF(w1, x1, acc);
            }
        }
        for (int i = 0; i < 16; ++i) acc[i] += cor[i];
    }
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

// ---------------------------------------------------------------------------------------------------------------------------
// part B: timing.  Every iteration handles 8 k-values of one 32x32 tile; `extra` independent v_fma_f32 per iteration beside it.
// ---------------------------------------------------------------------------------------------------------------------------
template <int MODE, int EXTRA>
__global__ __launch_bounds__(512) void time_kernel(float* out, int iters, float a, float b) {
    float v[8], e[8];
    for (int j = 0; j < 8; ++j) v[j] = a + threadIdx.x * 1e-3f + j, e[j] = b + j;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    u32x4 w1, w2, w3, x1, x2, x3;
    {
        float wv[8];
        for (int j = 0; j < 8; ++j) wv[j] = b * (j + 1) + threadIdx.x;
        split3(wv, w1, w2, w3);
        split3(v, x1, x2, x3);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w1[j & 3]), v[j], acc, 0, 0, 0);
            } else {
                if (MODE == 1) split3(v, x1, x2, x3);
                acc = MF(w3, x1, acc); acc = MF(w2, x2, acc); acc = MF(w1, x3, acc);
                acc = MF(w2, x1, acc); acc = MF(w1, x2, acc); acc = MF(w1, x1, acc);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * 1.0001f;                       // (the operands change: nothing hoists)
#pragma unroll
            for (int x = 0; x < EXTRA; ++x) e[x & 7] = __builtin_fmaf(e[x & 7], 0.999f, a);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int j = 0; j < 8; ++j) s += e[j] + __uint_as_float(x3[j & 3]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int EXTRA>
static float run(float* out, int threads) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((time_kernel<MODE, EXTRA>), dim3(256), dim3(threads), 0, 0, out, 10, 1.f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((time_kernel<MODE, EXTRA>), dim3(256), dim3(threads), 0, 0, out, 5000, 1.f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / (5000.f * 4.f);          // ns per 8-k step per wave
}

int main() {
    // ---- part A ----
    for (int K : {32, 128, 512}) {
        for (int dist = 0; dist < 2; ++dist) {
            std::vector<float> W(32 * K), X(K * 32), C(1024);
            srand(K + dist);
            auto rn = [] { float s = 0; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.f; };
            for (auto& w : W) w = rn() / sqrtf((float)K);
            for (auto& x : X) x = dist ? rn() * 1.5f + 0.3f : rn();
            float *dW, *dX, *dC;
            hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dC, 4096);
            hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
            printf("K=%4d %s:", K, dist ? "x~N(0.3,1.5)" : "x~N(0,1)    ");
            for (int mode = 0; mode < 4; ++mode) {
                hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dW, dX, dC, K, mode);
                hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
                double se = 0, sr = 0, mx = 0;
                for (int i = 0; i < 32; ++i)
                    for (int j = 0; j < 32; ++j) {
                        double ref = 0;
                        for (int k = 0; k < K; ++k) ref += (double)W[i * K + k] * (double)X[k * 32 + j];
                        const double d = C[i * 32 + j] - ref;
                        se += d * d, sr += ref * ref, mx = fmax(mx, fabs(d));
                    }
                const char* nm[] = {"fp32-mfma", "split6-1acc", "split6-2acc", "split3"};
                printf("  %s rel-rms %.3e max %.2e", nm[mode], sqrt(se / sr), mx);
            }
            printf("\n");
            hipFree(dW); hipFree(dX); hipFree(dC);
        }
    }
    // ---- part B ----
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    for (int threads : {256, 512}) {
        printf("%d threads per CU (%d wave(s) per SIMD); ns per 8-k step of a 32x32 tile per wave (8 fp32 MFMAs = 512 cycles)\n", threads, threads / 256);
        printf("  extra v_fma per step       0      32      64     128\n");
        printf("  8 x fp32 MFMA        %7.1f %7.1f %7.1f %7.1f\n", run<0, 0>(out, threads), run<0, 32>(out, threads), run<0, 64>(out, threads), run<0, 128>(out, threads));
        printf("  split + 6 x bf16     %7.1f %7.1f %7.1f %7.1f\n", run<1, 0>(out, threads), run<1, 32>(out, threads), run<1, 64>(out, threads), run<1, 128>(out, threads));
        printf("  6 x bf16 (pre-split) %7.1f %7.1f %7.1f %7.1f\n", run<2, 0>(out, threads), run<2, 32>(out, threads), run<2, 64>(out, threads), run<2, 128>(out, threads));
    }
    return 0;
}
