// Microbenchmark 2: the inner loop of the conv1x1 small-K kernels in isolation - B operands in a register strip,
// A operands read from LDS one group ahead, two MFMA chains - against the plain issue rate of mfma_peak.hip.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_lds.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// MODE 0: A and B both fixed registers; 1: B from a 64-entry register strip, A fixed; 2: A from LDS (prefetched), B strip;
// 3: as 2 with the accumulators stored (b64) after every 128 MFMAs
template <int MODE, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void kern(float* out, const float* in, int iters) {
    __shared__ float Wl[128 * 33];
    constexpr int KS = 64, WS = 33;
    const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5, ln = lane & 31;
    for (int i = tid; i < 128 * 33; i += NWAVE * 64) Wl[i] = in[i & 1023];
    __syncthreads();
    f32x2 xa[KS];
    for (int s = 0; s < KS; ++s) xa[s] = f32x2{in[(s * 64 + lane) & 1023], in[(s * 64 + lane + 7) & 1023]};
    f32x16 acc[2];
    for (int v = 0; v < 2; ++v) for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;
    const float* w = Wl + kh * WS + ln;
    float* o = out + (blockIdx.x * NWAVE * 64 + tid) * 2;
    for (int it = 0; it < iters; ++it) {
        float a[2][8];
        if (MODE >= 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[0][i] = w[i * 2 * WS];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[0][i] = a[1][i] = xa[0][0];
        }
#pragma unroll
        for (int grp = 0; grp < KS / 8; ++grp) {
            if (MODE >= 2 && grp + 1 < KS / 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[(grp + 1) & 1][i] = w[((grp + 1) * 8 + i) * 2 * WS];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int v = 0; v < 2; ++v)
                    acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[grp & 1][i], MODE >= 1 ? xa[grp * 8 + i][v] : xa[0][v], acc[v], 0, 0, 0);
        }
        if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                *reinterpret_cast<f32x2*>(o + (long)r * 1048576) = f32x2{acc[0][r], acc[1][r]};
                acc[0][r] = 0.f; acc[1][r] = 0.f;
            }
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int v = 0; v < 2; ++v) for (int r = 0; r < 16; ++r) s += acc[v][r];
    o[0] = s;
}
template <typename K>
void run(const char* name, K k, int nwave, int grid) {
    float *out, *in; hipMalloc(&out, 64u << 20); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
    const int iters = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(nwave * 64), 0, 0, out, in, 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(nwave * 64), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms  %.1f TFLOP/s\n", name, ms, (double)grid * nwave * iters * 128 * 4096 / ms * 1e-9);
    hipFree(out); hipFree(in);
}
int main() {
    run("fixed A,B            8 waves x 256", kern<0, 8>, 8, 256);
    run("B strip              8 waves x 256", kern<1, 8>, 8, 256);
    run("A from LDS, B strip  8 waves x 256", kern<2, 8>, 8, 256);
    run("  + stores           8 waves x 256", kern<3, 8>, 8, 256);
    run("A from LDS, B strip  4 waves x 512", kern<2, 4>, 4, 512);
    run("  + stores           4 waves x 512", kern<3, 4>, 4, 512);
    run("A from LDS, B strip  4 waves x 256 (1/SIMD)", kern<2, 4>, 4, 256);
    return 0;
}
