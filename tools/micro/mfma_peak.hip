// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 / 16x16x4 with 1, 2, 4 independent chains and 1 or 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float x = a + threadIdx.x, y = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
    float x = a + threadIdx.x, y = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 4; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int ch, double flops_per, int wgs_per_cu) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 16 * ch * flops_per;
    printf("%-10s chains=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", name, ch, wgs_per_cu, ms, fl / ms * 1e-9);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 2; ++w) {
        run("32x32x2", k32<1>, 1, 4096, w); run("32x32x2", k32<2>, 2, 4096, w); run("32x32x2", k32<4>, 4, 4096, w);
        run("16x16x4", k16<1>, 1, 2048, w); run("16x16x4", k16<2>, 2, 2048, w); run("16x16x4", k16<4>, 4, 2048, w);
    }
    return 0;
}
