// Issue rate of scalar and packed fp32 FMAs (register operands only, independent chains): does v_pk_fma_f32 run at the rate of v_fma_f32?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a[16];
    f2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i, p[i] = f2{a[i], a[i] + 1.f};
    const f2 s2 = {s, s * 1.0001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], s, 0.5f);                               // v_fma_f32 (16 per trip)
            else if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], s2, f2{0.5f, 0.25f});      // v_pk_fma_f32 (16 per trip)
            else { a[i] = __builtin_fmaf(a[i], s, 0.5f); p[i] = __builtin_elementwise_fma(p[i], s2, f2{0.5f, 0.25f}); }   // alternating (32 per trip)
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount, iters = 20000;
    float* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg = 1; wg <= 8; wg *= 2)
        for (int mode = 0; mode < 3; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(cus * wg), dim3(256), 0, 0, out, iters, 0.999f);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(cus * wg), dim3(256), 0, 0, out, iters, 0.999f);
                else hipLaunchKernelGGL(k<2>, dim3(cus * wg), dim3(256), 0, 0, out, iters, 0.999f);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double inst = (double)iters * (mode == 2 ? 32 : 16);          // wave instructions per wave
            const double waves_per_simd = wg;                                    // 4 waves per workgroup, 4 SIMDs
            printf("%s, %d waves per SIMD: %.2f cycles (2.4 GHz) per instruction and SIMD, %.1f TFLOP/s\n",
                   mode == 0 ? "v_fma_f32   " : mode == 1 ? "v_pk_fma_f32" : "alternating ", wg, ms * 1e-3 * 2.4e9 / (inst * waves_per_simd),
                   (double)cus * wg * 256 * iters * (mode == 0 ? 32 : mode == 1 ? 64 : 96) / (ms * 1e-3) / 1e12);
        }
    return 0;
}
