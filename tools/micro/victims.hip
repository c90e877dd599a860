// Noise kernels for tools/cross_stream_probe.py: loops of ONE kind of instruction (bf16 / fp32 matrix instructions, vector FMAs) with
// no LDS and no memory traffic inside the loop, to run BESIDE another stream's kernels.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/victims.hip -o abx/libvictims.so
// (Synthetic victims - a static pattern held in LDS / in vector registers, an LDS write-barrier-read exchange, duplicated FMA chains,
//  global loads of a constant table, global stores - showed 0 corrupted words beside every noise kernel; only real kernels are hit.)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(16))) float f32x16_;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_;
template <int MODE>
__global__ __launch_bounds__(256) void n_mfma(float* out, int iters) {
    extern __shared__ float sm[];
    bf16x8_ a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x ^ j)); }
    f32x16_ c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // four independent accumulators
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        } else {                    // a dependent chain of six, then the result goes to LDS (1) / memory (2) / nowhere (3)
#pragma unroll
            for (int k = 0; k < 6; ++k) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sm[r * 256 + threadIdx.x] = c0[r];
            } else if (MODE == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) out[((long)blockIdx.x * 16 + r) * 256 + threadIdx.x] = c0[r];
            }
        }
    }
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += c0[r] + c1[r] + c2[r] + c3[r];
    if (MODE == 1) acc += sm[threadIdx.x];
    out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}
typedef __attribute__((ext_vector_type(4))) float f32x4_;
template <int MODE>
__global__ __launch_bounds__(256) void n_other(float* out, int iters) {
    float acc = 0.f;
    if (MODE == 4) {                // vector ALU only
        float v[16];
        for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r;
        for (int it = 0; it < iters * 8; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaf(v[r], 0.9999f, 1e-4f);
        for (int r = 0; r < 16; ++r) acc += v[r];
    } else if (MODE == 5) {         // fp32 matrix instructions
        f32x16_ c0 = {0}, c1 = {0};
        const float a = threadIdx.x * 1e-3f, b = 1e-3f;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) acc += c0[r] + c1[r];
    } else {                        // 16x16x32 bf16
        bf16x8_ a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x ^ j)); }
        f32x4_ c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int it = 0; it < iters * 2; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) acc += c0[r] + c1[r] + c2[r] + c3[r];
    }
    out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}
extern "C" int noise(int mode, void* out, int blocks, int iters, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* o = static_cast<float*>(out);
    switch (mode) {
        case 0: hipLaunchKernelGGL(n_mfma<0>, dim3(blocks), dim3(256), 0, s, o, iters); break;
        case 1: hipLaunchKernelGGL(n_mfma<1>, dim3(blocks), dim3(256), 16 * 256 * 4, s, o, iters); break;
        case 2: hipLaunchKernelGGL(n_mfma<2>, dim3(blocks), dim3(256), 0, s, o, iters); break;
        case 3: hipLaunchKernelGGL(n_mfma<3>, dim3(blocks), dim3(256), 0, s, o, iters); break;
        case 4: hipLaunchKernelGGL(n_other<4>, dim3(blocks), dim3(256), 0, s, o, iters); break;
        case 5: hipLaunchKernelGGL(n_other<5>, dim3(blocks), dim3(256), 0, s, o, iters); break;
        case 6: hipLaunchKernelGGL(n_other<6>, dim3(blocks), dim3(256), 0, s, o, iters); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
