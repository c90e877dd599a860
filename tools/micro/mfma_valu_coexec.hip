// Microbenchmark: do fp32 MFMAs and fp32 vector-ALU instructions of two waves on one SIMD overlap, or do their cycles add?
// And what do v_fma_f32 vs v_pk_fma_f32 issue at?   512-thread workgroups, one per CU: waves 0-3 and 4-7 pair up on the 4 SIMDs.
//   mode 0: every wave runs the MFMA loop            mode 1: every wave runs the scalar-FMA loop (8 independent chains)
//   mode 2: waves 0-3 MFMA, waves 4-7 scalar FMA      mode 3: every wave runs the packed-FMA loop (v_pk_fma_f32)
//   mode 4: waves 0-3 MFMA, waves 4-7 packed FMA      mode 5: only waves 0-3 work (MFMA), 4-7 exit;  mode 6: only waves 4-7 (scalar FMA)
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float mfma_loop(int iters, float x, float y) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);      // 16 x 64 cycles
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    return s;
}
__device__ __forceinline__ float fma_loop(int iters, float x, float y) {
    float c[8];
    for (int j = 0; j < 8; ++j) c[j] = x + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_fmaf(c[j], y, x);                                      // 256 v_fma_f32
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += c[j];
    return s;
}
__device__ __forceinline__ float pkfma_loop(int iters, float x, float y) {
    f32x2 c[8];
    for (int j = 0; j < 8; ++j) c[j] = f32x2{x + j, x - j};
    const f32x2 yy = {y, y * 0.5f}, xx = {x, x + 1.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_elementwise_fma(c[j], yy, xx);                        // 256 v_pk_fma_f32
    }
    f32x2 s = 0.f;
    for (int j = 0; j < 8; ++j) s += c[j];
    return s.x + s.y;
}

__global__ __launch_bounds__(512) void k(float* out, int mode, int it_mfma, int it_valu, float a, float b) {
    const int wave = threadIdx.x >> 6;
    const float x = a + threadIdx.x * 1e-6f, y = b;
    float s = 0.f;
    const bool first = wave < 4;
    switch (mode) {
        case 0: s = mfma_loop(it_mfma, x, y); break;
        case 1: s = fma_loop(it_valu, x, y); break;
        case 2: s = first ? mfma_loop(it_mfma, x, y) : fma_loop(it_valu, x, y); break;
        case 3: s = pkfma_loop(it_valu, x, y); break;
        case 4: s = first ? mfma_loop(it_mfma, x, y) : pkfma_loop(it_valu, x, y); break;
        case 5: if (first) s = mfma_loop(it_mfma, x, y); break;
        case 6: if (!first) s = fma_loop(it_valu, x, y); break;
        case 7: if (!first) s = pkfma_loop(it_valu, x, y); break;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const char* names[] = {"8 waves MFMA", "8 waves v_fma", "4 MFMA + 4 v_fma", "8 waves v_pk_fma", "4 MFMA + 4 v_pk_fma", "4 waves MFMA alone",
                           "4 waves v_fma alone", "4 waves v_pk_fma alone"};
    const int it_mfma = 2000, it_valu = 2000;       // per wave: 2000 x 16 MFMAs x 64 cyc = 2.05 M cycles; 2000 x 256 FMAs x 4 cyc = 2.05 M cycles
    for (int mode = 0; mode < 8; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode, 10, 10, 1.f, 0.999f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode, it_mfma, it_valu, 1.f, 0.999f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d  %-24s %8.3f ms\n", mode, names[mode], ms);
    }
    hipFree(out);
    return 0;
}
