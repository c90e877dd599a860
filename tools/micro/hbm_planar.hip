// Microbenchmark 3: HBM read rate of the channel-planar access patterns of the conv1x1 kernels against a linear stream.
// A [C][P] fp32 array (C = 32 planes of 7.5M pixels, 0.96 GB) is read once; what varies is how many contiguous bytes of ONE
// plane a load instruction and a workgroup touch.
//   linear      : float4 per lane, consecutive lanes consecutive addresses, whole array as one stream
//   half<VEC>   : the MFMA B-operand pattern: lanes 0-31 read plane 2s, lanes 32-63 plane 2s+1, VEC floats per lane
//                 (32*VEC*4 contiguous bytes per plane and instruction)
//   full<VEC>   : all 64 lanes on one plane (64*VEC*4 contiguous bytes)
//   mixed<VEC,N>: half<VEC> reads of 32 planes + N planes written the same way (the level-1 to_hidden conv without its MFMAs)
// hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_planar.hip -o /tmp/hbm_planar && /tmp/hbm_planar
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int C = 32;
template <int VEC> struct V { typedef float t __attribute__((ext_vector_type(VEC))); };
template <> struct V<1> { typedef float t; };
template <int VEC> __device__ float hsum(typename V<VEC>::t v) { float s = 0.f; for (int i = 0; i < VEC; ++i) s += v[i]; return s; }
template <> __device__ float hsum<1>(float v) { return v; }

__global__ __launch_bounds__(256) void linear(const float4* in, float* out, long n4) {
    float s = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256 * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long j = i + (long)u * gridDim.x * 256; v[u] = in[j < n4 ? j : 0]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 1.2345f) out[threadIdx.x] = s;
}
template <int VEC, bool HALF>
__global__ __launch_bounds__(256) void planar(const float* in, float* out, long P) {
    typedef typename V<VEC>::t vt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5, ln = lane & 31;
    const int px_wave = (HALF ? 32 : 64) * VEC, px_wg = 4 * px_wave;
    const long tiles = P / px_wg;
    float s = 0.f;
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long pix = t * px_wg + wave * px_wave + (HALF ? ln : lane) * VEC;
        vt v[HALF ? C / 2 : C];
#pragma unroll
        for (int c = 0; c < (HALF ? C / 2 : C); ++c)
            v[c] = *reinterpret_cast<const vt*>(in + (long)(HALF ? 2 * c + kh : c) * P + pix);
#pragma unroll
        for (int c = 0; c < (HALF ? C / 2 : C); ++c) s += hsum<VEC>(v[c]);
    }
    if (s == 1.2345f) out[threadIdx.x] = s;
}
template <int VEC, int N>
__global__ __launch_bounds__(256) void mixed(const float* in, float* out, long P) {
    typedef typename V<VEC>::t vt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5, ln = lane & 31;
    const int px_wave = 32 * VEC, px_wg = 4 * px_wave;
    const long tiles = P / px_wg;
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long pix = t * px_wg + wave * px_wave + ln * VEC;
        vt v[C / 2];
#pragma unroll
        for (int c = 0; c < C / 2; ++c) v[c] = *reinterpret_cast<const vt*>(in + (long)(2 * c + kh) * P + pix);
        vt a = v[0];
#pragma unroll
        for (int c = 1; c < C / 2; ++c) a += v[c];
#pragma unroll 8
        for (int r = 0; r < N / 2; ++r) *reinterpret_cast<vt*>(out + (long)(2 * r + kh) * P + pix) = a + (float)r;
    }
}
template <typename F> void run(const char* name, F launch, double bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-34s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms * 1e-9);
}
int main() {
    const long P = 8L * 736 * 1280;
    float *in, *out, *big; hipMalloc(&in, C * P * 4); hipMalloc(&out, 4096); hipMemset(in, 0, C * P * 4); hipMalloc(&big, 152 * P * 4);
    const double bytes = (double)C * P * 4;
    for (int g : {512, 1024, 2048}) {
        printf("grid %d\n", g);
        run("linear float4", [&] { hipLaunchKernelGGL(linear, dim3(g), dim3(256), 0, 0, (const float4*)in, out, C * P / 4); }, bytes);
        run("half-wave planes, 4 B lanes (128 B)", [&] { hipLaunchKernelGGL((planar<1, true>), dim3(g), dim3(256), 0, 0, in, out, P); }, bytes);
        run("half-wave planes, 8 B lanes (256 B)", [&] { hipLaunchKernelGGL((planar<2, true>), dim3(g), dim3(256), 0, 0, in, out, P); }, bytes);
        run("half-wave planes, 16 B lanes (512 B)", [&] { hipLaunchKernelGGL((planar<4, true>), dim3(g), dim3(256), 0, 0, in, out, P); }, bytes);
        run("full-wave plane, 4 B lanes (256 B)", [&] { hipLaunchKernelGGL((planar<1, false>), dim3(g), dim3(256), 0, 0, in, out, P); }, bytes);
        run("full-wave plane, 8 B lanes (512 B)", [&] { hipLaunchKernelGGL((planar<2, false>), dim3(g), dim3(256), 0, 0, in, out, P); }, bytes);
        run("full-wave plane, 16 B lanes (1 KB)", [&] { hipLaunchKernelGGL((planar<4, false>), dim3(g), dim3(256), 0, 0, in, out, P); }, bytes);
    }
    for (int g : {512, 1024}) {
        printf("grid %d, read 32 planes + write N planes\n", g);
        run("mixed 8 B lanes, N = 32", [&] { hipLaunchKernelGGL((mixed<2, 32>), dim3(g), dim3(256), 0, 0, in, big, P); }, bytes * 2);
        run("mixed 8 B lanes, N = 86", [&] { hipLaunchKernelGGL((mixed<2, 86>), dim3(g), dim3(256), 0, 0, in, big, P); }, bytes * (32 + 86) / 32);
        run("mixed 8 B lanes, N = 152", [&] { hipLaunchKernelGGL((mixed<2, 152>), dim3(g), dim3(256), 0, 0, in, big, P); }, bytes * (32 + 152) / 32);
        run("mixed 16 B lanes, N = 152", [&] { hipLaunchKernelGGL((mixed<4, 152>), dim3(g), dim3(256), 0, 0, in, big, P); }, bytes * (32 + 152) / 32);
        run("mixed 4 B lanes, N = 152", [&] { hipLaunchKernelGGL((mixed<1, 152>), dim3(g), dim3(256), 0, 0, in, big, P); }, bytes * (32 + 152) / 32);
    }
    return 0;
}
