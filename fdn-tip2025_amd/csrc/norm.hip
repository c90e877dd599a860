// Channel LayerNorm over NCHW planes (WithBias_LayerNorm, FDN_arch.py:313-342): HBM-bound.
// 16-byte lanes matter: the float4 forms below reach 4.8-5.1 TB/s where the dword forms stop at 3.4-3.5 TB/s.
// One thread owns one pixel; the channel loop reads are coalesced across the wave (consecutive
// lanes = consecutive pixels of one channel plane).  Variance is accumulated around the first
// channel's value (shifted sums) so mean^2 cancellation cannot occur.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void chan_stats_kernel(const float* __restrict__ x, long xbs, float* __restrict__ stats,
                                                         int G, int E, int P) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y, b = blockIdx.z;
    if (p >= P) return;
    const float* src = x + (long)b * xbs + (long)g * E * P + p;
    const float x0 = src[0];
    float s = 0.f, ss = 0.f;
    for (int c = 1; c < E; ++c) {
        const float dlt = src[(long)c * P] - x0;
        s += dlt;
        ss += dlt * dlt;
    }
    const float inv = 1.0f / (float)E;
    const float md = s * inv;                       // mean - x0
    const float var = fmaxf(ss * inv - md * md, 0.f);
    float* dst = stats + ((long)b * G + g) * 2 * P;
    dst[p] = x0 + md;
    dst[P + p] = 1.0f / sqrtf(var + 1e-5f);
}

// float4 form: a thread owns 4 consecutive pixels, so a workgroup reads 4 KB contiguous per channel plane (16-byte
// lanes); used when P % 4 == 0 and the tensors are 16-byte aligned
__global__ __launch_bounds__(256) void chan_stats4_kernel(const float* __restrict__ x, long xbs, float* __restrict__ stats,
                                                          int G, int E, int P) {
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const int g = blockIdx.y, b = blockIdx.z;
    if (p >= P) return;
    const float* src = x + (long)b * xbs + (long)g * E * P + p;
    const float4 x0 = *reinterpret_cast<const float4*>(src);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), ss = s;
#pragma unroll 4
    for (int c = 1; c < E; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(src + (long)c * P);
        const float4 d = make_float4(v.x - x0.x, v.y - x0.y, v.z - x0.z, v.w - x0.w);
        s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
        ss.x += d.x * d.x; ss.y += d.y * d.y; ss.z += d.z * d.z; ss.w += d.w * d.w;
    }
    const float inv = 1.0f / (float)E;
    const float4 md = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
    float* dst = stats + ((long)b * G + g) * 2 * P;
    *reinterpret_cast<float4*>(dst + p) = make_float4(x0.x + md.x, x0.y + md.y, x0.z + md.z, x0.w + md.w);
    *reinterpret_cast<float4*>(dst + P + p) =
        make_float4(1.0f / sqrtf(fmaxf(ss.x * inv - md.x * md.x, 0.f) + 1e-5f), 1.0f / sqrtf(fmaxf(ss.y * inv - md.y * md.y, 0.f) + 1e-5f),
                    1.0f / sqrtf(fmaxf(ss.z * inv - md.z * md.z, 0.f) + 1e-5f), 1.0f / sqrtf(fmaxf(ss.w * inv - md.w * md.w, 0.f) + 1e-5f));
}

__global__ __launch_bounds__(256) void layernorm_chan_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ out, int C,
                                                             int P) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= P) return;
    const float* src = x + (long)b * C * P + p;
    float* dst = out + (long)b * C * P + p;
    const float x0 = src[0];
    float s = 0.f, ss = 0.f;
    for (int c = 1; c < C; ++c) {
        const float dlt = src[(long)c * P] - x0;
        s += dlt;
        ss += dlt * dlt;
    }
    const float inv = 1.0f / (float)C;
    const float md = s * inv;
    const float mu = x0 + md;
    const float rs = 1.0f / sqrtf(fmaxf(ss * inv - md * md, 0.f) + 1e-5f);
    for (int c = 0; c < C; ++c) dst[(long)c * P] = (src[(long)c * P] - mu) * rs * gamma[c] + beta[c];
}

__global__ __launch_bounds__(256) void layernorm_chan4_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ out, int C,
                                                              int P) {
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const int b = blockIdx.y;
    if (p >= P) return;
    const float* src = x + (long)b * C * P + p;
    float* dst = out + (long)b * C * P + p;
    const float4 x0 = *reinterpret_cast<const float4*>(src);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), ss = s;
#pragma unroll 4
    for (int c = 1; c < C; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(src + (long)c * P);
        const float4 d = make_float4(v.x - x0.x, v.y - x0.y, v.z - x0.z, v.w - x0.w);
        s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
        ss.x += d.x * d.x; ss.y += d.y * d.y; ss.z += d.z * d.z; ss.w += d.w * d.w;
    }
    const float inv = 1.0f / (float)C;
    const float4 md = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
    const float4 mu = make_float4(x0.x + md.x, x0.y + md.y, x0.z + md.z, x0.w + md.w);
    const float4 rs = make_float4(1.0f / sqrtf(fmaxf(ss.x * inv - md.x * md.x, 0.f) + 1e-5f), 1.0f / sqrtf(fmaxf(ss.y * inv - md.y * md.y, 0.f) + 1e-5f),
                                  1.0f / sqrtf(fmaxf(ss.z * inv - md.z * md.z, 0.f) + 1e-5f), 1.0f / sqrtf(fmaxf(ss.w * inv - md.w * md.w, 0.f) + 1e-5f));
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(src + (long)c * P);
        const float g = gamma[c], bt = beta[c];
        *reinterpret_cast<float4*>(dst + (long)c * P) =
            make_float4((v.x - mu.x) * rs.x * g + bt, (v.y - mu.y) * rs.y * g + bt, (v.z - mu.z) * rs.z * g + bt, (v.w - mu.w) * rs.w * g + bt);
    }
}

}  // namespace

extern "C" int fdn_chan_stats(const float* x, long xbs, float* stats, int B, int G, int E, int P, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && stats && B > 0 && G > 0 && E > 0 && P > 0 && G < 65536 && B < 65536);
    const bool v4 = P % 4 == 0 && xbs % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(stats)) & 15) == 0;
    if (v4)
        hipLaunchKernelGGL(chan_stats4_kernel, dim3(cdiv(P, 1024), G, B), dim3(256), 0, static_cast<hipStream_t>(stream), x, xbs, stats,
                           G, E, P);
    else
        hipLaunchKernelGGL(chan_stats_kernel, dim3(cdiv(P, 256), G, B), dim3(256), 0, static_cast<hipStream_t>(stream), x, xbs,
                           stats, G, E, P);
    return fdn_launch_status();
}

extern "C" int fdn_layernorm_chan(const float* x, const float* gamma, const float* beta, float* out, int B, int C, int P,
                                  fdn_stream_t stream) {
    FDN_CHECK_ARG(x && gamma && beta && out && B > 0 && C > 0 && P > 0 && B < 65536);
    if (P % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0)
        hipLaunchKernelGGL(layernorm_chan4_kernel, dim3(cdiv(P, 1024), B), dim3(256), 0, static_cast<hipStream_t>(stream), x, gamma,
                           beta, out, C, P);
    else
        hipLaunchKernelGGL(layernorm_chan_kernel, dim3(cdiv(P, 256), B), dim3(256), 0, static_cast<hipStream_t>(stream), x, gamma,
                           beta, out, C, P);
    return fdn_launch_status();
}
