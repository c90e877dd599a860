// Channel LayerNorm over NCHW planes (WithBias_LayerNorm, FDN_arch.py:313-342): HBM-bound.
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

}  // namespace

extern "C" int fdn_chan_stats(const float* x, long xbs, float* stats, int B, int G, int E, int P, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && stats && B > 0 && G > 0 && E > 0 && P > 0 && G < 65536 && B < 65536);
    hipLaunchKernelGGL(chan_stats_kernel, dim3(cdiv(P, 256), G, B), dim3(256), 0, static_cast<hipStream_t>(stream), x, xbs,
                       stats, G, E, P);
    return fdn_launch_status();
}

extern "C" int fdn_layernorm_chan(const float* x, const float* gamma, const float* beta, float* out, int B, int C, int P,
                                  fdn_stream_t stream) {
    FDN_CHECK_ARG(x && gamma && beta && out && B > 0 && C > 0 && P > 0 && B < 65536);
    hipLaunchKernelGGL(layernorm_chan_kernel, dim3(cdiv(P, 256), B), dim3(256), 0, static_cast<hipStream_t>(stream), x, gamma,
                       beta, out, C, P);
    return fdn_launch_status();
}
