// The step either side of the LPNet -> FDN forward, on the GPU:
//   fdn_pre_u8  : uint8 HWC image(s) -> /255 fp32 -> RGB CHW -> reflect-pad bottom/right to the x32 grid
//                 (inference_fdn_lolblur.py:47-62, basicsr/utils/img_util.py:9-33)
//   fdn_post_u8 : crop -> clamp(0,1) -> *255 -> round half-to-even -> uint8 HWC, RGB -> BGR
//                 (inference_fdn_lolblur.py:72-75, basicsr/utils/img_util.py:36-98)
// Pure byte <-> float reshuffles, HBM-bound: a thread owns one pixel (3 bytes in, 3 coalesced plane
// stores out, or the reverse).
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void pre_u8_kernel(const unsigned char* __restrict__ img, float* __restrict__ out, int h, int w,
                                                     int H, int W, int swap_rb) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= W) return;
    const int sy = y < h ? y : 2 * (h - 1) - y;                    // F.pad(mode='reflect'): no edge repeat
    const int sx = x < w ? x : 2 * (w - 1) - x;
    const unsigned char* p = img + (((long)b * h + sy) * w + sx) * 3;
    const float c0 = (float)p[0] / 255.0f, c1 = (float)p[1] / 255.0f, c2 = (float)p[2] / 255.0f;   // true division, as numpy's
    float* o = out + (long)b * 3 * H * W + (long)y * W + x;
    const long hw = (long)H * W;
    o[0] = swap_rb ? c2 : c0;
    o[hw] = c1;
    o[2 * hw] = swap_rb ? c0 : c2;
}

__global__ __launch_bounds__(256) void post_u8_kernel(const float* __restrict__ res, unsigned char* __restrict__ out, int h, int w,
                                                      int H, int W, int swap_rb) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= w) return;
    const float* r = res + (long)b * 3 * H * W + (long)y * W + x;
    const long hw = (long)H * W;
    unsigned char v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float f = r[c * hw];
        f = f < 0.f ? 0.f : (f > 1.f ? 1.f : f);                    // clamp_(0, 1)
        v[c] = (unsigned char)rintf(f * 255.0f);                    // numpy .round(): half to even
    }
    unsigned char* o = out + (((long)b * h + y) * w + x) * 3;
    o[0] = swap_rb ? v[2] : v[0];
    o[1] = v[1];
    o[2] = swap_rb ? v[0] : v[2];
}

}  // namespace

extern "C" int fdn_pre_u8(const unsigned char* img, float* out, int B, int h, int w, int H, int W, int swap_rb,
                          fdn_stream_t stream) {
    FDN_CHECK_ARG(img && out && B > 0 && h > 0 && w > 0 && H >= h && W >= w && B < 65536 && H < 65536);
    FDN_CHECK_ARG(H - h < h && W - w < w);                          // reflect padding needs pad < size
    hipLaunchKernelGGL(pre_u8_kernel, dim3(cdiv(W, 256), H, B), dim3(256), 0, static_cast<hipStream_t>(stream), img, out, h, w, H,
                       W, swap_rb);
    return fdn_launch_status();
}

extern "C" int fdn_post_u8(const float* res, unsigned char* out, int B, int h, int w, int H, int W, int swap_rb,
                           fdn_stream_t stream) {
    FDN_CHECK_ARG(res && out && B > 0 && h > 0 && w > 0 && H >= h && W >= w && B < 65536 && h < 65536);
    hipLaunchKernelGGL(post_u8_kernel, dim3(cdiv(w, 256), h, B), dim3(256), 0, static_cast<hipStream_t>(stream), res, out, h, w, H,
                       W, swap_rb);
    return fdn_launch_status();
}
