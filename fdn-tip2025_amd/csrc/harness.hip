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

// Tiled inference (ImageRestorationModel.grids / grids_inverse, image_restoration_model.py:261-339, scale 1):
// gather: tile t = x[:, i_t : i_t + ch, j_t : j_t + cw]; merge: every output pixel adds the tiles covering it in tile
// order (the order of the reference's accumulation loop, so the fp32 sum is the same) and divides by their count.
__global__ __launch_bounds__(256) void tiles_gather_kernel(const float* __restrict__ x, float* __restrict__ tiles,
                                                           const int* __restrict__ ij, int C, int H, int W, int ch, int cw) {
    const int px = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, t = blockIdx.z;
    if (px >= ch * cw) return;
    const int y = px / cw, xx = px - y * cw;
    const int i = ij[2 * t], j = ij[2 * t + 1];
    tiles[(((long)t * C + c) * ch + y) * cw + xx] = x[((long)c * H + i + y) * W + j + xx];
}

__global__ __launch_bounds__(256) void tiles_merge_kernel(const float* __restrict__ tiles, float* __restrict__ out,
                                                          const int* __restrict__ ij, int T, int C, int H, int W, int ch, int cw) {
    const int px = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (px >= H * W) return;
    const int y = px / W, xx = px - y * W;
    float acc = 0.f, cnt = 0.f;
    for (int t = 0; t < T; ++t) {
        const int dy = y - ij[2 * t], dx = xx - ij[2 * t + 1];
        if (dy >= 0 && dy < ch && dx >= 0 && dx < cw) {
            acc += tiles[(((long)t * C + c) * ch + dy) * cw + dx];
            cnt += 1.0f;
        }
    }
    out[(long)c * H * W + px] = acc / cnt;
}

}  // namespace

extern "C" int fdn_tiles_gather(const float* x, float* tiles, const int* ij, int T, int C, int H, int W, int ch, int cw,
                                fdn_stream_t stream) {
    FDN_CHECK_ARG(x && tiles && ij && T > 0 && C > 0 && H > 0 && W > 0 && ch > 0 && cw > 0 && ch <= H && cw <= W && C < 65536 && T < 65536);
    hipLaunchKernelGGL(tiles_gather_kernel, dim3(cdiv((long)ch * cw, 256), C, T), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       tiles, ij, C, H, W, ch, cw);
    return fdn_launch_status();
}

extern "C" int fdn_tiles_merge(const float* tiles, float* out, const int* ij, int T, int C, int H, int W, int ch, int cw,
                               fdn_stream_t stream) {
    FDN_CHECK_ARG(tiles && out && ij && T > 0 && C > 0 && H > 0 && W > 0 && ch > 0 && cw > 0 && ch <= H && cw <= W && C < 65536);
    hipLaunchKernelGGL(tiles_merge_kernel, dim3(cdiv((long)H * W, 256), C), dim3(256), 0, static_cast<hipStream_t>(stream), tiles, out,
                       ij, T, C, H, W, ch, cw);
    return fdn_launch_status();
}

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
