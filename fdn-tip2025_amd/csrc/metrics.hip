// In-repo image quality metrics on the GPU: the default validation metrics of the
// reference, basicsr/metrics/psnr_ssim.py:8-73 (calculate_psnr) and :163-197 (_ssim_3d, the ssim3d=True path of
// calculate_ssim).  HBM-bound reductions / separable 11-tap filters.
//   fdn_sse_max : sum of squared differences in fp64 (the reference squares and averages float64 arrays) + max of img1
//   fdn_ssim3d  : the 11x11x11 Gaussian window of _generate_3d_gaussian_kernel is the outer product of three 1-D
//                 cv2.getGaussianKernel(11, 1.5) kernels, applied here as three replicate-padded 1-D passes over
//                 (H, W, C) for the five fields x, y, x^2, y^2, xy; the last pass evaluates the SSIM map and reduces it
#include "common.hpp"

namespace {

__device__ __forceinline__ double block_sum(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    return red[0];
}

__global__ __launch_bounds__(256) void sse_max_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                      double* __restrict__ out) {
    __shared__ double red[256];
    __shared__ float redm[256];
    double s = 0.0;
    float m = 0.f;                                    // images are non-negative; max() <= 1 decides the peak value (:60)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const double d = (double)a[i] - (double)b[i];
        s += d * d;
        m = fmaxf(m, a[i]);
    }
    redm[threadIdx.x] = m;
    const double tot = block_sum(s, red);             // (barriers inside also order redm)
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) redm[threadIdx.x] = fmaxf(redm[threadIdx.x], redm[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(out, tot);
        atomicMax(reinterpret_cast<unsigned long long*>(out + 1), (unsigned long long)__double_as_longlong((double)redm[0]));
    }
}

struct G11 { float w[11]; };

// one replicate-padded 11-tap pass along the axis of stride `st` and length `len`; FIRST builds the five fields from a, b
template <bool FIRST>
__global__ __launch_bounds__(256) void gauss_pass_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ src, float* __restrict__ dst, long n, long st,
                                                         int len, G11 g) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int pos = (int)((i / st) % len);
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 11; ++t) {
        int p = pos + t - 5;
        p = p < 0 ? 0 : (p >= len ? len - 1 : p);     // padding_mode='replicate' (:157)
        const long j = i + (long)(p - pos) * st;
        if (FIRST) {
            const float x = a[j], y = b[j];
            acc[0] = fmaf(g.w[t], x, acc[0]);
            acc[1] = fmaf(g.w[t], y, acc[1]);
            acc[2] = fmaf(g.w[t], x * x, acc[2]);
            acc[3] = fmaf(g.w[t], y * y, acc[3]);
            acc[4] = fmaf(g.w[t], x * y, acc[4]);
        } else {
#pragma unroll
            for (int f = 0; f < 5; ++f) acc[f] = fmaf(g.w[t], src[f * n + j], acc[f]);
        }
    }
#pragma unroll
    for (int f = 0; f < 5; ++f) dst[f * n + i] = acc[f];
}

// last pass (along the channel axis) + SSIM map (:186-196) + sum
__global__ __launch_bounds__(256) void ssim_final_kernel(const float* __restrict__ src, long n, long st, int len, G11 g, float C1,
                                                         float C2, double* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int pos = (int)((i / st) % len);
        float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 11; ++t) {
            int p = pos + t - 5;
            p = p < 0 ? 0 : (p >= len ? len - 1 : p);
            const long j = i + (long)(p - pos) * st;
#pragma unroll
            for (int f = 0; f < 5; ++f) v[f] = fmaf(g.w[t], src[f * n + j], v[f]);
        }
        const float mu1 = v[0], mu2 = v[1];
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = v[2] - mu1_sq, s2 = v[3] - mu2_sq, s12 = v[4] - mu12;
        s += (double)(((2.f * mu12 + C1) * (2.f * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2)));
    }
    const double tot = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, tot);
}

// ---- Y channel (test_y_channel=True): metric_util.py:34-47 -> matlab_functions.py:207-238 (bgr2ycbcr, y_only) -------------------
// img [3][H][W] in B, G, R order, range [0, 255]: float32 image / 255, float64 dot with (24.966, 128.553, 65.481) + 16, / 255 and
// back to float32, * 255 in float32 - the reference's own mix of widths
__global__ __launch_bounds__(256) void y_channel_kernel(const float* __restrict__ img, float* __restrict__ out, long hw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= hw) return;
    const double b = (double)(img[i] / 255.0f), g = (double)(img[hw + i] / 255.0f), r = (double)(img[2 * hw + i] / 255.0f);
    const double y = b * 24.966 + g * 128.553 + r * 65.481 + 16.0;
    out[i] = (float)(y / 255.0) * 255.0f;
}

// ---- 2-D SSIM in float64 (the reference filters float64 arrays with cv2.filter2D): _ssim (:84-116) and _ssim_cly (:199-240) -----
struct G11d { double w[11]; };
__device__ __forceinline__ int border_index(int p, int len, int replicate) {
    if (replicate) return p < 0 ? 0 : (p >= len ? len - 1 : p);                      // BORDER_REPLICATE (:222)
    if (p < 0) p = -p;                                                                // BORDER_REFLECT_101, cv2.filter2D's default
    if (p >= len) p = 2 * (len - 1) - p;
    return p;
}
// pass along W: the five fields x, y, x^2, y^2, xy filtered horizontally -> ws [5][C*H*W] (double)
__global__ __launch_bounds__(256) void ssim2d_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ ws,
                                                          long n, int W, G11d g, int replicate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x0 = (int)(i % W);
    const long row = i - x0;
    double acc[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 11; ++t) {
        const int p = border_index(x0 + t - 5, W, replicate);
        const double x = (double)a[row + p], y = (double)b[row + p];
        acc[0] += g.w[t] * x;
        acc[1] += g.w[t] * y;
        acc[2] += g.w[t] * (x * x);
        acc[3] += g.w[t] * (y * y);
        acc[4] += g.w[t] * (x * y);
    }
#pragma unroll
    for (int f = 0; f < 5; ++f) ws[f * n + i] = acc[f];
}
// pass along H + SSIM map + sum; crop = 5 restricts the map to the valid region [5:-5, 5:-5] (:104-109)
__global__ __launch_bounds__(256) void ssim2d_cols_kernel(const double* __restrict__ ws, long n, int H, int W, G11d g, int replicate, int crop,
                                                          double C1, double C2, double* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x0 = (int)(i % W), y0 = (int)((i / W) % H);
        if (x0 < crop || x0 >= W - crop || y0 < crop || y0 >= H - crop) continue;
        const long plane = i - (long)y0 * W - x0;
        double v[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 11; ++t) {
            const long j = plane + (long)border_index(y0 + t - 5, H, replicate) * W + x0;
#pragma unroll
            for (int f = 0; f < 5; ++f) v[f] += g.w[t] * ws[f * n + j];
        }
        const double mu1 = v[0], mu2 = v[1];
        const double mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const double s1 = v[2] - mu1_sq, s2 = v[3] - mu2_sq, s12 = v[4] - mu12;
        s += ((2.0 * mu12 + C1) * (2.0 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
    }
    const double tot = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, tot);
}

}  // namespace

extern "C" int fdn_sse_max(const float* a, const float* b, long n, double* out2, fdn_stream_t stream) {
    FDN_CHECK_ARG(a && b && out2 && n > 0);
    long blocks = cdiv(n, 256L * 8);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sse_max_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, n, out2);
    return fdn_launch_status();
}

extern "C" int fdn_ssim3d(const float* a, const float* b, int C, int H, int W, float max_value, float* ws, double* out_sum,
                          fdn_stream_t stream) {
    FDN_CHECK_ARG(a && b && ws && out_sum && C > 0 && H > 0 && W > 0 && max_value > 0.f);
    G11 g;                                              // cv2.getGaussianKernel(11, 1.5): exp(-(i-5)^2 / (2 sigma^2)), normalised
    double w[11], sum = 0.0;
    for (int i = 0; i < 11; ++i) { w[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += w[i]; }
    for (int i = 0; i < 11; ++i) g.w[i] = (float)(w[i] / sum);
    const long n = (long)C * H * W;
    float* t1 = ws;
    float* t2 = ws + 5 * n;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)cdiv(n, 256L);
    // [C][H][W] tensors: W axis stride 1, H axis stride W, channel axis stride H*W (the reference filters the (H, W, C) volume)
    hipLaunchKernelGGL(gauss_pass_kernel<true>, dim3(blocks), dim3(256), 0, s, a, b, (const float*)nullptr, t1, n, 1L, W, g);
    hipLaunchKernelGGL(gauss_pass_kernel<false>, dim3(blocks), dim3(256), 0, s, (const float*)nullptr, (const float*)nullptr, t1, t2, n,
                       (long)W, H, g);
    const float C1 = (float)((0.01 * (double)max_value) * (0.01 * (double)max_value));   // python doubles, cast when they meet the fp32 maps
    const float C2 = (float)((0.03 * (double)max_value) * (0.03 * (double)max_value));
    long rb = cdiv(n, 256L * 4);
    if (rb > 4096) rb = 4096;
    hipLaunchKernelGGL(ssim_final_kernel, dim3((unsigned)rb), dim3(256), 0, s, t2, n, (long)H * W, C, g, C1, C2, out_sum);
    return fdn_launch_status();
}

extern "C" int fdn_y_channel(const float* img_bgr, float* out, int H, int W, fdn_stream_t stream) {
    FDN_CHECK_ARG(img_bgr && out && H > 0 && W > 0);
    const long hw = (long)H * W;
    hipLaunchKernelGGL(y_channel_kernel, dim3((unsigned)cdiv(hw, 256L)), dim3(256), 0, static_cast<hipStream_t>(stream), img_bgr, out, hw);
    return fdn_launch_status();
}

extern "C" int fdn_ssim2d(const float* a, const float* b, int C, int H, int W, float max_value, int replicate_no_crop, double* ws,
                          double* out_sum, fdn_stream_t stream) {
    FDN_CHECK_ARG(a && b && ws && out_sum && C > 0 && H > 0 && W > 0 && max_value > 0.f);
    const int crop = replicate_no_crop ? 0 : 5;
    FDN_CHECK_ARG(H > 2 * crop && W > 2 * crop && H >= 6 && W >= 6);                  // reflect-101 of a 5-pixel apron needs >= 6 pixels
    G11d g;
    double sum = 0.0;
    for (int i = 0; i < 11; ++i) { g.w[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += g.w[i]; }
    for (int i = 0; i < 11; ++i) g.w[i] /= sum;
    const long n = (long)C * H * W;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ssim2d_rows_kernel, dim3((unsigned)cdiv(n, 256L)), dim3(256), 0, s, a, b, ws, n, W, g, replicate_no_crop);
    const double C1 = (0.01 * (double)max_value) * (0.01 * (double)max_value), C2 = (0.03 * (double)max_value) * (0.03 * (double)max_value);
    long rb = cdiv(n, 256L * 4);
    if (rb > 4096) rb = 4096;
    hipLaunchKernelGGL(ssim2d_cols_kernel, dim3((unsigned)rb), dim3(256), 0, s, (const double*)ws, n, H, W, g, replicate_no_crop, crop, C1, C2,
                       out_sum);
    return fdn_launch_status();
}
