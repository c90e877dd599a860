// Dense k x k convolutions, resampling and the small MAR / LPNet helpers of the FDN path.
//   fdn_conv2d          : direct dense conv (any k, stride, pad; groups=1) with bias / residual /
//                         activation epilogue: OverlapPatchEmbed, Downsample, Upsample, output conv
//                         (FDN_arch.py:704,720,731,804), MAR's 3x3 / stride-2 convs (:57,:135,
//                         :174-175,:192-193,:196), LPNet's 7x7 s2, 3x3 and strided 1x1 convs
//                         (LPNet_arch.py:49-62,:91)
//   fdn_conv_transpose4x4s2 : ConvTranspose2d(k=4,s=2,p=1) + LeakyReLU (FDN_arch.py:21-23,:194-195)
//   fdn_upconv_gather     : the tap sum of Upsample's 3x3 conv over the bilinear x2 image, from the per-tap 1x1 products at low resolution (:726-734)
//   fdn_resample        : bilinear 1/2 and x2 (align_corners=False), nearest 1/2 and x2,
//                         PixelUnshuffle (FDN_arch.py:199-206,:230-233,:719,:730,:866)
// Thread = one output pixel (lanes contiguous along W), OCB output channels in registers; the
// weight address is wave-uniform so it rides the scalar cache.
#include "common.hpp"

// MFMA implicit-GEMM path for 3x3 / stride 1 / pad 1 (conv3x3.hip); FDN_ERR_UNSUPPORTED = not covered
int fdn_conv3x3_mfma(const float* x, const float* w, const float* bias, const float* res, float* out, int B, int Cin, int H,
                     int W, int Cout, int act, int res_before_act, float post_add, hipStream_t s);

namespace {

constexpr int OCB = 8;

struct ConvArgs {
    const float* x; const float* w; const float* bias; const float* res; float* out;
    int B, Cin, H, W, Cout, OH, OW, KH, KW, stride, pad;
    int act; int res_before_act; float post_add;
};

__global__ __launch_bounds__(256) void conv2d_kernel(ConvArgs a) {
    unsigned bx, by, bz;
    fdn_xcd_block3(bx, by, bz);                         // every XCD walks a contiguous run of pixel blocks: halo rows stay in one L2
    const long op = (long)bx * 256 + threadIdx.x;
    const int oc0 = by * OCB, b = bz;
    const long OP = (long)a.OH * a.OW;
    const bool live = op < OP;
    const int oy = live ? (int)(op / a.OW) : 0, ox = live ? (int)(op - (long)oy * a.OW) : 0;
    float acc[OCB];
#pragma unroll
    for (int o = 0; o < OCB; ++o) acc[o] = 0.f;
    const int nvalid = min(OCB, a.Cout - oc0);
    const long hw = (long)a.H * a.W;
    const int kk = a.KH * a.KW;
    const float* xb = a.x + (long)b * a.Cin * hw;
    for (int ci = 0; ci < a.Cin; ++ci) {
        const float* xc = xb + (long)ci * hw;
        for (int ky = 0; ky < a.KH; ++ky) {
            const int iy = oy * a.stride - a.pad + ky;
            const bool yok = iy >= 0 && iy < a.H;
            for (int kx = 0; kx < a.KW; ++kx) {
                const int ix = ox * a.stride - a.pad + kx;
                const float v = (live && yok && ix >= 0 && ix < a.W) ? xc[(long)iy * a.W + ix] : 0.f;
                const float* wp = a.w + ((long)oc0 * a.Cin + ci) * kk + ky * a.KW + kx;
#pragma unroll
                for (int o = 0; o < OCB; ++o)
                    if (o < nvalid) acc[o] = fmaf(v, wp[(long)o * a.Cin * kk], acc[o]);
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int o = 0; o < OCB; ++o) {
        if (o >= nvalid) break;
        const int oc = oc0 + o;
        float v = acc[o] + (a.bias ? a.bias[oc] : 0.f);
        const long oi = ((long)b * a.Cout + oc) * OP + op;
        if (a.res && a.res_before_act) v += a.res[oi];
        v = apply_act(v, a.act);
        if (a.res && !a.res_before_act) v += a.res[oi];
        a.out[oi] = v + a.post_add;
    }
}

// 3x3 / pad 1 / stride S specialisation for the narrow convs (Cout < 8 or stride 2; the wide stride-1 ones go
// to the MFMA kernel): the OCB-channel weight slice sits in LDS as [Cin][9][OCB] (one broadcast b128 read
// per 4 channels), the nine tap offsets and their validity are computed once per thread, and the nine loads
// of an input channel are issued together - the generic kernel above pays one global + OCB scalar round trips
// per tap.
template <int S, int OCB_>
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    unsigned bx, by, bz;
    fdn_xcd_block3(bx, by, bz);
    const int oc0 = by * OCB_, b = bz;
    const int nvalid = min(OCB_, a.Cout - oc0);
    for (int i = threadIdx.x; i < a.Cin * 9 * OCB_; i += 256) {
        const int o = i % OCB_, t = (i / OCB_) % 9, ci = i / (OCB_ * 9);
        wl[i] = o < nvalid ? a.w[((long)(oc0 + o) * a.Cin + ci) * 9 + t] : 0.f;
    }
    __syncthreads();
    const long op = (long)bx * 256 + threadIdx.x;
    const long OP = (long)a.OH * a.OW;
    const bool live = op < OP;
    const int oy = live ? (int)(op / a.OW) : 0, ox = live ? (int)(op - (long)oy * a.OW) : 0;
    int off[9];
    unsigned ok = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = oy * S - 1 + t / 3, ix = ox * S - 1 + t % 3;
        const bool v = live && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        off[t] = v ? iy * a.W + ix : 0;
        ok |= (v ? 1u : 0u) << t;
    }
    float acc[OCB_];
#pragma unroll
    for (int o = 0; o < OCB_; ++o) acc[o] = 0.f;
    const long hw = (long)a.H * a.W;
    const float* xc = a.x + (long)b * a.Cin * hw;
    for (int ci = 0; ci < a.Cin; ++ci, xc += hw) {
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = xc[off[t]];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float vv = ((ok >> t) & 1u) ? v[t] : 0.f;
            const float* wp = wl + (ci * 9 + t) * OCB_;
#pragma unroll
            for (int o = 0; o < OCB_; ++o) acc[o] = fmaf(vv, wp[o], acc[o]);
        }
    }
    if (!live) return;
#pragma unroll
    for (int o = 0; o < OCB_; ++o) {
        if (o >= nvalid) break;
        const int oc = oc0 + o;
        float v = acc[o] + (a.bias ? a.bias[oc] : 0.f);
        const long oi = ((long)b * a.Cout + oc) * OP + op;
        if (a.res && a.res_before_act) v += a.res[oi];
        v = apply_act(v, a.act);
        if (a.res && !a.res_before_act) v += a.res[oi];
        a.out[oi] = v + a.post_add;
    }
}

template <int S, int OCB_>
int launch_conv3x3_direct(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)a.Cin * 9 * OCB_ * sizeof(float);
    if (lds > 64 * 1024) return FDN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv3x3_direct_kernel<S, OCB_>), dim3(cdiv((long)a.OH * a.OW, 256), cdiv(a.Cout, OCB_), a.B), dim3(256),
                       lds, s, a);
    return fdn_launch_status();
}

// ConvTranspose2d(Cin, Cout, 4, stride 2, padding 1): weight [Cin][Cout][4][4]; OH = 2H, OW = 2W.
// stride 2, pad 1, k 4: output row oy takes input rows iy with 2*iy = oy + 1 - ky, i.e. the two taps
// ky = (oy+1)&1 and ky + 2 (same for columns): 4 of the 16 taps are live per output pixel.  The OCB-channel
// weight slice sits in LDS as [Cin][16][OCB]; the four loads of an input channel are issued together.
template <int OCB_>
__global__ __launch_bounds__(256) void convT_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ out, int Cin, int H,
                                                    int W, int Cout, int act) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    unsigned bx, by, bz;
    fdn_xcd_block3(bx, by, bz);
    const int oc0 = by * OCB_, b = bz;
    const int nvalid = min(OCB_, Cout - oc0);
    for (int i = threadIdx.x; i < Cin * 16 * OCB_; i += 256) {
        const int o = i % OCB_, t = (i / OCB_) % 16, ci = i / (OCB_ * 16);
        wl[i] = o < nvalid ? w[((long)ci * Cout + oc0 + o) * 16 + t] : 0.f;
    }
    __syncthreads();
    const int OH = 2 * H, OW = 2 * W;
    const long op = (long)bx * 256 + threadIdx.x;
    const long OP = (long)OH * OW;
    const bool live = op < OP;
    const int oy = live ? (int)(op / OW) : 0, ox = live ? (int)(op - (long)oy * OW) : 0;
    const int ky0 = (oy + 1) & 1, kx0 = (ox + 1) & 1;
    int off[4], tap[4];
    unsigned ok = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ky = ky0 + 2 * (j >> 1), kx = kx0 + 2 * (j & 1);
        const int ny = oy + 1 - ky, nx = ox + 1 - kx;
        const int iy = ny >> 1, ix = nx >> 1;
        const bool v = live && ny >= 0 && iy < H && nx >= 0 && ix < W;
        off[j] = v ? iy * W + ix : 0;
        tap[j] = (ky * 4 + kx) * OCB_;
        ok |= (v ? 1u : 0u) << j;
    }
    float acc[OCB_];
#pragma unroll
    for (int o = 0; o < OCB_; ++o) acc[o] = 0.f;
    const long hw = (long)H * W;
    const float* xc = x + (long)b * Cin * hw;
    for (int ci = 0; ci < Cin; ++ci, xc += hw) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = xc[off[j]];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float vv = ((ok >> j) & 1u) ? v[j] : 0.f;
            const float* wp = wl + ci * 16 * OCB_ + tap[j];
#pragma unroll
            for (int o = 0; o < OCB_; ++o) acc[o] = fmaf(vv, wp[o], acc[o]);
        }
    }
    if (!live) return;
#pragma unroll
    for (int o = 0; o < OCB_; ++o) {
        if (o >= nvalid) break;
        out[((long)b * Cout + oc0 + o) * OP + op] = apply_act(acc[o] + (bias ? bias[oc0 + o] : 0.f), act);
    }
}

// (round 5) The same transposed conv with a thread per INPUT pixel: its 2 x 2 output quad needs the 3 x 3 input window around it and every one of the
// 16 taps exactly once, so the tap of an FMA no longer depends on the lane's output parity - the weights become wave-uniform LDS broadcasts
// ([ci][tap][OCB] rows of 16 bytes) instead of per-lane gathers, no lane multiplies a masked zero, and a lane stores 8-byte row pairs.  Each output
// sums its four taps in the order of convT_kernel (ky0, kx0), (ky0, kx0 + 2), (ky0 + 2, kx0), (ky0 + 2, kx0 + 2): same bits.
// MAR's two up-convs (24 -> 12 at 368 x 640, 48 -> 24 at 184 x 320): 0.90 + 0.31 ms with convT_kernel.
template <int OCB_>
__global__ __launch_bounds__(256) void convT_quad_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int Cin, int H, int W, int Cout, int act) {
    // a thread takes TWO horizontally adjacent input pixels (ix0, ix0 + 1; W is even on this path): a 3 x 4 input window, two output quads, and
    // every weight row read from LDS feeds two FMAs - with one pixel per thread the kernel was bound by its 48 broadcast reads of 16 bytes per input
    // channel (the CU's one LDS pipe returns 1 KB per wave and read: 1.07 ms for MAR's two up-convs where the vector work needs ~0.3 ms)
    static_assert(OCB_ % 4 == 0, "weight rows are read as 16-byte words");
    extern __shared__ __attribute__((aligned(16))) float wl[];          // [ci][tap = ky * 4 + kx][OCB_]
    unsigned bx, by, bz;
    fdn_xcd_block3(bx, by, bz);
    const int oc0 = by * OCB_, b = bz;
    const int nvalid = min(OCB_, Cout - oc0);
    for (int i = threadIdx.x; i < Cin * 16 * OCB_; i += 256) {
        const int o = i % OCB_, t = (i / OCB_) % 16, ci = i / (OCB_ * 16);
        wl[i] = o < nvalid ? w[((long)ci * Cout + oc0 + o) * 16 + t] : 0.f;
    }
    __syncthreads();
    const long hw = (long)H * W;
    const int W2 = W >> 1;
    const long ip = (long)bx * 256 + threadIdx.x;                          // pixel PAIR index
    const bool live = ip < (long)H * W2;
    const int iy = live ? (int)(ip / W2) : 0, ix0 = live ? 2 * (int)(ip - (long)iy * W2) : 0;
    // the 3 x 4 window: offsets relative to the plane, 0 and a cleared flag outside the image
    int off[12];
    unsigned ok = 0;
#pragma unroll
    for (int d = 0; d < 12; ++d) {
        const int y = iy - 1 + d / 4, xx = ix0 - 1 + d % 4;
        const bool v = live && y >= 0 && y < H && xx >= 0 && xx < W;
        off[d] = v ? y * W + xx : 0;
        ok |= (v ? 1u : 0u) << d;
    }
    float acc[2][4][OCB_];                                                 // [pixel of the pair][2 py + px][channel]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int o = 0; o < OCB_; ++o) acc[u][q][o] = 0.f;
    const float* xc = x + (long)b * Cin * hw;
    for (int ci = 0; ci < Cin; ++ci, xc += hw) {
        float v[12];
#pragma unroll
        for (int d = 0; d < 12; ++d) v[d] = xc[off[d]];
#pragma unroll
        for (int d = 0; d < 12; ++d) v[d] = ((ok >> d) & 1u) ? v[d] : 0.f;
        const float* wc = wl + ci * 16 * OCB_;
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int ky0 = py ? 0 : 1, kx0 = px ? 0 : 1;             // (oy + 1) & 1 with oy = 2 iy + py
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ky = ky0 + 2 * (j >> 1), kx = kx0 + 2 * (j & 1);
                    // oy = 2 iy' - 1 + ky  =>  iy' = iy + (py + 1 - ky) / 2: window row 1 + (py + 1 - ky) / 2 (exact: the difference is even)
                    const int wy = 1 + (py + 1 - ky) / 2, wx = 1 + (px + 1 - kx) / 2;
                    const float* wp = wc + (ky * 4 + kx) * OCB_;
                    const float va = v[wy * 4 + wx], vb = v[wy * 4 + wx + 1];
#pragma unroll
                    for (int o = 0; o < OCB_; ++o) {
                        const float ww = wp[o];
                        acc[0][2 * py + px][o] = fmaf(va, ww, acc[0][2 * py + px][o]);
                        acc[1][2 * py + px][o] = fmaf(vb, ww, acc[1][2 * py + px][o]);
                    }
                }
            }
    }
    if (!live) return;
    const int OW = 2 * W;
    const long OP = 4 * hw;
#pragma unroll
    for (int o = 0; o < OCB_; ++o) {
        if (o < nvalid) {                                                  // (no `break`: the accumulators keep static register indices)
            const float bb = bias ? bias[oc0 + o] : 0.f;
            float* op = out + ((long)b * Cout + oc0 + o) * OP + (long)(2 * iy) * OW + 2 * ix0;
            *reinterpret_cast<float4*>(op) = make_float4(apply_act(acc[0][0][o] + bb, act), apply_act(acc[0][1][o] + bb, act),
                                                         apply_act(acc[1][0][o] + bb, act), apply_act(acc[1][1][o] + bb, act));
            *reinterpret_cast<float4*>(op + OW) = make_float4(apply_act(acc[0][2][o] + bb, act), apply_act(acc[0][3][o] + bb, act),
                                                              apply_act(acc[1][2][o] + bb, act), apply_act(acc[1][3][o] + bb, act));
        }
    }
}

__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, float* __restrict__ out, long planes, int H,
                                                       int W, int OH, int OW, int mode, int r) {
    // grid = (column blocks, output rows, output planes): no per-element division (the flat index form spent ~100 instructions
    // per element on two 64-bit divisions: 82 % vector-ALU-busy for a copy kernel)
    const int ox = blockIdx.x * 256 + threadIdx.x;
    if (ox >= OW) return;
    const int oy = blockIdx.y;
    const long pl = blockIdx.z;
    const long idx = (pl * OH + oy) * OW + ox;
    if (mode == FDN_RS_BILINEAR_HALF) {
        const float* s = x + pl * H * W + (long)(2 * oy) * W + 2 * ox;
        // area_pixel source index 2*o+0.5: lambda = 0.5 on both axes (exact 2x2 mean)
        out[idx] = 0.5f * (0.5f * s[0] + 0.5f * s[1]) + 0.5f * (0.5f * s[W] + 0.5f * s[W + 1]);
    } else if (mode == FDN_RS_BILINEAR_X2) {
        float sy = fmaxf(0.5f * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (ox + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = sy - y0, lx = sx - x0;
        const float* s = x + pl * H * W;
        out[idx] = (1.f - ly) * ((1.f - lx) * s[(long)y0 * W + x0] + lx * s[(long)y0 * W + x1]) +
                   ly * ((1.f - lx) * s[(long)y1 * W + x0] + lx * s[(long)y1 * W + x1]);
    } else if (mode == FDN_RS_NEAREST_HALF) {
        out[idx] = x[pl * H * W + (long)(2 * oy) * W + 2 * ox];
    } else if (mode == FDN_RS_NEAREST_X2) {
        out[idx] = x[pl * H * W + (long)(oy >> 1) * W + (ox >> 1)];
    } else {  // PixelUnshuffle(r): out plane = (b*C + c)*r*r + i*r + j ; planes counts output planes
        const long inpl = pl / (r * r);
        const int ij = (int)(pl - inpl * r * r);
        const int i = ij / r, j = ij - i * r;
        out[idx] = x[inpl * H * W + (long)(oy * r + i) * W + ox * r + j];
    }
}

// Four outputs per thread for the bilinear modes and nearest x2 (16-byte stores, the source window loaded once): the same arithmetic
// per value as resample_kernel, so the results are bit-identical to it.  grid = ((output rows x column quads) / 256, planes).
template <int MODE>
__global__ __launch_bounds__(256) void resample4_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int OH,
                                                        int OW) {
    // (row, quad) flattened over the threads of a plane: a row of OW / 4 quads rarely fills whole 256-thread blocks
    const int OWq = OW >> 2;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (unsigned)(OH * OWq)) return;
    const int oy = idx < (1u << 22) ? (int)(((float)idx + 0.5f) * (1.0f / (float)OWq)) : (int)(idx / (unsigned)OWq);   // exact below 2^22
    const int t = (int)idx - oy * OWq;
    const long pl = blockIdx.y;
    const float* s = x + pl * H * W;
    float4 o;
    if (MODE == FDN_RS_NEAREST_X2) {
        const float2 v = *reinterpret_cast<const float2*>(s + (long)(oy >> 1) * W + 2 * t);
        o = make_float4(v.x, v.x, v.y, v.y);
    } else if (MODE == FDN_RS_BILINEAR_HALF) {
        const float4 a0 = *reinterpret_cast<const float4*>(s + (long)(2 * oy) * W + 8 * t);
        const float4 a1 = *reinterpret_cast<const float4*>(s + (long)(2 * oy) * W + 8 * t + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(s + (long)(2 * oy + 1) * W + 8 * t);
        const float4 b1 = *reinterpret_cast<const float4*>(s + (long)(2 * oy + 1) * W + 8 * t + 4);
        o.x = 0.5f * (0.5f * a0.x + 0.5f * a0.y) + 0.5f * (0.5f * b0.x + 0.5f * b0.y);
        o.y = 0.5f * (0.5f * a0.z + 0.5f * a0.w) + 0.5f * (0.5f * b0.z + 0.5f * b0.w);
        o.z = 0.5f * (0.5f * a1.x + 0.5f * a1.y) + 0.5f * (0.5f * b1.x + 0.5f * b1.y);
        o.w = 0.5f * (0.5f * a1.z + 0.5f * a1.w) + 0.5f * (0.5f * b1.z + 0.5f * b1.w);
    } else {
        // align_corners=False x2: source x of output 4t + j is 2t + (2j - 1) / 4; the four outputs read columns 2t-1 .. 2t+2
        const float sy = fmaxf(0.5f * (oy + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)sy, y1 = min(y0 + 1, H - 1);
        const float ly = sy - y0;
        const float* r0 = s + (long)y0 * W;
        const float* r1 = s + (long)y1 * W;
        const int cm = max(2 * t - 1, 0), c0 = 2 * t, c1 = min(2 * t + 1, W - 1), c2 = min(2 * t + 2, W - 1);
        const float am = r0[cm], a0 = r0[c0], a1 = r0[c1], a2 = r0[c2];
        const float bm = r1[cm], b0 = r1[c0], b1 = r1[c1], b2 = r1[c2];
        auto mix = [&](float lx, float p, float q, float u, float v) { return (1.f - ly) * ((1.f - lx) * p + lx * q) + ly * ((1.f - lx) * u + lx * v); };
        o.x = t == 0 ? mix(0.f, a0, a1, b0, b1) : mix(0.75f, am, a0, bm, b0);      // (output 0 of a row: source x clamps to 0)
        o.y = mix(0.25f, a0, a1, b0, b1);
        o.z = mix(0.75f, a0, a1, b0, b1);
        o.w = mix(0.25f, a1, a2, b1, b2);
    }
    *reinterpret_cast<float4*>(out + (pl * OH + oy) * OW + 4 * t) = o;
}

// ------------------------------------------------------------------------------------------------
// Upsample (FDN_arch.py:726-734): Conv2d(C, C / 2, 3, padding 1) of the bilinear x2 image, WITHOUT the x2 image.  The conv's channel
// contraction commutes with the (per-channel, linear) upsampling: conv(up(x))[co] = sum over the nine taps of shift_tap(up(z_tap[co])) with
// z_tap = W_tap x, a 1x1 conv at LOW resolution (fdn_conv1x1, 9 Cout output channels: a quarter of the 3x3 conv's matrix work).  This
// kernel is the rest: thread = one low-resolution pixel (i, j) = one 2 x 2 output quad, COB output channels.  An output row 2i + p reads,
// through tap d, the x2 row 2i + p + d - 1 - one of FOUR rows (2i - 1 .. 2i + 2), each a two-tap blend of the low-resolution rows
// i - 1, i, i + 1 (align_corners = False: 0.75 / 0.25, the source index clamped at the image edge) or nothing at all (the conv's zero
// padding, outside the x2 image).  V[r][a] are those blends for the rows, U[c][b] for the columns:
//     out[co][2i + p][2j + q] = sum_{d, e} sum_{a, b} V[p + d][a] U[q + e][b] z[(3d + e) Cout + co][i + a][j + b].
// ------------------------------------------------------------------------------------------------
template <int COB>
__global__ __launch_bounds__(256, 2) void upconv_gather_kernel(const float* __restrict__ z, float* __restrict__ out, int Cout, int h, int w) {
    // every XCD walks a contiguous run of pixel blocks: a block's source rows i - 1 .. i + 1 are its neighbours' too:
    // 0.89 -> 0.83 ms at level 1.  (Two quads per thread - 8-byte loads, 16-byte stores, half the load instructions - measured slower: 1.05 ms)
    const unsigned idx = xcd_contiguous(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
    if (idx >= (unsigned)(h * w)) return;
    const int i = (int)(idx / (unsigned)w), j = (int)idx - i * w;
    const int co0 = blockIdx.y * COB, b = blockIdx.z;
    const long p_lo = (long)h * w, p_hi = 4 * p_lo;
    // blends of the four x2 rows / columns around the quad over the low-resolution offsets -1, 0, +1
    float V[4][3], U[4][3];
    {
        const bool t = i == 0, e_ = i == h - 1, l = j == 0, r = j == w - 1;
        V[0][0] = t ? 0.f : 0.75f; V[0][1] = t ? 0.f : 0.25f; V[0][2] = 0.f;                  // x2 row 2i - 1 (outside the image at i = 0)
        V[1][0] = t ? 0.f : 0.25f; V[1][1] = t ? 1.f : 0.75f; V[1][2] = 0.f;                  // 2i: rows i - 1 (clamped), i
        V[2][0] = 0.f; V[2][1] = e_ ? 1.f : 0.75f; V[2][2] = e_ ? 0.f : 0.25f;                // 2i + 1: rows i, i + 1 (clamped)
        V[3][0] = 0.f; V[3][1] = e_ ? 0.f : 0.25f; V[3][2] = e_ ? 0.f : 0.75f;                // 2i + 2 (outside at i = h - 1)
        U[0][0] = l ? 0.f : 0.75f; U[0][1] = l ? 0.f : 0.25f; U[0][2] = 0.f;
        U[1][0] = l ? 0.f : 0.25f; U[1][1] = l ? 1.f : 0.75f; U[1][2] = 0.f;
        U[2][0] = 0.f; U[2][1] = r ? 1.f : 0.75f; U[2][2] = r ? 0.f : 0.25f;
        U[3][0] = 0.f; U[3][1] = r ? 0.f : 0.25f; U[3][2] = r ? 0.f : 0.75f;
    }
    const int ro[3] = {max(i - 1, 0) * w, i * w, min(i + 1, h - 1) * w};
    const int cl[3] = {max(j - 1, 0), j, min(j + 1, w - 1)};
    const float* zb = z + (long)b * 9 * Cout * p_lo;
    float* ob = out + (long)b * Cout * p_hi + (long)(2 * i) * (2 * w) + 2 * j;
    // (two channels in flight per thread - 98 loads - at two waves per SIMD: 0.83 -> 0.78 ms; more waves with fewer registers lose: 1.02 ms at four)
#pragma unroll 2
    for (int c = 0; c < COB; ++c) {
        const int co = co0 + c;
        if (co >= Cout) break;
        float o00 = 0.f, o01 = 0.f, o10 = 0.f, o11 = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const float* zp = zb + (long)((3 * d + e) * Cout + co) * p_lo;
                // tap d touches the x2 rows d (p = 0) and d + 1 (p = 1): low-resolution offsets {-1, 0}, {-1, 0, 1}, {0, 1} for d = 0, 1, 2
#pragma unroll
                for (int a = (d == 2 ? 1 : 0); a < (d == 0 ? 2 : 3); ++a) {
                    float t0 = 0.f, t1 = 0.f;
#pragma unroll
                    for (int bb = (e == 2 ? 1 : 0); bb < (e == 0 ? 2 : 3); ++bb) {
                        const float v = zp[ro[a] + cl[bb]];
                        t0 = fmaf(U[e][bb], v, t0);
                        t1 = fmaf(U[e + 1][bb], v, t1);
                    }
                    o00 = fmaf(V[d][a], t0, o00);
                    o01 = fmaf(V[d][a], t1, o01);
                    o10 = fmaf(V[d + 1][a], t0, o10);
                    o11 = fmaf(V[d + 1][a], t1, o11);
                }
            }
        float* op = ob + (long)co * p_hi;
        *reinterpret_cast<float2*>(op) = make_float2(o00, o01);
        *reinterpret_cast<float2*>(op + 2 * w) = make_float2(o10, o11);
    }
}

// fourier_fuse.fpre[1]: Conv2d(n, n, 1, padding=1, groups=n): (H+2)x(W+2) map, bias-only border
__global__ __launch_bounds__(256) void dw1x1_pad1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out, int C, int H,
                                                         int W) {
    // grid = (column blocks, output rows, planes): no per-element division (cf. resample_kernel)
    const int OW = W + 2, OH = H + 2;
    const int ox = blockIdx.x * 256 + threadIdx.x;
    if (ox >= OW) return;
    const int oy = blockIdx.y;
    const long pl = blockIdx.z;
    const int c = (int)(pl % C);
    const bool in = oy >= 1 && oy <= H && ox >= 1 && ox <= W;
    out[(pl * OH + oy) * OW + ox] = (in ? w[c] * x[pl * H * W + (long)(oy - 1) * W + ox - 1] : 0.f) + bias[c];
}

// AvgPool2d(3, stride 2, padding 1), count_include_pad=True (LPNet_arch.py:94)
__global__ __launch_bounds__(256) void avgpool3s2_kernel(const float* __restrict__ x, float* __restrict__ out, long planes,
                                                         int H, int W, int OH, int OW) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * OH * OW) return;
    const int ox = (int)(idx % OW);
    const long t = idx / OW;
    const int oy = (int)(t % OH);
    const float* s = x + (t / OH) * H * W;
    float a = 0.f;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int y = 2 * oy + dy, xx = 2 * ox + dx;
            if (y >= 0 && y < H && xx >= 0 && xx < W) a += s[(long)y * W + xx];
        }
    out[idx] = a / 9.0f;
}

// mean over H*W of each plane (AdaptiveAvgPool2d(1)); one workgroup per plane
__global__ __launch_bounds__(256) void gap_kernel(const float* __restrict__ x, float* __restrict__ out, long P) {
    __shared__ float red[256];
    const float* s = x + (long)blockIdx.x * P;
    float a = 0.f;
    for (long i = threadIdx.x; i < P; i += 256) a += s[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] / (float)P;
}

// SE tail: relu(y * gate[plane] + shortcut)  (LPNet_arch.py:75-80)
__global__ __launch_bounds__(256) void se_apply_kernel(const float* __restrict__ y, const float* __restrict__ gate,
                                                       const float* __restrict__ sc, float* __restrict__ out, long P,
                                                       long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    out[idx] = fmaxf(fmaf(y[idx], gate[idx / P], sc[idx]), 0.f);
}

// x *= ratio[b]   (MAR_archa.forward, FDN_arch.py:213-219)
__global__ __launch_bounds__(256) void scale_batch_kernel(float* __restrict__ x, const float* __restrict__ ratio, long per_b,
                                                          long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < total) x[idx] *= ratio[idx / per_b];
}

// The per-bin MLPs of MAR's Fourier blocks (FDN_arch.py:93-94, :142-143): mag <- W2m lrelu(W1m mag + b1m) + b2m, pha likewise, in place, both in
// one launch.  As four fdn_conv1x1 launches these cost 0.43 ms per level-1 FreBlock (48 planes read and 48 written twice over, on MFMA tiles a
// quarter full at C = 12); here a thread owns one bin: its C values live in registers, every plane is read and written once - 4 C^2 FMAs per
// bin against 16 C bytes, i.e. the HBM rate of 4 C planes.  The weights wait in LDS ([W1 | W2 transposed | b1 | b2] per MLP, read at wave-uniform
// addresses = broadcasts of 16 bytes): as scalar operands they need 2 C registers per hidden unit, and the compiler spilled 322 of them at C = 48.
template <int C>
__global__ __launch_bounds__(256) void spectral_mlp2_kernel(float* __restrict__ mag, float* __restrict__ pha, const float* __restrict__ w1m,
                                                            const float* __restrict__ b1m, const float* __restrict__ w2m,
                                                            const float* __restrict__ b2m, const float* __restrict__ w1p,
                                                            const float* __restrict__ b1p, const float* __restrict__ w2p,
                                                            const float* __restrict__ b2p, long P, float slope) {
    constexpr int WS = 2 * C * C + 2 * C;                            // floats per MLP
    __shared__ __attribute__((aligned(16))) float wsh[2 * WS];
    for (int i = threadIdx.x; i < 2 * WS; i += 256) {
        const int m = i / WS, r = i - m * WS;
        const float* w1 = m ? w1p : w1m; const float* w2 = m ? w2p : w2m; const float* b1 = m ? b1p : b1m; const float* b2 = m ? b2p : b2m;
        float v;
        if (r < C * C) v = w1[r];                                    // W1[j][i]
        else if (r < 2 * C * C) { const int q = r - C * C, j = q / C, k = q - j * C; v = w2[k * C + j]; }      // W2 transposed: [j][k]
        else if (r < 2 * C * C + C) v = b1[r - 2 * C * C];
        else v = b2[r - 2 * C * C - C];
        wsh[i] = v;
    }
    __syncthreads();
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const long base = (long)blockIdx.y * C * P + p;
    auto mlp = [&](float* __restrict__ t, const float* __restrict__ ws) __attribute__((always_inline)) {
        const float* w1 = ws, *w2t = ws + C * C, *b1 = ws + 2 * C * C, *b2 = b1 + C;
        float v[C], o[C];
#pragma unroll
        for (int i = 0; i < C; ++i) v[i] = t[base + (long)i * P];
#pragma unroll
        for (int k = 0; k < C; ++k) o[k] = b2[k];
#pragma unroll 1
        for (int j = 0; j < C; ++j) {
            float s = b1[j];
#pragma unroll
            for (int i = 0; i < C; ++i) s = fmaf(w1[j * C + i], v[i], s);
            const float h = s > 0.f ? s : s * slope;
#pragma unroll
            for (int k = 0; k < C; ++k) o[k] = fmaf(w2t[j * C + k], h, o[k]);
        }
#pragma unroll
        for (int k = 0; k < C; ++k) t[base + (long)k * P] = o[k];
    };
    mlp(mag, wsh);
    mlp(pha, wsh + WS);
}

// gamma curve 1 - (1 - x)^(40 * i)   (MAR.forward, FDN_arch.py:282-284)
__global__ __launch_bounds__(256) void gamma_curve_kernel(const float* __restrict__ x, const float* __restrict__ im,
                                                          float* __restrict__ out, float scale, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < total) out[idx] = 1.0f - powf(1.0f - x[idx], im[idx] * scale);
}

// 7 x 7 / stride 2 / pad 3 with few input channels (LPNet's entry conv 3 -> 16, LPNet_arch.py:91; round 4): an 8 x 32 output tile per
// workgroup, its (2 * 8 + 5) x (2 * 32 + 5) input tile of every channel staged in LDS - columns split by parity, so the stride-2 reads of a
// row of lanes are consecutive words - and the weights as [ci][ky][kx][COUT] rows read as wave-uniform (broadcast) 16-byte words: one LDS
// word + COUT / 4 broadcast reads per tap for COUT FMAs, every input element fetched from memory once per tile (the generic kernel above
// issues a global load and OCB scalar loads per tap and reads the input once per 8 output channels: 0.98 ms, 3.4x its bytes).
// Accumulation order (ci, ky, kx) as the generic kernel: same sums, bit for bit.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv7x7s2_kernel(ConvArgs a) {
    constexpr int TR = 8, TCW = 32;                       // output tile
    constexpr int IR = 2 * TR + 5, IC = 2 * TCW + 5;      // input tile 21 x 69
    constexpr int HC = (IC + 1) / 2;                      // 35 columns per parity
    __shared__ float tile[CIN][IR][2][HC + 1];
    __shared__ __attribute__((aligned(16))) float wl[CIN * 49 * COUT];
    unsigned bx, by, bz;
    fdn_xcd_block3(bx, by, bz);
    const int tid = threadIdx.x, b = bz;
    const int oy0 = by * TR, ox0 = bx * TCW;
    for (int i = tid; i < CIN * 49 * COUT; i += 256) {
        const int o = i % COUT, t = i / COUT;             // t = ci * 49 + ky * 7 + kx
        wl[i] = a.w[(long)o * CIN * 49 + t];
    }
    const long hw = (long)a.H * a.W;
    const float* xb = a.x + (long)b * CIN * hw;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    for (int i = tid; i < CIN * IR * IC; i += 256) {
        const int c = i % IC, r = (i / IC) % IR, ci = i / (IC * IR);
        const int iy = iy0 + r, ix = ix0 + c;
        tile[ci][r][c & 1][c >> 1] = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? xb[(long)ci * hw + (long)iy * a.W + ix] : 0.f;
    }
    __syncthreads();
    const int ty = tid >> 5, tx = tid & 31;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
#pragma unroll 1
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float v = tile[ci][2 * ty + ky][kx & 1][tx + (kx >> 1)];
                const float* wp = wl + ((ci * 7 + ky) * 7 + kx) * COUT;
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] = fmaf(v, wp[o], acc[o]);
            }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= a.OH || ox >= a.OW) return;
    const long OP = (long)a.OH * a.OW, op = (long)oy * a.OW + ox;
#pragma unroll
    for (int oc = 0; oc < COUT; ++oc) {
        float v = acc[oc] + (a.bias ? a.bias[oc] : 0.f);
        const long oi = ((long)b * COUT + oc) * OP + op;
        if (a.res && a.res_before_act) v += a.res[oi];
        v = apply_act(v, a.act);
        if (a.res && !a.res_before_act) v += a.res[oi];
        a.out[oi] = v + a.post_add;
    }
}

}  // namespace

extern "C" int fdn_conv2d(const float* x, const float* w, const float* bias, const float* res, float* out, int B, int Cin,
                          int H, int W, int Cout, int KH, int KW, int stride, int pad, int act, int res_before_act,
                          float post_add, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && w && out && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.res = res; a.out = out;
    a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.OH = (H + 2 * pad - KH) / stride + 1;
    a.OW = (W + 2 * pad - KW) / stride + 1;
    FDN_CHECK_ARG(a.OH > 0 && a.OW > 0);
    a.act = act; a.res_before_act = res_before_act; a.post_add = post_add;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (KH == 3 && KW == 3 && pad == 1 && (stride == 1 || stride == 2)) {
        // measured at B=8 720p (tools/ab_conv3x3.py, tools/gpu_conv_shapes.py): Cin % 8 == 0 with at least 16 output channels
        // goes to the LDS-tiled split-bf16 MFMA form (64->32 @L1 2.3 ms against 5.5 ms direct, 24->24 0.21 against 0.89 ms);
        // the LDS-weight direct kernel keeps the narrow ones (12->12 0.47 ms against 2.0 ms on the flat MFMA form)
        int rc = FDN_ERR_UNSUPPORTED;
        if (stride == 1 && Cin % 8 == 0 && Cin >= 16 && Cout >= 16 && !fdn_matrix_pipe_f32()) {          // LDS-tiled MFMA form (split-bf16 operands, conv3x3.hip; the f32 gate is repeated inside)
            rc = fdn_conv3x3_mfma(x, w, bias, res, out, B, Cin, H, W, Cout, act, res_before_act, post_add, s);
            if (rc != FDN_ERR_UNSUPPORTED) return rc;
        }
        if (stride == 2) rc = Cout <= 8 ? launch_conv3x3_direct<2, 8>(a, s) : launch_conv3x3_direct<2, 16>(a, s);
        else if (Cout <= 4) rc = launch_conv3x3_direct<1, 4>(a, s);
        else if (Cout <= 8) rc = launch_conv3x3_direct<1, 8>(a, s);
        else if (Cout <= 64) rc = launch_conv3x3_direct<1, 16>(a, s);
        if (rc != FDN_ERR_UNSUPPORTED) return rc;
        if (stride == 1) {
            rc = fdn_conv3x3_mfma(x, w, bias, res, out, B, Cin, H, W, Cout, act, res_before_act, post_add, s);
            if (rc != FDN_ERR_UNSUPPORTED) return rc;
        }
    }
    if (KH == 7 && KW == 7 && stride == 2 && pad == 3 && Cin == 3 && Cout == 16) {           // LPNet's entry conv
        hipLaunchKernelGGL((conv7x7s2_kernel<3, 16>), dim3(cdiv(a.OW, 32), cdiv(a.OH, 8), B), dim3(256), 0, s, a);
        return fdn_launch_status();
    }
    hipLaunchKernelGGL(conv2d_kernel, dim3(cdiv((long)a.OH * a.OW, 256), cdiv(Cout, OCB), B), dim3(256), 0, s, a);
    return fdn_launch_status();
}

extern "C" int fdn_conv_transpose4x4s2(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int H,
                                       int W, int Cout, int act, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && w && out && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (Cout >= 12 && W % 2 == 0 && (size_t)Cin * 16 * 12 * sizeof(float) <= 48 * 1024 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {      // quad form, 12 channels per block
        hipLaunchKernelGGL(convT_quad_kernel<12>, dim3(cdiv((long)H * (W / 2), 256), cdiv(Cout, 12), B), dim3(256), (size_t)Cin * 16 * 12 * sizeof(float), s,
                           x, w, bias, out, Cin, H, W, Cout, act);
        return fdn_launch_status();
    }
    if (Cout > 8 && (size_t)Cin * 16 * 16 * sizeof(float) <= 64 * 1024) {
        hipLaunchKernelGGL(convT_kernel<16>, dim3(cdiv(4L * H * W, 256), cdiv(Cout, 16), B), dim3(256), (size_t)Cin * 16 * 16 * sizeof(float),
                           s, x, w, bias, out, Cin, H, W, Cout, act);
    } else {
        FDN_CHECK_ARG((size_t)Cin * 16 * 8 * sizeof(float) <= 64 * 1024);
        hipLaunchKernelGGL(convT_kernel<8>, dim3(cdiv(4L * H * W, 256), cdiv(Cout, 8), B), dim3(256), (size_t)Cin * 16 * 8 * sizeof(float), s,
                           x, w, bias, out, Cin, H, W, Cout, act);
    }
    return fdn_launch_status();
}

extern "C" int fdn_resample(const float* x, float* out, long planes, int H, int W, int mode, int r, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && out && planes > 0 && H > 0 && W > 0);
    int OH, OW;
    long oplanes = planes;
    switch (mode) {
        case FDN_RS_BILINEAR_HALF:
        case FDN_RS_NEAREST_HALF: FDN_CHECK_ARG(H % 2 == 0 && W % 2 == 0); OH = H / 2; OW = W / 2; break;
        case FDN_RS_BILINEAR_X2:
        case FDN_RS_NEAREST_X2: OH = 2 * H; OW = 2 * W; break;
        case FDN_RS_PIXEL_UNSHUFFLE: FDN_CHECK_ARG(r > 0 && H % r == 0 && W % r == 0); OH = H / r; OW = W / r; oplanes = planes * r * r; break;
        default: return FDN_ERR_ARG;
    }
    FDN_CHECK_ARG(OH <= 65535);
    // grid.z limit (the FDN path has <= 8 * 128 planes per call): chunks of whole input planes, i.e. a multiple of r * r output
    // planes for PixelUnshuffle - everything is validated before the first launch, a failed call writes nothing
    const long rr = mode == FDN_RS_PIXEL_UNSHUFFLE ? (long)r * r : 1;
    FDN_CHECK_ARG(rr <= 65535);
    const long chunk = 65535 / rr * rr;
    for (long p0 = 0; p0 < oplanes; p0 += chunk) {
        const long np = oplanes - p0 < chunk ? oplanes - p0 : chunk;
        const long in_pl = p0 / rr;
        const bool quad = OW % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 &&
                          (mode != FDN_RS_BILINEAR_HALF || W % 8 == 0);
        const dim3 qgrid((unsigned)cdiv((long)OH * (OW / 4), 256), (unsigned)np);
        if (quad && mode == FDN_RS_BILINEAR_X2)
            hipLaunchKernelGGL(resample4_kernel<FDN_RS_BILINEAR_X2>, qgrid, dim3(256), 0, static_cast<hipStream_t>(stream),
                               x + in_pl * H * W, out + p0 * OH * OW, H, W, OH, OW);
        else if (quad && mode == FDN_RS_BILINEAR_HALF)
            hipLaunchKernelGGL(resample4_kernel<FDN_RS_BILINEAR_HALF>, qgrid, dim3(256), 0, static_cast<hipStream_t>(stream),
                               x + in_pl * H * W, out + p0 * OH * OW, H, W, OH, OW);
        else if (quad && mode == FDN_RS_NEAREST_X2 && W % 2 == 0)
            hipLaunchKernelGGL(resample4_kernel<FDN_RS_NEAREST_X2>, qgrid, dim3(256), 0, static_cast<hipStream_t>(stream),
                               x + in_pl * H * W, out + p0 * OH * OW, H, W, OH, OW);
        else
        hipLaunchKernelGGL(resample_kernel, dim3(cdiv(OW, 256), OH, (unsigned)np), dim3(256), 0, static_cast<hipStream_t>(stream),
                           x + in_pl * H * W, out + p0 * OH * OW, np, H, W, OH, OW, mode, r);
    }
    return fdn_launch_status();
}

extern "C" int fdn_upconv_gather(const float* z, float* out, int B, int Cout, int h, int w, fdn_stream_t stream) {
    FDN_CHECK_ARG(z && out && B > 0 && Cout > 0 && h > 0 && w > 0 && B < 65536);
    FDN_CHECK_ARG((reinterpret_cast<uintptr_t>(out) & 7) == 0);
    if ((long)h * w > 0x7FFFFFFFL / 4) return FDN_ERR_UNSUPPORTED;
    constexpr int COB = 8;                 // (2 / 4 / 16 channels per thread: 0.88 / 0.95 / 0.80 ms against 0.82)
    hipLaunchKernelGGL(upconv_gather_kernel<COB>, dim3((unsigned)cdiv((long)h * w, 256), (unsigned)cdiv(Cout, COB), (unsigned)B), dim3(256), 0,
                       static_cast<hipStream_t>(stream), z, out, Cout, h, w);
    return fdn_launch_status();
}

extern "C" int fdn_dw1x1_pad1(const float* x, const float* w, const float* bias, float* out, int B, int C, int H, int W,
                              fdn_stream_t stream) {
    FDN_CHECK_ARG(x && w && bias && out && B > 0 && C > 0 && H > 0 && W > 0);
    FDN_CHECK_ARG(H + 2 <= 65535 && (long)B * C <= 65535);
    hipLaunchKernelGGL(dw1x1_pad1_kernel, dim3(cdiv(W + 2, 256), H + 2, (unsigned)(B * C)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, w, bias, out, C, H, W);
    return fdn_launch_status();
}

extern "C" int fdn_avgpool3s2(const float* x, float* out, long planes, int H, int W, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && out && planes > 0 && H > 0 && W > 0);
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(avgpool3s2_kernel, dim3(cdiv(planes * OH * OW, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       out, planes, H, W, OH, OW);
    return fdn_launch_status();
}

extern "C" int fdn_global_avgpool(const float* x, float* out, long planes, long P, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && out && planes > 0 && P > 0);
    hipLaunchKernelGGL(gap_kernel, dim3((unsigned)planes), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, P);
    return fdn_launch_status();
}

extern "C" int fdn_se_apply(const float* y, const float* gate, const float* shortcut, float* out, long planes, long P,
                            fdn_stream_t stream) {
    FDN_CHECK_ARG(y && gate && shortcut && out && planes > 0 && P > 0);
    hipLaunchKernelGGL(se_apply_kernel, dim3(cdiv(planes * P, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), y, gate,
                       shortcut, out, P, planes * P);
    return fdn_launch_status();
}

extern "C" int fdn_scale_batch(float* x, const float* ratio, int B, long per_batch, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && ratio && B > 0 && per_batch > 0);
    hipLaunchKernelGGL(scale_batch_kernel, dim3(cdiv(B * per_batch, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       ratio, per_batch, B * per_batch);
    return fdn_launch_status();
}

extern "C" int fdn_spectral_mlp2(float* mag, float* pha, const float* w1m, const float* b1m, const float* w2m, const float* b2m,
                                 const float* w1p, const float* b1p, const float* w2p, const float* b2p, int B, int C, long P, float slope,
                                 fdn_stream_t stream) {
    FDN_CHECK_ARG(mag && pha && w1m && b1m && w2m && b2m && w1p && b1p && w2p && b2p && B > 0 && C > 0 && P > 0 && B < 65536);
    const dim3 grid((unsigned)cdiv(P, 256), (unsigned)B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (C) {
        case 12: hipLaunchKernelGGL(spectral_mlp2_kernel<12>, grid, dim3(256), 0, s, mag, pha, w1m, b1m, w2m, b2m, w1p, b1p, w2p, b2p, P, slope); break;
        case 24: hipLaunchKernelGGL(spectral_mlp2_kernel<24>, grid, dim3(256), 0, s, mag, pha, w1m, b1m, w2m, b2m, w1p, b1p, w2p, b2p, P, slope); break;
        case 48: hipLaunchKernelGGL(spectral_mlp2_kernel<48>, grid, dim3(256), 0, s, mag, pha, w1m, b1m, w2m, b2m, w1p, b1p, w2p, b2p, P, slope); break;
        default: return FDN_ERR_UNSUPPORTED;
    }
    return fdn_launch_status();
}

extern "C" int fdn_gamma_curve(const float* x, const float* i_map, float* out, float scale, long total, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && i_map && out && total > 0);
    hipLaunchKernelGGL(gamma_curve_kernel, dim3(cdiv(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, i_map, out,
                       scale, total);
    return fdn_launch_status();
}
