// FDFFN front half in ONE launch: channel LayerNorm + project_in (1x1 conv C -> Hd on the matrix cores) + everything
// fdffn_mid_kernel does (spatial branch dw3x3 -> GELU -> dw3x3, frequency branch 8x8 rfft2 . filter . irfft2, sum;
// FDN_arch.py:456-470 with the LayerNorm of :672 in front) - the Hd-channel hidden tensor never exists in HBM (it was written
// and read back: 2 x 2.59 GB per block at level 1).
//
// Workgroup = one 8 x 32 pixel tile (1 x 4 patches) of one image, ALL Hd channels, 256 threads.
//   * the two depthwise convs need the hidden tensor on a 2-pixel halo: 12 x 36 = 432 pixels = 27 groups of 16.  The normalised
//     input of those pixels is loaded once, cut into three exact bf16 parts (common.hpp) and lives in registers as the B operand
//     of v_mfma_f32_16x16x32_bf16 (pixel on the lane, 8 consecutive channels per lane): 12 VGPRs per group and 32 channels,
//     seven groups per wave;
//   * the hidden channels are walked in chunks of 16 = the MFMA's rows (weights packed per chunk / k-step / part / lane by
//     fdn_fdffn_pack, LayerNorm affine folded in, bias = one more MFMA against bf16 ones so that pixels outside the image come
//     out 0 = the convs' zero padding): 7 MFMAs per group, results parked in a 16-plane LDS tile;
//   * two rounds of 8 channels per chunk, thread = (channel, patch, row) as in fdsa_fused_kernel: dw3x3 + GELU on the 10 x 34 ring
//     (240 threads x 12 pixels), forward rows from the tile centre | barrier | second dw3x3, column transforms with the
//     precomputed filter ffta e^{-i fftp} | barrier | inverse rows + spatial branch, 32-byte stores.  The spectra are double
//     buffered (two barriers per round); taps and filters of the next round are staged through LDS a round ahead.
// The 8 x 32 tile pays for the halo: the ring is 340 pixels per 256 (the 32 x 64 tile of fdffn_mid_kernel: 1.10 x) and
// project_in runs on 432 - on the bf16 matrix pipe, beside the vector ALU that bounds this kernel.
#include "patch_fft.hpp"

namespace {

constexpr int GT_H = 8, GT_W = 32;
constexpr int GHH = GT_H + 4, GHW = GT_W + 4;      // halo-2 tile 12 x 36
constexpr int GHP = GHH * GHW;                     // 432 pixels
constexpr int GNG = GHP / 16;                      // 27 groups of 16 pixels
constexpr int GPW = (GNG + 3) / 4;                 // groups per wave (7; wave 3 has 6)
constexpr int HRS = 37;                            // LDS row stride of a hidden plane
constexpr int HPL = GHH * HRS;                     // 444 floats per plane
constexpr int MRS = 35;                            // row stride of a ring plane (10 x 34)
constexpr int MPL = (GT_H + 2) * MRS;              // 350
constexpr int CHK = 16;                            // hidden channels per chunk = MFMA rows
constexpr int RND = 8;                             // channels per round
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4v mf16(fdn_u32x4 a, fdn_u32x4 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, a), __builtin_bit_cast(bf16x8v, b), c, 0, 0, 0);
}

struct GArgs {
    const float* x;
    long xbs;
    const float* stats;
    const fdn_u32x4* wpk;
    const float* w0;
    const float* w2;
    const float2* filt;
    void* out;
    int Hd, H, W, tiles_x, tiles_per_img, nchunks;
};

template <int C, bool LN, bool OBF>
__global__ __launch_bounds__(256, 2) void fdffn_fused_kernel(GArgs a) {
    constexpr int KST = (C + 31) / 32;             // MFMA k-steps (32 channels each)
    constexpr int KS = KST * 3 + 1;                // A-operand slots per chunk: (k-step, part) + the bias slot
    __shared__ float hid[CHK * HPL + 8];
    __shared__ float mid[RND * MPL + 8];
    __shared__ __attribute__((aligned(16))) float2 S[2][NP * PS];
    __shared__ float2 flt[2][RND * 40];
    __shared__ float tap[2][2 * RND * 9];          // [k0 | k2][channel][9]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, l16 = lane & 15;
    const int t_ = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = t_ / a.tiles_per_img, ti = t_ - b * a.tiles_per_img;
    const int ty0 = (ti / a.tiles_x) * GT_H, tx0 = (ti % a.tiles_x) * GT_W;
    const int Hd = a.Hd, H = a.H, W = a.W;
    const unsigned P = (unsigned)H * W, hw4 = P * 4u;
    const rsrc_t rx = mk_rsrc(a.x + (long)b * a.xbs, (unsigned)C * hw4);
    const rsrc_t rst = mk_rsrc(LN ? a.stats + (long)b * 2 * P : a.x, LN ? 2u * hw4 : 0u);
    constexpr unsigned OES = st_bytes<OBF>();
    const unsigned hwo = P * OES;
    const rsrc_t rout = mk_rsrc(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.out) + (long)b * Hd * P * OES), (unsigned)Hd * hwo);

    // ---- the wave's pixel groups of the normalised halo tile: B operands, resident for the whole workgroup ----
    fdn_u32x4 xb[GPW][KST][3];
    int pixoff[GPW];
    unsigned okbits = 0;                // bit i: group i's pixel of this lane is inside the image
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int gi = wave + 4 * i;
        const int p = gi * 16 + l16;
        const int r = p / GHW, c = p - r * GHW;
        const int gy = ty0 - 2 + r, gx = tx0 - 2 + c;
        const bool ok = gi < GNG && gy >= 0 && gy < H && gx >= 0 && gx < W;
        okbits |= ok ? (1u << i) : 0u;
        pixoff[i] = gi < GNG ? r * HRS + c : CHK * HPL;          // groups past the tile: the spare cells behind the planes
        const unsigned g = ok ? (unsigned)(gy * W + gx) * 4u : OOB;
        float mu = 0.f, rs = 1.f;
        if (LN) {
            mu = bload(rst, g, 0);
            rs = bload(rst, g, hw4);
        }
        float xs[KST][8];
#pragma unroll
        for (int ks = 0; ks < KST; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) xs[ks][j] = bload(rx, g + (unsigned)(8 * kq) * hw4, (unsigned)(32 * ks + j) * hw4);     // k >= C, pixels outside: 0
#pragma unroll
        for (int ks = 0; ks < KST; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v0 = xs[ks][2 * j], v1 = xs[ks][2 * j + 1];
                if (LN) v0 = (v0 - mu) * rs, v1 = (v1 - mu) * rs;
                unsigned p1, p2, p3;
                fdn_split3(v0, v1, p1, p2, p3);
                xb[i][ks][0][j] = p1, xb[i][ks][1][j] = p2, xb[i][ks][2][j] = p3;
            }
    }

    // VALU-phase coordinates (as fdsa_fused_kernel): lanes 0-31 / 32-63 of a wave take two different channels of the round
    const int el = wave * 2 + (lane >> 5);
    const int row = lane & 7, px = (lane >> 3) & 3;
    const int slot = el * 4 + px;
    const int gx0 = tx0 + px * 8;
    const unsigned opix = gx0 < W ? (unsigned)((ty0 + row) * W + gx0) * OES : OOB;
    // ring job of this thread (tid < 240): channel rel, ring row rr, column group cg (12 | 12 | 10 columns)
    const int rel = tid / 30, rem = tid - rel * 30, cg = rem / 10, rr = rem - cg * 10;
    const int ry = ty0 - 1 + rr;
    const bool ryok = tid < 240 && ry >= 0 && ry < H;
    const int rx0 = tx0 - 1 + cg * 12;

    // per-round parameters (taps of both depthwise convs, the complex filter), staged through LDS a round ahead
    float st_t = 0.f;
    float2 st_f0 = make_float2(0.f, 0.f), st_f1 = st_f0;
    auto stage_fetch = [&](int c0) {                                     // c0 = first channel of the round
        if (tid < 2 * RND * 9) {
            const int which = tid / (RND * 9), i = tid - which * (RND * 9);
            const int c = c0 + i / 9;
            st_t = (which ? a.w2 : a.w0)[(long)(c < Hd ? c : Hd - 1) * 9 + (i - (i / 9) * 9)];
        }
        const int c1 = c0 + tid / 40;
        st_f0 = a.filt[(long)(c1 < Hd ? c1 : Hd - 1) * 40 + (tid - (tid / 40) * 40)];
        if (tid < 64) {
            const int c2 = c0 + (tid + 256) / 40;
            st_f1 = a.filt[(long)(c2 < Hd ? c2 : Hd - 1) * 40 + ((tid + 256) - ((tid + 256) / 40) * 40)];
        }
    };
    auto stage_store = [&](int buf) {
        if (tid < 2 * RND * 9) tap[buf][tid] = st_t;
        flt[buf][tid] = st_f0;
        if (tid < 64) flt[buf][tid + 256] = st_f1;
    };
    fdn_u32x4 aw[KS];
    auto aw_fetch = [&](int ch) {
        const fdn_u32x4* wp = a.wpk + ((long)ch * KS) * 64 + lane;
#pragma unroll
        for (int j = 0; j < KS; ++j) aw[j] = wp[j * 64];
    };
    aw_fetch(0);
    stage_fetch(0);
    stage_store(0);                     // (visible behind the first barrier of the loop)

    int rnd = 0;                        // running round number: parameter / spectrum buffer = rnd & 1
    for (int ch = 0; ch < a.nchunks; ++ch) {
        // ---- project_in on the matrix cores: D[16 channels][16 halo pixels] per group -> LDS planes ----
#pragma unroll
        for (int i = 0; i < GPW; ++i) {
            if (wave + 4 * i < GNG) {                                     // wave-uniform
                f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {                          // the six leading products, small terms first (common.hpp)
                    acc = mf16(aw[3 * ks + 2], xb[i][ks][0], acc);
                    acc = mf16(aw[3 * ks + 1], xb[i][ks][1], acc);
                    acc = mf16(aw[3 * ks], xb[i][ks][2], acc);
                    acc = mf16(aw[3 * ks + 1], xb[i][ks][0], acc);
                    acc = mf16(aw[3 * ks], xb[i][ks][1], acc);
                    acc = mf16(aw[3 * ks], xb[i][ks][0], acc);
                }
                const bool one = ((okbits >> i) & 1u) && kq == 0;
                const fdn_u32x4 xone = {one ? 0x3F803F80u : 0u, one ? 0x00003F80u : 0u, 0u, 0u};
                acc = mf16(aw[KS - 1], xone, acc);                          // + bias: b1 + b2 + b3 against 1, 1, 1 (0 outside the image)
#pragma unroll
                for (int r = 0; r < 4; ++r) hid[(4 * kq + r) * HPL + pixoff[i]] = acc[r];
            }
        }
        if (ch + 1 < a.nchunks) aw_fetch(ch + 1);
        __syncthreads();

#pragma unroll 1
        for (int r2 = 0; r2 < CHK / RND; ++r2, ++rnd) {
            const int pb = rnd & 1;
            const int c0 = ch * CHK + r2 * RND;
            const int c = c0 + el;
            const bool more = c0 + RND < a.nchunks * CHK;
            if (more) stage_fetch(c0 + RND);

            // ---- A: first depthwise conv + GELU on the 10 x 34 ring (0 outside the image: the padding the second conv sees,
            // FDN_arch.py:439), 12 pixels per thread with a sliding 3 x 14 window; forward rows of the frequency branch
            if (tid < 240) {
                const float* hp = hid + (r2 * RND + rel) * HPL + rr * HRS + cg * 12;
                float k0[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) k0[i] = tap[pb][rel * 9 + i];
                float o[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) o[j] = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    float v[14];
#pragma unroll
                    for (int j = 0; j < 14; ++j) v[j] = hp[dy * HRS + j];
#pragma unroll
                    for (int j = 0; j < 12; ++j)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) o[j] = fmaf(k0[dy * 3 + dx], v[j + dx], o[j]);
                }
                float* mp = mid + rel * MPL + rr * MRS + cg * 12;
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const int xx = rx0 + j;
                    const float g = gelu_fast(o[j]);                      // unconditional: twelve independent chains, then a select
                    if (j < 10 || cg < 2) mp[j] = (ryok && xx >= 0 && xx < W) ? g : 0.f;
                }
            }
            {
                const float* hc = hid + (r2 * RND + el) * HPL + (2 + row) * HRS + 2 + px * 8;
                float r8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) r8[i] = hc[i];
                float2 o[5];
                rfft8_row(r8, o);
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) S[pb][slot * PS + kx * KXS + row] = o[kx];
            }
            __syncthreads();

            // ---- B: second depthwise conv; column transforms: forward, z * ffta e^{-i fftp}, inverse (FDN_arch.py:460-469) ----
            if (more) stage_store(pb ^ 1);
            float sp[8];
            {
                const float* mq = mid + el * MPL + row * MRS + px * 8;
                float k2[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) k2[i] = tap[pb][RND * 9 + el * 9 + i];
#pragma unroll
                for (int j = 0; j < 8; ++j) sp[j] = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    float v[10];
#pragma unroll
                    for (int j = 0; j < 10; ++j) v[j] = mq[dy * MRS + j];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) sp[j] = fmaf(k2[dy * 3 + dx], v[j + dx], sp[j]);
                }
            }
            if (tid < NP * 5) {
                const int pj = tid / 5, kx = tid - pj * 5;
                float2 z[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) z[i] = S[pb][pj * PS + kx * KXS + i];
                fft8<false>(z);
#pragma unroll
                for (int ky = 0; ky < 8; ++ky)
                    z[ky] = cmul(make_float2(rd1(z[ky].x), rd1(z[ky].y)), flt[pb][(pj >> 2) * 40 + ky * 5 + kx]);     // :461-468
                fft8<true>(z);
                constexpr float sc = 1.0f / 64.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) S[pb][pj * PS + kx * KXS + i] = make_float2(z[i].x * sc, z[i].y * sc);
            }
            __syncthreads();

            // ---- C: inverse rows + spatial branch, 32-byte segments to global (the next round works on the other spectrum buffer)
            {
                float2 xk[5];
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) xk[kx] = S[pb][slot * PS + kx * KXS + row];
                float r8[8];
                irfft8_row(xk, r8);
#pragma unroll
                for (int j = 0; j < 8; ++j) r8[j] += sp[j];                                                             // :470
                st_store8<OBF>(r8, rout, c < Hd ? opix + (unsigned)c * hwo : OOB, 0);
            }
        }
        // (the next chunk's MFMA phase rewrites `hid`: last read in phase A of the round above, two barriers ago)
    }
}

// fdn_fdffn_pack: project_in weights [Hd][C] (+ LayerNorm gamma / beta of the input) -> per (chunk of 16 channels, slot, lane)
// 16-byte A operands of v_mfma_f32_16x16x32_bf16 (slot 3 ks + part: the part-th bf16 part of w[row][32 ks + 8 kq + j] * gamma,
// last slot: the bias' three parts on k = 0, 1, 2), and the complex filter ffta e^{-i fftp} per (channel, ky, kx)
__global__ void fdffn_pack_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                  const float* __restrict__ ffta, const float* __restrict__ fftp, fdn_u32x4* __restrict__ wpk,
                                  float2* __restrict__ filt, int C, int Hd, int nchunks) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int KST = (C + 31) / 32, KS = KST * 3 + 1;
    if (idx < Hd * 40) {
        float sn, cs;
        fdn_sincos(fftp[idx], &sn, &cs);
        filt[idx] = make_float2(ffta[idx] * cs, -ffta[idx] * sn);
    }
    if (idx >= nchunks * KS * 64) return;
    const int lane = idx & 63, j = (idx >> 6) % KS, ch = (idx >> 6) / KS;
    const int m = lane & 15, kq = lane >> 4;
    const int e = ch * CHK + m;
    auto part_of = [](float x, int part) {
        for (int p = 0; p < part; ++p) x -= __uint_as_float(__float_as_uint(x) & 0xffff0000u);
        return __float_as_uint(x) >> 16;
    };
    fdn_u32x4 o = {0u, 0u, 0u, 0u};
    if (e < Hd) {
        const float* wr = w + (long)e * C;
        if (j < KS - 1) {
            const int ks = j / 3, part = j - 3 * ks;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned hl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k = 32 * ks + 8 * kq + 2 * q + u;
                    float v = k < C ? wr[k] : 0.f;
                    if (gamma && k < C) v *= gamma[k];
                    hl[u] = part_of(v, part);
                }
                o[q] = hl[0] | (hl[1] << 16);
            }
        } else if (kq == 0 && beta) {
            double sacc = 0.0;
            for (int k = 0; k < C; ++k) sacc += (double)wr[k] * (double)beta[k];
            const float bsum = (float)sacc;
            o[0] = part_of(bsum, 0) | (part_of(bsum, 1) << 16);
            o[1] = part_of(bsum, 2);
        }
    }
    wpk[idx] = o;
}

}  // namespace

extern "C" int fdn_fdffn_pack(const float* w, const float* gamma, const float* beta, const float* ffta, const float* fftp, void* wpk,
                              float* filt, int C, int Hd, fdn_stream_t stream) {
    FDN_CHECK_ARG(w && ffta && fftp && wpk && filt && C > 0 && Hd > 0 && (!gamma == !beta));
    const int nch = (Hd + CHK - 1) / CHK;
    const int total = max(nch * (((C + 31) / 32) * 3 + 1) * 64, Hd * 40);
    hipLaunchKernelGGL(fdffn_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w, gamma, beta, ffta, fftp,
                       static_cast<fdn_u32x4*>(wpk), reinterpret_cast<float2*>(filt), C, Hd, nch);
    return fdn_launch_status();
}

extern "C" int fdn_fdffn_fused(const float* x, long xbs, const float* stats, const void* wpk, const float* w0, const float* w2,
                               const float* filt, void* out, int B, int C, int Hd, int H, int W, int out_bf16, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && wpk && w0 && w2 && filt && out && B > 0 && Hd > 0 && H > 0 && W > 0);
    FDN_CHECK_ARG(H % 8 == 0 && W % 8 == 0);
    FDN_CHECK_ARG((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    FDN_CHECK_ARG(4ull * Hd * H * W < 0x80000000ull && 4ull * (C + 32) * H * W < 0x80000000ull);   // 32-bit byte offsets per image
    GArgs a;
    a.x = x; a.xbs = xbs; a.stats = stats; a.wpk = static_cast<const fdn_u32x4*>(wpk); a.w0 = w0; a.w2 = w2;
    a.filt = reinterpret_cast<const float2*>(filt); a.out = out;
    a.Hd = Hd; a.H = H; a.W = W;
    a.tiles_x = cdiv(W, GT_W);
    a.tiles_per_img = a.tiles_x * (H / GT_H);
    a.nchunks = (Hd + CHK - 1) / CHK;
    const long total = (long)B * a.tiles_per_img;
    FDN_CHECK_ARG(total < 0x7fffffffL);
    const dim3 grid((unsigned)total), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define FDN_FFN_CASE(CC)                                                                                           \
    case CC:                                                                                                       \
        if (stats && out_bf16) hipLaunchKernelGGL((fdffn_fused_kernel<CC, true, true>), grid, block, 0, s, a);     \
        else if (stats) hipLaunchKernelGGL((fdffn_fused_kernel<CC, true, false>), grid, block, 0, s, a);           \
        else if (out_bf16) hipLaunchKernelGGL((fdffn_fused_kernel<CC, false, true>), grid, block, 0, s, a);        \
        else hipLaunchKernelGGL((fdffn_fused_kernel<CC, false, false>), grid, block, 0, s, a);                     \
        break;
    switch (C) {
        FDN_FFN_CASE(24)
        FDN_FFN_CASE(32)
        default: return FDN_ERR_UNSUPPORTED;
    }
#undef FDN_FFN_CASE
    return fdn_launch_status();
}
