// 8x8-patch spectral kernels: the "one launch per block" fused FFT <-> pointwise <-> iFFT of the
// FDformer (FDSA core and FDFFN middle).  Everything between the global load of the activation
// tile and the global store of the result stays in LDS / registers:
//   depthwise 3x3 (halo tile) -> rfft2 per 8x8 patch (8-point radix-2 butterflies in registers,
//   row pass then column pass through LDS) -> amplitude/phase recombination without atan2/sincos
//   (e^{i(ang q - ang k)} = u_q * conj(u_k), e^{i ang v} = v/|v|) -> irfft2.
// Tile = 32 x 64 pixels (4 x 8 patches) of one channel per 256-thread workgroup.
#include "common.hpp"

namespace {

constexpr int TH = 32, TW = 64;
constexpr int NP = 32;             // patches per tile
constexpr int PS = 41;             // patch stride of the spectrum buffer in float2 (40 used + 1 pad)

// in-place 8-point complex FFT, natural order in and out.  INV: e^{+...}, unscaled.
template <bool INV>
__device__ __forceinline__ void fft8(float2 (&v)[8]) {
    constexpr float S = INV ? 1.f : -1.f;
    constexpr float C8 = 0.70710678118654752440f;
    // stage 1 (span 4), twiddle w8^i
    float2 a[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = make_float2(v[i].x + v[i + 4].x, v[i].y + v[i + 4].y);
        const float2 d = make_float2(v[i].x - v[i + 4].x, v[i].y - v[i + 4].y);
        float2 w;
        if (i == 0) w = make_float2(1.f, 0.f);
        else if (i == 1) w = make_float2(C8, S * C8);
        else if (i == 2) w = make_float2(0.f, S);
        else w = make_float2(-C8, S * C8);
        a[i + 4] = cmul(d, w);
    }
    // stage 2 (span 2), twiddle w4^i
    float2 c[8];
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            c[h + i] = make_float2(a[h + i].x + a[h + i + 2].x, a[h + i].y + a[h + i + 2].y);
            const float2 d = make_float2(a[h + i].x - a[h + i + 2].x, a[h + i].y - a[h + i + 2].y);
            c[h + i + 2] = (i == 0) ? d : cmul(d, make_float2(0.f, S));
        }
    }
    // stage 3 (span 1) and bit-reversed write-back
    constexpr int br[8] = {0, 4, 2, 6, 1, 5, 3, 7};
#pragma unroll
    for (int h = 0; h < 8; h += 2) {
        v[br[h]] = make_float2(c[h].x + c[h + 1].x, c[h].y + c[h + 1].y);
        v[br[h + 1]] = make_float2(c[h].x - c[h + 1].x, c[h].y - c[h + 1].y);
    }
}

// forward real row transform: 8 reals -> bins 0..4
__device__ __forceinline__ void rfft8_row(const float* r, float2 (&o)[5]) {
    float2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = make_float2(r[i], 0.f);
    fft8<false>(v);
#pragma unroll
    for (int i = 0; i < 5; ++i) o[i] = v[i];
}

// inverse c2r row transform from bins 0..4 (imag of bins 0 and 4 ignored, like pocketfft/MKL c2r)
__device__ __forceinline__ void irfft8_row(const float2 (&x)[5], float* r) {
    float2 v[8];
    v[0] = make_float2(x[0].x, 0.f);
    v[4] = make_float2(x[4].x, 0.f);
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        v[i] = x[i];
        v[8 - i] = make_float2(x[i].x, -x[i].y);
    }
    fft8<true>(v);
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = v[i].x;
}

__device__ __forceinline__ float2 unit(float2 z) {
    const float n = 1.0f / cabs2(z);
    return make_float2(z.x * n, z.y * n);
}

// ------------------------------------------------------------------------------------------
// FDSA core
// ------------------------------------------------------------------------------------------
constexpr int LS = 68;

__global__ __launch_bounds__(256) void fdsa_core_kernel(const float* __restrict__ hidden, const float* __restrict__ dww,
                                                        const float* __restrict__ fftw, float* __restrict__ out, int E,
                                                        int H, int W, int tiles_x) {
    // tin (halo tiles of q,k,v,vv) is dead once the stencils are done; the spectra alias it.
    __shared__ __attribute__((aligned(16))) float tin_raw[4 * (TH + 2) * LS];
    __shared__ __attribute__((aligned(16))) float D[3][TH][LS];
    float (*tin)[TH + 2][LS] = reinterpret_cast<float (*)[TH + 2][LS]>(tin_raw);
    float2* S = reinterpret_cast<float2*>(tin_raw);   // [3][NP][PS]  (3*32*41*8 B = 31.5 KB <= 37 KB)
    static_assert(3 * NP * PS * 8 <= 4 * (TH + 2) * LS * 4, "spectrum buffer must fit in the halo buffer");

    const int tid = threadIdx.x;
    const int e = blockIdx.y, b = blockIdx.z;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const long hw = (long)H * W;
    const long base = (long)b * 4 * E * hw;

    for (int t = 0; t < 4; ++t) {
        const float* src = hidden + base + (long)(t * E + e) * hw;
        for (int i = tid; i < (TH + 2) * (TW + 2); i += 256) {
            const int r = i / (TW + 2), c = i - r * (TW + 2);
            const int y = ty0 - 1 + r, x = tx0 - 1 + c;
            tin[t][r][c] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(long)y * W + x] : 0.f;
        }
    }
    __syncthreads();

    // ---- depthwise 3x3 (to_hidden_dw, FDN_arch.py:578) ------------------------------------
    {
        const int cx = tid & 63, r0 = (tid >> 6) * 8;
        const int gx = tx0 + cx;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float wk[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) wk[i] = dww[(t * E + e) * 9 + i];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float a = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) a = fmaf(wk[dy * 3 + dx], tin[t][r0 + i + dy][cx + dx], a);
                if (t < 3) D[t][r0 + i][cx] = a;
                else {
                    const int gy = ty0 + r0 + i;
                    if (gy < H && gx < W) out[base + (long)(3 * E + e) * hw + (long)gy * W + gx] = a;   // v_value
                }
            }
        }
    }
    __syncthreads();

    // ---- forward rows: thread = (patch, row) ------------------------------------------------
    const int patch = tid >> 3, rr = tid & 7;
    const int py = patch >> 3, px = patch & 7;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float r[8];
        const float4 lo = *reinterpret_cast<const float4*>(&D[t][py * 8 + rr][px * 8]);
        const float4 hi = *reinterpret_cast<const float4*>(&D[t][py * 8 + rr][px * 8 + 4]);
        r[0] = lo.x; r[1] = lo.y; r[2] = lo.z; r[3] = lo.w; r[4] = hi.x; r[5] = hi.y; r[6] = hi.z; r[7] = hi.w;
        float2 o[5];
        rfft8_row(r, o);
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) S[(t * NP + patch) * PS + kx * 8 + rr] = o[kx];
    }
    __syncthreads();

    // ---- columns: thread = (patch, kx): forward, recombine, inverse ---------------------------
    if (tid < NP * 5) {
        const int pj = tid / 5, kx = tid - pj * 5;
        float2 q[8], k[8], v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            q[i] = S[(0 * NP + pj) * PS + kx * 8 + i];
            k[i] = S[(1 * NP + pj) * PS + kx * 8 + i];
            v[i] = S[(2 * NP + pj) * PS + kx * 8 + i];
        }
        fft8<false>(q);
        fft8<false>(k);
        fft8<false>(v);
        float2 o1[8], o2[8], o3[8];
#pragma unroll
        for (int ky = 0; ky < 8; ++ky) {
            const float f = fftw[(e * 8 + ky) * 5 + kx];
            const float2 v1 = make_float2(rd1(v[ky].x * f), rd1(v[ky].y * f));            // :591-593
            float2 qk = cmul(q[ky], k[ky]);                                               // :595
            qk = make_float2(rd1(qk.x), rd1(qk.y));                                       // :597
            const float qka = cabs2(qk), va = cabs2(v1);                                  // :599,601
            const float2 uq = unit(make_float2(rd1(q[ky].x), rd1(q[ky].y)));              // :603,605
            const float2 uk = unit(make_float2(rd1(k[ky].x), rd1(k[ky].y)));              // :604,606
            const float2 u = cmulc(uq, uk);                                               // e^{i(qp-kp)} :607
            const float2 uv = make_float2(v1.x / va, v1.y / va);                          // e^{i v_p}
            o1[ky] = make_float2(va * u.x, va * u.y);                                     // :609-612
            o2[ky] = make_float2(qka * uv.x, qka * uv.y);                                 // :617-619
            o3[ky] = make_float2(qka * u.x, qka * u.y);                                   // :627-629
        }
        fft8<true>(o1);
        fft8<true>(o2);
        fft8<true>(o3);
        constexpr float sc = 1.0f / 64.0f;   // norm='backward'
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            S[(0 * NP + pj) * PS + kx * 8 + i] = make_float2(o1[i].x * sc, o1[i].y * sc);
            S[(1 * NP + pj) * PS + kx * 8 + i] = make_float2(o2[i].x * sc, o2[i].y * sc);
            S[(2 * NP + pj) * PS + kx * 8 + i] = make_float2(o3[i].x * sc, o3[i].y * sc);
        }
    }
    __syncthreads();

    // ---- inverse rows -> D ----------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float2 x[5];
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) x[kx] = S[(t * NP + patch) * PS + kx * 8 + rr];
        float r[8];
        irfft8_row(x, r);
        *reinterpret_cast<float4*>(&D[t][py * 8 + rr][px * 8]) = make_float4(r[0], r[1], r[2], r[3]);
        *reinterpret_cast<float4*>(&D[t][py * 8 + rr][px * 8 + 4]) = make_float4(r[4], r[5], r[6], r[7]);
    }
    __syncthreads();

    // ---- coalesced store of out1|out2|out3 -------------------------------------------------------
    {
        const int cx = tid & 63, r0 = (tid >> 6) * 8;
        const int gx = tx0 + cx;
        if (gx < W) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int gy = ty0 + r0 + i;
                    if (gy < H) out[base + (long)(t * E + e) * hw + (long)gy * W + gx] = D[t][r0 + i][cx];
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
// FDFFN middle: freq branch + (dw3x3 -> GELU -> dw3x3) spatial branch
// ------------------------------------------------------------------------------------------
constexpr int LS2 = 72;

__global__ __launch_bounds__(256) void fdffn_mid_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                        const float* __restrict__ w2, const float* __restrict__ ffta,
                                                        const float* __restrict__ fftp, float* __restrict__ out, int Hd,
                                                        int H, int W, int tiles_x) {
    __shared__ __attribute__((aligned(16))) float tin[TH + 4][LS2];   // halo 2
    __shared__ float mid[TH + 2][LS];                                  // gelu(dw0(x)) on halo 1
    __shared__ __attribute__((aligned(16))) float Fq[TH][LS];          // frequency-branch result
    __shared__ float2 S[NP * PS];

    const int tid = threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const long hw = (long)H * W;
    const float* src = x + ((long)b * Hd + c) * hw;

    for (int i = tid; i < (TH + 4) * (TW + 4); i += 256) {
        const int r = i / (TW + 4), cc = i - r * (TW + 4);
        const int y = ty0 - 2 + r, xx = tx0 - 2 + cc;
        tin[r][cc] = (y >= 0 && y < H && xx >= 0 && xx < W) ? src[(long)y * W + xx] : 0.f;
    }
    float k0[9], k2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        k0[i] = w0[c * 9 + i];
        k2[i] = w2[c * 9 + i];
    }
    __syncthreads();

    // first depthwise conv + GELU on the (TH+2) x (TW+2) ring; zero outside the image (= the
    // zero padding the second conv sees, FDN_arch.py:439)
    for (int i = tid; i < (TH + 2) * (TW + 2); i += 256) {
        const int r = i / (TW + 2), cc = i - r * (TW + 2);
        const int y = ty0 - 1 + r, xx = tx0 - 1 + cc;
        float a = 0.f;
        if (y >= 0 && y < H && xx >= 0 && xx < W) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) a = fmaf(k0[dy * 3 + dx], tin[r + dy][cc + dx], a);
            a = gelu_erf(a);
        }
        mid[r][cc] = a;
    }

    // forward rows of the frequency branch straight from the input tile (centre of tin)
    const int patch = tid >> 3, rr = tid & 7;
    const int py = patch >> 3, px = patch & 7;
    {
        float r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = tin[2 + py * 8 + rr][2 + px * 8 + i];
        float2 o[5];
        rfft8_row(r, o);
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) S[patch * PS + kx * 8 + rr] = o[kx];
    }
    __syncthreads();

    // second depthwise conv (kept in registers until the final add)
    float sp[8];
    {
        const int cx = tid & 63, r0 = (tid >> 6) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float a = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) a = fmaf(k2[dy * 3 + dx], mid[r0 + i + dy][cx + dx], a);
            sp[i] = a;
        }
    }

    // columns: forward, z * ffta * e^{-i fftp}, inverse  (FDN_arch.py:460-469)
    if (tid < NP * 5) {
        const int pj = tid / 5, kx = tid - pj * 5;
        float2 z[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = S[pj * PS + kx * 8 + i];
        fft8<false>(z);
#pragma unroll
        for (int ky = 0; ky < 8; ++ky) {
            const float a = ffta[(c * 8 + ky) * 5 + kx], ph = fftp[(c * 8 + ky) * 5 + kx];
            float sn, cs;
            sincosf(ph, &sn, &cs);
            const float2 zz = make_float2(rd1(z[ky].x), rd1(z[ky].y));                    // :461
            z[ky] = cmul(zz, make_float2(a * cs, -a * sn));
        }
        fft8<true>(z);
        constexpr float sc = 1.0f / 64.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) S[pj * PS + kx * 8 + i] = make_float2(z[i].x * sc, z[i].y * sc);
    }
    __syncthreads();

    {
        float2 xk[5];
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) xk[kx] = S[patch * PS + kx * 8 + rr];
        float r[8];
        irfft8_row(xk, r);
        *reinterpret_cast<float4*>(&Fq[py * 8 + rr][px * 8]) = make_float4(r[0], r[1], r[2], r[3]);
        *reinterpret_cast<float4*>(&Fq[py * 8 + rr][px * 8 + 4]) = make_float4(r[4], r[5], r[6], r[7]);
    }
    __syncthreads();

    {
        const int cx = tid & 63, r0 = (tid >> 6) * 8;
        const int gx = tx0 + cx;
        float* dst = out + ((long)b * Hd + c) * hw;
        if (gx < W) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int gy = ty0 + r0 + i;
                if (gy < H) dst[(long)gy * W + gx] = Fq[r0 + i][cx] + sp[i];               // :470
            }
        }
    }
}

}  // namespace

extern "C" int fdn_fdsa_core(const float* hidden, const float* dw_w, const float* fft_w, float* out, int B, int E, int H,
                             int W, fdn_stream_t stream) {
    FDN_CHECK_ARG(hidden && dw_w && fft_w && out && B > 0 && E > 0 && H > 0 && W > 0);
    FDN_CHECK_ARG(H % 8 == 0 && W % 8 == 0 && E < 65536 && B < 65536);
    const int tx = cdiv(W, TW), ty = cdiv(H, TH);
    hipLaunchKernelGGL(fdsa_core_kernel, dim3(tx * ty, E, B), dim3(256), 0, static_cast<hipStream_t>(stream), hidden, dw_w,
                       fft_w, out, E, H, W, tx);
    return fdn_launch_status();
}

extern "C" int fdn_fdffn_mid(const float* x, const float* w0, const float* w2, const float* ffta, const float* fftp,
                             float* out, int B, int Hd, int H, int W, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && w0 && w2 && ffta && fftp && out && B > 0 && Hd > 0 && H > 0 && W > 0);
    FDN_CHECK_ARG(H % 8 == 0 && W % 8 == 0 && Hd < 65536 && B < 65536);
    const int tx = cdiv(W, TW), ty = cdiv(H, TH);
    hipLaunchKernelGGL(fdffn_mid_kernel, dim3(tx * ty, Hd, B), dim3(256), 0, static_cast<hipStream_t>(stream), x, w0, w2,
                       ffta, fftp, out, Hd, H, W, tx);
    return fdn_launch_status();
}
