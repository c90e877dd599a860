// LDS-tiled 1x1-conv GEMM for the deep, MFMA-bound level-3 shapes (N = 128 output channels: FDSA project_out 459 -> 128 with
// the 3 x LayerNorm * v_value prologue, FDFFN project_out 345 -> 128, Fuse 128 -> 128; FDN_arch.py:633-639, :474, :685).
//
// The flat-strip kernels of gemm1x1.hip give every wave its own 32-pixel strip loaded straight into MFMA B registers; for these
// shapes that leaves one A-operand LDS read per MFMA, a barrier per 64 MFMAs and the prologue arithmetic on the MFMA waves'
// own issue slots (51 % MFMA + 22 % ALU busy at 459 -> 128, profiles/r02_c_summary.txt).  Here a workgroup owns a 128-pixel x
// 128-channel output tile, the K axis streams through LDS in 32-deep chunks of BOTH operands, the four waves form a 2 x 2 grid
// (two pixel strips x two channel tiles each: every LDS read feeds two MFMAs), and the prologue is applied once per element
// while the chunk is staged.  The next chunk's global loads are in flight during the MFMAs of the current one; LDS is single
// buffered (41 KB: three workgroups per CU cover each other's two barriers per chunk).
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mk_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(float v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

constexpr int TP = 128, TN = 128, KC = 32;
constexpr int LS = 161;                     // LDS row stride (floats): the two k of an MFMA step sit 33 banks apart, transposing weight writes hit 32 banks

struct TArgs {
    fdn_conv1x1_desc d;
    int tiles_per_img, total_tiles;
};

template <int PRO>
__global__ __launch_bounds__(256, 3) void gemm_tile_kernel(TArgs a) {
    const fdn_conv1x1_desc& d = a.d;
    // LN3_GATE walks K = 3E as triples k' = 3 e + g (g = LayerNorm group): the three k of a triple share one v_value operand and
    // each has a compile-time group, so a chunk is 10 triples = 30 k (15 MFMA k-steps); the weight rows are gathered to match
    constexpr bool TRI = PRO == FDN_PRO_LN3_GATE;
    constexpr int KCH = TRI ? 30 : KC;
    __shared__ float Xs[KC * LS];
    __shared__ float Ws[KC * LS];
    __shared__ float red[2][TP];              // statistics of the result: partial sums of the two channel halves
    const int K = d.K, N = d.N, E = d.ln_group;
    const unsigned P = (unsigned)d.P, P4 = P * 4u;
    const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5, ln = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform BY CONSTRUCTION: keeps k / soffset arithmetic scalar (no waterfall loops)
    const int wi = wave >> 1, wj = wave & 1;   // pixel strips 2 wi, 2 wi + 1; channel tiles 2 wj, 2 wj + 1
    const int nch = TRI ? (E + 9) / 10 : (K + KC - 1) / KC;

    const unsigned S = xcd_contiguous(blockIdx.x, (unsigned)a.total_tiles);
    const int b = (int)(S / (unsigned)a.tiles_per_img);
    const unsigned p0 = (S - (unsigned)b * a.tiles_per_img) * TP;

    // ---- staging roles: x element (row xr + 2 i | triple xr + 2 i, pixel xp), weight element (row wk, n = wn + 8 i) ----
    const int xp = tid & (TP - 1), xr = wave >> 1;
    const unsigned pix = min(p0 + (unsigned)xp, P - 1);                       // pixels past the image shadow the last one (never stored)
    const int wk = tid & 31, wn = tid >> 5;
    const rsrc_t rx = mk_rsrc(d.x[0] + (long)b * d.xbs[0], (unsigned)K * P4);
    const rsrc_t rv = mk_rsrc(TRI ? d.xb + (long)b * d.xbbs : d.x[0], TRI ? (unsigned)E * P4 : 0u);
    const rsrc_t rw = mk_rsrc(d.w, (unsigned)(N * K) * 4u);
    float sa[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};                  // (x - mean) * rstd = x * sa + sb
    if (TRI) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float* sp = d.stats + ((long)b * 3 + g) * 2 * P;
            sa[g] = sp[P + pix];
            sb[g] = -sp[pix] * sa[g];
        }
    }
    constexpr int NX = TRI ? 15 : 16;
    float xv[NX], vv[TRI ? 5 : 1], wv[16];
    float ga[TRI ? 15 : 1], be[TRI ? 15 : 1];     // gamma / beta of the chunk's elements: fetched with it (uniform loads), not between the barriers
    auto fetch = [&](int c) __attribute__((always_inline)) {
        if constexpr (TRI) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int e = c * 10 + xr + 2 * i;                             // wave-uniform; e >= E reads 0 (v) / finite garbage times 0 (x)
                vv[i] = bload(rv, pix * 4u, (unsigned)e * P4);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    xv[3 * i + g] = e < E ? bload(rx, pix * 4u, (unsigned)(g * E + e) * P4) : 0.f;
                    ga[3 * i + g] = e < E ? d.gamma[g * E + e] : 0.f;
                    be[3 * i + g] = e < E ? d.beta[g * E + e] : 0.f;
                }
            }
            const int kq = c * 30 + wk, e = kq / 3, g = kq - 3 * e;            // this thread's weight row k' = 3 e + g -> column g E + e
            const unsigned wo = (wk < 30 && e < E) ? (unsigned)(wn * K + g * E + e) * 4u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < 16; ++i) wv[i] = bload(rw, wo, (unsigned)(8 * i * K) * 4u);
        } else {
            const int k0 = c * KC;
#pragma unroll
            for (int i = 0; i < 16; ++i) xv[i] = bload(rx, pix * 4u, (unsigned)(k0 + xr + 2 * i) * P4);     // k >= K reads 0
            const unsigned wo = (k0 + wk < K) ? (unsigned)(wn * K + k0 + wk) * 4u : 0x80000000u;      // past K: outside the descriptor, reads 0
#pragma unroll
            for (int i = 0; i < 16; ++i) wv[i] = bload(rw, wo, (unsigned)(8 * i * K) * 4u);           // rows n >= N: past the end of w, 0
        }
    };
    auto stash = [&](int c) __attribute__((always_inline)) {
        if constexpr (TRI) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int e = c * 10 + xr + 2 * i;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float v = e < E ? fmaf(fmaf(xv[3 * i + g], sa[g], sb[g]), ga[3 * i + g], be[3 * i + g]) * vv[i] : 0.f;   // FDN_arch.py:633-638
                    Xs[(3 * (xr + 2 * i) + g) * LS + xp] = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) Xs[(xr + 2 * i) * LS + xp] = xv[i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) Ws[wk * LS + wn + 8 * i] = wv[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][t][r] = 0.f;

    fetch(0);
    stash(0);
    __syncthreads();
    const float* xb_ = Xs + kh * LS + wi * 64 + ln;
    const float* wb_ = Ws + kh * LS + wj * 64 + ln;
    for (int c = 0; c < nch; ++c) {
        const bool more = c + 1 < nch;
        if (more) fetch(c + 1);
#pragma unroll
        for (int s = 0; s < KCH / 2; ++s) {
            const float b0 = xb_[2 * s * LS], b1 = xb_[2 * s * LS + 32];
            const float a0 = wb_[2 * s * LS], a1 = wb_[2 * s * LS + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();                      // every wave has read this chunk
        if (more) stash(c + 1);
        __syncthreads();
    }

    // ---- epilogue: bias, residual, store; LayerNorm statistics of the result over its N <= 128 channels ----
    const rsrc_t ro = mk_rsrc(d.out + (long)b * d.obs, (unsigned)N * P4);
    const rsrc_t rr = mk_rsrc(d.epi == FDN_EPI_RES ? d.res + (long)b * d.rbs : d.out, d.epi == FDN_EPI_RES ? (unsigned)N * P4 : 0u);
    float psum[2] = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const unsigned p = p0 + (unsigned)(wi * 64 + s * 32 + ln);
        const unsigned voff = p < P ? (4u * kh * P + p) * 4u : 0x80000000u;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float rv_[16];
            if (d.epi == FDN_EPI_RES) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv_[r] = bload(rr, voff, (unsigned)((wj * 2 + t) * 32 + (r & 3) + 8 * (r >> 2)) * P4);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = (wj * 2 + t) * 32 + (r & 3) + 8 * (r >> 2);      // + 4 kh per lane
                float v = acc[s][t][r];
                if (d.bias) v += (nrow + 4 * kh < N) ? d.bias[nrow + 4 * kh] : 0.f;
                if (d.epi == FDN_EPI_RES) v += rv_[r];
                bstore(v, ro, voff, (unsigned)nrow * P4);                         // rows >= N fall outside the descriptor
                v = (nrow + 4 * kh < N) ? v : 0.f;
                acc[s][t][r] = v;
                psum[s] += v;
            }
        }
    }
    if (d.stats_out) {
        // two-pass mean / variance: the two waves of a pixel strip pair (wj = 0, 1) hold complementary channel halves
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            psum[s] += __shfl_xor(psum[s], 32);
            if (kh == 0) red[wj][wi * 64 + s * 32 + ln] = psum[s];
        }
        __syncthreads();
        float mean[2], q[2] = {0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pl = wi * 64 + s * 32 + ln;
            mean[s] = (red[0][pl] + red[1][pl]) / (float)N;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nrow = (wj * 2 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const float dl = acc[s][t][r] - mean[s];
                    q[s] += nrow < N ? dl * dl : 0.f;
                }
            q[s] += __shfl_xor(q[s], 32);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (kh == 0) red[wj][wi * 64 + s * 32 + ln] = q[s];
        __syncthreads();
        if (wj == 0 && kh == 0) {
            float* sp = d.stats_out + (long)b * 2 * P;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pl = wi * 64 + s * 32 + ln;
                const unsigned p = p0 + (unsigned)pl;
                if (p < P) {
                    sp[p] = mean[s];
                    sp[P + p] = 1.0f / sqrtf((red[0][pl] + red[1][pl]) / (float)N + 1e-5f);
                }
            }
        }
    }
}

template <int PRO>
int launch_tile(const fdn_conv1x1_desc& d, hipStream_t s) {
    TArgs a;
    a.d = d;
    a.tiles_per_img = cdiv(d.P, TP);
    a.total_tiles = d.B * a.tiles_per_img;
    hipLaunchKernelGGL(gemm_tile_kernel<PRO>, dim3((unsigned)a.total_tiles), dim3(256), 0, s, a);
    return fdn_launch_status();
}

}  // namespace

// FDN_ERR_UNSUPPORTED = not a shape of this kernel (fdn_conv1x1 then picks one of gemm1x1.hip's)
int fdn_gemm_tile(const fdn_conv1x1_desc& d, hipStream_t s) {
    if (d.N > TN || d.N < 96 || d.K < 96 || d.kseg[1] > 0 || d.kseg[2] > 0 || d.act != FDN_ACT_NONE || d.x_bf16 || d.out_bf16) return FDN_ERR_UNSUPPORTED;
    if (d.epi != FDN_EPI_NONE && d.epi != FDN_EPI_RES) return FDN_ERR_UNSUPPORTED;
    if ((long)d.B * cdiv(d.P, TP) > 0x7FFFFFFFL) return FDN_ERR_UNSUPPORTED;
    if (d.pro == FDN_PRO_NONE) return launch_tile<FDN_PRO_NONE>(d, s);
    if (d.pro == FDN_PRO_LN3_GATE) return launch_tile<FDN_PRO_LN3_GATE>(d, s);
    return FDN_ERR_UNSUPPORTED;
}
