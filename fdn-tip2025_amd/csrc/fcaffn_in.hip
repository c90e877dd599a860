// FCAFFN between the inverse FFT and the gated tail (FDN_arch.py:419-423) in one launch:
//
//   t = project_in( norm(xi) * x1 + x1 ) * conv3_mul(conv1_mul(img)) + conv3_add(conv1_add(img))
//
// xi = the irfft2 output, x1 = the block input, img = the 3-channel image guidance of the level.  The unfused path ran
// chan_stats(xi) -> img_mod_maps(img) -> conv1x1(LN*x1+x1 prologue, *mul+add epilogue): the GEMM streamed five C-plane
// operands (xi, x1, mul, add, out = 4.8 GB at level 1, at the HBM rate already) and the two maps were written and read back
// for every block.  Here
//   * the LayerNorm statistics come from the activation strip the wave holds anyway (all C channels of its pixels: two-pass
//     mean / variance in registers, lanes l and l^32 hold the two channel parities of a pixel);
//   * mul and add are two more MFMA chains: conv3(conv1(img)) has no bias, so it is one 3x3 convolution from 3 channels with
//     the folded weights w3[n][tap] * w1[n][c] - a K = 30 GEMM (3 channels x 10 tap slots, the tenth empty) against the image
//     patch of the pixel, which costs 15 k-steps per map and no HBM traffic worth naming (3 planes, cached);
//   * the epilogue is acc * mul + add on three accumulators.
// Three C-plane streams remain (xi, x1, out).  Same operand mapping as gemm1x1.hip (pixels on the MFMA column axis straight
// from NCHW, weights transposed in LDS, one register set per stream refilled for the next tile behind the MFMAs that read it).
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t mk_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float type __attribute__((ext_vector_type(1))); };
template <> struct VecT<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type bloadv(rsrc_t r, unsigned voff, unsigned soff) {
    typename VecT<VEC>::type f;
    if constexpr (VEC == 2) {
        const fdn_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        f[0] = __uint_as_float(u.x);
        f[1] = __uint_as_float(u.y);
    } else {
        f[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    }
    return f;
}
template <int VEC>
__device__ __forceinline__ void bstorev(typename VecT<VEC>::type f, rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (VEC == 2) __builtin_amdgcn_raw_buffer_store_b64(fdn_u32x2{__float_as_uint(f[0]), __float_as_uint(f[1])}, r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(f[0]), r, voff, soff, 0);
}

struct Args {
    const float *xi, *x1, *img, *w, *gamma, *beta, *w1m, *w3m, *w1a, *w3a;
    const float *stats1, *gamma1, *beta1;     // optional: x1 is norm(x1) with these per-pixel statistics [B][2][H*W] and affine parameters
    float* out;
    int B, C, H, W;
    int tiles_per_img, total_tiles;
};

constexpr int MS = 15;          // k-steps of a map chain: K = 3 channels x 10 tap slots (slot 9 empty)

template <int NCH, int VEC>
__global__ __launch_bounds__(256, 2) void fcaffn_in_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef typename VecT<VEC>::type vf;
    constexpr int C = 32 * NCH, KS = 16 * NCH, NS = C + 1, NT = 256;
    const int H = a.H, W = a.W;
    const unsigned P = (unsigned)(H * W), P4 = P * 4u;
    float* tg = smem;                            // gamma[C]
    float* tb = tg + C;                          // beta[C]
    float* Wl = tb + C;                          // [C][NS]   project_in, transposed
    float* Wm = Wl + C * NS;                     // [30][NS]  folded conv3_mul * conv1_mul, row j = c * 10 + tap
    float* Wa = Wm + 2 * MS * NS;                // [30][NS]  the same for add
    float* tg1 = Wa + 2 * MS * NS;               // gamma1[C], beta1[C] (x1 LayerNorm on load)
    float* tb1 = tg1 + C;
    const bool ln1 = a.stats1 != nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;
    for (int i = tid; i < C; i += NT) {
        tg[i] = a.gamma[i]; tb[i] = a.beta[i];
        tg1[i] = ln1 ? a.gamma1[i] : 1.f; tb1[i] = ln1 ? a.beta1[i] : 0.f;
    }
    for (int idx = tid; idx < C * C; idx += NT) {
        const int k = idx % C, n = idx / C;
        Wl[k * NS + n] = a.w[(long)n * C + k];
    }
    for (int idx = tid; idx < 2 * MS * C; idx += NT) {
        const int j = idx % (2 * MS), n = idx / (2 * MS), c = j / 10, t = j - c * 10;
        Wm[j * NS + n] = t < 9 ? a.w3m[n * 9 + t] * a.w1m[n * 3 + c] : 0.f;
        Wa[j * NS + n] = t < 9 ? a.w3a[n * 9 + t] * a.w1a[n * 3 + c] : 0.f;
    }
    __syncthreads();

    // this lane's tap of k-step s of a map chain: slot t = 2 (s % 5) + kh of channel s / 5
    unsigned tapoff[5], lmask[5], rmask[5];
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) {
        const int t = 2 * s5 + kh, dy = t / 3 - 1, dx = t % 3 - 1;
        tapoff[s5] = t < 9 ? (unsigned)((dy * W + dx) * 4) : 0x80000000u;        // the empty slot reads outside the descriptor: 0
        lmask[s5] = (t < 9 && dx == -1) ? 0xFFFFFFFFu : 0u;
        rmask[s5] = (t < 9 && dx == 1) ? 0xFFFFFFFFu : 0u;
    }
    const float r_W = 1.0f / (float)W;

    struct Tile { int b; unsigned pix; bool ok; };
    auto tile_setup = [&](int t) {
        Tile r;
        r.b = t / a.tiles_per_img;
        const unsigned p_ = (unsigned)(t - r.b * a.tiles_per_img) * (4 * 32 * VEC) + (wave * 32 + ln) * VEC;
        r.ok = p_ < P;                          // P % VEC == 0: a vector is inside or outside as a whole
        r.pix = r.ok ? p_ : P - VEC;
        return r;
    };
    vf xa[KS], xb[KS];                          // xi -> the GEMM operand; x1
    vf pa[MS];                                  // image patch: rows above / below the image read 0 through the descriptor
    auto strip_issue = [&](const float* base, const Tile& t, vf (&dst)[KS], int s0, int s1) __attribute__((always_inline)) {
        const rsrc_t r0 = mk_rsrc(base + (long)t.b * C * P, (unsigned)C * P4);
        const unsigned voff = (kh * P + t.pix) * 4u;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (s >= s0 && s < s1) dst[s] = bloadv<VEC>(r0, voff, (unsigned)(2 * s) * P4);
    };
    vf mu1 = 0.f, rs1 = 0.f;
    auto stats1_issue = [&](const Tile& t) __attribute__((always_inline)) {
        if (ln1) {
            const rsrc_t r1 = mk_rsrc(a.stats1 + (long)t.b * 2 * P, 2u * P4);
            mu1 = bloadv<VEC>(r1, t.pix * 4u, 0u);
            rs1 = bloadv<VEC>(r1, t.pix * 4u, P4);
        }
    };
    auto patch_issue = [&](const Tile& t) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const rsrc_t rc = mk_rsrc(a.img + ((long)t.b * 3 + s / 5) * P, P4);
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                pa[s][v] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rc, t.pix * 4u + 4u * v + tapoff[s % 5], 0u, 0));
        }
    };

    int tile = blockIdx.x;
    bool live = tile < a.total_tiles;
    Tile cur = tile_setup(live ? tile : 0);
    if (live) {
        strip_issue(a.xi, cur, xa, 0, KS);
        strip_issue(a.x1, cur, xb, 0, KS);
        stats1_issue(cur);
        patch_issue(cur);
    }
    while (live) {
        // ---- channel LayerNorm of xi (statistics over the strip), * x1 + x1 ---------------------------------------
        vf sm = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) sm += xa[s];
        vf mean, sq = 0.f, rstd;
#pragma unroll
        for (int v = 0; v < VEC; ++v) mean[v] = (sm[v] + __shfl_xor(sm[v], 32)) / (float)C;
#pragma unroll
        for (int s = 0; s < KS; ++s) { const vf dl = xa[s] - mean; sq += dl * dl; }
#pragma unroll
        for (int v = 0; v < VEC; ++v) rstd[v] = 1.0f / sqrtf((sq[v] + __shfl_xor(sq[v], 32)) / (float)C + 1e-5f);
        if (ln1) {                                                      // x1 = norm3(x): rebuilt from x and its statistics (uniform branch)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                asm volatile("" ::: "memory");
                xb[s] = (xb[s] - mu1) * rs1 * tg1[2 * s + kh] + tb1[2 * s + kh];
            }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            asm volatile("" ::: "memory");                              // table reads stay here (see conv1x1_smallk_vec_kernel)
            const float ga = tg[2 * s + kh], be = tb[2 * s + kh];
            xa[s] = ((xa[s] - mean) * rstd * ga + be) * xb[s] + xb[s];
        }
        // ---- left / right image border of this tile's pixels ----------------------------------------------------------
        {
            const unsigned y = P < (1u << 22) ? (unsigned)(((float)cur.pix + 0.5f) * r_W) : cur.pix / (unsigned)W;
            const unsigned x = cur.pix - y * (unsigned)W;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const unsigned lft = (x + v == 0u) ? 0xFFFFFFFFu : 0u, rgt = (x + v == (unsigned)W - 1u) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                for (int s5 = 0; s5 < 5; ++s5) {
                    const unsigned keep = ~((lmask[s5] & lft) | (rmask[s5] & rgt));
#pragma unroll
                    for (int c = 0; c < 3; ++c) pa[c * 5 + s5][v] = __uint_as_float(__float_as_uint(pa[c * 5 + s5][v]) & keep);
                }
            }
        }
        const int ntile = tile + gridDim.x;
        const bool nlive = ntile < a.total_tiles;
        const Tile nxt = tile_setup(nlive ? ntile : tile);
        if (nlive) { strip_issue(a.x1, nxt, xb, 0, KS); stats1_issue(nxt); }   // x1 of the next tile: its registers are free from here on

        const rsrc_t ro = mk_rsrc(a.out + (long)cur.b * C * P, (unsigned)C * P4);
        const unsigned voff = cur.ok ? (4u * kh * P + cur.pix) * 4u : 0x80000000u;     // outside pixels: stores dropped
#pragma unroll 1
        for (int m = 0; m < NCH; ++m) {
            const bool refill = nlive && m == NCH - 1;
            f32x16 acc[VEC], am[VEC], aa[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[v][r] = 0.f; am[v][r] = 0.f; aa[v][r] = 0.f; }
            // project_in: A operands one group of 8 k-steps ahead of their MFMAs, each feeds VEC of them
            const float* w = Wl + kh * NS + m * 32 + ln;
            float ar[2][8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ar[0][i] = w[i * 2 * NS];
#pragma unroll
            for (int grp = 0; grp < KS / 8; ++grp) {
                if (grp + 1 < KS / 8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) ar[(grp + 1) & 1][i] = w[((grp + 1) * 8 + i) * 2 * NS];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[grp & 1][i], xa[grp * 8 + i][v], acc[v], 0, 0, 0);
                if (refill) strip_issue(a.xi, nxt, xa, grp * 8, grp * 8 + 8);
            }
            // the two modulation maps of these 32 output channels
            const float* wm = Wm + kh * NS + m * 32 + ln;
            const float* wa = Wa + kh * NS + m * 32 + ln;
            float mr[MS], dr[MS];
#pragma unroll
            for (int s = 0; s < MS; ++s) { mr[s] = wm[s * 2 * NS]; dr[s] = wa[s * 2 * NS]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < MS; ++s)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    am[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(mr[s], pa[s][v], am[v], 0, 0, 0);
                    aa[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(dr[s], pa[s][v], aa[v], 0, 0, 0);
                }
            if (refill) patch_issue(nxt);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = m * 32 + (r & 3) + 8 * (r >> 2);
                vf o;
#pragma unroll
                for (int v = 0; v < VEC; ++v) o[v] = acc[v][r] * am[v][r] + aa[v][r];                       // FDN_arch.py:423
                bstorev<VEC>(o, ro, voff, (unsigned)nrow * P4);
            }
        }
        cur = nxt; tile = ntile; live = nlive;
    }
}

template <int NCH, int VEC>
int launch_fcaffn_in(Args a, hipStream_t s) {
    constexpr int C = 32 * NCH;
    const size_t lds = (4UL * C + (size_t)C * (C + 1) + 4UL * MS * (C + 1)) * sizeof(float);
    a.tiles_per_img = cdiv((long)a.H * a.W, 4 * 32 * VEC);
    a.total_tiles = a.B * a.tiles_per_img;
    auto kern = fcaffn_in_kernel<NCH, VEC>;
    if (lds > 48 * 1024 && !fdn_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return FDN_ERR_LAUNCH;
    const int cus = fdn_device_cus();
    if (cus <= 0) return FDN_ERR_LAUNCH;
    int per_cu = 0;
    if (!fdn_occupancy(&per_cu, reinterpret_cast<const void*>(kern), 256, lds) || per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    int grid = cus * per_cu;
    if (grid > a.total_tiles) grid = a.total_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    return fdn_launch_status();
}

}  // namespace

extern "C" int fdn_fcaffn_in(const float* xi, const float* x1, const float* stats1, const float* gamma1, const float* beta1,
                             const float* img, const float* w, const float* gamma, const float* beta, const float* w1_mul,
                             const float* w3_mul, const float* w1_add, const float* w3_add, float* out, int B, int C, int H, int W,
                             fdn_stream_t stream) {
    FDN_CHECK_ARG(!stats1 || (gamma1 && beta1 && (reinterpret_cast<uintptr_t>(stats1) & 7) == 0));
    FDN_CHECK_ARG(xi && x1 && img && w && gamma && beta && w1_mul && w3_mul && w1_add && w3_add && out);
    FDN_CHECK_ARG(B > 0 && H > 0 && W > 1 && (long)H * W < (1L << 28));
    if ((C != 32 && C != 64) || W % 2 != 0) return FDN_ERR_UNSUPPORTED;      // (C = 128: 98 KB of weights leave one wave per SIMD - 0.76 ms against 0.53 unfused)
    if (((reinterpret_cast<uintptr_t>(xi) | reinterpret_cast<uintptr_t>(x1) | reinterpret_cast<uintptr_t>(out)) & 7) != 0) return FDN_ERR_UNSUPPORTED;
    Args a = {xi, x1, img, w, gamma, beta, w1_mul, w3_mul, w1_add, w3_add, stats1, gamma1, beta1, out, B, C, H, W, 0, 0};
    hipStream_t s = static_cast<hipStream_t>(stream);
    return C == 32 ? launch_fcaffn_in<1, 2>(a, s) : launch_fcaffn_in<2, 1>(a, s);
}
