// Dense 3x3 convolution (stride 1, zero pad 1) as implicit GEMM on the matrix cores (a flat-pixel fp32-MFMA form for any
// shape, an LDS-tiled split-bf16 form for Cin % 8 == 0 - fp32-exact to rounding, further down): the FDformer's
// Downsample / Upsample convs (FDN_arch.py:720,731) and MAR's 3x3 convs (:57,:135,:174-175,:196).
//
// Same operand mapping as gemm1x1.hip: D[n][p] = sum_k' W[n][k'] * X[k'][p] with
// k' = tap * Cin + ci (tap-major, so the two k of an MFMA k-step share the tap when Cin is even);
// pixels sit on the MFMA lane axis and the B operand of lane (pixel y,x) is loaded straight from
// global memory at (ci, y+dy, x+dx) by a buffer load; the zero padding is a per-lane edge mask.
// Weights are streamed through LDS in 32-deep k' chunks (double buffered, one barrier per chunk),
// re-gathered from the [Cout][Cin][3][3] checkpoint layout on the fly.
#include "common.hpp"
#include <type_traits>

namespace {

constexpr int KC = 32;
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

struct C3Args {
    const float* x; const float* w; const float* bias; const float* res; float* out;
    int B, Cin, H, W, Cout;
    int act, res_before_act; float post_add;
    int tiles_per_img, total_tiles;
    int n0, CoutT;             // tiled kernel: first output channel of this launch / output channels of the tensor (plane strides)
};

template <int MT, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void conv3x3_kernel(C3Args a) {
    extern __shared__ __attribute__((aligned(16))) float Wl[];
    constexpr int NT = NW * 64;
    constexpr int WS = MT * 32 + 1;
    constexpr int CH = KC * WS;
    constexpr int WPT = (KC * MT * 32) / NT;
    const int Cin = a.Cin, N = a.Cout, W = a.W, H = a.H;
    const int Ce = (Cin + 1) & ~1;             // even channel count of the k' = tap*Ce + ci decomposition (pad channel = 0)
    const unsigned P = (unsigned)H * W, P4 = P * 4u;
    const int Kt = 9 * Ce;
    const int nch = (Kt + KC - 1) / KC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;
    const int npass = (N + MT * 32 - 1) / (MT * 32);

    for (int pass = 0; pass < npass; ++pass) {
        const int nbase = pass * MT * 32;
        float wr[WPT];
        auto w_fetch = [&](int c) {
            const int kp = c * KC + (tid & 31);
            const int tap = kp / Ce, ci = kp - tap * Ce;
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const int n = nbase + (tid >> 5) + (NT / 32) * i;
                wr[i] = (n < N && kp < Kt && ci < Cin) ? a.w[((long)n * Cin + ci) * 9 + tap] : 0.f;
            }
        };
        auto w_stash = [&](int buf) {
            float* dst = Wl + buf * CH + (tid & 31) * WS;
#pragma unroll
            for (int i = 0; i < WPT; ++i) dst[(tid >> 5) + (NT / 32) * i] = wr[i];
        };
        __syncthreads();
        w_fetch(0);
        w_stash(0);
        __syncthreads();

        int tile = blockIdx.x, c = 0;
        bool live = tile < a.total_tiles;
        struct Tile { int b; unsigned pix; bool ok, top, bot, lft, rgt; };
        auto tile_setup = [&](int t) {
            Tile r;
            r.b = t / a.tiles_per_img;
            const unsigned p_ = (unsigned)(t - r.b * a.tiles_per_img) * (NW * 32) + wave * 32 + ln;
            r.ok = p_ < P;
            r.pix = r.ok ? p_ : P - 1;
            const int y = r.pix / W, x = r.pix - y * W;
            r.top = y == 0; r.bot = y == H - 1; r.lft = x == 0; r.rgt = x == W - 1;
            return r;
        };
        float xa[16], xb[16];
        auto x_issue = [&](const Tile& t, int c_, float (&xv)[16]) {
            const rsrc_t rx = mk_rsrc(a.x + (long)t.b * Cin * P, (unsigned)Cin * P4);
            const unsigned vbase = (kh * P + t.pix) * 4u;
            int k0 = c_ * KC;
            int tap = k0 / Ce, ci = k0 - tap * Ce;             // wave-uniform
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                const bool bad = tap >= 9 || (dy < 0 && t.top) || (dy > 0 && t.bot) || (dx < 0 && t.lft) || (dx > 0 && t.rgt);
                const unsigned voff = vbase + (unsigned)((dy * W + dx) * 4);
                const float v = bload(rx, voff, (unsigned)ci * P4);
                xv[s] = (bad || ci + kh >= Cin) ? 0.f : v;        // pad channel / edge taps contribute nothing
                ci += 2;
                if (ci >= Ce) { ci -= Ce; ++tap; }
            }
        };

        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

        Tile cur = tile_setup(live ? tile : 0);
        if (live) x_issue(cur, 0, xa);
        int step = 0;
        while (live) {
            int ntile = tile, nc = c + 1;
            if (nc == nch) { nc = 0; ntile = tile + gridDim.x; }
            const bool nlive = ntile < a.total_tiles;
            Tile nxt = cur;
            if (nlive) {
                if (nc == 0) nxt = tile_setup(ntile);
                x_issue(nxt, nc, xb);
                w_fetch(nc);
            }
            const float* Wc = Wl + (step & 1) * CH;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float* wrow = Wc + (2 * s + kh) * WS + ln;
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[m * 32], xa[s], acc[m], 0, 0, 0);
            }
            if (c == nch - 1) {
                if (cur.ok) {
                    const unsigned nb4 = (unsigned)N * P4;
                    const rsrc_t ro = mk_rsrc(a.out + (long)cur.b * N * P, nb4);
                    const rsrc_t rr = mk_rsrc(a.res ? a.res + (long)cur.b * N * P : a.out, a.res ? nb4 : 0u);
                    const unsigned voff = (4u * kh * P + cur.pix) * 4u;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int nrow = nbase + m * 32 + (r & 3) + 8 * (r >> 2);
                            const unsigned soff = (unsigned)nrow * P4;
                            float v = acc[m][r];
                            if (a.bias) { const int n = nrow + 4 * kh; v += (n < N) ? a.bias[n] : 0.f; }
                            if (a.res && a.res_before_act) v += bload(rr, voff, soff);
                            v = apply_act(v, a.act);
                            if (a.res && !a.res_before_act) v += bload(rr, voff, soff);
                            bstore(v + a.post_add, ro, voff, soff);
                        }
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
            }
            if (nlive) w_stash((step + 1) & 1);
            __syncthreads();
            if (nlive && nc == 0) cur = nxt;
#pragma unroll
            for (int s = 0; s < 16; ++s) xa[s] = xb[s];
            tile = ntile; c = nc; live = nlive;
            ++step;
        }
    }
}

constexpr int TW = 32, CK = 8, HC = TW + 2;      // tile columns, input channels per chunk, halo columns

// ------------------------------------------------------------------------------------------------
// LDS-tiled form for Cin % 8 == 0, Cout >= 16 (the Downsample / Upsample body convs, 64 -> 32 at level 1 and 128 -> 64 at level 2:
// 278 GFLOP each; MAR's 24 -> 24 and 48 -> 48).  The flat-pixel kernel above loads every input value nine times from global
// memory; here a workgroup owns a 2*NW x 32 output tile and stages the halo tile of 8 input channels in LDS (zero-filled outside
// the image, double buffered, requested two chunks ahead).  Round 3: both operands are cut into three exact
// bf16 parts WHEN THEY ARE STAGED - a value is split once and then read by all nine taps - and the six leading products of a
// 16-deep k-step run on v_mfma_f32_32x32x16_bf16 (fp32 accumulate: fp32 accuracy, a quarter of the matrix-pipe cycles of
// v_mfma_f32_32x32x2_f32, and the pipe is not the vector ALU's): 64 -> 32 2.70 -> 2.32 ms, 128 -> 64 2.64 -> 1.90 ms,
// 24 -> 24 (was on the direct vector-ALU kernel) 0.89 -> 0.21 ms.  A k-step is TWO taps x
// 8 channels: lane half kh reads tap 2 s + kh (its own (dy, dx) offset into the staged tile), k = 8 kh + channel; the tenth
// tap of the fifth k-step has zero weights.  LDS holds, per buffer, the tile as [part][position] 16-byte cells (8 channels of
// one pixel: one ds_read_b128 per part and row) and the weights as [k-step][part][channel tile][lane] cells in operand order.
// Any Cout: channels past the tensor read zero weights and their rows fall outside the output descriptor.
// ------------------------------------------------------------------------------------------------
template <int MT, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 4 && MT == 1) ? 2 : 1) void conv3x3_split_kernel(C3Args a, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) fdn_u32x4 smem4[];
    constexpr int NT = NW * 64, TH_ = 2 * NW, HR_ = TH_ + 2, NPOS = HR_ * HC;
    constexpr int N = MT * 32, KS5 = 5;
    constexpr int XB = 3 * NPOS, WB = KS5 * 3 * MT * 64;      // cells per buffer
    constexpr int XE = (NPOS + NT - 1) / NT;                   // staged pixel positions per thread
    constexpr int WE = (N * 9 * CK + NT - 1) / NT;             // staged weights per thread (72 consecutive floats per output channel: coalesced)
    fdn_u32x4* Xs = smem4;                      // [2][3][NPOS]
    fdn_u32x4* Ws = smem4 + 2 * XB;             // [2][5][3][MT][64]
    const int Cin = a.Cin, H = a.H, W = a.W;
    const unsigned P = (unsigned)H * W, P4 = P * 4u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;
    const int nchunk = Cin / CK;
    int b, y0, x0;
    {
        const unsigned T = (unsigned)(tiles_x * tiles_y) * a.B;
        const unsigned S0 = xcd_contiguous(blockIdx.x, gridDim.x);
        const unsigned grp = S0 / T, S = S0 - grp * T;
        a.n0 = (int)grp * N;
        const unsigned per = (unsigned)(tiles_x * tiles_y);
        b = (int)(S / per);
        const unsigned t = S - (unsigned)b * per;
        const int ty = (int)(t / (unsigned)tiles_x);
        y0 = ty * TH_;
        x0 = (int)(t - (unsigned)ty * tiles_x) * TW;
    }
    const int nvalid = min(N, a.CoutT - a.n0);                 // output channels of this group that exist
    unsigned xg[XE];
    int xl[XE];
#pragma unroll
    for (int i = 0; i < XE; ++i) {
        const int e = tid + NT * i, r = e / HC, c = e - r * HC;
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        const bool in = e < NPOS, ok = in && gy >= 0 && gy < H && gx >= 0 && gx < W;
        xg[i] = ok ? (unsigned)(gy * W + gx) * 4u : 0x80000000u;
        xl[i] = in ? e : -1;
    }
    const rsrc_t rx = mk_rsrc(a.x + (long)b * Cin * P, (unsigned)Cin * P4);
    // two register stages: chunk c + 2 is requested while chunk c is multiplied (a chunk is ~1 us of matrix work, an HBM round trip
    // under load is longer), and parked in LDS a chunk later
    struct Stage { float xr[XE][CK], wr[WE]; };
    Stage stg[2];
    auto fetch = [&](int c, Stage& sg) __attribute__((always_inline)) {
        float (&xr)[XE][CK] = sg.xr;
        float (&wr)[WE] = sg.wr;
#pragma unroll
        for (int i = 0; i < XE; ++i)
#pragma unroll
            for (int ci = 0; ci < CK; ++ci) xr[i][ci] = bload(rx, xg[i], (unsigned)(c * CK + ci) * P4);
#pragma unroll
        for (int i = 0; i < WE; ++i) {
            const int e = tid + NT * i, n = e / (9 * CK), q = e - n * (9 * CK);
            wr[i] = (e < N * 9 * CK && n < nvalid) ? a.w[((long)(a.n0 + n) * Cin + c * CK) * 9 + q] : 0.f;
        }
    };
    auto stash = [&](int buf, const Stage& sg) __attribute__((always_inline)) {
        const float (&xr)[XE][CK] = sg.xr;
        const float (&wr)[WE] = sg.wr;
        fdn_u32x4* xd = Xs + buf * XB;
#pragma unroll
        for (int i = 0; i < XE; ++i) {
            fdn_u32x4 p1, p2, p3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned u1, u2, u3;
                fdn_split3(xr[i][2 * j], xr[i][2 * j + 1], u1, u2, u3);
                p1[j] = u1; p2[j] = u2; p3[j] = u3;
            }
            if (xl[i] >= 0) {
                xd[xl[i]] = p1;
                xd[NPOS + xl[i]] = p2;
                xd[2 * NPOS + xl[i]] = p3;
            }
        }
        // a weight (n, ci, tap) is one bf16 of the cell (k-step tap / 2, lane half tap % 2, lane n): three 16-bit stores, no pairing
        unsigned short* wd = reinterpret_cast<unsigned short*>(Ws + buf * WB);
#pragma unroll
        for (int i = 0; i < WE; ++i) {
            const int e = tid + NT * i, n = e / (9 * CK), q = e - n * (9 * CK), ci = q / 9, tap = q - ci * 9;
            const int cell = (n >> 5) * 64 + (tap & 1) * 32 + (n & 31), s_ = tap >> 1;
            const float w1 = fdn_trunc_bf16(wr[i]), r1 = wr[i] - w1, w2 = fdn_trunc_bf16(r1), w3 = r1 - w2;
            if (e < N * 9 * CK && n < nvalid) {
                wd[((s_ * 3 + 0) * (MT * 64) + cell) * 8 + ci] = (unsigned short)(__float_as_uint(w1) >> 16);
                wd[((s_ * 3 + 1) * (MT * 64) + cell) * 8 + ci] = (unsigned short)(__float_as_uint(w2) >> 16);
                wd[((s_ * 3 + 2) * (MT * 64) + cell) * 8 + ci] = (unsigned short)(__float_as_uint(w3) >> 16);
            }
        }
    };
    for (int i = tid; i < 2 * WB; i += NT) Ws[i] = fdn_u32x4{0u, 0u, 0u, 0u};      // the tenth tap and channels past the tensor stay zero
    __syncthreads();
    f32x16 acc[2][MT];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[st][m][r] = 0.f;

    fetch(0, stg[0]);
    if (nchunk > 1) fetch(1, stg[1]);
    stash(0, stg[0]);
    __syncthreads();
    const int row0 = 2 * wave;
    int toff[KS5];                              // this lane half's tap of every k-step, as an offset into the staged tile
#pragma unroll
    for (int s_ = 0; s_ < KS5; ++s_) {
        const int tap = min(2 * s_ + kh, 8), dy = tap / 3, dx = tap - dy * 3;
        toff[s_] = (row0 + dy) * HC + ln + dx;
    }
    auto chunk_step = [&](int c, Stage& mine, const Stage& nxt) __attribute__((always_inline)) {
        // `mine` held chunk c (parked in LDS already): refill it with chunk c + 2; `nxt` holds chunk c + 1
        const bool more = c + 1 < nchunk;
        if (c + 2 < nchunk) fetch(c + 2, mine);
        const fdn_u32x4* xb = Xs + (c & 1) * XB;
        const fdn_u32x4* wb = Ws + (c & 1) * WB + lane;
#pragma unroll
        for (int s_ = 0; s_ < KS5; ++s_) {
            const fdn_u32x4 b0[3] = {xb[toff[s_]], xb[NPOS + toff[s_]], xb[2 * NPOS + toff[s_]]};
            const fdn_u32x4 b1[3] = {xb[toff[s_] + HC], xb[NPOS + toff[s_] + HC], xb[2 * NPOS + toff[s_] + HC]};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const fdn_u32x4 av[3] = {wb[((s_ * 3 + 0) * MT + m) * 64], wb[((s_ * 3 + 1) * MT + m) * 64], wb[((s_ * 3 + 2) * MT + m) * 64]};
                acc[0][m] = fdn_mfma_split6(av, b0, acc[0][m]);
                acc[1][m] = fdn_mfma_split6(av, b1, acc[1][m]);
            }
        }
        if (more) stash((c + 1) & 1, nxt);
        __syncthreads();
    };
    for (int c = 0; c < nchunk; c += 2) {
        chunk_step(c, stg[0], stg[1]);
        if (c + 1 < nchunk) chunk_step(c + 1, stg[1], stg[0]);
    }
    const unsigned nb4 = (unsigned)nvalid * P4;
    const long obase = ((long)b * a.CoutT + a.n0) * P;
    const rsrc_t ro = mk_rsrc(a.out + obase, nb4);
    const rsrc_t rr = mk_rsrc(a.res ? a.res + obase : a.out, a.res ? nb4 : 0u);
    auto epilogue = [&](auto has_act) __attribute__((always_inline)) {
        const int act_ = decltype(has_act)::value ? a.act : (int)FDN_ACT_NONE;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int gy = y0 + row0 + st, gx = x0 + ln;
            const unsigned voff = (gy < H && gx < W) ? (4u * kh * P + (unsigned)(gy * W + gx)) * 4u : 0x80000000u;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float rv[16];
                if (a.res) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int nrow = m * 32 + (r & 3) + 8 * (r >> 2);
                        rv[r] = bload(rr, (nrow + 4 * kh < nvalid) ? voff : 0x80000000u, (unsigned)nrow * P4);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nrow = m * 32 + (r & 3) + 8 * (r >> 2);
                    float v = acc[st][m][r];
                    if (a.bias) v += a.bias[a.n0 + min(nrow + 4 * kh, nvalid - 1)];
                    if (a.res && a.res_before_act) v += rv[r];
                    v = apply_act(v, act_);
                    if (a.res && !a.res_before_act) v += rv[r];
                    bstore(v + a.post_add, ro, (nrow + 4 * kh < nvalid) ? voff : 0x80000000u, (unsigned)nrow * P4);      // (a channel past the tensor: dropped)
                }
            }
        }
    };
    if (a.act == FDN_ACT_NONE) epilogue(std::false_type{}); else epilogue(std::true_type{});
}

template <int MT, int NW>
int launch_split(const C3Args& a, hipStream_t s) {
    constexpr int N = MT * 32, TH_ = 2 * NW;
    const size_t lds = 2UL * (3UL * (TH_ + 2) * HC + 5UL * 3 * MT * 64) * sizeof(fdn_u32x4);
    auto kern = conv3x3_split_kernel<MT, NW>;
    if (lds > 48 * 1024 && !fdn_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return FDN_ERR_LAUNCH;
    const int tx = cdiv(a.W, TW), ty = cdiv(a.H, TH_);
    const long total = (long)tx * ty * a.B * cdiv(a.Cout, N);
    if (total > 0x7FFFFFFFL) return FDN_ERR_UNSUPPORTED;
    fdn_note_bf16_launch();
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(NW * 64), lds, s, a, tx, ty);
    return fdn_launch_status();
}

template <int MT, int NW>
int launch(C3Args a, hipStream_t s) {
    const size_t lds = 2UL * KC * (MT * 32 + 1) * sizeof(float);
    const int g_cus = fdn_device_cus();
    if (g_cus <= 0) return FDN_ERR_LAUNCH;
    a.tiles_per_img = cdiv((long)a.H * a.W, NW * 32);
    a.total_tiles = a.B * a.tiles_per_img;
    // 4-wave workgroups, two (independent) per CU: their per-chunk barriers drift apart (as in gemm1x1.hip)
    int per_cu = 0;
    if (!fdn_occupancy(&per_cu, reinterpret_cast<const void*>(conv3x3_kernel<MT, NW>), NW * 64, lds) || per_cu < 1) per_cu = 1;
    if (per_cu * NW > 16) per_cu = 16 / NW;
    int grid = g_cus * per_cu;
    if (grid > a.total_tiles) grid = a.total_tiles;
    hipLaunchKernelGGL((conv3x3_kernel<MT, NW>), dim3(grid), dim3(NW * 64), lds, s, a);
    return fdn_launch_status();
}

}  // namespace

// returns FDN_ERR_UNSUPPORTED when the shape is not covered (caller falls back to the direct kernel)
int fdn_conv3x3_mfma(const float* x, const float* w, const float* bias, const float* res, float* out, int B, int Cin, int H,
                     int W, int Cout, int act, int res_before_act, float post_add, hipStream_t s) {
    if (Cin < 2 || Cout < 8) return FDN_ERR_UNSUPPORTED;   // tiny Cout: the direct kernel wastes less
    const unsigned long long P4 = 4ull * H * W;
    if ((unsigned long long)(Cin + 2) * P4 > 0xFFFFFFFFull || (unsigned long long)(Cout + 200) * P4 > 0xFFFFFFFFull)
        return FDN_ERR_UNSUPPORTED;
    C3Args a;
    a.x = x; a.w = w; a.bias = bias; a.res = res; a.out = out;
    a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout;
    a.act = act; a.res_before_act = res_before_act; a.post_add = post_add;
    a.tiles_per_img = a.total_tiles = 0;
    a.n0 = 0; a.CoutT = Cout;
    // fdn_set_matrix_pipe(1) promises that NO bf16-MFMA kernel is launched: the gate lives here, so every caller keeps it (wide outputs then take the
    // flat fp32-MFMA form below)
    if (!fdn_matrix_pipe_f32() && Cin % CK == 0 && Cin >= 2 * CK && Cout >= 16)           // LDS-tiled, both operands as three bf16 parts on the bf16 matrix pipe
        return Cout <= 32 ? launch_split<1, 4>(a, s) : launch_split<2, 8>(a, s);      // (8 x 32 tiles, two workgroups per CU / 16 x 32, one)
    const int tiles = (Cout + 31) / 32;
    if (tiles == 1) return launch<1, 4>(a, s);
    if (tiles == 2) return launch<2, 4>(a, s);
    if (tiles == 3) return launch<3, 4>(a, s);
    return launch<4, 4>(a, s);
}
