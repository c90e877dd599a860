// Dense 3x3 convolution (stride 1, zero pad 1) as implicit GEMM on fp32 MFMA: the FDformer's
// Downsample / Upsample convs (FDN_arch.py:720,731) and MAR's 3x3 convs (:57,:135,:174-175,:196).
//
// Same operand mapping as gemm1x1.hip: D[n][p] = sum_k' W[n][k'] * X[k'][p] with
// k' = tap * Cin + ci (tap-major, so the two k of an MFMA k-step share the tap when Cin is even);
// pixels sit on the MFMA lane axis and the B operand of lane (pixel y,x) is loaded straight from
// global memory at (ci, y+dy, x+dx) by a buffer load; the zero padding is a per-lane edge mask.
// Weights are streamed through LDS in 32-deep k' chunks (double buffered, one barrier per chunk),
// re-gathered from the [Cout][Cin][3][3] checkpoint layout on the fly.
#include "common.hpp"

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
    const int tiles = (Cout + 31) / 32;
    if (tiles == 1) return launch<1, 4>(a, s);
    if (tiles == 2) return launch<2, 4>(a, s);
    if (tiles == 3) return launch<3, 4>(a, s);
    return launch<4, 4>(a, s);
}
