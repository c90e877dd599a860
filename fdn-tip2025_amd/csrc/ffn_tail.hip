// FDFFN / FCAFFN tail in one launch: the gated depthwise conv (Conv2d(C, 2C, 3, groups=C) then
// gelu(x1) * x2, FDN_arch.py:472-473 / :426-427), project_out (1x1, C -> N, :474 / :428), the residual
// add of TransformerBlock.forward (:673,:675) and the LayerNorm statistics of the result.
//
// The gated tensor (C channels) is never written to HBM: a workgroup owns an 8 x 32 pixel tile, stages
// the halo tiles of the 33 source planes of a 32-channel K-chunk in LDS, and every lane evaluates the
// two 3x3 stencils + GELU for its (pixel, k) pair straight into the register that feeds
// v_mfma_f32_32x32x2_f32 as the B operand (lane = pixel column, lane half = k parity, exactly the
// operand layout of gemm1x1.hip).  VALU (stencils) and MFMA (projection) run in the same waves; the
// next chunk's planes travel global -> registers while the current chunk is evaluated.
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
constexpr int TR = 8, TC = 32;            // pixel tile: one row per wave
constexpr int HR = TR + 2, HC = TC + 2;   // halo tile
constexpr int HPL = HR * HC;              // 340 floats per plane
constexpr int NA = 16, NB = 17, NPL = NA + NB;
constexpr int NW = 8, NT = NW * 64;

struct FtArgs {
    const float* y;       // [B][C][H][W]   (C = hidden width)
    const float* wdw;     // [2C][9]
    const float* w;       // [N][C]
    const float* res;     // [B][N][H][W] or null
    float* out;           // [B][N][H][W]
    float* stats_out;     // [B][2][H*W] or null
    int B, C, N, H, W;
    int tiles_x, tiles_per_img, total_tiles;
};

template <int MT>
__global__ __launch_bounds__(NT) void ffn_tail_kernel(FtArgs a) {
    constexpr int WS = MT * 32 + 1;
    constexpr int WPT = (32 * MT * 32) / NT;          // projection weights per thread per chunk
    __shared__ float planes[NPL * HPL];
    __shared__ float dwl[32 * 18];                     // [j local][wA(9) | wB(9)]
    __shared__ float Wl[32 * WS];                      // [k local][n]

    const int C = a.C, N = a.N, H = a.H, W = a.W;
    const long hw = (long)H * W;
    const int nch = (C + 31) / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;

    float pre[NPL], pdw[2], pw[WPT];             // thread < 340 owns one halo position of all 33 planes
    // ---- loaders ----------------------------------------------------------------------------------------
    auto fetch = [&](int tile, int c) {
        const int b = tile / a.tiles_per_img, t = tile - b * a.tiles_per_img;
        const int ty0 = (t / a.tiles_x) * TR, tx0 = (t % a.tiles_x) * TC;
        const int j0 = c * 32;
        const int abase = j0 >> 1, bbase = (C + j0) >> 1;
        // plane loads: per-thread halo position in voffset, plane (channel) in the scalar offset; positions
        // outside the image and channels >= C fall outside the descriptor and read 0
        const unsigned hw4 = (unsigned)hw * 4u;
        const rsrc_t ry = mk_rsrc(a.y + (long)b * C * hw, (unsigned)C * hw4);
        const int hr = tid / HC, hc = tid - hr * HC;
        const int gy = ty0 - 1 + hr, gx = tx0 - 1 + hc;
        const bool inimg = tid < HPL && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const unsigned voff = inimg ? (unsigned)(gy * W + gx) * 4u : 0x80000000u;
        if (tid < HPL) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                const int ch = pl < NA ? abase + pl : bbase + (pl - NA);
                pre[pl] = bload(ry, voff, (unsigned)ch * hw4);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                  // 576 depthwise weights of the chunk
            const int idx = tid + NT * i;
            const int jl = idx / 18, q = idx - jl * 18;
            const int j = j0 + jl;
            pdw[i] = (idx < 32 * 18 && j < C) ? a.wdw[(q < 9 ? (long)j * 9 + q : (long)(C + j) * 9 + (q - 9))] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {                // W[n][k], lanes along k
            const int kk = tid & 31, n = (tid >> 5) + (NT / 32) * i, k = j0 + kk;
            pw[i] = (n < N && k < C) ? a.w[(long)n * C + k] : 0.f;
        }
    };
    auto stash = [&]() {
        if (tid < HPL) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) planes[pl * HPL + tid] = pre[pl];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + NT * i;
            if (idx < 32 * 18) dwl[idx] = pdw[i];
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) Wl[(tid & 31) * WS + (tid >> 5) + (NT / 32) * i] = pw[i];
    };

    int tile = blockIdx.x;
    if (tile >= a.total_tiles) return;
    fetch(tile, 0);
    stash();
    __syncthreads();

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    int c = 0;
    while (true) {
        int ntile = tile, nc = c + 1;
        if (nc == nch) { nc = 0; ntile = tile + gridDim.x; }
        const bool nlive = ntile < a.total_tiles;
        if (nlive) fetch(ntile, nc);                               // flies during the stencils below

        // ---- this chunk: B operand = gelu(dwA(y_a)) * dwB(y_b) for k = j0 + 2s + kh at this lane's pixel ----
        {
            const int j0 = c * 32;
            const int bshift = ((C + j0) & 1);                     // (C + j0 + jl) >> 1 - bbase = (jl + bshift) >> 1
            const float* tl = planes + wave * HC + ln;             // halo origin (-1,-1): centre = +HC+1
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int jl = 2 * s + kh;
                const float* pa = tl + s * HPL;
                const float* pb = tl + (NA + ((jl + bshift) >> 1)) * HPL;
                const float* wv = dwl + jl * 18;
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        sa = fmaf(wv[dy * 3 + dx], pa[dy * HC + dx], sa);
                        sb = fmaf(wv[9 + dy * 3 + dx], pb[dy * HC + dx], sb);
                    }
                const float g = gelu_fast(sa) * sb;
                const float* wrow = Wl + jl * WS + ln;
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[m * 32], g, acc[m], 0, 0, 0);
            }
        }

        // ---- end of a tile: residual, store, statistics ---------------------------------------------------
        if (c == nch - 1) {
            const int b = tile / a.tiles_per_img, t = tile - b * a.tiles_per_img;
            const int gy = (t / a.tiles_x) * TR + wave, gx = (t % a.tiles_x) * TC + ln;
            const bool ok = gy < H && gx < W;
            const long pix = (long)gy * W + gx;
            const unsigned hw4 = (unsigned)hw * 4u, nb4 = (unsigned)N * hw4;
            const rsrc_t ro = mk_rsrc(a.out + (long)b * N * hw, nb4);
            const rsrc_t rr = mk_rsrc(a.res ? a.res + (long)b * N * hw : a.out, a.res ? nb4 : 0u);
            const unsigned vo = ok ? (unsigned)(4 * kh * hw + pix) * 4u : 0x80000000u;
            float sm = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float rres[16];                                      // residual operands of one 32-channel tile as a batch:
#pragma unroll                                                       // next to the stores every load is waited for on its own
                for (int r = 0; r < 16; ++r) rres[r] = bload(rr, vo, (unsigned)(m * 32 + (r & 3) + 8 * (r >> 2)) * hw4);   // 0 without res
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nrow = m * 32 + (r & 3) + 8 * (r >> 2);
                    const unsigned so = (unsigned)nrow * hw4;
                    float v = acc[m][r];
                    v += rres[r];
                    bstore(v, ro, vo, so);                           // rows >= N / pixels outside: dropped
                    v = (ok && nrow + 4 * kh < N) ? v : 0.f;
                    acc[m][r] = v;
                    sm += v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (a.stats_out) {
                sm += __shfl_xor(sm, 32);
                const float mean = sm / (float)N;
                float sq = 0.f;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        const float dlt = acc[m][r] - mean;
                        sq += (n < N) ? dlt * dlt : 0.f;
                    }
                sq += __shfl_xor(sq, 32);
                if (ok && kh == 0) {
                    float* sp = a.stats_out + (long)b * 2 * hw;
                    sp[pix] = mean;
                    sp[hw + pix] = 1.0f / sqrtf(sq / (float)N + 1e-5f);
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        }

        if (!nlive) break;
        __syncthreads();                                          // everyone is done reading this chunk
        stash();
        __syncthreads();
        tile = ntile; c = nc;
    }
}

template <int MT>
int launch(FtArgs a, hipStream_t s) {
    const int g_cus = fdn_device_cus();
    if (g_cus <= 0) return FDN_ERR_LAUNCH;
    a.tiles_x = cdiv(a.W, TC);
    a.tiles_per_img = a.tiles_x * cdiv(a.H, TR);
    a.total_tiles = a.B * a.tiles_per_img;
    int grid = g_cus * 2;
    if (grid > a.total_tiles) grid = a.total_tiles;
    hipLaunchKernelGGL(ffn_tail_kernel<MT>, dim3(grid), dim3(NT), 0, s, a);
    return fdn_launch_status();
}

}  // namespace

extern "C" int fdn_ffn_tail(const float* y, const float* dw_w, const float* w, const float* res, float* out, float* stats_out,
                            int B, int C, int N, int H, int W, fdn_stream_t stream) {
    FDN_CHECK_ARG(y && dw_w && w && out && B > 0 && C > 0 && N > 0 && H > 0 && W > 0);
    if (N > 128) return FDN_ERR_UNSUPPORTED;
    if ((unsigned long long)(C + 2) * 4ull * H * W >= 0x80000000ull || (unsigned long long)(N + 40) * 4ull * H * W >= 0x80000000ull)
        return FDN_ERR_UNSUPPORTED;                              // 32-bit buffer offsets with the 2 GiB out-of-image marker
    FtArgs a;
    a.y = y; a.wdw = dw_w; a.w = w; a.res = res; a.out = out; a.stats_out = stats_out;
    a.B = B; a.C = C; a.N = N; a.H = H; a.W = W;
    a.tiles_x = a.tiles_per_img = a.total_tiles = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int mt = (N + 31) / 32;
    if (mt == 1) return launch<1>(a, s);
    if (mt == 2) return launch<2>(a, s);
    if (mt == 3) return launch<3>(a, s);
    return launch<4>(a, s);
}
