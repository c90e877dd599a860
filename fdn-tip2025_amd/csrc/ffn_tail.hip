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

// ------------------------------------------------------------------------------------------------
// Sliding-window form (round 2).  Same fusion, different decomposition: the workgroup walks the hidden channels in PAIRS
// (j0, j1 = 2m, 2m + 1: exactly one k-step of v_mfma_f32_32x32x2_f32) for a fixed 2R x 64 pixel tile.
//   * a wave owns R rows x 32 columns; lane = (column n, k parity kh) evaluates channel 2m + kh at its column for the R rows
//     with a sliding 3 x 3 window per source plane (3 LDS reads per row and plane, as fdn_dwconv_gate), GELU and the gate
//     product - and that value IS the MFMA B operand of its row (lane = pixel, lane half = k), no transpose, no LDS hop;
//   * the R x MT accumulators (16 registers each) stay in registers over the whole channel loop; the epilogue adds the
//     residual, stores 128-byte row segments and emits the next LayerNorm's statistics;
//   * the source planes of pair m + 1 (2, or 3 when C is odd) travel global -> registers (16-byte lanes) during pair m and are
//     parked in the other half of a double-buffered LDS tile: one barrier per pair.
// Per pixel and pair: 2 x (9 + 9) FMAs + 2 GELUs on the vector ALU and MT MFMAs - 0.9 ms of SIMD time for the level-1 FDFFN
// tail against 0.9 ms of HBM time for its 150 planes; the gate tensor (C planes written + read: 5 GB at level 1) is gone.
// ------------------------------------------------------------------------------------------------
constexpr int SW_TC = 64;                 // tile columns (two 32-column wave strips)
constexpr int SW_LS = 72;                 // LDS row stride of a halo plane, in (A, B) cells: left halo at 3, interior from 4 (16-byte lanes), right halo at 68
constexpr unsigned SW_OOB = 0x80000000u;

// (round 4) The two source planes of a pair live in LDS as ONE plane of (A, B) cells and the taps as (wA, wB) pairs: a window cell is
// one 8-byte read and both stencils advance in one v_pk_fma_f32 - 36 packed FMAs per pair and lane where there were 72 scalar ones.
// CODD (odd C: the two channels of a pair read different B planes, FDN_lolv1's 129) keeps a second plane of (A, B1) cells for the odd lanes.
// epilogue of the sliding-window forms: residual, 128-byte row segments, the next LayerNorm's statistics
template <int MT, int R>
__device__ __forceinline__ void ffn_tail_epilogue(const FtArgs& a, f32x16 (&acc)[R][MT], int b, int ty0, int tx0, int r0, int col, int kh) {
    const int N = a.N, H = a.H, W = a.W;
    const unsigned P = (unsigned)H * W, hw4 = P * 4u;
    const int gx = tx0 + col;
    const unsigned nb4 = (unsigned)N * hw4;
    const rsrc_t ro = mk_rsrc(a.out + (long)b * N * P, nb4);
    const rsrc_t rr = mk_rsrc(a.res ? a.res + (long)b * N * P : a.out, a.res ? nb4 : 0u);
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int gy = ty0 + r0 + i;
        const bool ok = gy < H && gx < W;
        const unsigned pix = ok ? (unsigned)(gy * W + gx) : 0u;
        const unsigned vo = ok ? (4u * kh * P + pix) * 4u : SW_OOB;
        float rres[MT][16];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) rres[t][q] = bload(rr, vo, (unsigned)(t * 32 + (q & 3) + 8 * (q >> 2)) * hw4);     // 0 without a residual
        float sm = 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int nrow = t * 32 + (q & 3) + 8 * (q >> 2);
                float v = acc[i][t][q] + rres[t][q];
                bstore(v, ro, vo, (unsigned)nrow * hw4);                              // rows >= N fall outside the descriptor
                v = (nrow + 4 * kh < N) ? v : 0.f;
                acc[i][t][q] = v;
                sm += v;
            }
        if (a.stats_out) {
            sm += __shfl_xor(sm, 32);
            const float mean = sm / (float)N;
            float sq = 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float dl = acc[i][t][q] - mean;
                    sq += (t * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh < N) ? dl * dl : 0.f;
                }
            sq += __shfl_xor(sq, 32);
            if (kh == 0 && ok) {
                float* sp = a.stats_out + (long)b * 2 * P;
                sp[pix] = mean;
                sp[P + pix] = 1.0f / sqrtf(sq / (float)N + 1e-5f);
            }
        }
    }
}

// (round 6) tools/tail_trace.py: -DFDN_TAILSW_TRACE - every wave of 512 workgroups from the middle of the grid sums the s_memtime clocks it spends in the
// phases of a pair step (request + taps, window / stencil / GELU / MFMA issue, parking the next pair, barrier) and stamps prologue and epilogue;
// -DFDN_KOT_GELU / _MFMA / _LOADS / _STENCIL knock a phase out (results are wrong: timing only)
#ifdef FDN_TAILSW_TRACE
constexpr int TT_NWG = 512;
__device__ unsigned long long g_tail_trace[TT_NWG * 4 * 16];
#define TTR(var) const unsigned long long var = __builtin_amdgcn_s_memtime();
#else
#define TTR(var)
#endif
template <int MT, int R, bool IBF, bool CODD>
__global__ __launch_bounds__(256, (R * MT <= 8 && !(IBF && CODD && MT == 2)) ? 2 : 1) void ffn_tail_sw_kernel(FtArgs a) {      // (bf16 input, odd C, N = 64 - FDN_lolv1's level 2 in bf16 mode - needs > 256 registers: one workgroup per CU rather than 227 spilled)
#ifdef FDN_TAILSW_TRACE
    const unsigned long long tt_entry = __builtin_amdgcn_s_memtime();
    unsigned long long tt_sum[4] = {0, 0, 0, 0};
#endif
    constexpr int HRW = 2 * R + 2;                       // halo rows of the tile
    constexpr int PLN = HRW * SW_LS + 4;                 // cells per plane (+ spare cells for the threads without a lane / an edge cell)
    constexpr int NV4 = HRW * 16;                        // float4 of the 64 interior columns
    constexpr int V4T = (NV4 + 255) / 256;               // per thread
    constexpr int NPL = CODD ? 2 : 1;
    constexpr unsigned IES = st_bytes<IBF>();
    __shared__ __attribute__((aligned(16))) fdn_f32x2 planes[2][NPL][PLN];
    __shared__ __attribute__((aligned(16))) fdn_f32x2 dwl[2][2][10];      // [buffer][k parity][tap] = (wA, wB)
    __shared__ float wl[2][2][MT * 32];                                     // [buffer][k parity][output channel]: the projection's column pair w[n][2m], w[n][2m + 1]

    const int C = a.C, N = a.N, H = a.H, W = a.W;
    const unsigned P = (unsigned)H * W, hwi = P * IES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;
    const int wy = wave >> 1, wx = wave & 1;
    const int item = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = item / a.tiles_per_img, t_ = item - b * a.tiles_per_img;
    const int ty0 = (t_ / a.tiles_x) * (2 * R), tx0 = (t_ % a.tiles_x) * SW_TC;
    const rsrc_t rin = mk_rsrc(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.y) + (long)b * C * P * IES), (unsigned)C * hwi);
    constexpr bool codd = CODD;
    const int npairs = (C + 1) / 2;

    // ---- loaders: 16-byte lanes for the interior, dwords for the two edge columns (W % 4 == 0: a float4 is inside or outside) ----
    unsigned g4[V4T], ge;
    int s4[V4T], se;
#pragma unroll
    for (int i = 0; i < V4T; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx >> 4, c4 = idx & 15;
        const int y = ty0 - 1 + r, xx = tx0 + 4 * c4;
        const bool ok = idx < NV4 && y >= 0 && y < H && xx < W;
        g4[i] = ok ? (unsigned)(y * W + xx) * IES : SW_OOB;
        s4[i] = idx < NV4 ? r * SW_LS + 4 + 4 * c4 : HRW * SW_LS;
    }
    {
        const int er = tid >> 1, ec = (tid & 1) ? SW_TC + 1 : 0;      // image column tx0 - 1 + ec lives in cell 3 + ec
        const int ey = ty0 - 1 + er, ex = tx0 - 1 + ec;
        const bool eok = tid < 2 * HRW && ey >= 0 && ey < H && ex >= 0 && ex < W;
        ge = eok ? (unsigned)(ey * W + ex) * IES : SW_OOB;
        se = tid < 2 * HRW ? er * SW_LS + 3 + ec : HRW * SW_LS;
    }
    // two register stages: the planes of pair m + 2 are requested while pair m is evaluated (one pair of arithmetic is ~0.4 us,
    // an HBM round trip under load several times that), parked in LDS a pair later
    struct Stage { float q4[3][V4T][4], qe[3], pdw, pw; };
    Stage stg[2];
    // (round 6) EVERY pair step issues the same loads, unconditionally and without a branch around any of them.  As first written - the request skipped
    // when pair m + 2 does not exist, the taps and the projection column behind `if (tid < 40)` / `j < C` branches with a zero written first - the
    // number of loads in flight depended on the path, the compiler's wait-count pass merged the paths to "nothing younger in flight", and every pair
    // step began with s_waitcnt vmcnt(4) (all of pair m + 1's loads, behind a WAW on the zero) and parked behind vmcnt(0) (pair m + 2's too): the
    // two-stage prefetch was one of less than a step, and the kernel ran 29 % faster without its loads (tools/tail_trace.py, profiles/r06_tail_mid_trace.txt).
    // Now a pair that does not exist reads through a descriptor of zero records (returns 0, moves nothing), a lane without a tap / a column reads
    // outside the descriptor, and the waits count exactly.
    const rsrc_t rdead = mk_rsrc(a.y, 0u);
    const rsrc_t rdw = mk_rsrc(a.wdw, (unsigned)(2 * C * 9) * 4u), rwp = mk_rsrc(a.w, (unsigned)(N * C) * 4u);
    // two small values per thread and pair travel with the planes and are parked in LDS with them: threads 0-39 a depthwise tap, threads 64 .. a
    // projection weight w[n][2 m + par] (a lane reads its own column from LDS at the top of the step - as a per-lane global load carried to the next
    // trip in a register it cost a register copy behind vmcnt(0) at the loop's back edge, i.e. a full drain of the prefetch every second step)
    unsigned vdw, vpw;
    int dw_par, pw_par;
    {
        const int par = tid / 20, i = tid - par * 20, tap = i >> 1;
        dw_par = par;
        vdw = (tid < 40 && tap < 9) ? (unsigned)((((i & 1) ? C : 0) + par) * 9 + tap) * 4u : SW_OOB;      // + 72 m: taps of channel 2 m + par, (wA, wB) interleaved
        const int u = tid - 64, n = u % (MT * 32);
        pw_par = u / (MT * 32);
        vpw = (u >= 0 && u < 2 * MT * 32 && n < N) ? (unsigned)(n * C + pw_par) * 4u : SW_OOB;            // + 8 m: w[n][2 m + par] (threads 64 ..)
    }
    auto fetch = [&](int m, Stage& st) {
        float (&q4)[3][V4T][4] = st.q4;
        float (&qe)[3] = st.qe;
        float& pdw = st.pdw;
#ifdef FDN_KOT_LOADS
        const bool live = m < 2;
#else
        const bool live = m < npairs;                                              // uniform
#endif
        const rsrc_t ri = live ? rin : rdead, rd = live ? rdw : rdead, rw_ = live ? rwp : rdead;
        const int j0 = 2 * m, j1 = (2 * m + 1 < C) ? 2 * m + 1 : 2 * m;
        const int pa = m, pb0 = (C + j0) >> 1, pb1 = (C + j1) >> 1;            // grouped conv: output o reads input o / 2
#pragma unroll
        for (int i = 0; i < V4T; ++i) {
            st_load4<IBF>(q4[0][i], ri, g4[i], (unsigned)pa * hwi);
            st_load4<IBF>(q4[1][i], ri, g4[i], (unsigned)pb0 * hwi);
            if (codd) st_load4<IBF>(q4[2][i], ri, g4[i], (unsigned)pb1 * hwi);
        }
        qe[0] = st_load1<IBF>(ri, ge, (unsigned)pa * hwi);
        qe[1] = st_load1<IBF>(ri, ge, (unsigned)pb0 * hwi);
        if (codd) qe[2] = st_load1<IBF>(ri, ge, (unsigned)pb1 * hwi);
        // depthwise taps of channels 2m, 2m + 1 ([parity][tap](wA, wB), threads 0-39) / the projection's columns 2m, 2m + 1 (waves 1 .. MT)
        pdw = bload(rd, (!codd || 2 * m + dw_par < C) ? vdw : SW_OOB, (unsigned)(72 * m));
        st.pw = bload(rw_, (!codd || 2 * m + pw_par < C) ? vpw : SW_OOB, (unsigned)(8 * m));
    };
    auto stash = [&](int buf, const Stage& st) {
        const float (&q4)[3][V4T][4] = st.q4;
        const float (&qe)[3] = st.qe;
        const float pdw = st.pdw;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int i = 0; i < V4T; ++i)
#pragma unroll
                for (int e = 0; e < 4; e += 2)               // 16-byte lanes: cells (A, B)(A, B)
                    *reinterpret_cast<float4*>(&planes[buf][pl][s4[i] + e]) = float4{q4[0][i][e], q4[1 + pl][i][e], q4[0][i][e + 1], q4[1 + pl][i][e + 1]};
            planes[buf][pl][se] = fdn_f32x2{qe[0], qe[1 + pl]};
        }
        if (tid < 40) reinterpret_cast<float*>(dwl[buf][tid / 20])[tid % 20] = pdw;
        if (tid >= 64 && tid < 64 + 2 * MT * 32) (&wl[buf][0][0])[tid - 64] = st.pw;
    };

    f32x16 acc[R][MT];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][t][q] = 0.f;

    fetch(0, stg[0]);
    fetch(1, stg[1]);
    stash(0, stg[0]);
    __syncthreads();

    const int r0 = wy * R, col = wx * 32 + ln;
    auto pair_step = [&](int m, Stage& mine, const Stage& nxt) {
        // `mine` held pair m (already parked in LDS, its registers are free): refill it with pair m + 2; `nxt` holds pair m + 1
        const int buf = m & 1;
        TTR(tt0)
        fetch(m + 2, mine);
        // taps and projection column of this lane's channel
        fdn_f32x2 wab[9];
        float aw[MT];
        {
            const fdn_f32x2* dp = dwl[buf][kh];
#pragma unroll
            for (int i = 0; i < 9; ++i) wab[i] = dp[i];
#pragma unroll
            for (int t = 0; t < MT; ++t) aw[t] = wl[buf][kh][t * 32 + ln];
        }
        const fdn_f32x2* pAB = planes[buf][(codd && kh) ? NPL - 1 : 0] + r0 * SW_LS + 3 + col;
        // window rows live in a ring of four: row i + 3 is requested while row i (rows i .. i + 2) is evaluated, so the LDS
        // latency hides behind a row's arithmetic
        fdn_f32x2 wAB[4][3];
        auto load_row = [&](int hr, int k) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) wAB[k][dx] = pAB[hr * SW_LS + dx];
        };
        load_row(0, 0);
        load_row(1, 1);
        load_row(2, 2);
        TTR(tt1)
#ifndef FDN_GELU_SCALAR
        static_assert(R % 2 == 0, "rows are gated in pairs");
#pragma unroll
        for (int i = 0; i < R; i += 2) {                                            // two rows per trip: their GELU runs in packed fp32
            fdn_f32x2 sAB[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int ii = i + h2;
                if (ii + 3 < R + 2) load_row(ii + 3, (ii + 3) & 3);
                sAB[h2] = fdn_f32x2{0.f, 0.f};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
#ifdef FDN_KOT_STENCIL
                        if (dy != 1 || dx != 1) continue;
#endif
                        sAB[h2] = __builtin_elementwise_fma(wab[dy * 3 + dx], wAB[(ii + dy) & 3][dx], sAB[h2]);
                    }
            }
#ifdef FDN_KOT_GELU
            const fdn_f32x2 val = fdn_f32x2{sAB[0].x, sAB[1].x} * fdn_f32x2{sAB[0].y, sAB[1].y};
#else
            const fdn_f32x2 val = gelu_fast2(fdn_f32x2{sAB[0].x, sAB[1].x}) * fdn_f32x2{sAB[0].y, sAB[1].y};      // gelu(x1) * x2, FDN_arch.py:473 / :427
#endif
#ifdef FDN_KOT_MFMA
            acc[i][0][0] += val.x * aw[0];
            acc[i + 1][0][0] += val.y * aw[0];
#else
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[t], val.x, acc[i][t], 0, 0, 0);
                acc[i + 1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[t], val.y, acc[i + 1][t], 0, 0, 0);
            }
#endif
        }
        TTR(tt2)
#else
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if (i + 3 < R + 2) load_row(i + 3, (i + 3) & 3);
            fdn_f32x2 sAB = {0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) sAB = __builtin_elementwise_fma(wab[dy * 3 + dx], wAB[(i + dy) & 3][dx], sAB);
            const float val = gelu_fast(sAB.x) * sAB.y;                                   // gelu(x1) * x2, FDN_arch.py:473 / :427
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[t], val, acc[i][t], 0, 0, 0);
        }
#endif
        stash(buf ^ 1, nxt);                                 // (that half was last read two pairs ago, behind the previous barrier; behind the last pair: zeros)
        TTR(tt3)
        __syncthreads();
#ifdef FDN_TAILSW_TRACE
        const unsigned long long tt4 = __builtin_amdgcn_s_memtime();
        tt_sum[0] += tt1 - tt0; tt_sum[1] += tt2 - tt1; tt_sum[2] += tt3 - tt2; tt_sum[3] += tt4 - tt3;
#endif
    };
#ifdef FDN_TAILSW_TRACE
    const unsigned long long tt_loop = __builtin_amdgcn_s_memtime();
#endif
    {
        int m = 0;
        for (; m + 1 < npairs; m += 2) {                      // two steps per trip, no branch between them: the loads in flight are the same on every path
            pair_step(m, stg[0], stg[1]);
            pair_step(m + 1, stg[1], stg[0]);
        }
        if (m < npairs) pair_step(m, stg[0], stg[1]);
    }
#ifdef FDN_TAILSW_TRACE
    const unsigned long long tt_epi = __builtin_amdgcn_s_memtime();
#endif

    ffn_tail_epilogue<MT, R>(a, acc, b, ty0, tx0, r0, col, kh);
#ifdef FDN_TAILSW_TRACE
    {
        const unsigned rel = blockIdx.x - gridDim.x / 2;
        if (rel < (unsigned)TT_NWG && lane == 0) {
            unsigned long long* t = g_tail_trace + ((long)rel * 4 + wave) * 16;
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            t[0] = tt_loop - tt_entry; t[1] = tt_sum[0]; t[2] = tt_sum[1]; t[3] = tt_sum[2]; t[4] = tt_sum[3];
            t[5] = __builtin_amdgcn_s_memtime() - tt_epi; t[6] = tt_epi - tt_loop; t[7] = hwid; t[8] = (unsigned long long)npairs;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// (round 6) The sliding-window form with the source planes travelling global -> LDS DIRECTLY (buffer_load_dword ... lds), three pair steps ahead.
// In the form above a plane crosses the registers: requested two steps ahead, it must have LANDED by the end of the next step to be parked in LDS - the
// trace (tools/tail_trace.py) shows a quarter of a step spent in that park, most of it waiting - and a third register stage costs the third wave per SIMD.
// Here a step issues, per wave, the rows of pair m + 3 as 64-lane dword loads whose destination is the row record of LDS buffer (m + 3) % 4: no stage
// registers, no ds_write, and three steps for the data to arrive.  A row record is [A: 72 floats][B: 72 floats] (odd C: a second B plane), so a window cell
// is still ONE ds_read2_b32 (offsets 0, 72) and the packed stencil is unchanged.  Only the two edge columns (2 x 10 values per plane) keep a register stage:
// they sit in the record's padding, where a 64-lane load cannot put them.  Taps and projection column land in LDS the same way (one load per wave).
// Every wave issues the same number of vector-memory instructions per step (a row / a small block that does not exist goes through a descriptor of zero
// records into a dummy cell), so "pair m + 1 has landed" is s_waitcnt vmcnt(2 x that number) in front of the step's barrier - exact, and by hand: the
// compiler cannot see that another wave will read what this one's loads wrote.  Arithmetic and its order are the form above's: bit-identical.  fp32 input only.
// ------------------------------------------------------------------------------------------------
// Measured (tools/ab_libs.py, interleaved, against the register-staged form above, profiles/r06_tail_dma_ab.txt): 86 -> 32 1.352 against 1.351 ms, 172 -> 64 @L2
// 0.858 against 0.879 ms; step level 278.0 against 280.3 ms (two alternating pairs on one box, within the noise).  Three steps of slack bought nothing
// because nothing was waiting any more: after the load fix the kernel is bound by the ~170 instructions of a pair step (at three waves per SIMD the "park" share
// of the trace is issue time shared with the other waves, not a wait).  NOT the default: -DFDN_TAIL_DMA=1 selects it for fp32 inputs.
#ifndef FDN_TAIL_DMA
#define FDN_TAIL_DMA 0
#endif
template <int MT, int R, bool CODD>
__global__ __launch_bounds__(256, (R * MT <= 8) ? 2 : 1) void ffn_tail_dma_kernel(FtArgs a) {
    constexpr int HRW = 2 * R + 2;                       // halo rows of the tile
    constexpr int NP = CODD ? 3 : 2;                     // planes per row record: A, B (, B of the odd channel)
    constexpr int RS = NP * 72;                          // floats per row record: interior at 4 .. 67, left halo at 3, right halo at 68
    constexpr int PB = (HRW + 2) * RS;                   // (+ two rows nobody reads: waves 2, 3 have a third row slot (10, 11) that takes their OUT-OF-RANGE load -
                                                         //  every wave then issues the same loads with no branch or select around any of them)
    constexpr int SM = 4 * 64;                           // taps (wave 0: 40 used) | projection columns [k parity][output channel] (waves 1 .. MT) | unused
    constexpr int CNT = 3 * NP + 1;                      // direct-to-LDS loads per wave and step
    __shared__ __attribute__((aligned(16))) float pl0[PB], pl1[PB], pl2[PB], pl3[PB];
    __shared__ __attribute__((aligned(16))) float sm0[SM], sm1[SM], sm2[SM], sm3[SM];

    const int C = a.C, N = a.N, H = a.H, W = a.W;
    const unsigned P = (unsigned)H * W, hw4 = P * 4u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int wy = wave >> 1, wx = wave & 1;
    const int item = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = item / a.tiles_per_img, t_ = item - b * a.tiles_per_img;
    const int ty0 = (t_ / a.tiles_x) * (2 * R), tx0 = (t_ % a.tiles_x) * SW_TC;
    const rsrc_t rin = mk_rsrc(a.y + (long)b * C * P, (unsigned)C * hw4);
    const rsrc_t rdead = mk_rsrc(a.y, 0u);
    const rsrc_t rdw = mk_rsrc(a.wdw, (unsigned)(2 * C * 9) * 4u), rwp = mk_rsrc(a.w, (unsigned)(N * C) * 4u);
    const int npairs = (C + 1) / 2;

    // this wave's rows of a record (wave, wave + 4, wave + 8), lane = column
    unsigned vrow[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int rr = wave + 4 * j, y = ty0 - 1 + rr, x = tx0 + lane;
        vrow[j] = (rr < HRW && y >= 0 && y < H && x < W) ? (unsigned)(y * W + x) * 4u : SW_OOB;
    }
    // the two edge columns: threads 0 .. 2 HRW - 1 (image column tx0 - 1 -> cell 3, tx0 + 64 -> cell 68)
    unsigned ge;
    int se;
    {
        const int er = tid >> 1, side = tid & 1;
        const int ey = ty0 - 1 + er, ex = side ? tx0 + SW_TC : tx0 - 1;
        const bool eok = tid < 2 * HRW && ey >= 0 && ey < H && ex >= 0 && ex < W;
        ge = eok ? (unsigned)(ey * W + ex) * 4u : SW_OOB;
        se = tid < 2 * HRW ? er * RS + (side ? 68 : 3) : -1;
    }
    // the small block of this wave: wave 0 the 40 taps, waves 1 .. MT the projection's column pair
    unsigned vsm;
    int sm_par;
    {
        const int par = tid / 20, i = tid - par * 20, tap = i >> 1;
        sm_par = par;
        vsm = (tid < 40 && tap < 9) ? (unsigned)((((i & 1) ? C : 0) + par) * 9 + tap) * 4u : SW_OOB;       // + 72 m
        if (wv >= 1 && wv <= MT) {
            const int u = tid - 64, n = u % (MT * 32);
            sm_par = u / (MT * 32);
            vsm = n < N ? (unsigned)(n * C + sm_par) * 4u : SW_OOB;                                        // + 8 m
        }
    }
    typedef __attribute__((address_space(3))) void* lds_vp;
    float qe[2][NP];                                     // edge values: two register stages
    auto fetch_edges = [&](int m, float (&q)[NP]) {
        const bool live = m < npairs;
        const rsrc_t ri = live ? rin : rdead;
        const int j0 = 2 * m, j1 = (2 * m + 1 < C) ? 2 * m + 1 : 2 * m;
        q[0] = bload(ri, ge, (unsigned)m * hw4);
        q[1] = bload(ri, ge, (unsigned)((C + j0) >> 1) * hw4);
        if (CODD) q[NP - 1] = bload(ri, ge, (unsigned)((C + j1) >> 1) * hw4);
    };
    auto park_edges = [&](float* pl, const float (&q)[NP]) {
        if (se >= 0) {
#pragma unroll
            for (int p_ = 0; p_ < NP; ++p_) pl[se + 72 * p_] = q[p_];
        }
    };
    auto dma = [&](int m, float* pl, float* smb) {       // CNT loads per wave, whatever m and the wave
        const bool live = m < npairs;
        const rsrc_t ri = live ? rin : rdead;
        const int j0 = 2 * m, j1 = (2 * m + 1 < C) ? 2 * m + 1 : 2 * m;
        const unsigned so[3] = {(unsigned)m * hw4, (unsigned)((C + j0) >> 1) * hw4, (unsigned)((C + j1) >> 1) * hw4};
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int p_ = 0; p_ < NP; ++p_)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ri, (lds_vp)(pl + (wv + 4 * j) * RS + 72 * p_ + 4), 4, vrow[j],
                                                         so[p_ == 0 ? 0 : (CODD && p_ == 2) ? 2 : 1], 0, 0);
        const rsrc_t rs = !live ? rdead : (wv == 0 ? rdw : rwp);
        const unsigned v = (!CODD || 2 * m + sm_par < C) ? vsm : SW_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vp)(smb + 64 * wv), 4, v, (unsigned)((wv == 0 ? 72 : 8) * m), 0, 0);
    };

    f32x16 acc[R][MT];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][t][q] = 0.f;

    // prologue: pair 0 complete in buffer 0, pair 1 on its way into buffer 1 with its edges in stage 1, pair 2 on its way into buffer 2
    {
        float q0[NP];
        fetch_edges(0, q0);
        dma(0, pl0, sm0);
        fetch_edges(1, qe[1]);
        dma(1, pl1, sm1);
        dma(2, pl2, sm2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        park_edges(pl0, q0);
    }
    __syncthreads();

    const int r0 = wy * R, col = wx * 32 + ln;
    // step m: compute pair m from `cur`; edges of pair m + 2 -> stage m & 1; rows of pair m + 3 -> `far`; edges of pair m + 1 (stage (m + 1) & 1) -> `nxt`
    auto pair_step = [&](int m, const float* cur, const float* smc, float* nxt, float* far, float* smf, float (&qnew)[NP], const float (&qold)[NP]) {
        fetch_edges(m + 2, qnew);                        // (first: waiting for them later must not wait for this step's rows)
        dma(m + 3, far, smf);
        fdn_f32x2 wab[9];
        float aw[MT];
        {
            const fdn_f32x2* dp = reinterpret_cast<const fdn_f32x2*>(smc) + kh * 10;
#pragma unroll
            for (int i = 0; i < 9; ++i) wab[i] = dp[i];
#pragma unroll
            for (int t = 0; t < MT; ++t) aw[t] = smc[64 + kh * (MT * 32) + t * 32 + ln];
        }
        const float* pA = cur + r0 * RS + 3 + col;
        const float* pB = pA + ((CODD && kh) ? 144 : 72);
        fdn_f32x2 wAB[4][3];
        auto load_row = [&](int hr, int k) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) wAB[k][dx] = fdn_f32x2{pA[hr * RS + dx], pB[hr * RS + dx]};
        };
        load_row(0, 0);
        load_row(1, 1);
        load_row(2, 2);
        static_assert(R % 2 == 0, "rows are gated in pairs");
#pragma unroll
        for (int i = 0; i < R; i += 2) {
            fdn_f32x2 sAB[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int ii = i + h2;
                if (ii + 3 < R + 2) load_row(ii + 3, (ii + 3) & 3);
                sAB[h2] = fdn_f32x2{0.f, 0.f};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) sAB[h2] = __builtin_elementwise_fma(wab[dy * 3 + dx], wAB[(ii + dy) & 3][dx], sAB[h2]);
            }
            const fdn_f32x2 val = gelu_fast2(fdn_f32x2{sAB[0].x, sAB[1].x}) * fdn_f32x2{sAB[0].y, sAB[1].y};      // gelu(x1) * x2, FDN_arch.py:473 / :427
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[t], val.x, acc[i][t], 0, 0, 0);
                acc[i + 1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[t], val.y, acc[i + 1][t], 0, 0, 0);
            }
        }
        park_edges(nxt, qold);                           // pair m + 1's edge columns (requested a step ago, in front of that step's rows)
        // pair m + 1's rows (requested two steps ago) have landed once at most this step's and the previous step's loads are outstanding
        // (a bare barrier: __syncthreads() is a workgroup-scope release, and with direct-to-LDS loads in flight the compiler implements that as vmcnt(0))
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (CNT + NP)) : "memory");
    };
    {
        int m = 0;
        for (; m + 3 < npairs; m += 4) {                 // four steps per trip: the buffer of every step is a compile-time array
            pair_step(m, pl0, sm0, pl1, pl3, sm3, qe[0], qe[1]);
            pair_step(m + 1, pl1, sm1, pl2, pl0, sm0, qe[1], qe[0]);
            pair_step(m + 2, pl2, sm2, pl3, pl1, sm1, qe[0], qe[1]);
            pair_step(m + 3, pl3, sm3, pl0, pl2, sm2, qe[1], qe[0]);
        }
        if (m < npairs) pair_step(m, pl0, sm0, pl1, pl3, sm3, qe[0], qe[1]);
        if (m + 1 < npairs) pair_step(m + 1, pl1, sm1, pl2, pl0, sm0, qe[1], qe[0]);
        if (m + 2 < npairs) pair_step(m + 2, pl2, sm2, pl3, pl1, sm1, qe[0], qe[1]);
    }
    ffn_tail_epilogue<MT, R>(a, acc, b, ty0, tx0, r0, col, kh);
}

template <int MT, int R, bool IBF>
int launch_sw(FtArgs a, hipStream_t s) {
    a.tiles_x = cdiv(a.W, SW_TC);
    a.tiles_per_img = a.tiles_x * cdiv(a.H, 2 * R);
    a.total_tiles = a.B * a.tiles_per_img;
    if constexpr (FDN_TAIL_DMA && !IBF) {
        if (a.C & 1) hipLaunchKernelGGL((ffn_tail_dma_kernel<MT, R, true>), dim3((unsigned)a.total_tiles), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((ffn_tail_dma_kernel<MT, R, false>), dim3((unsigned)a.total_tiles), dim3(256), 0, s, a);
        return fdn_launch_status();
    }
    if (a.C & 1) hipLaunchKernelGGL((ffn_tail_sw_kernel<MT, R, IBF, true>), dim3((unsigned)a.total_tiles), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((ffn_tail_sw_kernel<MT, R, IBF, false>), dim3((unsigned)a.total_tiles), dim3(256), 0, s, a);
    return fdn_launch_status();
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

#ifdef FDN_TAILSW_TRACE
extern "C" int fdn_debug_tail_trace(void* host, long bytes, int clear) {          // trace builds only (tools/tail_trace.py); not part of the ABI
    static unsigned long long z[TT_NWG * 4 * 16];
    if (bytes > (long)sizeof(z)) return FDN_ERR_ARG;
    if (clear) return hipMemcpyToSymbol(HIP_SYMBOL(g_tail_trace), z, sizeof(z), 0, hipMemcpyHostToDevice) == hipSuccess ? FDN_OK : FDN_ERR_LAUNCH;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tail_trace), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? FDN_OK : FDN_ERR_LAUNCH;
}
#endif

extern "C" int fdn_ffn_tail(const void* y_, const float* dw_w, const float* w, const float* res, float* out, float* stats_out,
                            int B, int C, int N, int H, int W, int y_bf16, int form, fdn_stream_t stream) {
    const float* y = static_cast<const float*>(y_);
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
    if (form == 1) {                                             // sliding-window form: N <= 64, 16-byte input lanes
        if (W % 4 != 0 || (reinterpret_cast<uintptr_t>(y) & 15) != 0 || mt > 2) return FDN_ERR_UNSUPPORTED;
        if (mt == 1) return y_bf16 ? launch_sw<1, 4, true>(a, s) : launch_sw<1, 4, false>(a, s);
        // (64-wide outputs: 8 x 64 tiles as well - half the barriers per pixel and 10 / 8 instead of 6 / 4 halo rows beat the third
        //  workgroup per CU of the 4 x 64 tile: 0.636 -> 0.586 ms for the level-2 FCAFFN tail)
        return y_bf16 ? launch_sw<2, 4, true>(a, s) : launch_sw<2, 4, false>(a, s);
    }
    if (y_bf16) return FDN_ERR_UNSUPPORTED;
    if (mt == 1) return launch<1>(a, s);
    if (mt == 2) return launch<2>(a, s);
    if (mt == 3) return launch<3>(a, s);
    return launch<4>(a, s);
}
