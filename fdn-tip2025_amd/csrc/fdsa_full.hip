// The WHOLE FDSA sub-block in one launch (FDN_arch.py:575-639 + the residual of :671): channel LayerNorm of x, to_hidden (1x1), the
// depthwise 3x3, the three 8x8-patch rfft2, the amplitude / phase recombination, the three irfft2, the three channel LayerNorms of
// out1 | out2 | out3, the v_value gate, project_out (1x1, 3E -> C), `x + ...` and the LayerNorm statistics of the result.  Neither the
// 4E-channel hidden tensor nor the 4E-channel (out1|out2|out3|v_value) hand-off of fdn_fdsa_fused -> fdn_fdsa_out exists in HBM:
// the sub-block reads C planes (+ halo, mostly L2 hits) and the residual, and writes C planes: 3C elements per pixel instead of the
// pair's (C + 4E) + (4E + 2C) = 3C + 8E.
//
// The obstacle to this fusion is that project_out needs the LayerNorm of out_g over ALL E channels of a pixel, while the spectral part
// produces the channels chunk by chunk, and the 3E (+ E) values of a tile do not fit in LDS.  It is solved algebraically, with a pivot:
//     y[n] = sum_g rstd_g * ( S1_g[n] - d_g * S0_g[n] ) + Bt[n],
//     S1_g[n] = sum_e A_g[n][e] * vv_e * (o_ge - s_g),   S0_g[n] = sum_e A_g[n][e] * vv_e,   Bt[n] = sum_e (sum_g W[n][gE+e] beta_g[e]) * vv_e,
//     A_g[n][e] = W[n][gE+e] * gamma_g[e],  d_g = mean_e(o_ge) - s_g,  rstd_g = 1 / sqrt(mean_e (o_ge - s_g)^2 - d_g^2 + 1e-5),
// where s_g (per pixel and group) is the mean of o_g over the FIRST chunk's channels.  S1 accumulates chunk by chunk on the bf16 matrix
// pipe (exactly split fp32 operands, common.hpp) while the statistics accumulate beside it; S0 and Bt are products with v_value
// alone, formed once at the end from the v_value rows kept in LDS.  The un-pivoted form (s = 0) cancels catastrophically when a group's
// mean is large against its deviation; with the pivot |d_g| <= sigma_g * sqrt(E / CE) (the mean of CE of the E values cannot be further
// from the mean of all of them), so S1 - d S0 loses at most a factor sqrt(1 + E / CE) <= 2.4 against the direct evaluation, whereas the
// direct fp32 form of the reference subtracts a mean of any size from every value (error ~ eps |mu| / sigma).
//
// Workgroup = one tile of PT 8x8 patches side by side (8 x 16 pixels for C <= 32, 8 x 8 for C <= 64), 256 threads, two per CU.
// Per chunk of CE channels (8 resp. 16):
//   P0  to_hidden on the matrix cores for the chunk's 4 CE rows over the halo tile -> LDS `hid` (as fdsa_fused_kernel)       | barrier
//   P1..P5 are WAVE-LOCAL: a wave owns CE / 4 channels x PT patches = 4 (channel, patch) slots and walks
//       rows (stencil + row rfft of q, k, v; stencil of v_value -> LDS `VV`) -> forward column FFTs (60 lanes) -> recombination
//       (160 bins over 64 lanes) -> inverse column FFTs -> inverse rows -> the chunk's out_g rows, pixel-major, into `T` (which
//       overlays the wave's own spectra), with no workgroup barrier in between                                                   | barrier
//   P6  a wave owns 32 output pixels (x 32 output channels): reads the chunk's out_g / v_value of its pixels, subtracts the pivot,
//       accumulates the statistics, forms (o - s) * vv, cuts it into three bf16 parts and runs the S1 MFMAs.
// Epilogue: S0 / Bt MFMAs from `VV`, the combination above, residual, store, next LayerNorm's statistics.
#include "patch_fft.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_sync_lds() {
    // LDS operations of one wave execute in order; this keeps the COMPILER from moving LDS accesses across the phase boundary
    // (and drains the wave's outstanding LDS operations, which costs nothing measurable)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// K = 8 form of the split product (v_mfma_f32_32x32x8_bf16: lane (n, kh) holds k = 4 kh + j, j < 4): at 8 channels per chunk a lane
// then prepares 4 channels instead of 8 of which half are padding
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma8(fdn_u32x2 a, fdn_u32x2 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma8_split6(const fdn_u32x2 (&a)[3], const fdn_u32x2 (&b)[3], f32x16 c) {
    c = mfma8(a[2], b[0], c);
    c = mfma8(a[1], b[1], c);
    c = mfma8(a[0], b[2], c);
    c = mfma8(a[1], b[0], c);
    c = mfma8(a[0], b[1], c);
    return mfma8(a[0], b[0], c);
}

struct FullArgs {
    const float* x;         // [B][C][H][W]
    long xbs;
    const float* stats;     // [B][2][P] (mean, rstd) of x, or null: no LayerNorm in front
    const fdn_u32x4* wth;   // to_hidden operands (fdn_fdsa_pack layout, 8-channel chunks)
    const float* dww;       // [4E][9]
    const float* fftw;      // [E][40]
    const fdn_u32x4* ws1;   // project_out operands per chunk: [chunk][g][mt][part][lane]
    const fdn_u32x4* wep;   // project_out operands, dense in K: [ks][t = g0,g1,g2,beta][mt][part][lane]
    const float* res;       // [B][C][H][W] or null
    float* out;             // [B][C][H][W]
    float* stats_out;       // [B][2][P] or null
    int E, H, W, tiles_x, tiles_per_img, nchunks, N;
};

template <int C, int PT>
struct FullGeo {
    static constexpr int CHW = 4 / PT;                 // channels per wave and chunk
    static constexpr int CE = 4 * CHW;                 // channels per chunk: 8 (PT = 2) or 16 (PT = 1)
    static constexpr int MTH = CE / 8;                 // 32-row MFMA tiles of to_hidden per chunk
    static constexpr int TWP = 8 * PT;                 // tile width in pixels
    static constexpr int NPX = 8 * TWP;                // pixels per tile: 128 or 64
    static constexpr int HW_ = TWP + 2;                // halo tile 10 x HW_
    static constexpr int HP = 10 * HW_;                // 180 or 100 halo pixels
    static constexpr int NS = (HP + 31) / 32;          // 6 or 4 strips of 32 halo pixels
    static constexpr int NSW = (NS + 3) / 4;           // strips per wave: 2 or 1
    // LDS row stride of a hidden plane and floats per plane: with (19, 208) resp. (11, 120) the 32 lanes (slot, row) of a stencil read
    // fall into 32 distinct banks for every tap (tools/lds_conflicts_fdsa_full.py; 197 / 117 cost 2x / 4x the LDS cycles there)
    static constexpr int FRS = PT == 2 ? 19 : 11;
    static constexpr int FPL = PT == 2 ? 208 : 120;
    static constexpr int KST = (C + 15) / 16;          // k-steps of to_hidden
    static constexpr int KS = KST * 3 + 1;             // operand fragments per chunk of the fdn_fdsa_pack layout (3 parts per k-step + bias)
    static constexpr int EMAX = C * 6 / 5;             // int(1.2 C)
    static constexpr int NCH = (EMAX + CE - 1) / CE;   // chunks
    static constexpr int NMT = (C + 31) / 32;          // 32-row output tiles
    static constexpr int NOS = NPX / 32;               // output strips: 4 or 2  (NOS * NMT == 4: one (strip, tile) job per wave)
    static constexpr int KE = (EMAX + 15) / 16;        // k-steps of the dense epilogue products
    static constexpr int SKS = 16 * PS + 4;            // float2 stride between the q / k / v spectra: 724 = 20 mod 32, so the 60 column-FFT
                                                       // lanes (kind, slot, kx) -> 9 * lane + const keep the spacing of one kind across kinds
    static constexpr int VPS = NPX + 4;                // floats per v_value row
    static constexpr int WGS = 2;                      // workgroups per CU (LDS 68 / 79 KB).  A third one was tried at PT = 2 with the v_value
                                                       // rows in a scratch tensor instead of LDS (50 KB): 168 registers spill 46, 4.87 against 3.96 ms
    static_assert(NOS * NMT == 4, "one output job per wave");
};

template <int C, int PT, bool LN>
#ifndef FDN_FULL_WGS
#define FDN_FULL_WGS 0          // 0: the geometry's choice
#endif
__global__ __launch_bounds__(256, (FDN_FULL_WGS ? FDN_FULL_WGS : FullGeo<C, PT>::WGS)) void fdsa_full_kernel(FullArgs a) {
    typedef FullGeo<C, PT> G;
    constexpr int CHW = G::CHW, CE = G::CE, MTH = G::MTH, TWP = G::TWP, NPX = G::NPX, HW_ = G::HW_, HP = G::HP, NS = G::NS, NSW = G::NSW;
    constexpr int FRS = G::FRS, FPL = G::FPL, KST = G::KST, KS = G::KS, NCH = G::NCH, NMT = G::NMT, NOS = G::NOS, SKS = G::SKS, VPS = G::VPS;

    __shared__ float hid[4 * CE * FPL];
    __shared__ __attribute__((aligned(16))) float2 S[3 * SKS];          // spectra; the chunk's out_g rows (`T`) overlay them
    __shared__ __attribute__((aligned(16))) float VV[NCH * CE * VPS];   // v_value of the whole tile, all channels
    __shared__ float wks[2][4 * CE * 9];                                // depthwise taps of a chunk: [kind * CE + channel][9], by chunk parity
    __shared__ float fgs[2][CE * 40];                                   // fft gains: [channel][ky][kx]
    __shared__ float red[PT == 1 ? 4 * 64 : 1];                         // two-tile statistics exchange (C > 32)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;
    const int t_ = (int)xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = t_ / a.tiles_per_img, ti = t_ - b * a.tiles_per_img;
    const int ty0 = (ti / a.tiles_x) * 8, tx0 = (ti % a.tiles_x) * TWP;
    const int E = a.E, H = a.H, W = a.W, N = a.N;
    const unsigned P = (unsigned)H * W, hw4 = P * 4u;
    const rsrc_t rx = mk_rsrc(a.x + (long)b * a.xbs, (unsigned)C * hw4);
    const rsrc_t rst = mk_rsrc(LN ? a.stats + (long)b * 2 * P : a.x, LN ? 2u * hw4 : 0u);

    // ---- the wave's strips of the normalised halo tile: B operands of to_hidden, resident (as fdsa_fused_kernel) ----------------
    fdn_u32x4 xb[NSW][KST][3];
    unsigned onebits = 0;
    int pixoff[NSW];
#pragma unroll
    for (int si = 0; si < NSW; ++si) {
        const int s = wave + 4 * si;
        const int p = s * 32 + ln;
        const int r = p / HW_, c = p - r * HW_;
        const int gy = ty0 - 1 + r, gx = tx0 - 1 + c;
        const bool in_tile = s < NS && p < HP;
        const bool ok = in_tile && gy >= 0 && gy < H && gx >= 0 && gx < W;
        onebits |= (ok && kh == 0) ? (1u << si) : 0u;
        pixoff[si] = in_tile ? r * FRS + c : 10 * FRS;              // lanes past the tile: a pad cell behind row 9 of each plane
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
            for (int j = 0; j < 8; ++j) xs[ks][j] = bload(rx, g + (unsigned)(8 * kh) * hw4, (unsigned)(16 * ks + j) * hw4);
#pragma unroll
        for (int ks = 0; ks < KST; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v0 = xs[ks][2 * j], v1 = xs[ks][2 * j + 1];
                if (LN) v0 = (v0 - mu) * rs, v1 = (v1 - mu) * rs;
                unsigned p1, p2, p3;
                fdn_split3(v0, v1, p1, p2, p3);
                xb[si][ks][0][j] = p1, xb[si][ks][1][j] = p2, xb[si][ks][2][j] = p3;
            }
    }

    // ---- wave-local coordinates: lane = (half h, slot sl, row); slot = (channel c of the wave, patch p) --------------------------
    const int h = kh, sl = (lane >> 3) & 3, row = lane & 7;
    const int cw = sl / PT, pt = sl % PT;
    const int chl = wave * CHW + cw;                               // channel within the chunk
    const int slotg = wave * 4 + sl;                               // spectrum slot
    float* Sf = reinterpret_cast<float*>(S);
    // T[g][channel of the chunk][pixel] overlays the spectra of kind g of the wave that owns the channel
    auto t_off = [&](int g, int ch_, int px) { return (g * SKS + (ch_ / CHW) * 4 * PS) * 2 + (ch_ % CHW) * NPX + px; };
    // pixel (row, column) -> offset inside a T / VV row: the 16-byte slots of a tile row are XOR-swizzled with the row's 128-byte
    // half-line index, so that the eight rows a 16-byte store group covers land in eight different bank quads (4-way conflicts else)
    auto pxs = [&](int prow, int pcol) { return prow * TWP + ((((pcol >> 2) ^ ((prow * TWP) >> 5)) & (TWP / 4 - 1)) << 2) + (pcol & 3); };

    // output job of this wave: strip of 32 pixels x tile of 32 output channels
    const int ostrip = wave % NOS, omt = wave / NOS;
    const int opx = ostrip * 32 + ln;                              // pixel of the tile this lane owns in P6 / the epilogue
    const int opxs = pxs(opx / TWP, opx % TWP);                    // its (swizzled) offset inside a T / VV row
    f32x16 s1[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) s1[g][r] = 0.f;
    float piv[3] = {0.f, 0.f, 0.f}, sum1[3] = {0.f, 0.f, 0.f}, sum2[3] = {0.f, 0.f, 0.f};

    // per-chunk taps / gains: fetched a chunk ahead into registers, parked in the LDS buffer of the chunk's parity
    constexpr int NTAP = 4 * CE * 9, NGAIN = CE * 40;
    constexpr int TPT = (NTAP + 255) / 256, GPT = (NGAIN + 255) / 256;
    float st_w[TPT], st_f[GPT];
    auto stage_fetch = [&](int ch) {
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
            const int idx = tid + 256 * i;
            const int m = idx / 9, tap = idx - m * 9;                  // row m = kind * CE + channel
            const int ew = ch * CE + (m % CE);
            st_w[i] = idx < NTAP ? a.dww[(long)((m / CE) * E + (ew < E ? ew : E - 1)) * 9 + tap] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int idx = tid + 256 * i;
            const int c0 = idx / 40;
            const int e0_ = ch * CE + c0;
            st_f[i] = idx < NGAIN ? a.fftw[(e0_ < E ? e0_ : E - 1) * 40 + (idx - c0 * 40)] : 0.f;
        }
    };
    auto stage_store = [&](int ch) {
#pragma unroll
        for (int i = 0; i < TPT; ++i)
            if (tid + 256 * i < NTAP) wks[ch & 1][tid + 256 * i] = st_w[i];
#pragma unroll
        for (int i = 0; i < GPT; ++i)
            if (tid + 256 * i < NGAIN) fgs[ch & 1][tid + 256 * i] = st_f[i];
    };
    stage_fetch(0);
    stage_store(0);                      // (visible behind the first barrier)
    // the to_hidden operands of the NEXT chunk are requested behind the inverse rows, so that P0 does not start with an L2 round trip
    // (8-channel chunks: all 7 fragments, 28 registers; at 16-channel chunks the 26 fragments do not fit: read on the fly)
#ifndef FDN_FULL_AWA
#define FDN_FULL_AWA -1
#endif
    constexpr int AWN = FDN_FULL_AWA >= 0 ? FDN_FULL_AWA : (MTH == 1 ? KS : 0);      // fragments held ahead (0: none)
    fdn_u32x4 aw[AWN > 0 ? AWN : 1];
    auto aw_fetch = [&](int ch) {
        const fdn_u32x4* wp = a.wth + ((long)(ch * MTH) * KS) * 64 + lane;
#pragma unroll
        for (int j = 0; j < AWN; ++j) aw[j] = wp[j * 64];
    };
    aw_fetch(0);

    auto dw_row8 = [&](const float* hp, const float* wk9, float (&o8)[8]) __attribute__((always_inline)) {
        float wkt[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) wkt[i] = wk9[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            float v[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) v[j] = hp[dy * FRS + j];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) o8[j] = fmaf(wkt[dy * 3 + dx], v[j + dx], o8[j]);
        }
    };

    for (int ch = 0; ch < a.nchunks; ++ch) {
        const int ce0 = ch * CE;
        const bool more = ch + 1 < a.nchunks;                       // uniform
        if (more) stage_fetch(ch + 1);

        // ---- P0: to_hidden of the chunk on the matrix cores -> hid (0 outside the image: strip, statistics and `xone` read 0 there)
#pragma unroll
        for (int mh = 0; mh < MTH; ++mh) {
            const fdn_u32x4* wp_ = a.wth + ((long)(ch * MTH + mh) * KS) * 64 + lane;
#pragma unroll
            for (int si = 0; si < NSW; ++si) {
                if (wave + 4 * si < NS) {                           // wave-uniform
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks) {
                        auto frag = [&](int j) { return (mh == 0 && j < AWN) ? aw[j < AWN ? j : 0] : wp_[j * 64]; };
                        const fdn_u32x4 a3[3] = {frag(3 * ks), frag(3 * ks + 1), frag(3 * ks + 2)};
                        acc = fdn_mfma_split6(a3, xb[si][ks], acc);
                    }
                    const bool one = (onebits >> si) & 1u;
                    const fdn_u32x4 xone = {one ? 0x3F803F80u : 0u, one ? 0x00003F80u : 0u, 0u, 0u};
                    acc = fdn_mfma_bf16((mh == 0 && KS - 1 < AWN) ? aw[KS - 1 < AWN ? KS - 1 : 0] : wp_[(KS - 1) * 64], xone, acc);          // + bias (0 outside the image)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = (r & 3) + 8 * (r >> 2) + 4 * kh;              // row of the 32-row tile = kind * 8 + channel
                        hid[((m >> 3) * CE + mh * 8 + (m & 7)) * FPL + pixoff[si]] = acc[r];
                    }
                }
            }
        }
        __syncthreads();                                            // B1: hid, taps and gains of this chunk are in LDS

        const float* wk_ = wks[ch & 1];
        const float* fg_ = fgs[ch & 1];
        // ---- P1: rows.  Round A: q | k by lane half; round B: v | v_value -------------------------------------------------------
        {
            float o8[8];
            dw_row8(hid + (h * CE + chl) * FPL + row * FRS + pt * 8, wk_ + (h * CE + chl) * 9, o8);           // to_hidden_dw, FDN_arch.py:578
            float2 sp[5];
            rfft8_row(o8, sp);
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) S[h * SKS + slotg * PS + kx * KXS + row] = sp[kx];
            dw_row8(hid + ((2 + h) * CE + chl) * FPL + row * FRS + pt * 8, wk_ + ((2 + h) * CE + chl) * 9, o8);
            if (h == 0) {
                rfft8_row(o8, sp);
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) S[2 * SKS + slotg * PS + kx * KXS + row] = sp[kx];
            } else {
                float* vp = VV + (ce0 + chl) * VPS;
                *reinterpret_cast<f32x4*>(vp + pxs(row, pt * 8)) = f32x4{o8[0], o8[1], o8[2], o8[3]};
                *reinterpret_cast<f32x4*>(vp + pxs(row, pt * 8 + 4)) = f32x4{o8[4], o8[5], o8[6], o8[7]};
            }
        }
        wave_sync_lds();
        // ---- P2: forward column FFTs, lane = (kind, slot of the wave, kx): 60 of 64 lanes ---------------------------------------
        const int ck = lane / 20, cr = lane - ck * 20;               // kind, 5 * slot + kx
        float2* const colp = S + ck * SKS + wave * 4 * PS + (cr / 5) * PS + (cr % 5) * KXS;
        if (lane < 60) {
            float2 z[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) z[i] = colp[i];
            fft8<false>(z);
#pragma unroll
            for (int i = 0; i < 8; ++i) colp[i] = z[i];
        }
        wave_sync_lds();
        // ---- P3: recombination, 160 bins of the wave's four slots over 64 lanes (FDN_arch.py:591-630) ----------------------------
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int bin = lane + 64 * i;
            if (bin < 160) {
                const int bs = bin / 40, br = bin - bs * 40;         // slot of the wave, kx * 8 + ky
                const int kx = br >> 3, ky = br & 7;
                float2* const sp = S + (wave * 4 + bs) * PS + kx * KXS + ky;
                const float2 q = sp[0], k = sp[SKS], v = sp[2 * SKS];
                const float f = fg_[(wave * CHW + bs / PT) * 40 + ky * 5 + kx];
                const float2 v1 = make_float2(rd1(v.x * f), rd1(v.y * f));                  // :591-593
                float2 qk = cmul(q, k);                                                     // :595
                qk = make_float2(rd1(qk.x), rd1(qk.y));                                     // :597
                const float qk2 = qk.x * qk.x + qk.y * qk.y, v2 = v1.x * v1.x + v1.y * v1.y;
                const float qka = qk2 * rsq(qk2);                                           // |qk|  :599
                const float iv = rsq(v2), va = v2 * iv;                                     // |v|   :601
                const float2 qr = make_float2(rd1(q.x), rd1(q.y));                          // :603
                const float2 kr = make_float2(rd1(k.x), rd1(k.y));                          // :604
                const float nq = rsq(qr.x * qr.x + qr.y * qr.y), nk = rsq(kr.x * kr.x + kr.y * kr.y);
                const float2 u = cmulc(make_float2(qr.x * nq, qr.y * nq), make_float2(kr.x * nk, kr.y * nk));   // :605-607
                const float g_ = qka * iv;
                sp[0] = make_float2(va * u.x, va * u.y);                                    // out1 spectrum :609-612
                sp[SKS] = make_float2(g_ * v1.x, g_ * v1.y);                                // out2 :617-619
                sp[2 * SKS] = make_float2(qka * u.x, qka * u.y);                            // out3 :627-629
            }
        }
        wave_sync_lds();
        // ---- P4: inverse column FFTs ----------------------------------------------------------------------------------------------
        if (lane < 60) {
            float2 z[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) z[i] = colp[i];
            fft8<true>(z);
            constexpr float sc = 1.0f / 64.0f;                       // norm = 'backward'
#pragma unroll
            for (int i = 0; i < 8; ++i) colp[i] = make_float2(z[i].x * sc, z[i].y * sc);
        }
        wave_sync_lds();
        // the S1 operands of this chunk (L2 hits) are requested here: they arrive behind the inverse rows and the barrier
        fdn_u32x4 aop[CE == 8 ? 1 : 3][3];
        fdn_u32x2 aop8[CE == 8 ? 3 : 1][3];
        {
            const fdn_u32x4* wq = a.ws1 + ((long)(ch * 3) * NMT + omt) * 3 * 64 + lane;
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) {
                    if constexpr (CE == 8) aop8[g][pp] = *reinterpret_cast<const fdn_u32x2*>(&wq[((long)g * NMT * 3 + pp) * 64]);
                    else aop[g][pp] = wq[((long)g * NMT * 3 + pp) * 64];
                }
        }
        // ---- P5: inverse rows -> T[g][channel][pixel] (overlaying the wave's own spectra of kind g) ------------------------------
        {
            float2 xk[5];
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) xk[kx] = S[h * SKS + slotg * PS + kx * KXS + row];
            float2 xk2[5];
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) xk2[kx] = S[2 * SKS + slotg * PS + kx * KXS + row];     // (both halves read; the lower one uses it)
            wave_sync_lds();                                         // every read of the spectra precedes the first T write
            float r8[8];
            irfft8_row(xk, r8);
            float* tp = Sf + t_off(h, chl, 0);
            *reinterpret_cast<f32x4*>(tp + pxs(row, pt * 8)) = f32x4{r8[0], r8[1], r8[2], r8[3]};
            *reinterpret_cast<f32x4*>(tp + pxs(row, pt * 8 + 4)) = f32x4{r8[4], r8[5], r8[6], r8[7]};
            if (h == 0) {
                irfft8_row(xk2, r8);
                float* tp2 = Sf + t_off(2, chl, 0);
                *reinterpret_cast<f32x4*>(tp2 + pxs(row, pt * 8)) = f32x4{r8[0], r8[1], r8[2], r8[3]};
                *reinterpret_cast<f32x4*>(tp2 + pxs(row, pt * 8 + 4)) = f32x4{r8[4], r8[5], r8[6], r8[7]};
            }
        }
        if (more) stage_store(ch + 1);                               // (buffer of the other parity: last read before B2 of the chunk before)
        if (AWN > 0 && more) aw_fetch(ch + 1);
        __syncthreads();                                            // B2: T and VV rows of the chunk are complete; hid is free

        // ---- P6: S1 and the statistics for the wave's 32 pixels ------------------------------------------------------------------
        const bool whole = ce0 + CE <= E;                            // uniform: every channel of the chunk exists (all but the last chunk)
        if constexpr (CE == 8) {
            // lane (pixel ln, half kh) prepares channels 4 kh .. 4 kh + 3 of the chunk: the K = 8 MFMA has no padding k slots
            float vvv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) vvv[i] = VV[(ce0 + 4 * kh + i) * VPS + opxs];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = Sf[t_off(g, 4 * kh + i, opxs)];
                if (ch == 0) {                                       // the pivot: mean of the first chunk's channels at this pixel
                    float part = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) part += (4 * kh + i < E) ? o[i] : 0.f;
                    part += __shfl_xor(part, 32);
                    piv[g] = part * (1.0f / (float)(E < CE ? E : CE));
                }
                float pr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float d = o[i] - piv[g];
                    if (!whole) d = (ce0 + 4 * kh + i < E) ? d : 0.f;
                    sum1[g] += d;
                    sum2[g] = fmaf(d, d, sum2[g]);
                    pr[i] = d * vvv[i];                              // (o - s) * v_value  :636-638
                }
                fdn_u32x2 bq[3];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    unsigned p1, p2, p3;
                    fdn_split3(pr[2 * j], pr[2 * j + 1], p1, p2, p3);
                    bq[0][j] = p1, bq[1][j] = p2, bq[2][j] = p3;
                }
                s1[g] = mfma8_split6(aop8[g], bq, s1[g]);
            }
        } else {
            float vvv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) vvv[i] = VV[(ce0 + 8 * kh + i) * VPS + opxs];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = Sf[t_off(g, 8 * kh + i, opxs)];
                if (ch == 0) {
                    float part = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) part += (8 * kh + i < E) ? o[i] : 0.f;
                    part += __shfl_xor(part, 32);
                    piv[g] = part * (1.0f / (float)(E < CE ? E : CE));
                }
                float pr[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float d = o[i] - piv[g];
                    if (!whole) d = (ce0 + 8 * kh + i < E) ? d : 0.f;
                    sum1[g] += d;
                    sum2[g] = fmaf(d, d, sum2[g]);
                    pr[i] = d * vvv[i];
                }
                fdn_u32x4 bq[3];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned p1, p2, p3;
                    fdn_split3(pr[2 * j], pr[2 * j + 1], p1, p2, p3);
                    bq[0][j] = p1, bq[1][j] = p2, bq[2][j] = p3;
                }
                s1[g] = fdn_mfma_split6(aop[g], bq, s1[g]);
            }
        }
        // (the next chunk's P1 rewrites the spectra / T only behind its barrier B1, i.e. after every wave has finished these reads)
    }

    // ---- epilogue: S0_g and Bt from the v_value rows, dense in K --------------------------------------------------------------------
    f32x16 e4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) e4[t][r] = 0.f;
    const int ke = (E + 15) / 16;
    const int oy = opx / TWP, ox = opx - oy * TWP;
    const int gy = ty0 + oy, gx = tx0 + ox;
    const bool pok = gy < H && gx < W;
    const unsigned pixb = pok ? (unsigned)(gy * W + gx) * 4u : OOB;
    // (requesting the residual and the next k-step's operands ahead - a second fragment set, 48 registers - measured slower: 3.60 against
    //  3.52 ms at C = 32, 3.05 against 2.89 ms at C = 24; the kernel then sits at 255 registers)
    for (int ks = 0; ks < ke; ++ks) {
        float vvv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = 16 * ks + 8 * kh + i;
            const float v = VV[(e < NCH * CE ? e : 0) * VPS + opxs];
            vvv[i] = e < E ? v : 0.f;
        }
        fdn_u32x4 bq[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned p1, p2, p3;
            fdn_split3(vvv[2 * j], vvv[2 * j + 1], p1, p2, p3);
            bq[0][j] = p1, bq[1][j] = p2, bq[2][j] = p3;
        }
        const fdn_u32x4* wq = a.wep + ((long)(ks * 4) * NMT + omt) * 3 * 64 + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const fdn_u32x4 a3[3] = {wq[((long)t * NMT * 3) * 64], wq[((long)t * NMT * 3 + 1) * 64], wq[((long)t * NMT * 3 + 2) * 64]};
            e4[t] = fdn_mfma_split6(a3, bq, e4[t]);
        }
    }
    // statistics of the three groups at this lane's pixel (the two lane halves hold different channels)
    float dg[3], rg[3];
    const float invE = 1.0f / (float)E;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float t1 = (sum1[g] + __shfl_xor(sum1[g], 32)) * invE;
        const float t2 = (sum2[g] + __shfl_xor(sum2[g], 32)) * invE;
        dg[g] = t1;                                                   // mean - pivot
        const float var = fmaxf(t2 - t1 * t1, 0.f);
        rg[g] = rsq(var + 1e-5f);                                     // (1-ulp hardware form, as the other LayerNorm producers)
    }
    // y = sum_g rstd_g (S1_g - d_g S0_g) + Bt + residual; rows n = omt * 32 + (r & 3) + 8 (r >> 2) + 4 kh, pixel opx
    const unsigned vo = pok ? pixb + (unsigned)(4 * kh) * hw4 : OOB;
    const rsrc_t ro = mk_rsrc(a.out + (long)b * N * P, (unsigned)N * hw4);
    const rsrc_t rr = mk_rsrc(a.res ? a.res + (long)b * N * P : a.out, a.res ? (unsigned)N * hw4 : 0u);
    float rres[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rres[r] = bload(rr, vo, (unsigned)(omt * 32 + (r & 3) + 8 * (r >> 2)) * hw4);      // 0 without a residual
    float yv[16];
    float sm = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float y = e4[3][r];
#pragma unroll
        for (int g = 0; g < 3; ++g) y = fmaf(rg[g], fmaf(-dg[g], e4[g][r], s1[g][r]), y);
        y += rres[r];
        const int n = omt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), ro, n < N ? vo : OOB, (unsigned)(omt * 32 + (r & 3) + 8 * (r >> 2)) * hw4, 0);
        yv[r] = n < N ? y : 0.f;
        sm += yv[r];
    }
    if (a.stats_out) {                                                // LayerNorm statistics of the result (two-pass, as the other producers)
        sm += __shfl_xor(sm, 32);
        if (NMT == 2) {                                               // the other 32 output channels of this pixel live on wave ^ NOS
            if (kh == 0) red[wave * 64 + ln] = sm;
            __syncthreads();
            sm += red[(wave ^ NOS) * 64 + ln];
        }
        const float mean = sm / (float)N;
        float sq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = omt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const float dl = yv[r] - mean;
            sq += n < N ? dl * dl : 0.f;
        }
        sq += __shfl_xor(sq, 32);
        if (NMT == 2) {
            if (kh == 0) red[wave * 64 + 32 + ln] = sq;
            __syncthreads();
            sq += red[(wave ^ NOS) * 64 + 32 + ln];
        }
        if (kh == 0 && omt == 0 && pok) {
            float* sp = a.stats_out + (long)b * 2 * P + (long)gy * W + gx;
            sp[0] = mean;
            sp[P] = 1.0f / sqrtf(sq / (float)N + 1e-5f);
        }
    }
}

// fdn_fdsa_full_pack: project_out [N][3E] with the gammas / betas of norm1..3 -> the two operand images of fdsa_full_kernel.
//   ws1[((ch * 3 + g) * NMT + mt) * 3 + part][lane]: rows n = mt * 32 + (lane & 31), k = 8 (lane >> 5) + j <-> channel ch * CE + k (k < CE)
//   wep[((ks * 4 + t) * NMT + mt) * 3 + part][lane]: k = 16 ks + 8 (lane >> 5) + j <-> channel e; t < 3: W gamma_t, t = 3: sum_g W beta_g
__global__ void fdsa_full_pack_kernel(const float* __restrict__ w, const float* __restrict__ gamma3, const float* __restrict__ beta3,
                                      fdn_u32x4* __restrict__ ws1, fdn_u32x4* __restrict__ wep, int N, int E, int CE, int NMT, int nchunks, int ke) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n1 = nchunks * 3 * NMT * 3 * 64, n2 = ke * 4 * NMT * 3 * 64;
    if (idx >= n1 + n2) return;
    auto part_of = [](float x, int part) {
        for (int p = 0; p < part; ++p) x -= __uint_as_float(__float_as_uint(x) & 0xffff0000u);
        return __float_as_uint(x) >> 16;
    };
    const bool ep = idx >= n1;
    int r = ep ? idx - n1 : idx;
    const int lane = r & 63;
    r >>= 6;
    const int part = r % 3;
    r /= 3;
    const int mt = r % NMT;
    r /= NMT;
    const int t = ep ? r % 4 : r % 3;
    const int blk = ep ? r / 4 : r / 3;                               // k-step (epilogue image) or chunk
    const int n = mt * 32 + (lane & 31), kh = lane >> 5;
    fdn_u32x4 o = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned hl[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // chunk image at CE = 8: the K = 8 MFMA's operand - k = 4 kh + j, j < 4, in the first two dwords of the fragment
            const int k = (!ep && CE == 8) ? (q < 2 ? 4 * kh + 2 * q + u : CE) : 8 * kh + 2 * q + u;
            const int e = ep ? 16 * blk + k : (k < CE ? blk * CE + k : E);
            float v = 0.f;
            if (n < N && e < E) {
                if (t < 3) v = w[(long)n * 3 * E + t * E + e] * gamma3[t * E + e];
                else {
                    double sacc = 0.0;
                    for (int g = 0; g < 3; ++g) sacc += (double)w[(long)n * 3 * E + g * E + e] * (double)beta3[g * E + e];
                    v = (float)sacc;
                }
            }
            hl[u] = part_of(v, part);
        }
        o[q] = hl[0] | (hl[1] << 16);
    }
    (ep ? wep[idx - n1] : ws1[idx]) = o;
}

template <int C, int PT>
int launch_full(FullArgs a, int B, hipStream_t s) {
    typedef FullGeo<C, PT> G;
    a.tiles_x = a.W / G::TWP;
    a.tiles_per_img = a.tiles_x * (a.H / 8);
    a.nchunks = (a.E + G::CE - 1) / G::CE;
    const long total = (long)B * a.tiles_per_img;
    if (total > 0x7fffffffL) return FDN_ERR_UNSUPPORTED;
    fdn_note_bf16_launch();
    if (a.stats) hipLaunchKernelGGL((fdsa_full_kernel<C, PT, true>), dim3((unsigned)total), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((fdsa_full_kernel<C, PT, false>), dim3((unsigned)total), dim3(256), 0, s, a);
    return fdn_launch_status();
}

// geometry of the operand images for a width C (shared by the pack and the launch)
struct FullLayout {
    int CE, NMT, nchunks, nch8, ke;
    long th_frag, s1_frag, ep_frag;      // 16-byte fragments
};
FullLayout full_layout(int C, int E) {
    FullLayout L;
    L.CE = C <= 32 ? 8 : 16;
    L.NMT = (C + 31) / 32;
    L.nchunks = (E + L.CE - 1) / L.CE;
    L.nch8 = L.nchunks * (L.CE / 8);
    L.ke = (E + 15) / 16;
    L.th_frag = (long)L.nch8 * (((C + 15) / 16) * 3 + 1) * 64;
    L.s1_frag = (long)L.nchunks * 3 * L.NMT * 3 * 64;
    L.ep_frag = (long)L.ke * 4 * L.NMT * 3 * 64;
    return L;
}

}  // namespace

// defined in patchfft.hip: the to_hidden operand image for a given number of 8-channel chunks
int fdn_fdsa_pack_chunks(const float* w, const float* gamma, const float* beta, float* wpk, int C, int E, int nch8, hipStream_t s, int channel_major);

static bool full_supported(int C, int E) { return (C == 24 || C == 32 || C == 48 || C == 64) && E > 0 && E <= C * 6 / 5; }

extern "C" long fdn_fdsa_full_pack_bytes(int C, int E) {
    if (!full_supported(C, E)) return 0;
    const FullLayout L = full_layout(C, E);
    return (L.th_frag + L.s1_frag + L.ep_frag) * 16;
}

extern "C" int fdn_fdsa_full_pack(const float* w_hidden, const float* gamma, const float* beta, const float* w_out, const float* gamma3,
                                  const float* beta3, void* wpk, int C, int E, fdn_stream_t stream) {
    FDN_CHECK_ARG(w_hidden && w_out && gamma3 && beta3 && wpk && (!gamma == !beta));
    if (!full_supported(C, E)) return FDN_ERR_UNSUPPORTED;
    const FullLayout L = full_layout(C, E);
    hipStream_t s = static_cast<hipStream_t>(stream);
    fdn_u32x4* base = static_cast<fdn_u32x4*>(wpk);
    const int rc = fdn_fdsa_pack_chunks(w_hidden, gamma, beta, reinterpret_cast<float*>(base), C, E, L.nch8, s, 0);
    if (rc != FDN_OK) return rc;
    const long total = L.s1_frag + L.ep_frag;
    hipLaunchKernelGGL(fdsa_full_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w_out, gamma3, beta3, base + L.th_frag,
                       base + L.th_frag + L.s1_frag, C, E, L.CE, L.NMT, L.nchunks, L.ke);
    return fdn_launch_status();
}

extern "C" int fdn_fdsa_full(const float* x, long xbs, const float* stats, const void* wpk, const float* dw_w, const float* fft_w,
                             const float* res, float* out, float* stats_out, int B, int C, int E, int H, int W, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && wpk && dw_w && fft_w && out && B > 0 && E > 0 && H > 0 && W > 0);
    FDN_CHECK_ARG(H % 8 == 0 && W % 8 == 0);
    if (fdn_matrix_pipe_f32() || !full_supported(C, E)) return FDN_ERR_UNSUPPORTED;
    if (C <= 32 && W % 16 != 0) return FDN_ERR_UNSUPPORTED;                                   // 8 x 16 tiles
    if (4ull * (C + 8) * H * W >= 0x80000000ull) return FDN_ERR_UNSUPPORTED;                  // 32-bit byte offsets per image (OOB = 2^31)
    const FullLayout L = full_layout(C, E);
    FullArgs a;
    a.x = x; a.xbs = xbs; a.stats = stats; a.dww = dw_w; a.fftw = fft_w; a.res = res; a.out = out; a.stats_out = stats_out;
    a.wth = static_cast<const fdn_u32x4*>(wpk);
    a.ws1 = a.wth + L.th_frag;
    a.wep = a.ws1 + L.s1_frag;
    a.E = E; a.H = H; a.W = W; a.N = C;
    a.tiles_x = a.tiles_per_img = a.nchunks = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (C) {
        case 24: return launch_full<24, 2>(a, B, s);
        case 32: return launch_full<32, 2>(a, B, s);
        case 48: return launch_full<48, 1>(a, B, s);
        case 64: return launch_full<64, 1>(a, B, s);
    }
    return FDN_ERR_UNSUPPORTED;
}
