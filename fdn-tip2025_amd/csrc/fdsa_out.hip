// FDSA tail in one launch (FDN_arch.py:633-639 + the residual of :671): three channel LayerNorms of
// out1|out2|out3, the v_value gate, project_out (1x1, 3E -> C) and `x + ...`, plus the LayerNorm
// statistics of the result for the next sub-block.
//
// A wave owns 32 pixels.  It loads ALL 3E+E inputs of its pixels exactly once into registers
// (lane half kh holds the channels e = 2s + kh), derives the three (mean, rstd) pairs from those
// registers (two-pass, like the reference), normalises/gates in place and feeds the values straight to
// v_mfma_f32_32x32x2_f32 as the B operand; the transposed weights sit in LDS.  No statistics kernel,
// no second read of the 3E-channel tensor.  Register budget limits this form to E <= 38 (level 1,
// where this tail costs the most); levels 2 and 3 use fdn_chan_stats + fdn_conv1x1(PRO_LN3_GATE).
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

struct FoArgs {
    const float* o;        // [B][4E][P]: out1 | out2 | out3 | v_value
    const float* w;        // [N][3E]
    const float* gamma;    // [3E]
    const float* beta;     // [3E]
    const float* res;      // [B][N][P]
    float* out;            // [B][N][P]
    float* stats_out;      // [B][2][P] or null
    int B, E, N, P;
    int tiles_per_img, total_tiles;
};

constexpr int NW = 4;

// SH = ceil(E/2) k-steps per group, MT = ceil(N/32)
template <int SH, int MT>
__global__ __launch_bounds__(NW * 64, SH <= 19 ? 2 : 1) void fdsa_out_kernel(FoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int E2 = 2 * SH;
    constexpr int WS = MT * 32 + 1;
    float* tg = smem;                      // gamma [3][E2]
    float* tb = smem + 3 * E2;             // beta  [3][E2]
    float* Wl = smem + 6 * E2;             // [3][E2][WS]
    const int E = a.E, N = a.N;
    const unsigned P = (unsigned)a.P, P4 = P * 4u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;

    for (int i = tid; i < 3 * E2; i += NW * 64) {
        const int g = i / E2, e = i - g * E2;
        tg[i] = e < E ? a.gamma[g * E + e] : 0.f;
        tb[i] = e < E ? a.beta[g * E + e] : 0.f;
    }
    for (int idx = tid; idx < 3 * E2 * MT * 32; idx += NW * 64) {
        const int k = idx % (3 * E2), n = idx / (3 * E2);
        const int g = k / E2, e = k - g * E2;
        Wl[k * WS + n] = (n < N && e < E) ? a.w[(long)n * 3 * E + g * E + e] : 0.f;
    }
    __syncthreads();

    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        const int b = tile / a.tiles_per_img;
        const unsigned p_ = (unsigned)(tile - b * a.tiles_per_img) * (NW * 32) + wave * 32 + ln;
        const bool ok = p_ < P;
        const unsigned pix = ok ? p_ : P - 1;
        const float* ob = a.o + (long)b * 4 * E * P;
        const rsrc_t r0 = mk_rsrc(ob, (unsigned)E * P4);
        const rsrc_t r1 = mk_rsrc(ob + (long)E * P, (unsigned)E * P4);
        const rsrc_t r2 = mk_rsrc(ob + (long)2 * E * P, (unsigned)E * P4);
        const rsrc_t rv = mk_rsrc(ob + (long)3 * E * P, (unsigned)E * P4);
        const unsigned voff = (kh * P + pix) * 4u;

        float v0[SH], v1[SH], v2[SH], vv[SH];
#pragma unroll
        for (int s = 0; s < SH; ++s) {                    // channel e = 2s + kh; e >= E reads 0 (outside the descriptor)
            const unsigned so = (unsigned)(2 * s) * P4;
            v0[s] = bload(r0, voff, so);
            v1[s] = bload(r1, voff, so);
            v2[s] = bload(r2, voff, so);
            vv[s] = bload(rv, voff, so);
        }
        // ---- three LayerNorm statistics from registers (two-pass; lanes l and l^32 split the channels) ----
        float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int s = 0; s < SH; ++s) { m0 += v0[s]; m1 += v1[s]; m2 += v2[s]; }
        const float invE = 1.0f / (float)E;
        m0 = (m0 + __shfl_xor(m0, 32)) * invE;
        m1 = (m1 + __shfl_xor(m1, 32)) * invE;
        m2 = (m2 + __shfl_xor(m2, 32)) * invE;
        float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
        for (int s = 0; s < SH; ++s) {
            const bool live = 2 * s + kh < E;
            const float d0 = v0[s] - m0, d1 = v1[s] - m1, d2 = v2[s] - m2;
            q0 += live ? d0 * d0 : 0.f;
            q1 += live ? d1 * d1 : 0.f;
            q2 += live ? d2 * d2 : 0.f;
        }
        const float s0 = 1.0f / sqrtf((q0 + __shfl_xor(q0, 32)) * invE + 1e-5f);
        const float s1 = 1.0f / sqrtf((q1 + __shfl_xor(q1, 32)) * invE + 1e-5f);
        const float s2 = 1.0f / sqrtf((q2 + __shfl_xor(q2, 32)) * invE + 1e-5f);
        // ---- normalise, gate (in place) --------------------------------------------------------------------
#pragma unroll
        for (int s = 0; s < SH; ++s) {
            const int e = 2 * s + kh;
            v0[s] = ((v0[s] - m0) * s0 * tg[e] + tb[e]) * vv[s];                       // norm1(out1) * v_value  :633,636
            v1[s] = ((v1[s] - m1) * s1 * tg[E2 + e] + tb[E2 + e]) * vv[s];             // :634,637
            v2[s] = ((v2[s] - m2) * s2 * tg[2 * E2 + e] + tb[2 * E2 + e]) * vv[s];     // :635,638
        }
        // ---- project_out on MFMA -----------------------------------------------------------------------------
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
        for (int s = 0; s < SH; ++s) {
            const float* w0 = Wl + (2 * s + kh) * WS + ln;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[m * 32], v0[s], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[E2 * WS + m * 32], v1[s], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[2 * E2 * WS + m * 32], v2[s], acc[m], 0, 0, 0);
            }
        }
        // ---- epilogue: residual, store, next LayerNorm's statistics ------------------------------------------
        if (ok) {
            const unsigned nb4 = (unsigned)N * P4;
            const rsrc_t ro = mk_rsrc(a.out + (long)b * N * P, nb4);
            const rsrc_t rr = mk_rsrc(a.res ? a.res + (long)b * N * P : a.out, a.res ? nb4 : 0u);
            const unsigned vo = (4u * kh * P + pix) * 4u;
            float sm = 0.f;
            // the residual operand is requested as one batch before the stores: interleaved with them every load is
            // waited for on its own (16 memory round trips per tile instead of one; tools/gemm_trace.py)
            float rres[MT * 16];
#pragma unroll
            for (int i = 0; i < MT * 16; ++i)
                rres[i] = bload(rr, vo, (unsigned)((i >> 4) * 32 + (i & 3) + 8 * ((i & 15) >> 2)) * P4);   // 0 without a residual
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nrow = m * 32 + (r & 3) + 8 * (r >> 2);
                    const unsigned so = (unsigned)nrow * P4;
                    float v = acc[m][r];
                    v += rres[m * 16 + r];
                    bstore(v, ro, vo, so);
                    v = (nrow + 4 * kh < N) ? v : 0.f;
                    acc[m][r] = v;
                    sm += v;
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
                if (kh == 0) {
                    float* sp = a.stats_out + (long)b * 2 * P;
                    sp[pix] = mean;
                    sp[P + pix] = 1.0f / sqrtf(sq / (float)N + 1e-5f);
                }
            }
        }
    }
}

// 8-byte-lane form for level 1 (E <= 38, N <= 32, P % 4 == 0): a lane owns two consecutive pixels, so every load / store
// moves 8 bytes per lane (the dword form stops at ~3.2 TB/s).  To stay within two waves per SIMD the three LayerNorm
// groups are taken one after the other - v_value stays in registers, out_g is loaded (the next group while this one is
// normalised and multiplied), its statistics come from registers (two-pass), and its slice of project_out accumulates
// into the same two MFMA chains.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 bload2(rsrc_t r, unsigned voff, unsigned soff) {
    const u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return f32x2{__uint_as_float(u.x), __uint_as_float(u.y)};
}
__device__ __forceinline__ void bstore2(f32x2 v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v.x), __float_as_uint(v.y)}, r, voff, soff, 0);
}

// DB: prefetch the next group into a second register set (level 1); without it the other wave on the SIMD covers the
// load latency (level 2: 76-channel groups, two 32-row tiles - a second set would not leave two waves per SIMD).
template <bool BF>
__device__ __forceinline__ f32x2 oload2(rsrc_t r, unsigned voff, unsigned soff) {      // two pixels of an `o` plane (fp32 or bf16 storage)
    float v[2];
    st_load2<BF>(v, r, voff, soff);
    return f32x2{v[0], v[1]};
}

// PX pixels per lane: 2 (8-byte lanes, levels 1-2 as built in rounds 2-3) or 1 (4-byte lanes: half the data registers).
// PBF (round 4, with PX = 1 and NWV = 8 waves sharing one operand image): project_out on the bf16 matrix pipe.  The 76-channel form of level 2 is bound
// by its fp32 MFMAs - 0.80 ms, 0.53 without them (knock-out), 228 x 2 x 2 of them per 64 pixels at 64 cycles each on the vector ALU's datapath.  Here a
// lane's eight consecutive values of a group (channels 2 (8 q + i) + kh: exactly k = 8 kh + i of a 32x32x16 operand) are cut into three exact bf16
// parts (common.hpp: the six leading products are fp32-grade) and the weights wait in LDS as packed A operands, cut once per workgroup.
template <int PX> struct PxT;
template <> struct PxT<2> { typedef f32x2 T; };
template <> struct PxT<1> { typedef float T; };
__device__ __forceinline__ float pcomp(float v, int) { return v; }
__device__ __forceinline__ float pcomp(f32x2 v, int i) { return i ? v.y : v.x; }
__device__ __forceinline__ float xsum32(float v) { return v + __shfl_xor(v, 32); }
__device__ __forceinline__ f32x2 xsum32(f32x2 v) { return f32x2{v.x + __shfl_xor(v.x, 32), v.y + __shfl_xor(v.y, 32)}; }
__device__ __forceinline__ float rsqrt_eps(float v) { return 1.0f / sqrtf(v + 1e-5f); }
__device__ __forceinline__ f32x2 rsqrt_eps(f32x2 v) { return f32x2{1.0f / sqrtf(v.x + 1e-5f), 1.0f / sqrtf(v.y + 1e-5f)}; }
__device__ __forceinline__ void mkpx(float& o, const f32x16 (&acc)[1], int r) { o = acc[0][r]; }
__device__ __forceinline__ void mkpx(f32x2& o, const f32x16 (&acc)[2], int r) { o = f32x2{acc[0][r], acc[1][r]}; }
__device__ __forceinline__ void bloadp(float& v, rsrc_t r, unsigned voff, unsigned soff) { v = bload(r, voff, soff); }
__device__ __forceinline__ void bloadp(f32x2& v, rsrc_t r, unsigned voff, unsigned soff) { v = bload2(r, voff, soff); }
__device__ __forceinline__ void bstorep(float v, rsrc_t r, unsigned voff, unsigned soff) { bstore(v, r, voff, soff); }
__device__ __forceinline__ void bstorep(f32x2 v, rsrc_t r, unsigned voff, unsigned soff) { bstore2(v, r, voff, soff); }
template <bool BF> __device__ __forceinline__ void oloadp(float& v, rsrc_t r, unsigned voff, unsigned soff) { v = st_load1<BF>(r, voff, soff); }
template <bool BF> __device__ __forceinline__ void oloadp(f32x2& v, rsrc_t r, unsigned voff, unsigned soff) { v = oload2<BF>(r, voff, soff); }

// IBF: the (out1|out2|out3|v_value) planes are stored as bf16 (written so by fdn_fdsa_fused); statistics, products and the result stay fp32
template <int SH, int MT, bool DB, bool IBF, int PX, int NWV, bool PBF>
__global__ __launch_bounds__(NWV * 64, NWV == 8 ? 1 : 2) void fdsa_out_vec_kernel(FoArgs a) {
    typedef typename PxT<PX>::T T;
    static_assert(!PBF || PX == 1, "the bf16-pipe projection exists in the one-pixel form");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int E2 = 2 * SH;
    constexpr int WS = MT * 32 + 1;
    constexpr int NQ = (SH + 7) / 8;       // PBF: 16-deep k-steps per group
    float* tg = smem;                      // gamma [3][E2]
    float* tb = smem + 3 * E2;             // beta  [3][E2]
    float* Wl = smem + 6 * E2;             // [3][E2][WS]
    fdn_u32x4* Wp = reinterpret_cast<fdn_u32x4*>(smem + ((6 * E2 + 3) & ~3));      // PBF: [3][NQ][MT][part][lane] packed A operands
    const int E = a.E, N = a.N;
    const unsigned P = (unsigned)a.P, P4 = P * 4u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;

    for (int i = tid; i < 3 * E2; i += NWV * 64) {
        const int g = i / E2, e = i - g * E2;
        tg[i] = e < E ? a.gamma[g * E + e] : 0.f;
        tb[i] = e < E ? a.beta[g * E + e] : 0.f;
    }
    if constexpr (PBF) {
        for (int u = tid; u < 3 * NQ * MT * 3 * 64; u += NWV * 64) {
            const int l = u & 63, part = (u >> 6) % 3, mt = (u / 192) % MT, q = (u / (192 * MT)) % NQ, g = u / (192 * MT * NQ);
            const int n = mt * 32 + (l & 31), k2 = l >> 5;
            fdn_u32x4 o;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                unsigned hl[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int s_ = 8 * q + 2 * d + h, e = 2 * s_ + k2;
                    float x = (n < N && s_ < SH && e < E) ? a.w[(long)n * 3 * E + g * E + e] : 0.f;
                    for (int pp = 0; pp < part; ++pp) x -= fdn_trunc_bf16(x);
                    hl[h] = __float_as_uint(x) >> 16;
                }
                o[d] = hl[0] | (hl[1] << 16);
            }
            Wp[u] = o;
        }
    } else {
        for (int idx = tid; idx < 3 * E2 * MT * 32; idx += NWV * 64) {
            const int k = idx % (3 * E2), n = idx / (3 * E2);
            const int g = k / E2, e = k - g * E2;
            Wl[k * WS + n] = (n < N && e < E) ? a.w[(long)n * 3 * E + g * E + e] : 0.f;
        }
    }
    __syncthreads();
    const float invE = 1.0f / (float)E;

    constexpr unsigned IES = st_bytes<IBF>();
    const unsigned PI = P * IES;                              // bytes per `o` plane
    // tile -> image, pixel pair of this lane, descriptor of group g's E planes
    auto tile_b = [&](int tile) { return tile / a.tiles_per_img; };
    auto tile_p = [&](int tile, int b) { return (unsigned)(tile - b * a.tiles_per_img) * (NWV * 32 * PX) + (wave * 32 + ln) * PX; };
    auto oplanes = [&](int b, int g) {
        const char* ob = reinterpret_cast<const char*>(a.o) + (long)b * 4 * E * P * IES;
        return mk_rsrc(reinterpret_cast<const float*>(ob + (long)g * E * P * IES), (unsigned)E * PI);
    };
    T vv[SH], oa[SH], ob2[DB ? SH : 1];
    bool fetched = false;                                     // DB: v_value and group 0 of this tile were requested by the previous tile
    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        // The plane offsets 2 s PI / n P4 below are loop invariants: hoisted out of this persistent loop they are ~100 live scalars, and the
        // compiler parked them in vector-register lanes - 249 v_writelane + 307 v_readlane + 375 wait states in the 76-channel form.  Opaque
        // per-trip copies of the two strides keep each offset one s_mul beside its use.
        unsigned PIl = PI, P4l = P4;
        asm volatile("" : "+s"(PIl), "+s"(P4l));
        // (the same for the channel-range predicates 2 s + kh < E / n + 4 kh < N: hoisted, each is a 64-bit lane mask - 140 scalar registers)
        int khl = kh;
        asm volatile("" : "+v"(khl));
        const int b = tile_b(tile);
        const unsigned p_ = tile_p(tile, b);
        const bool ok = p_ < P;                               // P % 2 == 0: a pixel pair is inside or outside as a whole
        const unsigned pix = ok ? p_ : P - PX;
        const rsrc_t rg[3] = {oplanes(b, 0), oplanes(b, 1), oplanes(b, 2)};
        const rsrc_t rv = oplanes(b, 3);
        const unsigned voff = (kh * P + pix) * IES;          // channel e = 2s + kh; e >= E reads 0 (outside the descriptor)

        if (DB && fetched) {
#pragma unroll
            for (int s = 0; s < (DB ? SH : 1); ++s) oa[s] = ob2[s];
        } else {
#pragma unroll
            for (int s = 0; s < SH; ++s) {
                oloadp<IBF>(vv[s], rv, voff, (unsigned)(2 * s) * PIl);
                oloadp<IBF>(oa[s], rg[0], voff, (unsigned)(2 * s) * PIl);
            }
        }
        const unsigned nb4 = (unsigned)N * P4;
        const rsrc_t ro = mk_rsrc(a.out + (long)b * N * P, nb4);
        const rsrc_t rr = mk_rsrc(a.res ? a.res + (long)b * N * P : a.out, a.res ? nb4 : 0u);
        const unsigned vo = ok ? (4u * kh * P + pix) * 4u : 0x80000000u;
        T rres[MT][16];
        f32x16 acc[MT][PX];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int v = 0; v < PX; ++v)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][v][r] = 0.f;

#pragma unroll
        for (int g = 0; g < 3; ++g) {
            T* cur = (DB && (g & 1)) ? ob2 : oa;
            if (DB && g < 2) {
                T* nxt = (g & 1) ? oa : ob2;
#pragma unroll
                for (int s = 0; s < SH; ++s) oloadp<IBF>(nxt[s], rg[g + 1], voff, (unsigned)(2 * s) * PIl);
            }
            // LayerNorm statistics of this group from registers (two-pass; lanes l and l^32 split the channels)
            T m = 0.f;
#pragma unroll
            for (int s = 0; s < SH; ++s) m += cur[s];
            m = xsum32(m) * invE;
            T q = 0.f;
#pragma unroll
            for (int s = 0; s < SH; ++s) {
                const T dl = cur[s] - m;
                q += (2 * s + khl < E) ? dl * dl : T(0.f);
            }
            const T rs = rsqrt_eps(xsum32(q) * invE);
#pragma unroll
            for (int s = 0; s < SH; ++s) {
                asm volatile("" ::: "memory");                              // table reads stay here (registers)
                const int e = 2 * s + kh;
                cur[s] = ((cur[s] - m) * rs * tg[g * E2 + e] + tb[g * E2 + e]) * vv[s];      // norm_g(out_g) * v_value  :633-638
            }
            if (DB && g == 2) {
                // the NEXT tile's v_value and group 0 go into the two register sets that are dead from here (v_value was last used
                // just above, the idle set holds group 1): their round trip hides behind this tile's last MFMAs and its epilogue
                // (fp32 storage only: a bf16 load is followed by its unpacking, which would wait for the data right here - measured
                //  16.3 -> 30.9 ms per step at 1080p with the prefetch in the bf16 form)
                const int nt = tile + (int)gridDim.x;
                fetched = !IBF && nt < a.total_tiles;
                if constexpr (DB && !IBF) if (fetched) {
                    const int nb = tile_b(nt);
                    const unsigned np_ = tile_p(nt, nb);
                    const unsigned nvoff = (kh * P + (np_ < P ? np_ : P - PX)) * IES;
                    const rsrc_t nrv = oplanes(nb, 3), nr0 = oplanes(nb, 0);
#pragma unroll
                    for (int s = 0; s < SH; ++s) {
                        oloadp<IBF>(vv[s], nrv, nvoff, (unsigned)(2 * s) * PIl);
                        oloadp<IBF>(ob2[s], nr0, nvoff, (unsigned)(2 * s) * PIl);
                    }
                }
            }
            if (DB && g == 2) {      // (level 2 has no idle set: requested ahead it spills 39 registers, so it stays in the epilogue there)
                // the residual is requested here: v_value (and, with DB, the idle register set) is dead from this point, and the
                // round trip hides behind the last group's MFMAs instead of standing alone in the epilogue
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        bloadp(rres[mt][r], rr, vo, (unsigned)(mt * 32 + (r & 3) + 8 * (r >> 2)) * P4l);      // 0 without a residual
            }
            if constexpr (PBF) {
#pragma unroll
                for (int q8 = 0; q8 < NQ; ++q8) {
                    fdn_u32x4 bx[3];
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd) {
                        const int s0 = 8 * q8 + 2 * dd, s1 = s0 + 1;
                        unsigned p1, p2, p3;
                        fdn_split3(s0 < SH ? pcomp(cur[s0 < SH ? s0 : 0], 0) : 0.f, s1 < SH ? pcomp(cur[s1 < SH ? s1 : 0], 0) : 0.f, p1, p2, p3);
                        bx[0][dd] = p1, bx[1][dd] = p2, bx[2][dd] = p3;
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const fdn_u32x4* wp = Wp + (((g * NQ + q8) * MT + mt) * 3) * 64 + lane;
                        const fdn_u32x4 a3[3] = {wp[0], wp[64], wp[128]};
                        acc[mt][0] = fdn_mfma_split6(a3, bx, acc[mt][0]);
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < SH; ++s) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float wa = Wl[(g * E2 + 2 * s + kh) * WS + mt * 32 + ln];
#pragma unroll
                        for (int c = 0; c < PX; ++c) acc[mt][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, pcomp(cur[s], c), acc[mt][c], 0, 0, 0);
                    }
                }
            }
            if (!DB && g < 2) {
#pragma unroll
                for (int s = 0; s < SH; ++s) oloadp<IBF>(oa[s], rg[g + 1], voff, (unsigned)(2 * s) * PIl);
            }
        }
        // ---- epilogue: residual (one batch), store, next LayerNorm's statistics -------------------------------------
        T outv[MT][16];
        if (!DB) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    bloadp(rres[mt][r], rr, vo, (unsigned)(mt * 32 + (r & 3) + 8 * (r >> 2)) * P4l);      // 0 without a residual
        }
        T sm = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = mt * 32 + (r & 3) + 8 * (r >> 2);
                T o;
                mkpx(o, acc[mt], r);
                o += rres[mt][r];
                bstorep(o, ro, vo, (unsigned)nrow * P4l);
                outv[mt][r] = (nrow + 4 * khl < N) ? o : T(0.f);
                sm += outv[mt][r];
            }
        if (a.stats_out) {
            T sq = 0.f;
            const T mean = xsum32(sm) / (float)N;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const T dl = outv[mt][r] - mean;
                    sq += (mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * khl < N) ? dl * dl : T(0.f);
                }
            const T rstd = rsqrt_eps(xsum32(sq) / (float)N);
            if (kh == 0) {
                const rsrc_t rs_ = mk_rsrc(a.stats_out + (long)b * 2 * P, 2u * P4);
                const unsigned vs = ok ? pix * 4u : 0x80000000u;
                bstorep(mean, rs_, vs, 0u);
                bstorep(rstd, rs_, vs, P4);
            }
        }
    }
}

template <int SH, int MT, bool DB, bool IBF, int PX = 2, int NWV = NW, bool PBF = false>
int launch_vec(FoArgs a, hipStream_t s) {
    const size_t lds = PBF ? (((6UL * 2 * SH + 3) & ~3UL) * sizeof(float) + 3UL * ((SH + 7) / 8) * MT * 3 * 64 * 16)
                           : (6UL * 2 * SH + 3UL * 2 * SH * (MT * 32 + 1)) * sizeof(float);
    const int cus = fdn_device_cus();
    if (cus <= 0) return FDN_ERR_LAUNCH;
    if (lds > 48 * 1024 && !fdn_allow_dynamic_lds(reinterpret_cast<const void*>(fdsa_out_vec_kernel<SH, MT, DB, IBF, PX, NWV, PBF>), lds)) return FDN_ERR_LAUNCH;
    a.tiles_per_img = cdiv(a.P, NWV * 32 * PX);
    a.total_tiles = a.B * a.tiles_per_img;
    int grid = cus * (NWV == 8 ? 1 : 2);
    if (grid > a.total_tiles) grid = a.total_tiles;
    if (PBF) fdn_note_bf16_launch();
    hipLaunchKernelGGL((fdsa_out_vec_kernel<SH, MT, DB, IBF, PX, NWV, PBF>), dim3(grid), dim3(NWV * 64), lds, s, a);
    return fdn_launch_status();
}

template <int SH, int MT>
int launch(FoArgs a, hipStream_t s) {
    const size_t lds = (6UL * 2 * SH + 3UL * 2 * SH * (MT * 32 + 1)) * sizeof(float);
    const int g_cus = fdn_device_cus();
    if (g_cus <= 0) return FDN_ERR_LAUNCH;
    a.tiles_per_img = cdiv(a.P, NW * 32);
    a.total_tiles = a.B * a.tiles_per_img;
    auto kern = fdsa_out_kernel<SH, MT>;
    if (lds > 48 * 1024 && !fdn_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return FDN_ERR_LAUNCH;
    int per_cu = (int)((160 * 1024) / lds);
    const int want = SH <= 19 ? 2 : 1;
    if (per_cu > want) per_cu = want;
    if (per_cu < 1) per_cu = 1;
    int grid = g_cus * per_cu;
    if (grid > a.total_tiles) grid = a.total_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, a);
    return fdn_launch_status();
}

}  // namespace

extern "C" int fdn_fdsa_out(const void* o_, const float* w, const float* gamma3, const float* beta3, const float* res,
                            float* out, float* stats_out, int B, int E, int N, int P, int o_bf16, fdn_stream_t stream) {
    const float* o = static_cast<const float*>(o_);
    FDN_CHECK_ARG(o && w && gamma3 && beta3 && out && B > 0 && E > 0 && N > 0 && P > 0);
    if ((unsigned long long)(4 * E + 2) * 4ull * P > 0xFFFFFFFFull || (unsigned long long)(N + 40) * 4ull * P > 0xFFFFFFFFull)
        return FDN_ERR_UNSUPPORTED;
    FoArgs a;
    a.o = o; a.w = w; a.gamma = gamma3; a.beta = beta3; a.res = res; a.out = out; a.stats_out = stats_out;
    a.B = B; a.E = E; a.N = N; a.P = P;
    a.tiles_per_img = a.total_tiles = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int sh = (E + 1) / 2, mt = (N + 31) / 32;
    const bool vec_ok = P % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res) |
          reinterpret_cast<uintptr_t>(stats_out)) & 15) == 0;
    if (o_bf16) {                                               // bf16 storage: the same kernels reading 2-byte elements (storage only: tests/test_gpu_bf16.py)
        if (vec_ok && sh <= 19 && mt == 1) return launch_vec<19, 1, true, true>(a, s);
        if (fdn_matrix_pipe_wide() && vec_ok && sh > 19 && sh <= 38 && mt <= 2) return launch_vec<38, 2, false, true, 1, 8, true>(a, s);
        if (vec_ok && sh <= 38 && mt <= 2) return launch_vec<38, 2, false, true>(a, s);
        return FDN_ERR_UNSUPPORTED;
    }
    if (vec_ok && sh <= 19 && mt == 1) return launch_vec<19, 1, true, false>(a, s);       // level 1, 8-byte lanes
    // level 2.  (round 4) one pixel per lane, eight waves around one packed operand image, project_out on the bf16 matrix pipe - see the note at PxT.
    // The default since round 5 (fdn_set_matrix_pipe(2) keeps the fp32-MFMA form below for A/B runs): as accurate against float64 as that form
    // (tests/test_gpu_parity.py), a different - equally valid - rounding.
    if (fdn_matrix_pipe_wide() && vec_ok && sh > 19 && sh <= 38 && mt <= 2) return launch_vec<38, 2, false, false, 1, 8, true>(a, s);
    if (vec_ok && sh <= 38 && mt <= 2) return launch_vec<38, 2, false, false>(a, s);
    if (sh <= 19 && mt == 1) return launch<19, 1>(a, s);       // level 1: E = 38, C = 32
    if (sh <= 38 && mt <= 2) return launch<38, 2>(a, s);       // level 2: E = 76, C = 64 (one wave per SIMD, 490 registers: 1.74 vs 1.88 ms)
    return FDN_ERR_UNSUPPORTED;                                  // caller falls back to stats + conv1x1
}
