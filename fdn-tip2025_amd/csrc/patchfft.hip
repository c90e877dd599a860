// 8x8-patch spectral kernels: the "one launch per block" fused FFT <-> pointwise <-> iFFT of the
// FDformer (FDSA core and FDFFN middle).  Everything between the global load of the activation
// tile and the global store of the result stays in LDS / registers:
//   depthwise 3x3 (halo tile) -> rfft2 per 8x8 patch (real 8-point transforms via a 4-point
//   complex FFT in registers, row pass then column pass through LDS) -> amplitude/phase
//   recombination without atan2/sincos (e^{i(ang q - ang k)} = u_q * conj(u_k), e^{i ang v} = v/|v|;
//   SURVEY.md App. C; v_rsq_f32 for the normalisations) -> irfft2 -> 32-byte row-segment stores.
// Tile = 32 x 64 pixels (4 x 8 patches) of one channel per 256-thread workgroup; thread = (patch,
// row) computes its own 8 stencil outputs and feeds them straight into the row transform, so the
// only LDS traffic is the halo tile (one channel at a time, double buffered) and the spectra.
#include "patch_fft.hpp"
#include "fdsa_tail.hpp"
#include <type_traits>
#ifndef FDN_RD_VOTE
#define FDN_RD_VOTE 1      // replace_denormals by wave vote (fdsa_bin, patch_fft.hpp); 0: every compare / select as written (A/B builds)
#endif

namespace {

// ------------------------------------------------------------------------------------------
// halo tiles: (TH+2) x (TW+2) floats of one plane, row stride HS (odd: thread (patch,row) reads hit
// distinct banks); loaded through registers so the next channel's loads fly during the transforms
// ------------------------------------------------------------------------------------------
constexpr int HS = 65 + 2;         // 67: (r*67 + 8*px) mod 32 distinct for r<8, px<4
constexpr int HALO = (TH + 2) * (TW + 2);
constexpr int HPT = (HALO + 255) / 256;   // 9 elements per thread

// per-thread global byte offsets / LDS slots of its HPT halo elements (computed once, reused for every plane)
__device__ __forceinline__ void halo_offsets(int H, int W, int y0, int x0, unsigned (&goff)[HPT], int (&slot)[HPT]) {
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int rr = idx / (TW + 2), cc = idx - rr * (TW + 2);
        const int y = y0 - 1 + rr, x = x0 - 1 + cc;
        const bool ok = idx < HALO && y >= 0 && y < H && x >= 0 && x < W;
        goff[i] = ok ? (unsigned)(y * W + x) * 4u : OOB;
        slot[i] = idx < HALO ? rr * HS + cc : (TH + 2) * HS;          // spare cell behind the tile (a conditional store
                                                                         // lets hipcc sink the load into the branch: an extra round trip)
    }
}
__device__ __forceinline__ void halo_fetch(rsrc_t r, unsigned plane_off, const unsigned (&goff)[HPT], float (&v)[HPT]) {
#pragma unroll
    for (int i = 0; i < HPT; ++i) v[i] = bload(r, goff[i], plane_off);
}
__device__ __forceinline__ void halo_stash(float* t, const int (&slot)[HPT], const float (&v)[HPT]) {
#pragma unroll
    for (int i = 0; i < HPT; ++i) t[slot[i]] = v[i];       // unconditional: slots past the tile point at a spare cell
}
// 16-byte-lane form of the halo fetch (W % 4 == 0, 16-byte aligned planes): the 64 interior columns of the TH+2 rows
// travel as float4 (3 per thread), the two edge columns as dwords (threads 0..2(TH+2)-1): 4 load instructions per plane
// instead of 9, and a wave touches 1 KB contiguous
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 bload4(rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return f32x4{__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
}
struct HaloV4 {
    unsigned g4[3], ge;      // global byte offsets (OOB when outside the image / past the tile)
    int s4[3], se;           // LDS slots (spare cells when past the tile)
};
template <int HALO_W, int STRIDE, int ROWS>                 // HALO_W = halo width on each side (1 or 2)
__device__ __forceinline__ HaloV4 halo_v4_setup(int H, int W, int y0, int x0) {
    HaloV4 h;
    constexpr int NV = ROWS * 16;
    constexpr int SPARE = ROWS * STRIDE;                    // 4 spare floats behind the tile
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int r = idx >> 4, c4 = idx & 15;
        const int y = y0 - HALO_W + r, xx = x0 + 4 * c4;
        const bool ok = idx < NV && y >= 0 && y < H && xx < W;
        h.g4[i] = ok ? (unsigned)(y * W + xx) * 4u : OOB;
        h.s4[i] = idx < NV ? r * STRIDE + HALO_W + 4 * c4 : SPARE;
    }
    constexpr int NE = ROWS * 2 * HALO_W;                   // edge elements: HALO_W columns on each side
    const int e = threadIdx.x;
    const int er = e / (2 * HALO_W), ek = e - er * (2 * HALO_W);
    const int ec = ek < HALO_W ? ek : TW + ek;              // columns 0..HALO_W-1 and TW+HALO_W..TW+2*HALO_W-1
    const int ey = y0 - HALO_W + er, ex = x0 - HALO_W + ec;
    const bool eok = e < NE && ey >= 0 && ey < H && ex >= 0 && ex < W;
    h.ge = eok ? (unsigned)(ey * W + ex) * 4u : OOB;
    h.se = e < NE ? er * STRIDE + ec : SPARE;
    return h;
}
struct HaloV4Data { f32x4 q[3]; float e; };
__device__ __forceinline__ void halo_v4_fetch(rsrc_t r, unsigned plane_off, const HaloV4& h, HaloV4Data& d) {
#pragma unroll
    for (int i = 0; i < 3; ++i) d.q[i] = bload4(r, h.g4[i], plane_off);
    d.e = bload(r, h.ge, plane_off);
}
__device__ __forceinline__ void halo_v4_stash(float* t, const HaloV4& h, const HaloV4Data& d) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t[h.s4[i] + j] = d.q[i][j];
    t[h.se] = d.e;
}

// 3x3 stencil for the 8 pixels of row `row`, columns col0..col0+7 of the tile (halo origin -1,-1)
__device__ __forceinline__ void stencil_row8(const float* t, int row, int col0, const float (&w)[9], float (&o)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        float v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = t[(row + dy) * HS + col0 + j];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) o[j] = fmaf(w[dy * 3 + dx], v[j + dx], o[j]);
    }
}

// ------------------------------------------------------------------------------------------
// FDSA core
// ------------------------------------------------------------------------------------------
template <bool V4>
__global__ __launch_bounds__(256, 3) void fdsa_core_kernel(const float* __restrict__ hidden, const float* __restrict__ dww,
                                                           const float* __restrict__ fftw, float* __restrict__ out, int E,
                                                           int H, int W, int tiles_x, int ntiles, int EPB) {
    __shared__ float halo[2][(TH + 2) * HS + 4];
    __shared__ __attribute__((aligned(16))) float2 S[3 * NP * PS];

    const int tid = threadIdx.x;
    // (round 4) a workgroup walks EPB channels of one tile: fewer, longer workgroups
    const int ngrp = (E + EPB - 1) / EPB;
    const unsigned item = xcd_contiguous(blockIdx.x, gridDim.x);           // (b, channel group, tile), tile fastest
    const int tile = item % ntiles, e_first = ((item / ntiles) % ngrp) * EPB, b = item / (ntiles * ngrp);
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    const unsigned hw4 = (unsigned)H * W * 4u;                  // bytes per plane; 4E planes per image < 4 GB (checked by the host)
    const long base = (long)b * 4 * E * H * W;
    const rsrc_t rin = mk_rsrc(hidden + base, 4u * E * hw4);
    const rsrc_t rout = mk_rsrc(out + base, 4u * E * hw4);
    const int patch = tid >> 3, rr = tid & 7;
    const int py = patch >> 3, px = patch & 7;
    const int gy = ty0 + py * 8 + rr, gx = tx0 + px * 8;
    const bool inside = gy < H && gx < W;          // patches are entirely inside or outside (H, W % 8 == 0)
    const unsigned ooff = inside ? (unsigned)(gy * W + gx) * 4u : OOB;
    unsigned goff[HPT];
    int slot[HPT];
    float pre[HPT];
    HaloV4 hv;
    HaloV4Data hd;
    auto fetch = [&](int plane) {
        if (V4) halo_v4_fetch(rin, (unsigned)plane * hw4, hv, hd);
        else halo_fetch(rin, (unsigned)plane * hw4, goff, pre);
    };
    auto stash = [&](float* t) {
        if (V4) halo_v4_stash(t, hv, hd);
        else halo_stash(t, slot, pre);
    };
    if (V4) hv = halo_v4_setup<1, HS, TH + 2>(H, W, ty0, tx0);
    else halo_offsets(H, W, ty0, tx0, goff, slot);
    fetch(e_first);
    for (int e = e_first; e < e_first + EPB && e < E; ++e) {
    if (e != e_first) __syncthreads();                                     // the previous channel's inverse rows have read the spectra
    // the column-phase gains of this channel (thread = (patch, kx)) are requested first: their latency hides behind
    // the whole row phase instead of stalling the column phase
    float fg[8];
    {
        const int kxc = tid < NP * 5 ? tid % 5 : 0;
#pragma unroll
        for (int ky = 0; ky < 8; ++ky) fg[ky] = fftw[(e * 8 + ky) * 5 + kxc];
    }
    stash(halo[0]);                       // this channel's q plane: requested before the loop / behind the previous channel's column phase
    __syncthreads();

#pragma unroll
    for (int t = 0; t < 3; ++t) {
        fetch((t + 1) * E + e);
        float wk[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) wk[i] = dww[(t * E + e) * 9 + i];
        float o8[8];
        stencil_row8(halo[t & 1], py * 8 + rr, px * 8, wk, o8);          // to_hidden_dw, FDN_arch.py:578
        float2 sp[5];
        rfft8_row(o8, sp);
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) S[(t * NP + patch) * PS + kx * KXS + rr] = sp[kx];
        stash(halo[(t + 1) & 1]);
        __syncthreads();
    }
    // (halo[1] now holds the v_value plane: its depthwise conv goes straight out, from the wave the column phase leaves idle)
    // (round 5) the NEXT channel's q plane travels during this channel's column and inverse phases - it used to be requested at the top of the
    // channel and waited for on the spot, one exposed memory round trip per channel: level 3 0.501 -> 0.487 ms, bit-identical.  (Two planes in
    // flight throughout - a second register set, 168 registers - measured the same: 0.486 ms)
    if (e + 1 < e_first + EPB && e + 1 < E) fetch(e + 1);

    // ---- columns: thread = (patch, kx): forward, recombine, inverse ---------------------------
    if (tid < NP * 5) {
        const int pj = tid / 5, kx = tid - pj * 5;
        float2 o1[8], o2[8], o3[8];
        // the column as written (every replace_denormals on every component), or - first - without them and with a running minimum of the
        // magnitudes they would test (patch_fft.hpp, fdsa_bin); a wave that meets a value below 1e-10 evaluates its columns again as written
        auto column = [&](auto as_written) __attribute__((always_inline)) {
            constexpr bool AW = decltype(as_written)::value;
            float2 q[8], k[8], v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                q[i] = S[(0 * NP + pj) * PS + kx * KXS + i];
                k[i] = S[(1 * NP + pj) * PS + kx * KXS + i];
                v[i] = S[(2 * NP + pj) * PS + kx * KXS + i];
            }
            fft8<false>(q);
            fft8<false>(k);
            fft8<false>(v);
            float m = 1.0f;
#pragma unroll
            for (int ky = 0; ky < 8; ++ky) {
                float2 u, v1;
                float qka, g, va;
                if (ky % 4 == 0) fdsa_bin<AW, true>(q[ky], k[ky], v[ky], fg[ky], m, u, v1, qka, g, va);
                else fdsa_bin<AW, false>(q[ky], k[ky], v[ky], fg[ky], m, u, v1, qka, g, va);
                o1[ky] = make_float2(va * u.x, va * u.y);                                     // :609-612
                o2[ky] = make_float2(g * v1.x, g * v1.y);                                     // qka e^{i v_p} :617-619
                o3[ky] = make_float2(qka * u.x, qka * u.y);                                   // :627-629
            }
            return m;
        };
#if FDN_RD_VOTE
        if (__builtin_amdgcn_ballot_w64(column(std::false_type{}) < 1e-10f) != 0) column(std::true_type{});
#else
        column(std::true_type{});
#endif
        fft8<true>(o1);
        fft8<true>(o2);
        fft8<true>(o3);
        constexpr float sc = 1.0f / 64.0f;   // norm='backward'
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            S[(0 * NP + pj) * PS + kx * KXS + i] = make_float2(o1[i].x * sc, o1[i].y * sc);
            S[(1 * NP + pj) * PS + kx * KXS + i] = make_float2(o2[i].x * sc, o2[i].y * sc);
            S[(2 * NP + pj) * PS + kx * KXS + i] = make_float2(o3[i].x * sc, o3[i].y * sc);
        }
    } else if (tid >= 192) {
        // the 160 column jobs fill waves 0-2; wave 3 runs the v_value path of the whole tile meanwhile: four (patch, row) jobs per lane
        float wk[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) wk[i] = dww[(3 * E + e) * 9 + i];
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            const int J = (tid - 192) + 64 * i, pj = J >> 3, rj = J & 7;
            const int pyj = pj >> 3, pxj = pj & 7;
            const int gyj = ty0 + pyj * 8 + rj, gxj = tx0 + pxj * 8;
            float o8[8];
            stencil_row8(halo[1], pyj * 8 + rj, pxj * 8, wk, o8);
            bstore8(o8, rout, (gyj < H && gxj < W) ? (unsigned)(gyj * W + gxj) * 4u : OOB, (unsigned)(3 * E + e) * hw4);
        }
    }
    __syncthreads();

    // ---- inverse rows, 32-byte segments straight to global (out1|out2|out3) --------------------------
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float2 x[5];
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) x[kx] = S[(t * NP + patch) * PS + kx * KXS + rr];
        float r[8];
        irfft8_row(x, r);
        bstore8(r, rout, ooff, (unsigned)(t * E + e) * hw4);
    }
    }   // channels of this workgroup
}

// ------------------------------------------------------------------------------------------
// FDFFN middle: freq branch + (dw3x3 -> GELU -> dw3x3) spatial branch
// ------------------------------------------------------------------------------------------
constexpr int LS2 = 73;            // odd: lanes walking rows (ring) or (row, patch) pairs (row FFT) hit distinct banks
constexpr int LSM = 67;

// channels per workgroup (software-pipelined), chosen per launch: as many as leave ~8 rounds of workgroups on the chip, 4 .. 22.  Round 4,
// interleaved A/B at B = 8 720p: 4 channels 1.702 ms, 8 1.666, 11 1.654, 22 1.646 (level 2: 0.877 / 0.846 / 0.846 / 0.837) - fewer, longer
// workgroups: the fp32 form loses its time to workgroup turnover behind the store path, not inside the waves (profiles/r04_mid_pmc.txt)
constexpr int CPB_MIN = 4, CPB_MAX = 22;
constexpr int HALO2 = (TH + 4) * (TW + 4);
constexpr int HPT2 = (HALO2 + 255) / 256;               // 10 elements per thread

// (round 6) tools/tail_trace.py mid: -DFDN_MID_TRACE - per wave, summed over its channels, the s_memtime clocks of phase A (ring conv + GELU + forward
// rows), barrier, B (park the next halo + columns), barrier, C (inverse rows + second conv + stores), barrier; -DFDN_KOM_GELU / _CONV2 / _COLS / _STORE /
// _LOADS knock a piece out (results are wrong: timing only)
#ifdef FDN_MID_TRACE
constexpr int MT_NWG = 512;
__device__ unsigned long long g_mid_trace[MT_NWG * 4 * 16];
#define MTR(var) const unsigned long long var = __builtin_amdgcn_s_memtime();
#else
#define MTR(var)
#endif
template <bool V4, bool IBF, bool OBF>      // IBF / OBF: x / out are stored as bf16 (fp32 math either way; V4 needs fp32 input)
#ifndef FDN_MID_WGS
#define FDN_MID_WGS 3
#endif
#ifndef FDN_MID_CONV2_LATE
#define FDN_MID_CONV2_LATE 1
#endif

__global__ __launch_bounds__(256, FDN_MID_WGS) void fdffn_mid_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                           const float* __restrict__ w2, const float* __restrict__ ffta,
                                                           const float* __restrict__ fftp, float* __restrict__ out, int Hd,
                                                           int H, int W, int tiles_x, int ntiles, int CPB) {
    __shared__ float tin[(TH + 4) * LS2 + 4];       // halo 2 (+ spare cells)
    __shared__ float mid[(TH + 2) * LSM];           // gelu(dw0(x)) on halo 1
    __shared__ __attribute__((aligned(16))) float2 S[NP * PS];
    __shared__ float2 filt[40];                     // ffta * e^{-i fftp} per (ky, kx)
#ifdef FDN_MID_TRACE
    const unsigned long long mt_entry = __builtin_amdgcn_s_memtime();
    unsigned long long mt_sum[6] = {0, 0, 0, 0, 0, 0};
    int mt_nch = 0;
#endif

    const int tid = threadIdx.x;
    const int ngroups = (Hd + CPB - 1) / CPB;
    const unsigned item = xcd_contiguous(blockIdx.x, gridDim.x);           // (b, channel group, tile), tile fastest
    // (the run-time divisions go through the vector ALU: without the readfirstlanes the plane offsets below count as per-lane values and
    //  every halo load of the channel loop sits in a waterfall loop - 12 of them, ~150 issue slots per channel; tools/isa_waterfall.py)
    const int tile = __builtin_amdgcn_readfirstlane((int)(item % ntiles)), cbase = __builtin_amdgcn_readfirstlane((int)((item / ntiles) % ngroups) * CPB),
              b = __builtin_amdgcn_readfirstlane((int)(item / (ntiles * ngroups)));
    const int ty0 = __builtin_amdgcn_readfirstlane((tile / tiles_x) * TH), tx0 = __builtin_amdgcn_readfirstlane((tile % tiles_x) * TW);
    constexpr unsigned IES = st_bytes<IBF>(), OES = st_bytes<OBF>();
    const unsigned hw4 = (unsigned)H * W * IES;                 // bytes per input plane; Hd planes per image < 4 GB (checked by the host)
    const unsigned hwo = (unsigned)H * W * OES;
    const rsrc_t rin = mk_rsrc(reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + (long)b * Hd * H * W * IES), (unsigned)Hd * hw4);
    const rsrc_t rout = mk_rsrc(reinterpret_cast<const float*>(reinterpret_cast<const char*>(out) + (long)b * Hd * H * W * OES), (unsigned)Hd * hwo);
    const int patch = tid >> 3, rr = tid & 7;
    const int py = patch >> 3, px = patch & 7;
    const int gy = ty0 + py * 8 + rr, gx = tx0 + px * 8;
    const unsigned ooff = (gy < H && gx < W) ? (unsigned)(gy * W + gx) * OES : OOB;

    unsigned goff[HPT2];
    int slot[HPT2];
#pragma unroll
    for (int i = 0; i < HPT2; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / (TW + 4), cc = idx - r * (TW + 4);
        const int y = ty0 - 2 + r, xx = tx0 - 2 + cc;
        const bool ok = idx < HALO2 && y >= 0 && y < H && xx >= 0 && xx < W;
        goff[i] = ok ? (unsigned)(y * W + xx) * IES : OOB;
        slot[i] = idx < HALO2 ? r * LS2 + cc : (TH + 4) * LS2;
    }
    float pre[HPT2];
    HaloV4 hv;
    HaloV4Data hd;
    if (V4) hv = halo_v4_setup<2, LS2, TH + 4>(H, W, ty0, tx0);
    // bf16 storage: the halo tile starts on an even column of an even-width image, so it is loaded as dwords of two pixels
    // (single 2-byte loads made the bf16-input form 0.57 ms SLOWER than fp32 at level 1; as pairs it is 0.45 ms faster)
    constexpr int PW = (TW + 4) / 2, NPAIR = (TH + 4) * PW, HPP = (NPAIR + 255) / 256;
    unsigned goffp[IBF ? HPP : 1];
    int slotp[IBF ? HPP : 1];
    unsigned prep[IBF ? HPP : 1];
    if constexpr (IBF) {
#pragma unroll
        for (int i = 0; i < HPP; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx / PW, cp = idx - r * PW;
            const int y = ty0 - 2 + r, xx = tx0 - 2 + 2 * cp;
            const bool ok = idx < NPAIR && y >= 0 && y < H && xx >= 0 && xx < W;      // W and xx even: a pair is inside or outside as a whole
            goffp[i] = ok ? (unsigned)(y * W + xx) * 2u : OOB;
            slotp[i] = idx < NPAIR ? r * LS2 + 2 * cp : (TH + 4) * LS2;
        }
    }
    // (round 6) The loop issues the SAME loads on every path: a channel that does not exist is requested through a descriptor of zero records (returns 0,
    // moves nothing).  With the request inside `if (more)` the compiler's wait-count pass merged the two paths to "nothing younger in flight": the top of
    // every channel waited for one of the two stores just issued (vmcnt(1) in front of the halo request) and wave 0, which builds the filter, for both
    // (vmcnt(0)) - the kernel ran 19 % faster without its stores (tools/tail_trace.py, profiles/r06_tail_mid_trace.txt).
    const rsrc_t rdead = mk_rsrc(x, 0u);
    const rsrc_t rfa = mk_rsrc(ffta, (unsigned)Hd * 160u), rfp = mk_rsrc(fftp, (unsigned)Hd * 160u);
    auto fetch = [&](int c, bool live) {
#ifdef FDN_KOM_LOADS
        live = live && c == cbase;
#endif
        const rsrc_t ri = live ? rin : rdead;
        // (the plane offset must reach the loads in a scalar register: the compiler keeps the strength-reduced c * hw4 of the channel loop in a
        //  VECTOR register and then wraps every load in a waterfall loop - 12 per channel, tools/isa_waterfall.py)
        const unsigned po = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)c * hw4));
        if (V4) { halo_v4_fetch(ri, po, hv, hd); return; }
        if constexpr (IBF) {
#pragma unroll
            for (int i = 0; i < HPP; ++i) prep[i] = __builtin_amdgcn_raw_buffer_load_b32(ri, goffp[i], po, 0);
        } else {
#pragma unroll
            for (int i = 0; i < HPT2; ++i) pre[i] = st_load1<IBF>(ri, goff[i], po);
        }
    };
    auto stash = [&]() {
        if (V4) { halo_v4_stash(tin, hv, hd); return; }
        if constexpr (IBF) {
#pragma unroll
            for (int i = 0; i < HPP; ++i) {
                tin[slotp[i]] = bf16_lo(prep[i]);                  // (spare cells for slots past the tile)
                tin[slotp[i] + 1] = bf16_hi(prep[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < HPT2; ++i) tin[slot[i]] = pre[i];   // unconditional (spare cell for slots past the tile)
        }
    };
    fetch(cbase, true);
    stash();
    // (ffta, fftp) of the NEXT channel travel with its halo: loaded right here, in the middle of the loop, they would be
    // waited for with vmcnt(0), which also drains the halo prefetch issued just before (the counter is in-order)
    const unsigned ftoff = tid < 40 ? (unsigned)tid * 4u : OOB;
    float fa = bload(rfa, ftoff, (unsigned)cbase * 160u), fp = bload(rfp, ftoff, (unsigned)cbase * 160u);
    // the filter ffta e^{-i fftp} of a channel is built one channel ahead, in phase C of its predecessor (the first one here): at the top of the loop
    // it would wait for (fa, fp) with everything younger - the stores just issued - in front of it
    auto build_filter = [&]() {
        if (tid < 40) {
            float sn, cs;
            fdn_sincos(fp, &sn, &cs);
            filt[tid] = make_float2(fa * cs, -fa * sn);
        }
    };
    build_filter();
    __syncthreads();
    const bool ring_inside = ty0 >= 1 && ty0 + TH + 1 <= H && tx0 >= 1 && tx0 + TW + 1 <= W;
    for (int ci = 0; ci < CPB; ++ci) {
        const int c = cbase + ci;
        if (c >= Hd) break;                                   // uniform
        const bool more = ci + 1 < CPB && c + 1 < Hd;
        MTR(mt0)
        {
            fetch(c + 1, more);                               // next channel's halo flies during this one's math
            const rsrc_t ra = more ? rfa : rdead, rp = more ? rfp : rdead;
            const unsigned fo = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(c + 1) * 160u));
            fa = bload(ra, ftoff, fo);
            fp = bload(rp, ftoff, fo);
        }
        float k0[9], k2[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            k0[i] = w0[c * 9 + i];
            k2[i] = w2[c * 9 + i];
        }

        // ---- A: first depthwise conv + GELU on the (TH+2) x (TW+2) ring (zero outside the image = the
        // zero padding the second conv sees, FDN_arch.py:439); row segments of 8 share a 3 x 10 window
        auto ring_segment = [&](int job) {
            const int r = job % (TH + 2), c0 = (job / (TH + 2)) * 8;   // consecutive lanes -> consecutive rows
            float o8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o8[j] = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                float v[10];
#pragma unroll
                for (int j = 0; j < 10; ++j) v[j] = tin[(r + dy) * LS2 + c0 + j];
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) o8[j] = fmaf(k0[dy * 3 + dx], v[j + dx], o8[j]);
            }
#ifndef FDN_GELU_SCALAR
            if (ring_inside) {                                    // uniform: the whole ring of this tile lies in the image (all but border tiles)
#pragma unroll
                for (int j = 0; j < 8; j += 2) {                  // GELU on pairs: packed fp32 for the polynomial and the products
#ifdef FDN_KOM_GELU
                    const fdn_f32x2 gv = fdn_f32x2{o8[j], o8[j + 1]};
#else
                    const fdn_f32x2 gv = gelu_fast2(fdn_f32x2{o8[j], o8[j + 1]});
#endif
                    mid[r * LSM + c0 + j] = gv.x;
                    mid[r * LSM + c0 + j + 1] = gv.y;
                }
                return;
            }
#endif
            const int y = ty0 - 1 + r;
            const bool yok = y >= 0 && y < H;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int xx = tx0 - 1 + c0 + j;
                mid[r * LSM + c0 + j] = (yok && xx >= 0 && xx < W) ? gelu_fast(o8[j]) : 0.f;     // (a select around an unconditional GELU measured slower: 2.18 vs 1.83 ms)
            }
        };
        ring_segment(tid);                                        // 256 of the 272 segments
        // what is left of the ring - segments 256..271 (128 pixels) and columns 64, 65 of every row (68) - goes out ONE PIXEL per
        // thread to 196 threads of all four waves: as whole segments on threads 0-15 plus single pixels on the next 68 it kept
        // wave 0 busy for ~290 more instructions than waves 2 and 3, and the barrier below waits for the slowest wave
        if (tid < 128 + 2 * (TH + 2)) {
            int r, cc;
            if (tid < 128) {
                const int job = 256 + (tid >> 3);
                r = job % (TH + 2);
                cc = (job / (TH + 2)) * 8 + (tid & 7);
            } else {
                const int i = tid - 128;
                r = i >> 1;
                cc = 64 + (i & 1);
            }
            const int y = ty0 - 1 + r, xx = tx0 - 1 + cc;
            float a = 0.f;
            if (y >= 0 && y < H && xx >= 0 && xx < W) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) a = fmaf(k0[dy * 3 + dx], tin[(r + dy) * LS2 + cc + dx], a);
                a = gelu_fast(a);
            }
            mid[r * LSM + cc] = a;
        }
        // forward rows of the frequency branch straight from the input tile (centre of tin)
        {
            float r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = tin[(2 + py * 8 + rr) * LS2 + 2 + px * 8 + i];
            float2 o[5];
            rfft8_row(r, o);
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) S[patch * PS + kx * KXS + rr] = o[kx];
        }
        MTR(mt1)
        __syncthreads();
        MTR(mt2)

        // ---- B: tin is free: park the prefetched halo; second conv; column transforms --------------
        stash();                                                  // (behind the last channel: zeros, read by nobody)
        auto second_conv = [&](float (&sp)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sp[j] = 0.f;
#ifdef FDN_KOM_CONV2
            sp[0] = mid[(py * 8 + rr) * LSM + px * 8];
            return;
#endif
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                float v[10];
#pragma unroll
                for (int j = 0; j < 10; ++j) v[j] = mid[(py * 8 + rr + dy) * LSM + px * 8 + j];
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) sp[j] = fmaf(k2[dy * 3 + dx], v[j + dx], sp[j]);
            }
        };
#if FDN_MID_CONV2_LATE == 0
        float sp[8];
        second_conv(sp);
#endif
        // columns: forward, z * ffta * e^{-i fftp}, inverse  (FDN_arch.py:460-469)
#ifdef FDN_KOM_COLS
        if (false) {
#else
        if (tid < NP * 5) {
#endif
            const int pj = tid / 5, kx = tid - pj * 5;
            float2 z[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) z[i] = S[pj * PS + kx * KXS + i];
            fft8<false>(z);
#if FDN_RD_VOTE
            // replace_denormals (:460) touches a value only when |v| < 1e-10 - in practice the imaginary parts of the four self-conjugate bins
            // (exactly 0 for kx = 0, 4 at ky = 0, 4) and nothing else.  Those two are replaced as written; for the other 14 values the wave takes
            // the minimum magnitude and runs the compare / select pairs only when some lane needs one (same result bit for bit; 32 -> 13
            // instructions on the common path)
            z[0].y = rd1(z[0].y);
            z[4].y = rd1(z[4].y);
            {
                float m = fminf(fabsf(z[0].x), fabsf(z[4].x));
#pragma unroll
                for (int ky = 1; ky < 8; ++ky)
                    if (ky != 4) m = fminf(fminf(m, fabsf(z[ky].x)), fabsf(z[ky].y));
                if (__builtin_amdgcn_ballot_w64(m < 1e-10f) != 0) {
#pragma unroll
                    for (int ky = 0; ky < 8; ++ky) z[ky] = make_float2(rd1(z[ky].x), rd1(z[ky].y));
                }
            }
#pragma unroll
            for (int ky = 0; ky < 8; ++ky) {                                                      // :461-468
                // (the fused products spelled out - the form the compiler chose for the as-written code: bit-identical to it, profiles/r05_n_vote.txt)
                const float2 a = z[ky], b = filt[ky * 5 + kx];
                z[ky] = make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
            }
#else
#pragma unroll
            for (int ky = 0; ky < 8; ++ky)
                z[ky] = cmul(make_float2(rd1(z[ky].x), rd1(z[ky].y)), filt[ky * 5 + kx]);     // :461-468
#endif
            fft8<true>(z);
            constexpr float sc = 1.0f / 64.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) S[pj * PS + kx * KXS + i] = make_float2(z[i].x * sc, z[i].y * sc);
        }
        MTR(mt3)
        __syncthreads();
        MTR(mt4)

        // ---- C: inverse rows + spatial branch, 32-byte segments to global --------------------------------
        if (more) build_filter();                                 // the columns of this channel are done with `filt`; (fa, fp) of channel c + 1 landed long ago
        {
            float2 xk[5];
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) xk[kx] = S[patch * PS + kx * KXS + rr];
            float r[8];
            irfft8_row(xk, r);
#if FDN_MID_CONV2_LATE
            float sp[8];
            second_conv(sp);
#endif
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += sp[j];                                                              // :470
#ifdef FDN_KOM_STORE
            if (r[0] == 123.456f)
#endif
            st_store8<OBF>(r, rout, ooff, (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)c * hwo)));
        }
        MTR(mt5)
        __syncthreads();                                          // S, mid, filt are rewritten by the next channel
#ifdef FDN_MID_TRACE
        {
            const unsigned long long mt6 = __builtin_amdgcn_s_memtime();
            mt_sum[0] += mt1 - mt0; mt_sum[1] += mt2 - mt1; mt_sum[2] += mt3 - mt2; mt_sum[3] += mt4 - mt3; mt_sum[4] += mt5 - mt4; mt_sum[5] += mt6 - mt5;
            ++mt_nch;
        }
#endif
    }
#ifdef FDN_MID_TRACE
    {
        const unsigned rel = blockIdx.x - gridDim.x / 2;
        if (rel < (unsigned)MT_NWG && (tid & 63) == 0) {
            unsigned long long* t = g_mid_trace + ((long)rel * 4 + (tid >> 6)) * 16;
            const unsigned long long now = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i] = mt_sum[i];
            t[6] = (unsigned long long)mt_nch; t[7] = now - mt_entry;
        }
    }
#endif
}


// ------------------------------------------------------------------------------------------
// FDSA front half in ONE launch: channel LayerNorm + to_hidden (1x1 conv on the matrix cores) + everything
// fdsa_core_kernel does (FDN_arch.py:575-632) - the 4E-channel hidden tensor never exists in HBM.
//
// Workgroup = one 8 x 32 pixel tile (1 x 4 patches) of one image, ALL E channels, 256 threads.
//   * the normalised input strip of the 10 x 34 halo tile (C channels) is loaded once and lives in registers as the
//     B operand of v_mfma_f32_32x32x16_bf16 (pixels on the lane axis; three exact bf16 parts per value, so the product is
//     fp32 arithmetic on the bf16 matrix pipe, which - unlike the fp32 MFMA - runs beside the vector ALU): 11 strips of
//     32 halo pixels, three (waves 0-2) or two (wave 3) per wave, 12 VGPRs per 16 channels each;
//   * channels are walked in chunks of 8: the 32 MFMA rows of a chunk are (q,k,v,v_value) x 8 channels (weights
//     packed per chunk / k-step / part / lane by fdn_fdsa_pack, LayerNorm affine folded in), each wave runs 6 MFMAs per
//     16 channels and strip (+ 1 for the bias) and parks the 32 x 32 result in the LDS hidden tile (zero outside the image
//     = the conv's zero padding);
//   * then thread = (channel of the chunk, patch, row): stencil -> row rfft -> LDS spectra -> column phase
//     (160 threads) -> inverse rows -> 32-byte stores, the code of fdsa_core_kernel.
// HBM traffic: C planes in (+33 % halo, mostly L2 hits: each XCD owns a contiguous run of tiles), 4E planes out.
// Measured (MI355X, level 1, B = 8): 3.0-3.2 ms against 1.1-1.4 + 2.2-2.5 ms for fdn_conv1x1 + fdn_fdsa_core.  PMC view
// (profiles/r02_fdsa_fused_pmc.txt): 6.0k vector instructions per wave at 4 issue cycles each + 234 fp32 MFMAs at 64 cycles are 62 % of
// a wave's life; a second workgroup per CU buys nothing (one per CU: 3.02 ms, two: 3.06 ms) and staggering the pair changes
// nothing either - the fp32 MFMA and the vector ALU do not overlap on a SIMD, their cycles add.
// ------------------------------------------------------------------------------------------
constexpr int FT_H = 8, FT_W = 32;
constexpr int FHW = FT_W + 2, FHH = FT_H + 2;      // halo tile 10 x 34
constexpr int FHP = FHH * FHW;                     // 340 halo pixels
constexpr int FNS = (FHP + 31) / 32;               // 11 strips of 32 pixels
constexpr int FRS = 35;                            // LDS row stride of a hidden plane: (35 r + 8 px) mod 32 distinct for r < 8, px < 4
constexpr int FPL = FHH * FRS;                     // floats per plane
constexpr int FEG = 8;                             // channels per chunk (x 4 kinds = 32 MFMA rows)
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FusedArgs {
    const float* x;
    long xbs;
    const float* stats;
    const float* wpk;
    const float* dww;
    const float* fftw;
    float* out;
    int E, H, W, tiles_x, tiles_per_img, nchunks;
    // TAIL (fdn_fdsa_block): `out` is the per-tile scratch [tile][4E][8][32]; the tail's operand image, the residual, the result, its statistics
    const float* tw;
    const float* res;
    float* y;
    float* stats_out;
    int N;
    float* h;            // TAIL 3: the following FDFFN's project_in output [B][Hd][H][W] (h_bf16: stored as bf16)
    int Hd, h_bf16;
};
#ifndef FDN_RING
#define FDN_RING 1          // TAIL kernels: 1 = `out` is a RING of per-(CU, resident workgroup) blocks behind FDN_RING_SLOTS flag words (0 = free); 0 = one block per tile (A/B builds)
#endif
// (round 6) The tile-local hand-off as a ring.  With one block per TILE (4.6 GB per launch at level 1) every byte of it is written to HBM once; a
// workgroup only needs its block for its own lifetime, so the blocks are handed out per RESIDENT workgroup instead - CU (XCC_ID, SE_ID, SH_ID, CU_ID of
// HW_ID) x the two workgroups a CU holds (80 KB of LDS each) - and rewritten in place: ~80 MB that stay in the 256 MB Infinity Cache (tools/micro/ring_probe.hip).
// A slot is taken with two atomic exchanges in flight at once (one round trip; the loser of a race retries) and given back by the wave whose last
// read-back load landed last.  A collision of the id bits could only make a workgroup wait for a slot, never share one.
constexpr int FDN_RING_CUS = 2048;                 // 3 + 3 + 1 + 4 id bits
constexpr int FDN_RING_SLOTS = 2 * FDN_RING_CUS;
constexpr int FDN_RING_HDR = 16384;                // floats in front of the blocks (the flags, padded to 64 KB)

#ifndef FDN_FUSED_WGS
#define FDN_FUSED_WGS 2
#endif
#ifndef FDN_STAGE_OPAQUE
#define FDN_STAGE_OPAQUE 1      // 1: the tail + project_in kernels (TAIL 3) rebuild stage_fetch's indices per chunk (as C = 64 does) instead of carrying them: carried they
                                // are spilled there (a scratch reload per chunk behind vmcnt(0)): 3.575 -> 3.525 ms at C = 32, 0.755 -> 0.736 at C = 24 (profiles/r06_tail_ab6*.txt)
#endif
#ifndef FDN_FUSED_CELLS
#define FDN_FUSED_CELLS 1
#endif
#ifdef FDN_FUSED_TRACE      // tools/fused_trace.py: s_memtime stamps of every wave of FT_NWG workgroups from the middle of the grid (phase timeline per SIMD)
constexpr int FT_NWG = 512;
__device__ unsigned long long g_fused_trace[FT_NWG * 4 * 64];
#define FTR(i) if (trc && ch < 7) trc[ch * 8 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define FTR(i)
#endif
// TAIL (round 6): 0 = the (out1|out2|out3|v_value) planes go to the [B][4E][H][W] tensor for fdn_fdsa_out; 1 = they go to this tile's own 4E x 1 KB
// block of a scratch tensor and the workgroup runs fdn_fdsa_out's arithmetic on them itself (fdsa_tail.hpp): the whole sub-block in one launch
template <int C, bool LN, bool OBF, int TAIL = 0>          // OBF: the (out1|out2|out3|v_value) planes are stored as bf16
__global__ __launch_bounds__(256, FDN_FUSED_WGS) void fdsa_fused_kernel(FusedArgs a) {
    static_assert(!TAIL || !OBF, "the in-kernel tail reads fp32 planes");
    // (round 4) C <= 32: the hidden tile is ONE plane per channel of (q, k, v, v_value) CELLS - the MFMA rows of a chunk are channel-major,
    // so a lane's accumulator holds whole cells - and the taps are (wq, wk, wv, wvv) cells too: a window position is one 16-byte read
    // and the four depthwise convs advance as two v_pk_fma_f32 (144 packed FMAs per thread and chunk where the three row-phase stencils
    // took 216 scalar ones and wave 3 ran the fourth beside the column phase).  C >= 48 (108 / 144 registers of strips) keeps the planar form.
    constexpr bool CELLS = FDN_FUSED_CELLS && (C + 15) / 16 <= 2;
    __shared__ __attribute__((aligned(16))) float hid[32 * FPL + (FDN_FUSED_WGS == 1 ? 2048 : 0)];      // (A/B hook: a pad that leaves room for one workgroup per CU only)
    __shared__ __attribute__((aligned(16))) float2 S[3 * NP * PS];
    __shared__ __attribute__((aligned(16))) float wks[32 * 9];        // depthwise taps of the chunk: [kind * 8 + channel][9]; CELLS: [channel][tap][kind]
    __shared__ __attribute__((aligned(16))) float fgs[FEG * 40];      // fft gains of the chunk's channels: [channel][ky][kx]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;
    const int t_ = (int)xcd_contiguous(blockIdx.x, gridDim.x);            // every XCD walks a contiguous run of tiles
    const int b = t_ / a.tiles_per_img, ti = t_ - b * a.tiles_per_img;
    const int ty0 = (ti / a.tiles_x) * FT_H, tx0 = (ti % a.tiles_x) * FT_W;
    const int E = a.E, H = a.H, W = a.W;
    const unsigned P = (unsigned)H * W, hw4 = P * 4u;
    const rsrc_t rx = mk_rsrc(a.x + (long)b * a.xbs, (unsigned)C * hw4);
    const rsrc_t rst = mk_rsrc(LN ? a.stats + (long)b * 2 * P : a.x, LN ? 2u * hw4 : 0u);
    constexpr unsigned OES = st_bytes<OBF>();
    const unsigned hwo = TAIL ? 1024u : P * OES;                                          // bytes per output plane (TAIL: of this tile's block)
    const float* scr_tile = a.out + (long)t_ * 4 * E * 256;                                // (TAIL: this tile's block; ring: replaced behind the first barrier)
    rsrc_t rout = TAIL ? mk_rsrc(scr_tile, 4u * E * 1024u)
                       : mk_rsrc(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.out) + (long)b * 4 * E * P * OES), 4u * E * hwo);

#ifdef FDN_FUSED_TRACE
    unsigned long long* trc = nullptr;
    {
        const unsigned base = gridDim.x / 2, rel = blockIdx.x - base;
        if (rel < (unsigned)FT_NWG && lane == 0) {
            trc = g_fused_trace + ((long)rel * 4 + wave) * 64;
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            trc[60] = hwid; trc[61] = xcc; trc[62] = blockIdx.x; trc[63] = __builtin_amdgcn_s_memtime();
        }
    }
#endif
    __shared__ int ring_s[4];                      // (TAIL, ring) [0] the workgroup's slot, [1] waves whose read-back has landed, [2..3] its block's address
    if constexpr (TAIL != 0 && FDN_RING) {
        if (tid == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const unsigned cu = ((xcc & 7u) << 8) | (((hwid >> 13) & 7u) << 5) | (((hwid >> 12) & 1u) << 4) | ((hwid >> 8) & 0xFu);
            unsigned* f = reinterpret_cast<unsigned*>(a.out) + 2 * cu;
            int got = -1;
            while (got < 0) {
                const unsigned o0 = __hip_atomic_exchange(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned o1 = __hip_atomic_exchange(f + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (o0 == 0) {
                    got = 0;
                    if (o1 == 0) __hip_atomic_store(f + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // took both: one goes back
                } else if (o1 == 0) {
                    got = 1;
                } else {
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            ring_s[0] = (int)(2 * cu) + got;
            ring_s[1] = 0;
            const unsigned long long blk = reinterpret_cast<unsigned long long>(a.out + FDN_RING_HDR + (long)(2 * cu + got) * (4 * a.E * 256));
            ring_s[2] = (int)(unsigned)blk;
            ring_s[3] = (int)(unsigned)(blk >> 32);
        }
    }
    // (and: s_setprio 2 / 3 for the prologue - a fresh workgroup's waves are the youngest of their SIMDs - 2.380 / 2.375 against 2.322 ms: slower)
    // (round 5, measured and dropped: an L2 warm-up of the tile 2 x CUs blocks ahead - one dword per line of its body into a register nothing reads -
    //  changes nothing: 2.322 against 2.306 ms, level 2 1.441 against 1.390, although the build without the strip loads runs 12 % / 10 % faster)
    // ---- the wave's strips of the normalised halo tile: B operands of v_mfma_f32_32x32x16_bf16, resident for the whole
    // workgroup.  Lane (pixel ln, half kh) holds channels k = 16 ks + 8 kh + j, normalised and cut into three exact bf16
    // parts (common.hpp: fp32 arithmetic on the bf16 matrix pipe); channels k >= C read 0 and meet zero weights.
    constexpr int KST = (C + 15) / 16;
    fdn_u32x4 xb[3][KST][3];
    unsigned onebits = 0;               // bit si: B operand of the bias step is bf16 1.0 on k = 0, 1, 2 (the bias' three parts): this
                                        // lane is in the lower half and its pixel of strip si lies inside the image
    int pixoff[3];
#pragma unroll
    for (int si = 0; si < 3; ++si) {
        const int s = wave + 4 * si;
        const int p = s * 32 + ln;
        const int r = p / FHW, c = p - r * FHW;
        const int gy = ty0 - 1 + r, gx = tx0 - 1 + c;
        const bool in_tile = s < FNS && p < FHP;
        const bool ok = in_tile && gy >= 0 && gy < H && gx >= 0 && gx < W;
        onebits |= (ok && kh == 0) ? (1u << si) : 0u;
        pixoff[si] = in_tile ? r * FRS + c : FHW;                 // lanes past the tile: the unused pad cell of row 0 of each plane
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

    // VALU-phase coordinates: lanes 0-31 / 32-63 of a wave take two different channels of the chunk
    const int el = wave * 2 + kh;
    const int row = lane & 7, px = (lane >> 3) & 3;
    const int slot = el * 4 + px;
    const int gx0 = tx0 + px * 8;
    const unsigned opix = TAIL ? (unsigned)(row * 32 + px * 8) * 4u : gx0 < W ? (unsigned)((ty0 + row) * W + gx0) * OES : OOB;
    const float* hb = hid + el * FPL + row * FRS + px * 8;

    // Per-chunk operands.  A operands (packed per chunk / k-step / lane by fdn_fdsa_pack, last k-step = the bias row against
    // `xone`): registers, reloaded for the NEXT chunk right after this chunk's MFMAs so the loads fly during the spectral
    // phases.  Depthwise taps and fft gains: one element per thread, staged through LDS a chunk ahead.
    constexpr int KS = KST * 3 + 1;
    constexpr bool AW_AHEAD = KS <= 7;
    fdn_u32x4 aw[AW_AHEAD ? KS : 1];
    auto aw_fetch = [&](int ch) {
        const fdn_u32x4* wp = reinterpret_cast<const fdn_u32x4*>(a.wpk) + ((long)ch * KS) * 64 + lane;
#pragma unroll
        for (int j = 0; j < (AW_AHEAD ? KS : 1); ++j) aw[j] = wp[j * 64];
    };
    float st_w = 0.f, st_w1 = 0.f, st_f0 = 0.f, st_f1 = 0.f;
    // (round 5: descriptor loads with 32-bit offsets - as plain pointers the lane-invariant halves of the 64-bit addresses were hoisted out of the chunk
    //  loop and, at C = 64, spilled: five scratch reloads per chunk, each waiting for every load AND store in flight - scratch shares vmcnt)
    const rsrc_t rdw = mk_rsrc(a.dww, (unsigned)(4 * E * 9) * 4u), rfw = mk_rsrc(a.fftw, (unsigned)(E * 40) * 4u);
    auto stage_fetch = [&](int ch) {
        // (C = 64: the lane-invariant pieces of the indices below - tid / 9, tid / 40 and their remainders - were hoisted out of the chunk loop and
        //  spilled: 11 scratch reloads per chunk in four dependent groups, each waiting for vmcnt(0).  Behind an opaque copy of tid they are
        //  recomputed per chunk instead, ~30 vector instructions.  Together with the strips' LDS offsets (mfma_phase), wave 3's coordinates and the
        //  one-output-at-a-time recombination of the column phase: 19 -> 0 spilled registers, level 2 1.371 -> 1.327 ms, interleaved, bit-identical
        //  (profiles/r05_l_fused64_ab.txt).  The same treatment of stage_store's two addresses costs 10 %, and C <= 48 keeps the hoisted form:
        //  this change alone cost the C = 32 kernel 2.7 %)
        const int tid_o = tid;
        const int tid = [&] {
            if constexpr (KST >= 4 || (FDN_STAGE_OPAQUE && TAIL == 3)) { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
            else return tid_o;
        }();
        auto tap_of = [&](int i) {                                      // element i < 288: row m = kind * 8 + channel, tap i % 9
            if constexpr (CELLS) {                                      // ... or (channel, tap, kind)
                const int cl = i / 36, rem = i - cl * 36;
                const int ew = ch * FEG + cl;
                return bload(rdw, (unsigned)(((rem & 3) * E + (ew < E ? ew : E - 1)) * 9 + (rem >> 2)) * 4u, 0);
            }
            const int m = i / 9, tap = i - m * 9;
            const int ew = ch * FEG + (m & 7);
            return bload(rdw, (unsigned)(((m >> 3) * E + (ew < E ? ew : E - 1)) * 9 + tap) * 4u, 0);
        };
        st_w = tap_of(tid);
        st_w1 = tid < 32 ? tap_of(tid + 256) : 0.f;
        const int c0 = tid / 40, c1 = (tid + 256) / 40;                 // gains: 320 values
        const int e0_ = ch * FEG + c0, e1_ = ch * FEG + c1;
        st_f0 = bload(rfw, (unsigned)((e0_ < E ? e0_ : E - 1) * 40 + (tid - c0 * 40)) * 4u, 0);
        st_f1 = (tid < 64) ? bload(rfw, (unsigned)((e1_ < E ? e1_ : E - 1) * 40 + (tid + 256 - c1 * 40)) * 4u, 0) : 0.f;
    };
    auto stage_store = [&]() {
        wks[tid] = st_w;
        if (tid < 32) wks[tid + 256] = st_w1;
        fgs[tid] = st_f0;
        if (tid < 64) fgs[tid + 256] = st_f1;
    };
    // (C >= 48: the strips alone take 108 / 144 registers, so the A operands are not held in registers at all - each k-step reads its
    //  three fragments from L1 / L2: 1.43 ms against 1.53-1.56 ms at level 2 for the variants that hold them in registers.  No width spills
    //  since round 5 (C = 64 spilled 20 registers through round 4, profiles/r04_spills.txt))
    // ---- to_hidden of one chunk on the matrix cores: D[32 rows][32 halo pixels] per strip -> LDS planes (0 outside the image: the
    // strip, its statistics and `xone` all read 0 there)
    auto mfma_phase = [&](int ch) __attribute__((always_inline)) {
        const fdn_u32x4* wp_ = reinterpret_cast<const fdn_u32x4*>(a.wpk) + ((long)ch * KS) * 64 + lane;      // (!AW_AHEAD: operands straight from L1 / L2)
        int po[3];
        if constexpr (KST >= 4) {           // (C = 64: the LDS offsets of the strips' pixels are rebuilt here instead of living across the chunk)
            int t = threadIdx.x;
            asm volatile("" : "+v"(t));
#pragma unroll
            for (int si = 0; si < 3; ++si) {
                const int p = ((t >> 6) + 4 * si) * 32 + (t & 31);
                const int r = p / FHW, c = p - r * FHW;
                po[si] = p < FHP ? r * FRS + c : FHW;
            }
        } else {
#pragma unroll
            for (int si = 0; si < 3; ++si) po[si] = pixoff[si];
        }
        // (round 5, measured: issuing the three strips' chains in turn - three accumulators - changes nothing, 2.315 against 2.288 ms: the phase is not
        //  bound by the latency of a dependent chain)
#pragma unroll
        for (int si = 0; si < 3; ++si) {
            if (wave + 4 * si < FNS) {                              // wave-uniform
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    const fdn_u32x4 a3[3] = {AW_AHEAD ? aw[3 * ks] : wp_[(3 * ks) * 64], AW_AHEAD ? aw[3 * ks + 1] : wp_[(3 * ks + 1) * 64],
                                             AW_AHEAD ? aw[3 * ks + 2] : wp_[(3 * ks + 2) * 64]};
                    acc = fdn_mfma_split6(a3, xb[si][ks], acc);
                }
                const bool one = (onebits >> si) & 1u;
                const fdn_u32x4 xone = {one ? 0x3F803F80u : 0u, one ? 0x00003F80u : 0u, 0u, 0u};
                acc = fdn_mfma_bf16(AW_AHEAD ? aw[KS - 1] : wp_[(KS - 1) * 64], xone, acc);          // + bias: b1 + b2 + b3 against 1, 1, 1 (0 outside the image)
                // MFMA row R = (r & 3) + 8 (r >> 2) + 4 kh = 4 * channel + kind (fdn_fdsa_pack): registers 4g .. 4g + 3 are one cell of channel 2g + kh
                if constexpr (CELLS) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        reinterpret_cast<f32x4*>(hid)[(2 * g + kh) * FPL + po[si]] = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) hid[((r & 3) * 8 + 2 * (r >> 2) + kh) * FPL + po[si]] = acc[r];   // plane = kind * 8 + channel
                }
            }
        }
        if (AW_AHEAD && ch + 1 < a.nchunks) aw_fetch(ch + 1);
    };
    if (AW_AHEAD) aw_fetch(0);
    stage_fetch(0);
    stage_store();                      // (visible behind the first barrier of the loop)
    // (measured, tools/ab_libs.py: issuing the MFMA phase of chunk ch + 1 in front of chunk ch's inverse rows - the matrix pipe under
    //  that phase's vector instructions - loses: 2.47 against 2.36 ms at C = 32, 1.46 against 1.42 ms at C = 64; the accumulators then
    //  live across the inverse rows)
    for (int ch = 0; ch < a.nchunks; ++ch) {
        const int e0 = ch * FEG;
        const int e = e0 + el;
        const bool more = ch + 1 < a.nchunks;                       // uniform
        FTR(0)
        mfma_phase(ch);
        if (more) stage_fetch(ch + 1);
        FTR(1)
        __syncthreads();
        FTR(2)
        if constexpr (TAIL != 0 && FDN_RING) {
            // the slot (taken by thread 0 at entry: its round trip ran under the strip loads) is visible behind the first barrier.  Rebuilt per chunk
            // from the LDS word instead of carried through the loop: a loop-carried descriptor cost the chunk loop three spilled registers
            scr_tile = reinterpret_cast<const float*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(ring_s[3]) << 32) |
                                                      (unsigned)__builtin_amdgcn_readfirstlane(ring_s[2]));
            rout = mk_rsrc(scr_tile, 4u * E * 1024u);
        }

        // ---- rows: depthwise 3x3 (to_hidden_dw, FDN_arch.py:578) + forward row transforms of q, k, v (v_value: see the column phase)
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
        if constexpr (CELLS) {
            const f32x4* hc = reinterpret_cast<const f32x4*>(hid) + el * FPL + row * FRS + px * 8;
            f32x4 wk4[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) wk4[i] = reinterpret_cast<const f32x4*>(wks)[el * 9 + i];
            fdn_f32x2 aqk[8], avv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) aqk[j] = fdn_f32x2{0.f, 0.f}, avv[j] = fdn_f32x2{0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                f32x4 v[10];
#pragma unroll
                for (int j = 0; j < 10; ++j) v[j] = hc[dy * FRS + j];
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {                            // (each sum in the order of dw_row8: same bits)
                        aqk[j] = __builtin_elementwise_fma(wk4[dy * 3 + dx].xy, v[j + dx].xy, aqk[j]);
                        avv[j] = __builtin_elementwise_fma(wk4[dy * 3 + dx].zw, v[j + dx].zw, avv[j]);
                    }
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float o8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o8[j] = t == 0 ? aqk[j].x : t == 1 ? aqk[j].y : avv[j].x;
                float2 sp[5];
                rfft8_row(o8, sp);
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) S[(t * NP + slot) * PS + kx * KXS + row] = sp[kx];
            }
            float vv8[8];                                                      // v_value: no transform, straight out
#pragma unroll
            for (int j = 0; j < 8; ++j) vv8[j] = avv[j].y;
            st_store8<OBF>(vv8, rout, e < E ? opix + (unsigned)(3 * E + e) * hwo : OOB, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float o8[8];
                dw_row8(hb + t * 8 * FPL, wks + (t * 8 + el) * 9, o8);
                float2 sp[5];
                rfft8_row(o8, sp);
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) S[(t * NP + slot) * PS + kx * KXS + row] = sp[kx];
            }
        }
        FTR(3)
        __syncthreads();
        FTR(4)

        // ---- columns: thread = (slot, kx): forward, recombine, inverse (as fdsa_core_kernel) ------------------------
        if (tid < NP * 5) {
            const int pj = tid / 5, kx = tid - pj * 5;
            float fg[8];
#pragma unroll
            for (int ky = 0; ky < 8; ++ky) fg[ky] = fgs[(pj >> 2) * 40 + ky * 5 + kx];
            constexpr float sc = 1.0f / 64.0f;   // norm='backward'
            // (round 5) the recombination runs without the four replace_denormals of :593-604 first - only the self-conjugate bins' imaginary
            // parts are replaced, the minimum magnitude of everything else is tracked - and a wave that met a value below 1e-10 evaluates its
            // columns again as written, from the spectra still in LDS (patch_fft.hpp, fdsa_bin: bit-identical; 128 -> 45 instructions per column)
            // C = 64: 144 registers hold the strips, so the recombination keeps (u, |qk|, |qk| / |v|, |v|, v1) per bin - 56 registers where o1, o2,
            // o3 take 48 on top of q, k, v - and the three outputs are formed, transformed and parked one after the other.
            float2 u_[8], v1_[8];
            float qka_[8], g_[8], va_[8];
            auto column = [&](auto as_written) __attribute__((always_inline)) {
                constexpr bool AW = decltype(as_written)::value;
                float2 q[8], k[8], v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    q[i] = S[(0 * NP + pj) * PS + kx * KXS + i];
                    k[i] = S[(1 * NP + pj) * PS + kx * KXS + i];
                    v[i] = S[(2 * NP + pj) * PS + kx * KXS + i];
                }
                fft8<false>(q);
                if (KST >= 4) __builtin_amdgcn_sched_barrier(0);
                fft8<false>(k);
                if (KST >= 4) __builtin_amdgcn_sched_barrier(0);
                fft8<false>(v);
                if (KST >= 4) __builtin_amdgcn_sched_barrier(0);
                float m = 1.0f;
#pragma unroll
                for (int ky = 0; ky < 8; ++ky) {
                    if (ky % 4 == 0) fdsa_bin<AW, true>(q[ky], k[ky], v[ky], fg[ky], m, u_[ky], v1_[ky], qka_[ky], g_[ky], va_[ky]);
                    else fdsa_bin<AW, false>(q[ky], k[ky], v[ky], fg[ky], m, u_[ky], v1_[ky], qka_[ky], g_[ky], va_[ky]);
                    if (KST >= 4) __builtin_amdgcn_sched_barrier(0);     // C = 64: one bin at a time, or the kernel spills
                }
                return m;
            };
#if FDN_RD_VOTE
            // (C = 64 stays as written: the second evaluation path costs 14 spilled registers there, 1.36 -> 1.42 ms)
            if constexpr (KST >= 4) column(std::true_type{});
            else if (__builtin_amdgcn_ballot_w64(column(std::false_type{}) < 1e-10f) != 0) column(std::true_type{});
#else
            column(std::true_type{});
#endif
            if constexpr (KST >= 4) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    float2 o[8];
#pragma unroll
                    for (int ky = 0; ky < 8; ++ky)
                        o[ky] = t == 0 ? make_float2(va_[ky] * u_[ky].x, va_[ky] * u_[ky].y)                 // :609-612
                              : t == 1 ? make_float2(g_[ky] * v1_[ky].x, g_[ky] * v1_[ky].y)                 // :617-619
                                       : make_float2(qka_[ky] * u_[ky].x, qka_[ky] * u_[ky].y);              // :627-629
                    fft8<true>(o);
#pragma unroll
                    for (int i = 0; i < 8; ++i) S[(t * NP + pj) * PS + kx * KXS + i] = make_float2(o[i].x * sc, o[i].y * sc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                float2 o1[8], o2[8], o3[8];
#pragma unroll
                for (int ky = 0; ky < 8; ++ky) {
                    o1[ky] = make_float2(va_[ky] * u_[ky].x, va_[ky] * u_[ky].y);                            // :609-612
                    o2[ky] = make_float2(g_[ky] * v1_[ky].x, g_[ky] * v1_[ky].y);                            // :617-619
                    o3[ky] = make_float2(qka_[ky] * u_[ky].x, qka_[ky] * u_[ky].y);                          // :627-629
                }
                fft8<true>(o1);
                fft8<true>(o2);
                fft8<true>(o3);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    S[(0 * NP + pj) * PS + kx * KXS + i] = make_float2(o1[i].x * sc, o1[i].y * sc);
                    S[(1 * NP + pj) * PS + kx * KXS + i] = make_float2(o2[i].x * sc, o2[i].y * sc);
                    S[(2 * NP + pj) * PS + kx * KXS + i] = make_float2(o3[i].x * sc, o3[i].y * sc);
                }
            }
        } else if (!CELLS && wave == 3) {
            // the 160 column jobs fill waves 0-2: wave 3, idle otherwise, runs the whole chunk's v_value path meanwhile (depthwise
            // conv of the fourth kind straight to global: no transform) - four (channel, patch, row) jobs per lane
            int t3 = threadIdx.x;
            if constexpr (KST >= 4) asm volatile("" : "+v"(t3));       // (C = 64: row / px / kh rebuilt here, or the base address below is spilled)
            const int row3 = t3 & 7, px3 = (t3 >> 3) & 3, kh3 = (t3 >> 5) & 1;
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                const int elv = 2 * i + (KST >= 4 ? kh3 : kh), ev = e0 + elv;
                float o8[8];
                dw_row8(hid + (24 + elv) * FPL + (KST >= 4 ? row3 * FRS + px3 * 8 : row * FRS + px * 8), wks + (24 + elv) * 9, o8);
                st_store8<OBF>(o8, rout, ev < E ? opix + (unsigned)(3 * E + ev) * hwo : OOB, 0);
            }
        }
        FTR(5)
        __syncthreads();
        FTR(6)

        // ---- inverse rows, 32-byte segments straight to global (out1|out2|out3) --------------------------------------
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            float2 xk[5];
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) xk[kx] = S[(t * NP + slot) * PS + kx * KXS + row];
            float r8[8];
            irfft8_row(xk, r8);
            st_store8<OBF>(r8, rout, e < E ? opix + (unsigned)(t * E + e) * hwo : OOB, 0);
        }
        FTR(7)
        if (more) stage_store();        // taps and gains of the next chunk (this chunk's were last read before the third barrier)
        // (measured: an explicit s_waitcnt vmcnt(0) here - what a scratch reload implies - costs 1.8 % at C = 32, 1 % at C = 64: profiles/r05_l_fused64_ab.txt)
        // (the next chunk's row phase rewrites S behind the barrier at the top of the loop, i.e. after every thread has finished these reads)
    }
    if constexpr (TAIL == 1 || TAIL == 3) {
        // ---- the tail, on this workgroup's own planes (fdsa_tail.hpp).  `hid` is dead behind the last chunk's third barrier: the operand image
        // (gamma | beta | transposed project_out) goes global -> LDS directly, no registers; its arrival, and every store of the planes, is what
        // vmcnt(0) waits for; behind the barrier the planes are in L2 (or beyond), visible to the sc1 loads of every wave of this workgroup
        constexpr int SH = 19;
        constexpr bool PIN = TAIL == 3;          // + the next sub-block's project_in (operands behind the tail's own image)
        constexpr int NBLK0 = tl_image_floats(SH, 1) / 256;
        constexpr int NBLK = NBLK0 + (PIN ? tl_pin_floats(3) / 256 : 0);
#ifdef FDN_FUSED_TRACE
        if (trc) trc[40] = __builtin_amdgcn_s_memtime();
#endif
        const rsrc_t rtw = mk_rsrc(a.tw, (unsigned)NBLK * 1024u);
#pragma unroll
        for (int i = 0; i < (NBLK + 3) / 4; ++i) {
            const int blk = __builtin_amdgcn_readfirstlane(wave + 4 * i);
            if (blk < NBLK)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rtw, (__attribute__((address_space(3))) void*)(hid + blk * 256), 16, (unsigned)(blk * 1024 + lane * 16), 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef FDN_FUSED_TRACE
        if (trc) trc[41] = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef FDN_FUSED_TRACE
        if (trc) trc[42] = __builtin_amdgcn_s_memtime();
#endif
        TailIo io;
        io.scr = scr_tile;
        io.res = a.res ? a.res + (long)b * a.N * P : nullptr;
        io.y = a.y + (long)b * a.N * P;
        io.stats_out = a.stats_out ? a.stats_out + (long)b * 2 * P : nullptr;
        io.E = E; io.N = a.N; io.W = W; io.ty0 = ty0; io.tx0 = tx0; io.P = P;
        io.h = PIN ? reinterpret_cast<float*>(reinterpret_cast<char*>(a.h) + (long)b * a.Hd * P * (a.h_bf16 ? 2 : 4)) : nullptr;
        io.Hd = a.Hd;
        io.h_bf16 = a.h_bf16;
        io.ring_flag = FDN_RING ? reinterpret_cast<unsigned*>(a.out) + ring_s[0] : nullptr;
        io.ring_cnt = ring_s + 1;
        // (C = 32 is always the stock E = 38, N = 32: fdn_fdsa_fused_tail checks it; C = 24 keeps the channel-range predicates)
        constexpr bool FULL = C == 32;
#ifdef FDN_FUSED_TRACE
        fdsa_tail_px2<SH, PIN, 3, FULL>(io, hid, hid + NBLK0 * 256, trc);
#else
        fdsa_tail_px2<SH, PIN, 3, FULL>(io, hid, hid + NBLK0 * 256);
#endif
    }
    if constexpr (TAIL == 2 || TAIL == 4) {
        // ---- level 2 (fdsa_tail_px1): gamma -> wks, beta -> fgs, group 0's packed operands -> hid (all dead behind the last chunk's third barrier);
        // group 1's -> S, which the last inverse rows still read: behind the barrier
        constexpr int SH = 38, MT = 2, NQ = (SH + 7) / 8, GB = NQ * MT * 3;           // GB: KB (= wave-level DMA instructions) per group
        constexpr bool PIN2 = TAIL == 4;            // + the next sub-block's project_in (64 -> Hd), operands behind the tail's own image, read from global
        const rsrc_t rtw = mk_rsrc(a.tw, (unsigned)tl_image_floats_px1(SH, MT) * 4u);
        auto dma = [&](const void* dst, int src_blk, int nblk, int first = -1) {          // waves take the KB blocks in turn (first: starting wave)
            for (int i = first < 0 ? wave : (wave - first) & 3; i < nblk; i += 4) {
                const int blk = __builtin_amdgcn_readfirstlane(i);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rtw, (__attribute__((address_space(3))) void*)(reinterpret_cast<const char*>(dst) + blk * 1024), 16,
                                                         (unsigned)((src_blk + blk) * 1024 + lane * 16), 0, 0, 0);
            }
        };
#ifdef FDN_FUSED_TRACE
        if (trc) trc[40] = __builtin_amdgcn_s_memtime();
#endif
        dma(wks, 0, 1, 2);               // (waves 2 and 3 carry one block less of group 0's 30)
        dma(fgs, 1, 1, 3);
        dma(hid, 2, GB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef FDN_FUSED_TRACE
        if (trc) trc[41] = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef FDN_FUSED_TRACE
        if (trc) trc[42] = __builtin_amdgcn_s_memtime();
#endif
        dma(S, 2 + GB, GB);
        TailIo io;
        io.scr = scr_tile;
        io.res = a.res ? a.res + (long)b * a.N * P : nullptr;
        io.y = a.y + (long)b * a.N * P;
        io.stats_out = a.stats_out ? a.stats_out + (long)b * 2 * P : nullptr;
        io.E = E; io.N = a.N; io.W = W; io.ty0 = ty0; io.tx0 = tx0; io.P = P;
        io.h = PIN2 ? a.h + (long)b * a.Hd * P : nullptr;
        io.Hd = a.Hd;
        io.h_bf16 = 0;
        io.ring_flag = FDN_RING ? reinterpret_cast<unsigned*>(a.out) + ring_s[0] : nullptr;
        io.ring_cnt = ring_s + 1;
#ifdef FDN_FUSED_TRACE
        fdsa_tail_px1<SH, MT, C == 64, PIN2>(io, (lds_cf)wks, (lds_cf)fgs, (lds_cu4)hid, (lds_cu4)S, a.tw, trc);
#else
        fdsa_tail_px1<SH, MT, C == 64, PIN2>(io, (lds_cf)wks, (lds_cf)fgs, (lds_cu4)hid, (lds_cu4)S, a.tw);      // (C = 64 is always the stock E = 76: checked by the launcher)
#endif
    }
}

// fdn_fdsa_tail_pack: project_out [N][3E] + gamma3 / beta3 [3E] -> the tail's LDS image (fdsa_tail.hpp: gamma [3][E2] | beta [3][E2] | Wl [3][E2][WS])
__global__ void fdsa_tail_pack_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ img,
                                      int E, int N, int SH, int MT, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int E2 = 2 * SH, WS = MT * 32 + 1;
    float v = 0.f;
    if (i < 6 * E2) {
        const int j = i < 3 * E2 ? i : i - 3 * E2;
        const int g = j / E2, e = j - g * E2;
        if (e < E) v = (i < 3 * E2 ? gamma : beta)[g * E + e];
    } else if (i < 6 * E2 + 3 * E2 * WS) {
        const int j = i - 6 * E2;
        const int k = j / WS, n = j - k * WS;
        const int g = k / E2, e = k - g * E2;
        if (n < N && e < E) v = w[(long)n * 3 * E + g * E + e];
    }
    img[i] = v;
}

// fdn_fdsa_pack: [4E][C] weights (+ LayerNorm gamma / beta of the input) -> per (chunk, slot, lane) 16-byte A operands of
// v_mfma_f32_32x32x16_bf16: slot 3 ks + part = the part-th bf16 part of w[row][16 ks + 8 kh + j] * gamma, j = 0..7; the last
// slot carries the three parts of the bias row (W beta, fp64 sum) on k = 0, 1, 2 of the lower lane half
__global__ void fdsa_pack_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 fdn_u32x4* __restrict__ wpk, int C, int E, int nchunks, int channel_major) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int KST = (C + 15) / 16, KS = KST * 3 + 1;
    if (idx >= nchunks * KS * 64) return;
    const int lane = idx & 63, j = (idx >> 6) % KS, ch = (idx >> 6) / KS;
    const int m = lane & 31, kh = lane >> 5;
    // MFMA row m of a chunk: kind * 8 + channel (fdn_fdsa_full), or channel * 4 + kind (fdn_fdsa_fused: an accumulator register quad is one
    // (q, k, v, v_value) cell)
    const int e = ch * FEG + (channel_major ? m >> 2 : m & 7), kind = channel_major ? m & 3 : m >> 3;
    auto part_of = [](float x, int part) {
        for (int p = 0; p < part; ++p) x -= __uint_as_float(__float_as_uint(x) & 0xffff0000u);
        return __float_as_uint(x) >> 16;
    };
    fdn_u32x4 o = {0u, 0u, 0u, 0u};
    if (e < E) {
        const float* wr = w + (long)(kind * E + e) * C;
        if (j < KS - 1) {
            const int ks = j / 3, part = j - 3 * ks;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned hl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k = 16 * ks + 8 * kh + 2 * q + u;
                    float v = k < C ? wr[k] : 0.f;
                    if (gamma && k < C) v *= gamma[k];
                    hl[u] = part_of(v, part);
                }
                o[q] = hl[0] | (hl[1] << 16);
            }
        } else if (kh == 0 && beta) {
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

extern "C" int fdn_fdsa_core(const float* hidden, const float* dw_w, const float* fft_w, float* out, int B, int E, int H,
                             int W, fdn_stream_t stream) {
    FDN_CHECK_ARG(hidden && dw_w && fft_w && out && B > 0 && E > 0 && H > 0 && W > 0);
    FDN_CHECK_ARG(H % 8 == 0 && W % 8 == 0 && E < 65536 && B < 65536);
    FDN_CHECK_ARG((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    FDN_CHECK_ARG(16ull * E * H * W < 0x80000000ull);          // one image's 4E planes are addressed with 32-bit byte offsets
    const int tx = cdiv(W, TW), ty = cdiv(H, TH);
#ifndef FDN_CORE_EPB
#define FDN_CORE_EPB 0          // 0: chosen per launch
#endif
    int EPB = FDN_CORE_EPB;
    if (EPB <= 0) {
        const int cus = fdn_device_cus();
        if (cus <= 0) return FDN_ERR_LAUNCH;
        const long want = 8L * 3 * cus, per = (long)tx * ty * B;            // ~8 rounds of three workgroups per CU
        long groups = (want + per - 1) / per;
        groups = groups < 1 ? 1 : (groups > E ? E : groups);
        EPB = (int)((E + groups - 1) / groups);
        if (EPB > 8) EPB = 8;
    }
    const int ngrp = (E + EPB - 1) / EPB;
    if (W % 4 == 0 && (reinterpret_cast<uintptr_t>(hidden) & 15) == 0)
        hipLaunchKernelGGL(fdsa_core_kernel<true>, dim3((unsigned)(tx * ty) * ngrp * B), dim3(256), 0, static_cast<hipStream_t>(stream), hidden, dw_w,
                           fft_w, out, E, H, W, tx, tx * ty, EPB);
    else
        hipLaunchKernelGGL(fdsa_core_kernel<false>, dim3((unsigned)(tx * ty) * ngrp * B), dim3(256), 0, static_cast<hipStream_t>(stream), hidden, dw_w,
                           fft_w, out, E, H, W, tx, tx * ty, EPB);
    return fdn_launch_status();
}

extern "C" int fdn_fdffn_mid(const void* x_, const float* w0, const float* w2, const float* ffta, const float* fftp,
                             void* out_, int B, int Hd, int H, int W, int x_bf16, int out_bf16, fdn_stream_t stream) {
    const float* x = static_cast<const float*>(x_);
    float* out = static_cast<float*>(out_);
    FDN_CHECK_ARG(x && w0 && w2 && ffta && fftp && out && B > 0 && Hd > 0 && H > 0 && W > 0);
    FDN_CHECK_ARG(H % 8 == 0 && W % 8 == 0 && Hd < 65536 && B < 65536);
    FDN_CHECK_ARG((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    FDN_CHECK_ARG(4ull * Hd * H * W < 0x80000000ull);          // one image's Hd planes are addressed with 32-bit byte offsets
    const int tx = cdiv(W, TW), ty = cdiv(H, TH);
    // (the 16-byte-lane halo fetch, V4 = true, measured slower here - 1.95 vs 1.80 ms at level 1: this kernel is bound by
    // VALU issue, and the float4 stash costs four LDS writes per load)
    const int nt = tx * ty;
    int CPB = CPB_MIN;
    {
        const int cus = fdn_device_cus();
        if (cus <= 0) return FDN_ERR_LAUNCH;
        const long want = 8L * 3 * cus;                                    // ~8 rounds of three workgroups per CU
        long groups = (want + (long)nt * B - 1) / ((long)nt * B);         // channel groups per (image, tile) that reach it
        const long gmin = (Hd + CPB_MAX - 1) / CPB_MAX, gmax = (Hd + CPB_MIN - 1) / CPB_MIN;
        groups = groups < gmin ? gmin : (groups > gmax ? gmax : groups);
        CPB = (int)((Hd + groups - 1) / groups);
    }
    const dim3 grid((unsigned)nt * ((Hd + CPB - 1) / CPB) * B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // (a two-channels-per-thread packed-fp32 form measured slower - occupancy - see tools/experiments/fdffn_mid_pair_kernel.hip.inc)
    if (x_bf16 && out_bf16) hipLaunchKernelGGL((fdffn_mid_kernel<false, true, true>), grid, dim3(256), 0, s, x, w0, w2, ffta, fftp, out, Hd, H, W, tx, nt, CPB);
    else if (x_bf16) hipLaunchKernelGGL((fdffn_mid_kernel<false, true, false>), grid, dim3(256), 0, s, x, w0, w2, ffta, fftp, out, Hd, H, W, tx, nt, CPB);
    else if (out_bf16) hipLaunchKernelGGL((fdffn_mid_kernel<false, false, true>), grid, dim3(256), 0, s, x, w0, w2, ffta, fftp, out, Hd, H, W, tx, nt, CPB);
    else hipLaunchKernelGGL((fdffn_mid_kernel<false, false, false>), grid, dim3(256), 0, s, x, w0, w2, ffta, fftp, out, Hd, H, W, tx, nt, CPB);
    return fdn_launch_status();
}

// the operand image for nch8 >= ceil(E / 8) chunks of 8 channels (chunks past E are zeros): fdn_fdsa_full walks 16-channel chunks at C > 32
int fdn_fdsa_pack_chunks(const float* w, const float* gamma, const float* beta, float* wpk, int C, int E, int nch8, hipStream_t s, int channel_major) {
    FDN_CHECK_ARG(w && wpk && C > 0 && C % 2 == 0 && E > 0 && (!gamma == !beta) && nch8 * FEG >= E);
    const int total = nch8 * (((C + 15) / 16) * 3 + 1) * 64;
    hipLaunchKernelGGL(fdsa_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w, gamma, beta, reinterpret_cast<fdn_u32x4*>(wpk), C, E, nch8, channel_major);
    return fdn_launch_status();
}

extern "C" int fdn_fdsa_pack(const float* w, const float* gamma, const float* beta, float* wpk, int C, int E, fdn_stream_t stream) {
    return fdn_fdsa_pack_chunks(w, gamma, beta, wpk, C, E, (E + FEG - 1) / FEG, static_cast<hipStream_t>(stream), 1);
}

extern "C" int fdn_fdsa_fused(const float* x, long xbs, const float* stats, const float* wpk, const float* dw_w, const float* fft_w,
                              void* out_, int B, int C, int E, int H, int W, int out_bf16, fdn_stream_t stream) {
    float* out = static_cast<float*>(out_);
    FDN_CHECK_ARG(x && wpk && dw_w && fft_w && out && B > 0 && E > 0 && H > 0 && W > 0);
    FDN_CHECK_ARG(H % 8 == 0 && W % 8 == 0);
    FDN_CHECK_ARG((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    FDN_CHECK_ARG(16ull * E * H * W < 0x80000000ull && 4ull * C * H * W < 0x80000000ull);   // 32-bit byte offsets per image
    if (fdn_matrix_pipe_f32()) return FDN_ERR_UNSUPPORTED;      // (diagnostic switch: the caller takes fdn_conv1x1 + fdn_fdsa_core)
    FusedArgs a;
    a.x = x; a.xbs = xbs; a.stats = stats; a.wpk = wpk; a.dww = dw_w; a.fftw = fft_w; a.out = out;
    a.E = E; a.H = H; a.W = W;
    a.tiles_x = cdiv(W, FT_W);
    a.tiles_per_img = a.tiles_x * (H / FT_H);
    a.nchunks = (E + FEG - 1) / FEG;
    const long total = (long)B * a.tiles_per_img;
    FDN_CHECK_ARG(total < 0x7fffffffL);
    const dim3 grid((unsigned)total), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define FDN_FUSED_CASE(KERN, CC)                                                                      \
    case CC:                                                                                          \
        if (stats && out_bf16) hipLaunchKernelGGL((KERN<CC, true, true>), grid, block, 0, s, a);      \
        else if (stats) hipLaunchKernelGGL((KERN<CC, true, false>), grid, block, 0, s, a);            \
        else if (out_bf16) hipLaunchKernelGGL((KERN<CC, false, true>), grid, block, 0, s, a);         \
        else hipLaunchKernelGGL((KERN<CC, false, false>), grid, block, 0, s, a);                      \
        break;
    switch (C) {
        FDN_FUSED_CASE(fdsa_fused_kernel, 24)
        FDN_FUSED_CASE(fdsa_fused_kernel, 32)
        FDN_FUSED_CASE(fdsa_fused_kernel, 48)
        FDN_FUSED_CASE(fdsa_fused_kernel, 64)
        default: return FDN_ERR_UNSUPPORTED;
    }
#undef FDN_FUSED_CASE
    fdn_note_bf16_launch();
    return fdn_launch_status();
}

// project_in's operands behind the level-1 image (fdsa_tail_px2<.., PIN>): [NT][2 k-steps][3 parts][64 lanes] 16-byte A operands of the LayerNorm-folded
// weights wf [Hd][C] - lane (n, kh) holds the part-th bf16 part of wf[32 t + n][16 ks + 8 kh + 0..7] - then NT * 32 folded bias values, padded to one KB
static __global__ void fdsa_tail_pack_pin_kernel(const float* __restrict__ wf, const float* __restrict__ bf, float* __restrict__ img, int C, int Hd, int NT, int NKS) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nop = NT * NKS * 3 * 64;
    if (i < nop) {
        const int l = i & 63, part = (i >> 6) % 3, ks = (i / 192) % NKS, t = i / (192 * NKS);
        const int n = t * 32 + (l & 31), kh = l >> 5;
        fdn_u32x4 o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned hl[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = 16 * ks + 8 * kh + 2 * d + h;
                float x = (n < Hd && k < C) ? wf[(long)n * C + k] : 0.f;
                for (int pp = 0; pp < part; ++pp) x -= fdn_trunc_bf16(x);
                hl[h] = __float_as_uint(x) >> 16;
            }
            o[d] = hl[0] | (hl[1] << 16);
        }
        reinterpret_cast<fdn_u32x4*>(img)[i] = o;
    } else if (i < nop + 256) {
        const int j = i - nop;
        img[nop * 4 + j] = (bf && j < Hd && j < NT * 32) ? bf[j] : 0.f;
    }
}

// the level-2 image: [gamma 3 E2 | pad][beta 3 E2 | pad][Wp [3][NQ][MT][part][64] A operands of v_mfma_f32_32x32x16_bf16]: lane l = (n = mt 32 + (l & 31), k2 = l >> 5)
// holds channels e = 2 (8 q + 2 d + h) + k2, d = 0..3, h = 0..1, cut into the part-th bf16 part (fdsa_out_vec_kernel's in-kernel packing, done once here)
static __global__ void fdsa_tail_pack_px1_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ img,
                                          int E, int N, int SH, int MT) {
    const int E2 = 2 * SH, NQ = (SH + 7) / 8;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 512) {
        const int j = i & 255;
        const int g = j / E2, e = j - g * E2;
        img[i] = (j < 3 * E2 && e < E) ? (i < 256 ? gamma : beta)[g * E + e] : 0.f;
        return;
    }
    const int u = i - 512;
    if (u >= 3 * NQ * MT * 3 * 64) return;
    const int l = u & 63, part = (u >> 6) % 3, mt = (u / 192) % MT, q = (u / (192 * MT)) % NQ, g = u / (192 * MT * NQ);
    const int n = mt * 32 + (l & 31), k2 = l >> 5;
    fdn_u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned hl[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int s_ = 8 * q + 2 * d + h, e = 2 * s_ + k2;
            float x = (n < N && s_ < SH && e < E) ? w[(long)n * 3 * E + g * E + e] : 0.f;
            for (int pp = 0; pp < part; ++pp) x -= fdn_trunc_bf16(x);
            hl[h] = __float_as_uint(x) >> 16;
        }
        o[d] = hl[0] | (hl[1] << 16);
    }
    reinterpret_cast<fdn_u32x4*>(img + 512)[u] = o;
}

// ---- the FDSA sub-block in one launch: fdn_fdsa_fused with fdn_fdsa_out's arithmetic run by the producing workgroup (fdsa_tail.hpp) ----
static int fdsa_tail_form(int C, int E, int N, int* sh, int* mt) {      // 0 = no in-kernel tail for this width
    if ((C == 24 || C == 32) && E <= 38 && N <= 32) { *sh = 19; *mt = 1; return 1; }
    if ((C == 48 || C == 64) && E > 38 && E <= 76 && N > 32 && N <= 64) { *sh = 38; *mt = 2; return 2; }
    return 0;
}
static bool fdsa_tail_pin_ok(int form, int C, int Hd) {      // fdn_conv1x1's strip<2> shapes at level 1, strip<4> at C = 64
    return Hd > 0 && 2 * Hd >= 5 * C && ((form == 1 && Hd <= 96) || (form == 2 && C == 64 && Hd <= 192));
}
extern "C" long fdn_fdsa_tail_pack_floats(int C, int E, int N, int Hd) {
    int sh, mt;
    const int form = fdsa_tail_form(C, E, N, &sh, &mt);
    if (Hd > 0 && !fdsa_tail_pin_ok(form, C, Hd)) return 0;
    return form == 1 ? tl_image_floats(sh, mt) + (Hd > 0 ? tl_pin_floats(3) : 0) : form == 2 ? tl_image_floats_px1(sh, mt) + (Hd > 0 ? tl_pin_floats_l2(6) : 0) : 0;
}
extern "C" long fdn_fdsa_scratch_floats(int B, int E, int H, int W) {
    if (B <= 0 || E <= 0 || H <= 0 || W <= 0 || H % 8) return 0;
    if (FDN_RING) return FDN_RING_HDR + (long)FDN_RING_SLOTS * 4 * E * 256;             // independent of the image: per resident workgroup
    return (long)B * (H / FT_H) * cdiv(W, FT_W) * 4 * E * 256;
}
extern "C" int fdn_fdsa_tail_pack(const float* w, const float* gamma3, const float* beta3, const float* pin_w, const float* pin_b, float* img, int C,
                                  int E, int N, int Hd, fdn_stream_t stream) {
    FDN_CHECK_ARG(w && gamma3 && beta3 && img && E > 0 && N > 0 && (Hd == 0) == (pin_w == nullptr));
    int sh, mt;
    const int form = fdsa_tail_form(C, E, N, &sh, &mt);
    if (!form || (Hd > 0 && !fdsa_tail_pin_ok(form, C, Hd))) return FDN_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (form == 2) {
        const int nthreads = 512 + (tl_image_floats_px1(sh, mt) - 512) / 4;     // 512 header floats + one thread per 16-byte operand
        hipLaunchKernelGGL(fdsa_tail_pack_px1_kernel, dim3(cdiv(nthreads, 256)), dim3(256), 0, s, w, gamma3, beta3, img, E, N, sh, mt);
        if (Hd > 0)
            hipLaunchKernelGGL(fdsa_tail_pack_pin_kernel, dim3(cdiv(6 * 4 * 3 * 64 + 256, 256)), dim3(256), 0, s, pin_w, pin_b, img + tl_image_floats_px1(sh, mt), C, Hd, 6, 4);
        return fdn_launch_status();
    }
    const int total = tl_image_floats(sh, mt);
    hipLaunchKernelGGL(fdsa_tail_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w, gamma3, beta3, img, E, N, sh, mt, total);
    if (Hd > 0)
        hipLaunchKernelGGL(fdsa_tail_pack_pin_kernel, dim3(cdiv(3 * 2 * 3 * 64 + 256, 256)), dim3(256), 0, s, pin_w, pin_b, img + total, C, Hd, 3, 2);
    return fdn_launch_status();
}
extern "C" int fdn_fdsa_fused_tail(const float* x, long xbs, const float* stats, const float* wpk, const float* dw_w, const float* fft_w,
                                   const float* tail_img, const float* res, float* out, float* stats_out, float* scratch, void* h_out_, int B,
                                   int C, int E, int H, int W, int Hd, int h_bf16, fdn_stream_t stream) {
    float* h_out = static_cast<float*>(h_out_);
    FDN_CHECK_ARG(x && wpk && dw_w && fft_w && tail_img && out && scratch && B > 0 && E > 0 && H > 0 && W > 0);
    FDN_CHECK_ARG(H % 8 == 0 && W % 8 == 0);
    FDN_CHECK_ARG(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(stats_out) |
                    reinterpret_cast<uintptr_t>(scratch) | reinterpret_cast<uintptr_t>(tail_img)) & 15) == 0);
    FDN_CHECK_ARG(out != x);                                                     // the result must not overwrite x: neighbouring tiles read its halo
    FDN_CHECK_ARG(4ull * (C + 40) * H * W < 0x80000000ull);                      // 32-bit byte offsets per image (x, res / out rows n + 4 kh)
    if (fdn_matrix_pipe_f32()) return FDN_ERR_UNSUPPORTED;
    int sh, mt;
    const int form = fdsa_tail_form(C, E, C, &sh, &mt);
    if (!form || (form == 1 && W % 2)) return FDN_ERR_UNSUPPORTED;
    if ((C == 32 && E != 38) || (C == 64 && E != 76)) return FDN_ERR_UNSUPPORTED;                          // (the C = 32 tail is compiled for the stock E = int(1.2 C) = 38: no channel-range predicates)
    if (form == 2 && !fdn_matrix_pipe_wide()) return FDN_ERR_UNSUPPORTED;       // fdn_set_matrix_pipe(2): the level-2 tail keeps its fp32-MFMA form (fdn_fdsa_out)
    FDN_CHECK_ARG((Hd == 0) == (h_out == nullptr) && (reinterpret_cast<uintptr_t>(h_out) & 15) == 0);
    if (Hd > 0 && !fdsa_tail_pin_ok(form, C, Hd)) return FDN_ERR_UNSUPPORTED;
    if (h_bf16 && (form != 1 || Hd <= 0)) return FDN_ERR_UNSUPPORTED;           // bf16 h: the level-1 form only
    FDN_CHECK_ARG(4ull * (Hd + 40) * H * W < 0x80000000ull);
    FusedArgs a;
    a.x = x; a.xbs = xbs; a.stats = stats; a.wpk = wpk; a.dww = dw_w; a.fftw = fft_w; a.out = scratch;
    a.E = E; a.H = H; a.W = W;
    a.tiles_x = cdiv(W, FT_W);
    a.tiles_per_img = a.tiles_x * (H / FT_H);
    a.nchunks = (E + FEG - 1) / FEG;
    a.tw = tail_img; a.res = res; a.y = out; a.stats_out = stats_out; a.N = C;
    a.h = h_out; a.Hd = Hd; a.h_bf16 = h_bf16;
    const long total = (long)B * a.tiles_per_img;
    FDN_CHECK_ARG(total < 0x7fffffffL);
    const dim3 grid((unsigned)total), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define FDN_FUSED_TAIL_CASE(CC, TT)                                                                          \
    case CC:                                                                                                \
        if (stats) hipLaunchKernelGGL((fdsa_fused_kernel<CC, true, false, TT>), grid, block, 0, s, a);      \
        else hipLaunchKernelGGL((fdsa_fused_kernel<CC, false, false, TT>), grid, block, 0, s, a);           \
        break;
    switch (C) {
        FDN_FUSED_TAIL_CASE(48, 2)
        case 64:
            if (Hd > 0) {
                switch (C) { FDN_FUSED_TAIL_CASE(64, 4) }
            } else {
                switch (C) { FDN_FUSED_TAIL_CASE(64, 2) }
            }
            break;
        case 24:
        case 32:
            if (Hd > 0) {
                switch (C) {
                    FDN_FUSED_TAIL_CASE(24, 3)
                    FDN_FUSED_TAIL_CASE(32, 3)
                }
            } else {
                switch (C) {
                    FDN_FUSED_TAIL_CASE(24, 1)
                    FDN_FUSED_TAIL_CASE(32, 1)
                }
            }
            break;
        default: return FDN_ERR_UNSUPPORTED;
    }
#undef FDN_FUSED_TAIL_CASE
    fdn_note_bf16_launch();
    return fdn_launch_status();
}

#ifdef FDN_MID_TRACE
extern "C" int fdn_debug_mid_trace(void* host, long bytes, int clear) {          // trace builds only (tools/tail_trace.py); not part of the ABI
    static unsigned long long z[MT_NWG * 4 * 16];
    if (bytes > (long)sizeof(z)) return FDN_ERR_ARG;
    if (clear) return hipMemcpyToSymbol(HIP_SYMBOL(g_mid_trace), z, sizeof(z), 0, hipMemcpyHostToDevice) == hipSuccess ? FDN_OK : FDN_ERR_LAUNCH;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mid_trace), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? FDN_OK : FDN_ERR_LAUNCH;
}
#endif
#ifdef FDN_FUSED_TRACE
extern "C" int fdn_debug_fused_trace(void* host, long bytes) {          // trace builds only (tools/fused_trace.py); not part of the ABI
    if (bytes > (long)sizeof(unsigned long long) * FT_NWG * 4 * 64) return FDN_ERR_ARG;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fused_trace), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? FDN_OK : FDN_ERR_LAUNCH;
}
extern "C" int fdn_debug_fused_trace_clear(void) {
    static unsigned long long z[FT_NWG * 4 * 64];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fused_trace), z, sizeof(z), 0, hipMemcpyHostToDevice) == hipSuccess ? FDN_OK : FDN_ERR_LAUNCH;
}
#endif
