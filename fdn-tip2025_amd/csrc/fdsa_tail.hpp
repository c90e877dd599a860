// FDSA tail INSIDE fdn_fdsa_fused's workgroup (round 6; FDN_arch.py:633-639 + the residual of :671): the workgroup that has just produced
// all 4E planes of its 8 x 32 tile wrote them tile-contiguous ([plane][8][32] floats: whole 1-KB rows instead of 32-byte image segments),
// every wave drained its stores (s_waitcnt vmcnt(0)), the workgroup met at a barrier - and now the SAME workgroup reads them back (L2 /
// Infinity Cache: `sc1` loads, this CU's L1 is bypassed) and runs fdn_fdsa_out's arithmetic on them: three LayerNorms over E from registers,
// x v_value, project_out, + residual, the next LayerNorm's statistics.  No second launch, no read of the hand-off from HBM; nothing crosses
// workgroups, so no agent-scope fence is involved (MI355X_MICROARCH.md, inter-workgroup visibility: not needed for same-CU data behind a
// drained store + barrier when the loads go to L2).
//
// The arithmetic is fdsa_out_vec_kernel's, operation for operation (fdsa_out.hip: the same sums in the same order, the same MFMA chains), so
// the result equals the two-launch route bit for bit; only the pixel -> lane map differs (per-pixel arithmetic does not depend on it).
#pragma once
#include "common.hpp"

namespace {

typedef __amdgpu_buffer_rsrc_t trsrc_t;
__device__ __forceinline__ trsrc_t tl_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
constexpr int TL_SC1 = 16;                 // aux bit 4 = sc1 on gfx940+: served by L2, this CU's L1 bypassed
__device__ __forceinline__ fdn_f32x2 tl_load2(trsrc_t r, unsigned voff, unsigned soff) {
    const fdn_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, TL_SC1);
    return fdn_f32x2{__uint_as_float(u.x), __uint_as_float(u.y)};
}
__device__ __forceinline__ float tl_load1(trsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, TL_SC1));
}
__device__ __forceinline__ fdn_f32x2 tl_xsum32(fdn_f32x2 v) { return fdn_f32x2{v.x + __shfl_xor(v.x, 32), v.y + __shfl_xor(v.y, 32)}; }
__device__ __forceinline__ float tl_xsum32(float v) { return v + __shfl_xor(v, 32); }
__device__ __forceinline__ fdn_f32x2 tl_rsqrt_eps(fdn_f32x2 v) { return fdn_f32x2{1.0f / sqrtf(v.x + 1e-5f), 1.0f / sqrtf(v.y + 1e-5f)}; }
__device__ __forceinline__ float tl_rsqrt_eps(float v) { return 1.0f / sqrtf(v + 1e-5f); }

// operand image of the tail in LDS / global (fdn_fdsa_tail_pack): gamma [3][E2] | beta [3][E2] | Wl [3][E2][WS], E2 = 2 ceil(E / 2), WS = 32 MT + 1,
// padded to whole KB (one LDS-DMA wave instruction moves 64 lanes x 16 bytes)
__host__ __device__ constexpr int tl_image_floats(int SH, int MT) { return ((6 * 2 * SH + 3 * 2 * SH * (MT * 32 + 1) + 255) / 256) * 256; }

struct TailIo {
    const float* scr;        // this tile's planes: [4E][256] floats (pixel = 32 row + col)
    const float* res;        // image base [N][P] or null
    float* y;                // image base [N][P]
    float* stats_out;        // image base [2][P] or null
    int E, N, W, ty0, tx0;
    unsigned P;
};

// Level-1 form (E <= 2 SH <= 38, N <= 32): a lane owns two horizontally adjacent pixels (8-byte lanes), a wave two tile rows; fp32 MFMA.
// Mirrors fdsa_out_vec_kernel<19, 1, true, false, 2, 4, false>.
#ifdef FDN_FUSED_TRACE
#define TLTR(i) if (trc) trc[40 + (i)] = __builtin_amdgcn_s_memtime();       // (tools/fused_trace.py --tail: the tail's stamps sit behind five chunks' worth)
#else
#define TLTR(i)
#endif
template <int SH>
__device__ __forceinline__ void fdsa_tail_px2(const TailIo& io, const float* lds, unsigned long long* trc = nullptr) {
    typedef fdn_f32x2 T;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int E2 = 2 * SH, WS = 33;
    const float* tg = lds;
    const float* tb = lds + 3 * E2;
    const float* Wl = lds + 6 * E2;
    const int E = io.E, N = io.N;
    const unsigned P = io.P, P4 = P * 4u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5, ln = lane & 31;
    const int row = 2 * wave + (ln >> 4), col = 2 * (ln & 15);
    const bool ok = io.tx0 + col < io.W;
    const unsigned pix = (unsigned)((io.ty0 + row) * io.W + io.tx0 + col);
    constexpr unsigned PI = 1024u;                                   // bytes per plane of the tile
    const trsrc_t rg[3] = {tl_rsrc(io.scr, (unsigned)E * PI), tl_rsrc(io.scr + (long)E * 256, (unsigned)E * PI),
                           tl_rsrc(io.scr + (long)2 * E * 256, (unsigned)E * PI)};
    const trsrc_t rv = tl_rsrc(io.scr + (long)3 * E * 256, (unsigned)E * PI);
    const unsigned voff = kh * PI + (unsigned)(row * 32 + col) * 4u;    // channel e = 2s + kh; e >= E reads 0 (outside the descriptor)
    const float invE = 1.0f / (float)E;

    T vv[SH], oa[SH], ob2[SH];
#pragma unroll
    for (int s = 0; s < SH; ++s) {
        vv[s] = tl_load2(rv, voff, (unsigned)(2 * s) * PI);
        oa[s] = tl_load2(rg[0], voff, (unsigned)(2 * s) * PI);
    }
    const unsigned nb4 = (unsigned)N * P4;
    const trsrc_t ro = tl_rsrc(io.y, nb4);
    const trsrc_t rr = tl_rsrc(io.res ? io.res : io.y, io.res ? nb4 : 0u);
    const unsigned vo = ok ? (4u * kh * P + pix) * 4u : 0x80000000u;
    T rres[16];
    f32x16 acc[2];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;

#pragma unroll
    for (int g = 0; g < 3; ++g) {
        T* cur = (g & 1) ? ob2 : oa;
        if (g < 2) {
            T* nxt = (g & 1) ? oa : ob2;
#pragma unroll
            for (int s = 0; s < SH; ++s) nxt[s] = tl_load2(rg[g + 1], voff, (unsigned)(2 * s) * PI);
        }
        T m = 0.f;
#pragma unroll
        for (int s = 0; s < SH; ++s) m += cur[s];
        m = tl_xsum32(m) * invE;
        T q = 0.f;
#pragma unroll
        for (int s = 0; s < SH; ++s) {
            const T dl = cur[s] - m;
            q += (2 * s + kh < E) ? dl * dl : T(0.f);
        }
        const T rs = tl_rsqrt_eps(tl_xsum32(q) * invE);
#pragma unroll
        for (int s = 0; s < SH; ++s) {
            asm volatile("" ::: "memory");
            const int e = 2 * s + kh;
            cur[s] = ((cur[s] - m) * rs * tg[g * E2 + e] + tb[g * E2 + e]) * vv[s];      // norm_g(out_g) * v_value  :633-638
        }
        if (g == 0) { TLTR(3) }
        if (g == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const fdn_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(rr, vo, (unsigned)((r & 3) + 8 * (r >> 2)) * P4, 0);      // 0 without a residual
                rres[r] = T{__uint_as_float(u.x), __uint_as_float(u.y)};
            }
        }
#pragma unroll
        for (int s = 0; s < SH; ++s) {
            const float wa = Wl[(g * E2 + 2 * s + kh) * WS + ln];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, cur[s].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, cur[s].y, acc[1], 0, 0, 0);
        }
        if (g == 0) { TLTR(4) }
        if (g == 2) { TLTR(5) }
    }
    // ---- epilogue: residual, store, next LayerNorm's statistics (fdsa_out_vec_kernel's) -------------------------------------
    T outv[16];
    T sm = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int nrow = (r & 3) + 8 * (r >> 2);
        T o = T{acc[0][r], acc[1][r]};
        o += rres[r];
        __builtin_amdgcn_raw_buffer_store_b64(fdn_u32x2{__float_as_uint(o.x), __float_as_uint(o.y)}, ro, vo, (unsigned)nrow * P4, 0);
        outv[r] = (nrow + 4 * kh < N) ? o : T(0.f);
        sm += outv[r];
    }
    if (io.stats_out) {
        T sq = 0.f;
        const T mean = tl_xsum32(sm) / (float)N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const T dl = outv[r] - mean;
            sq += ((r & 3) + 8 * (r >> 2) + 4 * kh < N) ? dl * dl : T(0.f);
        }
        const T rstd = tl_rsqrt_eps(tl_xsum32(sq) / (float)N);
        if (kh == 0) {
            const trsrc_t rs_ = tl_rsrc(io.stats_out, 2u * P4);
            const unsigned vs = ok ? pix * 4u : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b64(fdn_u32x2{__float_as_uint(mean.x), __float_as_uint(mean.y)}, rs_, vs, 0u, 0);
            __builtin_amdgcn_raw_buffer_store_b64(fdn_u32x2{__float_as_uint(rstd.x), __float_as_uint(rstd.y)}, rs_, vs, P4, 0);
        }
    }
    TLTR(6)
}

}  // namespace
