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

// (dl * dl rounded, then added: fdsa_out_vec_kernel's select between the product and the sum keeps the two apart; without the select - FULL: every
// channel of the group exists - the compiler would contract them into one fma and the statistics would round differently)
template <bool FULL, typename T>
__device__ __forceinline__ T tl_sq_acc(T q, T dl, bool live) {
#pragma clang fp contract(off)
    const T p = dl * dl;
    if constexpr (FULL) return q + p;
    else return q + (live ? p : T(0.f));
}

// operand image of the tail in LDS / global (fdn_fdsa_tail_pack): gamma [3][E2] | beta [3][E2] | Wl [3][E2][WS], E2 = 2 ceil(E / 2), WS = 32 MT + 1,
// padded to whole KB (one LDS-DMA wave instruction moves 64 lanes x 16 bytes)
__host__ __device__ constexpr int tl_image_floats(int SH, int MT) { return ((6 * 2 * SH + 3 * 2 * SH * (MT * 32 + 1) + 255) / 256) * 256; }

__host__ __device__ constexpr int tl_image_floats_px1_c(int SH, int MT) { return 512 + 3 * ((SH + 7) / 8) * MT * 3 * 64 * 4; }      // level-2 image without project_in
struct TailIo {
    const float* scr;        // this tile's planes: [4E][256] floats (pixel = 32 row + col)
    const float* res;        // image base [N][P] or null
    float* y;                // image base [N][P]
    float* stats_out;        // image base [2][P] or null
    int E, N, W, ty0, tx0;
    unsigned P;
    float* h;                // PIN: image base [Hd][P] of the NEXT sub-block's project_in output (FDFFN, FDN_arch.py:456), Hd rows
    int Hd;
    int h_bf16;              // PIN: h is stored as bf16 (bf16-storage mode: round-to-nearest-even of the fp32 result, two pixels per dword)
    unsigned* ring_flag;     // ring mode: this workgroup's slot flag (given back when the last wave's read-back has landed) or null
    int* ring_cnt;           // LDS counter of waves whose read-back has landed
};
// the wave whose read-back landed last frees the workgroup's block of the ring (every load of it has returned: the next owner may write)
__device__ __forceinline__ void tl_ring_release(const TailIo& io) {
    if (io.ring_flag && (threadIdx.x & 63) == 0) {
        if (__hip_atomic_fetch_add(io.ring_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 3)
            __hip_atomic_store(io.ring_flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Level-1 form (E <= 2 SH <= 38, N <= 32): a lane owns two horizontally adjacent pixels (8-byte lanes), a wave two tile rows; fp32 MFMA.
// Mirrors fdsa_out_vec_kernel<19, 1, true, false, 2, 4, false>.
#ifdef FDN_FUSED_TRACE
#define TLTR(i) if (trc) trc[40 + (i)] = __builtin_amdgcn_s_memtime();       // (tools/fused_trace.py --tail: the tail's stamps sit behind five chunks' worth)
#else
#define TLTR(i)
#endif
// PIN (round 6): the FDFFN that follows every FDSA starts with project_in(LayerNorm(y)) (FDN_arch.py:456, :673) - per pixel, on exactly the values and
// statistics this epilogue holds in registers.  It runs here as gemm_split_strip_kernel<2, FDN_PRO_LN> does (fdn_conv1x1's kernel for this shape: the
// normalised values cut into three bf16 parts, six products per 16-deep k-step on v_mfma_f32_32x32x16_bf16, bias in the accumulator, same k slots, same
// order: the same bits), and the C-plane read + LayerNorm + split of that launch are gone.  An MFMA result has channels (r & 3) + 8 (r >> 2) + 4 kh in
// register r, its B operand wants channels 16 ks + 8 kh + j: four v_permlane32_swap per 16 channels move the two middle quarters across the lane halves.
// lds_pin: [NT tiles][2 k-steps][3 parts][64 lanes] 16-byte A operands of the LayerNorm-folded weights, then NT * 32 bias floats.
// FULL: E == 2 SH and N == 32 (the stock level 1: E = 38, C = 32) - no channel-range predicates (205 selects + 36 compares of the 1,350 vector instructions)
template <int SH, bool PIN = false, int NT = 3, bool FULL = false>
__device__ __forceinline__ void fdsa_tail_px2(const TailIo& io, const float* lds, const float* lds_pin = nullptr, unsigned long long* trc = nullptr) {
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
            q = tl_sq_acc<FULL>(q, dl, 2 * s + kh < E);
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
            tl_ring_release(io);                 // (the last group's values are in registers: nothing of the block is read any more)
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
        outv[r] = (FULL || nrow + 4 * kh < N) ? o : T(0.f);
        sm += outv[r];
    }
    if (io.stats_out || PIN) {
        T sq = 0.f;
        const T mean = tl_xsum32(sm) / (float)N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const T dl = outv[r] - mean;
            sq = tl_sq_acc<FULL>(sq, dl, (r & 3) + 8 * (r >> 2) + 4 * kh < N);
        }
        const T rstd = tl_rsqrt_eps(tl_xsum32(sq) / (float)N);
        if (io.stats_out && kh == 0) {
            const trsrc_t rs_ = tl_rsrc(io.stats_out, 2u * P4);
            const unsigned vs = ok ? pix * 4u : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b64(fdn_u32x2{__float_as_uint(mean.x), __float_as_uint(mean.y)}, rs_, vs, 0u, 0);
            __builtin_amdgcn_raw_buffer_store_b64(fdn_u32x2{__float_as_uint(rstd.x), __float_as_uint(rstd.y)}, rs_, vs, P4, 0);
        }
        if constexpr (PIN) {
            TLTR(6)
            fdn_u32x4 Bf[2][2][3];                                   // [pixel of the pair][k-step][part]
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float sa = c ? rstd.y : rstd.x, sb = -(c ? mean.y : mean.x) * sa;          // gemm_split: v = fmaf(v, rstd, -mean * rstd)
                float xn[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) xn[r] = fmaf(c ? outv[r].y : outv[r].x, sa, sb);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[8];
#pragma unroll
                    for (int b4 = 0; b4 < 4; ++b4) {
                        // register 8 ks + b4 holds channels 16 ks + b4 (+ 4 on the upper lanes), register 8 ks + 4 + b4 channels 16 ks + 8 + b4 (+ 4):
                        // swap the first one's upper lanes with the second one's lower lanes
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xn[8 * ks + b4]), __float_as_uint(xn[8 * ks + 4 + b4]), false, false);
                        v[b4] = __uint_as_float(sw[0]);
                        v[4 + b4] = __uint_as_float(sw[1]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned p1, p2, p3;
                        fdn_split3(v[2 * j], v[2 * j + 1], p1, p2, p3);
                        Bf[c][ks][0][j] = p1, Bf[c][ks][1][j] = p2, Bf[c][ks][2][j] = p3;
                    }
                }
            }
            const fdn_u32x4* wpin = reinterpret_cast<const fdn_u32x4*>(lds_pin) + lane;
            const float* bpin = lds_pin + NT * 2 * 3 * 64 * 4;
            const trsrc_t rh = tl_rsrc(io.h, (unsigned)io.Hd * P4);
            const unsigned vh = ok ? (4u * kh * P + pix) * 4u : 0x80000000u;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x16 ah[2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(&bpin[t * 32 + 8 * g + 4 * kh]);
#pragma unroll
                    for (int c = 0; c < 2; ++c) ah[c][4 * g] = bv.x, ah[c][4 * g + 1] = bv.y, ah[c][4 * g + 2] = bv.z, ah[c][4 * g + 3] = bv.w;
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const fdn_u32x4 a3[3] = {wpin[((t * 2 + ks) * 3 + 0) * 64], wpin[((t * 2 + ks) * 3 + 1) * 64], wpin[((t * 2 + ks) * 3 + 2) * 64]};
                    ah[0] = fdn_mfma_split6(a3, Bf[0][ks], ah[0]);
                    ah[1] = fdn_mfma_split6(a3, Bf[1][ks], ah[1]);
                }
                if (io.h_bf16) {                                        // (uniform) bf16 storage: the pixel pair is one dword
                    const trsrc_t rhb = tl_rsrc(io.h, (unsigned)io.Hd * P * 2u);
                    const unsigned vhb = ok ? (4u * kh * P + pix) * 2u : 0x80000000u;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(pack_bf16(ah[0][r], ah[1][r]), rhb, vhb, (unsigned)(t * 32 + (r & 3) + 8 * (r >> 2)) * P * 2u, 0);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)                        // rows >= Hd fall outside the descriptor
                        __builtin_amdgcn_raw_buffer_store_b64(fdn_u32x2{__float_as_uint(ah[0][r]), __float_as_uint(ah[1][r])}, rh, vh,
                                                              (unsigned)(t * 32 + (r & 3) + 8 * (r >> 2)) * P4, 0);
                }
            }
        }
    }
    if constexpr (PIN) { TLTR(7) } else { TLTR(6) }
}

// Level-2 form (E <= 2 SH <= 76, N <= 64): one pixel per lane, a wave takes its two tile rows one after the other; project_out on the bf16 matrix pipe
// (three exact bf16 parts per operand, six products: fp32 arithmetic).  Mirrors fdsa_out_vec_kernel<38, 2, false, false, 1, 8, true> - the same sums, the
// same operand cuts, the same MFMA order - so the result equals fdn_fdsa_fused + fdn_fdsa_out bit for bit.
// Operand image (fdn_fdsa_tail_pack): [gamma 3 E2 | pad to 256 floats][beta 3 E2 | pad to 256][Wp [3][NQ][MT][part][64 lanes] 16-byte A operands].
// The workgroup's 80 KB of LDS hold groups 0 and 1 (one in the dead hidden tile, one in the dead spectra: two arrays, so the compiler keeps the waits
// of the two LDS-DMA batches apart); group 2's operands are read from the image itself (30 KB, L1 / L2 hits: the data loads bypass L1).
// (the tail as a real function - to keep its register allocation away from the chunk loop's - measured slower: its callee-saved registers go through scratch,
//  profiles/r06_tail_ab3.txt; it is inlined)
#define FDN_TAIL_FN __forceinline__
typedef __attribute__((address_space(3))) const float* lds_cf;
typedef __attribute__((address_space(3))) const fdn_u32x4* lds_cu4;
// PIN (C = 64 only): the following FDFFN's project_in (64 -> Hd <= 32 NT) as gemm_split_strip_kernel<4, FDN_PRO_LN> computes it - see fdsa_tail_px2; its
// packed operands ([NT][4 k-steps][3 parts][64 lanes] x 16 bytes, then 32 NT bias floats) are read from the image in global memory (72 KB: L1 / L2 hits)
template <int SH, int MT, bool FULL = false, bool PIN = false, int NT = 6>          // FULL: E == 2 SH and N == 32 MT (the stock level 2: E = 76, C = 64): no channel-range predicates
__device__ FDN_TAIL_FN void fdsa_tail_px1(const TailIo& io, lds_cf tg, lds_cf tb, lds_cu4 W0, lds_cu4 W1, const float* gimg,
                                              unsigned long long* trc = nullptr) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int E2 = 2 * SH, NQ = (SH + 7) / 8;
    const int E = io.E, N = io.N;
    const unsigned P = io.P, P4 = P * 4u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5, ln = lane & 31;
    constexpr unsigned PI = 1024u;
    const trsrc_t rg[3] = {tl_rsrc(io.scr, (unsigned)E * PI), tl_rsrc(io.scr + (long)E * 256, (unsigned)E * PI),
                           tl_rsrc(io.scr + (long)2 * E * 256, (unsigned)E * PI)};
    const trsrc_t rv = tl_rsrc(io.scr + (long)3 * E * 256, (unsigned)E * PI);
    const trsrc_t rw2 = tl_rsrc(gimg + 512 + 2 * NQ * MT * 3 * 64 * 4, (unsigned)(NQ * MT * 3 * 64) * 16u);      // group 2's operands
    const float invE = 1.0f / (float)E;
    const unsigned nb4 = (unsigned)N * P4;
    const trsrc_t ro = tl_rsrc(io.y, nb4);
    const trsrc_t rr = tl_rsrc(io.res ? io.res : io.y, io.res ? nb4 : 0u);
    const bool ok = io.tx0 + ln < io.W;

    float vv[SH], buf[2][SH];
    auto voff_of = [&](int rnd) { return kh * PI + (unsigned)((2 * wave + rnd) * 32 + ln) * 4u; };
    {
        const unsigned voff = voff_of(0);
        unsigned PIl = PI;
        asm volatile("" : "+s"(PIl));
#pragma unroll
        for (int s = 0; s < SH; ++s) {
            vv[s] = tl_load1(rv, voff, (unsigned)(2 * s) * PIl);
            buf[0][s] = tl_load1(rg[0], voff, (unsigned)(2 * s) * PIl);
        }
    }
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
        // (as in fdsa_out_vec_kernel: the plane offsets n P4 and the channel-range predicates are loop invariants the compiler would otherwise park in
        //  ~190 scalar registers / vector lanes; opaque per-row copies keep each one s_mul / one compare beside its use)
        unsigned P4l = P4;
        asm volatile("" : "+s"(P4l));
        int khl = kh;
        asm volatile("" : "+v"(khl));
        const unsigned voff = voff_of(rnd);
        const unsigned pix = (unsigned)((io.ty0 + 2 * wave + rnd) * io.W + io.tx0 + ln);
        const unsigned vo = ok ? (4u * kh * P + pix) * 4u : 0x80000000u;
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
        float rres[MT][16];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            float* cur = buf[(rnd + g) & 1];
            float* nxt = buf[(rnd + g + 1) & 1];
            unsigned PIl = PI;                      // (the plane offsets 2 s PI are loop invariants too: a scalar register each unless rebuilt beside the load)
            asm volatile("" : "+s"(PIl));
            if (g < 2) {
#pragma unroll
                for (int s = 0; s < SH; ++s) nxt[s] = tl_load1(rg[g + 1], voff, (unsigned)(2 * s) * PIl);
            }
            float m = 0.f;
#pragma unroll
            for (int s = 0; s < SH; ++s) m += cur[s];
            m = tl_xsum32(m) * invE;
            float q = 0.f;
#pragma unroll
            for (int s = 0; s < SH; ++s) {
                const float dl = cur[s] - m;
                q = tl_sq_acc<FULL>(q, dl, 2 * s + khl < E);
            }
            const float rs = tl_rsqrt_eps(tl_xsum32(q) * invE);
#pragma unroll
            for (int s = 0; s < SH; ++s) {
                asm volatile("" ::: "memory");
                const int e = 2 * s + kh;
                cur[s] = ((cur[s] - m) * rs * tg[g * E2 + e] + tb[g * E2 + e]) * vv[s];      // norm_g(out_g) * v_value  :633-638
            }
            if (g == 0 && rnd == 0) { TLTR(3) }
            if (g == 2 && rnd == 0) {          // the second row's v_value and group 0: v_value was last used just above, the idle set held group 1
                const unsigned voff1 = voff_of(1);
#pragma unroll
                for (int s = 0; s < SH; ++s) {
                    vv[s] = tl_load1(rv, voff1, (unsigned)(2 * s) * PIl);
                    nxt[s] = tl_load1(rg[0], voff1, (unsigned)(2 * s) * PIl);
                }
            }
            if (g == 2 && rnd == 1) tl_ring_release(io);      // the second row's last group is in registers
#pragma unroll
            for (int q8 = 0; q8 < NQ; ++q8) {
                fdn_u32x4 bx[3];
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    const int s0 = 8 * q8 + 2 * dd, s1 = s0 + 1;
                    unsigned p1, p2, p3;
                    fdn_split3(s0 < SH ? cur[s0 < SH ? s0 : 0] : 0.f, s1 < SH ? cur[s1 < SH ? s1 : 0] : 0.f, p1, p2, p3);
                    bx[0][dd] = p1, bx[1][dd] = p2, bx[2][dd] = p3;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    fdn_u32x4 a3[3];
                    if (g < 2) {
                        lds_cu4 wp = (g == 0 ? W0 : W1) + ((q8 * MT + mt) * 3) * 64 + lane;
                        a3[0] = wp[0]; a3[1] = wp[64]; a3[2] = wp[128];
                    } else {
                        const unsigned o = (unsigned)(((q8 * MT + mt) * 3) * 64 + lane) * 16u;
#pragma unroll
                        for (int part = 0; part < 3; ++part) a3[part] = __builtin_amdgcn_raw_buffer_load_b128(rw2, o, (unsigned)part * 1024u, 0);
                    }
                    acc[mt] = fdn_mfma_split6(a3, bx, acc[mt]);
                }
            }
            if (g == 0 && rnd == 0) {
                TLTR(4)
                // group 1's operands were requested (LDS-DMA into the spectra's array) by all four waves behind the barrier in front of this function:
                // every wave's share has to have landed before any wave reads them
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
        if (rnd == 0) { TLTR(5) }
        // ---- epilogue: residual (one batch), store, next LayerNorm's statistics (fdsa_out_vec_kernel's) ----
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                rres[mt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, vo, (unsigned)(mt * 32 + (r & 3) + 8 * (r >> 2)) * P4l, 0));      // 0 without a residual
        float outv[MT][16];
        float sm = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = mt * 32 + (r & 3) + 8 * (r >> 2);
                float o = acc[mt][r];
                o += rres[mt][r];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), ro, vo, (unsigned)nrow * P4l, 0);
                outv[mt][r] = (FULL || nrow + 4 * khl < N) ? o : 0.f;
                sm += outv[mt][r];
            }
        if (io.stats_out || PIN) {
            float sq = 0.f;
            const float mean = tl_xsum32(sm) / (float)N;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dl = outv[mt][r] - mean;
                    sq = tl_sq_acc<FULL>(sq, dl, mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * khl < N);
                }
            const float rstd = tl_rsqrt_eps(tl_xsum32(sq) / (float)N);
            if (io.stats_out && kh == 0) {
                const trsrc_t rs_ = tl_rsrc(io.stats_out, 2u * P4);
                const unsigned vs = ok ? pix * 4u : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mean), rs_, vs, 0u, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rstd), rs_, vs, P4, 0);
            }
            if constexpr (PIN) {
                static_assert(MT == 2, "project_in behind the level-2 tail: K = 64");
                const float sa = rstd, sb = -mean * sa;                                   // gemm_split: v = fmaf(v, rstd, -mean * rstd)
                fdn_u32x4 Bf[4][3];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int mt = ks >> 1, h8 = 8 * (ks & 1);
                    float v[8];
#pragma unroll
                    for (int b4 = 0; b4 < 4; ++b4) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(fmaf(outv[mt][h8 + b4], sa, sb)), __float_as_uint(fmaf(outv[mt][h8 + 4 + b4], sa, sb)), false, false);
                        v[b4] = __uint_as_float(sw[0]);
                        v[4 + b4] = __uint_as_float(sw[1]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned p1, p2, p3;
                        fdn_split3(v[2 * j], v[2 * j + 1], p1, p2, p3);
                        Bf[ks][0][j] = p1, Bf[ks][1][j] = p2, Bf[ks][2][j] = p3;
                    }
                }
                const trsrc_t rpw = tl_rsrc(gimg + tl_image_floats_px1_c(SH, MT), (unsigned)(NT * 4 * 3 * 64 * 16 + NT * 32 * 4));
                const trsrc_t rh = tl_rsrc(io.h, (unsigned)io.Hd * P4);
                const unsigned vh = ok ? (4u * kh * P + pix) * 4u : 0x80000000u;
                auto aload = [&](int t, int ks, fdn_u32x4 (&a3)[3]) {
#pragma unroll
                    for (int part = 0; part < 3; ++part)
                        a3[part] = __builtin_amdgcn_raw_buffer_load_b128(rpw, (unsigned)lane * 16u, (unsigned)(((t * 4 + ks) * 3 + part) * 1024), 0);
                };
                // the operands of tile t + 1 and its bias are requested in front of tile t's MFMAs (two register sets): read on the spot they cost an
                // exposed L2 round trip per tile and the fused project_in was no faster than its own launch
                fdn_u32x4 a3[2][4][3], bv[2][4];
                auto tload = [&](int t, int bufi) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) aload(t, ks, a3[bufi][ks]);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        bv[bufi][g4] = __builtin_amdgcn_raw_buffer_load_b128(rpw, (unsigned)(8 * g4 + 4 * kh) * 4u, (unsigned)(NT * 4 * 3 * 1024 + t * 128), 0);
                };
                tload(0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (t + 1 < NT) tload(t + 1, (t + 1) & 1);
                    f32x16 ah;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const fdn_u32x4 b_ = bv[t & 1][g4];
                        ah[4 * g4] = __uint_as_float(b_.x), ah[4 * g4 + 1] = __uint_as_float(b_.y), ah[4 * g4 + 2] = __uint_as_float(b_.z), ah[4 * g4 + 3] = __uint_as_float(b_.w);
                    }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) ah = fdn_mfma_split6(a3[t & 1][ks], Bf[ks], ah);
#pragma unroll
                    for (int r = 0; r < 16; ++r)                        // rows >= Hd fall outside the descriptor
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ah[r]), rh, vh, (unsigned)(t * 32 + (r & 3) + 8 * (r >> 2)) * P4l, 0);
                }
            }
        }
    }
    TLTR(6)
}
__host__ __device__ constexpr int tl_pin_floats_l2(int NT) { return NT * 4 * 3 * 64 * 4 + 256; }     // level 2: four k-steps
__host__ __device__ constexpr int tl_pin_floats(int NT) { return NT * 2 * 3 * 64 * 4 + 256; }        // A operands + one KB of bias
__host__ __device__ constexpr int tl_image_floats_px1(int SH, int MT) { return 512 + 3 * ((SH + 7) / 8) * MT * 3 * 64 * 4; }

}  // namespace
