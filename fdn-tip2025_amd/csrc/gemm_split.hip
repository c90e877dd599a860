// fp32 1x1-conv GEMM on the bf16 matrix pipe, for the deep level-3 shapes (to_hidden 128 -> 612, FDFFN project_in 128 -> 345 and
// project_out 345 -> 128, FDSA project_out 459 -> 128 with the 3 x LayerNorm * v_value prologue, FCAFFN / Fuse 128 -> 128;
// FDN_arch.py:576, :456, :474, :633-639, :421, :685).
//
// Why: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate and shares the vector ALU's datapath, so the
// level-3 GEMMs sat at 60-77 % "MFMA busy" with nothing left to gain.  v_mfma_f32_32x32x16_bf16 is a pipe of its own at 16x that
// rate.  An fp32 value splits EXACTLY into three bf16 values by truncation (8 + 8 + 8 significant bits: x = x1 + x2 + x3), so
//     w x = w1 x1 + (w1 x2 + w2 x1) + (w1 x3 + w2 x2 + w3 x1) + O(2^-24 |w x|)
// and the six bf16 products, each exact in the fp32 accumulator, reproduce the fp32 product to below one fp32 rounding; the sums
// are fp32 like the fmaf chain's.  Measured against float64 (tools/micro/split_bf16_mfma.hip): relative RMS error 8.5e-8 / 1.7e-7
// / 3.5e-7 at K = 32 / 128 / 512 against 1.05e-7 / 1.9e-7 / 4.0e-7 for the fp32 MFMA chain; 8 k of a 32 x 32 tile take 134 ns
// (split of the activations included) against 270 ns.  This is fp32 arithmetic carried out on bf16 multipliers, not bf16 precision.
//
// Layout: a workgroup owns 128 pixels x 128 output channels; K streams through LDS in chunks of 32 (two MFMA k-steps of 16).
// Activations are loaded as fp32 (thread = pixel, 16 consecutive k), get their prologue, are split and written to LDS as three bf16
// planes in MFMA operand order ([part][k-step][lane half][pixel] x 16 bytes: every operand read is one conflict-free ds_read_b128).
// The weights are split once, off line, by fdn_conv1x1_pack into the same order, so staging them is a 16-byte copy.  Waves form a
// 2 x 2 grid (two pixel strips x two channel tiles each): 24 MFMAs per k-step per wave behind 12 operand reads.
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mk_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(float v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
__device__ __forceinline__ f32x16 mf(fdn_u32x4 a, fdn_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int TP = 128, TN = 128, KC = 32;
constexpr int BLK = 3 * 2 * 2 * 128;          // 16-byte units of one operand chunk: [part][k-step][lane half][row]
constexpr int STRIP_MAX_N = 1024;             // widest output of the strip kernel (its bias lives in LDS)
constexpr int TRI_E = 10;                     // LN3_GATE: channels e per chunk (10 triples = 30 k + 2 zero columns)

struct SArgs {
    fdn_conv1x1_desc d;
    int tiles_per_img, total_ptiles, ntiles;
    // FC form (fdn_fcaffn_in_packed): xb = x1 is normalised on load (stats [B][2][P], gamma / beta [K]; null: used as it is) and the
    // epilogue's mul / add maps are evaluated from the 3-channel image (folded 3x3 o 1x1 weights as MFMA operands in `mapw`)
    const float* xb_stats; const float* xb_gamma; const float* xb_beta;
    const float* img; const fdn_u32x4* mapw; int img_w, img_h;
};

// k' (position in the packed K axis) -> source column of w / channel of x.  Natural order, or for LN3_GATE the triple order
// k' = 32 c + 16 hh + 3 j + g  ->  e = 10 c + 5 hh + j, channel g E + e  (position 15 of each half is a zero column)
__host__ __device__ __forceinline__ int tri_src(int kp, int E) {
    const int c = kp >> 5, hh = (kp >> 4) & 1, q = kp & 15;
    if (q == 15) return -1;
    const int j = q / 3, g = q - 3 * j, e = TRI_E * c + 5 * hh + j;
    return e < E ? g * E + e : -1;
}

template <int PRO, bool FC = false>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(SArgs a) {
    const fdn_conv1x1_desc& d = a.d;
    constexpr bool TRI = PRO == FDN_PRO_LN3_GATE, LN = PRO == FDN_PRO_LN, LNM = PRO == FDN_PRO_LN_MULADD;
    static_assert(!FC || LNM, "the FCAFFN form is the LN_MULADD prologue plus its own epilogue");
    __shared__ fdn_u32x4 Xs[BLK];
    __shared__ fdn_u32x4 Ws[BLK];
    __shared__ float red[2][TP];
    const int K = d.K, N = d.N, E = d.ln_group;
    const unsigned P = (unsigned)d.P, P4 = P * 4u;
    const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5, ln = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int nch = TRI ? (E + TRI_E - 1) / TRI_E : (K + KC - 1) / KC;

    // item S = (pixel tile, channel tile), channel tile fastest: the channel tiles of one pixel tile run on the same XCD back to
    // back and share its activations in L2
    const unsigned S = xcd_contiguous(blockIdx.x, (unsigned)(a.total_ptiles * a.ntiles));
    const unsigned pt = S / (unsigned)a.ntiles;
    const int nt = (int)(S - pt * a.ntiles);
    const int b = (int)(pt / (unsigned)a.tiles_per_img);
    const unsigned p0 = (pt - (unsigned)b * a.tiles_per_img) * TP;
    const int n0 = nt * TN;

    // ---- staging roles: thread = (pixel xp, k-step hh) handles the 16 k of that step; weights: six 16-byte units ----
    const int xp = tid & (TP - 1), hh = wave >> 1;
    const unsigned pix = min(p0 + (unsigned)xp, P - 1);
    // (round 5) two concatenated inputs (Fuse.conv over [enc | dnc], FDN_arch.py:685): a 32-deep chunk lies in one of them (the host checks
    // kseg[0] % 32 == 0), so the descriptor and the plane offset are chosen per chunk, uniformly
    const int K0 = d.kseg[1] > 0 ? d.kseg[0] : K;
    const rsrc_t rx = mk_rsrc(d.x[0] + (long)b * d.xbs[0], (unsigned)K0 * P4);
    const rsrc_t rx1 = d.kseg[1] > 0 ? mk_rsrc(d.x[1] + (long)b * d.xbs[1], (unsigned)(K - K0) * P4) : rx;
    const rsrc_t rv = mk_rsrc((TRI || LNM) ? d.xb + (long)b * d.xbbs : d.x[0], TRI ? (unsigned)E * P4 : LNM ? (unsigned)K * P4 : 0u);
    const fdn_u32x4* wsrc = reinterpret_cast<const fdn_u32x4*>(d.wpk) + (long)nt * nch * BLK;
    float sa[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};                  // (x - mean) * rstd = x * sa + sb
    if ((TRI || LN || LNM) && d.stats) {
#pragma unroll
        for (int g = 0; g < (TRI ? 3 : 1); ++g) {
            const float* sp = d.stats + ((long)b * (TRI ? 3 : 1) + g) * 2 * P;
            sa[g] = sp[P + pix];
            sb[g] = -sp[pix] * sa[g];
        }
    }
    if constexpr (TRI || LNM) {
        // (round 5) stats == NULL: the LayerNorm statistics of this pixel tile are taken HERE, in a pass over the tile's K planes before the
        // product (the fdn_chan_stats launch and its HBM read of the 3E / K planes go away; the product's own read of the tile then comes from
        // L2 / the Infinity Cache).  Shifted sums around the group's first channel, as fdn_chan_stats does (norm.hip); the two k-step halves of a
        // pixel (threads xp and xp + 128) each take their own channels and meet in LDS.
        if (!d.stats) {
            constexpr int G = TRI ? 3 : 1;
            const int Eg = TRI ? E : K;                                       // channels per LayerNorm group
            float x0[G], s[G], ss[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                x0[g] = bload(rx, pix * 4u, (unsigned)(g * Eg) * P4);
                s[g] = ss[g] = 0.f;
            }
            constexpr int PER = TRI ? 5 : 16, STEP = TRI ? TRI_E : KC;        // channels per thread and per chunk
            const int nfull = Eg / STEP;                                      // chunks whose channels all exist
#pragma unroll 2
            for (int c = 0; c < nfull; ++c) {
                float v[G][PER];
#pragma unroll
                for (int j = 0; j < PER; ++j)
#pragma unroll
                    for (int g = 0; g < G; ++g) v[g][j] = bload(rx, pix * 4u, (unsigned)(g * Eg + c * STEP + PER * hh + j) * P4);
#pragma unroll
                for (int j = 0; j < PER; ++j)
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float dl = v[g][j] - x0[g];
                        s[g] += dl;
                        ss[g] = fmaf(dl, dl, ss[g]);
                    }
            }
            for (int e = nfull * STEP + PER * hh; e < min(Eg, nfull * STEP + PER * hh + PER); ++e)       // the ragged last chunk (uniform bounds)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float dl = bload(rx, pix * 4u, (unsigned)(g * Eg + e) * P4) - x0[g];
                    s[g] += dl;
                    ss[g] = fmaf(dl, dl, ss[g]);
                }
            float* sx = reinterpret_cast<float*>(Xs);                         // [half][group][s | ss][pixel]: Xs is not in use yet
#pragma unroll
            for (int g = 0; g < G; ++g) {
                sx[((hh * G + g) * 2 + 0) * TP + xp] = s[g];
                sx[((hh * G + g) * 2 + 1) * TP + xp] = ss[g];
            }
            __syncthreads();
            const float inv = 1.0f / (float)Eg;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float st = sx[((0 * G + g) * 2 + 0) * TP + xp] + sx[((1 * G + g) * 2 + 0) * TP + xp];
                const float sst = sx[((0 * G + g) * 2 + 1) * TP + xp] + sx[((1 * G + g) * 2 + 1) * TP + xp];
                const float md = st * inv;                                    // mean - x0
                sa[g] = 1.0f / sqrtf(fmaxf(sst * inv - md * md, 0.f) + 1e-5f);
                sb[g] = -(x0[g] + md) * sa[g];
            }
            __syncthreads();                                                  // Xs is staged next
        }
    }
    const bool x1ln = FC && a.xb_stats != nullptr;                            // uniform
    float sa1 = 1.f, sb1 = 0.f;
    if (x1ln) {
        const float* sp = a.xb_stats + (long)b * 2 * P;
        sa1 = sp[P + pix];
        sb1 = -sp[pix] * sa1;
    }
    float xv[16], vv[(TRI || LNM) ? (TRI ? 5 : 16) : 1];
    float ga[(TRI || LNM) ? 16 : 1], be[(TRI || LNM) ? 16 : 1];
    float g1[FC ? 16 : 1], b1[FC ? 16 : 1];
    fdn_u32x4 wv[6];
    auto fetch = [&](int c) __attribute__((always_inline)) {
        if constexpr (TRI) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int e = c * TRI_E + 5 * hh + j;                         // wave-uniform; e >= E: outside the descriptors, reads 0
                vv[j] = e < E ? bload(rv, pix * 4u, (unsigned)e * P4) : 0.f;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    xv[3 * j + g] = e < E ? bload(rx, pix * 4u, (unsigned)(g * E + e) * P4) : 0.f;
                    ga[3 * j + g] = e < E ? d.gamma[g * E + e] : 0.f;
                    be[3 * j + g] = e < E ? d.beta[g * E + e] : 0.f;
                }
            }
        } else {
            const int k0 = c * KC + 16 * hh;
            const bool second = PRO == FDN_PRO_NONE && d.kseg[1] > 0 && k0 >= K0;      // wave-uniform; one input: k >= K stays on rx, outside its descriptor (reads 0)
            const rsrc_t rxc = second ? rx1 : rx;
            const int kb = second ? k0 - K0 : k0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                xv[i] = bload(rxc, pix * 4u, (unsigned)(kb + i) * P4);        // k >= K reads 0
                if constexpr (LNM) {
                    vv[i] = bload(rv, pix * 4u, (unsigned)(k0 + i) * P4);
                    ga[i] = k0 + i < K ? d.gamma[k0 + i] : 0.f;
                    be[i] = k0 + i < K ? d.beta[k0 + i] : 0.f;
                    if constexpr (FC) {
                        g1[i] = k0 + i < K ? (x1ln ? a.xb_gamma[k0 + i] : 1.f) : 0.f;
                        b1[i] = (x1ln && k0 + i < K) ? a.xb_beta[k0 + i] : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) wv[i] = wsrc[(long)c * BLK + tid + 256 * i];
    };
    auto stash = [&](int c) __attribute__((always_inline)) {
        float v[16];
        if constexpr (TRI) {
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    v[3 * j + g] = fmaf(fmaf(xv[3 * j + g], sa[g], sb[g]), ga[3 * j + g], be[3 * j + g]) * vv[j];   // FDN_arch.py:633-638 (e >= E: ga = be = 0)
            v[15] = 0.f;
        } else {
            const int k0 = c * KC + 16 * hh;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (LN) v[i] = k0 + i < K ? fmaf(xv[i], sa[0], sb[0]) : 0.f;       // affine part folded into the weights
                else if constexpr (LNM) {
                    float x1 = vv[i];
                    if constexpr (FC) x1 = fmaf(fmaf(x1, sa1, sb1), g1[i], b1[i]);            // x1 = norm3(block input), FDN_arch.py:675
                    v[i] = fmaf(fmaf(fmaf(xv[i], sa[0], sb[0]), ga[i], be[i]), x1, x1);       // norm(x) * x1 + x1, FDN_arch.py:420
                }
                else v[i] = xv[i];
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            fdn_u32x4 p1, p2, p3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = v[8 * h + 2 * j], x1 = v[8 * h + 2 * j + 1];
                p1[j] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
                const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u), r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
                p2[j] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
                const float s0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), s1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
                p3[j] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
            }
            Xs[((0 * 2 + hh) * 2 + h) * 128 + xp] = p1;
            Xs[((1 * 2 + hh) * 2 + h) * 128 + xp] = p2;
            Xs[((2 * 2 + hh) * 2 + h) * 128 + xp] = p3;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) Ws[tid + 256 * i] = wv[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][t][r] = 0.f;

    fetch(0);
    stash(0);
    __syncthreads();
    const fdn_u32x4* xb_ = Xs + kh * 128 + wi * 64 + ln;
    const fdn_u32x4* wb_ = Ws + kh * 128 + wj * 64 + ln;
    for (int c = 0; c < nch; ++c) {
        const bool more = c + 1 < nch;
        if (more) fetch(c + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            fdn_u32x4 A[2][3], B[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    B[s][p] = xb_[((p * 2 + ks) * 2) * 128 + s * 32];
                    A[s][p] = wb_[((p * 2 + ks) * 2) * 128 + s * 32];
                }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) {                                 // small terms first
                    acc[s][t] = mf(A[t][2], B[s][0], acc[s][t]);
                    acc[s][t] = mf(A[t][1], B[s][1], acc[s][t]);
                    acc[s][t] = mf(A[t][0], B[s][2], acc[s][t]);
                    acc[s][t] = mf(A[t][1], B[s][0], acc[s][t]);
                    acc[s][t] = mf(A[t][0], B[s][1], acc[s][t]);
                    acc[s][t] = mf(A[t][0], B[s][0], acc[s][t]);
                }
        }
        __syncthreads();                      // every wave has read this chunk
        if (more) stash(c + 1);
        __syncthreads();
    }

    // ---- epilogue: bias, residual | mul-add, store; LayerNorm statistics of the result (one channel tile only) ----
    const int Nt = min(N - n0, TN);           // channels of this tile
    const rsrc_t ro = mk_rsrc(d.out + (long)b * d.obs + (long)n0 * P, (unsigned)Nt * P4);
    const bool RES = d.epi == FDN_EPI_RES, MA = d.epi == FDN_EPI_MULADD;
    const rsrc_t rr = mk_rsrc(RES ? d.res + (long)b * d.rbs + (long)n0 * P : MA ? d.mul + (long)b * d.mbs + (long)n0 * P : d.out, (RES || MA) ? (unsigned)Nt * P4 : 0u);
    const rsrc_t ra = mk_rsrc(MA ? d.add + (long)b * d.mbs + (long)n0 * P : d.out, MA ? (unsigned)Nt * P4 : 0u);
    const rsrc_t rbias = mk_rsrc(d.bias ? d.bias + n0 : d.w, d.bias ? (unsigned)Nt * 4u : 0u);     // no bias: empty descriptor, reads 0
    if constexpr (FC) {
        // x * conv3_mul(conv1_mul(img)) + conv3_add(conv1_add(img))  (FDN_arch.py:423): the two maps of this wave's 64 pixels x 64
        // channels as MFMA chains over the 27 (tap, image channel) products, k = 3 tap + channel, zero outside the image
        const int Wd = a.img_w, Hh = a.img_h;
        const rsrc_t ri = mk_rsrc(a.img + (long)b * 3 * P, 3u * P4);
        fdn_u32x4 Bm[2][2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned pc = min(p0 + (unsigned)(wi * 64 + s * 32 + ln), P - 1);
            const int y = (int)(pc / (unsigned)Wd), x = (int)pc - y * Wd;
            const bool up = y > 0, dn = y < Hh - 1, lf = x > 0, rt = x < Wd - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kk0 = 16 * ks + j, kk1 = kk0 + 8;                       // lane half 0 / 1
                    const int t0 = kk0 / 3, c0 = kk0 - 3 * t0, dy0 = t0 / 3 - 1, dx0 = t0 % 3 - 1;
                    const int t1 = kk1 / 3, c1 = kk1 - 3 * t1, dy1 = t1 / 3 - 1, dx1 = t1 % 3 - 1;
                    const bool in0 = kk0 < 27 && (dy0 < 0 ? up : dy0 > 0 ? dn : true) && (dx0 < 0 ? lf : dx0 > 0 ? rt : true);
                    const bool in1 = kk1 < 27 && (dy1 < 0 ? up : dy1 > 0 ? dn : true) && (dx1 < 0 ? lf : dx1 > 0 ? rt : true);
                    const int off0 = c0 * (int)P + dy0 * Wd + dx0, off1 = c1 * (int)P + dy1 * Wd + dx1;
                    const bool in = kh ? in1 : in0;
                    const int off = kh ? off1 : off0;
                    v[j] = bload(ri, in ? (unsigned)((int)pc + off) * 4u : 0x80000000u, 0u);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned u1, u2, u3;
                    fdn_split3(v[2 * j], v[2 * j + 1], u1, u2, u3);
                    Bm[s][ks][0][j] = u1; Bm[s][ks][1][j] = u2; Bm[s][ks][2][j] = u3;
                }
            }
        }
        const int nt32 = (N + 31) / 32;
        const rsrc_t rb = mk_rsrc(d.bias ? d.bias + n0 : d.w, d.bias ? (unsigned)min(N - n0, TN) * 4u : 0u);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x16 mac[2][2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mac[s][t][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int ct = min(n0 / 32 + wj * 2 + t, nt32 - 1);                // (a tile past N: its rows are never stored)
                    const fdn_u32x4* mp = a.mapw + ((long)((m * 2 + ks) * 3) * nt32 + ct) * 64 + lane;
                    const fdn_u32x4 A3[3] = {mp[0], mp[(long)nt32 * 64], mp[(long)2 * nt32 * 64]};
#pragma unroll
                    for (int s = 0; s < 2; ++s) mac[s][t] = fdn_mfma_split6(A3, Bm[s][ks], mac[s][t]);
                }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (m == 0) {
                            const float bi = bload(rb, (unsigned)(4 * kh) * 4u, (unsigned)((wj * 2 + t) * 32 + (r & 3) + 8 * (r >> 2)) * 4u);
                            acc[s][t][r] = (acc[s][t][r] + bi) * mac[s][t][r];
                        } else {
                            acc[s][t][r] += mac[s][t][r];
                        }
                    }
        }
    }
    float psum[2] = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const unsigned p = p0 + (unsigned)(wi * 64 + s * 32 + ln);
        const unsigned voff = p < P ? (4u * kh * P + p) * 4u : 0x80000000u;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float rv_[16], av_[16], bi_[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bi_[r] = FC ? 0.f : bload(rbias, (unsigned)(4 * kh) * 4u, (unsigned)((wj * 2 + t) * 32 + (r & 3) + 8 * (r >> 2)) * 4u);
            if (RES || MA) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv_[r] = bload(rr, voff, (unsigned)((wj * 2 + t) * 32 + (r & 3) + 8 * (r >> 2)) * P4);
            }
            if (MA) {
#pragma unroll
                for (int r = 0; r < 16; ++r) av_[r] = bload(ra, voff, (unsigned)((wj * 2 + t) * 32 + (r & 3) + 8 * (r >> 2)) * P4);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = (wj * 2 + t) * 32 + (r & 3) + 8 * (r >> 2);      // + 4 kh per lane
                float v = acc[s][t][r] + bi_[r];
                if (RES) v += rv_[r];
                if (MA) v = fmaf(v, rv_[r], av_[r]);
                bstore(v, ro, voff, (unsigned)nrow * P4);                         // rows >= Nt fall outside the descriptor
                v = (nrow + 4 * kh < Nt) ? v : 0.f;
                acc[s][t][r] = v;
                psum[s] += v;
            }
        }
    }
    if (d.stats_out) {
        // two-pass mean / variance: the two waves of a pixel strip pair (wj = 0, 1) hold complementary channel halves
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            psum[s] += __shfl_xor(psum[s], 32);
            if (kh == 0) red[wj][wi * 64 + s * 32 + ln] = psum[s];
        }
        __syncthreads();
        float mean[2], q[2] = {0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pl = wi * 64 + s * 32 + ln;
            mean[s] = (red[0][pl] + red[1][pl]) / (float)N;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nrow = (wj * 2 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const float dl = acc[s][t][r] - mean[s];
                    q[s] += nrow < N ? dl * dl : 0.f;
                }
            q[s] += __shfl_xor(q[s], 32);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (kh == 0) red[wj][wi * 64 + s * 32 + ln] = q[s];
        __syncthreads();
        if (wj == 0 && kh == 0) {
            float* sp = d.stats_out + (long)b * 2 * P;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pl = wi * 64 + s * 32 + ln;
                const unsigned p = p0 + (unsigned)pl;
                if (p < P) {
                    sp[p] = mean[s];
                    sp[P + p] = 1.0f / sqrtf((red[0][pl] + red[1][pl]) / (float)N + 1e-5f);
                }
            }
        }
    }
}

template <int PRO>
int launch_split(const fdn_conv1x1_desc& d, hipStream_t s) {
    SArgs a = {};
    a.d = d;
    a.tiles_per_img = cdiv(d.P, TP);
    a.total_ptiles = d.B * a.tiles_per_img;
    a.ntiles = cdiv(d.N, TN);
    fdn_note_bf16_launch();
    hipLaunchKernelGGL(gemm_split_kernel<PRO>, dim3((unsigned)(a.total_ptiles * a.ntiles)), dim3(256), 0, s, a);
    return fdn_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// Short K, many output channels (to_hidden 128 -> 612, FDFFN project_in 128 -> 345; FDN_arch.py:576, :456): a K loop of four chunks
// cannot cover its own fill and drain (the tiled kernel above ran these at 25 % of the matrix rate).  Here the ACTIVATIONS stay put:
// a wave loads its 32-pixel strip over all K once, normalises and splits it, and keeps the three bf16 parts in registers as MFMA
// B operands (K = 128: 96 registers); the weights stream past it through LDS in 32-channel tiles (24 KB, double buffered, one
// barrier per tile), 6 K/16 MFMAs per tile and wave, and every tile's 32 x 32 result goes straight to memory.
// ---------------------------------------------------------------------------------------------------------------------------
template <int NKS, int PRO>
__global__ __launch_bounds__(256, 2) void gemm_split_strip_kernel(SArgs a) {
    const fdn_conv1x1_desc& d = a.d;
    constexpr int UNITS = 3 * NKS * 2 * 32;              // 16-byte units of one 32-channel weight tile
    constexpr int PER = (UNITS + 255) / 256;
    __shared__ fdn_u32x4 Ws[2][UNITS];
    __shared__ __attribute__((aligned(16))) float bs[STRIP_MAX_N];
    const int K = d.K, N = d.N;
    const unsigned P = (unsigned)d.P, P4 = P * 4u;
    const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5, ln = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nch = (K + KC - 1) / KC;
    const unsigned S = xcd_contiguous(blockIdx.x, (unsigned)a.total_ptiles);
    const int b = (int)(S / (unsigned)a.tiles_per_img);
    const unsigned p0 = (S - (unsigned)b * a.tiles_per_img) * TP;
    const unsigned p = p0 + (unsigned)(wave * 32 + ln), pix = min(p, P - 1);
    const rsrc_t rx = mk_rsrc(d.x[0] + (long)b * d.xbs[0], (unsigned)K * P4);
    const fdn_u32x4* wsrc = reinterpret_cast<const fdn_u32x4*>(d.wpk);

    // weight tile t (32 channels) as laid out by fdn_conv1x1_pack: unit u = ((part NKS + ks8) 2 + kh) 32 + n of the tile
    fdn_u32x4 wv[PER];
    auto fetch = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int u = tid + 256 * i;
            const int n = u & 31, h = (u >> 5) & 1, q = u >> 6, ks8 = q % NKS, part = q / NKS;
            const long src = ((((long)(t >> 2) * nch + (ks8 >> 1)) * 3 + part) * 2 + (ks8 & 1)) * 256 + h * 128 + (t & 3) * 32 + n;
            if (UNITS % 256 == 0 || u < UNITS) wv[i] = wsrc[src];
        }
    };
    auto stash = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (UNITS % 256 == 0 || tid + 256 * i < UNITS) Ws[buf][tid + 256 * i] = wv[i];
    };
    fetch(0);

    // ---- the strip: lane (pixel ln, half kh) holds k = 16 ks + 8 kh + j of its pixel, normalised, as three bf16 parts ----
    fdn_u32x4 Bf[NKS][3];
    {
        float sa = 1.f, sb = 0.f;
        if (PRO == FDN_PRO_LN) {
            const float* sp = d.stats + (long)b * 2 * P;
            sa = sp[P + pix];
            sb = -sp[pix] * sa;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = bload(rx, pix * 4u + (unsigned)(kh * 8) * P4, (unsigned)(ks * 16 + j) * P4);     // k >= K reads 0
            if (PRO == FDN_PRO_LN) {                 // (k >= K: any finite value will do, its packed weight is 0)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], sa, sb);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = v[2 * j], x1 = v[2 * j + 1];
                Bf[ks][0][j] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
                const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u), r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
                Bf[ks][1][j] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
                const float s0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), s1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
                Bf[ks][2][j] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
            }
        }
    }
    stash(0);
    for (int i = tid; i < STRIP_MAX_N; i += 256) bs[i] = (d.bias && i < N) ? d.bias[i] : 0.f;
    __syncthreads();

    const rsrc_t ro = mk_rsrc(d.out + (long)b * d.obs, (unsigned)N * P4);
    const unsigned voff = p < P ? (4u * kh * P + p) * 4u : 0x80000000u;
    const int ntl = (N + 31) / 32;
    for (int t = 0; t < ntl; ++t) {
        if (t + 1 < ntl) fetch(t + 1);
        const fdn_u32x4* wb = Ws[t & 1] + kh * 32 + ln;
        f32x16 acc;                               // starts from the bias (LDS copy: a global load here would wait behind the stores)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4*>(&bs[t * 32 + 8 * g + 4 * kh]);
            acc[4 * g] = bv.x, acc[4 * g + 1] = bv.y, acc[4 * g + 2] = bv.z, acc[4 * g + 3] = bv.w;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const fdn_u32x4 a1 = wb[((0 * NKS + ks) * 2) * 32], a2 = wb[((1 * NKS + ks) * 2) * 32], a3 = wb[((2 * NKS + ks) * 2) * 32];
            acc = mf(a3, Bf[ks][0], acc);
            acc = mf(a2, Bf[ks][1], acc);
            acc = mf(a1, Bf[ks][2], acc);
            acc = mf(a2, Bf[ks][0], acc);
            acc = mf(a1, Bf[ks][1], acc);
            acc = mf(a1, Bf[ks][0], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nrow = t * 32 + (r & 3) + 8 * (r >> 2);          // + 4 kh per lane
            bstore(acc[r], ro, voff, (unsigned)nrow * P4);             // rows >= N fall outside the descriptor
        }
        if (t + 1 < ntl) stash((t + 1) & 1);
        __syncthreads();
    }
}

// (round 4) The same kernel with the weight tiles travelling global -> LDS directly (buffer_load_dwordx4 ... lds: no staging registers,
// no ds_write) and TWO accumulators: tile t + 1's MFMA chain runs while tile t's 16 stores are issued between its MFMAs.  Knock-outs of the
// form above (tools/ab_gemm_l3.py, 128 -> 612: 0.43 ms as shipped, 0.28 without its stores, 0.34 without barrier + stash, 0.20 without
// both = the MFMA time) showed its waves marching in step: chain, stores and barrier added up instead of overlapping.
template <int NKS, int PRO>
__global__ __launch_bounds__(256, 2) void gemm_split_strip2_kernel(SArgs a, unsigned wbytes) {
    const fdn_conv1x1_desc& d = a.d;
    constexpr int UNITS = 3 * NKS * 2 * 32;              // 16-byte units of one 32-channel weight tile
    static_assert(UNITS % 256 == 0, "whole waves of 16-byte lanes");
    constexpr int PER = UNITS / 256;
    // (two arrays, not one [2][UNITS]: the compiler orders every LDS read behind ALL outstanding direct-to-LDS loads it cannot prove disjoint -
    //  with a run-time buffer index that is a vmcnt(0) in front of each tile's first operand read, i.e. no prefetch and a wait for the stores)
    __shared__ __attribute__((aligned(16))) fdn_u32x4 Ws0[UNITS];
    __shared__ __attribute__((aligned(16))) fdn_u32x4 Ws1[UNITS];
    __shared__ __attribute__((aligned(16))) float bs[STRIP_MAX_N];
    const int K = d.K, N = d.N;
    const unsigned P = (unsigned)d.P, P4 = P * 4u;
    const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5, ln = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nch = (K + KC - 1) / KC;
    const unsigned S = xcd_contiguous(blockIdx.x, (unsigned)a.total_ptiles);
    const int b = (int)(S / (unsigned)a.tiles_per_img);
    const unsigned p0 = (S - (unsigned)b * a.tiles_per_img) * TP;
    const unsigned p = p0 + (unsigned)(wave * 32 + ln), pix = min(p, P - 1);
    const rsrc_t rx = mk_rsrc(d.x[0] + (long)b * d.xbs[0], (unsigned)K * P4);
    const rsrc_t rw = mk_rsrc(static_cast<const float*>(d.wpk), wbytes);

    // weight tile t -> LDS buffer t & 1, unit u = tid + 256 i (layout as in gemm_split_strip_kernel): lane-contiguous 16-byte cells, so a
    // wave's 64 units are one direct-to-LDS load
    auto dma = [&](int t, fdn_u32x4* Wd) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int u = tid + 256 * i;
            const int n = u & 31, h = (u >> 5) & 1, q = u >> 6, ks8 = q % NKS, part = q / NKS;
            const int src = ((((t >> 2) * nch + (ks8 >> 1)) * 3 + part) * 2 + (ks8 & 1)) * 256 + h * 128 + (t & 3) * 32 + n;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(&Wd[256 * i + 64 * wave]), 16, src * 16, 0, 0, 0);
        }
    };
    dma(0, Ws0);

    fdn_u32x4 Bf[NKS][3];
    {
        float sa = 1.f, sb = 0.f;
        if (PRO == FDN_PRO_LN) {
            const float* sp = d.stats + (long)b * 2 * P;
            sa = sp[P + pix];
            sb = -sp[pix] * sa;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = bload(rx, pix * 4u + (unsigned)(kh * 8) * P4, (unsigned)(ks * 16 + j) * P4);     // k >= K reads 0
            if (PRO == FDN_PRO_LN) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], sa, sb);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned p1, p2, p3;
                fdn_split3(v[2 * j], v[2 * j + 1], p1, p2, p3);
                Bf[ks][0][j] = p1, Bf[ks][1][j] = p2, Bf[ks][2][j] = p3;
            }
        }
    }
    for (int i = tid; i < STRIP_MAX_N; i += 256) bs[i] = (d.bias && i < N) ? d.bias[i] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const rsrc_t ro = mk_rsrc(d.out + (long)b * d.obs, (unsigned)N * P4);
    const unsigned voff = p < P ? (4u * kh * P + p) * 4u : 0x80000000u;
    const int ntl = (N + 31) / 32;
    f32x16 acc[2];
    // tile t into acc[cur] while the 16 stores of tile t - 1 (acc[cur ^ 1]) go out, two behind every k-step's six MFMAs
    auto tile = [&](int t, const fdn_u32x4* Wc, fdn_u32x4* Wn, f32x16& cur, const f32x16& prev, bool has_prev) __attribute__((always_inline)) {
        if (t + 1 < ntl) dma(t + 1, Wn);
        const fdn_u32x4* wb = Wc + kh * 32 + ln;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4*>(&bs[t * 32 + 8 * g + 4 * kh]);
            cur[4 * g] = bv.x, cur[4 * g + 1] = bv.y, cur[4 * g + 2] = bv.z, cur[4 * g + 3] = bv.w;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const fdn_u32x4 a1 = wb[((0 * NKS + ks) * 2) * 32], a2 = wb[((1 * NKS + ks) * 2) * 32], a3 = wb[((2 * NKS + ks) * 2) * 32];
            cur = mf(a3, Bf[ks][0], cur);
            cur = mf(a2, Bf[ks][1], cur);
            cur = mf(a1, Bf[ks][2], cur);
            cur = mf(a2, Bf[ks][0], cur);
            cur = mf(a1, Bf[ks][1], cur);
            cur = mf(a1, Bf[ks][0], cur);
            if (has_prev) {
#pragma unroll
                for (int r = (16 * ks) / NKS; r < (16 * (ks + 1)) / NKS; ++r)
                    bstore(prev[r], ro, voff, (unsigned)((t - 1) * 32 + (r & 3) + 8 * (r >> 2)) * P4);      // rows >= N fall outside the descriptor
            }
            __builtin_amdgcn_sched_barrier(0);                // the stores stay between the k-steps
        }
        // the direct-to-LDS loads of tile t + 1 are older than the stores just issued: in-order completion, so "at most 16 outstanding" covers them
        if (has_prev) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    tile(0, Ws0, Ws1, acc[0], acc[1], false);
    int t = 1;
    for (; t + 1 < ntl; t += 2) {
        tile(t, Ws1, Ws0, acc[1], acc[0], true);
        tile(t + 1, Ws0, Ws1, acc[0], acc[1], true);
    }
    if (t < ntl) {
        tile(t, Ws1, Ws0, acc[1], acc[0], true);
        ++t;
    }
    const f32x16& last = (ntl & 1) ? acc[0] : acc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) bstore(last[r], ro, voff, (unsigned)((ntl - 1) * 32 + (r & 3) + 8 * (r >> 2)) * P4);
}

template <int NKS, int PRO>
int launch_strip(const fdn_conv1x1_desc& d, hipStream_t s) {
    SArgs a = {};
    a.d = d;
    a.tiles_per_img = cdiv(d.P, TP);
    a.total_ptiles = d.B * a.tiles_per_img;
    a.ntiles = 1;
#ifndef FDN_STRIP2
#define FDN_STRIP2 1
#endif
    if constexpr (FDN_STRIP2 && (3 * NKS * 2 * 32) % 256 == 0 && NKS >= 6) {
        const long wbytes = (long)cdiv(d.N, TN) * ((d.K + KC - 1) / KC) * BLK * 16;      // = fdn_conv1x1_pack_bytes(N, K, 0)
        if (wbytes < 0x7FFFFFFFL) {
            fdn_note_bf16_launch();
            hipLaunchKernelGGL((gemm_split_strip2_kernel<NKS, PRO>), dim3((unsigned)a.total_ptiles), dim3(256), 0, s, a, (unsigned)wbytes);
            return fdn_launch_status();
        }
    }
    fdn_note_bf16_launch();
    hipLaunchKernelGGL((gemm_split_strip_kernel<NKS, PRO>), dim3((unsigned)a.total_ptiles), dim3(256), 0, s, a);
    return fdn_launch_status();
}

// fdn_fcaffn_in_pack: the two folded image maps as MFMA A operands: cell (map, k-step, part, 32-channel tile, lane) holds the part-th
// bf16 part of w3[c][tap] * w1[c][ch] (one fp32 rounding, as fdn_fcaffn_in folds them) for k = 16 ks + 8 kh + j = 3 tap + ch < 27
__global__ void pack_imgmod_kernel(const float* __restrict__ w1m, const float* __restrict__ w3m, const float* __restrict__ w1a,
                                   const float* __restrict__ w3a, fdn_u32x4* __restrict__ out, int C, int nt32, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63), ct = (int)((i >> 6) % nt32);
    const long q = (i >> 6) / nt32;
    const int part = (int)(q % 3), ks = (int)((q / 3) % 2), m = (int)(q / 6);
    const int c = ct * 32 + (lane & 31), kh = lane >> 5;
    const float* w1 = m ? w1a : w1m;
    const float* w3 = m ? w3a : w3m;
    fdn_u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kk = 16 * ks + 8 * kh + 2 * j + u;
            float x = (c < C && kk < 27) ? w3[(long)c * 9 + kk / 3] * w1[(long)c * 3 + kk % 3] : 0.f;
            for (int p = 0; p < part; ++p) x -= __uint_as_float(__float_as_uint(x) & 0xffff0000u);
            v[u] = x;
        }
        o[j] = (__float_as_uint(v[0]) >> 16) | (__float_as_uint(v[1]) & 0xffff0000u);
    }
    out[i] = o;
}

// one thread per 16-byte unit of the packed weights: 8 consecutive k' of one output row, one of the three bf16 parts
__global__ void pack_split_kernel(const float* __restrict__ w, fdn_u32x4* __restrict__ out, int N, int K, int E, int nch, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = (int)(i & 127), h = (int)((i >> 7) & 1), ks = (int)((i >> 8) & 1);
    const long q = i >> 9;
    const int part = (int)(q % 3), c = (int)((q / 3) % nch), nt = (int)(q / 3 / nch);
    const int row = nt * TN + n;
    fdn_u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kp = c * KC + ks * 16 + h * 8 + 2 * j + u;
            const int k = E > 0 ? tri_src(kp, E) : (kp < K ? kp : -1);
            float x = (row < N && k >= 0) ? w[(long)row * K + k] : 0.f;
            for (int p = 0; p < part; ++p) x -= __uint_as_float(__float_as_uint(x) & 0xffff0000u);
            v[u] = x;
        }
        o[j] = (__float_as_uint(v[0]) >> 16) | (__float_as_uint(v[1]) & 0xffff0000u);
    }
    out[i] = o;
}

}  // namespace

// FDN_ERR_UNSUPPORTED = not a shape of this kernel (fdn_conv1x1 then picks another)
int fdn_gemm_split(const fdn_conv1x1_desc& d, hipStream_t s) {
    if (fdn_matrix_pipe_f32()) return FDN_ERR_UNSUPPORTED;
    // lanes past the pixel count are masked with a byte offset of 2^31: it must stay outside every descriptor of this kernel
    if ((unsigned long long)(d.N > d.K ? d.N : d.K) * 4ull * (unsigned long long)d.P > 0x7FFFFFFFull) return FDN_ERR_UNSUPPORTED;
    if (!d.wpk || d.kseg[2] > 0 || d.act != FDN_ACT_NONE || d.x_bf16 || d.out_bf16) return FDN_ERR_UNSUPPORTED;
    const bool two = d.kseg[1] > 0;                       // two inputs: the K-streaming kernel only, plain prologue, whole chunks per input
    if (two && (d.pro != FDN_PRO_NONE || d.kseg[0] % KC != 0 || d.K < 96 || d.N < 96)) return FDN_ERR_UNSUPPORTED;
    if ((long)d.B * cdiv(d.P, TP) * cdiv(d.N, TN) > 0x7FFFFFFFL) return FDN_ERR_UNSUPPORTED;
    // short K, wide N, no epilogue: the activation strip stays in registers and the weights stream
    const bool strip = !two && d.N <= STRIP_MAX_N && d.epi == FDN_EPI_NONE && !d.stats_out && (d.pro == FDN_PRO_NONE || d.pro == FDN_PRO_LN);
    // (round 3) the project_in convs of levels 1-2 as well (32 -> 86, 64 -> 172; FDN_lolv1 24 -> 64, 48 -> 129): on the fp32 MFMA they kept
    // the vector ALU's datapath 60-90 % busy (64 -> 172: 0.49 -> 0.41 ms, 32 -> 86: 0.82 -> 0.75 ms here)
    if (strip && d.K > 16 && d.K <= 64 && 2 * d.N >= 5 * d.K) {
        const bool ln = d.pro == FDN_PRO_LN;
        if (d.K > 48) return ln ? launch_strip<4, FDN_PRO_LN>(d, s) : launch_strip<4, FDN_PRO_NONE>(d, s);
        if (d.K > 32) return ln ? launch_strip<3, FDN_PRO_LN>(d, s) : launch_strip<3, FDN_PRO_NONE>(d, s);
        return ln ? launch_strip<2, FDN_PRO_LN>(d, s) : launch_strip<2, FDN_PRO_NONE>(d, s);
    }
    if (d.K < 96 || d.N < 96 || (d.stats_out && d.N > TN)) return FDN_ERR_UNSUPPORTED;
    if (d.K <= 128 && d.N >= 256 && strip) {
        const bool ln = d.pro == FDN_PRO_LN;
        if (d.K > 112) return ln ? launch_strip<8, FDN_PRO_LN>(d, s) : launch_strip<8, FDN_PRO_NONE>(d, s);
        if (d.K > 96) return ln ? launch_strip<7, FDN_PRO_LN>(d, s) : launch_strip<7, FDN_PRO_NONE>(d, s);
        return ln ? launch_strip<6, FDN_PRO_LN>(d, s) : launch_strip<6, FDN_PRO_NONE>(d, s);          // K = 96 (FDN_lolv1)
    }
    switch (d.pro) {
        case FDN_PRO_NONE: return launch_split<FDN_PRO_NONE>(d, s);
        case FDN_PRO_LN: return launch_split<FDN_PRO_LN>(d, s);
        case FDN_PRO_LN3_GATE: return launch_split<FDN_PRO_LN3_GATE>(d, s);
        case FDN_PRO_LN_MULADD: return launch_split<FDN_PRO_LN_MULADD>(d, s);
    }
    return FDN_ERR_UNSUPPORTED;
}

extern "C" long fdn_conv1x1_pack_bytes(int N, int K, int ln3_E);
extern "C" long fdn_fcaffn_in_pack_bytes(int C) {
    return fdn_conv1x1_pack_bytes(C, C, 0) + 2L * 2 * 3 * cdiv(C, 32) * 64 * 16;
}

extern "C" int fdn_fcaffn_in_pack(const float* w, const float* w1_mul, const float* w3_mul, const float* w1_add, const float* w3_add, int C,
                                  void* wpk, fdn_stream_t stream) {
    FDN_CHECK_ARG(w && w1_mul && w3_mul && w1_add && w3_add && wpk && C > 0);
    if (int e = fdn_conv1x1_pack(w, C, C, 0, wpk, stream)) return e;
    const int nt32 = cdiv(C, 32);
    const long total = 2L * 2 * 3 * nt32 * 64;
    fdn_u32x4* maps = reinterpret_cast<fdn_u32x4*>(static_cast<char*>(wpk) + fdn_conv1x1_pack_bytes(C, C, 0));
    hipLaunchKernelGGL(pack_imgmod_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w1_mul, w3_mul,
                       w1_add, w3_add, maps, C, nt32, total);
    return fdn_launch_status();
}

extern "C" int fdn_fcaffn_in_packed(const float* xi, const float* stats_xi, const float* x1, const float* stats1, const float* gamma1,
                                    const float* beta1, const float* img, const void* wpk, const float* gamma, const float* beta, float* out,
                                    int B, int C, int H, int W, fdn_stream_t stream) {
    FDN_CHECK_ARG(xi && x1 && img && wpk && gamma && beta && out && B > 0 && C > 0 && H > 0 && W > 0);      // stats_xi NULL: taken in the kernel
    FDN_CHECK_ARG((stats1 && gamma1 && beta1) || (!stats1 && !gamma1 && !beta1));
    const long P = (long)H * W;
    if (fdn_matrix_pipe_f32()) return FDN_ERR_UNSUPPORTED;
    if (C < 96 || (unsigned long long)(C + 4) * 4ull * P > 0x7FFFFFFFull) return FDN_ERR_UNSUPPORTED;      // narrower: fdn_fcaffn_in
    if ((long)B * cdiv(P, TP) * cdiv(C, TN) > 0x7FFFFFFFL) return FDN_ERR_UNSUPPORTED;
    SArgs a = {};
    fdn_conv1x1_desc& d = a.d;
    d.x[0] = xi; d.xbs[0] = (long)C * P; d.kseg[0] = C;
    d.w = static_cast<const float*>(wpk);          // (not read: the packed form below carries the weights)
    d.wpk = wpk;
    d.out = out; d.obs = (long)C * P;
    d.B = B; d.K = C; d.N = C; d.P = (int)P;
    d.pro = FDN_PRO_LN_MULADD; d.stats = stats_xi; d.gamma = gamma; d.beta = beta;
    d.xb = x1; d.xbbs = (long)C * P;
    d.act = FDN_ACT_NONE; d.epi = FDN_EPI_NONE;
    a.xb_stats = stats1; a.xb_gamma = gamma1; a.xb_beta = beta1;
    a.img = img; a.img_w = W; a.img_h = H;
    a.mapw = reinterpret_cast<const fdn_u32x4*>(static_cast<const char*>(wpk) + fdn_conv1x1_pack_bytes(C, C, 0));
    a.tiles_per_img = cdiv((int)P, TP);
    a.total_ptiles = B * a.tiles_per_img;
    a.ntiles = cdiv(C, TN);
    fdn_note_bf16_launch();
    hipLaunchKernelGGL((gemm_split_kernel<FDN_PRO_LN_MULADD, true>), dim3((unsigned)(a.total_ptiles * a.ntiles)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    return fdn_launch_status();
}

extern "C" long fdn_conv1x1_pack_bytes(int N, int K, int ln3_E) {
    const long nch = ln3_E > 0 ? (ln3_E + TRI_E - 1) / TRI_E : (K + KC - 1) / KC;
    return (long)cdiv(N, TN) * nch * BLK * 16;
}

extern "C" int fdn_conv1x1_pack(const float* w, int N, int K, int ln3_E, void* wpk, fdn_stream_t stream) {
    FDN_CHECK_ARG(w && wpk && N > 0 && K > 0 && (ln3_E == 0 || 3 * ln3_E == K));
    const int nch = ln3_E > 0 ? (ln3_E + TRI_E - 1) / TRI_E : (K + KC - 1) / KC;
    const long total = (long)cdiv(N, TN) * nch * BLK;
    hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                       static_cast<fdn_u32x4*>(wpk), N, K, ln3_E, nch, total);
    return fdn_launch_status();
}
