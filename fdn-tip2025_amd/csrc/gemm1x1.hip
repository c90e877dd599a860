// 1x1 convolution (the 79 % of FDN's MACs) as an exact-fp32 MFMA GEMM on NCHW planes.
//
//   out[b][n][p] = epi( act( sum_k W[n][k] * pro(x[b][k][p]) + bias[n] ) )
//
// Replaces the F.conv2d(kernel_size=1) calls of FDN_arch.py:576 (to_hidden), :639 (attn
// project_out), :456/:474 (FDFFN project_in/out), :421/:428 (FCAFFN), :685-686 (Fuse) and the
// 62 1x1 convs of MAR (:78-86, :125-134, :168-190), with the surrounding channel-LayerNorm
// (:313-342), the v_value gating (:633-638), `norm(x)*x1+x1` (:420), the residual adds (:671-675)
// and `x*mul+add` (:423) folded into the operand path / epilogue.
//
// Mapping (CDNA4, v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-exact fmaf chain):
//   D[n][p] = A[n][k] * B[k][p],  A = weights, B = activations.
// Pixels are the contiguous axis of NCHW and sit on the MFMA column (lane) axis, so
//   * the B operand of lane l for k-step s is x[k = 2s + (l>>5)][p = p0 + (l&31)]: one coalesced
//     global dword load (two 128-B lines per wave-instruction) straight into the VGPR the MFMA
//     reads - activations never touch LDS and each wave owns its 32-pixel strip exclusively;
//   * the accumulator of reg r is row n = 32*mt + (r&3) + 8*(r>>2) + 4*(l>>5), column p, so every
//     store instruction writes two full 128-byte lines;
//   * only the weights (shared by all waves) live in LDS, transposed to [k][n]; when the K x (MT*32)
//     slice fits they are loaded once per workgroup ("resident") and the persistent tile loop runs
//     without any barrier, otherwise 32-deep K chunks are double-buffered with one barrier a step.
// The next step's activations are prefetched into a second register set while the current step's
// MFMAs issue.  Prologues (LayerNorm, 3xLayerNorm * v_value, LayerNorm * x1 + x1) are applied in
// registers between the load and the MFMA.
#include "common.hpp"
#include <type_traits>

namespace {

constexpr int KC = 32;     // K chunk (16 MFMA k-steps)
typedef float f32x16 __attribute__((ext_vector_type(16)));

// The activation / epilogue kind are wave-uniform runtime values of the descriptor.  Tested per element they cost a chain
// of scalar compare-and-branch per output value (the switch of apply_act alone is ~8 branches: 5.3k of a 23.5k-cycle output
// tile of the level-3 to_hidden conv went there, tools/gemm_trace2.py).  Each kernel therefore writes its epilogue once as a
// generic lambda over two integral constants and picks the copy once per tile.
template <int V> using IC = std::integral_constant<int, V>;
#define FDN_EPI_MODES(act_c, epi_c)                                                                     \
    const int act_ = decltype(act_c)::value ? d.act : (int)FDN_ACT_NONE;                                \
    const int epi_ = decltype(epi_c)::value >= 0 ? (int)decltype(epi_c)::value : d.epi;
#define FDN_EPI_DISPATCH(f)                                                                             \
    do {                                                                                                \
        if (d.act != FDN_ACT_NONE) f(IC<1>(), IC<-1>());                                                \
        else if (d.epi == FDN_EPI_NONE) f(IC<0>(), IC<FDN_EPI_NONE>());                                 \
        else if (d.epi == FDN_EPI_RES) f(IC<0>(), IC<FDN_EPI_RES>());                                   \
        else f(IC<0>(), IC<FDN_EPI_MULADD>());                                                          \
    } while (0)
#define FDN_ACT_DISPATCH(f)                                                                             \
    do {                                                                                                \
        if (d.act != FDN_ACT_NONE) f(IC<1>(), IC<-1>());                                                \
        else f(IC<0>(), IC<-1>());                                                                      \
    } while (0)

struct Geo {
    int tiles_per_img;     // pixel tiles (NW*32 px) per image
    int total_tiles;       // B * tiles_per_img
    int resident;          // whole K x (MT*32) weight slice kept in LDS
    int bias_off;          // float offset of the bias table in dynamic LDS (generic kernel)
};

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mk_rsrc(const float* base, unsigned bytes) {
    // raw buffer, stride 0: offsets >= bytes read 0 / drop the store (K, N and plane tails for free)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(float v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

template <int MT, int PRO, int NW, bool EARLY>
__global__ __launch_bounds__(NW * 64, (NW == 4 && !EARLY) ? ((MT <= 2 && PRO == FDN_PRO_NONE) ? 3 : 2) : 1) void conv1x1_kernel(fdn_conv1x1_desc d, Geo g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = NW * 64;
    constexpr int WS = MT * 32 + 1;            // LDS row stride of the transposed weight chunk
    constexpr int CH = KC * WS;                // floats per chunk buffer
    const int K = d.K, N = d.N;
    const unsigned P = (unsigned)d.P, P4 = P * 4u;
    // LN3_GATE walks K = 3E as (e, E+e, 2E+e) triples, 5 channel pairs e = 2j+kh per chunk (15 k-steps): the three
    // k-steps of a triple share one v_value operand (a third of the loads and registers of the plain k order) and
    // their LayerNorm group is a compile-time constant
    constexpr bool TRI = PRO == FDN_PRO_LN3_GATE;
    constexpr int KS = TRI ? 15 : 16;          // MFMA k-steps per chunk
    const int E = d.ln_group;
    const int nch = TRI ? ((E + 1) / 2 + 4) / 5 : (K + KC - 1) / KC;
    const int Kp = nch * KC;
    auto kbase = [&](int c_, int s_) { return TRI ? (s_ % 3) * E + 2 * (c_ * 5 + s_ / 3) : c_ * KC + 2 * s_; };   // even-lane k of a step
    float* tg = smem;                          // gamma[Kp]
    float* tb = smem + Kp;                     // beta[Kp]
    float* Wl = smem + 2 * Kp;                 // weight chunks

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5, ln = lane & 31;

    for (int i = tid; i < Kp; i += NT) {
        tg[i] = (PRO >= FDN_PRO_LN3_GATE && i < K) ? d.gamma[i] : 0.f;       // (PRO_LN: gamma / beta are folded into w / bias by the caller)
        tb[i] = (PRO >= FDN_PRO_LN3_GATE && i < K) ? d.beta[i] : 0.f;
    }
    float* bl = smem + g.bias_off;             // bias (zeros without one), read from LDS in the epilogue
    for (int i = tid; i < ((N + 31) & ~31) + 32 * 5; i += NT) bl[i] = (d.bias && i < N) ? d.bias[i] : 0.f;

    const int ks0 = d.kseg[0], ks01 = d.kseg[0] + d.kseg[1];
    const int npass = (N + MT * 32 - 1) / (MT * 32);
    constexpr int WPT = (KC * MT * 32) / NT;   // weight elements per thread per chunk

    for (int pass = 0; pass < npass; ++pass) {
        const int nbase = pass * MT * 32;

        // ---- weight chunk loader: W[n][k] (lanes along k) -> regs -> Wl[buf][k][n] ---------------
        float wr[WPT];
        auto w_fetch = [&](int c) {
            const int kk = tid & 31;           // LDS row = 2 * k-step + lane half
            const int k = kbase(c, kk >> 1) + (kk & 1);
            const bool kok = TRI ? ((kk >> 1) < KS && 2 * (c * 5 + (kk >> 1) / 3) + (kk & 1) < E) : k < K;
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const int n = nbase + (tid >> 5) + (NT / 32) * i;
                wr[i] = (n < N && kok) ? d.w[(long)n * K + k] : 0.f;
            }
        };
        auto w_stash = [&](int buf) {
            const int kk = tid & 31;
            float* dst = Wl + buf * CH + kk * WS;
#pragma unroll
            for (int i = 0; i < WPT; ++i) dst[(tid >> 5) + (NT / 32) * i] = wr[i];
        };
        __syncthreads();                       // previous pass finished reading Wl; tables written
        if (g.resident) {
            for (int c = 0; c < nch; ++c) { w_fetch(c); w_stash(c); }
        } else {
            w_fetch(0);
            w_stash(0);
        }
        __syncthreads();

        // ---- persistent loop over (tile, chunk) steps ---------------------------------------------
        int tile = blockIdx.x, c = 0;
        bool live = tile < g.total_tiles;

        float xa[16], xb[16];                  // current / prefetched activations
        float ya[16], yb[16];                  // second operand (v_value / x1) for PRO 2,3
        float mu[3], rs[3];                    // LayerNorm statistics of this lane's pixel

        struct Tile { int b; unsigned pix; bool ok; };
        auto tile_setup = [&](int t) {
            Tile r;
            r.b = t / g.tiles_per_img;
            const unsigned p_ = (unsigned)(t - r.b * g.tiles_per_img) * (NW * 32) + wave * 32 + ln;
            r.ok = p_ < P;
            r.pix = r.ok ? p_ : P - 1;         // clamp: harmless loads, never stored
            return r;
        };
        auto x_issue = [&](const Tile& t, int c_, float (&xv)[16], float (&yv)[16]) {
            const rsrc_t r0 = mk_rsrc(d.x[0] + (long)t.b * d.xbs[0], (unsigned)ks0 * P4);
            const rsrc_t r1 = mk_rsrc(d.x[1] + (long)t.b * d.xbs[1], (unsigned)d.kseg[1] * P4);
            const rsrc_t r2 = mk_rsrc(d.x[2] + (long)t.b * d.xbs[2], (unsigned)d.kseg[2] * P4);
            const unsigned voff = (kh * P + t.pix) * 4u;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int k = kbase(c_, s);                         // wave-uniform
                if (TRI || k < ks0) xv[s] = bload(r0, voff, (unsigned)k * P4);      // (LN3_GATE: one segment; k >= K reads 0)
                else if (k < ks01) xv[s] = bload(r1, voff, (unsigned)(k - ks0) * P4);
                else xv[s] = bload(r2, voff, (unsigned)(k - ks01) * P4);
            }
            if (PRO == FDN_PRO_LN3_GATE) {
                const rsrc_t ry = mk_rsrc(d.xb + (long)t.b * d.xbbs, (unsigned)E * P4);
#pragma unroll
                for (int j = 0; j < 5; ++j) yv[j] = bload(ry, voff, (unsigned)(2 * (c_ * 5 + j)) * P4);   // v_value[e], e = 2j + kh
            } else if (PRO == FDN_PRO_LN_MULADD) {
                const rsrc_t ry = mk_rsrc(d.xb + (long)t.b * d.xbbs, (unsigned)K * P4);
#pragma unroll
                for (int s = 0; s < 16; ++s) yv[s] = bload(ry, voff, (unsigned)(c_ * KC + 2 * s) * P4);
            }
        };
        auto stats_load = [&](const Tile& t) {
            if (PRO == FDN_PRO_NONE) return;
            constexpr int G = (PRO == FDN_PRO_LN3_GATE) ? 3 : 1;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (q < G) {
                    const float* sp = d.stats + ((long)t.b * G + q) * 2 * P;
                    mu[q] = sp[t.pix];
                    rs[q] = sp[P + t.pix];
                } else { mu[q] = 0.f; rs[q] = 0.f; }
            }
        };

        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

        Tile cur = tile_setup(live ? tile : 0);
        if (live) {
            stats_load(cur);
            x_issue(cur, 0, xa, ya);
        }
        int step = 0;
#ifdef FDN_GEMM_TRACE   // tools/gemm_trace.py: per-step s_memtime stamps of waves 0 and 4 of workgroup 0 into d.mul (vec4 = 12345)
        unsigned long long* trc = (d.vec4 == 12345 && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0)
                                      ? reinterpret_cast<unsigned long long*>(const_cast<float*>(d.mul)) + (wave ? 1024 : 0) : nullptr;
#define TRC(i) if (trc && step < 120) trc[step * 8 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define TRC(i)
#endif
        while (live) {
            TRC(0)
            // next step
            int ntile = tile, nc = c + 1;
            if (nc == nch) { nc = 0; ntile = tile + gridDim.x; }
            const bool nlive = ntile < g.total_tiles;
            Tile nxt = cur;
            if (nlive) {
                if (nc == 0) nxt = tile_setup(ntile);
                x_issue(nxt, nc, xb, yb);                          // prefetch: overlaps the MFMAs below
                if (!g.resident) w_fetch(nc);
            }

            // EARLY (narrow outputs, K <= 64): the epilogue's residual / mul / add operands are requested before
            // the MFMAs of the tile's last chunk, so their latency hides behind the matrix work instead of
            // stalling every tile (the K = N = C FCAFFN convs are a load-latency chain otherwise: 2.4 -> 1.1 ms)
            float e0[EARLY ? MT * 16 : 1], e1[EARLY ? MT * 16 : 1];
            if (EARLY && c == nch - 1 && d.epi != FDN_EPI_NONE) {
                const unsigned nb4 = (unsigned)N * P4;
                const unsigned voff = (4u * kh * P + cur.pix) * 4u;
                if (d.epi == FDN_EPI_RES) {
                    const rsrc_t rr = mk_rsrc(d.res + (long)cur.b * d.rbs, nb4);
#pragma unroll
                    for (int i = 0; i < MT * 16; ++i)
                        e0[i] = bload(rr, voff, (unsigned)(nbase + (i >> 4) * 32 + (i & 3) + 8 * ((i & 15) >> 2)) * P4);
                } else {
                    const rsrc_t rm = mk_rsrc(d.mul + (long)cur.b * d.mbs, nb4);
                    const rsrc_t rd = mk_rsrc(d.add + (long)cur.b * d.mbs, nb4);
#pragma unroll
                    for (int i = 0; i < MT * 16; ++i) {
                        const unsigned soff = (unsigned)(nbase + (i >> 4) * 32 + (i & 3) + 8 * ((i & 15) >> 2)) * P4;
                        e0[i] = bload(rm, voff, soff);
                        e1[i] = bload(rd, voff, soff);
                    }
                }
            }

            TRC(1)
            // ---- compute step (tile, c) -----------------------------------------------------------
            const float* Wc = Wl + (g.resident ? c : (step & 1)) * CH;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                float bv = xa[s];
                if (PRO != FDN_PRO_NONE) {
                    const int k = min(kbase(c, s) + kh, Kp - 1);    // (tables are zero past K)
                    asm volatile("" ::: "memory");                  // table reads stay inside the step (see conv1x1_smallk_vec_kernel)
                    const float ga = PRO == FDN_PRO_LN ? 1.f : tg[k], be = PRO == FDN_PRO_LN ? 0.f : tb[k];
                    if (PRO == FDN_PRO_LN) {
                        bv = (bv - mu[0]) * rs[0];
                    } else if (PRO == FDN_PRO_LN3_GATE) {
                        bv = ((bv - mu[s % 3]) * rs[s % 3] * ga + be) * ya[s / 3];
                    } else {
                        bv = ((bv - mu[0]) * rs[0] * ga + be) * ya[s] + ya[s];
                    }
                }
                const float* wrow = Wc + (2 * s + kh) * WS + ln;
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[m * 32], bv, acc[m], 0, 0, 0);
            }

            TRC(2)
            // ---- epilogue at the last chunk of a tile --------------------------------------------------
            if (c == nch - 1) {
                if (cur.ok) {
                    const unsigned nb4 = (unsigned)N * P4;
                    const rsrc_t ro = mk_rsrc(d.out + (long)cur.b * d.obs, nb4);
                    const rsrc_t rr = mk_rsrc(d.res ? d.res + (long)cur.b * d.rbs : d.out, d.res ? nb4 : 0u);
                    const rsrc_t rm = mk_rsrc(d.mul ? d.mul + (long)cur.b * d.mbs : d.out, d.mul ? nb4 : 0u);
                    const rsrc_t rd = mk_rsrc(d.add ? d.add + (long)cur.b * d.mbs : d.out, d.add ? nb4 : 0u);
                    const unsigned voff = (4u * kh * P + cur.pix) * 4u;
                    // the residual / mul / add operands of one 32-channel tile are requested as a batch before its
                    // stores: interleaved with the stores hipcc waits for every load separately (it cannot prove res
                    // and out distinct) - 64 memory round trips per tile at MT = 4, 92k of a tile's 228k cycles
                    // (tools/gemm_trace.py)
                    // batch size by register budget: MT = 1 lives on 128 registers (2 workgroups per CU); the wide LN3_GATE
                    // kernels sit at 256 already and keep the one-by-one form
                    // (this kernel keeps the runtime tests of d.act / d.epi per element: resolved per tile as in the other kernels, the
                    // freer schedule spills 33-110 registers at MT >= 4 and K = 345 -> 128 went from 0.56 to 0.96 ms)
                    constexpr int EB = MT == 1 ? 4 : 16;
                    if constexpr (EB == 1) {
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int nrow = nbase + m * 32 + (r & 3) + 8 * (r >> 2);   // + 4*kh per lane
                                const unsigned soff = (unsigned)nrow * P4;
                                float v = acc[m][r];
                                v += bl[nrow + 4 * kh];
                                v = apply_act(v, d.act);
                                if (d.epi == FDN_EPI_RES) v += bload(rr, voff, soff);
                                else if (d.epi == FDN_EPI_MULADD) v = v * bload(rm, voff, soff) + bload(rd, voff, soff);
                                bstore(v, ro, voff, soff);
                                acc[m][r] = (nrow + 4 * kh < N) ? v : 0.f;
                            }
                    } else
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
#pragma unroll
                        for (int r0 = 0; r0 < 16; r0 += EB) {
                            float l0[EB], l1[EB];
                            if (!EARLY && d.epi != FDN_EPI_NONE) {
#pragma unroll
                                for (int j = 0; j < EB; ++j) {
                                    const int r = r0 + j;
                                    const unsigned soff = (unsigned)(nbase + m * 32 + (r & 3) + 8 * (r >> 2)) * P4;
                                    if (d.epi == FDN_EPI_RES) l0[j] = bload(rr, voff, soff);
                                    else { l0[j] = bload(rm, voff, soff); l1[j] = bload(rd, voff, soff); }
                                }
                            }
#pragma unroll
                            for (int j = 0; j < EB; ++j) {
                                const int r = r0 + j;
                                const int nrow = nbase + m * 32 + (r & 3) + 8 * (r >> 2);   // + 4*kh per lane
                                const unsigned soff = (unsigned)nrow * P4;
                                float v = acc[m][r];
                                v += bl[nrow + 4 * kh];
                                v = apply_act(v, d.act);
                                if (d.epi == FDN_EPI_RES) v += EARLY ? e0[m * 16 + r] : l0[j];
                                else if (d.epi == FDN_EPI_MULADD) v = EARLY ? v * e0[m * 16 + r] + e1[m * 16 + r] : v * l0[j] + l1[j];
                                bstore(v, ro, voff, soff);          // rows >= N fall outside the descriptor
                                acc[m][r] = (nrow + 4 * kh < N) ? v : 0.f;
                            }
                            if (EB > 1) __builtin_amdgcn_sched_barrier(0);      // keep the next batch from being hoisted up here
                        }
                    }
                    if (d.stats_out && npass == 1) {
                        // channel LayerNorm statistics of the tile just written (two-pass, registers only):
                        // lane l and l^32 hold complementary rows of the same pixel
                        float sm = 0.f;
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int r = 0; r < 16; ++r) sm += acc[m][r];
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
                            float* sp = d.stats_out + (long)cur.b * 2 * P;
                            sp[cur.pix] = mean;
                            sp[P + cur.pix] = 1.0f / sqrtf(sq / (float)N + 1e-5f);
                        }
                    }
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
            }

            TRC(3)
            if (!g.resident) {
                if (nlive) w_stash((step + 1) & 1);
                TRC(4)
                __syncthreads();
            }
            TRC(5)
            // advance
            if (nlive && nc == 0) { cur = nxt; stats_load(cur); }
#pragma unroll
            for (int s = 0; s < 16; ++s) { xa[s] = xb[s]; ya[s] = yb[s]; }
            tile = ntile; c = nc; live = nlive;
            ++step;
        }
    }
}

// One accumulator, STEPS k-steps: acc += W[k][n] * x[k] with the A operand read from LDS eight k-steps
// ahead of its MFMA into a second register set (hipcc otherwise reuses one register pair and exposes
// the full ds_read latency every two MFMAs).  `w` points at this lane's element of k-step 0;
// consecutive k-steps are `stride` floats apart.
template <int STEPS>
__device__ __forceinline__ void mfma_chain(f32x16& acc, const float* w, int stride, const float* x) {
    static_assert(STEPS % 8 == 0, "k-steps come in groups of 8");
    float a[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[0][i] = w[i * stride];
#pragma unroll
    for (int g = 0; g < STEPS / 8; ++g) {
        if (g + 1 < STEPS / 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[(g + 1) & 1][i] = w[((g + 1) * 8 + i) * stride];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][i], x[g * 8 + i], acc, 0, 0, 0);
            // 1 MFMA then (while it runs) 1 LDS read of the next group: keep the two streams interleaved
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Small-K variant (K <= 128, whole transposed weight matrix resident in LDS): the wave keeps its
// 32-pixel activation strip for ALL K in registers (K/2 VGPRs), applies the prologue once, and then
// walks the output-channel tiles one at a time: 16*NCH MFMAs into a single 16-register accumulator,
// 16 stores, next tile.  Only 16 accumulator registers are live (5-6 waves per SIMD instead of 2),
// the stores of tile m drain while tile m+1's MFMAs issue, and there is no pass loop and no barrier
// after the weights are loaded.  Used for to_hidden / project_in (K = C) where N is 2.7-4.8 x K.
// ------------------------------------------------------------------------------------------------
template <int NCH, int PRO, int NW>
__global__ __launch_bounds__(NW * 64) void conv1x1_smallk_kernel(fdn_conv1x1_desc d, Geo g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = NW * 64;
    constexpr int Kp = NCH * KC;
    const int K = d.K, N = d.N;
    const unsigned P = (unsigned)d.P, P4 = P * 4u;
    const int ntiles = (N + 31) / 32;
    const int NS = ntiles * 32 + 1;
    float* tg = smem;
    float* tb = smem + Kp;
    float* Wl = smem + 2 * Kp;                 // [Kp][NS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;

    for (int i = tid; i < Kp; i += NT) {
        tg[i] = (PRO >= FDN_PRO_LN3_GATE && i < K) ? d.gamma[i] : 0.f;       // (PRO_LN: gamma / beta are folded into w / bias by the caller)
        tb[i] = (PRO >= FDN_PRO_LN3_GATE && i < K) ? d.beta[i] : 0.f;
    }
    // weights: W[n][k] (lanes along k) -> Wl[k][n]
    for (int idx = tid; idx < Kp * ntiles * 32; idx += NT) {
        const int k = idx % Kp, n = idx / Kp;
        Wl[k * NS + n] = (n < N && k < K) ? d.w[(long)n * K + k] : 0.f;
    }
    float* bl = Wl + Kp * NS;                  // bias[ntiles*32] (zeros without one): read from LDS in the epilogue - a global
    for (int i = tid; i < ntiles * 32; i += NT) bl[i] = (d.bias && i < N) ? d.bias[i] : 0.f;   // load next to the stores is waited for alone
    __syncthreads();

    const int ks0 = d.kseg[0], ks01 = d.kseg[0] + d.kseg[1];
    struct Tile { int b; unsigned pix; bool ok; };
    auto tile_setup = [&](int t) {
        Tile r;
        r.b = t / g.tiles_per_img;
        const unsigned p_ = (unsigned)(t - r.b * g.tiles_per_img) * (NW * 32) + wave * 32 + ln;
        r.ok = p_ < P;
        r.pix = r.ok ? p_ : P - 1;
        return r;
    };
    float xa[NCH * 16], xb[NCH * 16], yb[NCH * 16];
    float mu_n = 0.f, rs_n = 0.f;
    auto x_issue = [&](const Tile& t) {
        const rsrc_t r0 = mk_rsrc(d.x[0] + (long)t.b * d.xbs[0], (unsigned)ks0 * P4);
        const rsrc_t r1 = mk_rsrc(d.x[1] + (long)t.b * d.xbs[1], (unsigned)d.kseg[1] * P4);
        const rsrc_t r2 = mk_rsrc(d.x[2] + (long)t.b * d.xbs[2], (unsigned)d.kseg[2] * P4);
        const unsigned voff = (kh * P + t.pix) * 4u;
#pragma unroll
        for (int s = 0; s < NCH * 16; ++s) {
            const int k = 2 * s;
            if (k < ks0) xb[s] = bload(r0, voff, (unsigned)k * P4);
            else if (k < ks01) xb[s] = bload(r1, voff, (unsigned)(k - ks0) * P4);
            else xb[s] = bload(r2, voff, (unsigned)(k - ks01) * P4);
        }
        if (PRO == FDN_PRO_LN_MULADD) {
            const rsrc_t ry = mk_rsrc(d.xb + (long)t.b * d.xbbs, (unsigned)K * P4);
#pragma unroll
            for (int s = 0; s < NCH * 16; ++s) yb[s] = bload(ry, voff, (unsigned)(2 * s) * P4);
        }
        if (PRO != FDN_PRO_NONE) {
            const float* sp = d.stats + (long)t.b * 2 * P;
            mu_n = sp[t.pix];
            rs_n = sp[P + t.pix];
        }
    };

    int tile = blockIdx.x;
    bool live = tile < g.total_tiles;
    Tile cur = tile_setup(live ? tile : 0);
    if (live) x_issue(cur);
    while (live) {
        // take ownership of the prefetched strip, apply the prologue once
#pragma unroll
        for (int s = 0; s < NCH * 16; ++s) {
            float v = xb[s];
            if (PRO != FDN_PRO_NONE) {
                if (PRO == FDN_PRO_LN) {
                    v = (v - mu_n) * rs_n;
                } else {
                    const float ga = tg[2 * s + kh], be = tb[2 * s + kh];
                    v = (v - mu_n) * rs_n * ga + be;
                }
                if (PRO == FDN_PRO_LN_MULADD) v = v * yb[s] + yb[s];
            }
            xa[s] = v;
        }
        const int ntile = tile + gridDim.x;
        const bool nlive = ntile < g.total_tiles;
        const Tile nxt = tile_setup(nlive ? ntile : tile);
        if (nlive) x_issue(nxt);                                   // lands while this tile's MFMAs run

        const unsigned nb4 = (unsigned)N * P4;
        const rsrc_t ro = mk_rsrc(d.out + (long)cur.b * d.obs, nb4);
        const rsrc_t rr = mk_rsrc(d.res ? d.res + (long)cur.b * d.rbs : d.out, d.res ? nb4 : 0u);
        const rsrc_t rm = mk_rsrc(d.mul ? d.mul + (long)cur.b * d.mbs : d.out, d.mul ? nb4 : 0u);
        const rsrc_t rd = mk_rsrc(d.add ? d.add + (long)cur.b * d.mbs : d.out, d.add ? nb4 : 0u);
        const unsigned voff = (4u * kh * P + cur.pix) * 4u;
        for (int m = 0; m < ntiles; ++m) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            mfma_chain<NCH * 16>(acc, Wl + kh * NS + m * 32 + ln, 2 * NS, xa);
            auto epilogue = [&](auto act_c, auto epi_c) __attribute__((always_inline)) {
                FDN_EPI_MODES(act_c, epi_c)
            if (cur.ok) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nrow = m * 32 + (r & 3) + 8 * (r >> 2);
                    const unsigned soff = (unsigned)nrow * P4;
                    float v = acc[r];
                    v += bl[nrow + 4 * kh];
                    v = apply_act(v, act_);
                    if (epi_ == FDN_EPI_RES) v += bload(rr, voff, soff);
                    else if (epi_ == FDN_EPI_MULADD) v = v * bload(rm, voff, soff) + bload(rd, voff, soff);
                    bstore(v, ro, voff, soff);
                }
            }
            };
            FDN_EPI_DISPATCH(epilogue);
        }
        cur = nxt; tile = ntile; live = nlive;
    }
}

// Vectorised small-K variant (the HBM-bound level-1/2 to_hidden / project_in convs): a lane owns VEC consecutive
// pixels, so every buffer load / store moves 16 (VEC = 4) or 8 (VEC = 2) bytes per lane and a half-wave touches 512 /
// 256 contiguous bytes of a channel plane instead of 128 - the dword form stops at ~3.8 TB/s, 16-byte lanes reach ~5
// (same effect as in norm.hip).  Register v of a loaded vector is the B operand of MFMA chain v; the VEC chains share
// every A operand read from LDS.  Plain / LN prologue, no epilogue operand, one input segment, P % VEC == 0.
template <int VEC> struct VecT;
template <> struct VecT<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecT<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type bloadv(rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned uv __attribute__((ext_vector_type(VEC)));
    uv u;
    if constexpr (VEC == 4) u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    else u = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    typename VecT<VEC>::type f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = __uint_as_float(u[i]);
    return f;
}
template <int VEC>
__device__ __forceinline__ void bstorev(typename VecT<VEC>::type f, rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned uv __attribute__((ext_vector_type(VEC)));
    uv u;
#pragma unroll
    for (int i = 0; i < VEC; ++i) u[i] = __float_as_uint(f[i]);
    if constexpr (VEC == 4) __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(u, r, voff, soff, 0);
}

// the same for an activation tensor kept in bf16 STORAGE (BF = true: VEC bf16 values = VEC * 2 bytes per lane, widened to fp32 /
// rounded to nearest-even; the arithmetic is fp32 either way)
template <int VEC, bool BF>
__device__ __forceinline__ typename VecT<VEC>::type sloadv(rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (!BF) return bloadv<VEC>(r, voff, soff);
    else {
        static_assert(VEC == 2, "bf16 storage: pixel pairs");
        float v[2];
        st_load2<true>(v, r, voff, soff);
        typename VecT<VEC>::type f;
        f[0] = v[0];
        f[1] = v[1];
        return f;
    }
}
template <int VEC, bool BF>
__device__ __forceinline__ void sstorev(typename VecT<VEC>::type f, rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (!BF) bstorev<VEC>(f, r, voff, soff);
    else {
        static_assert(VEC == 2, "bf16 storage: pixel pairs");
        const float v[2] = {f[0], f[1]};
        st_store2<true>(v, r, voff, soff);
    }
}
__device__ __forceinline__ const float* byte_advance(const float* p, long bytes) {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + bytes);
}

// XBF: x[0] is stored as bf16; OBF: out is stored as bf16 (non-TAIL form).  Statistics, residual and TAIL output stay fp32.
template <int NCH, int PRO, int VEC, bool TAIL, bool XBF = false, bool OBF = false>
__global__ __launch_bounds__(256, 2) void conv1x1_smallk_vec_kernel(fdn_conv1x1_desc d, Geo g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(VEC == 2, "only the 8-byte-lane form is validated: a 16-byte-lane build gave wrong results at full size (round 1) and is not shipped");
    typedef typename VecT<VEC>::type vf;
    constexpr int NT = 256, NWV = 4;
    constexpr int Kp = NCH * KC, KS = NCH * 16;
    constexpr unsigned XES = st_bytes<XBF>(), OES = st_bytes<OBF>();
    const int K = d.K, N = d.N;
    const unsigned P = (unsigned)d.P, P4 = P * 4u, PX = P * XES, PO = P * OES;
    const int ntiles = (N + 31) / 32;
    const int NS = ntiles * 32 + 1;
    float* tg = smem;
    float* tb = smem + Kp;
    float* Wl = smem + 2 * Kp;                 // [Kp][NS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;

    for (int i = tid; i < Kp; i += NT) {
        tg[i] = (PRO >= FDN_PRO_LN3_GATE && i < K) ? d.gamma[i] : 0.f;       // (PRO_LN: gamma / beta are folded into w / bias by the caller)
        tb[i] = (PRO >= FDN_PRO_LN3_GATE && i < K) ? d.beta[i] : 0.f;
    }
    for (int idx = tid; idx < Kp * ntiles * 32; idx += NT) {
        const int k = idx % Kp, n = idx / Kp;
        Wl[k * NS + n] = (n < N && k < K) ? d.w[(long)n * K + k] : 0.f;
    }
    float* bl = Wl + Kp * NS;                  // bias[ntiles*32] (zeros without one): read from LDS in the epilogue - a global
    for (int i = tid; i < ntiles * 32; i += NT) bl[i] = (d.bias && i < N) ? d.bias[i] : 0.f;   // load next to the stores is waited for alone
    __syncthreads();

    struct Tile { int b; unsigned pix; bool ok; };
    auto tile_setup = [&](int t) {
        Tile r;
        r.b = t / g.tiles_per_img;
        const unsigned p_ = (unsigned)(t - r.b * g.tiles_per_img) * (NWV * 32 * VEC) + (wave * 32 + ln) * VEC;
        r.ok = p_ < P;                          // P % VEC == 0: a vector is inside or outside as a whole
        r.pix = r.ok ? p_ : P - VEC;
        return r;
    };
    // ONE register set for the activation strip: the next tile's element s is requested into xa[s] right after the last
    // MFMA group that reads it (during the last output tile), so no second (prefetch) buffer is live - with the LN
    // prologue a double buffer does not fit two waves per SIMD at 16-byte lanes
    vf xa[KS];
    vf mu_n, rs_n;
    auto stats_issue = [&](const Tile& t) {
        if (PRO != FDN_PRO_NONE) {
            const rsrc_t rs_ = mk_rsrc(d.stats + (long)t.b * 2 * P, 2u * P4);
            mu_n = bloadv<VEC>(rs_, t.pix * 4u, 0u);
            rs_n = bloadv<VEC>(rs_, t.pix * 4u, P4);
        }
    };

    int tile = blockIdx.x;
    bool live = tile < g.total_tiles;
    Tile cur = tile_setup(live ? tile : 0);
    if (live) {
        const rsrc_t r0 = mk_rsrc(byte_advance(d.x[0], (long)cur.b * d.xbs[0] * XES), (unsigned)K * PX);
        const unsigned voff = (kh * P + cur.pix) * XES;
#pragma unroll
        for (int s = 0; s < KS; ++s) xa[s] = sloadv<VEC, XBF>(r0, voff, (unsigned)(2 * s) * PX);      // k >= K reads 0
        stats_issue(cur);
    }
    while (live) {
        if (PRO != FDN_PRO_NONE) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                asm volatile("" ::: "memory");      // keeps the table reads here: hoisted out of the persistent tile loop (with
                                                      // rs*gamma products precomputed per element) they cost ~130 registers
                xa[s] = (xa[s] - mu_n) * rs_n;                       // PRO_LN: gamma / beta live in w / bias
            }
        }
        const int ntile = tile + gridDim.x;
        const bool nlive = ntile < g.total_tiles;
        const Tile nxt = tile_setup(nlive ? ntile : tile);
        const rsrc_t rn = mk_rsrc(byte_advance(d.x[0], (long)nxt.b * d.xbs[0] * XES), (unsigned)K * PX);
        const unsigned voffn = (kh * P + nxt.pix) * XES;
        if (nlive) stats_issue(nxt);                               // (mu, rstd) of the next tile: consumed after this one

        const rsrc_t ro = mk_rsrc(byte_advance(d.out, (long)cur.b * d.obs * OES), (unsigned)N * PO);
        const unsigned voff = cur.ok ? (4u * kh * P + cur.pix) * 4u : 0x80000000u;     // outside pixels: stores dropped (fp32 operands)
        const unsigned voffo = cur.ok ? (4u * kh * P + cur.pix) * OES : 0x80000000u;   // the same in bytes of the output's storage type
        for (int m = 0; m < ntiles; ++m) {
            const bool refill = nlive && m == ntiles - 1;
            f32x16 acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;
            // TAIL: the residual of this tile is requested before the MFMA groups, so its latency hides behind them (after
            // them it was the one exposed round trip of every tile)
            vf rres[TAIL ? 16 : 1];
            if (TAIL && d.epi == FDN_EPI_RES) {
                const rsrc_t rr = mk_rsrc(d.res + (long)cur.b * d.rbs, (unsigned)N * P4);
#pragma unroll
                for (int r = 0; r < (TAIL ? 16 : 1); ++r) rres[r] = bloadv<VEC>(rr, voff, (unsigned)((r & 3) + 8 * (r >> 2)) * P4);
            }
            // A operands one group of 8 k-steps ahead (as mfma_chain); each feeds VEC MFMAs
            const float* w = Wl + kh * NS + m * 32 + ln;
            float a[2][8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[0][i] = w[i * 2 * NS];
#pragma unroll
            for (int grp = 0; grp < KS / 8; ++grp) {
                if (grp + 1 < KS / 8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) a[(grp + 1) & 1][i] = w[((grp + 1) * 8 + i) * 2 * NS];
                }
                __builtin_amdgcn_sched_barrier(0);      // (reads stay one group ahead of their use, see conv1x1_smallk_stream_vec_kernel)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[grp & 1][i], xa[grp * 8 + i][v], acc[v], 0, 0, 0);
                if (refill) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) xa[grp * 8 + i] = sloadv<VEC, XBF>(rn, voffn, (unsigned)(2 * (grp * 8 + i)) * PX);
                }
            }
            auto epilogue = [&](auto act_c, auto epi_c) __attribute__((always_inline)) {
                FDN_EPI_MODES(act_c, epi_c)
            if constexpr (!TAIL) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nrow = m * 32 + (r & 3) + 8 * (r >> 2);
                    vf o;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o[v] = acc[v][r];
                    o += bl[nrow + 4 * kh];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o[v] = apply_act(o[v], act_);
                    sstorev<VEC, OBF>(o, ro, voffo, (unsigned)nrow * PO);        // rows >= N fall outside the descriptor
                }
            } else {
                // TAIL (N <= 32, the narrow project_out convs): residual as one batch of vector loads, then the stores and
                // the channel-LayerNorm statistics of the result (two-pass, registers only; lanes l and l^32 hold
                // complementary rows of the same pixels)
                vf outv[16];                    // (rres: requested ahead of the MFMA groups, below)
                vf sm = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nrow = (r & 3) + 8 * (r >> 2);
                    vf o;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o[v] = acc[v][r];
                    o += bl[nrow + 4 * kh];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o[v] = apply_act(o[v], act_);
                    if (epi_ == FDN_EPI_RES) o += rres[r];
                    bstorev<VEC>(o, ro, voff, (unsigned)nrow * P4);
                    outv[r] = (nrow + 4 * kh < N) ? o : vf(0.f);
                    sm += outv[r];
                }
                if (d.stats_out) {
                    vf mean, sq = 0.f;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) mean[v] = (sm[v] + __shfl_xor(sm[v], 32)) / (float)N;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const vf dl = outv[r] - mean;
                        sq += ((r & 3) + 8 * (r >> 2) + 4 * kh < N) ? dl * dl : vf(0.f);
                    }
                    vf rstd;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) rstd[v] = 1.0f / sqrtf((sq[v] + __shfl_xor(sq[v], 32)) / (float)N + 1e-5f);
                    if (kh == 0) {
                        const rsrc_t rs_ = mk_rsrc(d.stats_out + (long)cur.b * 2 * P, 2u * P4);
                        const unsigned vs = cur.ok ? cur.pix * 4u : 0x80000000u;
                        bstorev<VEC>(mean, rs_, vs, 0u);
                        bstorev<VEC>(rstd, rs_, vs, P4);
                    }
                }
            }
            };
            if constexpr (TAIL) FDN_EPI_DISPATCH(epilogue); else FDN_ACT_DISPATCH(epilogue);
        }
        cur = nxt; tile = ntile; live = nlive;
    }
}

// Small-K, large-N variant whose weight matrix does NOT fit LDS (level 3: 128 -> 612 / 345): same
// register-resident activation strip, but the 32-channel weight tiles stream through a double
// buffer in LDS (one barrier per output tile).  The strip of the NEXT pixel tile is requested a
// whole tile (ntiles x 64*NCH MFMA cycles) ahead, the next weight tile one step ahead, so neither
// HBM nor L2 latency is exposed to the MFMA pipe.
template <int NCH, int PRO, int NW>
__global__ __launch_bounds__(NW * 64) void conv1x1_smallk_stream_kernel(fdn_conv1x1_desc d, Geo g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = NW * 64;
    constexpr int Kp = NCH * KC;
    constexpr int WS = 33;
    constexpr int WPT = (Kp * 32) / NT;
    const int K = d.K, N = d.N;
    const unsigned P = (unsigned)d.P, P4 = P * 4u;
    const int ntiles = (N + 31) / 32;
    float* tg = smem;
    float* tb = smem + Kp;
    float* Wl = smem + 2 * Kp;                 // [2][Kp][WS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;

    for (int i = tid; i < Kp; i += NT) {
        tg[i] = (PRO >= FDN_PRO_LN3_GATE && i < K) ? d.gamma[i] : 0.f;       // (PRO_LN: gamma / beta are folded into w / bias by the caller)
        tb[i] = (PRO >= FDN_PRO_LN3_GATE && i < K) ? d.beta[i] : 0.f;
    }
    float wr[WPT];
    // element idx = tid + NT*i of the [32 n][Kp k] weight tile (k fastest: coalesced rows of W[n][:])
    auto w_fetch = [&](int m) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + NT * i;
            const int k = idx % Kp, n = m * 32 + idx / Kp;
            wr[i] = (n < N && k < K) ? d.w[(long)n * K + k] : 0.f;
        }
    };
    auto w_stash = [&](int buf) {
        float* dst = Wl + buf * (Kp * WS);
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + NT * i;
            dst[(idx % Kp) * WS + idx / Kp] = wr[i];
        }
    };
    w_fetch(0);
    w_stash(0);
    float* bl = Wl + 2 * Kp * WS;              // bias[ntiles*32] (zeros without one), read from LDS in the epilogue
    for (int i = tid; i < ntiles * 32; i += NT) bl[i] = (d.bias && i < N) ? d.bias[i] : 0.f;
    __syncthreads();

    struct Tile { int b; unsigned pix; bool ok; };
    auto tile_setup = [&](int t) {
        Tile r;
        r.b = t / g.tiles_per_img;
        const unsigned p_ = (unsigned)(t - r.b * g.tiles_per_img) * (NW * 32) + wave * 32 + ln;
        r.ok = p_ < P;
        r.pix = r.ok ? p_ : P - 1;
        return r;
    };
    float xa[NCH * 16], xb[NCH * 16];
    float mu_n = 0.f, rs_n = 0.f;
    auto x_issue = [&](const Tile& t) {
        const rsrc_t r0 = mk_rsrc(d.x[0] + (long)t.b * d.xbs[0], (unsigned)K * P4);
        const unsigned voff = (kh * P + t.pix) * 4u;
#pragma unroll
        for (int s = 0; s < NCH * 16; ++s) xb[s] = bload(r0, voff, (unsigned)(2 * s) * P4);
        if (PRO != FDN_PRO_NONE) {
            const float* sp = d.stats + (long)t.b * 2 * P;
            mu_n = sp[t.pix];
            rs_n = sp[P + t.pix];
        }
    };

    int tile = blockIdx.x;
    bool live = tile < g.total_tiles;
    Tile cur = tile_setup(live ? tile : 0);
    if (live) x_issue(cur);
    int wstep = 0;                              // running output-tile counter -> weight buffer parity
    while (live) {
#pragma unroll
        for (int s = 0; s < NCH * 16; ++s) {
            float v = xb[s];
            if (PRO != FDN_PRO_NONE) v = (v - mu_n) * rs_n;
            xa[s] = v;
        }
        const int ntile = tile + gridDim.x;
        const bool nlive = ntile < g.total_tiles;
        const Tile nxt = tile_setup(nlive ? ntile : tile);
        if (nlive) x_issue(nxt);

        const unsigned nb4 = (unsigned)N * P4;
        const rsrc_t ro = mk_rsrc(d.out + (long)cur.b * d.obs, nb4);
        const rsrc_t rr = mk_rsrc(d.res ? d.res + (long)cur.b * d.rbs : d.out, d.res ? nb4 : 0u);
        const unsigned voff = (4u * kh * P + cur.pix) * 4u;
        for (int m = 0; m < ntiles; ++m, ++wstep) {
            const int mnext = (m + 1 < ntiles) ? m + 1 : 0;
            const bool more = (m + 1 < ntiles) || nlive;
            if (more) w_fetch(mnext);                              // next weight tile: L2 -> registers
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            mfma_chain<NCH * 16>(acc, Wl + (wstep & 1) * (Kp * WS) + kh * WS + ln, 2 * WS, xa);
            auto epilogue = [&](auto act_c, auto epi_c) __attribute__((always_inline)) {
                FDN_EPI_MODES(act_c, epi_c)
            if (cur.ok) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nrow = m * 32 + (r & 3) + 8 * (r >> 2);
                    const unsigned soff = (unsigned)nrow * P4;
                    float v = acc[r];
                    v += bl[nrow + 4 * kh];
                    v = apply_act(v, act_);
                    if (epi_ == FDN_EPI_RES) v += bload(rr, voff, soff);
                    bstore(v, ro, voff, soff);
                }
            }
            };
            FDN_EPI_DISPATCH(epilogue);
            if (more) w_stash((wstep + 1) & 1);
            __syncthreads();
        }
        cur = nxt; tile = ntile; live = nlive;
    }
}

// 8-byte-lane form of the streaming small-K kernel (level-3 to_hidden / project_in, K <= 128): a lane owns two consecutive
// pixels, the two MFMA chains share every A operand, ONE register set holds the strip (refilled for the next pixel tile
// during the last output tile, as in conv1x1_smallk_vec_kernel; the dword kernel above keeps two sets and spills 88
// registers at K = 128 with the LN prologue).  Plain / LN prologue, no epilogue operand.
template <int NCH, int PRO>
__global__ __launch_bounds__(512) void conv1x1_smallk_stream_vec_kernel(fdn_conv1x1_desc d, Geo g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int VEC = 2, NW = 8, NT = NW * 64;
    typedef typename VecT<VEC>::type vf;
    constexpr int Kp = NCH * KC, KS = NCH * 16;
    constexpr int WS = 33;
    constexpr int WPT = (Kp * 32) / NT;
    const int K = d.K, N = d.N;
    const unsigned P = (unsigned)d.P, P4 = P * 4u;
    const int ntiles = (N + 31) / 32;
    float* tg = smem;
    float* tb = smem + Kp;
    float* Wl = smem + 2 * Kp;                 // [2][Kp][WS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;

    for (int i = tid; i < Kp; i += NT) {
        tg[i] = (PRO >= FDN_PRO_LN3_GATE && i < K) ? d.gamma[i] : 0.f;       // (PRO_LN: gamma / beta are folded into w / bias by the caller)
        tb[i] = (PRO >= FDN_PRO_LN3_GATE && i < K) ? d.beta[i] : 0.f;
    }
    float wr[WPT];
    auto w_fetch = [&](int m) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + NT * i;
            const int k = idx % Kp, n = m * 32 + idx / Kp;
            wr[i] = (n < N && k < K) ? d.w[(long)n * K + k] : 0.f;
        }
    };
    auto w_stash = [&](int buf) {
        float* dst = Wl + buf * (Kp * WS);
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + NT * i;
            dst[(idx % Kp) * WS + idx / Kp] = wr[i];
        }
    };
    w_fetch(0);
    w_stash(0);
    float* bl = Wl + 2 * Kp * WS;              // bias[ntiles*32] (zeros without one), read from LDS in the epilogue
    for (int i = tid; i < ntiles * 32; i += NT) bl[i] = (d.bias && i < N) ? d.bias[i] : 0.f;
    __syncthreads();

    struct Tile { int b; unsigned pix; bool ok; };
    auto tile_setup = [&](int t) {
        Tile r;
        r.b = t / g.tiles_per_img;
        const unsigned p_ = (unsigned)(t - r.b * g.tiles_per_img) * (NW * 32 * VEC) + (wave * 32 + ln) * VEC;
        r.ok = p_ < P;
        r.pix = r.ok ? p_ : P - VEC;
        return r;
    };
    vf xa[KS];
    vf mu_n, rs_n;
    auto stats_issue = [&](const Tile& t) {
        if (PRO != FDN_PRO_NONE) {
            const rsrc_t rs_ = mk_rsrc(d.stats + (long)t.b * 2 * P, 2u * P4);
            mu_n = bloadv<VEC>(rs_, t.pix * 4u, 0u);
            rs_n = bloadv<VEC>(rs_, t.pix * 4u, P4);
        }
    };
    int tile = blockIdx.x;
    bool live = tile < g.total_tiles;
    Tile cur = tile_setup(live ? tile : 0);
    if (live) {
        const rsrc_t r0 = mk_rsrc(d.x[0] + (long)cur.b * d.xbs[0], (unsigned)K * P4);
        const unsigned voff = (kh * P + cur.pix) * 4u;
#pragma unroll
        for (int s = 0; s < KS; ++s) xa[s] = bloadv<VEC>(r0, voff, (unsigned)(2 * s) * P4);
        stats_issue(cur);
    }
    int wstep = 0;
#ifdef FDN_GEMM_TRACE   // tools/gemm_trace2.py: s_memtime stamps per output tile (waves 0 and 4 of workgroup 0) into d.mul (vec4 = 12345)
    unsigned long long* trc2 = (d.vec4 == 12345 && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0)
                                   ? reinterpret_cast<unsigned long long*>(const_cast<float*>(d.mul)) + (wave ? 1024 : 0) : nullptr;
#define TR2(i) if (trc2 && wstep < 120) trc2[wstep * 8 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define TR2(i)
#endif
    while (live) {
        TR2(6)
        if (PRO != FDN_PRO_NONE) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                asm volatile("" ::: "memory");                     // table reads stay inside the tile loop
                xa[s] = (xa[s] - mu_n) * rs_n;
            }
        }
        const int ntile = tile + gridDim.x;
        const bool nlive = ntile < g.total_tiles;
        const Tile nxt = tile_setup(nlive ? ntile : tile);
        const rsrc_t rn = mk_rsrc(d.x[0] + (long)nxt.b * d.xbs[0], (unsigned)K * P4);
        const unsigned voffn = (kh * P + nxt.pix) * 4u;
        if (nlive) stats_issue(nxt);
        const rsrc_t ro = mk_rsrc(d.out + (long)cur.b * d.obs, (unsigned)N * P4);
        const unsigned voff = cur.ok ? (4u * kh * P + cur.pix) * 4u : 0x80000000u;
        for (int m = 0; m < ntiles; ++m, ++wstep) {
            const int mnext = (m + 1 < ntiles) ? m + 1 : 0;
            const bool more = (m + 1 < ntiles) || nlive;
            const bool refill = nlive && m == ntiles - 1;
            TR2(0)
            if (more) w_fetch(mnext);                              // next weight tile: L2 -> registers
            f32x16 acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;
            const float* w = Wl + (wstep & 1) * (Kp * WS) + kh * WS + ln;
            float a[2][8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[0][i] = w[i * 2 * WS];
#pragma unroll
            for (int grp = 0; grp < KS / 8; ++grp) {
                if (grp + 1 < KS / 8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) a[(grp + 1) & 1][i] = w[((grp + 1) * 8 + i) * 2 * WS];
                }
                __builtin_amdgcn_sched_barrier(0);      // the reads stay a whole group (16 MFMAs) ahead of their use: left alone the
                                                        // scheduler sinks them next to it and the LDS latency stalls the matrix pipe
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[grp & 1][i], xa[grp * 8 + i][v], acc[v], 0, 0, 0);
                if (refill) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) xa[grp * 8 + i] = bloadv<VEC>(rn, voffn, (unsigned)(2 * (grp * 8 + i)) * P4);
                }
            }
            TR2(1)
            auto epilogue = [&](auto act_c, auto epi_c) __attribute__((always_inline)) {
                FDN_EPI_MODES(act_c, epi_c)
                (void)epi_;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = m * 32 + (r & 3) + 8 * (r >> 2);
                vf o;
#pragma unroll
                for (int v = 0; v < VEC; ++v) o[v] = acc[v][r];
                o += bl[nrow + 4 * kh];
#pragma unroll
                for (int v = 0; v < VEC; ++v) o[v] = apply_act(o[v], act_);
                bstorev<VEC>(o, ro, voff, (unsigned)nrow * P4);
            }
            };
            FDN_ACT_DISPATCH(epilogue);
            TR2(2)
            if (more) w_stash((wstep + 1) & 1);
            TR2(3)
            __syncthreads();
            TR2(4)
        }
        cur = nxt; tile = ntile; live = nlive;
    }
}

// 8-byte-lane form of the K-streaming kernel for the plain deep-K convs (FDFFN project_out at levels 2/3, Fuse):
// a lane owns two consecutive pixels, so each A operand read from LDS feeds two MFMA chains per output tile and every
// load / store moves 8 bytes per lane.  One input segment, no prologue, residual or no epilogue operand, statistics
// of the result when N fits one pass.  Weights resident in LDS or double-buffered per 32-deep chunk, as the generic kernel.
template <int MT, bool XBF = false>          // XBF: x[0] is stored as bf16
__global__ __launch_bounds__(256, 2) void conv1x1_kstream_vec_kernel(fdn_conv1x1_desc d, Geo g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int VEC = 2, NW = 4, NT = NW * 64;
    constexpr unsigned XES = st_bytes<XBF>();
    typedef typename VecT<VEC>::type vf;
    constexpr int WS = MT * 32 + 1;
    constexpr int CH = KC * WS;
    constexpr int WPT = (KC * MT * 32) / NT;
    const int K = d.K, N = d.N;
    const unsigned P = (unsigned)d.P, P4 = P * 4u;
    const int nch = (K + KC - 1) / KC;
    float* Wl = smem;
    float* bl = smem + g.bias_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, ln = lane & 31;
    for (int i = tid; i < MT * 32; i += NT) bl[i] = (d.bias && i < N) ? d.bias[i] : 0.f;

    float wr[WPT];
    auto w_fetch = [&](int c) {
        const int kk = tid & 31, k = c * KC + kk;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int n = (tid >> 5) + (NT / 32) * i;
            wr[i] = (n < N && k < K) ? d.w[(long)n * K + k] : 0.f;
        }
    };
    auto w_stash = [&](int buf) {
        float* dst = Wl + buf * CH + (tid & 31) * WS;
#pragma unroll
        for (int i = 0; i < WPT; ++i) dst[(tid >> 5) + (NT / 32) * i] = wr[i];
    };
    if (g.resident) {
        for (int c = 0; c < nch; ++c) { w_fetch(c); w_stash(c); }
    } else {
        w_fetch(0);
        w_stash(0);
    }
    __syncthreads();

    struct Tile { int b; unsigned pix; bool ok; };
    auto tile_setup = [&](int t) {
        Tile r;
        r.b = t / g.tiles_per_img;
        const unsigned p_ = (unsigned)(t - r.b * g.tiles_per_img) * (NW * 32 * VEC) + (wave * 32 + ln) * VEC;
        r.ok = p_ < P;
        r.pix = r.ok ? p_ : P - VEC;
        return r;
    };
    vf xa[16], xb[16];
    auto x_issue = [&](const Tile& t, int c_, vf (&xv)[16]) {
        const rsrc_t r0 = mk_rsrc(byte_advance(d.x[0], (long)t.b * d.xbs[0] * XES), (unsigned)K * P * XES);
        const unsigned voff = (kh * P + t.pix) * XES;
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) xv[s_] = sloadv<VEC, XBF>(r0, voff, (unsigned)(c_ * KC + 2 * s_) * P * XES);   // k >= K reads 0
    };
    f32x16 acc[MT][VEC];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < VEC; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][v][r] = 0.f;

    int tile = blockIdx.x, c = 0, step = 0;
    bool live = tile < g.total_tiles;
    Tile cur = tile_setup(live ? tile : 0);
    if (live) x_issue(cur, 0, xa);
    while (live) {
        int ntile = tile, nc = c + 1;
        if (nc == nch) { nc = 0; ntile = tile + gridDim.x; }
        const bool nlive = ntile < g.total_tiles;
        Tile nxt = cur;
        if (nlive) {
            if (nc == 0) nxt = tile_setup(ntile);
            x_issue(nxt, nc, xb);
            if (!g.resident) w_fetch(nc);
        }
        const float* Wc = Wl + (g.resident ? c : (step & 1)) * CH;
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            const float* wrow = Wc + (2 * s_ + kh) * WS + ln;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float a = wrow[m * 32];
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[m][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xa[s_][v], acc[m][v], 0, 0, 0);
            }
        }
        if (c == nch - 1) {
            const unsigned nb4 = (unsigned)N * P4;
            const rsrc_t ro = mk_rsrc(d.out + (long)cur.b * d.obs, nb4);
            const rsrc_t rr = mk_rsrc(d.res ? d.res + (long)cur.b * d.rbs : d.out, d.res ? nb4 : 0u);
            const unsigned voff = cur.ok ? (4u * kh * P + cur.pix) * 4u : 0x80000000u;
            vf sm = 0.f;
            auto epilogue = [&](auto act_c, auto epi_c) __attribute__((always_inline)) {
                FDN_EPI_MODES(act_c, epi_c)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                vf l0[16];                                              // residual of one 32-channel tile as a batch
                if (epi_ == FDN_EPI_RES) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) l0[r] = bloadv<VEC>(rr, voff, (unsigned)(m * 32 + (r & 3) + 8 * (r >> 2)) * P4);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nrow = m * 32 + (r & 3) + 8 * (r >> 2);
                    vf o;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o[v] = acc[m][v][r];
                    o += bl[nrow + 4 * kh];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o[v] = apply_act(o[v], act_);
                    if (epi_ == FDN_EPI_RES) o += l0[r];
                    bstorev<VEC>(o, ro, voff, (unsigned)nrow * P4);
                    o = (nrow + 4 * kh < N) ? o : vf(0.f);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[m][v][r] = o[v];
                    sm += o;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            };
            if constexpr (MT <= 2) FDN_ACT_DISPATCH(epilogue); else epilogue(IC<1>(), IC<-1>());      // (MT = 3 spills 41 registers with the activation resolved)
            if (d.stats_out) {
                vf mean, sq = 0.f, rstd;
#pragma unroll
                for (int v = 0; v < VEC; ++v) mean[v] = (sm[v] + __shfl_xor(sm[v], 32)) / (float)N;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool in = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh < N;
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const float dl = acc[m][v][r] - mean[v];
                            sq[v] += in ? dl * dl : 0.f;
                        }
                    }
#pragma unroll
                for (int v = 0; v < VEC; ++v) rstd[v] = 1.0f / sqrtf((sq[v] + __shfl_xor(sq[v], 32)) / (float)N + 1e-5f);
                if (kh == 0) {
                    const rsrc_t rs_ = mk_rsrc(d.stats_out + (long)cur.b * 2 * P, 2u * P4);
                    const unsigned vs = cur.ok ? cur.pix * 4u : 0x80000000u;
                    bstorev<VEC>(mean, rs_, vs, 0u);
                    bstorev<VEC>(rstd, rs_, vs, P4);
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int v = 0; v < VEC; ++v)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][v][r] = 0.f;
        }
        if (!g.resident) {
            if (nlive) w_stash((step + 1) & 1);
            __syncthreads();
        }
        if (nlive && nc == 0) cur = nxt;
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) xa[s_] = xb[s_];
        tile = ntile; c = nc; live = nlive;
        ++step;
    }
}

int pick_mt(int N) {
    // fewest computed 32-row tiles, then fewest passes; MT <= 5 keeps the accumulator at 80 VGPRs
    const int tiles = (N + 31) / 32;
    int best = 1, best_cost = 1 << 30;
    for (int mt = 1; mt <= 5; ++mt) {
        const int passes = (tiles + mt - 1) / mt;
        const int cost = passes * mt * 100 + passes;
        if (cost < best_cost || (cost == best_cost && mt > best)) { best_cost = cost; best = mt; }
    }
    return best;
}

// launch-time facts are cached per (kernel, device) in capi.hip: no occupancy / attribute query on the hot path
#define FDN_NUM_CU_OR_FAIL()               \
    const int g_num_cu = fdn_device_cus(); \
    if (g_num_cu <= 0) return FDN_ERR_LAUNCH;

template <int MT, int PRO, int NW, bool EARLY>
int launch(const fdn_conv1x1_desc& d, hipStream_t s) {
    const int nch = PRO == FDN_PRO_LN3_GATE ? ((d.ln_group + 1) / 2 + 4) / 5 : (d.K + KC - 1) / KC;   // as in the kernel
    const size_t tab = 2UL * nch * KC * sizeof(float);
    const size_t chunk = (size_t)KC * (MT * 32 + 1) * sizeof(float);
    Geo g;
    g.resident = (tab + nch * chunk <= 96 * 1024) ? 1 : 0;
    size_t lds = tab + (g.resident ? nch : 2) * chunk;
    g.bias_off = (int)(lds / sizeof(float));
    lds += (size_t)(((d.N + 31) & ~31) + 32 * 5) * sizeof(float);     // bias table (+ the rows a partial last pass still indexes)
    g.tiles_per_img = cdiv(d.P, NW * 32);
    g.total_tiles = d.B * g.tiles_per_img;
    auto kern = conv1x1_kernel<MT, PRO, NW, EARLY>;
    if (lds > 48 * 1024 && !fdn_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return FDN_ERR_LAUNCH;
    FDN_NUM_CU_OR_FAIL()
    // persistent grid: as many workgroups as can be co-resident (LDS / register limited), capped by the work
    int per_cu = 0;
    if (!fdn_occupancy(&per_cu, reinterpret_cast<const void*>(kern), NW * 64, lds)) per_cu = 1;
    if (per_cu * NW > 16) per_cu = 16 / NW;                     // 4 waves per SIMD are enough to hide the loads
    if (per_cu < 1) per_cu = 1;
    int grid = g_num_cu * per_cu;
    if (grid > g.total_tiles) grid = g.total_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, d, g);
    return fdn_launch_status();
}

template <int NCH, int PRO>
int launch_smallk(const fdn_conv1x1_desc& d, hipStream_t s) {
    const int ntiles = (d.N + 31) / 32;
    const size_t lds = (2UL * NCH * KC + (size_t)NCH * KC * (ntiles * 32 + 1) + ntiles * 32) * sizeof(float);
    FDN_NUM_CU_OR_FAIL()
    constexpr int NW = 8;
    Geo g;
    g.resident = 1;
    g.tiles_per_img = cdiv(d.P, NW * 32);
    g.total_tiles = d.B * g.tiles_per_img;
    auto kern = conv1x1_smallk_kernel<NCH, PRO, NW>;
    if (lds > 48 * 1024 && !fdn_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return FDN_ERR_LAUNCH;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 3) per_cu = 3;
    if (per_cu < 1) per_cu = 1;
    int grid = g_num_cu * per_cu;
    if (grid > g.total_tiles) grid = g.total_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, d, g);
    return fdn_launch_status();
}

template <int NCH, int PRO>
int launch_smallk_stream(const fdn_conv1x1_desc& d, hipStream_t s) {
    constexpr int NW = 8;
    const size_t lds = (2UL * NCH * KC + 2UL * NCH * KC * 33 + ((d.N + 31) / 32) * 32) * sizeof(float);
    FDN_NUM_CU_OR_FAIL()
    Geo g;
    g.resident = 0;
    g.tiles_per_img = cdiv(d.P, NW * 32);
    g.total_tiles = d.B * g.tiles_per_img;
    int grid = g_num_cu;                                           // 8 waves x ~190 VGPRs: one workgroup per CU
    if (grid > g.total_tiles) grid = g.total_tiles;
    hipLaunchKernelGGL((conv1x1_smallk_stream_kernel<NCH, PRO, NW>), dim3(grid), dim3(NW * 64), lds, s, d, g);
    return fdn_launch_status();
}

template <int NCH, int PRO, int VEC, bool TAIL = false, bool XBF = false, bool OBF = false>
int launch_smallk_vec(const fdn_conv1x1_desc& d, hipStream_t s) {
    const int ntiles = (d.N + 31) / 32;
    const size_t lds = (2UL * NCH * KC + (size_t)NCH * KC * (ntiles * 32 + 1) + ntiles * 32) * sizeof(float);
    FDN_NUM_CU_OR_FAIL()
    Geo g;
    g.resident = 1;
    g.tiles_per_img = cdiv(d.P, 4 * 32 * VEC);
    g.total_tiles = d.B * g.tiles_per_img;
    auto kern = conv1x1_smallk_vec_kernel<NCH, PRO, VEC, TAIL, XBF, OBF>;
    if (lds > 48 * 1024 && !fdn_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return FDN_ERR_LAUNCH;
    int per_cu = 0;
    if (!fdn_occupancy(&per_cu, reinterpret_cast<const void*>(kern), 256, lds) || per_cu < 1)
        per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    int grid = g_num_cu * per_cu;
    if (grid > g.total_tiles) grid = g.total_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, d, g);
    return fdn_launch_status();
}

// the vectorised kernel covers: small-K shapes (see smallk_ok) with one segment, no epilogue operand, plain / LN prologue,
// P a multiple of 4 and 16-byte aligned tensors
bool smallk_vec_ok(const fdn_conv1x1_desc& d) {
    if (d.kseg[1] > 0 || d.kseg[2] > 0 || d.epi != FDN_EPI_NONE || (d.pro != FDN_PRO_NONE && d.pro != FDN_PRO_LN)) return false;
    if (d.P % 4 != 0 || d.xbs[0] % 4 != 0 || d.obs % 4 != 0) return false;
    uintptr_t a = reinterpret_cast<uintptr_t>(d.x[0]) | reinterpret_cast<uintptr_t>(d.out);
    if (d.pro != FDN_PRO_NONE) a |= reinterpret_cast<uintptr_t>(d.stats);
    return (a & 15) == 0;
}

template <int NCH, int PRO>
int launch_smallk_stream_vec(const fdn_conv1x1_desc& d, hipStream_t s) {
    const size_t lds = (2UL * NCH * KC + 2UL * NCH * KC * 33 + ((d.N + 31) / 32) * 32) * sizeof(float);
    FDN_NUM_CU_OR_FAIL()
    Geo g;
    g.resident = 0;
    g.tiles_per_img = cdiv(d.P, 8 * 32 * 2);
    g.total_tiles = d.B * g.tiles_per_img;
    int grid = g_num_cu;
    if (grid > g.total_tiles) grid = g.total_tiles;
    hipLaunchKernelGGL((conv1x1_smallk_stream_vec_kernel<NCH, PRO>), dim3(grid), dim3(512), lds, s, d, g);
    return fdn_launch_status();
}

template <int MT, bool XBF = false>
int launch_kstream_vec(const fdn_conv1x1_desc& d, hipStream_t s) {
    const int nch = (d.K + KC - 1) / KC;
    const size_t chunk = (size_t)KC * (MT * 32 + 1) * sizeof(float);
    Geo g;
    g.resident = (nch * chunk <= 64 * 1024) ? 1 : 0;
    size_t lds = (g.resident ? nch : 2) * chunk;
    g.bias_off = (int)(lds / sizeof(float));
    lds += (size_t)MT * 32 * sizeof(float);
    g.tiles_per_img = cdiv(d.P, 4 * 32 * 2);
    g.total_tiles = d.B * g.tiles_per_img;
    auto kern = conv1x1_kstream_vec_kernel<MT, XBF>;
    if (lds > 48 * 1024 && !fdn_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return FDN_ERR_LAUNCH;
    FDN_NUM_CU_OR_FAIL()
    int per_cu = 0;
    if (!fdn_occupancy(&per_cu, reinterpret_cast<const void*>(kern), 256, lds) || per_cu < 1)
        per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    int grid = g_num_cu * per_cu;
    if (grid > g.total_tiles) grid = g.total_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, d, g);
    return fdn_launch_status();
}

// plain deep-K convs for the 8-byte-lane K-streaming kernel
bool kstream_vec_ok(const fdn_conv1x1_desc& d) {
    if (d.pro != FDN_PRO_NONE || d.kseg[1] > 0 || d.kseg[2] > 0 || d.K <= 96 || d.N > 96) return false;   // (N = 128 spills: slower)
    if (d.epi != FDN_EPI_NONE && d.epi != FDN_EPI_RES) return false;
    if (d.P % 4 != 0 || d.xbs[0] % 4 != 0 || d.obs % 4 != 0 || (d.epi == FDN_EPI_RES && d.rbs % 4 != 0)) return false;
    uintptr_t a = reinterpret_cast<uintptr_t>(d.x[0]) | reinterpret_cast<uintptr_t>(d.out);
    if (d.epi == FDN_EPI_RES) a |= reinterpret_cast<uintptr_t>(d.res);
    if (d.stats_out) a |= reinterpret_cast<uintptr_t>(d.stats_out);
    return (a & 15) == 0;                          // measured: 172 -> 64 at level 2 10.2 -> 7.0 ms (71 TFLOP/s)
}

// K <= 128, N >= 2K, single input segment, plain/LN prologue, no muladd epilogue, weights too big for LDS
bool smallk_stream_ok(const fdn_conv1x1_desc& d) {
    if (d.K > 128 || d.K <= 64 || d.stats_out || d.kseg[1] > 0) return false;
    if (d.pro != FDN_PRO_NONE && d.pro != FDN_PRO_LN) return false;
    if (d.epi == FDN_EPI_MULADD) return false;
    return d.N >= 2 * d.K;
}

// narrow project_out convs for the vectorised kernel's TAIL form: K <= 96, N <= 32, no prologue, residual or no
// epilogue operand, statistics allowed
bool narrow_vec_ok(const fdn_conv1x1_desc& d) {
    if (d.K > 96 || d.N > 32 || d.pro != FDN_PRO_NONE || d.kseg[1] > 0 || d.kseg[2] > 0) return false;
    if (d.epi != FDN_EPI_NONE && d.epi != FDN_EPI_RES) return false;
    if (d.P % 4 != 0 || d.xbs[0] % 4 != 0 || d.obs % 4 != 0 || (d.epi == FDN_EPI_RES && d.rbs % 4 != 0)) return false;
    uintptr_t a = reinterpret_cast<uintptr_t>(d.x[0]) | reinterpret_cast<uintptr_t>(d.out);
    if (d.epi == FDN_EPI_RES) a |= reinterpret_cast<uintptr_t>(d.res);
    if (d.stats_out) a |= reinterpret_cast<uintptr_t>(d.stats_out);
    return (a & 15) == 0;
}

// true when the small-K kernel covers this problem
bool smallk_ok(const fdn_conv1x1_desc& d) {
    if (d.K > 64 || d.stats_out || d.pro == FDN_PRO_LN3_GATE) return false;
    if (d.N < 2 * d.K || d.N < 64) return false;                  // made for N >> K
    const int nch = (d.K + KC - 1) / KC, ntiles = (d.N + 31) / 32;
    const size_t lds = (2UL * nch * KC + (size_t)nch * KC * (ntiles * 32 + 1)) * sizeof(float);
    return lds <= 100 * 1024;
}

template <int PRO>
int launch_smallk_nch(const fdn_conv1x1_desc& d, hipStream_t s) {
    const int nch = (d.K + KC - 1) / KC;
    if (nch == 1) return launch_smallk<1, PRO>(d, s);
    return launch_smallk<2, PRO>(d, s);
}

template <int MT, int PRO>
int launch_early(const fdn_conv1x1_desc& d, hipStream_t s) {
    // narrow, shallow problems with an epilogue operand are load-latency bound: 4-wave workgroups (finer
    // register granularity per CU) that fetch the epilogue operands ahead of the MFMAs
    if constexpr (MT <= 2) {
        if (d.epi != FDN_EPI_NONE && d.K <= 64) return launch<MT, PRO, 4, true>(d, s);
    }
    // measured (tools/gpu_gemm_shapes.py): 64-wide plain GEMMs (FDFFN project_out at level 2, 172 -> 64) gain from
    // 4-wave workgroups at 3 waves per SIMD (13.5 -> 10.9 ms); wider tiles spill at that register budget
    if constexpr (MT == 2 && PRO == FDN_PRO_NONE) return launch<MT, PRO, 4, false>(d, s);
    // wide tiles: two independent 4-wave workgroups per CU instead of one of 8 - their per-chunk barriers drift apart,
    // so one workgroup's MFMAs fill the other's load-issue / barrier phase (345 -> 128: 12.1 -> 11.3 ms)
    if constexpr (MT >= 3 && PRO != FDN_PRO_LN_MULADD) return launch<MT, PRO, 4, false>(d, s);     // (LN_MULADD spills at that budget)
    return launch<MT, PRO, 8, false>(d, s);
}

template <int MT>
int launch_pro(const fdn_conv1x1_desc& d, hipStream_t s) {
    switch (d.pro) {
        case FDN_PRO_NONE: return launch_early<MT, FDN_PRO_NONE>(d, s);
        case FDN_PRO_LN: return launch_early<MT, FDN_PRO_LN>(d, s);
        case FDN_PRO_LN3_GATE:
            if constexpr (MT >= 2) return launch<MT, FDN_PRO_LN3_GATE, 4, false>(d, s);
            return launch<MT, FDN_PRO_LN3_GATE, 8, false>(d, s);
        case FDN_PRO_LN_MULADD: return launch_early<MT, FDN_PRO_LN_MULADD>(d, s);
        default: return FDN_ERR_ARG;
    }
}

}  // namespace

// gemm_tile.hip: LDS-tiled kernel for the deep N = 128 shapes; FDN_ERR_UNSUPPORTED = not one of them
int fdn_gemm_tile(const fdn_conv1x1_desc& d, hipStream_t s);
// gemm_split.hip: the same shapes and the N > 128 ones on the bf16 matrix pipe (split operands), when the caller supplies packed weights
int fdn_gemm_split(const fdn_conv1x1_desc& d, hipStream_t s);

extern "C" int fdn_conv1x1(const fdn_conv1x1_desc* dp, fdn_stream_t stream) {
    FDN_CHECK_ARG(dp != nullptr);
    fdn_conv1x1_desc d = *dp;
    FDN_CHECK_ARG(d.B > 0 && d.K > 0 && d.N > 0 && d.P > 0);
    FDN_CHECK_ARG(d.x[0] && d.w && d.out);
    FDN_CHECK_ARG(d.kseg[0] + d.kseg[1] + d.kseg[2] == d.K);
    FDN_CHECK_ARG(d.kseg[1] == 0 || d.x[1]);
    FDN_CHECK_ARG(d.kseg[2] == 0 || d.x[2]);
    // stats == NULL with a LayerNorm prologue: the K-streaming split-bf16 kernel takes the statistics itself (LN3_GATE / LN_MULADD with packed
    // weights on a deep shape); every other kernel wants them from fdn_chan_stats or a producer's epilogue
    const bool own_stats = d.pro != FDN_PRO_NONE && !d.stats;
    if (own_stats) FDN_CHECK_ARG(d.pro == FDN_PRO_LN3_GATE || d.pro == FDN_PRO_LN_MULADD);
    if (d.pro >= FDN_PRO_LN3_GATE) FDN_CHECK_ARG(d.gamma && d.beta);
    if (d.pro == FDN_PRO_LN3_GATE) FDN_CHECK_ARG(d.xb && d.ln_group * 3 == d.K && d.kseg[0] == d.K);
    if (d.pro == FDN_PRO_LN_MULADD) FDN_CHECK_ARG(d.xb);
    if (d.epi == FDN_EPI_RES) FDN_CHECK_ARG(d.res);
    if (d.epi == FDN_EPI_MULADD) FDN_CHECK_ARG(d.mul && d.add);
    if (d.stats_out) FDN_CHECK_ARG(d.N <= 160);
#ifndef FDN_GEMM_TRACE
    d.vec4 = 0;
#endif
    // 32-bit buffer offsets: every per-image plane set must stay below 4 GiB (incl. the padded K / N tails)
    {
        const unsigned long long lim = 0xFFFFFFFFull, P4 = 4ull * d.P;
        if ((unsigned long long)(d.K + 40) * P4 > lim || (unsigned long long)(d.N + 200) * P4 > lim) return FDN_ERR_UNSUPPORTED;
        if (d.kseg[1] > 0 && ((d.kseg[0] & 1) || (d.kseg[1] & 1))) return FDN_ERR_UNSUPPORTED;   // k-step pairs must not straddle segments
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // bf16 STORAGE of one operand (the block-internal FDFFN tensors of levels 1-2): the pixel-pair kernels only -
    //   x_bf16  : the project_out convs (narrow TAIL form / K-streaming form), fp32 result;
    //   out_bf16: the project_in convs (small-K form, plain or LayerNorm prologue), fp32 input.
    if (d.x_bf16 || d.out_bf16) {
        if ((d.x_bf16 && d.out_bf16) || own_stats) return FDN_ERR_UNSUPPORTED;
        if (d.x_bf16) {
            if (kstream_vec_ok(d)) {
                const int tiles = (d.N + 31) / 32;
                if (tiles == 1) return launch_kstream_vec<1, true>(d, s);
                if (tiles == 2) return launch_kstream_vec<2, true>(d, s);
                return launch_kstream_vec<3, true>(d, s);
            }
            if (narrow_vec_ok(d)) {
                if (d.K <= KC) return launch_smallk_vec<1, FDN_PRO_NONE, 2, true, true, false>(d, s);
                if (d.K <= 2 * KC) return launch_smallk_vec<2, FDN_PRO_NONE, 2, true, true, false>(d, s);
                return launch_smallk_vec<3, FDN_PRO_NONE, 2, true, true, false>(d, s);
            }
            return FDN_ERR_UNSUPPORTED;
        }
        if (smallk_ok(d) && smallk_vec_ok(d)) {
            const int ntiles = (d.N + 31) / 32;
            if (d.K <= KC) {
                if (d.pro == FDN_PRO_LN) return launch_smallk_vec<1, FDN_PRO_LN, 2, false, false, true>(d, s);
                return launch_smallk_vec<1, FDN_PRO_NONE, 2, false, false, true>(d, s);
            }
            if ((2UL * 2 * KC + 2UL * KC * (ntiles * 32 + 1)) * sizeof(float) <= 52 * 1024) {
                if (d.pro == FDN_PRO_LN) return launch_smallk_vec<2, FDN_PRO_LN, 2, false, false, true>(d, s);
                return launch_smallk_vec<2, FDN_PRO_NONE, 2, false, false, true>(d, s);
            }
        }
        return FDN_ERR_UNSUPPORTED;
    }
    {
        const int rc = fdn_gemm_split(d, s);            // level 3 with packed weights: fp32 on the bf16 matrix pipe (gemm_split.hip)
        if (rc != FDN_ERR_UNSUPPORTED) return rc;
    }
    if (own_stats) return FDN_ERR_UNSUPPORTED;
    {
        const int rc = fdn_gemm_tile(d, s);             // 459 -> 128 (LN3 * v_value), 345 -> 128, 128 -> 128 at level 3
        if (rc != FDN_ERR_UNSUPPORTED) return rc;
    }
    // K = 64 with a weight matrix too big to sit in LDS three times per CU (level-2 to_hidden, 64 -> 304): stream the weights too
    if (d.K > KC && d.K <= 2 * KC && d.N >= 256 && !d.stats_out && d.kseg[1] == 0 && d.epi == FDN_EPI_NONE &&
        (d.pro == FDN_PRO_NONE || d.pro == FDN_PRO_LN) && smallk_vec_ok(d)) {               // 15.1 -> 12.4 ms
        if (d.pro == FDN_PRO_LN) return launch_smallk_stream_vec<2, FDN_PRO_LN>(d, s);
        return launch_smallk_stream_vec<2, FDN_PRO_NONE>(d, s);
    }
    if (smallk_stream_ok(d) && d.epi == FDN_EPI_NONE && smallk_vec_ok(d)) {           // 128->612: 21.3 -> 20.0 ms, 128->345: 13.1 -> 10.9 ms
        const int nch = (d.K + KC - 1) / KC;
        if (d.pro == FDN_PRO_LN) return nch == 3 ? launch_smallk_stream_vec<3, FDN_PRO_LN>(d, s) : launch_smallk_stream_vec<4, FDN_PRO_LN>(d, s);
        return nch == 3 ? launch_smallk_stream_vec<3, FDN_PRO_NONE>(d, s) : launch_smallk_stream_vec<4, FDN_PRO_NONE>(d, s);
    }
    if (smallk_stream_ok(d)) {
        const int nch = (d.K + KC - 1) / KC;
        if (d.pro == FDN_PRO_LN) return nch == 3 ? launch_smallk_stream<3, FDN_PRO_LN>(d, s) : launch_smallk_stream<4, FDN_PRO_LN>(d, s);
        return nch == 3 ? launch_smallk_stream<3, FDN_PRO_NONE>(d, s) : launch_smallk_stream<4, FDN_PRO_NONE>(d, s);
    }
    if (kstream_vec_ok(d)) {
        const int tiles = (d.N + 31) / 32;
        if (tiles == 1) return launch_kstream_vec<1>(d, s);
        if (tiles == 2) return launch_kstream_vec<2>(d, s);
        return launch_kstream_vec<3>(d, s);
    }
    if (narrow_vec_ok(d)) {
        if (d.K <= KC) return launch_smallk_vec<1, FDN_PRO_NONE, 2, true>(d, s);
        if (d.K <= 2 * KC) return launch_smallk_vec<2, FDN_PRO_NONE, 2, true>(d, s);
        return launch_smallk_vec<3, FDN_PRO_NONE, 2, true>(d, s);
    }
    if (smallk_ok(d) && smallk_vec_ok(d)) {
        // measured (tools/bench_kernels.py to_hidden ffn_in, B=8 720p): 8-byte lanes win for K <= 32 (32->152: 1.65 -> 1.44 ms,
        // 32->86: 0.92 -> 0.76 ms) and for K <= 64 while the weight matrix leaves room for 3 workgroups per CU (64->172:
        // 0.74 -> 0.59 ms; 64->304 is slower vectorised); 16-byte lanes spill with the LN prologue
        const int ntiles = (d.N + 31) / 32;
        if (d.K <= KC) {
            if (d.pro == FDN_PRO_LN) return launch_smallk_vec<1, FDN_PRO_LN, 2>(d, s);       // (16-byte lanes measure the same here)
            return launch_smallk_vec<1, FDN_PRO_NONE, 2>(d, s);       // (8-byte lanes only: the kernel static_asserts VEC == 2)
        }
        if ((2UL * 2 * KC + 2UL * KC * (ntiles * 32 + 1)) * sizeof(float) <= 52 * 1024) {
            if (d.pro == FDN_PRO_LN) return launch_smallk_vec<2, FDN_PRO_LN, 2>(d, s);
            return launch_smallk_vec<2, FDN_PRO_NONE, 2>(d, s);
        }
    }
    if (smallk_ok(d)) {
        switch (d.pro) {
            case FDN_PRO_NONE: return launch_smallk_nch<FDN_PRO_NONE>(d, s);
            case FDN_PRO_LN: return launch_smallk_nch<FDN_PRO_LN>(d, s);
            default: return launch_smallk_nch<FDN_PRO_LN_MULADD>(d, s);
        }
    }
    switch (pick_mt(d.N)) {
        case 1: return launch_pro<1>(d, s);
        case 2: return launch_pro<2>(d, s);
        case 3: return launch_pro<3>(d, s);
        case 4: return launch_pro<4>(d, s);
        default: return launch_pro<5>(d, s);
    }
}
