// 1x1 convolution (the 79 % of FDN's MACs) as an exact-fp32 MFMA GEMM on NCHW planes.
//
//   out[b][n][p] = epi( act( sum_k W[n][k] * pro(x[b][k][p]) + bias[n] ) )
//
// Replaces the F.conv2d(kernel_size=1) calls of FDN_arch.py:576 (to_hidden), :639 (attn
// project_out), :456/:474 (FDFFN project_in/out), :421/:428 (FCAFFN), :685-686 (Fuse) and the
// 62 1x1 convs of MAR (:78-86, :125-134, :168-190), with the surrounding channel-LayerNorm
// (:313-342), the v_value gating (:633-638), `norm(x)*x1+x1` (:420), the residual adds (:671-675)
// and `x*mul+add` (:423) folded into the operand staging / epilogue.
//
// Mapping (CDNA4): pixels are the contiguous axis of NCHW, so the GEMM is
//   D[n][p] = A[n][k] * B[k][p],  A = weights, B = activations,
// on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-exact fmaf chain).  One 256-thread
// workgroup owns 128 consecutive pixels of one image; wave w owns pixels [32w, 32w+32) and MT
// 32-row tiles of output channels, so the accumulator of reg r / lane l is
//   n = 32*mt + (r&3) + 8*(r>>2) + 4*(l>>5),  p = 32*w + (l&31)
// and every store instruction writes two full 128-byte lines.  K is streamed through LDS in
// chunks of 32 channels (global -> registers while the previous chunk's MFMAs run, registers ->
// LDS after the barrier).  When N needs more than one pass of MT tiles the workgroup loops over
// the passes itself so the activation tile is re-read by the same CU (L1/L2 hit), never by a
// workgroup on another XCD.
#include "common.hpp"

namespace {

constexpr int BP = 128;   // pixels per workgroup
constexpr int KC = 32;    // K chunk staged in LDS
constexpr int WPAD = 1;   // Ws row padding (transposing store: bank = k + n)

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Pro {
    // per-thread constants of the prologue for its 4 pixels
    float mu[3][4], rs[3][4];
};

template <int MT>
__global__ __launch_bounds__(256) void conv1x1_kernel(fdn_conv1x1_desc d) {
    __shared__ float Xs[KC][BP];
    __shared__ float Ws[KC][MT * 32 + WPAD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = blockIdx.y;
    const long p0 = (long)blockIdx.x * BP;
    const int P = d.P, K = d.K, N = d.N;

    // ---- staging geometry for X: thread owns pixel quad pq (4 px) and rows kr + 8*i -----------
    const int pq = (tid & 31) * 4;
    const int kr = tid >> 5;                  // 0..7
    const long pg = p0 + pq;                  // first global pixel of the quad
    const bool vec = d.vec4 && (pg + 3 < P);
    const int nvalid = pg >= P ? 0 : (pg + 4 <= P ? 4 : (int)(P - pg));

    const int pro = d.pro;
    const int E = d.ln_group;                 // channels per LN group (LN3_GATE: E; else K)
    Pro st;
    if (pro != FDN_PRO_NONE) {
        const int G = (pro == FDN_PRO_LN3_GATE) ? 3 : 1;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = (g < G) && (j < nvalid);
                const float* sp = d.stats + ((long)b * G + (g < G ? g : 0)) * 2 * P;
                st.mu[g][j] = ok ? sp[pg + j] : 0.f;
                st.rs[g][j] = ok ? sp[P + pg + j] : 0.f;
            }
    }

    auto load_x = [&](int k, float (&v)[4]) {
        // raw fetch of channel k (concat of up to 3 segments) for this thread's pixel quad
        const float* src;
        if (k < d.kseg[0]) src = d.x[0] + (long)b * d.xbs[0] + (long)k * P;
        else if (k < d.kseg[0] + d.kseg[1]) src = d.x[1] + (long)b * d.xbs[1] + (long)(k - d.kseg[0]) * P;
        else src = d.x[2] + (long)b * d.xbs[2] + (long)(k - d.kseg[0] - d.kseg[1]) * P;
        if (vec) {
            const float4 t = *reinterpret_cast<const float4*>(src + pg);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = j < nvalid ? src[pg + j] : 0.f;
        }
    };
    auto load_aux = [&](int k, float (&v)[4]) {  // second operand (v_value / x1), channel k of d.xb
        const float* src = d.xb + (long)b * d.xbbs + (long)k * P;
        if (vec) {
            const float4 t = *reinterpret_cast<const float4*>(src + pg);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = j < nvalid ? src[pg + j] : 0.f;
        }
    };

    const int nchunks = (K + KC - 1) / KC;
    const int npass = (N + MT * 32 - 1) / (MT * 32);

    for (int pass = 0; pass < npass; ++pass) {
        const int nbase = pass * MT * 32;
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

        float xr[4][4];          // staged X: rows kr + 8*i
        float wr[MT * 4];        // staged W: element e = tid + 256*i of the [MT*32][32] chunk

        auto fetch = [&](int kc) {
            const int k0 = kc * KC;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + kr + 8 * i;
                if (k < K) {
                    load_x(k, xr[i]);
                    if (pro == FDN_PRO_LN) {
                        const float ga = d.gamma[k], be = d.beta[k];
#pragma unroll
                        for (int j = 0; j < 4; ++j) xr[i][j] = (xr[i][j] - st.mu[0][j]) * st.rs[0][j] * ga + be;
                    } else if (pro == FDN_PRO_LN3_GATE) {
                        const int g = k / E, e = k - g * E;
                        const float ga = d.gamma[k], be = d.beta[k];
                        float vv[4];
                        load_aux(e, vv);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float mu = g == 0 ? st.mu[0][j] : (g == 1 ? st.mu[1][j] : st.mu[2][j]);
                            const float rs = g == 0 ? st.rs[0][j] : (g == 1 ? st.rs[1][j] : st.rs[2][j]);
                            xr[i][j] = ((xr[i][j] - mu) * rs * ga + be) * vv[j];
                        }
                    } else if (pro == FDN_PRO_LN_MULADD) {
                        const float ga = d.gamma[k], be = d.beta[k];
                        float x1[4];
                        load_aux(k, x1);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            xr[i][j] = ((xr[i][j] - st.mu[0][j]) * st.rs[0][j] * ga + be) * x1[j] + x1[j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j >= nvalid) xr[i][j] = 0.f;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) xr[i][j] = 0.f;
                }
            }
            // weights: W[n][k], lanes run along k (coalesced 128-byte rows)
            const int kk = tid & 31;
#pragma unroll
            for (int i = 0; i < MT * 4; ++i) {
                const int nl = (tid >> 5) + 8 * i;
                const int n = nbase + nl, k = k0 + kk;
                wr[i] = (n < N && k < K) ? d.w[(long)n * K + k] : 0.f;
            }
        };
        auto stash = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4*>(&Xs[kr + 8 * i][pq]) = make_float4(xr[i][0], xr[i][1], xr[i][2], xr[i][3]);
            const int kk = tid & 31;
#pragma unroll
            for (int i = 0; i < MT * 4; ++i) Ws[kk][(tid >> 5) + 8 * i] = wr[i];
        };

        fetch(0);
        for (int kc = 0; kc < nchunks; ++kc) {
            __syncthreads();            // previous chunk's reads are done
            stash();
            __syncthreads();
            if (kc + 1 < nchunks) fetch(kc + 1);
            const int kh = lane >> 5, ln = lane & 31;
#pragma unroll
            for (int kk = 0; kk < KC; kk += 2) {
                const float bv = Xs[kk + kh][wave * 32 + ln];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float av = Ws[kk + kh][m * 32 + ln];
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
                }
            }
        }

        // ---- epilogue ----------------------------------------------------------------------
        const long p = p0 + wave * 32 + (lane & 31);
        if (p < P) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = nbase + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (n < N) {
                        float v = acc[m][r];
                        if (d.bias) v += d.bias[n];
                        v = apply_act(v, d.act);
                        const long o = (long)n * P + p;
                        if (d.epi == FDN_EPI_RES) v += d.res[(long)b * d.rbs + o];
                        else if (d.epi == FDN_EPI_MULADD) v = v * d.mul[(long)b * d.mbs + o] + d.add[(long)b * d.mbs + o];
                        d.out[(long)b * d.obs + o] = v;
                    }
                }
        }
    }
}

int pick_mt(int N) {
    // fewest wasted 32-row tiles, then fewest passes; MT <= 5 keeps the accumulator at 80 VGPRs
    const int tiles = (N + 31) / 32;
    int best = 1, best_cost = 1 << 30;
    for (int mt = 1; mt <= 5; ++mt) {
        const int passes = (tiles + mt - 1) / mt;
        const int cost = passes * mt * 100 + passes;   // computed tiles dominate, then passes
        if (cost < best_cost || (cost == best_cost && mt > best)) { best_cost = cost; best = mt; }
    }
    return best;
}

}  // namespace

extern "C" int fdn_conv1x1(const fdn_conv1x1_desc* dp, fdn_stream_t stream) {
    FDN_CHECK_ARG(dp != nullptr);
    fdn_conv1x1_desc d = *dp;
    FDN_CHECK_ARG(d.B > 0 && d.K > 0 && d.N > 0 && d.P > 0);
    FDN_CHECK_ARG(d.x[0] && d.w && d.out);
    FDN_CHECK_ARG(d.kseg[0] + d.kseg[1] + d.kseg[2] == d.K);
    FDN_CHECK_ARG(d.kseg[1] == 0 || d.x[1]);
    FDN_CHECK_ARG(d.kseg[2] == 0 || d.x[2]);
    if (d.pro != FDN_PRO_NONE) FDN_CHECK_ARG(d.stats && d.gamma && d.beta);
    if (d.pro == FDN_PRO_LN3_GATE) FDN_CHECK_ARG(d.xb && d.ln_group * 3 == d.K && d.kseg[0] == d.K);
    if (d.pro == FDN_PRO_LN_MULADD) FDN_CHECK_ARG(d.xb);
    if (d.epi == FDN_EPI_RES) FDN_CHECK_ARG(d.res);
    if (d.epi == FDN_EPI_MULADD) FDN_CHECK_ARG(d.mul && d.add);
    // float4 path: every plane base 16-byte aligned and P a multiple of 4
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    bool vec = (d.P % 4 == 0) && al(d.x[0]) && (d.xbs[0] % 4 == 0);
    if (d.kseg[1]) vec = vec && al(d.x[1]) && (d.xbs[1] % 4 == 0);
    if (d.kseg[2]) vec = vec && al(d.x[2]) && (d.xbs[2] % 4 == 0);
    if (d.xb) vec = vec && al(d.xb) && (d.xbbs % 4 == 0);
    d.vec4 = vec ? 1 : 0;

    dim3 grid(cdiv(d.P, BP), d.B), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (pick_mt(d.N)) {
        case 1: hipLaunchKernelGGL(conv1x1_kernel<1>, grid, block, 0, s, d); break;
        case 2: hipLaunchKernelGGL(conv1x1_kernel<2>, grid, block, 0, s, d); break;
        case 3: hipLaunchKernelGGL(conv1x1_kernel<3>, grid, block, 0, s, d); break;
        case 4: hipLaunchKernelGGL(conv1x1_kernel<4>, grid, block, 0, s, d); break;
        default: hipLaunchKernelGGL(conv1x1_kernel<5>, grid, block, 0, s, d); break;
    }
    return fdn_launch_status();
}
