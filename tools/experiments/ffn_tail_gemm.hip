// EXPERIMENT (round 3, not part of the library): built, correct (10 shapes against float64 through fdn_ffn_tail's test), not faster:
//   level 3 (345 -> 128, 184 x 320, B = 8): 0.74 ms against 0.58 ms for fdn_dwconv_gate + fdn_conv1x1
//   level 2 (172 -> 64, 368 x 640):         1.36 ms against 1.07 ms
// with the gate as one pixel per thread (31 LDS reads per gated value), with the depthwise taps through LDS instead of scalar loads
// (0.93 -> 0.74 ms at level 3) and with the gate as (channel pair, four pixels) per thread on 16-byte window reads (no change):
// the step time (~11 k cycles at level 3) is about twice what its vector instructions, LDS traffic and MFMAs add up to, the three
// barriers per 16-channel step with two or three workgroups per CU being the suspect.  To try it again: copy the file to
// fdn-tip2025_amd/csrc/, declare fdn_ffn_tail_packed in include/fdn_hip.h and call it with wpk = fdn_conv1x1_pack(w, N, C, 0).
// FFN tail of the DEEP layers in one launch (FDN_arch.py:472-474, :673): Conv2d(C, 2C, 3, groups = C) -> gelu(x1) * x2 -> project_out
// (1x1, C -> N) + residual (+ the next LayerNorm's statistics), for the shapes where the sliding-window kernel of ffn_tail.hip
// loses to gate + GEMM (level 2: 172 -> 64, level 3: 345 -> 128): there the projection is 2-4 x the work of the level-1 tail and,
// as v_mfma_f32_32x32x2_f32, competes with the stencils for the vector ALU's datapath.  Here the projection runs on the bf16 matrix
// pipe (operands cut exactly into three bf16 parts, six products: fp32 arithmetic) while the vector ALU does
// nothing but the gate: the gated tensor (C planes written and read back: 0.65 GB per call at level 3) never exists.
//
// A workgroup owns 128 consecutive pixels x all N <= 128 output channels and walks the hidden channels 16 at a time (one MFMA k-step):
//   stage   the 8 + 9 input planes of the step (output j reads plane j / 2, output C + j plane (C + j) / 2) as three 130-pixel
//           segments (rows y - 1, y, y + 1 in flattened pixel order: neighbour (dy, dx) of pixel p is element p + dy W + dx;
//           pixels outside the tensor read 0, the two column-wrap cases are masked per pixel), requested one step ahead;
//   gate    thread (pixel, half) evaluates its 8 channels: two 3 x 3 stencils from LDS, GELU, product - splits the values and writes
//           them as MFMA B-operand cells (3 parts x 16 bytes);
//   matrix  the four waves (2 pixel halves x 2 channel halves) run 6 MFMAs per accumulator tile on the cells and on the weight
//           cells of the step (the layout of fdn_conv1x1_pack: no second packing).
// The vector ALU is the bound (about 58 instruction slots per gated value); three workgroups per CU overlap one's matrix step and
// barriers with the others' stencils.
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
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

constexpr int TP = 128;                      // pixels per workgroup
constexpr int KS = 16;                       // hidden channels per step
constexpr int NPL = 17;                      // staged planes per step: 8 for x1, up to 9 for x2
constexpr int SEG = 132;                     // floats per staged row segment (130 used)
constexpr int PLF = 3 * SEG;                 // floats per staged plane
constexpr int CHUNK = 3 * 2 * 2 * 128;       // 16-byte cells of one 32-deep chunk of fdn_conv1x1_pack
constexpr unsigned OOB = 0x80000000u;

struct TgArgs {
    const float* y;            // [B][C][P]
    const float* wdw;          // [2C][9]
    const fdn_u32x4* wpk;      // project_out [N][C] as packed by fdn_conv1x1_pack(w, N, C, 0)
    const float* res;          // [B][N][P] or null
    float* out;                // [B][N][P]
    float* stats_out;          // [B][2][P] or null
    int B, C, N, H, W, tiles_per_img;
};

// NT2 = 32-channel tiles per wave (N <= 64: 1, N <= 128: 2)
template <int NT2>
__global__ __launch_bounds__(256, NT2 == 1 ? 3 : 2) void ffn_tail_gemm_kernel(TgArgs a) {
    __shared__ __attribute__((aligned(16))) float planes[NPL * PLF];
    __shared__ fdn_u32x4 Xs[3 * 2 * 128];          // [part][k half][pixel]
    __shared__ fdn_u32x4 Ws[3 * 2 * 128];          // [part][k half][channel row]
    __shared__ float red[2][TP];
    __shared__ float dwl[KS * 18 + 32];            // taps of the step's channels: [channel][wA(9) | wB(9)] (scalar loads per channel stalled the gate)
    const int C = a.C, N = a.N, W = a.W;
    const unsigned P = (unsigned)a.H * W, P4 = P * 4u;
    const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5, ln = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const unsigned S = xcd_contiguous(blockIdx.x, gridDim.x);
    const int b = (int)(S / (unsigned)a.tiles_per_img);
    const unsigned p0 = (S - (unsigned)b * a.tiles_per_img) * TP;
    const rsrc_t ry = mk_rsrc(a.y + (long)b * C * P, (unsigned)C * P4);
    const int nks = (C + KS - 1) / KS;

    // ---- staging plan of this thread: positions r = tid and tid + 256 of the 3 x 130 segment block, the same for all 17 planes ----
    unsigned sg[2];                                  // byte offset of the pixel inside a plane, OOB outside the tensor
    int sl[2];                                       // LDS float offset inside a staged plane (-1: no element)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = tid + 256 * h, seg = r / 130, pos = r - seg * 130;
        const long q = (long)p0 + (long)(seg - 1) * W + pos - 1;
        const bool in = r < 390;
        sg[h] = (in && q >= 0 && q < (long)P) ? (unsigned)q * 4u : OOB;
        sl[h] = in ? seg * SEG + pos : -1;
    }
    float stg[NPL][2], dst[2] = {0.f, 0.f};
    fdn_u32x4 wst[3];
    auto fetch = [&](int k) __attribute__((always_inline)) {
        const int j0 = k * KS, pa = j0 >> 1, pb = (C + j0) >> 1;               // first plane of the x1 / x2 inputs of this step
#pragma unroll
        for (int slot = 0; slot < NPL; ++slot) {
            const int pl = slot < 8 ? pa + slot : pb + slot - 8;               // (planes past the tensor fall outside the descriptor: 0)
#pragma unroll
            for (int h = 0; h < 2; ++h) stg[slot][h] = bload(ry, sg[h], (unsigned)pl * P4);
        }
        const fdn_u32x4* src = a.wpk + (long)(k >> 1) * CHUNK + (k & 1) * 256;  // cells ((part 2 + ks) 2 + half) 128 + row of the chunk
#pragma unroll
        for (int i = 0; i < 3; ++i) wst[i] = src[i * 512 + tid];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int idx = tid + 256 * h, ch = idx / 18, t = idx - ch * 18, j = j0 + ch;       // tap t of channel j (t < 9: x1 branch, else x2)
            dst[h] = (idx < KS * 18 && j < C) ? a.wdw[(long)((t < 9 ? 0 : C) + j) * 9 + (t < 9 ? t : t - 9)] : 0.f;
        }
    };
    auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int slot = 0; slot < NPL; ++slot)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (sl[h] >= 0) planes[slot * PLF + sl[h]] = stg[slot][h];
#pragma unroll
        for (int i = 0; i < 3; ++i) Ws[i * 256 + tid] = wst[i];
        dwl[tid] = dst[0];
        if (tid < KS * 18 - 256) dwl[256 + tid] = dst[1];
    };

    // ---- gate role: thread = (channel pair cp of the step, group g of four consecutive pixels): a 3 x 6 window per input plane feeds the
    // four pixels (two 16-byte / 8-byte LDS reads per row), the taps of the pair sit in registers for the step ----
    const int cp = tid >> 5, g4 = (tid & 31) * 4;
    float ml[4], mr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned pg = min(p0 + (unsigned)(g4 + i), P - 1);
        const int gx = (int)(pg % (unsigned)W);
        ml[i] = gx > 0 ? 1.f : 0.f;                                            // a row's first / last pixel has no left / right neighbour
        mr[i] = gx < W - 1 ? 1.f : 0.f;
    }
    unsigned* Xs32 = reinterpret_cast<unsigned*>(Xs);

    f32x16 acc[2][NT2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][t][r] = 0.f;

    fetch(0);
    stash();
    __syncthreads();
    for (int k = 0; k < nks; ++k) {
        const bool more = k + 1 < nks;
        if (more) fetch(k + 1);
        // ---- gate: channels j0 + 2 cp, + 1 at pixels g4 .. g4 + 3 ----
        {
            const int jA = k * KS + 2 * cp;
            const int pbase = (C + k * KS) >> 1;
            auto window = [&](int slot, float (&wv)[3][6]) __attribute__((always_inline)) {
                const float* pp = planes + slot * PLF + g4;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float4 a4 = *reinterpret_cast<const float4*>(pp + dy * SEG);
                    const float2 a2 = *reinterpret_cast<const float2*>(pp + dy * SEG + 4);
                    wv[dy][0] = a4.x; wv[dy][1] = a4.y; wv[dy][2] = a4.z; wv[dy][3] = a4.w; wv[dy][4] = a2.x; wv[dy][5] = a2.y;
                }
            };
            auto stencil = [&](const float (&wv)[3][6], const float* tp, float (&o)[4]) __attribute__((always_inline)) {
                float t[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) t[i] = tp[i];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float l = 0.f, c = 0.f, r = 0.f;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        l = fmaf(t[dy * 3], wv[dy][i], l);
                        c = fmaf(t[dy * 3 + 1], wv[dy][i + 1], c);
                        r = fmaf(t[dy * 3 + 2], wv[dy][i + 2], r);
                    }
                    o[i] = fmaf(ml[i], l, fmaf(mr[i], r, c));
                }
            };
            float wA[3][6], x1v[2][4], x2v[2][4];
            window(cp, wA);                                                    // both channels of the pair read input plane (j0 + 2 cp) / 2
            stencil(wA, dwl + (2 * cp) * 18, x1v[0]);
            stencil(wA, dwl + (2 * cp + 1) * 18, x1v[1]);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float wB[3][6];
                window(8 + ((C + jA + u) >> 1) - pbase, wB);
                stencil(wB, dwl + (2 * cp + u) * 18 + 9, x2v[u]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned u1, u2, u3;
                fdn_split3(gelu_fast(x1v[0][i]) * x2v[0][i], gelu_fast(x1v[1][i]) * x2v[1][i], u1, u2, u3);      // gelu(x1) * x2, FDN_arch.py:473
                const int cell = ((cp >> 2) * 128 + g4 + i) * 4 + (cp & 3);    // [k half][pixel] cell, pair cp & 3 of its eight channels
                Xs32[cell] = u1;
                Xs32[cell + 2 * 128 * 4] = u2;
                Xs32[cell + 4 * 128 * 4] = u3;
            }
        }
        __syncthreads();                                        // the cells of this step are complete
        // ---- matrix step ----
        {
            fdn_u32x4 A[NT2][3], Bv[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int s = 0; s < 2; ++s) Bv[s][p] = Xs[(p * 2 + kh) * 128 + wi * 64 + s * 32 + ln];
#pragma unroll
                for (int t = 0; t < NT2; ++t) A[t][p] = Ws[(p * 2 + kh) * 128 + (wj * NT2 + t) * 32 + ln];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < NT2; ++t) acc[s][t] = fdn_mfma_split6(A[t], Bv[s], acc[s][t]);
        }
        __syncthreads();                                        // planes, cells and weight cells of this step have been read
        if (more) {
            stash();
            __syncthreads();
        }
    }

    // ---- epilogue: residual, store, LayerNorm statistics of the result ----
    const rsrc_t ro = mk_rsrc(a.out + (long)b * N * P, (unsigned)N * P4);
    const rsrc_t rr = mk_rsrc(a.res ? a.res + (long)b * N * P : a.out, a.res ? (unsigned)N * P4 : 0u);
    float psum[2] = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const unsigned p = p0 + (unsigned)(wi * 64 + s * 32 + ln);
        const unsigned voff = p < P ? (4u * kh * P + p) * 4u : OOB;
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = (wj * NT2 + t) * 32 + (r & 3) + 8 * (r >> 2);
                rv[r] = bload(rr, (nrow + 4 * kh < N) ? voff : OOB, (unsigned)nrow * P4);         // 0 without a residual
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = (wj * NT2 + t) * 32 + (r & 3) + 8 * (r >> 2);
                const bool rok = nrow + 4 * kh < N;
                float v = acc[s][t][r] + rv[r];
                bstore(v, ro, rok ? voff : OOB, (unsigned)nrow * P4);
                v = rok ? v : 0.f;
                acc[s][t][r] = v;
                psum[s] += v;
            }
        }
    }
    if (a.stats_out) {
        // two-pass mean / variance: the two waves of a pixel half (wj = 0, 1) hold complementary channel halves
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            psum[s] += __shfl_xor(psum[s], 32);
            if (kh == 0) red[wj][wi * 64 + s * 32 + ln] = psum[s];
        }
        __syncthreads();
        float mean[2], qq[2] = {0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pl = wi * 64 + s * 32 + ln;
            mean[s] = (red[0][pl] + red[1][pl]) / (float)N;
#pragma unroll
            for (int t = 0; t < NT2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nrow = (wj * NT2 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const float dl = acc[s][t][r] - mean[s];
                    qq[s] += nrow < N ? dl * dl : 0.f;
                }
            qq[s] += __shfl_xor(qq[s], 32);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (kh == 0) red[wj][wi * 64 + s * 32 + ln] = qq[s];
        __syncthreads();
        if (wj == 0 && kh == 0) {
            float* sp_ = a.stats_out + (long)b * 2 * P;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pl = wi * 64 + s * 32 + ln;
                const unsigned p = p0 + (unsigned)pl;
                if (p < P) {
                    sp_[p] = mean[s];
                    sp_[P + p] = 1.0f / sqrtf((red[0][pl] + red[1][pl]) / (float)N + 1e-5f);
                }
            }
        }
    }
}

}  // namespace

extern "C" int fdn_ffn_tail_packed(const float* y, const float* dw_w, const void* wpk, const float* res, float* out, float* stats_out,
                                   int B, int C, int N, int H, int W, fdn_stream_t stream) {
    FDN_CHECK_ARG(y && dw_w && wpk && out && B > 0 && C > 0 && N > 0 && H > 0 && W > 0);
    const long P = (long)H * W;
    if (N > 128 || C < 32 || W < 2) return FDN_ERR_UNSUPPORTED;
    if ((unsigned long long)(C + 2) * 4ull * P >= 0x80000000ull || (unsigned long long)(N + 8) * 4ull * P >= 0x80000000ull) return FDN_ERR_UNSUPPORTED;
    TgArgs a;
    a.y = y; a.wdw = dw_w; a.wpk = static_cast<const fdn_u32x4*>(wpk); a.res = res; a.out = out; a.stats_out = stats_out;
    a.B = B; a.C = C; a.N = N; a.H = H; a.W = W;
    a.tiles_per_img = cdiv((int)P, TP);
    const long total = (long)B * a.tiles_per_img;
    if (total > 0x7FFFFFFFL) return FDN_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N <= 64) hipLaunchKernelGGL(ffn_tail_gemm_kernel<1>, dim3((unsigned)total), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(ffn_tail_gemm_kernel<2>, dim3((unsigned)total), dim3(256), 0, s, a);
    return fdn_launch_status();
}
