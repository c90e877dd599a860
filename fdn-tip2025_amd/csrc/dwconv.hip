// Depthwise 3x3 convolutions of the FDN path (HBM-bound stencil kernels, LDS halo tiles).
//   fdn_dwconv3x3    : plain depthwise 3x3 (+activation)            FDN_arch.py:396-399,435-441
//   fdn_dwconv_gate  : Conv2d(C,2C,3,groups=C) then gelu(x1)*x2     FDN_arch.py:472-473, 426-427
//   fdn_img_mod_maps : conv3_{mul,add}(conv1_{mul,add}(x_img))      FDN_arch.py:423
// Tile = 32 rows x 64 columns of one (b, c) plane per 256-thread workgroup; lane = column, so
// global loads/stores and LDS reads are contiguous across the wave.
#include "common.hpp"

namespace {

constexpr int TH = 32, TW = 64;
constexpr int LW = TW + 2;       // halo width
constexpr int LS = 68;           // LDS row stride (floats)

// load a (TH+2) x (TW+2) halo tile of plane `src` (H x W) into LDS, zero outside the image
__device__ __forceinline__ void load_halo(float (*t)[LS], const float* __restrict__ src, int H, int W, int y0, int x0) {
    for (int i = threadIdx.x; i < (TH + 2) * LW; i += 256) {
        const int r = i / LW, c = i - r * LW;
        const int y = y0 - 1 + r, x = x0 - 1 + c;
        t[r][c] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(long)y * W + x] : 0.f;
    }
}

__device__ __forceinline__ float stencil(const float (*t)[LS], int r, int c, const float* w) {
    // output at tile-local (r, c) ; halo origin is (-1,-1)
    float a = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) a = fmaf(w[dy * 3 + dx], t[r + dy][c + dx], a);
    return a;
}

__global__ __launch_bounds__(256) void dw3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                    float* __restrict__ out, int C, int H, int W, int act, int tiles_x) {
    __shared__ float t[TH + 2][LS];
    const int c = blockIdx.y, b = blockIdx.z;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const long plane = ((long)b * C + c) * H * W;
    load_halo(t, x + plane, H, W, ty0, tx0);
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = w[c * 9 + i];
    __syncthreads();
    const int cx = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * 8;
    const int gx = tx0 + cx;
    if (gx >= W) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int gy = ty0 + r0 + i;
        if (gy < H) out[plane + (long)gy * W + gx] = apply_act(stencil(t, r0 + i, cx, wk), act);
    }
}

// buffer addressing: per-lane byte offsets are computed once (invalid lanes get an out-of-range offset, which
// loads 0 and drops stores) and the channel plane is a scalar offset, so plane walks cost no vector ALU work
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ rsrc_t mk_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(float v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
__device__ __forceinline__ float __attribute__((ext_vector_type(4))) bload4(rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const u4 u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return f4{__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
}
constexpr int HALO = (TH + 2) * LW;
constexpr int HPT = (HALO + 255) / 256;   // 9 halo elements per thread

// One workgroup produces the output pair (2m, 2m+1): both read input channel m for the GELU
// branch, and (C+2m)/2, (C+2m+1)/2 for the gate branch (the same channel when C is even), so every
// input plane is read once instead of twice.  A thread walks 8 rows of one column with a sliding 3x3
// window per plane (3 LDS reads per row and plane instead of 9).
template <bool SAME, bool OBF>
__device__ __forceinline__ void dw_gate_rows(const float* ta, const float* tb0, const float* tb1, const float (&wa0)[9],
                                             const float (&wb0)[9], const float (&wa1)[9], const float (&wb1)[9], bool has1, int r0,
                                             int cx, rsrc_t rout, unsigned voff0, unsigned row4, int rows_ok, unsigned plane0,
                                             unsigned plane1) {
    float a[3][3], p[3][3], q[3][3];
    auto load_row = [&](int hr, int k) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            a[k][dx] = ta[hr * LS + cx + dx];
            p[k][dx] = tb0[hr * LS + cx + dx];
            if (!SAME) q[k][dx] = tb1[hr * LS + cx + dx];
        }
    };
    load_row(r0, 0);
    load_row(r0 + 1, 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        load_row(r0 + i + 2, (i + 2) % 3);
        float s0 = 0.f, g0 = 0.f, s1 = 0.f, g1 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int k = (i + dy) % 3;
                s0 = fmaf(wa0[dy * 3 + dx], a[k][dx], s0);
                g0 = fmaf(wb0[dy * 3 + dx], p[k][dx], g0);
                s1 = fmaf(wa1[dy * 3 + dx], a[k][dx], s1);
                g1 = fmaf(wb1[dy * 3 + dx], SAME ? p[k][dx] : q[k][dx], g1);
            }
        if (i < rows_ok) {                                                   // wave-uniform
#ifndef FDN_GELU_SCALAR
            const fdn_f32x2 ge = gelu_fast2(fdn_f32x2{s0, s1}) * fdn_f32x2{g0, g1};      // both channels of the pair in packed fp32
            st_store1<OBF>(ge.x, rout, voff0, plane0 + (unsigned)i * row4);
            if (has1) st_store1<OBF>(ge.y, rout, voff0, plane1 + (unsigned)i * row4);
#else
            st_store1<OBF>(gelu_fast(s0) * g0, rout, voff0, plane0 + (unsigned)i * row4);
            if (has1) st_store1<OBF>(gelu_fast(s1) * g1, rout, voff0, plane1 + (unsigned)i * row4);
#endif
        }
    }
}

template <bool V4, bool IBF, bool OBF>      // IBF / OBF: x / out are stored as bf16 (fp32 math either way)
__global__ __launch_bounds__(256) void dw_gate_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ out, int C, int H, int W, int tiles_x, int ntiles) {
    constexpr unsigned IES = st_bytes<IBF>(), OES = st_bytes<OBF>();
    const int npairs = (C + 1) / 2;
    const unsigned item = xcd_contiguous(blockIdx.x, gridDim.x);           // (b, output pair, tile), tile fastest
    __shared__ float ta[(TH + 2) * LS + 4];
    __shared__ float tb[2][(TH + 2) * LS + 4];
    const int tile = item % ntiles, m = (item / ntiles) % npairs, b = item / (ntiles * npairs);
    const int j0 = 2 * m, j1 = 2 * m + 1;
    const bool has1 = j1 < C;
    const int cb0 = (C + j0) >> 1, cb1 = (C + j1) >> 1;     // grouped conv: output o reads input o/2
    const bool same = cb1 == cb0 || !has1;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    const unsigned hw4 = (unsigned)H * W * IES, hwo = (unsigned)H * W * OES;
    const rsrc_t rin = mk_rsrc(reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + (long)b * C * H * W * IES), (unsigned)C * hw4);
    const rsrc_t rout = mk_rsrc(reinterpret_cast<const float*>(reinterpret_cast<const char*>(out) + (long)b * C * H * W * OES), (unsigned)C * hwo);

    float wa0[9], wb0[9], wa1[9], wb1[9];                   // scalar loads, requested ahead of the halo so both latencies overlap
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        wa0[i] = w[j0 * 9 + i];
        wb0[i] = w[(C + j0) * 9 + i];
        wa1[i] = has1 ? w[j1 * 9 + i] : 0.f;
        wb1[i] = has1 ? w[(C + j1) * 9 + i] : 0.f;
    }
    if (V4) {
        // 16-byte lanes for the 64 interior columns of the halo rows (W % 4 == 0, planes 16-byte aligned: a float4 is inside
        // or outside the image as a whole), dword loads only for the two edge columns: 4 load instructions per plane
        // instead of 9, and a wave touches 1 KB contiguous
        constexpr int NV = (TH + 2) * 16;                   // 544 float4 of the interior
        float qa[3][4], qb[3][4], qc[3][4];
        int qs[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int r = idx >> 4, c4 = idx & 15;
            const int y = ty0 - 1 + r, xx = tx0 + 4 * c4;
            const bool ok = idx < NV && y >= 0 && y < H && xx < W;
            const unsigned g = ok ? (unsigned)(y * W + xx) * IES : OOB;
            qs[i] = idx < NV ? r * LS + 1 + 4 * c4 : (TH + 2) * LS;        // spare cells (the 4 floats behind the tile)
            st_load4<IBF>(qa[i], rin, g, (unsigned)m * hw4);
            st_load4<IBF>(qb[i], rin, g, (unsigned)cb0 * hw4);
            if (same) { qc[i][0] = qc[i][1] = qc[i][2] = qc[i][3] = 0.f; }
            else st_load4<IBF>(qc[i], rin, g, (unsigned)cb1 * hw4);
        }
        const int er = threadIdx.x >> 1, ec = (threadIdx.x & 1) ? TW + 1 : 0;     // edge columns: threads 0..67
        const int ey = ty0 - 1 + er, ex = tx0 - 1 + ec;
        const bool eok = threadIdx.x < 2 * (TH + 2) && ey >= 0 && ey < H && ex >= 0 && ex < W;
        const unsigned eg = eok ? (unsigned)(ey * W + ex) * IES : OOB;
        const int es = threadIdx.x < 2 * (TH + 2) ? er * LS + ec : (TH + 2) * LS;
        const float ea = st_load1<IBF>(rin, eg, (unsigned)m * hw4), eb = st_load1<IBF>(rin, eg, (unsigned)cb0 * hw4);
        const float ecv = same ? 0.f : st_load1<IBF>(rin, eg, (unsigned)cb1 * hw4);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ta[qs[i] + j] = qa[i][j];
                tb[0][qs[i] + j] = qb[i][j];
                if (!same) tb[1][qs[i] + j] = qc[i][j];
            }
        ta[es] = ea;
        tb[0][es] = eb;
        if (!same) tb[1][es] = ecv;
    } else {
    float va[HPT], vb[HPT], vc[HPT];
    int slot[HPT];
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int r = idx / LW, c = idx - r * LW;
        const int y = ty0 - 1 + r, xx = tx0 - 1 + c;
        const bool ok = idx < HALO && y >= 0 && y < H && xx >= 0 && xx < W;
        const unsigned g = ok ? (unsigned)(y * W + xx) * IES : OOB;
        slot[i] = idx < HALO ? r * LS + c : (TH + 2) * LS;     // spare cell: stores stay unconditional (no sunk load)
        va[i] = st_load1<IBF>(rin, g, (unsigned)m * hw4);
        vb[i] = st_load1<IBF>(rin, g, (unsigned)cb0 * hw4);
        vc[i] = same ? 0.f : st_load1<IBF>(rin, g, (unsigned)cb1 * hw4);
    }
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
        ta[slot[i]] = va[i];
        tb[0][slot[i]] = vb[i];
        if (!same) tb[1][slot[i]] = vc[i];
    }
    }
    __syncthreads();
    const int cx = threadIdx.x & 63;
    const int r0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 8;
    const int gx = tx0 + cx, gy0 = ty0 + r0;
    const unsigned voff0 = (gx < W && gy0 < H) ? (unsigned)(gy0 * W + gx) * OES : OOB;
    const int rows_ok = H - gy0;                              // rows of this wave inside the image (>= 8: all)
    const unsigned row4 = (unsigned)W * OES;
    if (same)
        dw_gate_rows<true, OBF>(ta, tb[0], tb[0], wa0, wb0, wa1, wb1, has1, r0, cx, rout, voff0, row4, rows_ok, (unsigned)j0 * hwo, (unsigned)j1 * hwo);
    else
        dw_gate_rows<false, OBF>(ta, tb[0], tb[1], wa0, wb0, wa1, wb1, has1, r0, cx, rout, voff0, row4, rows_ok, (unsigned)j0 * hwo, (unsigned)j1 * hwo);
}

// mul/add maps: per output channel c:  sum_tap w3[c][tap] * (sum_i w1[c][i] * img[i][p+tap])
// The three image planes' halo tiles are fetched as one batch of buffer loads (zero outside the image, which is
// what the zero padding of the depthwise conv sees since conv1 has no bias); a thread walks 8 rows of one column
// with a sliding 3x3x3 window and evaluates the 8 channels of its workgroup per row.
__global__ __launch_bounds__(256) void img_maps_kernel(const float* __restrict__ img, const float* __restrict__ w1m,
                                                       const float* __restrict__ w3m, const float* __restrict__ w1a,
                                                       const float* __restrict__ w3a, float* __restrict__ mul,
                                                       float* __restrict__ add, int C, int H, int W, int tiles_x, int ntiles) {
    __shared__ float t[3][(TH + 2) * LS + 1];
    const int ncg = (C + 7) / 8;
    const unsigned item = xcd_contiguous(blockIdx.x, gridDim.x);           // (b, channel chunk, tile), tile fastest
    const int tile = item % ntiles, cgi = (item / ntiles) % ncg, b = item / (ntiles * ncg);
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    const unsigned hw4 = (unsigned)H * W * 4u;
    const rsrc_t rin = mk_rsrc(img + (long)b * 3 * H * W, 3u * hw4);
    {
        float v[3][HPT];
        int slot[HPT];
#pragma unroll
        for (int i = 0; i < HPT; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int r = idx / LW, c = idx - r * LW;
            const int y = ty0 - 1 + r, xx = tx0 - 1 + c;
            const bool ok = idx < HALO && y >= 0 && y < H && xx >= 0 && xx < W;
            const unsigned g = ok ? (unsigned)(y * W + xx) * 4u : OOB;
            slot[i] = idx < HALO ? r * LS + c : (TH + 2) * LS;     // spare cell: unconditional stores
#pragma unroll
            for (int p = 0; p < 3; ++p) v[p][i] = bload(rin, g, (unsigned)p * hw4);
        }
#pragma unroll
        for (int i = 0; i < HPT; ++i) {
#pragma unroll
            for (int p = 0; p < 3; ++p) t[p][slot[i]] = v[p][i];
        }
    }
    __syncthreads();
    const int cx = threadIdx.x & 63;
    const int r0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 8;
    const int gx = tx0 + cx, gy0 = ty0 + r0;
    const unsigned voff0 = (gx < W && gy0 < H) ? (unsigned)(gy0 * W + gx) * 4u : OOB;
    const int rows_ok = H - gy0;
    const unsigned row4 = (unsigned)W * 4u;
    const rsrc_t rmul = mk_rsrc(mul + (long)b * C * H * W, (unsigned)C * hw4);
    const rsrc_t radd = mk_rsrc(add + (long)b * C * H * W, (unsigned)C * hw4);
    const int c0 = cgi * 8, nc = min(8, C - c0);                      // channel chunk of this workgroup
    float win[3][3][3];                                                 // [window row][plane][dx]
    auto load_row = [&](int hr, int k) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) win[k][p][dx] = t[p][hr * LS + cx + dx];
    };
    load_row(r0, 0);
    load_row(r0 + 1, 1);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        load_row(r0 + r + 2, (r + 2) % 3);
        if (r >= rows_ok) continue;                                     // wave-uniform
        for (int cc = 0; cc < nc; ++cc) {
            const int c = c0 + cc;
            float am = 0.f, aa = 0.f;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                // inner sum over the 3 image channels first (same association as conv1 then conv3)
                float sm = 0.f, sa = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float v = win[(r + tp / 3) % 3][i][tp % 3];
                    sm = fmaf(w1m[c * 3 + i], v, sm);
                    sa = fmaf(w1a[c * 3 + i], v, sa);
                }
                am = fmaf(w3m[c * 9 + tp], sm, am);
                aa = fmaf(w3a[c * 9 + tp], sa, aa);
            }
            bstore(am, rmul, voff0, (unsigned)c * hw4 + (unsigned)r * row4);
            bstore(aa, radd, voff0, (unsigned)c * hw4 + (unsigned)r * row4);
        }
    }
}

}  // namespace

extern "C" int fdn_dwconv3x3(const float* x, const float* w, float* out, int B, int C, int H, int W, int act,
                             fdn_stream_t stream) {
    FDN_CHECK_ARG(x && w && out && B > 0 && C > 0 && H > 0 && W > 0 && C < 65536 && B < 65536);
    const int tx = cdiv(W, TW), ty = cdiv(H, TH);
    hipLaunchKernelGGL(dw3x3_kernel, dim3(tx * ty, C, B), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, out, C, H, W,
                       act, tx);
    return fdn_launch_status();
}

extern "C" int fdn_dwconv_gate(const void* x_, const float* w, void* out_, int B, int C, int H, int W, int x_bf16, int out_bf16,
                               fdn_stream_t stream) {
    const float* x = static_cast<const float*>(x_);
    float* out = static_cast<float*>(out_);
    FDN_CHECK_ARG(x && w && out && B > 0 && C > 0 && H > 0 && W > 0 && C < 65536 && B < 65536);
    FDN_CHECK_ARG(4ull * C * H * W < 0x80000000ull);          // one image's C planes are addressed with 32-bit byte offsets
    const int tx = cdiv(W, TW), ty = cdiv(H, TH);
    const bool v4 = W % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const dim3 grid((unsigned)(tx * ty) * ((C + 1) / 2) * B);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define FDN_GATE(V, I, O) hipLaunchKernelGGL((dw_gate_kernel<V, I, O>), grid, dim3(256), 0, s, x, w, out, C, H, W, tx, tx * ty)
    if (v4) {
        if (x_bf16 && out_bf16) FDN_GATE(true, true, true);
        else if (x_bf16) FDN_GATE(true, true, false);
        else if (out_bf16) FDN_GATE(true, false, true);
        else FDN_GATE(true, false, false);
    } else {
        if (x_bf16 && out_bf16) FDN_GATE(false, true, true);
        else if (x_bf16) FDN_GATE(false, true, false);
        else if (out_bf16) FDN_GATE(false, false, true);
        else FDN_GATE(false, false, false);
    }
#undef FDN_GATE
    return fdn_launch_status();
}

extern "C" int fdn_img_mod_maps(const float* img, const float* w1_mul, const float* w3_mul, const float* w1_add,
                                const float* w3_add, float* mul, float* add, int B, int C, int H, int W,
                                fdn_stream_t stream) {
    FDN_CHECK_ARG(img && w1_mul && w3_mul && w1_add && w3_add && mul && add && B > 0 && C > 0 && H > 0 && W > 0);
    FDN_CHECK_ARG(4ull * C * H * W < 0x80000000ull);          // one image's C planes are addressed with 32-bit byte offsets
    const int tx = cdiv(W, TW), ty = cdiv(H, TH);
    hipLaunchKernelGGL(img_maps_kernel, dim3((unsigned)(tx * ty) * cdiv(C, 8) * B), dim3(256), 0, static_cast<hipStream_t>(stream), img,
                       w1_mul, w3_mul, w1_add, w3_add, mul, add, C, H, W, tx, tx * ty);
    return fdn_launch_status();
}
