// Shared helpers for the FDN gfx950 kernels (device math, launch checks, error codes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fdn_hip.h"

#define FDN_CHECK_ARG(cond) \
    do {                    \
        if (!(cond)) return FDN_ERR_ARG; \
    } while (0)

static inline int fdn_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FDN_OK : FDN_ERR_LAUNCH;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Launch-time device facts, cached per device ordinal (a process may drive several GPUs from several threads; plain
// function-local statics would freeze the first device's answer).  fdn_device_cus: compute units of the CURRENT device
// (<= 0 on error).  fdn_allow_dynamic_lds: raise a kernel's dynamic-LDS limit once per (kernel, device).
int fdn_device_cus();
bool fdn_allow_dynamic_lds(const void* kernel, size_t bytes);
bool fdn_matrix_pipe_f32();                                   // fdn_set_matrix_pipe(1): no bf16-MFMA kernel is launched
void fdn_note_bf16_launch();                                  // called by every launcher of a bf16-MFMA kernel (fdn_bf16_mfma_launches)
bool fdn_matrix_pipe_wide();                                  // default mode: fdn_fdsa_out's level-2 shape on the bf16 pipe too (mode 2 = the ABI-10 default keeps it on fp32 MFMAs)
bool fdn_occupancy(int* blocks_per_cu, const void* kernel, int threads, size_t lds);   // hipOccupancyMaxActiveBlocksPerMultiprocessor, cached

__device__ __forceinline__ float gelu_erf(float x) {
    // F.gelu default (erf form), FDN_arch.py:427,438,473
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// GELU for the HBM/VALU-bound stencil kernels: erf via Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7 on
// erf, i.e. at fp32 rounding level on 1+erf), 1+erf formed without cancellation on the negative side.
// ~12 VALU instructions instead of ~40 for erff.
__device__ __forceinline__ float gelu_fast(float x) {
    // 0.5 x erfc(-x / sqrt 2) with erfc(z) = t (a1 + t (a2 + ...)) e^{-z^2}, t = 1 / (1 + p z), z = |x| / sqrt 2: the scale of z is
    // folded into p and into the exponent (e^{-x^2/2} = 2^{-(k x)^2}, k^2 = log2(e) / 2), the 0.5 into the coefficients: 15
    // instructions, two of them transcendental (v_rcp_f32, v_exp_f32)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x), 1.0f));
    const float poly = t * (0.5f * 0.254829592f + t * (0.5f * -0.284496736f + t * (0.5f * 1.421413741f + t * (0.5f * -1.453152027f + t * (0.5f * 1.061405429f)))));
    const float u = x * 0.84932180028801904272f;                      // sqrt(log2(e) / 2)
    const float pe = poly * __builtin_amdgcn_exp2f(-(u * u));         // 0.5 erfc(|x| / sqrt 2)
    return x * (x >= 0.f ? 1.0f - pe : pe);
}

// The same GELU on a PAIR of values: the polynomial, the argument scaling and the final products run as packed fp32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32: ~4.5 issue cycles for two lanes of work against 4 for one), the two reciprocals and the two
// exponentials stay scalar: 21 instructions per pair instead of 30.  Bit-identical to gelu_fast per component.
typedef float fdn_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fdn_f32x2 gelu_fast2(fdn_f32x2 x) {
    const fdn_f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    const fdn_f32x2 den = __builtin_elementwise_fma(ax, fdn_f32x2(0.3275911f * 0.70710678118654752440f), fdn_f32x2(1.0f));
    const fdn_f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    fdn_f32x2 p = __builtin_elementwise_fma(t, fdn_f32x2(0.5f * 1.061405429f), fdn_f32x2(0.5f * -1.453152027f));
    p = __builtin_elementwise_fma(t, p, fdn_f32x2(0.5f * 1.421413741f));
    p = __builtin_elementwise_fma(t, p, fdn_f32x2(0.5f * -0.284496736f));
    p = __builtin_elementwise_fma(t, p, fdn_f32x2(0.5f * 0.254829592f));
    const fdn_f32x2 poly = t * p;
    const fdn_f32x2 u = x * 0.84932180028801904272f;
    const fdn_f32x2 u2 = u * u;
    const fdn_f32x2 ex = {__builtin_amdgcn_exp2f(-u2.x), __builtin_amdgcn_exp2f(-u2.y)};
    const fdn_f32x2 pe = poly * ex;
    const fdn_f32x2 om = 1.0f - pe;
    const fdn_f32x2 f = {x.x >= 0.f ? om.x : pe.x, x.y >= 0.f ? om.y : pe.y};
    return x * f;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    // act is wave-uniform: every test below is a scalar compare-and-branch PER VALUE, so the common kinds come first (the
    // switch form walked ~8 branches even for FDN_ACT_NONE).  Kernels with long epilogues resolve act once per tile instead.
    if (act == FDN_ACT_NONE) return v;
    if (act <= FDN_ACT_RELU) return v > 0.f ? v : (act == FDN_ACT_LEAKY ? 0.1f * v : 0.f);   // LeakyReLU(0.1) FDN_arch.py:28 / ReLU
    return act == FDN_ACT_SIGMOID ? 1.0f / (1.0f + expf(-v)) : gelu_erf(v);
}

// sin and cos together, ~1 ulp over the whole float range and without a library call (a call site costs the caller its
// register allocation: the column FFT kernel evaluates 32 of these in straight-line code).
//   |x| < 8192 : three-constant Cody-Waite reduction by pi/2 in fp32
//   otherwise  : |x| = m * 2^e with a 24-bit integer m, and x * 2/pi mod 4 = m * T[e] mod 4 with T[e] = (2^e * 2/pi) mod 4
//                tabulated as double-double (sincos_table.inc, e = -10 .. 104): one exact fp64 product instead of a
//                Payne-Hanek loop; the reduced argument is good to ~2^-60.  Inf / NaN give NaN.
// then the classic single-precision minimax polynomials on [-pi/4, pi/4].
static __device__ const double fdn_two_over_pi_mod4[115][2] = {
#include "sincos_table.inc"
};
// FULL = false: the fp32 reduction only, valid for |x| < 8192 (callers that evaluate many arguments in straight-line code
// run this form, note whether any argument was out of range and redo that rare case with the full form)
template <bool FULL = true>
__device__ __forceinline__ void fdn_sincos(float x, float* sn, float* cs) {
    float r;
    int q;
    if (!FULL || __builtin_expect(fabsf(x) < 8192.0f, 1)) {
        const float k = rintf(x * 0.63661977236758134308f);
        r = fmaf(k, -1.5707962513e+0f, x);
        r = fmaf(k, -7.5497894159e-08f, r);
        r = fmaf(k, -5.3903029534e-15f, r);
        q = (int)k;
    } else {
        const unsigned bits = __float_as_uint(x) & 0x7FFFFFFFu;
        const int eb = (int)(bits >> 23);                               // >= 140: |x| = m * 2^(eb - 150)
        const double m = (double)(int)((bits & 0x7FFFFFu) | 0x800000u);
        const int idx = (eb > 254 ? 254 : eb) - 140;
        const double fh = fdn_two_over_pi_mod4[idx][0], fl = fdn_two_over_pi_mod4[idx][1];
        const double p = m * fh, pe = fma(m, fh, -p);                   // p + pe = m * fh exactly, p < 2^26
        const double kq = rint(p);
        const double fr = (p - kq) + fma(m, fl, pe);                    // |fr| <= 1/2 (+ 2^-27)
        r = (float)(fr * 1.57079632679489661923);
        q = (int)kq;
        if (x < 0.0f) { r = -r; q = -q; }
        if (eb == 255) r = __uint_as_float(0x7FC00000u);
    }
    const float r2 = r * r;
    const float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    const float s0 = fmaf(r * r2, ps, r);
    const float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    const float c0 = fmaf(r2 * r2, pc, fmaf(-0.5f, r2, 1.0f));
    const float a = (q & 1) ? c0 : s0, b = (q & 1) ? s0 : c0;
    *sn = (q & 2) ? -a : a;
    *cs = ((q + 1) & 2) ? -b : b;
}

// replace_denormals on one component: (-1e-10, 1e-10) incl. +-0 -> +1e-10 (FDN_arch.py:548-553)
__device__ __forceinline__ float rd1(float v) { return (v < 1e-10f && v > -1e-10f) ? 1e-10f : v; }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {  // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float cabs2(float2 a) { return sqrtf(a.x * a.x + a.y * a.y); }

// Workgroup -> work item for tiled kernels.  The dispatcher deals workgroups round-robin over the 8 XCDs (blocks b and b + 8
// share one, MI355X_MICROARCH.md), so with the natural order horizontally adjacent tiles of a plane land on different XCDs and
// the 128-byte lines both touch (a halo row of 64 + 2 floats spans 4 lines, 2 of them shared) are fetched from the fabric twice:
// the stencil kernels read 2.1-2.2x their input that way (profiles/r02_a_traffic_groups.json).  xcd_contiguous maps block L of T
// to item S so that every XCD walks a contiguous run of items in dispatch order; callers decode S with the tile index fastest.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned L, unsigned T) {
    const unsigned xcd = L & 7u, idx = L >> 3, per = T >> 3, rem = T & 7u;
    return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

// The same for a 3-D grid of workgroups whose x axis walks the pixels of a plane (the direct convs of convs.hip): the linear
// workgroup id x + gx (y + gy z) decides the XCD, so it is the linear id that is remapped and then cut back into (x, y, z).  With the
// plain order the 256-pixel blocks either side of a halo row sit on different XCDs and every input row is fetched by three L2s
// (conv2d[32->3] read 3.2x its bytes, profiles/r03_n_summary.txt).
__device__ __forceinline__ void fdn_xcd_block3(unsigned& bx, unsigned& by, unsigned& bz) {
    const unsigned gx = gridDim.x, gy = gridDim.y, T = gx * gy * gridDim.z;
    const unsigned S = xcd_contiguous(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), T);
    bx = S % gx;
    const unsigned r = S / gx;
    by = r % gy;
    bz = r / gy;
}

// ------------------------------------------------------------------------------------------------
// bf16 STORAGE of block-internal activations (BASELINE.json configs[2]).  bf16 is only a
// memory format here: a value is widened to fp32 when loaded (exact) and rounded to nearest-even when stored
// (v_cvt_pk_bf16_f32, NaN stays NaN); every product, sum, FFT and LayerNorm statistic is fp32.
// ------------------------------------------------------------------------------------------------
typedef unsigned fdn_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned fdn_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
// element loads / stores through a raw buffer resource; `voff` / `soff` are BYTE offsets of the storage type
template <bool BF>
__device__ __forceinline__ float st_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (BF) return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0) << 16);
    else return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <bool BF>
__device__ __forceinline__ void st_store1(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (BF) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pack_bf16(v, 0.f) & 0xFFFFu), r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
template <bool BF>
__device__ __forceinline__ void st_load2(float (&v)[2], __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (BF) {
        const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
        v[0] = bf16_lo(u);
        v[1] = bf16_hi(u);
    } else {
        const fdn_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        v[0] = __uint_as_float(u.x);
        v[1] = __uint_as_float(u.y);
    }
}
template <bool BF>
__device__ __forceinline__ void st_store2(const float (&v)[2], __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (BF) __builtin_amdgcn_raw_buffer_store_b32(pack_bf16(v[0], v[1]), r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(fdn_u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}, r, voff, soff, 0);
}
template <bool BF>
__device__ __forceinline__ void st_load4(float (&v)[4], __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (BF) {
        const fdn_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        v[0] = bf16_lo(u.x); v[1] = bf16_hi(u.x); v[2] = bf16_lo(u.y); v[3] = bf16_hi(u.y);
    } else {
        const fdn_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        v[0] = __uint_as_float(u.x); v[1] = __uint_as_float(u.y); v[2] = __uint_as_float(u.z); v[3] = __uint_as_float(u.w);
    }
}
#ifndef FDN_ST_AUX
#define FDN_ST_AUX 0      // cache policy bits of the 32-byte row-segment stores (A/B: 2 = non-temporal)
#endif
template <bool BF>
__device__ __forceinline__ void st_store8(const float (&v)[8], __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (BF) {
        __builtin_amdgcn_raw_buffer_store_b128(fdn_u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])},
                                               r, voff, soff, FDN_ST_AUX);
    } else {
        __builtin_amdgcn_raw_buffer_store_b128(fdn_u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                                               r, voff, soff, FDN_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(fdn_u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])},
                                               r, voff + 16u, soff, 0);
    }
}
template <bool BF> constexpr unsigned st_bytes() { return BF ? 2u : 4u; }

// ------------------------------------------------------------------------------------------------
// fp32 operands for the bf16 matrix pipe (gemm_split.hip has the why).  An fp32 value is cut EXACTLY into three bf16
// values by truncation: x = x1 + x2 + x3 (8 + 8 + 8 significant bits), and w x = w1 x1 + (w1 x2 + w2 x1) +
// (w1 x3 + w2 x2 + w3 x1) to below one fp32 rounding.  fdn_split3 cuts the PAIR (a, b) and packs each part as
// (a | b << 16), the order of two consecutive k of an MFMA operand.
// ------------------------------------------------------------------------------------------------
typedef float fdn_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float fdn_trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ unsigned fdn_pack_hi16(float a, float b) {
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ void fdn_split3(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = fdn_pack_hi16(a, b);
    const float ra = a - fdn_trunc_bf16(a), rb = b - fdn_trunc_bf16(b);
    p2 = fdn_pack_hi16(ra, rb);
    p3 = fdn_pack_hi16(ra - fdn_trunc_bf16(ra), rb - fdn_trunc_bf16(rb));
}
__device__ __forceinline__ fdn_f32x16 fdn_mfma_bf16(fdn_u32x4 a, fdn_u32x4 b, fdn_f32x16 c) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// NOTE (MI355X, ROCm 7.2): a kernel that issues v_mfma_f32_32x32x16_bf16 must not share the GPU with kernels of
// ANOTHER HIP stream - the neighbours (this library's row FFT, rocFFT, a channel LayerNorm) then return wrong rows in up to half of
// their launches; a compiler-generated loop of nothing but those MFMAs is enough (tools/cross_stream_probe.py), fp32 MFMA and
// vector-ALU neighbours are harmless.  Hence one stream per GPU for this path (fdn_hip/pipeline.py).
// the six leading products of one 16-deep k-step, small terms first
__device__ __forceinline__ fdn_f32x16 fdn_mfma_split6(const fdn_u32x4 (&a)[3], const fdn_u32x4 (&b)[3], fdn_f32x16 c) {
    c = fdn_mfma_bf16(a[2], b[0], c);
    c = fdn_mfma_bf16(a[1], b[1], c);
    c = fdn_mfma_bf16(a[0], b[2], c);
    c = fdn_mfma_bf16(a[1], b[0], c);
    c = fdn_mfma_bf16(a[0], b[1], c);
    return fdn_mfma_bf16(a[0], b[0], c);
}
