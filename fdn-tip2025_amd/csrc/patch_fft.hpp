// Shared pieces of the 8x8-patch spectral kernels (patchfft.hip; tools/experiments/fdffn_fused.hip): 8-point transforms on (re, im) pairs,
// the real row transforms of an 8 x 8 patch, spectrum strides, raw-buffer helpers.
#pragma once
#include "common.hpp"

namespace {

constexpr int TH = 32, TW = 64;
constexpr int NP = 32;             // patches per tile
constexpr int KXS = 9;             // stride between the kx columns of a patch spectrum, in float2 (8 used + 1 pad)
constexpr int PS = 5 * KXS;        // patch stride: column-phase thread t = 5 * patch + kx sits at 9 t float2 = 18 t dwords, so the
                                   // 8-byte accesses of 16 / 32 consecutive threads fall into distinct banks (with 8 / 41 the kx = 0
                                   // and kx = 4 columns of a patch shared their banks: 2-3x the LDS cycles in the column phase)
constexpr float C8 = 0.70710678118654752440f;

// The three 8-point transforms are written on (re, im) PAIRS (ext-vector float2): a complex add / subtract is one
// v_pk_add_f32, a multiplication by +-i a swizzle the compiler folds into op_sel / neg modifiers, a real scale one v_pk_mul_f32.
// On gfx950 a wave64 v_pk_* instruction takes ~4.5 issue cycles against 4 for a scalar one (tools/micro/mfma_valu_coexec.hip), so
// the packed forms cost a little over half the vector-ALU time of the component-wise ones (fft8: 37 instructions instead of 64).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mul_pi(f2 a) { return f2{-a.y, a.x}; }      // * (+i)
__device__ __forceinline__ f2 mul_ni(f2 a) { return f2{a.y, -a.x}; }      // * (-i)
__device__ __forceinline__ f2 cconj(f2 a) { return f2{a.x, -a.y}; }
__device__ __forceinline__ f2 tof2(float2 a) { return f2{a.x, a.y}; }
__device__ __forceinline__ float2 tofloat2(f2 a) { return make_float2(a.x, a.y); }

// in-place 8-point complex FFT, natural order in and out.  INV: e^{+...}, unscaled.
template <bool INV>
__device__ __forceinline__ void fft8(float2 (&vv)[8]) {
    f2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = tof2(vv[i]);
    auto rot = [](f2 d) { return INV ? mul_pi(d) : mul_ni(d); };         // * e^{-+ i pi/2}
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = v[i] + v[i + 4];
        const f2 d = v[i] - v[i + 4];
        if (i == 0) a[4] = d;
        else if (i == 1) a[5] = C8 * (d + rot(d));                         // * e^{-+ i pi/4}
        else if (i == 2) a[6] = rot(d);
        else a[7] = C8 * (rot(d) - d);                                     // * e^{-+ 3 i pi/4}
    }
    f2 c[8];
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
        c[h] = a[h] + a[h + 2];
        c[h + 2] = a[h] - a[h + 2];
        c[h + 1] = a[h + 1] + a[h + 3];
        c[h + 3] = rot(a[h + 1] - a[h + 3]);
    }
    constexpr int br[8] = {0, 4, 2, 6, 1, 5, 3, 7};
#pragma unroll
    for (int h = 0; h < 8; h += 2) {
        vv[br[h]] = tofloat2(c[h] + c[h + 1]);
        vv[br[h + 1]] = tofloat2(c[h] - c[h + 1]);
    }
}

// forward real row transform: 8 reals -> bins 0..4, through one 4-point complex FFT of
// z[n] = x[2n] + i x[2n+1] and the split  X[k] = E[k] + W8^k O[k]
__device__ __forceinline__ void rfft8_row(const float (&x)[8], float2 (&o)[5]) {
    const f2 z0 = {x[0], x[1]}, z1 = {x[2], x[3]}, z2 = {x[4], x[5]}, z3 = {x[6], x[7]};
    const f2 s02 = z0 + z2, d02 = z0 - z2, s13 = z1 + z3, d13 = z1 - z3;
    const f2 Z0 = s02 + s13, Z2 = s02 - s13;
    const f2 Z1 = d02 + mul_ni(d13);                  // d02 - i d13
    const f2 Z3 = d02 + mul_pi(d13);                  // d02 + i d13
    o[0] = make_float2(Z0.x + Z0.y, 0.f);
    o[4] = make_float2(Z0.x - Z0.y, 0.f);
    o[2] = tofloat2(cconj(Z2));
    // k = 1: E = (Z1 + conj Z3)/2, D = (Z1 - conj Z3)/2, O = -i D, X1 = E + W8 O, W8 = (c, -c);  k = 3: X3 = conj(E - W8 O)
    const f2 cz3 = cconj(Z3);
    const f2 e = 0.5f * (Z1 + cz3), d = 0.5f * (Z1 - cz3);
    const f2 oo = mul_ni(d);
    const f2 t = C8 * (oo + mul_ni(oo));              // W8 * O = c (oo.x + oo.y, oo.y - oo.x)
    o[1] = tofloat2(e + t);
    o[3] = tofloat2(cconj(e - t));
}

// inverse c2r row transform from bins 0..4 (imag of bins 0 and 4 ignored, like pocketfft/MKL c2r),
// unscaled: returns 8 * x, through one 4-point complex inverse FFT
__device__ __forceinline__ void irfft8_row(const float2 (&X)[5], float (&x)[8]) {
    const f2 Z0 = {X[0].x + X[4].x, X[0].x - X[4].x};
    const f2 Z2 = 2.f * cconj(tof2(X[2]));
    // k = 1: E' = X1 + conj X3, D' = X1 - conj X3, O' = D' * (c, c), Z1 = E' + i O';  k = 3: Z3 = conj(E') + i O3', O3' = conj-mirrored
    const f2 x1 = tof2(X[1]), cx3 = cconj(tof2(X[3]));
    const f2 e1 = x1 + cx3, d1 = x1 - cx3;
    const f2 o1 = C8 * (d1 + mul_pi(d1));             // c (d1.x - d1.y, d1.x + d1.y)
    const f2 Z1 = e1 + mul_pi(o1);                    // (e1.x - o1.y, e1.y + o1.x)
    const f2 Z3 = cconj(e1 - mul_pi(o1));             // the k = 3 bin of the half-length transform
    const f2 s02 = Z0 + Z2, d02 = Z0 - Z2, s13 = Z1 + Z3, d13 = Z1 - Z3;
    const f2 r0 = s02 + s13, r2 = s02 - s13;
    const f2 r1 = d02 + mul_pi(d13);                  // d02 + i d13
    const f2 r3 = d02 + mul_ni(d13);                  // d02 - i d13
    x[0] = r0.x; x[1] = r0.y; x[2] = r1.x; x[3] = r1.y; x[4] = r2.x; x[5] = r2.y; x[6] = r3.x; x[7] = r3.y;
}

__device__ __forceinline__ float rsq(float v) { return __builtin_amdgcn_rsqf(v); }

// One bin (ky of a column kx) of FDSA's recombination, FDN_arch.py:591-629: from the column spectra q, k, v and the gain f = self.fft to
// u = e^{i(qp - kp)}, v1 = replace_denormals(v f), |qk|, |qk| / |v| and |v| (out1 = |v| u, out2 = |qk| / |v| v1, out3 = |qk| u).
//   AS_WRITTEN: the four replace_denormals of :593, :597, :603, :604 on every component (16 compare / select instructions per bin).
//   else      : replace_denormals touches a component only when its magnitude is below 1e-10 - in practice the imaginary parts of the
//               self-conjugate bins (ky = 0, 4 of the columns kx = 0, 4: exactly 0) and nothing else.  Those four are replaced as written
//               (SELF_CONJ = ky % 4 == 0); of the other components only the minimum magnitude is tracked in `m`.  The caller takes a wave
//               vote on m < 1e-10 behind the column and, if some lane needed a replacement, evaluates the column again AS_WRITTEN: the
//               result is the same bit for bit, the common path issues 45 instructions for the four calls instead of 128.
// The products that are fused into an fma are spelled out: left to the compiler's contraction the two evaluation paths round differently (its
// choice depends on the shape of the code around the expression).  The forms below are the ones it had chosen for the as-written code of
// rounds 1-4 - a b + c d = fma(a, b, c d) everywhere, except the real part of q k, which came out as two rounded products and a subtraction -
// so the results are those kernels' bit for bit (tools/ab_libs.py, profiles/r05_n_vote.txt).
__device__ __forceinline__ float fdn_dot2(float a, float b, float c, float d) { return fmaf(a, b, c * d); }      // a b + c d
__device__ __forceinline__ float fdn_sub_prod(float a, float b, float c, float d) {                             // a b - c d, both products rounded
#pragma clang fp contract(off)
    const float p = a * b, r = c * d;
    return p - r;
}
template <bool AS_WRITTEN, bool SELF_CONJ>
__device__ __forceinline__ void fdsa_bin(float2 q, float2 k, float2 v, float f, float& m, float2& u, float2& v1, float& qka, float& g, float& va) {
    v1 = make_float2(v.x * f, v.y * f);                                               // :591
    float2 qk = make_float2(fdn_sub_prod(q.x, k.x, q.y, k.y), fdn_dot2(q.x, k.y, q.y, k.x));      // :595
    if constexpr (AS_WRITTEN) {
        v1 = make_float2(rd1(v1.x), rd1(v1.y));                                       // :593
        qk = make_float2(rd1(qk.x), rd1(qk.y));                                       // :597
        q = make_float2(rd1(q.x), rd1(q.y));                                          // :603
        k = make_float2(rd1(k.x), rd1(k.y));                                          // :604
    } else if constexpr (SELF_CONJ) {
        v1.y = rd1(v1.y); qk.y = rd1(qk.y); q.y = rd1(q.y); k.y = rd1(k.y);
        m = fminf(fminf(m, fabsf(v1.x)), fabsf(qk.x));
        m = fminf(fminf(m, fabsf(q.x)), fabsf(k.x));
    } else {
        m = fminf(fminf(m, fabsf(v1.x)), fabsf(v1.y));
        m = fminf(fminf(m, fabsf(qk.x)), fabsf(qk.y));
        m = fminf(fminf(m, fabsf(q.x)), fabsf(q.y));
        m = fminf(fminf(m, fabsf(k.x)), fabsf(k.y));
    }
    const float qk2 = fdn_dot2(qk.x, qk.x, qk.y, qk.y), v2 = fdn_dot2(v1.x, v1.x, v1.y, v1.y);
    qka = qk2 * rsq(qk2);                                                             // |qk|  :599
    const float iv = rsq(v2);
    va = v2 * iv;                                                                     // |v|   :601
    const float nq = rsq(fdn_dot2(q.x, q.x, q.y, q.y)), nk = rsq(fdn_dot2(k.x, k.x, k.y, k.y));
    const float2 a = make_float2(q.x * nq, q.y * nq), b = make_float2(k.x * nk, k.y * nk);
    u = make_float2(fdn_dot2(a.x, b.x, a.y, b.y), fdn_dot2(a.y, b.x, -a.x, b.y));            // a conj(b) = e^{i(qp - kp)}  :605-607
    g = qka * iv;
}

// buffer resources: per-lane byte offsets are computed once per workgroup (invalid lanes get an offset past
// num_records, which loads as 0 and drops stores); the channel plane is a scalar offset, so walking planes
// costs no vector ALU work (these kernels are VALU-issue bound, not HBM bound)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;       // images are limited to < 2 GB per tensor so that OOB (+ small immediates) stays out of range
__device__ __forceinline__ rsrc_t mk_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore8(const float (&v)[8], rsrc_t r, unsigned voff, unsigned soff) {
    u32x4 a, b;
    a.x = __float_as_uint(v[0]); a.y = __float_as_uint(v[1]); a.z = __float_as_uint(v[2]); a.w = __float_as_uint(v[3]);
    b.x = __float_as_uint(v[4]); b.y = __float_as_uint(v[5]); b.z = __float_as_uint(v[6]); b.w = __float_as_uint(v[7]);
    __builtin_amdgcn_raw_buffer_store_b128(a, r, voff, soff, 0);
    __builtin_amdgcn_raw_buffer_store_b128(b, r, voff + 16u, soff, 0);
}

}  // namespace
