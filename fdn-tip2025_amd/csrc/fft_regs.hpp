// Register-resident FFT building blocks with compile-time plans: every index, root of unity and butterfly of a
// transform is known when the kernel is compiled, so a pass costs its arithmetic and nothing else (the generic Stockham
// passes of fft2d.hip spend 3-4 vector instructions on index arithmetic, masks and LDS round trips per useful one).
//
//   dft_odd_c<R, INV>     R-point DFT of an odd R on (re, im) pairs, roots folded into the instruction stream
//   fft_dif<N, INV>       radix-2 decimation in frequency: natural order in, bit-reversed order out
//   fft_dit<N, INV>       radix-2 decimation in time: bit-reversed order in, natural order out
//   dft_nat<N, INV>       any small N (2^a 3^b 5^c ...), natural order in and out, radix 4 where it divides
//
// A forward fft_dif followed by a pointwise operation and an inverse fft_dit therefore needs no permutation at all.
// Values are packed (re, im) ext-vectors so adds / scalar multiplies become v_pk_* instructions.
#pragma once
#include <type_traits>

#include "common.hpp"

namespace fftr {

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mul_pi(f2 a) { return f2{-a.y, a.x}; }      // * (+i)
__device__ __forceinline__ f2 mul_ni(f2 a) { return f2{a.y, -a.x}; }      // * (-i)
__device__ __forceinline__ f2 cmul(f2 a, f2 b) { return f2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ f2 cmulc(f2 a, f2 b) { return f2{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }   // a * conj(b)

template <int B, int E, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        sfor<B + 1, E>(f);
    }
}

// ---- cos / sin of 2 pi m / n in double at compile time: octant reduction, then Taylor series on [0, pi/4] ----
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double sin_series(double x) {
    double x2 = x * x, term = x, sum = x;
    for (int n = 1; n <= 12; ++n) { term *= -x2 / (double)((2 * n) * (2 * n + 1)); sum += term; }
    return sum;
}
constexpr double cos_series(double x) {
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int n = 1; n <= 12; ++n) { term *= -x2 / (double)((2 * n - 1) * (2 * n)); sum += term; }
    return sum;
}
constexpr double sin2pi(long m, long n);
constexpr double cos2pi(long m, long n) {
    m %= n; if (m < 0) m += n;
    if (m == 0) return 1.0;
    if (2 * m > n) return cos2pi(n - m, n);                 // cos(2 pi - x)
    if (4 * m == n) return 0.0;
    if (2 * m == n) return -1.0;
    if (4 * m > n) return -cos2pi(n - 2 * m, 2 * n);        // cos(pi - y) = -cos y
    if (8 * m > n) return sin_series(2.0 * kPi * (double)(n - 4 * m) / (double)(4 * n));   // cos x = sin(pi/2 - x)
    return cos_series(2.0 * kPi * (double)m / (double)n);
}
constexpr double sin2pi(long m, long n) {
    m %= n; if (m < 0) m += n;
    if (m == 0 || 2 * m == n) return 0.0;
    if (2 * m > n) return -sin2pi(n - m, n);
    if (4 * m == n) return 1.0;
    if (4 * m > n) return sin2pi(n - 2 * m, 2 * n);         // sin(pi - y) = sin y
    if (8 * m > n) return cos_series(2.0 * kPi * (double)(n - 4 * m) / (double)(4 * n));
    return sin_series(2.0 * kPi * (double)m / (double)n);
}

// v * W_N^M with W_N = e^{-2 pi i / N} (INV: e^{+2 pi i / N}); the axis roots cost no multiply
template <int N, int M, bool INV>
__device__ __forceinline__ f2 mul_root(f2 v) {
    constexpr int m = ((M % N) + N) % N;
    if constexpr (m == 0) return v;
    else if constexpr (2 * m == N) return -v;
    else if constexpr (4 * m == N) return INV ? mul_pi(v) : mul_ni(v);
    else if constexpr (4 * m == 3 * N) return INV ? mul_ni(v) : mul_pi(v);
    else {
        constexpr float c = (float)cos2pi(m, N);
        constexpr float s = (float)(INV ? sin2pi(m, N) : -sin2pi(m, N));
        return v * c + mul_pi(v) * s;                        // (v.x c - v.y s, v.y c + v.x s)
    }
}

constexpr int brev(int k, int n) {          // bit reversal of k within log2(n) bits
    int r = 0;
    for (int b = 1; b < n; b <<= 1) { r = (r << 1) | (k & 1); k >>= 1; }
    return r;
}

template <int N, bool INV>
__device__ __forceinline__ void fft_dif(f2* u) {
    if constexpr (N >= 2) {
        sfor<0, N / 2>([&](auto j) {
            constexpr int J = decltype(j)::value;
            const f2 a = u[J], b = u[J + N / 2];
            u[J] = a + b;
            u[J + N / 2] = mul_root<N, J, INV>(a - b);
        });
        fft_dif<N / 2, INV>(u);
        fft_dif<N / 2, INV>(u + N / 2);
    }
}

template <int N, bool INV>
__device__ __forceinline__ void fft_dit(f2* u) {
    if constexpr (N >= 2) {
        fft_dit<N / 2, INV>(u);
        fft_dit<N / 2, INV>(u + N / 2);
        sfor<0, N / 2>([&](auto j) {
            constexpr int J = decltype(j)::value;
            const f2 a = u[J], b = mul_root<N, J, INV>(u[J + N / 2]);
            u[J] = a + b;
            u[J + N / 2] = a - b;
        });
    }
}

// R-point DFT of an odd R, natural order in and out.  Conjugate symmetry of the roots: with a_r = u_r + u_{R-r},
// b_r = u_r - u_{R-r} (r = 1..h),  X_o = P_o + i S_o,  X_{R-o} = P_o - i S_o,  P_o = u_0 + sum_r a_r cos(2 pi r o / R),
// S_o = sum_r b_r Im(w^{r o}):  (R-1)^2 / 2 packed FMAs.
template <int R, bool INV>
__device__ __forceinline__ void dft_odd_c(f2* u) {
    constexpr int h = (R - 1) / 2;
    f2 a[h], b[h];
    f2 x0 = u[0];
    sfor<1, h + 1>([&](auto r) {
        constexpr int Rr = decltype(r)::value;
        a[Rr - 1] = u[Rr] + u[R - Rr];
        b[Rr - 1] = u[Rr] - u[R - Rr];
        x0 += a[Rr - 1];
    });
    const f2 u0 = u[0];
    sfor<1, h + 1>([&](auto o) {
        constexpr int O = decltype(o)::value;
        f2 Pv = u0, S = f2{0.f, 0.f};
        sfor<1, h + 1>([&](auto r) {
            constexpr int Rr = decltype(r)::value;
            constexpr int m = (Rr * O) % R;
            constexpr float wx = (float)cos2pi(m, R);
            constexpr float wy = (float)(INV ? sin2pi(m, R) : -sin2pi(m, R));
            Pv += a[Rr - 1] * wx;
            S += b[Rr - 1] * wy;
        });
        u[O] = Pv + mul_pi(S);
        u[R - O] = Pv - mul_pi(S);
    });
    u[0] = x0;
}


// N-point DFT of any small N, natural order in and out: Cooley-Tukey on the smallest prime factor, every twiddle a constant.
constexpr int spf(int n) {
    for (int f = 2; f * f <= n; ++f)
        if (n % f == 0) return f;
    return n;
}

template <int N, bool INV>
__device__ __forceinline__ void dft_nat(f2* u) {
    if constexpr (N == 2) {
        const f2 a = u[0], b = u[1];
        u[0] = a + b;
        u[1] = a - b;
    } else if constexpr (N == 4) {
        const f2 a = u[0] + u[2], b = u[0] - u[2], c = u[1] + u[3], d = u[1] - u[3];
        const f2 jd = INV ? mul_pi(d) : mul_ni(d);
        u[0] = a + c;
        u[1] = b + jd;
        u[2] = a - c;
        u[3] = b - jd;
    } else if constexpr (N > 2 && spf(N) == N) {
        dft_odd_c<N, INV>(u);
    } else if constexpr (N > 4) {
        constexpr int N1 = (N % 4 == 0) ? 4 : spf(N), N2 = N / N1;       // n = N2 n1 + n2,  k = k1 + N1 k2
        f2 t[N];
        sfor<0, N2>([&](auto n2) {
            constexpr int Q = decltype(n2)::value;
            f2 w[N1];
            sfor<0, N1>([&](auto n1) { w[decltype(n1)::value] = u[N2 * decltype(n1)::value + Q]; });
            dft_nat<N1, INV>(w);
            sfor<0, N1>([&](auto k1) { t[decltype(k1)::value * N2 + Q] = mul_root<N, Q * decltype(k1)::value, INV>(w[decltype(k1)::value]); });
        });
        sfor<0, N1>([&](auto k1) {
            constexpr int K = decltype(k1)::value;
            dft_nat<N2, INV>(t + K * N2);
            sfor<0, N2>([&](auto k2) { u[K + N1 * decltype(k2)::value] = t[K * N2 + decltype(k2)::value]; });
        });
    }
}

}  // namespace fftr
