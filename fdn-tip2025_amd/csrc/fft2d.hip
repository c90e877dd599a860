// Full-image real 2-D FFT pipeline (norm='backward') for FCAFFN, the FDN guidance transforms and
// MAR (FDN_arch.py:90-98, :139-147, :411-418, :882-914): mixed-radix Stockham autosort FFTs staged
// in LDS, any length (720p needs radices 2/4/5/23, 1080p 3/5/17, fourier_fuse 41/641/...).
//
//   rows   : real row of length W -> half-length complex FFT (M = W/2) + split post-pass -> W/2+1 bins
//   columns: a workgroup owns TC adjacent columns x all H rows of one (b,c) plane in LDS, runs the
//            forward FFT, the pointwise spectral op and (FCAFFN) the inverse FFT without leaving
//            the CU: FFT <-> pointwise <-> iFFT fused in one launch
//   rows^-1: Hermitian merge pre-pass + half-length inverse FFT -> real row (+ residual epilogue)
//
// A Stockham pass of radix R maps N/R butterflies onto the threads; radix 2 and 4 are register
// butterflies, every other prime uses the gather form out[m] = sum_r in[i + r*N/R] * W_L^{r*m}.
// Twiddles come from one immutable table W_N^t per length, built on first use with exact
// values on the axes (so DC/Nyquist bins of real data stay exactly real).
#include <math.h>

#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "fft_regs.hpp"

namespace {

constexpr int MAX_STAGES = 16;
constexpr int NT = 256;

struct Plan {
    int N;                     // transform length
    int nst;
    int radix[MAX_STAGES];
    const float2* tw;          // W_tabN^t, t in [0, tabN)
    int tab_mul;               // tabN / N
};

// ------------------------------------------------------------------------------------------
// host: plan cache
// ------------------------------------------------------------------------------------------
std::mutex g_mu;
std::map<std::pair<int, int>, const float2*> g_tables;   // (device, N) -> device table

const float2* get_table(int N) {
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_tables.find({devid, N});
    if (it != g_tables.end()) return it->second;
    std::vector<float2> h(N);
    for (int t = 0; t < N; ++t) {
        double c, s;
        if ((4L * t) % N == 0) {                      // exact on the axes
            const int q = (int)((4L * t) / N);        // quarter turns
            c = (q == 0) ? 1.0 : (q == 2 ? -1.0 : 0.0);
            s = (q == 1) ? 1.0 : (q == 3 ? -1.0 : 0.0);
        } else {
            const double a = 2.0 * M_PI * (double)t / (double)N;
            c = cos(a);
            s = sin(a);
        }
        h[t] = make_float2((float)c, (float)(-s) + 0.0f);   // e^{-2 pi i t / N}; "+0" keeps zeros positive
        if (h[t].y == 0.0f) h[t].y = 0.0f;
        if (h[t].x == 0.0f) h[t].x = 0.0f;
    }
    float2* d = nullptr;
    if (hipMalloc(&d, sizeof(float2) * N) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), sizeof(float2) * N, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    g_tables[{devid, N}] = d;
    return d;
}

bool make_plan(int N, int tabN, Plan* p) {
    p->N = N;
    p->nst = 0;
    int n = N;
    auto push = [&](int r) { if (p->nst < MAX_STAGES) p->radix[p->nst++] = r; };
    int n2 = n, odd[MAX_STAGES], nodd = 0;
    while (n2 % 2 == 0) n2 /= 2;
    for (int f = 3; (long)f * f <= n2; f += 2)
        while (n2 % f == 0) { if (nodd < MAX_STAGES) odd[nodd++] = f; n2 /= f; }
    if (n2 > 1 && nodd < MAX_STAGES) odd[nodd++] = n2;
    for (int i = nodd - 1; i >= 0; --i) { push(odd[i]); n /= odd[i]; }     // largest odd prime first
    while (n % 4 == 0) { push(4); n /= 4; }
    while (n % 2 == 0) { push(2); n /= 2; }
    int prod = 1;
    for (int i = 0; i < p->nst; ++i) prod *= p->radix[i];
    if (prod != N || tabN % N != 0) return false;
    p->tw = get_table(tabN);
    p->tab_mul = tabN / N;
    return p->tw != nullptr;
}

// Rader's algorithm for a prime length p whose p-1 factors into the radices above (641 = W/2 of the
// (W+2)-wide maps of fourier_fuse at 720p, FDN_arch.py:126,139): with a primitive root g,
//   X[g^-q] = x[0] + sum_m x[g^m] * w^(g^(m-q)),  w = e^{-2 pi i / p}
// is a cyclic convolution of length p-1, done with two Stockham FFTs of length p-1 and a pointwise product
// with the precomputed spectrum of b[m] = w^(g^-m) (scaled by 1/(p-1)); X[0] = x[0] + sum_m x[g^m].
struct Rader {
    int p;                     // 0 = unused
    const int* perm_in;        // [p-1]  g^q mod p
    const int* perm_out;       // [p-1]  g^-q mod p
    const float2* bhat;        // [p-1]  FFT_{p-1}(b) / (p-1)
    Plan sub;                  // length p-1, its own table (tab_mul 1)
};
std::map<std::pair<int, int>, Rader> g_rader;

bool is_prime(int n) {
    if (n < 2) return false;
    for (int f = 2; (long)f * f <= n; ++f)
        if (n % f == 0) return false;
    return true;
}

// returns false when p is not prime / p-1 needs a radix without a register butterfly (caller keeps the gather pass)
bool get_rader(int p, Rader* out) {
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess) return false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_rader.find({devid, p});
        if (it != g_rader.end()) { *out = it->second; return out->p != 0; }
    }
    Rader r = {};
    auto fail = [&]() { std::lock_guard<std::mutex> lk(g_mu); g_rader[{devid, p}] = Rader{}; return false; };
    if (p < 29 || !is_prime(p)) return fail();
    const int n = p - 1;
    if (!make_plan(n, n, &r.sub)) return fail();
    for (int i = 0; i < r.sub.nst; ++i) {
        const int R = r.sub.radix[i];
        if (!(R == 2 || R == 3 || R == 4 || R == 5 || R == 7 || R == 17 || R == 23)) return fail();
    }
    auto powmod = [&](long b, long e) { long x = 1; b %= p; while (e) { if (e & 1) x = x * b % p; b = b * b % p; e >>= 1; } return x; };
    int g = 0;
    for (int c = 2; c < p && !g; ++c) {
        bool ok = true;
        int m = n;
        for (int f = 2; f <= m && ok; ++f)
            if (m % f == 0) { if (powmod(c, n / f) == 1) ok = false; while (m % f == 0) m /= f; }
        if (ok) g = c;
    }
    if (!g) return fail();
    const long ginv = powmod(g, p - 2);
    std::vector<int> pin(n), pout(n);
    long a = 1, b = 1;
    for (int q = 0; q < n; ++q) { pin[q] = (int)a; pout[q] = (int)b; a = a * g % p; b = b * ginv % p; }
    std::vector<double> br(n), bi(n);
    for (int m = 0; m < n; ++m) {                     // b[m] = w^(g^-m)
        const double ang = -2.0 * M_PI * (double)pout[m] / (double)p;
        br[m] = cos(ang); bi[m] = sin(ang);
    }
    std::vector<float2> bh(n);
    for (int k = 0; k < n; ++k) {                     // plain DFT in double: n^2 = 4e5 terms, once per (device, p)
        double sr = 0, si = 0;
        for (int m = 0; m < n; ++m) {
            const double ang = -2.0 * M_PI * (double)((long)k * m % n) / (double)n;
            const double c = cos(ang), sn = sin(ang);
            sr += br[m] * c - bi[m] * sn;
            si += br[m] * sn + bi[m] * c;
        }
        bh[k] = make_float2((float)(sr / n), (float)(si / n));
    }
    int *dpi = nullptr, *dpo = nullptr;
    float2* dbh = nullptr;
    if (hipMalloc(&dpi, sizeof(int) * n) != hipSuccess || hipMalloc(&dpo, sizeof(int) * n) != hipSuccess ||
        hipMalloc(&dbh, sizeof(float2) * n) != hipSuccess)
        return fail();
    if (hipMemcpy(dpi, pin.data(), sizeof(int) * n, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dpo, pout.data(), sizeof(int) * n, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dbh, bh.data(), sizeof(float2) * n, hipMemcpyHostToDevice) != hipSuccess)
        return fail();
    r.p = p; r.perm_in = dpi; r.perm_out = dpo; r.bhat = dbh;
    std::lock_guard<std::mutex> lk(g_mu);
    g_rader[{devid, p}] = r;
    *out = r;
    return true;
}

// ------------------------------------------------------------------------------------------
// device: Stockham passes over sequences held in LDS
//   element idx of sequence s lives at  s*ss + idx*es
// ------------------------------------------------------------------------------------------
// i / d for the index ranges of these kernels without the ~25-instruction integer division sequence: with r = 1.0f / d,
// floor((i + 0.5) * r) is exact while i < 2^22 and d < 2^12 (the rounding error of the product, ~1.2e-7 * i / d, stays below the
// 0.5 / d margin the half adds).  The passes used to spend most of their vector instructions on `job / nseq` and `i % Ns`.
__device__ __forceinline__ int fdiv(int i, float r) { return (int)(((float)i + 0.5f) * r); }

template <bool INV>
__device__ __forceinline__ float2 twd(const float2* __restrict__ tw, int idx) {
    const float2 w = tw[idx];
    return INV ? make_float2(w.x, -w.y) : w;
}

// DFT_R of an odd R in registers, using the conjugate symmetry of the roots: with a_r = u_r + u_{R-r},
// b_r = u_r - u_{R-r} (r = 1..h, h = (R-1)/2) and w = e^{-+2 pi i m/R},
//   X_o = P_o + Q_o,  X_{R-o} = P_o - Q_o,  P_o = u_0 + sum_r a_r Re(w_{ro}),  Q_o = i sum_r b_r Im(w_{ro})
// i.e. (R-1)^2 real FMAs instead of the 4 (R-1) R of the plain complex matrix product (3.4x fewer for R = 23),
// and only h roots live in registers.  `wR[m-1]` = w_m for m = 1..h.
template <int R>
__device__ __forceinline__ void dft_odd(float2 (&u)[R], const float2 (&wR)[(R - 1) / 2]) {
    constexpr int h = (R - 1) / 2;
    float2 a[h], b[h];
    float2 x0 = u[0];
#pragma unroll
    for (int r = 1; r <= h; ++r) {
        a[r - 1] = make_float2(u[r].x + u[R - r].x, u[r].y + u[R - r].y);
        b[r - 1] = make_float2(u[r].x - u[R - r].x, u[r].y - u[R - r].y);
        x0.x += a[r - 1].x;
        x0.y += a[r - 1].y;
    }
#pragma unroll
    for (int o = 1; o <= h; ++o) {
        float2 P = u[0], Q = make_float2(0.f, 0.f);
#pragma unroll
        for (int r = 1; r <= h; ++r) {
            const int m = (r * o) % R;
            const float wx = wR[(m <= h ? m : R - m) - 1].x;
            const float wy = (m <= h) ? wR[m - 1].y : -wR[R - m - 1].y;
            P.x = fmaf(a[r - 1].x, wx, P.x);
            P.y = fmaf(a[r - 1].y, wx, P.y);
            Q.x = fmaf(-b[r - 1].y, wy, Q.x);
            Q.y = fmaf(b[r - 1].x, wy, Q.y);
        }
        u[o] = make_float2(P.x + Q.x, P.y + Q.y);
        u[R - o] = make_float2(P.x - Q.x, P.y - Q.y);
    }
    u[0] = x0;
}

// odd-prime pass with the whole radix-R butterfly in registers
template <bool INV, int R>
__device__ void fft_pass_reg(const float2* src, float2* dst, int N, int Ns, int nseq, int ss, int es, bool seq_fast,
                             const float2* __restrict__ tw, int tab_mul) {
    const int T = N / R;
    const int L = Ns * R;
    const int tws = tab_mul * (N / L);
    const int twr = tab_mul * (N / R);
    float2 wR[(R - 1) / 2];
#pragma unroll
    for (int t = 1; t <= (R - 1) / 2; ++t) wR[t - 1] = twd<INV>(tw, t * twr);
    const int jobs = T * nseq;
    const float r_nseq = 1.0f / (float)nseq, r_T = 1.0f / (float)T, r_Ns = 1.0f / (float)Ns;
    for (int job = threadIdx.x; job < jobs; job += NT) {
        int i, s;
        if (seq_fast) { i = fdiv(job, r_nseq); s = job - i * nseq; }
        else { s = fdiv(job, r_T); i = job - s * T; }
        const int k = i - fdiv(i, r_Ns) * Ns;
        const int j = (i - k) * R + k;
        const float2* sp = src + s * ss;
        float2* dp = dst + s * ss;
        float2 u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = sp[(i + r * T) * es];
        if (k) {
#pragma unroll
            for (int r = 1; r < R; ++r) u[r] = cmul(u[r], twd<INV>(tw, r * k * tws));
        }
        dft_odd<R>(u, wR);
#pragma unroll
        for (int o = 0; o < R; ++o) dp[(j + o * Ns) * es] = u[o];
    }
}

// BIG: 0 = register butterflies for the radices 2-7 only, 1 = + 17 / 23 (1080p / 720p lengths), 2 = + 13 / 37 / 41 (the (H + 2)-row
// maps of fourier_fuse: 738 = 41 * 18, 370 = 37 * 10, 546 = 13 * 42) - tiers, because the widest butterfly sets the kernel's register count
template <bool INV, int BIG>
__device__ void fft_pass(const float2* src, float2* dst, int N, int Ns, int R, int nseq, int ss, int es, bool seq_fast,
                         const float2* __restrict__ tw, int tab_mul) {
    const int T = N / R;
    const int L = Ns * R;
    const int tws = tab_mul * (N / L);        // W_L^e = tw[e * tws]
    const int jobs = T * nseq;
    const float r_nseq = 1.0f / (float)nseq, r_T = 1.0f / (float)T, r_Ns = 1.0f / (float)Ns, r_jobs = 1.0f / (float)jobs;
    switch (R) {
        case 3: fft_pass_reg<INV, 3>(src, dst, N, Ns, nseq, ss, es, seq_fast, tw, tab_mul); return;
        case 5: fft_pass_reg<INV, 5>(src, dst, N, Ns, nseq, ss, es, seq_fast, tw, tab_mul); return;
        case 7: fft_pass_reg<INV, 7>(src, dst, N, Ns, nseq, ss, es, seq_fast, tw, tab_mul); return;
        case 13: if (BIG == 2) { fft_pass_reg<INV, 13>(src, dst, N, Ns, nseq, ss, es, seq_fast, tw, tab_mul); return; } break;   // 546 = 13 * 42 (1080p)
        case 17: if (BIG) { fft_pass_reg<INV, 17>(src, dst, N, Ns, nseq, ss, es, seq_fast, tw, tab_mul); return; } break;
        case 23: if (BIG) { fft_pass_reg<INV, 23>(src, dst, N, Ns, nseq, ss, es, seq_fast, tw, tab_mul); return; } break;
        case 37: if (BIG == 2) { fft_pass_reg<INV, 37>(src, dst, N, Ns, nseq, ss, es, seq_fast, tw, tab_mul); return; } break;
        case 41: if (BIG == 2) { fft_pass_reg<INV, 41>(src, dst, N, Ns, nseq, ss, es, seq_fast, tw, tab_mul); return; } break;
        default: break;
    }
    if (R != 2 && R != 4) {
        // gather form, one output element per job: out[m] = sum_r in[i + r*T] * W_L^{r*m}
        const int jobs2 = jobs * R;
        for (int job2 = threadIdx.x; job2 < jobs2; job2 += NT) {
            const int o = fdiv(job2, r_jobs), job = job2 - o * jobs;
            int i, s;
            if (seq_fast) { i = fdiv(job, r_nseq); s = job - i * nseq; }
            else { s = fdiv(job, r_T); i = job - s * T; }
            const int k = i - fdiv(i, r_Ns) * Ns;
            const int j = (i - k) * R + k;
            const float2* sp = src + s * ss;
            const int m = k + o * Ns;              // output position inside the length-L block
            float2 acc = sp[i * es];
            int e = 0;
            for (int r = 1; r < R; ++r) {
                e += m;
                if (e >= L) e -= L;
                const float2 w = twd<INV>(tw, e * tws);
                const float2 v = sp[(i + r * T) * es];
                acc.x = fmaf(v.x, w.x, fmaf(-v.y, w.y, acc.x));
                acc.y = fmaf(v.x, w.y, fmaf(v.y, w.x, acc.y));
            }
            dst[s * ss + (j + o * Ns) * es] = acc;
        }
        return;
    }
    for (int job = threadIdx.x; job < jobs; job += NT) {
        int i, s;
        if (seq_fast) { i = fdiv(job, r_nseq); s = job - i * nseq; }
        else { s = fdiv(job, r_T); i = job - s * T; }
        const int k = i - fdiv(i, r_Ns) * Ns;
        const int j = (i - k) * R + k;
        const float2* sp = src + s * ss;
        float2* dp = dst + s * ss;
        if (R == 4) {
            float2 u0 = sp[i * es];
            float2 u1 = sp[(i + T) * es], u2 = sp[(i + 2 * T) * es], u3 = sp[(i + 3 * T) * es];
            if (k) {
                u1 = cmul(u1, twd<INV>(tw, k * tws));
                u2 = cmul(u2, twd<INV>(tw, 2 * k * tws));
                u3 = cmul(u3, twd<INV>(tw, 3 * k * tws));
            }
            const float2 a = make_float2(u0.x + u2.x, u0.y + u2.y), b = make_float2(u0.x - u2.x, u0.y - u2.y);
            const float2 c = make_float2(u1.x + u3.x, u1.y + u3.y), d = make_float2(u1.x - u3.x, u1.y - u3.y);
            // forward: -i*d = (d.y, -d.x) ; inverse: +i*d = (-d.y, d.x)
            const float2 jd = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
            dp[j * es] = make_float2(a.x + c.x, a.y + c.y);
            dp[(j + Ns) * es] = make_float2(b.x + jd.x, b.y + jd.y);
            dp[(j + 2 * Ns) * es] = make_float2(a.x - c.x, a.y - c.y);
            dp[(j + 3 * Ns) * es] = make_float2(b.x - jd.x, b.y - jd.y);
        } else {
            const float2 u0 = sp[i * es];
            float2 u1 = sp[(i + T) * es];
            if (k) u1 = cmul(u1, twd<INV>(tw, k * tws));
            dp[j * es] = make_float2(u0.x + u1.x, u0.y + u1.y);
            dp[(j + Ns) * es] = make_float2(u0.x - u1.x, u0.y - u1.y);
        }
    }
}

// run all passes; returns the buffer holding the result
template <bool INV, int BIG>
__device__ float2* fft_run(float2* a, float2* b, const Plan& p, const float2* tw, int nseq, int ss, int es, bool seq_fast) {
    int Ns = 1;
    float2* src = a;
    float2* dst = b;
    for (int st = 0; st < p.nst; ++st) {
        __syncthreads();
        fft_pass<INV, BIG>(src, dst, p.N, Ns, p.radix[st], nseq, ss, es, seq_fast, tw, p.tab_mul);
        Ns *= p.radix[st];
        float2* t = src; src = dst; dst = t;
    }
    __syncthreads();
    return src;
}

// ------------------------------------------------------------------------------------------
// In-place passes for the column kernels: every thread pulls the inputs of all its butterflies into
// registers, the workgroup synchronises, then the outputs go back into the SAME LDS buffer at their
// Stockham positions.  Halves the LDS footprint (no ping-pong buffer) so 2-3 workgroups share a CU.
// Needs (N/R)*nseq <= NT*JMAX for every pass (checked on the host).
// ------------------------------------------------------------------------------------------
constexpr int EMAX = 36;                   // complex values a thread may hold across the barrier

template <bool INV, int R>
__device__ void fft_pass_inplace(float2* buf, int N, int Ns, int nseq, const float2* __restrict__ tw) {
    constexpr int JMAX = EMAX / R;
    const int T = N / R;
    const int L = Ns * R;
    const int tws = N / L, twr = N / R;
    const int jobs = T * nseq;
    const float r_nseq = 1.0f / (float)nseq, r_Ns = 1.0f / (float)Ns;
    float2 u[JMAX][R];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int job = threadIdx.x + NT * j;
        if (job < jobs) {
            const int i = fdiv(job, r_nseq), s = job - i * nseq;
            const int k = i - fdiv(i, r_Ns) * Ns;
#pragma unroll
            for (int r = 0; r < R; ++r) u[j][r] = buf[s + (i + r * T) * nseq];
            if (k) {
#pragma unroll
                for (int r = 1; r < R; ++r) u[j][r] = cmul(u[j][r], twd<INV>(tw, r * k * tws));
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int job = threadIdx.x + NT * j;
        if (job < jobs) {
            const int i = fdiv(job, r_nseq), s = job - i * nseq;
            const int k = i - fdiv(i, r_Ns) * Ns;
            float2* dp = buf + s + ((i - k) * R + k) * nseq;
            if (R == 2) {
                dp[0] = make_float2(u[j][0].x + u[j][1].x, u[j][0].y + u[j][1].y);
                dp[Ns * nseq] = make_float2(u[j][0].x - u[j][1].x, u[j][0].y - u[j][1].y);
            } else if (R == 4) {
                const float2 a = make_float2(u[j][0].x + u[j][2].x, u[j][0].y + u[j][2].y);
                const float2 b = make_float2(u[j][0].x - u[j][2].x, u[j][0].y - u[j][2].y);
                const float2 c = make_float2(u[j][1].x + u[j][3].x, u[j][1].y + u[j][3].y);
                const float2 d = make_float2(u[j][1].x - u[j][3].x, u[j][1].y - u[j][3].y);
                const float2 jd = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
                dp[0] = make_float2(a.x + c.x, a.y + c.y);
                dp[Ns * nseq] = make_float2(b.x + jd.x, b.y + jd.y);
                dp[2 * Ns * nseq] = make_float2(a.x - c.x, a.y - c.y);
                dp[3 * Ns * nseq] = make_float2(b.x - jd.x, b.y - jd.y);
            } else if constexpr (R % 2 == 1) {
                float2 wR[(R - 1) / 2];                                           // wave-uniform: LDS broadcast
#pragma unroll
                for (int t = 1; t <= (R - 1) / 2; ++t) wR[t - 1] = twd<INV>(tw, t * twr);
                dft_odd<R>(u[j], wR);
#pragma unroll
                for (int o = 0; o < R; ++o) dp[o * Ns * nseq] = u[j][o];
            }
        }
    }
    __syncthreads();
}

template <bool INV, int BIG>
__device__ void fft_run_inplace(float2* buf, const Plan& p, const float2* tw, int nseq) {
    int Ns = 1;
    __syncthreads();
    for (int st = 0; st < p.nst; ++st) {
        const int R = p.radix[st];
        switch (R) {
            case 2: fft_pass_inplace<INV, 2>(buf, p.N, Ns, nseq, tw); break;
            case 3: fft_pass_inplace<INV, 3>(buf, p.N, Ns, nseq, tw); break;
            case 4: fft_pass_inplace<INV, 4>(buf, p.N, Ns, nseq, tw); break;
            case 5: fft_pass_inplace<INV, 5>(buf, p.N, Ns, nseq, tw); break;
            case 7: fft_pass_inplace<INV, 7>(buf, p.N, Ns, nseq, tw); break;
            case 17: if (BIG) fft_pass_inplace<INV, 17>(buf, p.N, Ns, nseq, tw); break;
            case 23: if (BIG) fft_pass_inplace<INV, 23>(buf, p.N, Ns, nseq, tw); break;
            default: break;
        }
        Ns *= R;
    }
}

// ------------------------------------------------------------------------------------------
// rows: r2c
// ------------------------------------------------------------------------------------------
template <bool BIG>
__global__ __launch_bounds__(NT) void rfft_rows_kernel(const float* __restrict__ in, float2* __restrict__ out, int W, long R,
                                                       int rpb, const Plan p, const Rader rd, int pitch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = W / 2, Wf = M + 1;
    float2* A = reinterpret_cast<float2*>(smem);
    float2* Bf = A + (long)rpb * M;
    float2* twl = Bf + (long)rpb * M;
    float2* tws = twl + W;                              // Rader: table of the length M-1 sub-transform, then x[0] per row
    for (int i = threadIdx.x; i < W; i += NT) twl[i] = p.tw[i];
    if (rd.p)
        for (int i = threadIdx.x; i < M - 1; i += NT) tws[i] = rd.sub.tw[i];
    const long row0 = (long)blockIdx.x * rpb;
    const int nrow = (int)min((long)rpb, R - row0);
    {   // batches of unconditional loads (clamped address, value masked afterwards): with a branch per element hipcc
        // waits for every load separately - one memory round trip per element
        constexpr int U = 8;
        const float2* in2 = reinterpret_cast<const float2*>(in + row0 * W);      // row s, element m at in2[s * M + m]
        const int tot = rpb * M, live = nrow * M;
        if (M % 2 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {        // 16-byte lanes: two complex inputs per load
            const float4* in4 = reinterpret_cast<const float4*>(in2);
            float4* A4 = reinterpret_cast<float4*>(A);
            const int tot4 = tot / 2, live4 = live / 2;
            for (int base = threadIdx.x; base < tot4; base += NT * 4) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + NT * u;
                    v[u] = in4[idx < live4 ? idx : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + NT * u;
                    if (idx < tot4) A4[idx] = idx < live4 ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        } else
        for (int base = threadIdx.x; base < tot; base += NT * U) {
            float2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + NT * u;
                v[u] = in2[idx < live ? idx : 0];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + NT * u;
                if (idx < tot) A[idx] = idx < live ? v[u] : make_float2(0.f, 0.f);
            }
        }
    }
    float2* Z;
    if (rd.p) {
        const int n = M - 1;
        float2* z0 = tws + n;
        __syncthreads();
        for (int idx = threadIdx.x; idx < rpb * n; idx += NT) {           // a[q] = z[g^q]
            const int s = idx / n, q = idx - s * n;
            Bf[s * M + q] = A[s * M + rd.perm_in[q]];
        }
        if (threadIdx.x < rpb) z0[threadIdx.x] = A[threadIdx.x * M];
        float2* Y = fft_run<false, BIG>(Bf, A, rd.sub, tws, rpb, M, 1, false);
        float2* other = (Y == Bf) ? A : Bf;
        __syncthreads();
        for (int idx = threadIdx.x; idx < rpb * n; idx += NT) {           // spectrum of the convolution
            const int s = idx / n, k = idx - s * n;
            const float2 y = Y[s * M + k];
            if (k == 0) other[s * M + n] = make_float2(z0[s].x + y.x, z0[s].y + y.y);   // X[0] = x[0] + sum a, parked in the spare slot
            Y[s * M + k] = cmul(y, rd.bhat[k]);
        }
        __syncthreads();
        float2 x0 = make_float2(0.f, 0.f);                                 // X[0] of row `threadIdx.x`, kept across the inverse
        const bool keeper = threadIdx.x < rpb;
        if (keeper) x0 = other[threadIdx.x * M + n];
        float2* Cv = fft_run<true, BIG>(Y, other, rd.sub, tws, rpb, M, 1, false);
        float2* dst = (Cv == Y) ? other : Y;
        __syncthreads();
        for (int idx = threadIdx.x; idx < rpb * n; idx += NT) {           // X[g^-q] = x[0] + c[q]
            const int s = idx / n, q = idx - s * n;
            const float2 c = Cv[s * M + q];
            dst[s * M + rd.perm_out[q]] = make_float2(z0[s].x + c.x, z0[s].y + c.y);
        }
        if (keeper) dst[threadIdx.x * M] = x0;
        __syncthreads();
        Z = dst;
    } else {
        Z = fft_run<false, BIG>(A, Bf, p, twl, rpb, M, 1, false);
    }
    // split: X[k] = E[k] + W_N^k O[k],  E = (Z[k]+conj Z[M-k])/2,  O = -i (Z[k]-conj Z[M-k])/2
    const int tw1 = p.tab_mul / 2;              // table is W_W^t:  tab_mul = W / M = 2  -> stride 1
    const float r_Wf = 1.0f / (float)Wf;
#pragma unroll 4
    for (int idx = threadIdx.x; idx < nrow * Wf; idx += NT) {
        const int s = fdiv(idx, r_Wf), k = idx - s * Wf;
        const float2 zk = Z[s * M + (k == M ? 0 : k)];
        const float2 zc = Z[s * M + (k == 0 ? 0 : M - k)];
        const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
        const float2 d = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y));   // (Z - conj Zc)/2
        const float2 o = make_float2(d.y, -d.x);                                     // -i * d
        float2 x;
        if (k == 0) x = make_float2(e.x + o.x, 0.0f);
        else if (k == M) x = make_float2(e.x - o.x, 0.0f);
        else {
            const float2 w = twl[k * tw1];
            x = make_float2(e.x + (o.x * w.x - o.y * w.y), e.y + (o.x * w.y + o.y * w.x));
        }
        out[(row0 + s) * pitch + k] = x;
    }
    for (int idx = threadIdx.x; idx < nrow * (pitch - Wf); idx += NT) {         // bins of a padded row past W/2: zeros
        const int s = idx / (pitch - Wf), k = Wf + idx - s * (pitch - Wf);
        out[(row0 + s) * pitch + k] = make_float2(0.f, 0.f);
    }
}

// rows: c2r.  in rows have stride in_ws bins (>= M+1: leading-slice crop of a wider spectrum)
template <bool BIG>
__global__ __launch_bounds__(NT) void irfft_rows_kernel(const float2* __restrict__ in, long in_ws, long in_plane_rows,
                                                        long in_plane_stride, float* __restrict__ out, int W, int H, long R,
                                                        int rpb, float scale, const float* __restrict__ res, float alpha,
                                                        const Plan p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = W / 2;
    float2* A = reinterpret_cast<float2*>(smem);
    float2* Bf = A + (long)rpb * M;
    float2* twl = Bf + (long)rpb * M;
    for (int i = threadIdx.x; i < W; i += NT) twl[i] = p.tw[i];
    __syncthreads();
    const long row0 = (long)blockIdx.x * rpb;
    const int nrow = (int)min((long)rpb, R - row0);
    const int tw1 = p.tab_mul / 2;
    constexpr int UL = 4;
    const float r_M = 1.0f / (float)M, r_H = 1.0f / (float)H;
    for (int base = threadIdx.x; base < rpb * M; base += NT * UL) {
      float2 xks[UL], xcs[UL];
#pragma unroll
      for (int u = 0; u < UL; ++u) {                               // batched, unconditional loads (clamped row)
        const int idx = base + NT * u;
        const int s0 = fdiv(idx, r_M), k = idx - s0 * M;
        const int s = min(s0, nrow - 1);
        const long row = row0 + s;
        const long plane = row < (1L << 22) ? (long)fdiv((int)row, r_H) : row / H, h = row - plane * H;
        const float2* src = in + plane * in_plane_stride + h * in_ws;
        xks[u] = src[k];
        xcs[u] = src[M - k];
      }
#pragma unroll
      for (int u = 0; u < UL; ++u) {
        const int idx = base + NT * u;
        if (idx >= rpb * M) continue;
        const int s = fdiv(idx, r_M), k = idx - s * M;
        float2 z = make_float2(0.f, 0.f);
        if (s < nrow) {
            float2 xk = xks[u], xc = xcs[u];
            if (k == 0) { xk.y = 0.f; xc.y = 0.f; }            // c2r ignores Im of DC and Nyquist
            // E = (X[k] + conj X[M-k])/2 ; O = (X[k] - conj X[M-k])/2 * W_N^{-k} ; Z = E + i O
            const float2 e = make_float2(0.5f * (xk.x + xc.x), 0.5f * (xk.y - xc.y));
            const float2 d = make_float2(0.5f * (xk.x - xc.x), 0.5f * (xk.y + xc.y));
            const float2 w = twl[k * tw1];                      // conj -> W^{-k}
            const float2 o = make_float2(d.x * w.x + d.y * w.y, d.y * w.x - d.x * w.y);
            z = make_float2(e.x - o.y, e.y + o.x);
        }
        A[idx] = z;
      }
    }
    float2* Z = fft_run<true, BIG>(A, Bf, p, twl, rpb, M, 1, false);
    {   // rows row0.. are contiguous in out / res: element idx of the block sits at float2 index row0 * M + idx
        constexpr int US = 8;
        const float2* res2 = res ? reinterpret_cast<const float2*>(res + row0 * W) : nullptr;
        float2* out2 = reinterpret_cast<float2*>(out + row0 * W);
        const int live = nrow * M;
        if (M % 2 == 0 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res)) & 15) == 0) {   // 16-byte lanes
            const float4* res4 = reinterpret_cast<const float4*>(res2);
            float4* out4 = reinterpret_cast<float4*>(out2);
            const float4* Z4 = reinterpret_cast<const float4*>(Z);
            const int live4 = live / 2;
            for (int base = threadIdx.x; base < live4; base += NT * 4) {
                float4 r[4];
                if (res4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) r[u] = res4[min(base + NT * u, live4 - 1)];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + NT * u;
                    if (idx >= live4) continue;
                    float4 v = Z4[idx];
                    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
                    if (res4) {
                        v.x = fmaf(alpha, r[u].x, v.x); v.y = fmaf(alpha, r[u].y, v.y);
                        v.z = fmaf(alpha, r[u].z, v.z); v.w = fmaf(alpha, r[u].w, v.w);
                    }
                    out4[idx] = v;
                }
            }
        } else
        for (int base = threadIdx.x; base < live; base += NT * US) {
            float2 r[US];
            if (res2) {
#pragma unroll
                for (int u = 0; u < US; ++u) r[u] = res2[min(base + NT * u, live - 1)];      // one batch, then the stores
            }
#pragma unroll
            for (int u = 0; u < US; ++u) {
                const int idx = base + NT * u;
                if (idx >= live) continue;
                float2 v = Z[idx];
                v.x *= scale;
                v.y *= scale;
                if (res2) {
                    v.x = fmaf(alpha, r[u].x, v.x);
                    v.y = fmaf(alpha, r[u].y, v.y);
                }
                out2[idx] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// columns
// ------------------------------------------------------------------------------------------
struct ColArgs {
    // spectrum planes: [planes][H][Wf] complex
    float2* z;                 // in/out (FCAFFN), in (fwd), out (inv polar)
    int H, Wf, C;              // C = channels per batch item (plane = b*C + c)
    int tc, tcs;               // columns per workgroup (a power of two) and its log2
    long planes;               // total planes (a multiple of C)
    // FCAFFN modulation (FDN_arch.py:412-417)
    const float4* guide;       // [B][H][Wf][2] float4: (amp0, amp1, amp2, pha0), (pha1, pha2, -, -)  (fdn_pack_guidance)
    const float* wxa;          // [C][3]
    const float* wxp;          // [C][3]
    // forward outputs (real planes [planes][H][Wf])
    float* out_abs;
    float* out_ang;
    int rd_before;             // replace_denormals before abs/angle
    int fix_real;              // force Im = +0 at the self-conjugate bins of a real transform
    // inverse-from-polar inputs: planes [planes][Hin][Wfin], leading (H, Wf) slice used
    const float* in_mag;
    const float* in_pha;
    int Hin, Wfin;
};

enum { COL_FCAFFN = 0, COL_FWD = 1, COL_INV_POLAR = 2 };

template <int MODE, int BIG, bool INPL>
__global__ __launch_bounds__(NT, (INPL ? 2 : 1)) void fft_cols_kernel(ColArgs a, const Plan p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = a.H, Wf = a.Wf, tc = a.tc;
    float2* A = reinterpret_cast<float2*>(smem);
    float2* Bf = A + (long)H * tc;                       // unused when INPL
    float2* twl = INPL ? Bf : Bf + (long)H * tc;
    for (int i = threadIdx.x; i < H; i += NT) twl[i] = p.tw[i];
    // XCD-aware work order: workgroups i and i+8 share an XCD (round-robin dispatch), so each XCD walks a
    // contiguous range of work items ordered (batch, column tile, channel): consecutive workgroups of an XCD
    // read the same guidance records (shared by all C channels of a batch item) out of that XCD's L2
    int plane, col0;
    {
        const int ntile = (Wf + tc - 1) / tc;
        const long total = (long)ntile * a.planes, per_xcd = (total + 7) / 8;
        const long w = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (w >= total) return;                                              // uniform: before any barrier
        const int C = a.C;
        const long grp = w / C;
        const int ch = (int)(w - grp * C);
        plane = (int)(grp / ntile) * C + ch;
        col0 = (int)(grp % ntile) * tc;
    }
    const int ncol = min(tc, Wf - col0);
    float2* zp = a.z + (long)plane * H * Wf;

    // ---- load: batches of independent requests (a plain loop would pay one memory round trip per element) ----
    const int per = H * tc, tcs = a.tcs, tcm = tc - 1;                 // tc is a power of two
    if (MODE == COL_INV_POLAR) {
        constexpr int U = 4;
        for (int base = threadIdx.x; base < per; base += NT * U) {
            float mg[U], ph[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + NT * u, h = idx >> tcs, c = idx & tcm;
                mg[u] = 0.f; ph[u] = 0.f;
                if (idx < per && c < ncol) {
                    const long o = ((long)plane * a.Hin + h) * a.Wfin + col0 + c;
                    mg[u] = a.in_mag[o];
                    ph[u] = a.in_pha[o];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + NT * u;
                if (idx < per) {
                    float sn, cs;
                    fdn_sincos(ph[u], &sn, &cs);
                    A[idx] = make_float2(mg[u] * cs, mg[u] * sn);                  // FDN_arch.py:95-97
                }
            }
        }
    } else {
        constexpr int U = 12;        // two round trips for 24 elements per thread (H * tc = 5888)
        for (int base = threadIdx.x; base < per; base += NT * U) {
            float2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + NT * u, h = idx >> tcs, c = idx & tcm;
                v[u] = make_float2(0.f, 0.f);
                if (idx < per && c < ncol) v[u] = zp[(long)h * Wf + col0 + c];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + NT * u;
                if (idx < per) A[idx] = v[u];
            }
        }
    }
    float2* Z = A;
    if (MODE != COL_INV_POLAR) {
        if (INPL) fft_run_inplace<false, BIG>(A, p, twl, tc);
        else Z = fft_run<false, BIG>(A, Bf, p, twl, tc, 1, tc, true);
    }
    float2* other = (Z == A) ? Bf : A;

    if (MODE == COL_FWD) {
        const bool evenH = (H % 2) == 0;
#pragma unroll 2
        for (int idx = threadIdx.x; idx < per; idx += NT) {
            const int h = idx >> tcs, c = idx & tcm;
            if (c >= ncol) continue;
            float2 v = Z[idx];
            const int col = col0 + c;
            if (a.fix_real && (col == 0 || col == Wf - 1) && (h == 0 || (evenH && h == H / 2))) v.y = 0.0f;
            if (a.rd_before) v = make_float2(rd1(v.x), rd1(v.y));
            const long o = ((long)plane * H + h) * Wf + col;
            if (a.out_abs) a.out_abs[o] = cabs2(v);
            if (a.out_ang) a.out_ang[o] = atan2f(v.y, v.x);
        }
        return;
    }

    if (MODE == COL_FCAFFN) {
        const int b = plane / a.C, ch = plane - b * a.C;
        const float wa0 = a.wxa[ch * 3], wa1 = a.wxa[ch * 3 + 1], wa2 = a.wxa[ch * 3 + 2];
        const float wp0 = a.wxp[ch * 3], wp1 = a.wxp[ch * 3 + 1], wp2 = a.wxp[ch * 3 + 2];
        const float4* gb = a.guide + (long)b * H * Wf * 2;
        constexpr int U = 12;       // registers are free here (the FFT passes are separate calls)
        for (int base = threadIdx.x; base < per; base += NT * U) {
            float4 g0[U], g1[U];                                   // one 32-byte record per bin: rows of a tile are contiguous
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + NT * u, h = idx >> tcs, c = idx & tcm;
                const bool ok = idx < per && c < ncol;
                const long o = ok ? (long)h * Wf + col0 + c : 0;
                g0[u] = gb[2 * o];
                g1[u] = gb[2 * o + 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + NT * u, c = idx & tcm;
                if (idx < per && c < ncol) {
                    const float A_ = wa0 * g0[u].x + wa1 * g0[u].y + wa2 * g0[u].z;       // conv1_xa(x_high)
                    const float ph = wp0 * g0[u].w + wp1 * g1[u].x + wp2 * g1[u].y;       // conv1_xp(xp2)
                    float sn, cs;
                    fdn_sincos(ph, &sn, &cs);
                    const float2 v = Z[idx];
                    const float2 r = make_float2(rd1(v.x), rd1(v.y));                       // :412
                    Z[idx] = cmul(r, make_float2(A_ * cs, -A_ * sn));   // |z| A e^{i(ang z - ph)}  :413-417
                }
            }
        }
    }
    float2* Y = Z;
    if (INPL) fft_run_inplace<true, BIG>(Z, p, twl, tc);
    else Y = fft_run<true, BIG>(Z, other, p, twl, tc, 1, tc, true);
#pragma unroll 8
    for (int idx = threadIdx.x; idx < per; idx += NT) {
        const int h = idx >> tcs, c = idx & tcm;
        if (c < ncol) zp[(long)h * Wf + col0 + c] = Y[idx];
    }
}

// ------------------------------------------------------------------------------------------
// Columns of length H = R * P (R an odd prime, P = 8 / 16 / 32) with a compile-time plan: 736 / 368 / 184 rows of the 720p
// pyramid (R = 23), 544 / 272 / 136 of 1080p (R = 17).  Cooley-Tukey with n = P n1 + n2, k = k1 + R k2:
//   1  thread (n2, column) loads x[P n1 + n2], n1 < R, straight into registers and runs the R-point DFT over n1
//   2  -> LDS  Y[k1][n2][column]                                                        (barrier)
//   3  thread (k1, column group) takes the P values of its 32 / P columns, multiplies by W_H^{n2 k1} and runs the P-point
//      FFT over n2 in registers (decimation in frequency: bit-reversed k2) -> the bins h = k1 + R k2 of those columns
//   4  the pointwise spectral operation on those registers (FCAFFN modulation / abs + angle out / polar in)
//   5  inverse P-point FFT over k2 (decimation in time: takes the bit-reversed order as it is), * W_H^{-n2 k1}
//   6  -> LDS, same cells                                                               (barrier)
//   7  thread (n2, column) runs the inverse R-point DFT over k1 and stores x[P n1 + n2]
// Two barriers and two LDS round trips per element (the generic in-place passes: 8 barriers-pairs and 18 LDS accesses for
// 736 = 23*4*4*2), no index arithmetic in the passes, all roots of unity inside the transforms are instruction constants.
// A workgroup owns 256 / P adjacent columns, so every step-1 wave is full; step 3 has R * 8 jobs (184 of 256 threads for R = 23):
// the idle wave rotates with the workgroup index.
// ------------------------------------------------------------------------------------------
using fftr::f2;
using fftr::sfor;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t cols_rsrc(const void* base, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes > 0x7FFFFFFFL ? 0x7FFFFFFFL : bytes), 0x00020000);
}
__device__ __forceinline__ f2 bload_f2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const fdn_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return f2{__uint_as_float(u.x), __uint_as_float(u.y)};
}
__device__ __forceinline__ void bstore_f2(f2 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b64(fdn_u32x2{__float_as_uint(v.x), __float_as_uint(v.y)}, r, voff, soff, 0);
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024 && !fdn_allow_dynamic_lds(reinterpret_cast<const void*>(kernel), bytes)) return FDN_ERR_LAUNCH;
    return FDN_OK;
}

// R may be composite (34 = 2 * 17 for the 1088 rows of 1080p level 1: dft_nat<R>); R * 8 jobs then exceed 256 threads and the
// workgroup gets a fifth wave that only takes part in the P-point stage.
template <int R, int P> struct ColPlan {
    static constexpr int NJ = R * 8, NT = NJ > 256 ? 320 : 256, KS = 256 + 4;
    static constexpr size_t lds = ((size_t)R * KS + (size_t)(P - 1) * R) * sizeof(float2);
};

#ifndef FDN_COLS_NB3
#define FDN_COLS_NB3 16
#endif
#ifndef FDN_COLS_WGS3
#define FDN_COLS_WGS3 1
#endif
template <int R, int P, int MODE>
// (three workgroups per CU where the LDS allows it, R <= 23.  Round 3 ran the FCAFFN mode at two: inlined 32 times, its full-range sincos
//  redo took 256 registers, and capped at 168 the hot path spilled - 0.99 against 0.78 ms.  Round 4: the redo is a rolled loop over the
//  thread's LDS cells and the guidance records come in batches of 2 instead of 8: 168 registers, no scratch, 9.34 -> 8.22 ms per step)
__global__ __launch_bounds__((ColPlan<R, P>::NT), (FDN_COLS_WGS3 && 3 * ColPlan<R, P>::lds <= 160 * 1024) ? 3 : 2) void fft_cols_rp_kernel(ColArgs a, const float2* __restrict__ twT) {
    constexpr int H = R * P, TC = 256 / P, CJ = 32 / P, NG = 8, KS = ColPlan<R, P>::KS, NJ = R * NG, NT = ColPlan<R, P>::NT;
    static_assert(TC / CJ == NG, "8 column groups");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cols[];
    f2* Y = reinterpret_cast<f2*>(smem_cols);       // [R][KS]
    f2* twl = Y + R * KS;                           // twl[(n2 - 1) * R + k1] = W_H^{n2 k1}, n2 >= 1  (R = 23: 53.5 KB)
    const int tid = threadIdx.x, Wf = a.Wf;
    const bool io = NT == 256 || tid < 256;         // threads with a (row class, column) of the tile
    for (int i = tid; i < (P - 1) * R; i += NT) twl[i] = f2{twT[R + i].x, twT[R + i].y};
    // work order: every XCD walks a contiguous run of items ordered (batch, chunk of 8 channels, column tile, channel in chunk):
    // the tiles either side of a shared 128-byte line run within a few workgroups of each other on one L2, and the guidance
    // records of a tile (shared by all channels of a batch item) are re-read from memory once per chunk only
    int plane, col0;
    {
        const int ntile = (Wf + TC - 1) / TC;
        const long total = (long)ntile * a.planes, per_xcd = (total + 7) / 8;
        const long w = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (w >= total) return;                                              // uniform: before any barrier
        const int C = a.C, CH = (C % 8 == 0) ? 8 : 1;
        const long grp = w / CH;
        const int ci = (int)(w - grp * CH);
        const long rest = grp / ntile;
        col0 = (int)(grp - rest * ntile) * TC;
        const int nchunk = C / CH;
        const long b = rest / nchunk;
        plane = (int)(b * C + (rest - b * nchunk) * CH + ci);
    }
    const int c = tid & (TC - 1), n2 = (tid & 255) / TC;
    const bool live = col0 + c < Wf;
    const int col = live ? col0 + c : Wf - 1;                                // dead lanes shadow the last column (columns never mix)
    const long plane_bins = (long)H * Wf;                                     // < 2^28: byte offsets inside a plane fit 32 bits
    const __amdgpu_buffer_rsrc_t rz = cols_rsrc(a.z + (long)plane * plane_bins, plane_bins * 8);
    const unsigned zoff = (unsigned)(n2 * Wf + col) * 8u, zstep = (unsigned)(P * Wf) * 8u;

    if (MODE != COL_INV_POLAR && io) {
        f2 u[R];
        sfor<0, R>([&](auto n1) { u[decltype(n1)::value] = bload_f2(rz, zoff, (unsigned)decltype(n1)::value * zstep); });
        fftr::dft_nat<R, false>(u);
        sfor<0, R>([&](auto k1) { Y[decltype(k1)::value * KS + tid] = u[decltype(k1)::value]; });
    }

    int jt = tid + 64 * (int)((blockIdx.x >> 3) % (NT / 64));
    if (jt >= NT) jt -= NT;
    const bool worker = jt < NJ;
    const int k1 = worker ? jt >> 3 : 0, cg = jt & 7;
    // column of value slot q (q / P = column inside the group): clamped like `col`
    unsigned binq[CJ];                              // k1 * Wf + column: bin offset of slot (cc, k2 = 0) inside a plane
    bool liveq[CJ];
#pragma unroll
    for (int cc = 0; cc < CJ; ++cc) {
        liveq[cc] = col0 + cc * NG + cg < Wf;                      // column cc of group cg = cc * 8 + cg: lanes of neighbouring groups hold
        binq[cc] = (unsigned)(k1 * Wf + (liveq[cc] ? col0 + cc * NG + cg : Wf - 1));      // neighbouring columns (their 32-byte guidance records share lines)
    }
    const unsigned kstep = (unsigned)(R * Wf);      // bins between k2 and k2 + 1
    // guidance records in batches, one batch in flight ahead of the arithmetic: 8 per batch (96 registers for the two buffers) with two
    // workgroups per CU, 4 per batch (48) where the LDS leaves room for a third workgroup
    constexpr int NB = (FDN_COLS_WGS3 && 3 * ColPlan<R, P>::lds <= 160 * 1024) ? FDN_COLS_NB3 : 4, BS = 32 / NB;
    fdn_u32x4 g0[2][BS];
    fdn_u32x2 g1[2][BS];
    __amdgpu_buffer_rsrc_t rg = rz;
    float wa0 = 0, wa1 = 0, wa2 = 0, wp0 = 0, wp1 = 0, wp2 = 0;
    auto gload = [&](auto bb) __attribute__((always_inline)) {
        constexpr int Bb = decltype(bb)::value;
        sfor<0, BS>([&](auto i) {
            constexpr int q = Bb * BS + decltype(i)::value, cc = q / P, k2 = fftr::brev(q % P, P);
            g0[Bb & 1][decltype(i)::value] = __builtin_amdgcn_raw_buffer_load_b128(rg, binq[cc] * 32u, (unsigned)k2 * kstep * 32u, 0);
            g1[Bb & 1][decltype(i)::value] = __builtin_amdgcn_raw_buffer_load_b64(rg, binq[cc] * 32u, (unsigned)k2 * kstep * 32u + 16u, 0);
        });
    };
    if (MODE == COL_FCAFFN) {
        const int b = plane / a.C, ch = plane - b * a.C;
        wa0 = a.wxa[ch * 3]; wa1 = a.wxa[ch * 3 + 1]; wa2 = a.wxa[ch * 3 + 2];
        wp0 = a.wxp[ch * 3]; wp1 = a.wxp[ch * 3 + 1]; wp2 = a.wxp[ch * 3 + 2];
        rg = cols_rsrc(a.guide + (long)b * plane_bins * 2, plane_bins * 32);
        if (worker) gload(std::integral_constant<int, 0>{});
    }
    __syncthreads();

    if (worker) {
        f2 v[32];                                   // slot q = cc * P + (n2 | bit-reversed k2)
        f2* yb = Y + k1 * KS + cg;
        // steps 3 and 4 for the FCAFFN mode.  FULL = false evaluates the 32 sincos with the fp32 range reduction only
        // (branch-free: 135 registers; with a range check at every site the allocator spills) and reports whether any
        // phase was outside its range; that case is redone from the LDS copy with the full-range form.
        auto forward = [&]() __attribute__((always_inline)) {
            sfor<0, P>([&](auto n) {
                constexpr int N2 = decltype(n)::value;
#pragma unroll
                for (int cc = 0; cc < CJ; ++cc) v[cc * P + N2] = yb[N2 * TC + cc * NG];
                if constexpr (N2 > 0) {
                    const f2 w = twl[(N2 - 1) * R + k1];
#pragma unroll
                    for (int cc = 0; cc < CJ; ++cc) v[cc * P + N2] = fftr::cmul(v[cc * P + N2], w);
                }
            });
#pragma unroll
            for (int cc = 0; cc < CJ; ++cc) fftr::fft_dif<P, false>(v + cc * P);
        };
        auto modulate = [&](auto full) __attribute__((always_inline)) -> bool {
            constexpr bool FULL = decltype(full)::value;
            bool big = false;
            sfor<0, NB>([&](auto bb) {
                constexpr int Bb = decltype(bb)::value;
                if constexpr (Bb + 1 < NB) gload(std::integral_constant<int, Bb + 1>{});
                sfor<0, BS>([&](auto i) {
                    constexpr int I = decltype(i)::value, q = Bb * BS + I;
                    const fdn_u32x4 ga = g0[Bb & 1][I];
                    const fdn_u32x2 gp = g1[Bb & 1][I];
                    const float A_ = wa0 * __uint_as_float(ga.x) + wa1 * __uint_as_float(ga.y) + wa2 * __uint_as_float(ga.z);   // conv1_xa(x_high)
                    const float ph = wp0 * __uint_as_float(ga.w) + wp1 * __uint_as_float(gp.x) + wp2 * __uint_as_float(gp.y);   // conv1_xp(xp2)
                    float sn, cs;
                    fdn_sincos<FULL>(ph, &sn, &cs);
                    if (!FULL) big |= !(fabsf(ph) < 8192.0f);
                    const f2 r = f2{rd1(v[q].x), rd1(v[q].y)};                        // FDN_arch.py:412
                    v[q] = fftr::cmul(r, f2{A_ * cs, -A_ * sn});                      // |z| A e^{i(ang z - ph)}  :413-417
                });
            });
            return big;
        };
        if (MODE != COL_INV_POLAR) forward();
        if (MODE == COL_FWD) {
            const bool evenH = (H % 2) == 0;
            const __amdgpu_buffer_rsrc_t ra = cols_rsrc(a.out_abs ? a.out_abs + (long)plane * plane_bins : nullptr, a.out_abs ? plane_bins * 4 : 0);
            const __amdgpu_buffer_rsrc_t rp = cols_rsrc(a.out_ang ? a.out_ang + (long)plane * plane_bins : nullptr, a.out_ang ? plane_bins * 4 : 0);
            sfor<0, 32>([&](auto qq) {
                constexpr int q = decltype(qq)::value, cc = q / P, k2 = fftr::brev(q % P, P);
                if (liveq[cc]) {
                    f2 x = v[q];
                    const int cl = (int)binq[cc] - k1 * Wf;
                    if (a.fix_real && (cl == 0 || cl == Wf - 1) && k1 == 0 && (k2 == 0 || (evenH && 2 * k2 == P))) x.y = 0.0f;
                    if (a.rd_before) x = f2{rd1(x.x), rd1(x.y)};
                    // (a null plane has a zero-length descriptor: the store is dropped)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sqrtf(x.x * x.x + x.y * x.y)), ra, binq[cc] * 4u, (unsigned)k2 * kstep * 4u, 0);
                    if (a.out_ang) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(atan2f(x.y, x.x)), rp, binq[cc] * 4u, (unsigned)k2 * kstep * 4u, 0);
                }
            });
            return;                                 // no barrier follows in this mode
        }
        if (MODE == COL_FCAFFN) {
            if (__builtin_expect(modulate(std::false_type{}), 0)) {
                // (round 4) the redo is a ROLLED loop over the thread's own LDS cells: inlined 32 times, the full-range sincos took the whole
                // kernel to 256 registers (two workgroups per CU); like this the cold path needs fewer registers than the hot one
                asm volatile("" ::: "memory");      // start over from memory: nothing of the first pass is kept alive for this one
                forward();                          // Y still holds the step-2 values of this thread's cells
                sfor<0, 32>([&](auto qq) { constexpr int q = decltype(qq)::value; yb[(q % P) * TC + (q / P) * NG] = v[q]; });
#pragma unroll 1
                for (int q = 0; q < 32; ++q) {
                    const int cc = q / P, n = q % P, k2 = (int)(__brev((unsigned)n) >> (P == 32 ? 27 : P == 16 ? 28 : 29));
                    unsigned bq = binq[0];
#pragma unroll
                    for (int j = 1; j < CJ; ++j) bq = cc == j ? binq[j] : bq;
                    const fdn_u32x4 ga = __builtin_amdgcn_raw_buffer_load_b128(rg, bq * 32u, (unsigned)k2 * kstep * 32u, 0);
                    const fdn_u32x2 gp = __builtin_amdgcn_raw_buffer_load_b64(rg, bq * 32u, (unsigned)k2 * kstep * 32u + 16u, 0);
                    const float A_ = wa0 * __uint_as_float(ga.x) + wa1 * __uint_as_float(ga.y) + wa2 * __uint_as_float(ga.z);
                    const float ph = wp0 * __uint_as_float(ga.w) + wp1 * __uint_as_float(gp.x) + wp2 * __uint_as_float(gp.y);
                    float sn, cs;
                    fdn_sincos<true>(ph, &sn, &cs);
                    const f2 x = yb[n * TC + cc * NG];
                    yb[n * TC + cc * NG] = fftr::cmul(f2{rd1(x.x), rd1(x.y)}, f2{A_ * cs, -A_ * sn});
                }
                sfor<0, 32>([&](auto qq) { constexpr int q = decltype(qq)::value; v[q] = yb[(q % P) * TC + (q / P) * NG]; });
            }
        }
        if (MODE == COL_INV_POLAR) {
            const long in_bins = (long)a.Hin * a.Wfin;
            const __amdgpu_buffer_rsrc_t rm = cols_rsrc(a.in_mag + (long)plane * in_bins, in_bins * 4);
            const __amdgpu_buffer_rsrc_t rp = cols_rsrc(a.in_pha + (long)plane * in_bins, in_bins * 4);
            unsigned inq[CJ];
#pragma unroll
            for (int cc = 0; cc < CJ; ++cc) inq[cc] = (unsigned)(k1 * a.Wfin + ((int)binq[cc] - k1 * Wf)) * 4u;
            const unsigned instep = (unsigned)(R * a.Wfin) * 4u;
            sfor<0, NB>([&](auto bb) {
                constexpr int Bb = decltype(bb)::value;
                float mg[BS], ph[BS];
                sfor<0, BS>([&](auto i) {
                    constexpr int I = decltype(i)::value, q = Bb * BS + I, cc = q / P, k2 = fftr::brev(q % P, P);
                    mg[I] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rm, inq[cc], (unsigned)k2 * instep, 0));
                    ph[I] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rp, inq[cc], (unsigned)k2 * instep, 0));
                });
                sfor<0, BS>([&](auto i) {
                    constexpr int I = decltype(i)::value, q = Bb * BS + I;
                    float sn, cs;
                    fdn_sincos(ph[I], &sn, &cs);
                    v[q] = f2{mg[I] * cs, mg[I] * sn};                                // FDN_arch.py:95-97
                });
            });
        }
#pragma unroll
        for (int cc = 0; cc < CJ; ++cc) fftr::fft_dit<P, true>(v + cc * P);
        asm volatile("" ::: "memory");              // re-read the twiddles: carrying the forward copies across the modulation costs 62 registers
        sfor<0, P>([&](auto n) {
            constexpr int N2 = decltype(n)::value;
            if constexpr (N2 > 0) {
                const f2 w = twl[(N2 - 1) * R + k1];
#pragma unroll
                for (int cc = 0; cc < CJ; ++cc) v[cc * P + N2] = fftr::cmulc(v[cc * P + N2], w);
            }
#pragma unroll
            for (int cc = 0; cc < CJ; ++cc) yb[N2 * TC + cc * NG] = v[cc * P + N2];
        });
    } else if (MODE == COL_FWD) {
        return;
    }
    __syncthreads();
    if (io) {
        f2 u[R];
        sfor<0, R>([&](auto k) { u[decltype(k)::value] = Y[decltype(k)::value * KS + tid]; });
        fftr::dft_nat<R, true>(u);
        if (live) sfor<0, R>([&](auto n1) { bstore_f2(u[decltype(n1)::value], rz, zoff, (unsigned)decltype(n1)::value * zstep); });
    }
}

// transposed twiddles of the R x P split: tab[n2 * R + k1] = W_H^{n2 k1}, exact on the axes
const float2* get_table_rp(int R, int P) {
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    const int key = -(R * 1024 + P);
    auto it = g_tables.find({devid, key});
    if (it != g_tables.end()) return it->second;
    const int H = R * P;
    std::vector<float2> h((size_t)R * P);
    for (int n2 = 0; n2 < P; ++n2)
        for (int k1 = 0; k1 < R; ++k1) {
            const long t = ((long)n2 * k1) % H;
            double c, s;
            if ((4 * t) % H == 0) {
                const int q = (int)((4 * t) / H);
                c = (q == 0) ? 1.0 : (q == 2 ? -1.0 : 0.0);
                s = (q == 1) ? 1.0 : (q == 3 ? -1.0 : 0.0);
            } else {
                const double ang = 2.0 * M_PI * (double)t / (double)H;
                c = cos(ang); s = sin(ang);
            }
            float2 w = make_float2((float)c, (float)(-s) + 0.0f);
            if (w.y == 0.0f) w.y = 0.0f;
            if (w.x == 0.0f) w.x = 0.0f;
            h[(size_t)n2 * R + k1] = w;
        }
    float2* d = nullptr;
    if (hipMalloc(&d, sizeof(float2) * h.size()) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    g_tables[{devid, key}] = d;
    return d;
}

template <int R, int P, int MODE>
int launch_cols_rp(ColArgs a, long planes, fdn_stream_t stream) {
    const float2* tw = get_table_rp(R, P);
    if (!tw) return FDN_ERR_LAUNCH;
    a.planes = planes;
    if (a.C <= 0 || planes % a.C != 0) a.C = 1;
    constexpr int TC = 256 / P;
    const long total = (long)cdiv(a.Wf, TC) * planes, per_xcd = (total + 7) / 8;
    if (per_xcd * 8 > 0x7FFFFFFFL) return FDN_ERR_UNSUPPORTED;
    typedef ColPlan<R, P> CP;
    constexpr size_t lds = CP::lds;
    constexpr int nt = CP::NT;
    if (int e = set_lds(fft_cols_rp_kernel<R, P, MODE>, lds)) return e;
    hipLaunchKernelGGL((fft_cols_rp_kernel<R, P, MODE>), dim3((unsigned)(per_xcd * 8)), dim3(nt), lds, static_cast<hipStream_t>(stream), a, tw);
    return fdn_launch_status();
}

// the compile-time plans: H = 23 * {32, 16, 8} (720p pyramid), 17 * {32, 16, 8} (1080p levels 2, 3 and 544-row inputs)
template <int MODE>
int launch_cols_planned(const ColArgs& a, long planes, fdn_stream_t stream, bool* done) {
    *done = true;
    switch (a.H) {
        case 23 * 32: return launch_cols_rp<23, 32, MODE>(a, planes, stream);
        case 23 * 16: return launch_cols_rp<23, 16, MODE>(a, planes, stream);
        case 23 * 8: return launch_cols_rp<23, 8, MODE>(a, planes, stream);
        case 17 * 32: return launch_cols_rp<17, 32, MODE>(a, planes, stream);
        case 17 * 16: return launch_cols_rp<17, 16, MODE>(a, planes, stream);
        case 17 * 8: return launch_cols_rp<17, 8, MODE>(a, planes, stream);
        case 34 * 32: return launch_cols_rp<34, 32, MODE>(a, planes, stream);       // 1088 rows: 1080p level 1
        // the shapes the reference's own drivers feed: LOL-Blur frames 640 x 1120 (inference_fdn_lolblur.py:16-17, already x32) and
        // LOL-v1 400 x 600 padded to 416 x 608 (inference_fdn_lolv1.py:52-64)
        case 20 * 32: return launch_cols_rp<20, 32, MODE>(a, planes, stream);
        case 20 * 16: return launch_cols_rp<20, 16, MODE>(a, planes, stream);
        case 20 * 8: return launch_cols_rp<20, 8, MODE>(a, planes, stream);
        case 13 * 32: return launch_cols_rp<13, 32, MODE>(a, planes, stream);
        case 13 * 16: return launch_cols_rp<13, 16, MODE>(a, planes, stream);
        case 13 * 8: return launch_cols_rp<13, 8, MODE>(a, planes, stream);
        default: break;
    }
    *done = false;
    return FDN_OK;
}

bool plan_big(const Plan& p) {
    for (int i = 0; i < p.nst; ++i)
        if (p.radix[i] == 17 || p.radix[i] == 23) return true;
    return false;
}
bool plan_big2(const Plan& p) {
    for (int i = 0; i < p.nst; ++i)
        if (p.radix[i] == 13 || p.radix[i] == 37 || p.radix[i] == 41) return true;
    return false;
}

int pick_tc(int H) {
    const long per_col = 2L * H * sizeof(float2);
    if (per_col * 16 <= 56 * 1024) return 16;
    if (per_col * 8 <= 140 * 1024) return 8;
    if (per_col * 4 <= 140 * 1024) return 4;
    if (per_col * 2 <= 140 * 1024) return 2;
    return 0;
}


int pick_rpb(int M) {
    // rows per workgroup from an LDS budget for the ping-pong buffers.  Swept on the B=8 720p forward (row kernels, ms per
    // step r2c / c2r): 16 KiB 16.6 / 15.0, 24 KiB 13.3 / 13.4, 32 KiB 13.3 / 12.0, 40 KiB 14.0 / 13.4, 48 KiB 14.1 / 13.4,
    // 64 KiB 15.4 / 14.5 - the passes are latency bound, so workgroups per CU count for more than rows per workgroup
    // (M = 640: 3 rows, 41 KiB with the tables, 3 workgroups per CU, 480 radix-4 jobs for 256 threads)
    int rpb = (int)((32 * 1024) / (2L * M * sizeof(float2)));
    if (rpb > 8) rpb = 8;
    if (rpb < 1) rpb = 1;
    return rpb;
}

// in-place passes possible: every radix has a register butterfly and its jobs fit NT*JMAX threads-slots
int inplace_tc(const Plan& p, int H) {
    for (int tc = 32; tc >= 8; tc >>= 1) {
        if ((long)H * tc > (long)NT * 24) continue;               // keep the buffer <= 48 KiB: 3 workgroups per CU
        bool ok = true;
        for (int i = 0; i < p.nst && ok; ++i) {
            const int R = p.radix[i];
            if (!(R == 2 || R == 3 || R == 4 || R == 5 || R == 7 || R == 17 || R == 23)) ok = false;
            else if ((long)(H / R) * tc > (long)NT * (EMAX / R)) ok = false;
        }
        if (ok) return tc;
    }
    return 0;
}

template <int MODE, int BIG, bool INPL>
int launch_cols_k(ColArgs a, const Plan& p, long planes, size_t lds, fdn_stream_t stream) {
    a.planes = planes;
    if (a.C <= 0 || planes % a.C != 0) a.C = 1;
    if (int e = set_lds(fft_cols_kernel<MODE, BIG, INPL>, lds)) return e;
    const long total = (long)cdiv(a.Wf, a.tc) * planes, per_xcd = (total + 7) / 8;
    if (per_xcd * 8 > 0x7FFFFFFFL) return FDN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((fft_cols_kernel<MODE, BIG, INPL>), dim3((unsigned)(per_xcd * 8)), dim3(NT), lds,
                       static_cast<hipStream_t>(stream), a, p);
    return fdn_launch_status();
}

template <int MODE>
int launch_cols(ColArgs a, long planes, fdn_stream_t stream) {
    {
        bool done = false;
        const int e = launch_cols_planned<MODE>(a, planes, stream, &done);
        if (done) return e;
    }
    Plan p;
    if (!make_plan(a.H, a.H, &p)) return FDN_ERR_UNSUPPORTED;
    const int itc = inplace_tc(p, a.H);
    if (itc > 0) {
        a.tc = itc;
        a.tcs = __builtin_ctz(itc);
        const size_t lds = ((size_t)a.H * a.tc + a.H) * sizeof(float2);
        return plan_big(p) ? launch_cols_k<MODE, 1, true>(a, p, planes, lds, stream)
                           : launch_cols_k<MODE, 0, true>(a, p, planes, lds, stream);
    }
    a.tc = pick_tc(a.H);
    if (a.tc == 0) return FDN_ERR_UNSUPPORTED;
    a.tcs = __builtin_ctz(a.tc);
    const size_t lds = (2UL * a.H * a.tc + a.H) * sizeof(float2);
    if (plan_big2(p)) return launch_cols_k<MODE, 2, false>(a, p, planes, lds, stream);
    return plan_big(p) ? launch_cols_k<MODE, 1, false>(a, p, planes, lds, stream)
                       : launch_cols_k<MODE, 0, false>(a, p, planes, lds, stream);
}

// ------------------------------------------------------------------------------------------
// Rows with a compile-time plan: half-length M = R1 * P with R1 = 20 (720p: W = 1280 / 640 / 320) or 30 (1080p: 1920 /
// 960 / 480) and P = 32 / 16 / 8.  Same scheme as the planned columns: thread (n2, row) runs the R1-point DFT over n1 on
// values loaded straight from memory, LDS transposes, thread (k1, row group) runs the P-point FFTs of its 32 / P rows, and the
// r2c split (c2r merge) works on natural-order rows in LDS with coalesced stores (loads).  256 / P rows per workgroup, two
// LDS round trips; the generic passes needed five (R = 5, 4, 4, 4, 2 for M = 640) plus a table copy per 3 rows.
//   LDS: Y[row][k1][n2] with k1 stride P + 1 (conflict-free for both access directions), reused as Z[row][k]; then the
//   transposed twiddles tw2[(n2 - 1) R1 + k1] = W_M^{n2 k1} and the split twiddles W_W^k, k <= M.
// ------------------------------------------------------------------------------------------
// (round 5) R1 = 35 - the reference's own LOL-Blur frames, 640 x 1120: W = 1120 / 560 / 280 = 2 x 35 x {16, 8, 4} - does not fit eight row groups: 35 x 8 = 280
// second-stage jobs for 256 threads and 85 KB of LDS.  The number of row groups NG is therefore part of the plan: the largest that keeps R1 * NG <= 256
// jobs AND two workgroups per CU (<= 80 KB): 7 at P = 16 / 8 (14 / 28 rows per workgroup, 245 jobs, 75 KB), 6 at P = 4 (48 rows, 210 jobs, 69 KB).
constexpr int row_plan_groups(int R1, int P) {
    int ng = 8;
    while (ng > 1 && (R1 * ng > 256 || ((long)ng * (32 / P) * R1 * (P + 1) + (P - 1) * R1 + R1 * P + 1) * 8 > 80 * 1024)) --ng;
    return ng;
}
template <int R1, int P>
struct RowPlan {
    static constexpr int NG = row_plan_groups(R1, P);
    static constexpr int M = R1 * P, W = 2 * M, Wf = M + 1, CJ = 32 / P, RW = NG * CJ, PS = P + 1, RS = R1 * PS, NJ = R1 * NG;
    static constexpr int NTW = (P - 1) * R1 + Wf;
    static constexpr size_t lds = ((size_t)RW * RS + NTW) * sizeof(float2);
    static_assert(RW * P <= 256 && NJ <= 256 && CJ * P <= 32, "first stage: one thread per (row, n2); second stage: one per (row group, k1) with 32 values");
};

// LN: the input rows are planes of x [B][C][H][W] and the transform is taken of the channel LayerNorm of x
// ((x - mean) * rstd * gamma_c + beta_c with the per-pixel statistics ln.stats [B][2][H*W]) - FCAFFN's norm3 in front of rfft2
// (FDN_arch.py:675, :411) without the normalised tensor ever being written.
struct RowLN {
    const float* stats;
    const float* gamma;
    const float* beta;
    int C, H;
};

template <int R1, int P, bool LN = false>
__global__ __launch_bounds__(256, 2) void rfft_rows_rp_kernel(const float* __restrict__ in, float2* __restrict__ out, long R,
                                                              const float2* __restrict__ tab, RowLN lnp, int pitch) {
    typedef RowPlan<R1, P> L;
    constexpr int M = L::M, W = L::W, Wf = L::Wf, RW = L::RW, CJ = L::CJ, PS = L::PS, RS = L::RS, NJ = L::NJ;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f2* Y = reinterpret_cast<f2*>(smem);
    f2* tw2 = Y + RW * RS;
    f2* tws = tw2 + (P - 1) * R1;
    const int tid = threadIdx.x;
    for (int i = tid; i < L::NTW; i += 256) tw2[i] = f2{tab[i].x, tab[i].y};
    const long row0 = (long)blockIdx.x * RW;
    const int nrow = (int)min((long)RW, R - row0);
    {
        const int n2 = tid & (P - 1), rw = tid / P < RW ? tid / P : RW - 1;  // (plans with RW P < 256: the threads past the last row shadow it - same
        const int rwc = rw < nrow ? rw : nrow - 1;                           //  values to the same cells); rows past the end shadow the last one
        const __amdgpu_buffer_rsrc_t rin = cols_rsrc(in + row0 * W, (long)nrow * W * 4);
        const unsigned voff = (unsigned)(rwc * W + 2 * n2) * 4u;
        f2 u[R1];
        sfor<0, R1>([&](auto n1) { u[decltype(n1)::value] = bload_f2(rin, voff, (unsigned)(decltype(n1)::value * P * 8)); });
        if constexpr (LN) {
            const long row = row0 + rwc, plane = row / lnp.H;
            const int h = (int)(row - plane * lnp.H), c = (int)(plane % lnp.C);
            const long b = plane / lnp.C, HW = (long)lnp.H * W;
            // ONE descriptor over the whole statistics tensor, the batch item in the per-lane offset: the rows of a workgroup may belong to two
            // batch items, so a descriptor per item is a per-lane value and every load behind it sits in a waterfall loop (40-60 per thread,
            // tools/isa_waterfall.py).  (The host refuses tensors past 2^31 bytes.)
            const __amdgpu_buffer_rsrc_t rs_ = cols_rsrc(lnp.stats, (R / ((long)lnp.C * lnp.H)) * 2 * HW * 4);
            const unsigned so = (unsigned)(b * 2 * HW + h * W + 2 * n2) * 4u;
            const float ga = lnp.gamma[c], be = lnp.beta[c];
            f2 mu[R1], rs[R1];
            sfor<0, R1>([&](auto n1) {
                constexpr int N1 = decltype(n1)::value;
                mu[N1] = bload_f2(rs_, so, (unsigned)(N1 * P * 8));
                rs[N1] = bload_f2(rs_, so + (unsigned)HW * 4u, (unsigned)(N1 * P * 8));
            });
            sfor<0, R1>([&](auto n1) {
                constexpr int N1 = decltype(n1)::value;
                u[N1] = (u[N1] - mu[N1]) * rs[N1] * ga + be;
            });
        }
        fftr::dft_nat<R1, false>(u);
        sfor<0, R1>([&](auto k1) { Y[rw * RS + decltype(k1)::value * PS + n2] = u[decltype(k1)::value]; });
    }
    __syncthreads();
    const int jt = (tid + 64 * (int)(blockIdx.x & 3)) & 255;
    const bool worker = jt < NJ;
    const int rg = jt / R1, k1 = jt - rg * R1;
    f2 v[32];
    if (worker) {
        const f2* yb = Y + rg * CJ * RS + k1 * PS;
        sfor<0, P>([&](auto n) {
            constexpr int N2 = decltype(n)::value;
#pragma unroll
            for (int cc = 0; cc < CJ; ++cc) v[cc * P + N2] = yb[cc * RS + N2];
            if constexpr (N2 > 0) {
                const f2 w = tw2[(N2 - 1) * R1 + k1];
#pragma unroll
                for (int cc = 0; cc < CJ; ++cc) v[cc * P + N2] = fftr::cmul(v[cc * P + N2], w);
            }
        });
#pragma unroll
        for (int cc = 0; cc < CJ; ++cc) fftr::dft_nat<P, false>(v + cc * P);
    }
    __syncthreads();                                                        // every Y cell has been read
    if (worker) {
        f2* zb = Y + rg * CJ * RS + k1;
        sfor<0, P>([&](auto k2) {
#pragma unroll
            for (int cc = 0; cc < CJ; ++cc) zb[cc * RS + R1 * decltype(k2)::value] = v[cc * P + decltype(k2)::value];
        });
    }
    __syncthreads();
    // split: X[k] = E[k] + W_W^k O[k],  E = (Z[k] + conj Z[M-k]) / 2,  O = -i (Z[k] - conj Z[M-k]) / 2
    // (rows may be padded to `pitch` >= Wf bins - a multiple of 16 keeps every row on a 128-byte line for the column pass; the
    //  padding is written as zeros)
    f2* out2 = reinterpret_cast<f2*>(out) + row0 * pitch;
    const int live = nrow * pitch;
    const float r_Wf = 1.0f / (float)pitch;
#pragma unroll 4
    for (int idx = tid; idx < live; idx += 256) {
        const int s = fdiv(idx, r_Wf), k = idx - s * pitch;
        if (k >= Wf) { out2[idx] = f2{0.f, 0.f}; continue; }
        const f2 zk = Y[s * RS + (k == M ? 0 : k)];
        const f2 zc = Y[s * RS + (k == 0 ? 0 : M - k)];
        const f2 e = 0.5f * f2{zk.x + zc.x, zk.y - zc.y};
        const f2 d = 0.5f * f2{zk.x - zc.x, zk.y + zc.y};                   // (Z - conj Zc) / 2
        const f2 o = fftr::mul_ni(d);
        f2 x;
        if (k == 0) x = f2{e.x + o.x, 0.0f};
        else if (k == M) x = f2{e.x - o.x, 0.0f};
        else x = e + fftr::cmul(o, tws[k]);
        out2[idx] = x;
    }
}

template <int R1, int P>
__global__ __launch_bounds__(256, 2) void irfft_rows_rp_kernel(const float2* __restrict__ in, long in_ws, long in_plane_stride,
                                                               float* __restrict__ out, int H, long R, float scale,
                                                               const float* __restrict__ res, float alpha,
                                                               const float2* __restrict__ tab) {
    typedef RowPlan<R1, P> L;
    constexpr int M = L::M, W = L::W, RW = L::RW, CJ = L::CJ, PS = L::PS, RS = L::RS, NJ = L::NJ;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f2* Y = reinterpret_cast<f2*>(smem);
    f2* tw2 = Y + RW * RS;
    f2* tws = tw2 + (P - 1) * R1;
    __shared__ long rowoff[RW];
    const int tid = threadIdx.x;
    for (int i = tid; i < L::NTW; i += 256) tw2[i] = f2{tab[i].x, tab[i].y};
    const long row0 = (long)blockIdx.x * RW;
    const int nrow = (int)min((long)RW, R - row0);
    if (tid < RW) {
        const long row = row0 + min(tid, nrow - 1);
        const long plane = row / H, h = row - plane * H;
        rowoff[tid] = plane * in_plane_stride + h * in_ws;
    }
    __syncthreads();
    {   // merge on pairs (k, M - k), every bin loaded once:  E = (X[k] + conj X[M-k]) / 2,  O = (X[k] - conj X[M-k]) / 2 * W_W^{-k},
        // Z[k] = E + i O,  Z[M-k] = conj(E - i O)  (W_W^{M-k} = -conj W_W^k): one twiddle product per pair
        constexpr int UL = 4, NP = M / 2 + 1, TOT = RW * NP;
        const f2* in2 = reinterpret_cast<const f2*>(in);
        const float r_NP = 1.0f / (float)NP;
        for (int base = tid; base < TOT; base += 256 * UL) {
            f2 xks[UL], xcs[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int idx = min(base + 256 * u, TOT - 1), s = fdiv(idx, r_NP), k = idx - s * NP;
                const f2* src = in2 + rowoff[s];
                xks[u] = src[k];
                xcs[u] = src[M - k];
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int idx = base + 256 * u;
                if (idx >= TOT) continue;
                const int s = fdiv(idx, r_NP), k = idx - s * NP;
                f2 xk = xks[u], xc = xcs[u];
                if (k == 0) { xk.y = 0.f; xc.y = 0.f; }                     // c2r ignores Im of DC and Nyquist
                const f2 e = 0.5f * f2{xk.x + xc.x, xk.y - xc.y};
                const f2 d = 0.5f * f2{xk.x - xc.x, xk.y + xc.y};
                const f2 io = fftr::mul_pi(fftr::cmulc(d, tws[k]));           // i * O
                f2* row = Y + s * RS;
                row[k] = e + io;
                if (k != 0) {                                                // (2k = M: the same value twice)
                    const f2 c = e - io;
                    row[M - k] = f2{c.x, -c.y};
                }
            }
        }
    }
    __syncthreads();
    const int jt = (tid + 64 * (int)(blockIdx.x & 3)) & 255;
    const bool worker = jt < NJ;
    const int rg = jt / R1, k1 = jt - rg * R1;
    f2 v[32];
    if (worker) {
        const f2* zb = Y + rg * CJ * RS + k1;
        sfor<0, P>([&](auto k2) {
#pragma unroll
            for (int cc = 0; cc < CJ; ++cc) v[cc * P + decltype(k2)::value] = zb[cc * RS + R1 * decltype(k2)::value];
        });
#pragma unroll
        for (int cc = 0; cc < CJ; ++cc) fftr::dft_nat<P, true>(v + cc * P);
    }
    __syncthreads();                                                        // every Z cell has been read
    if (worker) {
        f2* yb = Y + rg * CJ * RS + k1 * PS;
        sfor<0, P>([&](auto n) {
            constexpr int N2 = decltype(n)::value;
            if constexpr (N2 > 0) {
                const f2 w = tw2[(N2 - 1) * R1 + k1];
#pragma unroll
                for (int cc = 0; cc < CJ; ++cc) v[cc * P + N2] = fftr::cmulc(v[cc * P + N2], w);
            }
#pragma unroll
            for (int cc = 0; cc < CJ; ++cc) yb[cc * RS + N2] = v[cc * P + N2];
        });
    }
    __syncthreads();
    {
        const int n2 = tid & (P - 1), rw = tid / P;
        const int rwl = rw < RW ? rw : RW - 1;
        f2 u[R1];
        sfor<0, R1>([&](auto k) { u[decltype(k)::value] = Y[rwl * RS + decltype(k)::value * PS + n2]; });
        fftr::dft_nat<R1, true>(u);
        if (rw < nrow) {
            const __amdgpu_buffer_rsrc_t ro = cols_rsrc(out + row0 * W, (long)nrow * W * 4);
            const __amdgpu_buffer_rsrc_t rr = cols_rsrc(res ? res + row0 * W : out, res ? (long)nrow * W * 4 : 0);
            const unsigned voff = (unsigned)(rw * W + 2 * n2) * 4u;
            f2 r[R1];
            if (res) sfor<0, R1>([&](auto n1) { r[decltype(n1)::value] = bload_f2(rr, voff, (unsigned)(decltype(n1)::value * P * 8)); });
            sfor<0, R1>([&](auto n1) {
                constexpr int N1 = decltype(n1)::value;
                f2 x = u[N1] * scale;
                if (res) x += alpha * r[N1];
                bstore_f2(x, ro, voff, (unsigned)(N1 * P * 8));
            });
        }
    }
}

// [ (P-1) * R1 transposed twiddles W_M^{n2 k1} | M + 1 split twiddles W_W^k ], exact on the axes
const float2* get_table_rows_rp(int R1, int P) {
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    const int key = -(1 << 24) - (R1 * 1024 + P);
    auto it = g_tables.find({devid, key});
    if (it != g_tables.end()) return it->second;
    const int M = R1 * P, W = 2 * M;
    auto root = [](long t, long N) {
        t %= N;
        double c, s;
        if ((4 * t) % N == 0) {
            const int q = (int)((4 * t) / N);
            c = (q == 0) ? 1.0 : (q == 2 ? -1.0 : 0.0);
            s = (q == 1) ? 1.0 : (q == 3 ? -1.0 : 0.0);
        } else {
            const double ang = 2.0 * M_PI * (double)t / (double)N;
            c = cos(ang); s = sin(ang);
        }
        float2 w = make_float2((float)c, (float)(-s) + 0.0f);
        if (w.y == 0.0f) w.y = 0.0f;
        if (w.x == 0.0f) w.x = 0.0f;
        return w;
    };
    std::vector<float2> h;
    for (int n2 = 1; n2 < P; ++n2)
        for (int k1 = 0; k1 < R1; ++k1) h.push_back(root((long)n2 * k1, M));
    for (int k = 0; k <= M; ++k) h.push_back(root(k, W));
    float2* d = nullptr;
    if (hipMalloc(&d, sizeof(float2) * h.size()) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    g_tables[{devid, key}] = d;
    return d;
}

template <int R1, int P>
int launch_rfft_rp(const float* in, float* out_c, long rows, fdn_stream_t stream, int pitch, const RowLN* ln = nullptr) {
    typedef RowPlan<R1, P> L;
    const float2* tab = get_table_rows_rp(R1, P);
    if (!tab) return FDN_ERR_LAUNCH;
    if (ln) {
        if (int e = set_lds(rfft_rows_rp_kernel<R1, P, true>, L::lds)) return e;
        hipLaunchKernelGGL((rfft_rows_rp_kernel<R1, P, true>), dim3(cdiv(rows, L::RW)), dim3(256), L::lds, static_cast<hipStream_t>(stream),
                           in, reinterpret_cast<float2*>(out_c), rows, tab, *ln, pitch);
        return fdn_launch_status();
    }
    if (int e = set_lds(rfft_rows_rp_kernel<R1, P, false>, L::lds)) return e;
    hipLaunchKernelGGL((rfft_rows_rp_kernel<R1, P, false>), dim3(cdiv(rows, L::RW)), dim3(256), L::lds, static_cast<hipStream_t>(stream), in,
                       reinterpret_cast<float2*>(out_c), rows, tab, RowLN{}, pitch);
    return fdn_launch_status();
}

template <int R1, int P>
int launch_irfft_rp(const float* in_c, long in_row_bins, long in_plane_bins, float* out, long planes, int H, float scale,
                    const float* res, float alpha, fdn_stream_t stream) {
    typedef RowPlan<R1, P> L;
    const float2* tab = get_table_rows_rp(R1, P);
    if (!tab) return FDN_ERR_LAUNCH;
    if (int e = set_lds(irfft_rows_rp_kernel<R1, P>, L::lds)) return e;
    const long rows = planes * H;
    hipLaunchKernelGGL((irfft_rows_rp_kernel<R1, P>), dim3(cdiv(rows, L::RW)), dim3(256), L::lds, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float2*>(in_c), in_row_bins, in_plane_bins, out, H, rows, scale, res, alpha, tab);
    return fdn_launch_status();
}

// W -> (R1, P) of the planned row kernels; 0 = none
bool rows_plan(int W, int* R1, int* P) {
    for (int r : {20, 30})
        for (int p : {32, 16, 8})
            if (W == 2 * r * p) { *R1 = r; *P = p; return true; }
    for (int p : {16, 8})                                   // LOL-v1 padded: W = 608 / 304 (19 x 16, 19 x 8; 152 = 2 x 19 x 4 stays generic)
        if (W == 2 * 19 * p) { *R1 = 19; *P = p; return true; }
    for (int p : {16, 8, 4})                                // LOL-Blur frames (inference_fdn_lolblur.py:16-17): W = 1120 / 560 / 280
        if (W == 2 * 35 * p) { *R1 = 35; *P = p; return true; }
    return false;
}
#define FDN_ROWS_DISPATCH(CALL)                                        \
    switch (R1 * 64 + P) {                                             \
        case 20 * 64 + 32: return CALL(20, 32);                        \
        case 20 * 64 + 16: return CALL(20, 16);                        \
        case 20 * 64 + 8: return CALL(20, 8);                          \
        case 30 * 64 + 32: return CALL(30, 32);                        \
        case 30 * 64 + 16: return CALL(30, 16);                        \
        case 30 * 64 + 8: return CALL(30, 8);                          \
        case 19 * 64 + 16: return CALL(19, 16);                        \
        case 19 * 64 + 8: return CALL(19, 8);                          \
        case 35 * 64 + 16: return CALL(35, 16);                        \
        case 35 * 64 + 8: return CALL(35, 8);                          \
        case 35 * 64 + 4: return CALL(35, 4);                          \
        default: break;                                                \
    }

}  // namespace

extern "C" int fdn_fft_prepare(int n) {
    FDN_CHECK_ARG(n > 0);
    for (int R : {23, 17, 20, 13})
        for (int P : {32, 16, 8})
            if (n == R * P && !get_table_rp(R, P)) return FDN_ERR_LAUNCH;      // column lengths with a compile-time plan
    if (n == 34 * 32 && !get_table_rp(34, 32)) return FDN_ERR_LAUNCH;
    {
        int R1 = 0, P = 0;                                                      // row widths with a compile-time plan
        if (rows_plan(n, &R1, &P) && !get_table_rows_rp(R1, P)) return FDN_ERR_LAUNCH;
    }
    return get_table(n) ? FDN_OK : FDN_ERR_LAUNCH;
}

namespace {
__global__ __launch_bounds__(256) void sincos_kernel(const float* __restrict__ x, float* __restrict__ sn, float* __restrict__ cs, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        float s, c;
        fdn_sincos(x[i], &s, &c);
        sn[i] = s;
        cs[i] = c;
    }
}
}  // namespace

extern "C" int fdn_sincos_f32(const float* x, float* sn, float* cs, long n, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && sn && cs && n > 0 && n < (1L << 39));
    hipLaunchKernelGGL(sincos_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, sn, cs, n);
    return fdn_launch_status();
}

extern "C" int fdn_rfft_rows(const float* in, float* out_c, long rows, int W, long out_row_bins, fdn_stream_t stream) {
    FDN_CHECK_ARG(in && out_c && rows > 0 && W >= 2 && W % 2 == 0 && (out_row_bins == 0 || (out_row_bins >= W / 2 + 1 && out_row_bins < (1L << 20))));
    const int pitch = out_row_bins ? (int)out_row_bins : W / 2 + 1;
    {
        int R1 = 0, P = 0;
        if (rows_plan(W, &R1, &P) && (reinterpret_cast<uintptr_t>(in) & 7) == 0) {
#define FDN_CALL(a, b) launch_rfft_rp<a, b>(in, out_c, rows, stream, pitch)
            FDN_ROWS_DISPATCH(FDN_CALL)
#undef FDN_CALL
        }
    }
    Plan p;
    if (!make_plan(W / 2, W, &p)) return FDN_ERR_UNSUPPORTED;
    Rader rd = {};
    const bool rader = p.nst == 1 && get_rader(W / 2, &rd);            // prime half-length: convolution form instead of the O(N^2) gather
    if (!rader) rd = Rader{};
    const int rpb = pick_rpb(W / 2);
    const size_t lds = (2UL * rpb * (W / 2) + W + (rader ? (size_t)(W / 2) + rpb : 0)) * sizeof(float2);
    if (lds > 160 * 1024) return FDN_ERR_UNSUPPORTED;
    if (plan_big(p) || (rader && plan_big(rd.sub))) {
        if (int e = set_lds(rfft_rows_kernel<true>, lds)) return e;
        hipLaunchKernelGGL(rfft_rows_kernel<true>, dim3(cdiv(rows, rpb)), dim3(NT), lds, static_cast<hipStream_t>(stream), in,
                           reinterpret_cast<float2*>(out_c), W, rows, rpb, p, rd, pitch);
    } else {
        if (int e = set_lds(rfft_rows_kernel<false>, lds)) return e;
        hipLaunchKernelGGL(rfft_rows_kernel<false>, dim3(cdiv(rows, rpb)), dim3(NT), lds, static_cast<hipStream_t>(stream), in,
                           reinterpret_cast<float2*>(out_c), W, rows, rpb, p, rd, pitch);
    }
    return fdn_launch_status();
}

extern "C" int fdn_rfft_rows_ln(const float* x, const float* stats, const float* gamma, const float* beta, float* out_c, int B, int C,
                                int H, int W, long out_row_bins, fdn_stream_t stream) {
    FDN_CHECK_ARG(x && stats && gamma && beta && out_c && B > 0 && C > 0 && H > 0 && W >= 2 && W % 2 == 0);
    FDN_CHECK_ARG(out_row_bins == 0 || (out_row_bins >= W / 2 + 1 && out_row_bins < (1L << 20)));
    const int pitch = out_row_bins ? (int)out_row_bins : W / 2 + 1;
    FDN_CHECK_ARG((long)H * W < (1L << 28));
    int R1 = 0, P = 0;
    if (!rows_plan(W, &R1, &P) || (reinterpret_cast<uintptr_t>(x) & 7) != 0 || (reinterpret_cast<uintptr_t>(stats) & 7) != 0)
        return FDN_ERR_UNSUPPORTED;                           // widths with a compile-time plan only: else fdn_layernorm_chan + fdn_rfft_rows
    if ((long)B * 2 * H * W * 4 > 0x7FFFFFFFL) return FDN_ERR_UNSUPPORTED;      // the statistics of all batch items sit behind one 2 GB descriptor
    const RowLN ln = {stats, gamma, beta, C, H};
    const long rows = (long)B * C * H;
#define FDN_CALL(a, b) launch_rfft_rp<a, b>(x, out_c, rows, stream, pitch, &ln)
    FDN_ROWS_DISPATCH(FDN_CALL)
#undef FDN_CALL
    return FDN_ERR_UNSUPPORTED;
}

extern "C" int fdn_irfft_rows(const float* in_c, long in_row_bins, long in_plane_bins, float* out, long planes, int H, int W,
                              float scale, const float* res, float alpha, fdn_stream_t stream) {
    FDN_CHECK_ARG(in_c && out && planes > 0 && H > 0 && W >= 2 && W % 2 == 0 && in_row_bins >= W / 2 + 1);
    {
        int R1 = 0, P = 0;
        if (rows_plan(W, &R1, &P) && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res)) & 7) == 0) {
#define FDN_CALL(a, b) launch_irfft_rp<a, b>(in_c, in_row_bins, in_plane_bins, out, planes, H, scale, res, alpha, stream)
            FDN_ROWS_DISPATCH(FDN_CALL)
#undef FDN_CALL
        }
    }
    Plan p;
    if (!make_plan(W / 2, W, &p)) return FDN_ERR_UNSUPPORTED;
    const int rpb = pick_rpb(W / 2);
    const size_t lds = (2UL * rpb * (W / 2) + W) * sizeof(float2);
    if (lds > 160 * 1024) return FDN_ERR_UNSUPPORTED;
    const long rows = planes * H;
    if (plan_big(p)) {
        if (int e = set_lds(irfft_rows_kernel<true>, lds)) return e;
        hipLaunchKernelGGL(irfft_rows_kernel<true>, dim3(cdiv(rows, rpb)), dim3(NT), lds, static_cast<hipStream_t>(stream),
                           reinterpret_cast<const float2*>(in_c), in_row_bins, (long)H, in_plane_bins, out, W, H, rows, rpb,
                           scale, res, alpha, p);
    } else {
        if (int e = set_lds(irfft_rows_kernel<false>, lds)) return e;
        hipLaunchKernelGGL(irfft_rows_kernel<false>, dim3(cdiv(rows, rpb)), dim3(NT), lds, static_cast<hipStream_t>(stream),
                           reinterpret_cast<const float2*>(in_c), in_row_bins, (long)H, in_plane_bins, out, W, H, rows, rpb,
                           scale, res, alpha, p);
    }
    return fdn_launch_status();
}

__global__ __launch_bounds__(256) void pack_guidance_kernel(const float* __restrict__ amp, const float* __restrict__ pha,
                                                            float4* __restrict__ out, int H, int Wf, int pitch, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;           // (b, h, w) flat over padded rows
    if (i >= total) return;
    const long row = i / pitch;
    const int w = (int)(i - row * pitch);
    if (w >= Wf) {                                                  // padding of a row: zeros (phase 0, amplitude 0)
        out[2 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
        out[2 * i + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const long b = row / H, bins = (long)H * Wf, o = (row - b * H) * Wf + w;
    const float* ap = amp + b * 3 * bins + o;
    const float* pp = pha + b * 3 * bins + o;
    out[2 * i] = make_float4(ap[0], ap[bins], ap[2 * bins], pp[0]);
    out[2 * i + 1] = make_float4(pp[bins], pp[2 * bins], 0.f, 0.f);
}

extern "C" int fdn_pack_guidance(const float* amp, const float* pha, float* packed, int B, int H, int Wf, long row_bins,
                                 fdn_stream_t stream) {
    FDN_CHECK_ARG(amp && pha && packed && B > 0 && H > 0 && Wf > 0 && (row_bins == 0 || (row_bins >= Wf && row_bins < (1L << 20))));
    FDN_CHECK_ARG((reinterpret_cast<uintptr_t>(packed) & 15) == 0);
    const int pitch = row_bins ? (int)row_bins : Wf;
    const long total = (long)B * H * pitch;
    hipLaunchKernelGGL(pack_guidance_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), amp,
                       pha, reinterpret_cast<float4*>(packed), H, Wf, pitch, total);
    return fdn_launch_status();
}

extern "C" int fdn_fft_cols_fcaffn(float* z, const float* guide, const float* wxa, const float* wxp, int B, int C, int H,
                                   int Wf, fdn_stream_t stream) {
    FDN_CHECK_ARG(z && guide && wxa && wxp && B > 0 && C > 0 && H > 0 && Wf > 0);
    FDN_CHECK_ARG((reinterpret_cast<uintptr_t>(guide) & 15) == 0);
    ColArgs a = {};
    a.z = reinterpret_cast<float2*>(z);
    a.H = H; a.Wf = Wf; a.C = C;
    a.guide = reinterpret_cast<const float4*>(guide); a.wxa = wxa; a.wxp = wxp;
    return launch_cols<COL_FCAFFN>(a, (long)B * C, stream);
}

extern "C" int fdn_fft_cols_fwd(const float* z, float* out_abs, float* out_ang, long planes, int H, int Wf, int rd_before,
                                int fix_real, fdn_stream_t stream) {
    FDN_CHECK_ARG(z && (out_abs || out_ang) && planes > 0 && H > 0 && Wf > 0);
    ColArgs a = {};
    a.z = reinterpret_cast<float2*>(const_cast<float*>(z));
    a.H = H; a.Wf = Wf; a.C = 1;
    a.out_abs = out_abs; a.out_ang = out_ang; a.rd_before = rd_before; a.fix_real = fix_real;
    return launch_cols<COL_FWD>(a, planes, stream);
}

extern "C" int fdn_fft_cols_inv_polar(const float* mag, const float* pha, int Hin, int Wfin, float* z_out, long planes, int H,
                                      int Wf, fdn_stream_t stream) {
    FDN_CHECK_ARG(mag && pha && z_out && planes > 0 && H > 0 && Wf > 0 && Hin >= H && Wfin >= Wf);
    ColArgs a = {};
    a.z = reinterpret_cast<float2*>(z_out);
    a.H = H; a.Wf = Wf; a.C = 1;
    a.in_mag = mag; a.in_pha = pha; a.Hin = Hin; a.Wfin = Wfin;
    return launch_cols<COL_INV_POLAR>(a, planes, stream);
}
