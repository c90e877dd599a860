/* fdn_hip.h -- C ABI of libfdn_hip.so, the MI355X (gfx950) kernels behind the FDN inference path.
 *
 * Drop-in boundary: the reference is pure PyTorch, so "the reference FFI" for
 * this path is the set of aten op sequences inside basicsr/models/archs/FDN_arch.py and
 * LPNet_arch.py.  Each entry point below replaces one such sequence (cited per function, paths
 * relative to the reference root) and is called from the host-side mirror of those modules
 * (fdn-tip2025_amd/basicsr/models/archs/FDN_arch.py) through ctypes.
 *
 * Conventions
 *   - every tensor is a raw DEVICE pointer to contiguous fp32 NCHW data owned by the caller
 *     (PyTorch's caching allocator); complex spectra are interleaved (re, im) float pairs;
 *   - `*_bs` fields are batch strides in ELEMENTS (lets the caller pass channel slices);
 *   - no entry point allocates, frees or synchronises; work is enqueued on `stream`
 *     (a hipStream_t; NULL = the default stream).  The one exception is the immutable FFT twiddle
 *     table of a size, built on first use of that size (fdn_fft_prepare does it explicitly);
 *   - return value: FDN_OK or an FDN_ERR_* code; the Python layer maps non-zero to RuntimeError.
 */
#ifndef FDN_HIP_H
#define FDN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fdn_stream_t; /* hipStream_t */

enum { FDN_OK = 0, FDN_ERR_ARG = 1, FDN_ERR_LAUNCH = 2, FDN_ERR_WORKSPACE = 3, FDN_ERR_UNSUPPORTED = 4 };
enum { FDN_ACT_NONE = 0, FDN_ACT_LEAKY = 1, FDN_ACT_RELU = 2, FDN_ACT_SIGMOID = 3, FDN_ACT_GELU = 4 };
enum { FDN_PRO_NONE = 0, FDN_PRO_LN = 1, FDN_PRO_LN3_GATE = 2, FDN_PRO_LN_MULADD = 3 };
enum { FDN_EPI_NONE = 0, FDN_EPI_RES = 1, FDN_EPI_MULADD = 2 };

/* library version / build info: returns the ABI version (bumped on any signature change) */
int fdn_abi_version(void);
const char* fdn_error_string(int code);

/* ------------------------------------------------------------------------------------------
 * 1x1 convolution as fp32-MFMA GEMM with fused prologue / epilogue.
 * Replaces F.conv2d(k=1) at FDN_arch.py:576,639 (FDSA), :456,:474 (FDFFN), :421,:428 (FCAFFN),
 * :685-686 (Fuse), MAR's 1x1 convs (:78-86,:125-134,:168-190) together with the channel
 * LayerNorm in front (:313-342), the v_value gate (:633-638), `norm(x)*x1+x1` (:420), LeakyReLU
 * (:28), the residual adds (:671-675) and `x*mul+add` (:423).
 *   out[b][n][p] = epi( act( sum_k w[n][k] * pro(x[b][k][p]) + bias[n] ) )
 * x is the channel-concatenation of up to three segments (torch.cat at :137,:243,:250,:689).
 * pro: NONE | LN (stats/gamma/beta over K) | LN3_GATE (x = [o1|o2|o3] each ln_group channels,
 * three LayerNorms, times xb = v_value[ln_group]) | LN_MULADD (LN(x)*xb + xb).
 * stats: [B][G][2][P] = (mean, rstd) from fdn_chan_stats, G = 3 for LN3_GATE else 1.
 * epi: NONE | RES (+res) | MULADD (*mul + add).  act is applied before epi. */
typedef struct fdn_conv1x1_desc {
    const float* x[3];
    long xbs[3];
    int kseg[3];
    const float* w;     /* [N][K] */
    const float* bias;  /* [N] or NULL */
    float* out;
    long obs;
    int B, K, N, P;
    int pro, ln_group;
    const float* stats;
    const float* gamma; /* [K] */
    const float* beta;  /* [K] */
    const float* xb;
    long xbbs;
    int act, epi;
    const float* res;
    long rbs;
    const float* mul;
    const float* add;
    long mbs;
    int vec4; /* set by the library */
} fdn_conv1x1_desc;
int fdn_conv1x1(const fdn_conv1x1_desc* d, fdn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Channel LayerNorm pieces (WithBias_LayerNorm over the channel axis, FDN_arch.py:313-342).
 * fdn_chan_stats: x [B][G*E][P] (batch stride xbs) -> stats [B][G][2][P] (mean, 1/sqrt(var+1e-5)).
 * fdn_layernorm_chan: out = (x-mean)*rstd*gamma+beta, C channels. */
int fdn_chan_stats(const float* x, long xbs, float* stats, int B, int G, int E, int P, fdn_stream_t stream);
int fdn_layernorm_chan(const float* x, const float* gamma, const float* beta, float* out, int B, int C, int P,
                       fdn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * FDSA spectral core: to_hidden_dw (depthwise 3x3 over 4E channels) + 8x8-patch rfft2 of q,k,v +
 * amplitude/phase recombination + three irfft2 (FDN_arch.py:578-632), one launch.
 * hidden [B][4E][H][W] (q|k|v|v_value), dw_w [4E][9], fft_w [E][8][5]
 * out [B][4E][H][W] = (out1|out2|out3|v_value_dw), all before the LayerNorms of :633-635. */
int fdn_fdsa_core(const float* hidden, const float* dw_w, const float* fft_w, float* out, int B, int E, int H, int W,
                  fdn_stream_t stream);

/* FDFFN middle: spatial branch dw3x3 -> GELU -> dw3x3 (FDN_arch.py:435-441,457) plus frequency
 * branch 8x8 rfft2 -> replace_denormals -> amplitude*ffta, phase-fftp -> irfft2 (:458-469), summed
 * (:470).  x [B][Hd][H][W], w0,w2 [Hd][9], ffta,fftp [Hd][8][5] -> out [B][Hd][H][W]. */
int fdn_fdffn_mid(const float* x, const float* w0, const float* w2, const float* ffta, const float* fftp, float* out,
                  int B, int Hd, int H, int W, fdn_stream_t stream);

/* Gated depthwise conv: Conv2d(C, 2C, 3, groups=C) then gelu(x1)*x2 (FDN_arch.py:472-473,:426-427).
 * x [B][C][H][W], w [2C][9] -> out [B][C][H][W];
 * out[j] = gelu(dw(x[j/2], w[j])) * dw(x[(C+j)/2], w[C+j]). */
int fdn_dwconv_gate(const float* x, const float* w, float* out, int B, int C, int H, int W, fdn_stream_t stream);

/* Plain depthwise 3x3 (zero pad 1), optional activation.  x,out [B][C][H][W], w [C][9]. */
int fdn_dwconv3x3(const float* x, const float* w, float* out, int B, int C, int H, int W, int act, fdn_stream_t stream);

/* FCAFFN spatial modulation maps (FDN_arch.py:423): mul = conv3_mul(conv1_mul(img)),
 * add = conv3_add(conv1_add(img)).  img [B][3][H][W]; w1_* [C][3]; w3_* [C][9]; outs [B][C][H][W]. */
int fdn_img_mod_maps(const float* img, const float* w1_mul, const float* w3_mul, const float* w1_add,
                     const float* w3_add, float* mul, float* add, int B, int C, int H, int W, fdn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FDN_HIP_H */
