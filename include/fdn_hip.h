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
 *
 * Storage formats.  Everything is fp32 (the reference's arithmetic, BASELINE.json configs[1]) unless an entry point has an
 * `*_bf16` flag and the caller sets it: that one tensor is then STORED as bf16 (2 bytes per element, same [B][C][H][W]
 * layout, widened to fp32 on load, rounded to nearest-even on store).  Only block-internal activations of FDSA / FDFFN have
 * such a flag (out1|out2|out3|v_value; the FDFFN hidden tensors); all arithmetic, FFTs, LayerNorm statistics, the residual
 * stream and every other tensor stay fp32 (BASELINE.json configs[2], "bf16 storage").
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
enum { FDN_RS_BILINEAR_HALF = 0, FDN_RS_BILINEAR_X2 = 1, FDN_RS_NEAREST_HALF = 2, FDN_RS_NEAREST_X2 = 3, FDN_RS_PIXEL_UNSHUFFLE = 4 };

/* library version / build info: returns the ABI version (bumped on any signature change) */
int fdn_abi_version(void);
const char* fdn_error_string(int code);
/* Diagnostic switch, process-wide, default 0 = every matrix product that has a split-bf16 form runs on the bf16 matrix pipe
 * (fp32 arithmetic: operands cut exactly into three bf16 parts, six products, fp32 accumulation).  mode = 1: no kernel that issues
 * v_mfma_f32_32x32x16_bf16 / 32x32x8_bf16 is launched - the 1x1 / 3x3 convs and fdn_fdsa_out run their fp32-MFMA forms, fdn_fdsa_fused /
 * fdn_fdsa_full / fdn_fcaffn_in_packed return FDN_ERR_UNSUPPORTED (the host mirror then takes the unfused launches).  Exists so that the
 * cross-stream finding of DESIGN.md can be bisected; same results to fp32 rounding, slower.  mode = 2: as 0, except that the level-2 FDSA
 * tail (fdn_fdsa_out with E in 39..76) keeps its fp32-MFMA form - the default of ABI 10, kept for A/B runs (0.84 against 0.71 ms per launch).
 * Other values: FDN_ERR_ARG.  (No reference counterpart.) */
int fdn_set_matrix_pipe(int mode);
/* ABI 11: number of bf16-MFMA kernel launches this process has enqueued so far (every launcher counts; what mode 1 promises is that
 * this number stands still).  Diagnostic; no reference counterpart. */
long fdn_bf16_mfma_launches(void);

/* ------------------------------------------------------------------------------------------
 * 1x1 convolution as fp32-MFMA GEMM with fused prologue / epilogue.
 * Replaces F.conv2d(k=1) at FDN_arch.py:576,639 (FDSA), :456,:474 (FDFFN), :421,:428 (FCAFFN),
 * :685-686 (Fuse), MAR's 1x1 convs (:78-86,:125-134,:168-190) together with the channel
 * LayerNorm in front (:313-342), the v_value gate (:633-638), `norm(x)*x1+x1` (:420), LeakyReLU
 * (:28), the residual adds (:671-675) and `x*mul+add` (:423).
 *   out[b][n][p] = epi( act( sum_k w[n][k] * pro(x[b][k][p]) + bias[n] ) )
 * x is the channel-concatenation of up to three segments (torch.cat at :137,:243,:250,:689).
 * pro: NONE | LN ((x - mean) * rstd over K; the LayerNorm's affine part is linear and is folded by the caller:
 * w = W * diag(gamma), bias = W * beta (+ bias) - gamma / beta are ignored) | LN3_GATE (x = [o1|o2|o3] each
 * ln_group channels, three LayerNorms with gamma / beta [K], times xb = v_value[ln_group]) |
 * LN_MULADD (LN(x)*xb + xb, gamma / beta [K]).
 * stats: [B][G][2][P] = (mean, rstd) from fdn_chan_stats, G = 3 for LN3_GATE else 1.  ABI 13: NULL with LN3_GATE / LN_MULADD = the kernel
 * takes the statistics of its pixel tile itself - only the K-streaming split-bf16 kernel does (packed weights, K and N >= 96); every
 * other shape returns FDN_ERR_UNSUPPORTED and wants the fdn_chan_stats launch.
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
    float* stats_out; /* optional [B][1][2][P]: (mean, rstd) over the N output channels of `out` (needs N <= 160):
                         the LayerNorm statistics the NEXT block needs, produced in this epilogue */
    int x_bf16;       /* x[0] is stored as bf16 (xbs in elements): FDFFN project_out forms only, else FDN_ERR_UNSUPPORTED */
    int out_bf16;     /* out is stored as bf16 (obs in elements): FDFFN project_in forms only (K <= 64, N >= 2K) */
    const void* wpk;  /* optional: `w` as packed by fdn_conv1x1_pack (same N, K; ln3_E = ln_group for FDN_PRO_LN3_GATE, else 0).
                         Deep shapes (K >= 96, N >= 96, one x segment, no activation, fp32 storage) then run on the split-bf16
                         kernel; every other shape ignores it.  `w` must still be given. */
} fdn_conv1x1_desc;
int fdn_conv1x1(const fdn_conv1x1_desc* d, fdn_stream_t stream);

/* Weights of a 1x1 conv split for the bf16 matrix pipe (gemm_split.hip): every fp32 weight is cut EXACTLY into three bf16 parts
 * (w = w1 + w2 + w3 by truncation) and laid out in MFMA operand order per (128-channel tile, 32-deep K chunk).  The kernel splits
 * the activations the same way and accumulates the six leading products in fp32: fp32 arithmetic to below one rounding, at the
 * bf16 matrix rate (the fp32 MFMA runs at the vector rate).  Replaces nothing in the reference: it is F.conv2d's weight operand
 * (FDN_arch.py:576, :456, :474, :639) in the form the kernel wants, built once per weight.
 * w [N][K] fp32 (LayerNorm affine part already folded in for FDN_PRO_LN); ln3_E = E for the FDN_PRO_LN3_GATE order (K = 3E),
 * else 0; wpk: fdn_conv1x1_pack_bytes(N, K, ln3_E) bytes. */
long fdn_conv1x1_pack_bytes(int N, int K, int ln3_E);
int fdn_conv1x1_pack(const float* w, int N, int K, int ln3_E, void* wpk, fdn_stream_t stream);

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

/* FDSA front half in ONE launch (FDN_arch.py:575-632 + the LayerNorm of :668 in front): channel LayerNorm of x,
 * to_hidden (1x1 conv C -> 4E on the matrix cores, evaluated on the 1-pixel halo of each 8x32 tile) and everything
 * fdn_fdsa_core does; the 4E-channel hidden tensor never reaches HBM.
 * fdn_fdsa_pack: to_hidden weight w [4E][C] (+ optional LayerNorm gamma, beta [C], folded in: w*diag(gamma), w@beta)
 *   -> wpk [ceil(E/8)][3*ceil(C/16)+1][64][4] dwords (bf16 MFMA A operands per chunk of 8 channels, (k-step, part), lane: every
 *   weight is cut exactly into three bf16 parts, see fdn_conv1x1_pack; the last slot carries the bias row's three parts).
 * fdn_fdsa_fused: x [B][C][H][W] (batch stride xbs), stats [B][2][P] = (mean, rstd) of x over C or NULL (no
 *   LayerNorm; then pack without gamma / beta), dw_w [4E][9], fft_w [E][8][5]
 *   -> out [B][4E][H][W] = (out1|out2|out3|v_value_dw), exactly fdn_fdsa_core's output.
 * C in {24, 32, 48, 64} (the two upper levels of FDN and FDN_lolv1), else FDN_ERR_UNSUPPORTED.
 * out_bf16: store `out` as bf16 (bf16-storage mode, see "Storage formats" above). */
int fdn_fdsa_pack(const float* w, const float* gamma, const float* beta, float* wpk, int C, int E, fdn_stream_t stream);
int fdn_fdsa_fused(const float* x, long xbs, const float* stats, const float* wpk, const float* dw_w, const float* fft_w,
                   void* out, int B, int C, int E, int H, int W, int out_bf16, fdn_stream_t stream);

/* The WHOLE FDSA sub-block in one launch (FDN_arch.py:575-639, the LayerNorm of :668 in front and the residual add of :671 behind):
 * fdn_fdsa_fused and fdn_fdsa_out in one kernel, without the [B][4E][H][W] tensor between them - C planes in, C planes out.
 * The three LayerNorms over E are applied AFTER project_out through sums that are linear in the channels (pivot-shifted, so nothing
 * cancels: csrc/fdsa_full.hip has the algebra and its error bound); results agree with the two-kernel route to fp32 rounding.
 * fdn_fdsa_full_pack: to_hidden w_hidden [4E][C] (+ optional LayerNorm gamma, beta [C] of the input), project_out w_out [C][3E],
 *   gamma3 / beta3 [3E] of norm1|norm2|norm3 -> wpk (fdn_fdsa_full_pack_bytes(C, E) bytes; 0 = unsupported width).
 * fdn_fdsa_full: x [B][C][H][W] (batch stride xbs), stats [B][2][P] = (mean, rstd) of x or NULL (then pack without gamma / beta),
 *   dw_w [4E][9], fft_w [E][8][5], res [B][C][H][W] or NULL -> out [B][C][H][W] = res + project_out(...); stats_out [B][2][P] or NULL =
 *   (mean, rstd) of out over C.  C in {24, 32, 48, 64}, E <= int(1.2 C), H % 8 == 0, W % 16 == 0 (C <= 32) or W % 8 == 0; anything
 *   else (and fdn_set_matrix_pipe(1)) returns FDN_ERR_UNSUPPORTED: use fdn_fdsa_fused + fdn_fdsa_out. */
long fdn_fdsa_full_pack_bytes(int C, int E);
int fdn_fdsa_full_pack(const float* w_hidden, const float* gamma, const float* beta, const float* w_out, const float* gamma3,
                       const float* beta3, void* wpk, int C, int E, fdn_stream_t stream);
int fdn_fdsa_full(const float* x, long xbs, const float* stats, const void* wpk, const float* dw_w, const float* fft_w,
                  const float* res, float* out, float* stats_out, int B, int C, int E, int H, int W, fdn_stream_t stream);

/* The FDSA sub-block in one launch WITHOUT new arithmetic (round 6; FDN_arch.py:575-639, the LayerNorm of :668 in front, the residual of :671
 * behind): fdn_fdsa_fused's kernel, whose workgroup writes the (out1|out2|out3|v_value) planes of its 8 x 32 tile tile-contiguous into `scratch`,
 * drains its stores, meets at a barrier and then runs fdn_fdsa_out's arithmetic on them itself (L2 / Infinity-Cache read-back of its own bytes:
 * no second launch, no HBM read of the hand-off).  Results equal fdn_fdsa_fused + fdn_fdsa_out bit for bit.
 * fdn_fdsa_tail_pack: project_out w_out [N][3E], gamma3 / beta3 [3E] -> img (fdn_fdsa_tail_pack_floats(C, E, N, Hd) floats; 0 = no in-kernel tail
 *   for this width).  fdn_fdsa_scratch_floats(B, E, H, W): floats of `scratch`: a RING of one block ([4E][8][32] floats) per RESIDENT workgroup
 *   (compute unit x 2), taken at entry and given back when the tail has read it - ~80 MB in use, rewritten in place, resident in the 256 MB Infinity
 *   Cache (one block per tile would send 4.6 GB to HBM per launch at level 1).  The size does not depend on the image; the buffer starts with the slot
 *   flags and MUST BE ZERO-FILLED ONCE by the caller (every launch leaves them zero again); one buffer serves every launch of a stream.
 * fdn_fdsa_fused_tail: x, xbs, stats, wpk, dw_w, fft_w as fdn_fdsa_fused; res [B][C][H][W] or NULL (may be x); out [B][C][H][W] (must not be x);
 *   stats_out [B][2][P] or NULL.  C in {24, 32} with E <= 38 and W even, or C in {48, 64} with E in 39..76 (default matrix-pipe mode only: the
 *   level-2 tail on the bf16 pipe); anything else (and fdn_set_matrix_pipe(1)) returns FDN_ERR_UNSUPPORTED.
 * Hd > 0 (level 1 only: C <= 32, Hd <= 96, 2 Hd >= 5 C): the tail also runs the project_in of the FDFFN that follows the FDSA (FDN_arch.py:456 behind
 *   the LayerNorm of :673) on the values and statistics it holds in registers - h_out [B][Hd][H][W] = pin_w' LN(out) + pin_b', bit for bit what
 *   fdn_conv1x1(FDN_PRO_LN) returns for these operands; pin_w [Hd][C] / pin_b [Hd] are the LayerNorm-FOLDED weights (w diag(gamma), w beta) given to
 *   fdn_fdsa_tail_pack.  Hd = 0: pin_w = pin_b = h_out = NULL.  h_bf16 (level 1 only): h_out is stored as bf16 - round-to-nearest-even of the same fp32 result. */
long fdn_fdsa_tail_pack_floats(int C, int E, int N, int Hd);
long fdn_fdsa_scratch_floats(int B, int E, int H, int W);
int fdn_fdsa_tail_pack(const float* w_out, const float* gamma3, const float* beta3, const float* pin_w, const float* pin_b, float* img, int C, int E,
                       int N, int Hd, fdn_stream_t stream);
int fdn_fdsa_fused_tail(const float* x, long xbs, const float* stats, const float* wpk, const float* dw_w, const float* fft_w,
                        const float* tail_img, const float* res, float* out, float* stats_out, float* scratch, void* h_out, int B, int C,
                        int E, int H, int W, int Hd, int h_bf16, fdn_stream_t stream);

/* FDSA tail in one launch (FDN_arch.py:633-639 and the residual add of :671): norm1/2/3 over the E channels
 * of out1|out2|out3, times v_value, project_out (3E -> N) + res, and (optionally) the channel LayerNorm
 * statistics of the result.  o [B][4E][P] as written by fdn_fdsa_core; w [N][3E]; gamma3,beta3 [3E];
 * stats_out [B][2][P] or NULL.  Register-resident form, E <= 76 and N <= 64: other sizes return
 * FDN_ERR_UNSUPPORTED (use fdn_chan_stats + fdn_conv1x1 with FDN_PRO_LN3_GATE).  o_bf16: `o` is stored as bf16. */
int fdn_fdsa_out(const void* o, const float* w, const float* gamma3, const float* beta3, const float* res, float* out,
                 float* stats_out, int B, int E, int N, int P, int o_bf16, fdn_stream_t stream);

/* FDFFN middle: spatial branch dw3x3 -> GELU -> dw3x3 (FDN_arch.py:435-441,457) plus frequency
 * branch 8x8 rfft2 -> replace_denormals -> amplitude*ffta, phase-fftp -> irfft2 (:458-469), summed
 * (:470).  x [B][Hd][H][W], w0,w2 [Hd][9], ffta,fftp [Hd][8][5] -> out [B][Hd][H][W].  x_bf16 / out_bf16: storage of x / out. */
int fdn_fdffn_mid(const void* x, const float* w0, const float* w2, const float* ffta, const float* fftp, void* out,
                  int B, int Hd, int H, int W, int x_bf16, int out_bf16, fdn_stream_t stream);

/* Gated depthwise conv: Conv2d(C, 2C, 3, groups=C) then gelu(x1)*x2 (FDN_arch.py:472-473,:426-427).
 * x [B][C][H][W], w [2C][9] -> out [B][C][H][W];
 * out[j] = gelu(dw(x[j/2], w[j])) * dw(x[(C+j)/2], w[C+j]).  x_bf16 / out_bf16: storage of x / out. */
int fdn_dwconv_gate(const void* x, const float* w, void* out, int B, int C, int H, int W, int x_bf16, int out_bf16,
                    fdn_stream_t stream);

/* FDFFN / FCAFFN tail in one launch: gated depthwise conv (FDN_arch.py:472-473, :426-427) + project_out
 * (:474, :428) + residual (:673, :675) + LayerNorm statistics of the result; the gated tensor never reaches HBM.
 * y [B][C][H][W] (y_bf16: stored as bf16); dw_w [2C][9]; w [N][C]; res/out [B][N][H][W]; stats_out [B][2][H*W] or NULL.
 * form 1: sliding-window kernel (channel pairs walked per 16x64 / 8x64 pixel tile, accumulators resident; N <= 64, W % 4 == 0);
 * form 0: chunked kernel of round 1 (N <= 128, fp32 y).  Unsupported combinations return FDN_ERR_UNSUPPORTED. */
int fdn_ffn_tail(const void* y, const float* dw_w, const float* w, const float* res, float* out, float* stats_out, int B,
                 int C, int N, int H, int W, int y_bf16, int form, fdn_stream_t stream);

/* Plain depthwise 3x3 (zero pad 1), optional activation.  x,out [B][C][H][W], w [C][9]. */
int fdn_dwconv3x3(const float* x, const float* w, float* out, int B, int C, int H, int W, int act, fdn_stream_t stream);

/* FCAFFN between the inverse FFT and the gated tail in one launch (FDN_arch.py:419-423):
 *   out = project_in( norm(xi) * x1 + x1 ) * conv3_mul(conv1_mul(img)) + conv3_add(conv1_add(img))
 * replaces fdn_chan_stats(xi) -> fdn_img_mod_maps(img) -> fdn_conv1x1(FDN_PRO_LN_MULADD, FDN_EPI_MULADD): the LayerNorm statistics
 * come from the activation strip in registers, the two maps are MFMA chains on the image patch (never written to memory).
 * xi, x1, out [B][C][H][W]; img [B][3][H][W]; w [C][C]; gamma, beta [C]; w1_* [C][3]; w3_* [C][9] (all convs without bias).
 * stats1 / gamma1 / beta1 (all three or none): x1 is the channel LayerNorm of the tensor passed as x1, rebuilt on load from its
 *   per-pixel statistics [B][2][H*W] - the block's norm3 (FDN_arch.py:675) without a normalised copy in memory.
 * C = 32 or 64, W even, tensors 8-byte aligned; anything else returns FDN_ERR_UNSUPPORTED (use the three calls above). */
int fdn_fcaffn_in(const float* xi, const float* x1, const float* stats1, const float* gamma1, const float* beta1, const float* img,
                  const float* w, const float* gamma, const float* beta, const float* w1_mul, const float* w3_mul,
                  const float* w1_add, const float* w3_add, float* out, int B, int C, int H, int W, fdn_stream_t stream);

/* The same sub-block for wide layers (C >= 96: level 3, where the register strip of fdn_fcaffn_in does not fit), on the split-bf16
 * GEMM of fdn_conv1x1 (128 pixels x 128 channels per workgroup): x1's LayerNorm on load, `norm(xi) * x1 + x1` as the GEMM prologue,
 * and in the epilogue the two image maps as MFMA chains over the 27 (tap, image channel) products of the folded weights
 * w3[c][tap] * w1[c][ch] - replaces fdn_layernorm_chan(x1) + fdn_img_mod_maps(img) + fdn_conv1x1(FDN_PRO_LN_MULADD, FDN_EPI_MULADD)
 * (FDN_arch.py:419-423, :675), whose 3 C planes of intermediates are never written.
 * fdn_fcaffn_in_pack: w [C][C], w1_* [C][3], w3_* [C][9] -> wpk (fdn_fcaffn_in_pack_bytes(C) bytes), once per weight set.
 * fdn_fcaffn_in_packed: stats_xi [B][2][H*W] from fdn_chan_stats(xi), or (ABI 13) NULL: taken in the kernel; stats1 / gamma1 / beta1 as in fdn_fcaffn_in (all three or
 *   none).  C < 96 returns FDN_ERR_UNSUPPORTED (use fdn_fcaffn_in). */
long fdn_fcaffn_in_pack_bytes(int C);
int fdn_fcaffn_in_pack(const float* w, const float* w1_mul, const float* w3_mul, const float* w1_add, const float* w3_add, int C,
                       void* wpk, fdn_stream_t stream);
int fdn_fcaffn_in_packed(const float* xi, const float* stats_xi, const float* x1, const float* stats1, const float* gamma1,
                         const float* beta1, const float* img, const void* wpk, const float* gamma, const float* beta, float* out,
                         int B, int C, int H, int W, fdn_stream_t stream);

/* FCAFFN spatial modulation maps (FDN_arch.py:423): mul = conv3_mul(conv1_mul(img)),
 * add = conv3_add(conv1_add(img)).  img [B][3][H][W]; w1_* [C][3]; w3_* [C][9]; outs [B][C][H][W]. */
int fdn_img_mod_maps(const float* img, const float* w1_mul, const float* w3_mul, const float* w1_add,
                     const float* w3_add, float* mul, float* add, int B, int C, int H, int W, fdn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Full-image real 2-D FFT pipeline, norm='backward' (torch.fft.rfft2 / irfft2 at FDN_arch.py:90,98,
 * :139,147, :411,418, :882-911).  Spectra are [planes][H][W/2+1] interleaved complex.
 * Any even W and any H are accepted (mixed radix; large primes use an O(N*R) gather pass).
 * fdn_fft_prepare(n): build the immutable twiddle table of length n now (else first use does). */
int fdn_fft_prepare(int n);
/* sn[i] = sin(x[i]), cs[i] = cos(x[i]) with the library's own range reduction (what the polar <-> complex steps of the column
 * kernels evaluate for torch.cos / torch.sin at FDN_arch.py:95-97, :414-416): ~1 ulp over the whole float range, NaN for
 * Inf / NaN.  Exposed so the tests can pin it against a float64 reference at large arguments. */
int fdn_sincos_f32(const float* x, float* sn, float* cs, long n, fdn_stream_t stream);
/* r2c along rows: in [rows][W] real -> out_c [rows][out_row_bins] complex, bins W/2+1 .. out_row_bins-1 of a row written as zeros
 * (out_row_bins = 0: dense rows of W/2+1).  A pitch that is a multiple of 16 bins starts every spectrum row on a 128-byte line:
 * the column pass then runs on the padded width (its tiles no longer straddle lines: 1.08 -> 0.80 ms at level 1) and
 * fdn_irfft_rows takes the pitch as in_row_bins. */
int fdn_rfft_rows(const float* in, float* out_c, long rows, int W, long out_row_bins, fdn_stream_t stream);
/* r2c along rows of the channel LayerNorm of x, normalised on load: x [B][C][H][W], stats [B][2][H*W] = (mean, rstd) of x over C,
 * gamma / beta [C] -> out_c [B*C*H][W/2+1] = rfft(norm(x)) (FDN_arch.py:675 + :411).  Widths with a compile-time plan only
 * (W = 2 * {20, 30} * {32, 16, 8}); else FDN_ERR_UNSUPPORTED: fdn_layernorm_chan + fdn_rfft_rows. */
int fdn_rfft_rows_ln(const float* x, const float* stats, const float* gamma, const float* beta, float* out_c, int B, int C, int H,
                     int W, long out_row_bins, fdn_stream_t stream);
/* c2r along rows: spectrum rows of `in_row_bins` bins (>= W/2+1; leading-slice crop of
 * irfft2(s=(H,W)), FDN_arch.py:147), planes `in_plane_bins` apart -> out [planes][H][W] real,
 * out = scale * c2r(in) + alpha * res  (res may be NULL).  Im of bins 0 and W/2 is ignored. */
int fdn_irfft_rows(const float* in_c, long in_row_bins, long in_plane_bins, float* out, long planes, int H, int W,
                   float scale, const float* res, float alpha, fdn_stream_t stream);
/* Interleave the FCAFFN guidance (x_high = amplitude, xp2 = phase, FDN_arch.py:413-414; both [B][3][H][Wf]) into
 * one 32-byte record per bin: packed [B][H][row_bins][8] = (amp0, amp1, amp2, pha0, pha1, pha2, 0, 0), records Wf .. row_bins-1 of a
 * row zero (row_bins = 0: dense rows of Wf; use the spectrum's pitch, see fdn_rfft_rows).  Done once per level and forward; every
 * encoder block of the level reads it (a column tile's rows become contiguous). */
int fdn_pack_guidance(const float* amp, const float* pha, float* packed, int B, int H, int Wf, long row_bins, fdn_stream_t stream);
/* FCAFFN spectral core, in place on z [B*C][H][Wf] (already row-transformed): column FFT ->
 * replace_denormals -> * conv1_xa(amp) * exp(-i conv1_xp(pha)) -> column iFFT (FDN_arch.py:411-418,
 * SURVEY App. C).  guide = fdn_pack_guidance output; wxa,wxp [C][3].  Unscaled (scale in fdn_irfft_rows). */
int fdn_fft_cols_fcaffn(float* z, const float* guide, const float* wxa, const float* wxp, int B, int C, int H, int Wf,
                        fdn_stream_t stream);
/* Forward column FFT of z [planes][H][Wf] -> |z| and/or angle(z) as real planes (FDN_arch.py:91-92,
 * :140-141, :883-884, :904).  rd_before: replace_denormals first; fix_real: force Im=+0 at the four
 * self-conjugate bins of a real input (what a real-FFT library returns; keeps angle=+pi there). */
int fdn_fft_cols_fwd(const float* z, float* out_abs, float* out_ang, long planes, int H, int Wf, int rd_before,
                     int fix_real, fdn_stream_t stream);
/* mag,pha real planes [planes][Hin][Wfin] -> z = mag*e^{i pha} on the leading (H,Wf) slice ->
 * inverse column FFT -> z_out [planes][H][Wf] (FDN_arch.py:95-98, :144-147). */
int fdn_fft_cols_inv_polar(const float* mag, const float* pha, int Hin, int Wfin, float* z_out, long planes, int H, int Wf,
                           fdn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense convolutions, resampling and small MAR / LPNet helpers.
 * fdn_conv2d: direct conv, groups=1, weight [Cout][Cin][KH][KW] (FDN_arch.py:57,:135,:174-175,
 * :192-193,:196,:704,:720,:731,:804; LPNet_arch.py:49-62,:91).
 *   v = conv + bias; if res && res_before_act: v += res; v = act(v); if res && !res_before_act:
 *   v += res; out = v + post_add   (post_add = 1e-8 for the MAR heads, FDN_arch.py:241,248,255). */
int fdn_conv2d(const float* x, const float* w, const float* bias, const float* res, float* out, int B, int Cin, int H,
               int W, int Cout, int KH, int KW, int stride, int pad, int act, int res_before_act, float post_add,
               fdn_stream_t stream);
/* ABI 14.  Upsample = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) + Conv2d(C, C/2, 3, padding=1, bias=False)
 * (FDN_arch.py:726-734) without the x2 image: the conv's channel contraction commutes with the upsampling, so the caller first forms the
 * nine per-tap products z[(3 dy + dx) Cout + co] = sum_ci w[co][ci][dy][dx] x[ci] at LOW resolution (one fdn_conv1x1 with the weight
 * rearranged to [9 Cout][Cin]) and this entry point sums, for every output pixel, the nine taps' bilinear samples of them (zero outside the
 * x2 image = the conv's padding; source index clamped at the edge as align_corners=False does).
 * z [B][9 Cout][h][w] -> out [B][Cout][2h][2w]. */
int fdn_upconv_gather(const float* z, float* out, int B, int Cout, int h, int w, fdn_stream_t stream);
/* ConvTranspose2d(Cin, Cout, 4, stride=2, padding=1) + act; weight [Cin][Cout][4][4] (FDN_arch.py:194-195). */
int fdn_conv_transpose4x4s2(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int H, int W,
                            int Cout, int act, fdn_stream_t stream);
/* mode FDN_RS_*: bilinear 1/2, bilinear x2 (align_corners=False), nearest 1/2 ([::2,::2]), nearest x2,
 * PixelUnshuffle(r) (FDN_arch.py:199-206,:230-233,:273-274,:719,:730,:875-876). `planes` = input planes. */
int fdn_resample(const float* x, float* out, long planes, int H, int W, int mode, int r, fdn_stream_t stream);
/* fourier_fuse.fpre[1] = Conv2d(n,n,1,padding=1,groups=n): out [B][C][H+2][W+2] (FDN_arch.py:126). */
int fdn_dw1x1_pad1(const float* x, const float* w, const float* bias, float* out, int B, int C, int H, int W,
                   fdn_stream_t stream);
/* AvgPool2d(3,2,1) count_include_pad (LPNet_arch.py:94); AdaptiveAvgPool2d(1) (:71,:99); SE tail
 * relu(y*gate+shortcut) (:75-80). */
int fdn_avgpool3s2(const float* x, float* out, long planes, int H, int W, fdn_stream_t stream);
int fdn_global_avgpool(const float* x, float* out, long planes, long P, fdn_stream_t stream);
int fdn_se_apply(const float* y, const float* gate, const float* shortcut, float* out, long planes, long P,
                 fdn_stream_t stream);
/* ABI 12.  The two per-bin MLPs of a FreBlock / fourier_fuse in ONE launch, in place: mag <- process1(mag), pha <- process2(pha), each
 * Conv2d(C, C, 1) -> LeakyReLU(slope) -> Conv2d(C, C, 1) with bias over the C channels of a spectrum bin (FDN_arch.py:79-94, :127-143).
 * mag, pha [B][C][P] (P bins per plane); w1*, w2* [C][C], b1*, b2* [C].  C in {12, 24, 48}; other widths return FDN_ERR_UNSUPPORTED
 * (four fdn_conv1x1 calls do the same). */
int fdn_spectral_mlp2(float* mag, float* pha, const float* w1m, const float* b1m, const float* w2m, const float* b2m, const float* w1p,
                      const float* b1p, const float* w2p, const float* b2p, int B, int C, long P, float slope, fdn_stream_t stream);
/* x[b] *= ratio[b] (FDN_arch.py:213-219); out = 1-(1-x)^(scale*i_map) (FDN_arch.py:282-284). */
int fdn_scale_batch(float* x, const float* ratio, int B, long per_batch, fdn_stream_t stream);
int fdn_gamma_curve(const float* x, const float* i_map, float* out, float scale, long total, fdn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The step either side of the path, batched.
 * fdn_pre_u8: uint8 HWC images [B][h][w][3] -> fp32 /255 -> CHW -> reflect-pad bottom/right to [B][3][H][W]
 *   (inference_fdn_lolblur.py:47-62, basicsr/utils/img_util.py:9-33).  swap_rb = 1 for BGR input (cv2.imread),
 *   0 when the buffer is already RGB.
 * fdn_post_u8: [B][3][H][W] fp32 -> crop [:h,:w] -> clamp(0,1) -> *255 -> round half-to-even -> uint8 HWC
 *   [B][h][w][3] (inference_fdn_lolblur.py:72-75, basicsr/utils/img_util.py:36-98); swap_rb = 1 writes BGR. */
int fdn_pre_u8(const unsigned char* img, float* out, int B, int h, int w, int H, int W, int swap_rb, fdn_stream_t stream);
int fdn_post_u8(const float* res, unsigned char* out, int B, int h, int w, int H, int W, int swap_rb, fdn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Tiled inference (ImageRestorationModel.grids / grids_inverse,
 * basicsr/models/image_restoration_model.py:261-339, scale = 1).  ij = device int32 [T][2] tile origins (i, j).
 * fdn_tiles_gather: x [C][H][W] -> tiles [T][C][ch][cw].
 * fdn_tiles_merge : tiles [T][C][ch][cw] -> out [C][H][W] = (sum of the tiles covering a pixel, in tile order) / count. */
int fdn_tiles_gather(const float* x, float* tiles, const int* ij, int T, int C, int H, int W, int ch, int cw, fdn_stream_t stream);
int fdn_tiles_merge(const float* tiles, float* out, const int* ij, int T, int C, int H, int W, int ch, int cw, fdn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Validation metrics on the GPU (basicsr/metrics/psnr_ssim.py:8-73, :163-197).
 * fdn_sse_max: out2[0] += sum (a-b)^2 in fp64, out2[1] = max(out2[1], max(a)) over n floats (caller zeroes out2);
 *   PSNR = 20 log10(peak / sqrt(out2[0] / n)), peak = 1 if max <= 1 else 255 (:58-61).
 * fdn_ssim3d: the ssim3d=True path of calculate_ssim on [C][H][W] fp32 images: *out_sum += sum of the SSIM map over
 *   the (H, W, C) volume (mean = sum / (C*H*W)); ws = 10*C*H*W floats of workspace (caller zeroes out_sum). */
int fdn_sse_max(const float* a, const float* b, long n, double* out2, fdn_stream_t stream);
int fdn_ssim3d(const float* a, const float* b, int C, int H, int W, float max_value, float* ws, double* out_sum,
               fdn_stream_t stream);
/* The other branches of the same two functions.
 * fdn_y_channel: test_y_channel=True (psnr_ssim.py:55-57, :275-277 -> metric_util.py:34-47 -> matlab_functions.py:207-238, y_only):
 *   img_bgr [3][H][W] in B, G, R order, range [0, 255] -> out [H][W] = Y of ITU-R BT.601 in [16, 235], with the reference's mix of
 *   float32 / float64 steps.  PSNR on Y = fdn_sse_max of two Y planes.
 * fdn_ssim2d: the 2-D Gaussian SSIM in float64, per channel: replicate_no_crop = 0 is _ssim (:84-116; cv2.filter2D's default
 *   reflect-101 border, map restricted to [5:-5, 5:-5]; mean = sum / (C (H-10) (W-10))), 1 is _ssim_cly (:199-240; BORDER_REPLICATE,
 *   whole map; mean = sum / (C H W), used on the Y plane with max_value 255).  ws = 5*C*H*W doubles (caller zeroes out_sum). */
int fdn_y_channel(const float* img_bgr, float* out, int H, int W, fdn_stream_t stream);
int fdn_ssim2d(const float* a, const float* b, int C, int H, int W, float max_value, int replicate_no_crop, double* ws,
               double* out_sum, fdn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FDN_HIP_H */
