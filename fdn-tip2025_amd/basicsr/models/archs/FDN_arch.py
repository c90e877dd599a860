"""MI355X-native FDN (drop-in for the reference's basicsr/models/archs/FDN_arch.py).

Same module surface as the reference: class `FDN` (no-arg ctor, `forward(inp_img, ori=None,
device=None, ratio_i=None, mode=1)` -> 4-tuple, FDN_arch.py:847-921) and the same parameter tree,
so `load_state_dict(torch.load(path)['params'], strict=True)` of a reference checkpoint (1503
keys) works unchanged.  The nn.Conv2d / Parameter objects below are only weight containers: every
compute step runs in libfdn_hip.so (hand-written gfx950 kernels, include/fdn_hip.h) through
`fdn_hip.ops`.  There is no CPU / eager fallback - tensors must be fp32 on a ROCm device.

Differences from the reference that are deliberate: no hard-coded `fourier_gamma.pth` load in
`FDN.__init__` (FDN_arch.py:860-862; a full checkpoint overwrites net_a anyway), no stray print
(:211), dead classes (AFF, SpaBlock, Mlp, get_p, ...) are not re-created.
"""
import numbers  # noqa: F401  (star-exported by the reference module)

import numpy as np  # noqa: F401
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401  (inference_fdn_lolblur.py uses F.pad from the star import)
from einops import rearrange  # noqa: F401

import fdn_hip
from fdn_hip import ACT_LEAKY, ACT_NONE, ACT_SIGMOID, ops


# ---------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------
_Cache = ops.WeightCache          # derived weights live on their module (stream-safe, see fdn_hip/ops.py)


def _w(p):
    return p.detach()


class WithBias_LayerNorm(nn.Module):
    """Parameter holder for the channel LayerNorm (reference FDN_arch.py:313-329)."""

    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class LayerNorm(nn.Module):
    """LayerNorm over channels of NCHW (reference FDN_arch.py:332-342) -> fdn_layernorm_chan."""

    def __init__(self, dim, LayerNorm_type="WithBias"):
        super().__init__()
        self.body = WithBias_LayerNorm(dim)

    def params(self):
        return _w(self.body.weight), _w(self.body.bias)

    def forward(self, x):
        return ops.layernorm_chan(x, *self.params())


# ---------------------------------------------------------------------------------------------
# FDformer blocks
# ---------------------------------------------------------------------------------------------
class FDSA(nn.Module):
    """Frequency-domain self attention (reference FDN_arch.py:556-641).
    HIP path: MFMA GEMM (to_hidden, LN fused) -> fdn_fdsa_core (dw3x3 + 8x8 FFT recombination)
    -> channel stats -> MFMA GEMM (3 LayerNorms * v_value fused in, residual fused out)."""

    def __init__(self, dim, bias=False):
        super().__init__()
        self.inner = 4
        self.expand_dim = int(dim * 1.2)
        e = self.expand_dim
        self.to_hidden = nn.Conv2d(dim, e * 4, kernel_size=1, bias=bias)
        self.to_hidden_dw = nn.Conv2d(e * 4, e * 4, kernel_size=3, padding=1, groups=e * 4, bias=bias)
        self.project_out = nn.Conv2d(e * 3, dim, kernel_size=1, bias=bias)
        self.norm1 = LayerNorm(e)
        self.norm2 = LayerNorm(e)
        self.norm3 = LayerNorm(e)
        self.patch_size = 8
        self.fft = nn.Parameter(torch.ones((e, 1, 1, 8, 5)))
        self._c = _Cache()

    def fused(self, x, ln=None, res=None, pin=None):
        """pin = (FDFFN module, (gamma, beta) of the LayerNorm in front of it): the sub-block that follows; where the one-launch route can, it also
        produces that FDFFN's project_in output and attaches it to the result (`._fdn_pin`, consumed by FDFFN.fused)."""
        e = self.expand_dim
        norms = (self.norm1, self.norm2, self.norm3)
        gam = self._c.get("g", [n.body.weight for n in norms], lambda: torch.cat([n.body.weight.detach() for n in norms]))
        bet = self._c.get("b", [n.body.bias for n in norms], lambda: torch.cat([n.body.bias.detach() for n in norms]))
        if (ops.FDSA_FULL and x.shape[1] in ops.FDSA_FUSED_C and x.shape[1] <= ops.FDSA_FULL_MAX_C and fdn_hip.matrix_pipe() == "bf16"
                and fdn_hip.storage_dtype() == "f32" and (res is None or res.is_contiguous())):
            # levels 1-2: the whole sub-block in one launch - no hidden tensor and no (out1|out2|out3|v_value) hand-off in HBM
            srcs = [self.to_hidden.weight, self.project_out.weight] + [n.body.weight for n in norms] + [n.body.bias for n in norms] \
                + ([ln[1], ln[2]] if ln is not None else [])
            wfull = self._c.get("full" if ln is not None else "full0", srcs, lambda: ops.fdsa_full_pack(
                _w(self.to_hidden.weight), *((ln[1], ln[2]) if ln is not None else (None, None)), _w(self.project_out.weight), gam, bet))
            if wfull is not None:
                y = ops.fdsa_full(x, ln[0] if ln is not None else None, wfull, _w(self.to_hidden_dw.weight), _w(self.fft), res=res,
                                  want_stats=res is not None)
                if y is not None:
                    return y
        if x.shape[1] in ops.FDSA_FUSED_C and fdn_hip.matrix_pipe() == "bf16":      # levels 1-2: LayerNorm + to_hidden + core in one launch, no hidden tensor in HBM
            srcs = [self.to_hidden.weight] + ([ln[1], ln[2]] if ln is not None else [])
            wpk = self._c.get("pk" if ln is not None else "pk0", srcs, lambda: ops.fdsa_pack(
                _w(self.to_hidden.weight), *((ln[1], ln[2]) if ln is not None else (None, None))))
            if ops.FDSA_TAIL and (res is None or res.is_contiguous()):       # (both storage modes: the tile-local scratch is fp32 either way - no hand-off tensor to store)
                # (round 6) one launch: the workgroup that produced a tile's (out1|out2|out3|v_value) planes runs the tail on them itself
                tsrc = [self.project_out.weight] + [n.body.weight for n in norms] + [n.body.bias for n in norms]
                img, hd = None, 0
                hst = ops.block_storage(x.shape[1], x.shape[2] * x.shape[3], hidden=pin[0].project_in.weight.shape[0]) if pin is not None else torch.float32
                if pin is not None and ops.FDSA_TAIL_PIN and x.shape[1] <= ops.FDSA_TAIL_PIN_MAX_C and res is not None and (hst == torch.float32 or x.shape[1] <= 32):
                    pw = pin[0].project_in.weight
                    img = self._c.get("tlp", tsrc + [pw, pin[1][0], pin[1][1]], lambda: ops.fdsa_tail_pack(
                        _w(self.project_out.weight), gam, bet, x.shape[1], pin=ops.fold_ln(_w(pw), None, pin[1][0], pin[1][1])))
                    hd = pw.shape[0] if img is not None else 0
                if img is None:
                    img = self._c.get("tl", tsrc, lambda: ops.fdsa_tail_pack(_w(self.project_out.weight), gam, bet, x.shape[1]))
                if img is not None:
                    y = ops.fdsa_fused_tail(x, ln[0] if ln is not None else None, wpk, _w(self.to_hidden_dw.weight), _w(self.fft), img,
                                            res=res, want_stats=res is not None, Hd=hd, h_dtype=hst if hd else torch.float32)
                    if y is not None:
                        return y
            o = ops.fdsa_fused(x, ln[0] if ln is not None else None, wpk, _w(self.to_hidden_dw.weight), _w(self.fft),
                               out_dtype=ops.block_storage(x.shape[1], x.shape[2] * x.shape[3]))
        else:
            hidden = ops.conv1x1(x, _w(self.to_hidden.weight), ln=ln, cache=(self._c, "th"))
            o = ops.fdsa_core(hidden, _w(self.to_hidden_dw.weight), _w(self.fft))
        y = ops.fdsa_out(o, _w(self.project_out.weight), gam, bet, res=res, want_stats=res is not None)   # levels 1, 2
        if y is not None:
            return y
        stats = None if ops.GEMM_OWN_STATS else ops.chan_stats(o[:, :3 * e], groups=3)      # None: the GEMM takes them in a pass over its tile
        return ops.conv1x1(o[:, :3 * e], _w(self.project_out.weight), ln3_gate=(stats, gam, bet, o[:, 3 * e:]), res=res,
                           want_stats=res is not None, cache=(self._c, "po"))

    def forward(self, x):
        return self.fused(x)


class FDFFN(nn.Module):
    """Frequency-domain FFN (reference FDN_arch.py:430-475).
    HIP path: GEMM (project_in) -> fdn_fdffn_mid -> fdn_ffn_tail (gate + project_out + residual in one launch)."""

    def __init__(self, dim, bias=False, r=2.7, use_light=True, use_img=True):
        super().__init__()
        hidden = int(r * dim)
        self.space = nn.Sequential(
            nn.Conv2d(hidden, hidden, kernel_size=3, padding=1, groups=hidden, bias=bias),
            nn.GELU(),
            nn.Conv2d(hidden, hidden, kernel_size=3, padding=1, groups=hidden, bias=bias))
        self.patch_size = 8
        self.ffta = nn.Parameter(torch.ones((hidden, 1, 1, 8, 5)))
        self.fftp = nn.Parameter(torch.zeros((hidden, 1, 1, 8, 5)))
        self.dwconv = nn.Conv2d(hidden, hidden * 2, kernel_size=3, padding=1, groups=hidden, bias=bias)
        self.project_in = nn.Conv2d(dim, hidden, kernel_size=1, bias=bias)
        self.project_out = nn.Conv2d(hidden, dim, kernel_size=1, bias=bias)
        self._c = _Cache()

    def fused(self, x, ln=None, res=None):
        st = ops.block_storage(x.shape[1], x.shape[2] * x.shape[3], hidden=self.project_in.weight.shape[0])      # hidden tensors: fp32, or bf16 storage (levels 1-2 in bf16 mode)
        h = x.__dict__.pop("_fdn_pin", None) if hasattr(x, "__dict__") else None      # (round 6) the FDSA launch in front already ran project_in(LN(x)) on its registers
        if h is None or ln is None:
            h = ops.conv1x1(x, _w(self.project_in.weight), ln=ln, cache=(self._c, "pi"), out_dtype=st)
        y = ops.fdffn_mid(h, _w(self.space[0].weight), _w(self.space[2].weight), _w(self.ffta), _w(self.fftp))
        return ops.ffn_tail(y, _w(self.dwconv.weight), _w(self.project_out.weight), res=res, want_stats=res is not None,
                            cache=(self._c, "po"))

    def forward(self, x, x_high=None, xp2=None, x_img=None):
        return self.fused(x)


class FCAFFN(nn.Module):
    """Fourier cross-attention FFN of the encoder blocks (reference FDN_arch.py:381-429).
    HIP path: rows r2c -> fdn_fft_cols_fcaffn (column FFT, modulation, column iFFT in one launch)
    -> rows c2r -> GEMM with LN(x)*x1+x1 prologue and x*mul+add epilogue -> gate -> GEMM."""

    def __init__(self, dim, bias=False, r=1.0, use_light=True, use_img=True):
        super().__init__()
        hidden = int(r * dim)
        self.project_in = nn.Conv2d(dim, hidden, kernel_size=1, bias=bias)
        self.project_out = nn.Conv2d(dim, hidden, kernel_size=1, bias=bias)
        self.conv1_xa = nn.Conv2d(3, hidden, kernel_size=1, bias=bias)
        self.conv1_xp = nn.Conv2d(3, hidden, kernel_size=1, bias=bias)
        self.conv1_add = nn.Conv2d(3, hidden, kernel_size=1, bias=bias)
        self.conv1_mul = nn.Conv2d(3, hidden, kernel_size=1, bias=bias)
        self.conv3_add = nn.Conv2d(hidden, hidden, kernel_size=3, padding=1, groups=hidden, bias=bias)
        self.conv3_mul = nn.Conv2d(hidden, hidden, kernel_size=3, padding=1, groups=hidden, bias=bias)
        self.norm = LayerNorm(hidden)
        self.dwconv = nn.Conv2d(hidden, hidden * 2, kernel_size=3, padding=1, groups=hidden, bias=bias)
        self._c = _Cache()

    def fused(self, xn, x_high, xp2, x_img, res=None, ln=None):
        """xn: the block input (already norm3-normalised), kept as x1 (FDN_arch.py:410).  ln = (stats, gamma, beta): xn is the
        UN-normalised input and norm3 is applied on load by the row FFT and by fdn_fcaffn_in (no normalised copy)."""
        _, _, h, w = xn.shape
        wp = ops.spec_pitch(w // 2 + 1)            # spectrum rows padded to whole 128-byte lines for the column pass
        z = ops.rfft_rows_ln(xn, *ln, pitch=wp) if ln is not None else ops.rfft_rows(xn, pitch=wp)
        ops.fft_cols_fcaffn(z, x_high, xp2, _w(self.conv1_xa.weight), _w(self.conv1_xp.weight))
        xi = ops.irfft_rows(z, h, w, 2.0 / (h * w))
        gam, bet = self.norm.params()
        if xi.shape[1] in ops.FCAFFN_IN_C and w % 2 == 0:
            t = ops.fcaffn_in(xi, xn, x_img, _w(self.project_in.weight), gam, bet, _w(self.conv1_mul.weight),
                              _w(self.conv3_mul.weight), _w(self.conv1_add.weight), _w(self.conv3_add.weight), x1_ln=ln)
        elif xi.shape[1] >= ops.FCAFFN_PACKED_MIN_C and fdn_hip.matrix_pipe() == "bf16":          # level 3: the same sub-block on the split-bf16 GEMM, one launch
            srcs = [self.project_in.weight, self.conv1_mul.weight, self.conv3_mul.weight, self.conv1_add.weight, self.conv3_add.weight]
            wpk = self._c.get("fcpk", srcs, lambda: ops.fcaffn_in_pack(*[_w(p) for p in srcs]))
            t = ops.fcaffn_in_packed(xi, None if ops.GEMM_OWN_STATS else ops.chan_stats(xi), xn, x_img, wpk, gam, bet, x1_ln=ln)
        else:
            stats = ops.chan_stats(xi)
            mul, add = ops.img_mod_maps(x_img, _w(self.conv1_mul.weight), _w(self.conv3_mul.weight),
                                        _w(self.conv1_add.weight), _w(self.conv3_add.weight))
            t = ops.conv1x1(xi, _w(self.project_in.weight), ln_muladd=(stats, gam, bet, xn), muladd=(mul, add), cache=(self._c, "pi"))
        return ops.ffn_tail(t, _w(self.dwconv.weight), _w(self.project_out.weight), res=res, want_stats=res is not None,
                            cache=(self._c, "po"))

    def forward(self, x, x_high, xp2, x_img=None):
        return self.fused(x, x_high, xp2, x_img)


class TransformerBlock(nn.Module):
    """Reference FDN_arch.py:646-677; tuple in / tuple out (x, x_high, x_p, x_img)."""

    def __init__(self, dim, ffn_expansion_factor=2.66, mode=1, bias=False, LayerNorm_type="WithBias", att=False,
                 use_light=True, use_img=True):
        super().__init__()
        self.use_light = use_light
        self.att = att
        if att:
            self.norm1 = LayerNorm(dim)
            self.attn = FDSA(dim, bias)
        self.norm2 = LayerNorm(dim)
        self.ffn = FDFFN(dim, bias, use_light=use_light, use_img=use_img)
        if use_light:
            self.norm3 = LayerNorm(dim)
            self.ffn2 = FCAFFN(dim, bias, use_light=use_light, use_img=use_img)

    def forward(self, xt):
        x, x_high, x_p, x_img = xt
        if self.att:
            x = self.attn.fused(x, ln=(ops.stats_of(x),) + self.norm1.params(), res=x, pin=(self.ffn, self.norm2.params()))
        x = self.ffn.fused(x, ln=(ops.stats_of(x),) + self.norm2.params(), res=x)
        if self.use_light:
            if (x.shape[1] in ops.FCAFFN_IN_C or (x.shape[1] >= ops.FCAFFN_PACKED_MIN_C and fdn_hip.matrix_pipe() == "bf16")) and ops.rows_ln_ok(x):      # norm3 on load: no normalised copy of x
                x = self.ffn2.fused(x, x_high, x_p, x_img, res=x, ln=(ops.stats_of(x),) + self.norm3.params())
            else:
                x = self.ffn2.fused(self.norm3(x), x_high, x_p, x_img, res=x)
        return x, x_high, x_p, x_img


class Fuse(nn.Module):
    """Reference FDN_arch.py:679-695.  conv2 + split + add is one GEMM with folded weights
    (e + d = (W_e + W_d) x + (b_e + b_d))."""

    def __init__(self, n_feat):
        super().__init__()
        self.n_feat = n_feat
        self.att_channel = TransformerBlock(dim=n_feat * 2, use_light=False, use_img=False)
        self.conv = nn.Conv2d(n_feat * 2, n_feat * 2, 1, 1, 0)
        self.conv2 = nn.Conv2d(n_feat * 2, n_feat * 2, 1, 1, 0)
        self._c = _Cache()

    def forward(self, enc, dnc, x_high=None, x_high_p=None, x_img=None):
        n = self.n_feat
        x = ops.conv1x1([enc, dnc], _w(self.conv.weight), _w(self.conv.bias), want_stats=True, cache=(self._c, "c1"))   # (norm2 of att_channel reads the statistics: no fdn_chan_stats pass)
        x = self.att_channel((x, x_high, x_high_p, x_img))[0]
        wf = self._c.get("w", [self.conv2.weight], lambda: self.conv2.weight.detach()[:n] + self.conv2.weight.detach()[n:])
        bf = self._c.get("b", [self.conv2.bias], lambda: self.conv2.bias.detach()[:n] + self.conv2.bias.detach()[n:])
        return ops.conv1x1(x, wf, bf, cache=(self._c, "c2"), want_stats=True)                        # (norm1 of the next decoder block)


class OverlapPatchEmbed(nn.Module):
    def __init__(self, in_c=3, embed_dim=48, bias=False):
        super().__init__()
        self.proj = nn.Conv2d(in_c, embed_dim, kernel_size=3, stride=1, padding=1, bias=bias)

    def forward(self, x):
        return ops.conv2d(x, _w(self.proj.weight), pad=1)


class Downsample(nn.Module):
    """bilinear 1/2 then 3x3 conv C->2C (reference FDN_arch.py:715-723)."""

    def __init__(self, n_feat):
        super().__init__()
        self.body = nn.Sequential(nn.Identity(), nn.Conv2d(n_feat, n_feat * 2, 3, stride=1, padding=1, bias=False))

    def forward(self, x):
        return ops.conv2d(ops.resample(x, ops.RS_BILINEAR_HALF), _w(self.body[1].weight), pad=1)


class Upsample(nn.Module):
    """bilinear x2 then 3x3 conv C->C/2 (reference FDN_arch.py:726-734)."""

    def __init__(self, n_feat):
        super().__init__()
        self.body = nn.Sequential(nn.Identity(), nn.Conv2d(n_feat, n_feat // 2, 3, stride=1, padding=1, bias=False))
        self._c = _Cache()

    def forward(self, x):
        if ops.UPCONV_GATHER:               # the channel contraction first, at low resolution; the x2 image is never formed
            return ops.upsample_conv3x3(x, _w(self.body[1].weight), cache=(self._c, "up"))
        return ops.conv2d(ops.resample(x, ops.RS_BILINEAR_X2), _w(self.body[1].weight), pad=1)


class FDformer(nn.Module):
    """U-shaped Fourier transformer (reference FDN_arch.py:753-842)."""

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[6, 6, 12, 8], num_refinement_blocks=4,
                 ffn_expansion_factor=3, bias=False):
        super().__init__()

        def stage(d, n, enc):
            return nn.Sequential(*[TransformerBlock(dim=d, ffn_expansion_factor=ffn_expansion_factor, bias=bias, att=True,
                                                    mode=1 if enc else 2, use_light=enc, use_img=enc) for _ in range(n)])

        self.patch_embed = OverlapPatchEmbed(inp_channels, dim)
        self.encoder_level1 = stage(dim, num_blocks[0], True)
        self.down1_2 = Downsample(dim)
        self.encoder_level2 = stage(dim * 2, num_blocks[1], True)
        self.down2_3 = Downsample(dim * 2)
        self.encoder_level3 = stage(dim * 4, num_blocks[2], True)
        self.decoder_level3 = stage(dim * 4, num_blocks[2], False)
        self.up3_2 = Upsample(dim * 4)
        self.reduce_chan_level2 = nn.Conv2d(dim * 4, dim * 2, kernel_size=1, bias=bias)   # in the checkpoint, never called
        self.decoder_level2 = stage(dim * 2, num_blocks[1], False)
        self.up2_1 = Upsample(dim * 2)
        self.decoder_level1 = stage(dim, num_blocks[0], False)
        self.refinement = stage(dim, num_refinement_blocks, False)
        self.fuse2 = Fuse(dim * 2)
        self.fuse1 = Fuse(dim)
        self.output = nn.Conv2d(dim, out_channels, kernel_size=3, stride=1, padding=1, bias=bias)
        self.norm = LayerNorm(3)                                                         # in the checkpoint, never called

    def forward(self, inp_img, ori_img=None, x_high1=None, x_high2=None, x_high3=None, x_high12=None, x_high22=None,
                x_high32=None, x1=None, x2=None, x3=None):
        e1 = self.encoder_level1((self.patch_embed(inp_img), x_high1, x_high12, x1))[0]
        e2 = self.encoder_level2((self.down1_2(e1), x_high2, x_high22, x2))[0]
        e3 = self.encoder_level3((self.down2_3(e2), x_high3, x_high32, x3))[0]
        d3 = self.decoder_level3((e3, x_high3, x_high32, x3))[0]
        d2 = self.fuse2(self.up3_2(d3), e2, x_high2, x_high22, x2)
        d2 = self.decoder_level2((d2, x_high2, x_high22, x2))[0]
        d1 = self.fuse1(self.up2_1(d2), e1, x_high1, x_high12, x1)
        d1 = self.decoder_level1((d1, x_high1, x_high12, x1))[0]
        d1 = self.refinement((d1, x_high1, x_high12, x1))[0]
        res = inp_img if ori_img is None else ori_img
        return ops.conv2d(d1, _w(self.output.weight), pad=1, res=res)


# ---------------------------------------------------------------------------------------------
# MAR (amplitude / gamma-curve pre-net, reference FDN_arch.py:16-286)
# ---------------------------------------------------------------------------------------------
class BasicConv(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, stride, bias=True, relu=True, transpose=False):
        super().__init__()
        self.k, self.stride, self.relu, self.transpose = kernel_size, stride, relu, transpose
        if transpose:
            conv = nn.ConvTranspose2d(in_channel, out_channel, kernel_size, padding=kernel_size // 2 - 1, stride=stride, bias=bias)
        else:
            conv = nn.Conv2d(in_channel, out_channel, kernel_size, padding=kernel_size // 2, stride=stride, bias=bias)
        self.main = nn.Sequential(conv)

    def forward(self, x, res=None, res_before_act=False, act=None, post_add=0.0):
        c = self.main[0]
        a = (ACT_LEAKY if self.relu else ACT_NONE) if act is None else act
        if self.transpose:
            return ops.conv_transpose4x4s2(x, _w(c.weight), _w(c.bias), a)
        if self.k == 1 and self.stride == 1:
            return ops.conv1x1(x, _w(c.weight), _w(c.bias), act=a)
        return ops.conv2d(x, _w(c.weight), _w(c.bias), stride=self.stride, pad=self.k // 2, act=a, res=res,
                          res_before_act=res_before_act, post_add=post_add)


class FAM(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.merge1 = nn.Conv2d(channel * 2, channel, kernel_size=1)
        self.merge2 = nn.Conv2d(channel, channel, kernel_size=3, stride=1, padding=1)

    def forward(self, x1, x2):
        t = ops.conv1x1([x1, x2], _w(self.merge1.weight), _w(self.merge1.bias))
        return ops.conv2d(t, _w(self.merge2.weight), _w(self.merge2.bias), pad=1)


def _mlp2(seq, x):
    t = ops.conv1x1(x, _w(seq[0].weight), _w(seq[0].bias), act=ACT_LEAKY)
    return ops.conv1x1(t, _w(seq[2].weight), _w(seq[2].bias))


def _spectral_mlps(y, process1, process2, H, W):
    """rfft2 -> (|.|, angle) -> per-bin 1x1 MLPs -> polar -> column iFFT (FDN_arch.py:90-97)."""
    z = ops.rfft_rows(y)
    mag, pha = ops.fft_cols_fwd(z, True, True, rd_before=False, fix_real=True)
    if ops.SPECTRAL_MLP_FUSED and mag.shape[1] in ops.SPECTRAL_MLP_C:              # both MLPs in one launch, in place (the four 1x1 convs read and write 4 C planes each)
        ops.spectral_mlp2(mag, pha, _w(process1[0].weight), _w(process1[0].bias), _w(process1[2].weight), _w(process1[2].bias),
                          _w(process2[0].weight), _w(process2[0].bias), _w(process2[2].weight), _w(process2[2].bias), slope=0.1)
    else:
        mag = _mlp2(process1, mag)
        pha = _mlp2(process2, pha)
    return ops.fft_cols_inv_polar(mag, pha, H, W // 2 + 1)


class FreBlock(nn.Module):
    def __init__(self, nc):
        super().__init__()
        self.fpre = nn.Conv2d(nc, nc, 1, 1, 0)
        self.process1 = nn.Sequential(nn.Conv2d(nc, nc, 1, 1, 0), nn.LeakyReLU(0.1), nn.Conv2d(nc, nc, 1, 1, 0))
        self.process2 = nn.Sequential(nn.Conv2d(nc, nc, 1, 1, 0), nn.LeakyReLU(0.1), nn.Conv2d(nc, nc, 1, 1, 0))

    def forward(self, x, skip_gain=1.0):
        """irfft2(...) + skip_gain * x  (skip_gain=1: FreBlock alone, :100; 2: inside ProcessBlock, :118)."""
        _, _, H, W = x.shape
        y = ops.conv1x1(x, _w(self.fpre.weight), _w(self.fpre.bias))
        zz = _spectral_mlps(y, self.process1, self.process2, H, W)
        return ops.irfft_rows(zz, H, W, 2.0 / (H * W), res=x, alpha=skip_gain)


class ProcessBlock(nn.Module):
    def __init__(self, in_nc, spatial=False):
        super().__init__()
        self.frequency_process = FreBlock(in_nc)
        self.cat = nn.Conv2d(in_nc, in_nc, 1, 1, 0)      # in the checkpoint, never called (spatial=False, :117-118)

    def forward(self, x):
        return self.frequency_process(x, skip_gain=2.0)


class fourier_fuse(nn.Module):
    def __init__(self, in_nc, out_nc):
        super().__init__()
        self.fpre = nn.Sequential(nn.Conv2d(in_nc, out_nc, 1, 1), nn.Conv2d(out_nc, out_nc, 1, 1, 1, groups=out_nc))
        self.process1 = nn.Sequential(nn.Conv2d(out_nc, out_nc, 1, 1, 0), nn.LeakyReLU(0.1), nn.Conv2d(out_nc, out_nc, 1, 1, 0))
        self.process2 = nn.Sequential(nn.Conv2d(out_nc, out_nc, 1, 1, 0), nn.LeakyReLU(0.1), nn.Conv2d(out_nc, out_nc, 1, 1, 0))
        self.fourier_out = nn.Conv2d(out_nc, out_nc, 3, 1, 1)
        self._c = _Cache()

    def forward(self, x1, x2, x4):
        return self._rest(ops.conv1x1([x1, x2, x4], _w(self.fpre[0].weight), _w(self.fpre[0].bias)))

    def forward_multires(self, same, up1, up2):
        """forward(cat(same), nearest_x2(up1), nearest_x4(up2)) without the resized tensors (round 5): fpre[0] is a 1x1 conv and nearest
        replication commutes with it, so every source is contracted at ITS OWN resolution and the (narrow) partial sums are replicated:
        W [same | up1 | up2] = W_s same + x2(W_1 up1 + x2(W_2 up2)).  MAR_archa.forward's z21 / z41 / z42 (FDN_arch.py:231-236: 72 + 48 planes
        at the finer levels) are never formed; the sums differ from the reference's single 84-term dot product in association only."""
        wfull, bias = _w(self.fpre[0].weight), _w(self.fpre[0].bias)
        n = wfull.shape[0]
        ks = sum(t.shape[1] for t in same)
        k1 = 0 if up1 is None else up1.shape[1]
        w2 = wfull.reshape(n, -1)
        parts = self._c.get("split", [self.fpre[0].weight], lambda: (w2[:, :ks].contiguous(), w2[:, ks:ks + k1].contiguous(), w2[:, ks + k1:].contiguous()))
        acc = ops.conv1x1(up2, parts[2])                                               # two levels down
        if up1 is not None:
            acc = ops.conv1x1(up1, parts[1], res=ops.resample(acc, ops.RS_NEAREST_X2))   # one level down
        return self._rest(ops.conv1x1(same, parts[0], bias, res=ops.resample(acc, ops.RS_NEAREST_X2)))

    def _rest(self, y):
        H, W = y.shape[-2:]
        y = ops.dw1x1_pad1(y, _w(self.fpre[1].weight), _w(self.fpre[1].bias))          # (H+2) x (W+2), :126
        zz = _spectral_mlps(y, self.process1, self.process2, H, W)                      # leading-slice crop, :147
        xo = ops.irfft_rows(zz, H, W, 2.0 / (H * W))
        return ops.conv2d(xo, _w(self.fourier_out.weight), _w(self.fourier_out.bias), pad=1)


class MAR_archa(nn.Module):
    def __init__(self, use_ratio=True, block=None, apply_ratio=True):
        """block: the ProcessBlock class (the LOL-v1 variant, fdnlol24_arch.py, has a live `.cat` conv).
        apply_ratio: the reference multiplies by `ratio` unconditionally here (FDN_arch.py:213-219, `use_ratio`
        is ignored) and conditionally in the LOL-v1 file (fdnlol24_arch.py:160-169)."""
        super().__init__()
        ProcessBlock_ = ProcessBlock if block is None else block
        self.use_ratio = use_ratio
        self.apply_ratio = apply_ratio
        c = 12
        self.Encoder = nn.ModuleList([ProcessBlock_(c), ProcessBlock_(c * 2), ProcessBlock_(c * 4)])
        self.Decoder = nn.ModuleList([ProcessBlock_(c * 4), ProcessBlock_(c * 2), ProcessBlock_(c)])
        self.Convs = nn.ModuleList([BasicConv(c * 4, c * 2, 1, 1, relu=True), BasicConv(c * 2, c, 1, 1, relu=True)])
        self.ConvsOut = nn.ModuleList([BasicConv(c * 4, 3, 3, 1, relu=False), BasicConv(c * 2, 3, 3, 1, relu=False)])
        self.AFFs = nn.ModuleList([fourier_fuse(c * 7, c), fourier_fuse(c * 7, c * 2)])
        self.FAM1 = FAM(c * 4)
        self.f1 = nn.Sequential(nn.Conv2d(3 * 16, c * 4, 1, 1, 0), ProcessBlock_(c * 4))
        self.f2 = nn.Sequential(nn.Conv2d(3 * 4, c * 2, 1, 1, 0), ProcessBlock_(c * 2))
        self.f3 = nn.Sequential(nn.Conv2d(3, c, 1, 1, 0), ProcessBlock_(c))
        self.f3_down = BasicConv(c, c * 2, 3, 2, relu=True)
        self.f2_down = BasicConv(c * 2, c * 4, 3, 2, relu=True)
        self.f2_up = BasicConv(c * 4, c * 2, 4, 2, relu=True, transpose=True)
        self.f3_up = BasicConv(c * 2, c, 4, 2, relu=True, transpose=True)
        self.out = BasicConv(c, 3, 3, 1, relu=False)
        self.FAM2 = FAM(c * 2)

    def _stem(self, seq, x, ratio):
        t = seq[1](ops.conv1x1(x, _w(seq[0].weight), _w(seq[0].bias)))
        return ops.scale_batch_(t, ratio) if self.apply_ratio else t

    def forward(self, x, ratio):
        """ratio: flat [B] tensor; always applied (FDN_arch.py:213-219)."""
        x_2 = ops.resample(x, ops.RS_NEAREST_HALF)
        x_4 = ops.resample(x_2, ops.RS_NEAREST_HALF)
        z2 = self._stem(self.f2, ops.resample(x, ops.RS_PIXEL_UNSHUFFLE, 2), ratio)
        z4 = self._stem(self.f1, ops.resample(x, ops.RS_PIXEL_UNSHUFFLE, 4), ratio)
        x_ = self._stem(self.f3, x, ratio)
        res1 = self.Encoder[0](x_)
        res2 = self.Encoder[1](self.FAM2(self.f3_down(res1), z2))
        z = self.Encoder[2](self.FAM1(self.f2_down(res2), z4))
        z12 = ops.resample(res1, ops.RS_NEAREST_HALF)
        if ops.AFF_MULTIRES:                 # the 1x1 conv of each fourier_fuse applied per source resolution: no z21 / z42 / z41
            res2n = self.AFFs[1].forward_multires([z12, res2], None, z)
            res1 = self.AFFs[0].forward_multires([res1], res2, z)
            res2 = res2n
        else:
            z21 = ops.resample(res2, ops.RS_NEAREST_X2)
            z42 = ops.resample(z, ops.RS_NEAREST_X2)
            z41 = ops.resample(z42, ops.RS_NEAREST_X2)
            res2 = self.AFFs[1](z12, res2, z42)
            res1 = self.AFFs[0](res1, z21, z41)
        z = self.Decoder[0](z)
        o4 = self.ConvsOut[0](z, res=x_4, res_before_act=True, act=ACT_SIGMOID, post_add=1e-8)
        z = self.f2_up(z)
        c0 = self.Convs[0].main[0]
        z = self.Decoder[1](ops.conv1x1([z, res2], _w(c0.weight), _w(c0.bias), act=ACT_LEAKY))
        o2 = self.ConvsOut[1](z, res=x_2, res_before_act=True, act=ACT_SIGMOID, post_add=1e-8)
        z = self.f3_up(z)
        c1 = self.Convs[1].main[0]
        z = self.Decoder[2](ops.conv1x1([z, res1], _w(c1.weight), _w(c1.bias), act=ACT_LEAKY))
        o1 = self.out(z, res=x, res_before_act=True, act=ACT_SIGMOID, post_add=1e-8)
        return [o4, o2, o1]


class MAR(nn.Module):
    def __init__(self, use_ratio=True):
        super().__init__()
        self.net = MAR_archa(use_ratio=True)
        self.scale = 40.0
        self.use_ratio = use_ratio

    def forward(self, x, ratio=None):
        ratio = ratio.reshape(-1).contiguous()
        x1 = x
        x2 = ops.resample(x1, ops.RS_BILINEAR_HALF)
        x3 = ops.resample(x2, ops.RS_BILINEAR_HALF)
        i3, i2, i1 = self.net(x, ratio)
        return ops.gamma_curve(x3, i3, self.scale), ops.gamma_curve(x2, i2, self.scale), ops.gamma_curve(x1, i1, self.scale)


# ---------------------------------------------------------------------------------------------
# FDN
# ---------------------------------------------------------------------------------------------
class FDN(nn.Module):
    """FDN = MAR + FDformer + 3 LayerNorm(3) (reference FDN_arch.py:847-921)."""

    def __init__(self):
        super().__init__()
        self.net_a = MAR(use_ratio=True)
        self.net_p = FDformer(inp_channels=3, out_channels=3, dim=32, num_blocks=[6, 6, 10], num_refinement_blocks=4,
                              ffn_expansion_factor=3, bias=False)
        for p in self.net_a.parameters():
            p.requires_grad = False
        self.norm1 = LayerNorm(3)
        self.norm2 = LayerNorm(3)
        self.norm3 = LayerNorm(3)

    @staticmethod
    def _spectrum(x, want_abs, want_ang, rd):
        return ops.fft_cols_fwd(ops.rfft_rows(x), want_abs, want_ang, rd_before=rd, fix_real=True)

    def forward(self, inp_img, ori=None, device=None, ratio_i=None, mode=1):
        if ratio_i is None:
            raise ValueError("FDN.forward needs ratio_i of shape (B, 1) (reference FDN_arch.py:872)")
        inp_img = inp_img.contiguous()
        ratio = ratio_i.reshape(-1).to(torch.float32).contiguous()
        norms = (self.norm1, self.norm2, self.norm3)
        # phase guidance from the input pyramid (:874-892)
        p1 = inp_img
        p2 = ops.resample(p1, ops.RS_BILINEAR_HALF)
        p3 = ops.resample(p2, ops.RS_BILINEAR_HALF)
        phas = [self._spectrum(n(p), False, True, True)[1] for n, p in zip(norms, (p1, p2, p3))]
        # amplitude guidance from the MAR outputs (:895-914)
        q3, q2, q1 = self.net_a(inp_img, ratio)
        amps = [self._spectrum(n(q), True, False, False)[0] for n, q in zip(norms, (q1, q2, q3))]
        out = self.net_p(inp_img, ori_img=inp_img, x_high1=amps[0], x_high2=amps[1], x_high3=amps[2],
                         x_high12=phas[0], x_high22=phas[1], x_high32=phas[2], x1=q1, x2=q2, x3=q3)
        return out, q1, q2, q3
