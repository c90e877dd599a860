"""CPU oracle for the FDN inference hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product path (fdn-tip2025_amd/) never routes through it and has no CPU fallback.

What it is: a functional (no nn.Module) PyTorch-CPU restatement of the reference's
`LPNet -> FDN` forward, written against the flat checkpoint key layout (1503 keys for FDN,
292 for LPNet) so the very state dict the reference produces can be fed in.  Every function
cites the reference lines it follows (paths relative to /root/reference).  It is dtype-generic:
run it in float32 to mimic the reference, or in float64 to get the "truth" used by the
conditioning-aware tolerance of tests/ (the reference itself cannot run in fp64: hard
`.float()` casts, FDN_arch.py:411,460,585-589).

Parity pinning: the reference has no tests or golden vectors of its own.
This oracle is pinned against outputs of the reference itself, imported in the build
container by tests/golden/make_golden.py, stored as small fixtures in tests/golden/*.npz and
checked by tests/test_oracle_golden.py.  The trained FDN checkpoint and the LOL-Blur data are
absent from the reference checkout, so trained-checkpoint PSNR is UNPINNED.

Third-party arithmetic the reference delegates to and that is used here as-is: torch.fft
(rfft2/irfft2, norm='backward'), F.conv2d / conv_transpose2d, F.gelu (erf form).
"""
import math

import torch
import torch.nn.functional as F

PATCH = 8  # FDN_arch.py:442,571


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------
def replace_denormals(z, thr=1e-10):
    """FDN_arch.py:548-553 -- real and imag separately; (-thr, thr) incl. +-0 -> +thr."""
    re, im = z.real, z.imag
    re = torch.where((re < thr) & (re > -thr), torch.full_like(re, thr), re)
    im = torch.where((im < thr) & (im > -thr), torch.full_like(im, thr), im)
    return torch.complex(re, im)


def ln_chan(x, w, b, eps=1e-5):
    """WithBias_LayerNorm over the channel axis of NCHW (FDN_arch.py:313-342)."""
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)  # biased, :328
    return (x - mu) / torch.sqrt(var + eps) * w.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def _ln(x, P, pfx):
    return ln_chan(x, P[pfx + ".body.weight"], P[pfx + ".body.bias"])


def to_patches(x):
    """'b c (h p1) (w p2) -> b c h w p1 p2' (FDN_arch.py:458,579-584)."""
    b, c, H, W = x.shape
    return x.view(b, c, H // PATCH, PATCH, W // PATCH, PATCH).permute(0, 1, 2, 4, 3, 5)


def from_patches(x):
    """'b c h w p1 p2 -> b c (h p1) (w p2)' (FDN_arch.py:470,621-632)."""
    b, c, h, w, p1, p2 = x.shape
    return x.permute(0, 1, 2, 4, 3, 5).reshape(b, c, h * p1, w * p2)


def polar(a, p):
    return torch.complex(a * torch.cos(p), a * torch.sin(p))


def bilinear_half(x):
    """nn.Upsample(0.5, bilinear, align_corners=False) == exact 2x2 mean (FDN_arch.py:719,866)."""
    return F.interpolate(x, scale_factor=0.5, mode="bilinear", align_corners=False)


def bilinear_x2(x):
    """nn.Upsample(2, bilinear, align_corners=False) (FDN_arch.py:730)."""
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


# --------------------------------------------------------------------------------------
# FDformer blocks
# --------------------------------------------------------------------------------------
def fdsa(x, P, pfx, taps=None):
    """Frequency-domain self attention, FDN_arch.py:575-641.  taps: optional dict that receives the inputs of
    norm1/2/3 (:633-635) as 'o1','o2','o3' and v_value as 'vv' (what a forward hook on the reference sees)."""
    w_h = P[pfx + ".to_hidden.weight"]
    hidden = F.conv2d(x, w_h)                                                    # :576
    hidden = F.conv2d(hidden, P[pfx + ".to_hidden_dw.weight"], padding=1,
                      groups=w_h.shape[0])                                       # :578
    q, k, v, vv = hidden.chunk(4, dim=1)
    qf = torch.fft.rfft2(to_patches(q))                                          # :585
    kf = torch.fft.rfft2(to_patches(k))                                          # :587
    vf = torch.fft.rfft2(to_patches(v))                                          # :589
    vf = replace_denormals(vf * P[pfx + ".fft"])                                 # :591-593
    qk = replace_denormals(qf * kf)                                              # :595-597
    qka = qk.abs()                                                               # :599
    v_a, v_p = vf.abs(), vf.angle()                                              # :601-602
    qkp = replace_denormals(qf).angle() - replace_denormals(kf).angle()          # :603-607
    s = (PATCH, PATCH)
    o1 = from_patches(torch.fft.irfft2(polar(v_a, qkp), s=s))                    # :609-614
    o2 = from_patches(torch.fft.irfft2(polar(qka, v_p), s=s))                    # :617-620
    o3 = from_patches(torch.fft.irfft2(polar(qka, qkp), s=s))                    # :627-630
    if taps is not None:
        taps.update(o1=o1, o2=o2, o3=o3, vv=vv)
    o1 = _ln(o1, P, pfx + ".norm1") * vv                                         # :633,636
    o2 = _ln(o2, P, pfx + ".norm2") * vv
    o3 = _ln(o3, P, pfx + ".norm3") * vv
    return F.conv2d(torch.cat([o1, o2, o3], dim=1), P[pfx + ".project_out.weight"])  # :639


def fdffn(x, P, pfx, taps=None):
    """Frequency-domain feed-forward, FDN_arch.py:453-475 (x_high/xp2/x_img are ignored there).
    taps: optional dict that receives the input of `dwconv` (:472) as 'mid'."""
    x = F.conv2d(x, P[pfx + ".project_in.weight"])                               # :456
    hd = x.shape[1]
    s = F.conv2d(x, P[pfx + ".space.0.weight"], padding=1, groups=hd)
    s = F.conv2d(F.gelu(s), P[pfx + ".space.2.weight"], padding=1, groups=hd)    # :457
    z = replace_denormals(torch.fft.rfft2(to_patches(x)))                        # :458-461
    z = polar(z.abs() * P[pfx + ".ffta"], z.angle() - P[pfx + ".fftp"])          # :462-468
    x = from_patches(torch.fft.irfft2(z, s=(PATCH, PATCH))) + s                  # :469-470
    if taps is not None:
        taps.update(mid=x)
    x1, x2 = F.conv2d(x, P[pfx + ".dwconv.weight"], padding=1, groups=hd).chunk(2, dim=1)  # :472
    return F.conv2d(F.gelu(x1) * x2, P[pfx + ".project_out.weight"])             # :473-474


def fcaffn(x, amp, pha, img, P, pfx, taps=None):
    """Fourier cross-attention FFN (encoder blocks only), FDN_arch.py:405-429.
    taps: optional dict that receives the input of `norm` (:420), i.e. the irfft2 result, as 'xi'."""
    h, w = x.shape[-2:]
    x1 = x
    z = replace_denormals(torch.fft.rfft2(x))                                    # :411-412
    z_p = z.angle() - F.conv2d(pha, P[pfx + ".conv1_xp.weight"])                 # :413
    z_a = F.conv2d(amp, P[pfx + ".conv1_xa.weight"]) * z.abs()                   # :414-415
    x = torch.fft.irfft2(polar(z_a, z_p), s=(h, w))                              # :417-418
    if taps is not None:
        taps.update(xi=x)
    x = _ln(x, P, pfx + ".norm") * x1 + x1                                       # :420
    x = F.conv2d(x, P[pfx + ".project_in.weight"])                               # :421
    c = x.shape[1]
    mul = F.conv2d(F.conv2d(img, P[pfx + ".conv1_mul.weight"]), P[pfx + ".conv3_mul.weight"],
                   padding=1, groups=c)
    add = F.conv2d(F.conv2d(img, P[pfx + ".conv1_add.weight"]), P[pfx + ".conv3_add.weight"],
                   padding=1, groups=c)
    x = x * mul + add                                                            # :423
    a, g = F.conv2d(x, P[pfx + ".dwconv.weight"], padding=1, groups=c).chunk(2, dim=1)  # :426
    return F.conv2d(F.gelu(a) * g, P[pfx + ".project_out.weight"])               # :427-428


def tblock(x, amp, pha, img, P, pfx, att, light):
    """TransformerBlock.forward, FDN_arch.py:666-677."""
    if att:
        x = x + fdsa(_ln(x, P, pfx + ".norm1"), P, pfx + ".attn")
    x = x + fdffn(_ln(x, P, pfx + ".norm2"), P, pfx + ".ffn")
    if light:
        x = x + fcaffn(_ln(x, P, pfx + ".norm3"), amp, pha, img, P, pfx + ".ffn2")
    return x


def fuse(enc, dnc, P, pfx):
    """Fuse.forward, FDN_arch.py:688-695."""
    x = F.conv2d(torch.cat([enc, dnc], dim=1), P[pfx + ".conv.weight"], P[pfx + ".conv.bias"])
    x = tblock(x, None, None, None, P, pfx + ".att_channel", att=False, light=False)
    x = F.conv2d(x, P[pfx + ".conv2.weight"], P[pfx + ".conv2.bias"])
    n = x.shape[1] // 2
    return x[:, :n] + x[:, n:]


def downsample(x, P, pfx):
    """Downsample, FDN_arch.py:715-723."""
    return F.conv2d(bilinear_half(x), P[pfx + ".body.1.weight"], padding=1)


def upsample(x, P, pfx):
    """Upsample, FDN_arch.py:726-734."""
    return F.conv2d(bilinear_x2(x), P[pfx + ".body.1.weight"], padding=1)


def _count_blocks(P, pfx):
    n = 0
    while f"{pfx}.{n}.norm2.body.weight" in P:
        n += 1
    return n


def fdformer(inp, ori, amps, phas, imgs, P, pfx="net_p"):
    """FDformer.forward, FDN_arch.py:810-842.  amps/phas/imgs: lists for levels 1..3."""
    def stage(x, name, lvl, light):
        for i in range(_count_blocks(P, f"{pfx}.{name}")):
            x = tblock(x, amps[lvl], phas[lvl], imgs[lvl], P, f"{pfx}.{name}.{i}", True, light)
        return x

    e1 = F.conv2d(inp, P[pfx + ".patch_embed.proj.weight"], padding=1)          # :813
    e1 = stage(e1, "encoder_level1", 0, True)
    e2 = stage(downsample(e1, P, pfx + ".down1_2"), "encoder_level2", 1, True)
    e3 = stage(downsample(e2, P, pfx + ".down2_3"), "encoder_level3", 2, True)
    d3 = stage(e3, "decoder_level3", 2, False)
    d2 = fuse(upsample(d3, P, pfx + ".up3_2"), e2, P, pfx + ".fuse2")            # :824-826
    d2 = stage(d2, "decoder_level2", 1, False)
    d1 = fuse(upsample(d2, P, pfx + ".up2_1"), e1, P, pfx + ".fuse1")            # :829-831
    d1 = stage(d1, "decoder_level1", 0, False)
    d1 = stage(d1, "refinement", 0, False)
    return F.conv2d(d1, P[pfx + ".output.weight"], padding=1) + ori             # :836-841


# --------------------------------------------------------------------------------------
# MAR (amplitude / gamma-curve pre-net)
# --------------------------------------------------------------------------------------
def _conv(x, P, pfx, **kw):
    return F.conv2d(x, P[pfx + ".weight"], P[pfx + ".bias"], **kw)


def _mlp2(x, P, pfx):
    """1x1 -> LeakyReLU(0.1) -> 1x1 (FDN_arch.py:79-86,127-134)."""
    return _conv(F.leaky_relu(_conv(x, P, pfx + ".0"), 0.1), P, pfx + ".2")


def freblock(x, P, pfx):
    """FreBlock.forward, FDN_arch.py:88-100 (no denormal fix: angle of +-0 imag as rfft2 gives it)."""
    H, W = x.shape[-2:]
    z = torch.fft.rfft2(_conv(x, P, pfx + ".fpre"))
    z = polar(_mlp2(z.abs(), P, pfx + ".process1"), _mlp2(z.angle(), P, pfx + ".process2"))
    return torch.fft.irfft2(z, s=(H, W)) + x


def processblock(x, P, pfx, cat=False):
    """ProcessBlock(spatial=False).forward.  FDN_arch.py:109-118: FreBlock(x)+x, `.cat` is dead.
    cat=True is the LOL-v1 variant, fdnlol24_arch.py:769-776: cat(FreBlock(x)) + x (a live 1x1 conv)."""
    if cat:
        return _conv(freblock(x, P, pfx + ".frequency_process"), P, pfx + ".cat") + x
    return freblock(x, P, pfx + ".frequency_process") + x


def fourier_fuse(x1, x2, x4, P, pfx):
    """fourier_fuse.forward, FDN_arch.py:136-148.  fpre.1 is a 1x1 *depthwise* conv with
    padding=1 (:126): the map grows to (H+2, W+2) with a bias-only border; irfft2(s=(H,W))
    then crops the spectrum to [:H, :W//2+1]."""
    x = torch.cat([x1, x2, x4], dim=1)
    H, W = x.shape[-2:]
    y = _conv(x, P, pfx + ".fpre.0")
    y = _conv(y, P, pfx + ".fpre.1", padding=1, groups=y.shape[1])
    z = torch.fft.rfft2(y)
    z = polar(_mlp2(z.abs(), P, pfx + ".process1"), _mlp2(z.angle(), P, pfx + ".process2"))
    return _conv(torch.fft.irfft2(z, s=(H, W)), P, pfx + ".fourier_out", padding=1)


def _basic(x, P, pfx, relu, **kw):
    y = _conv(x, P, pfx + ".main.0", **kw)
    return F.leaky_relu(y, 0.1) if relu else y


def _basic_t(x, P, pfx):
    """BasicConv(transpose=True, k=4, s=2): padding = k//2-1 = 1 (FDN_arch.py:21-23), LeakyReLU."""
    y = F.conv_transpose2d(x, P[pfx + ".main.0.weight"], P[pfx + ".main.0.bias"], stride=2, padding=1)
    return F.leaky_relu(y, 0.1)


def mar_arch(x, ratio, P, pfx, cat=False):
    """MAR_archa.forward, FDN_arch.py:203-257.  ratio: (B,1,1,1); always applied (:213-219).
    cat=True: fourier_multi_scale_gamma2 of the LOL-v1 variant (fdnlol24_arch.py:147-209; use_ratio=True there,
    :991), identical wiring with the live `.cat` ProcessBlock."""
    processblock = lambda t, P_, pf: globals()["processblock"](t, P_, pf, cat)
    x_2 = x[:, :, ::2, ::2]                       # nearest 0.5, :205
    x_4 = x_2[:, :, ::2, ::2]                     # :206
    z2 = processblock(_conv(F.pixel_unshuffle(x, 2), P, pfx + ".f2.0"), P, pfx + ".f2.1") * ratio
    z4 = processblock(_conv(F.pixel_unshuffle(x, 4), P, pfx + ".f1.0"), P, pfx + ".f1.1") * ratio
    x_ = processblock(_conv(x, P, pfx + ".f3.0"), P, pfx + ".f3.1") * ratio
    res1 = processblock(x_, P, pfx + ".Encoder.0")                                # :220
    z = _basic(res1, P, pfx + ".f3_down", True, stride=2, padding=1)              # :222
    z = _conv(_conv(torch.cat([z, z2], 1), P, pfx + ".FAM2.merge1"), P, pfx + ".FAM2.merge2", padding=1)
    res2 = processblock(z, P, pfx + ".Encoder.1")                                 # :224
    z = _basic(res2, P, pfx + ".f2_down", True, stride=2, padding=1)              # :226
    z = _conv(_conv(torch.cat([z, z4], 1), P, pfx + ".FAM1.merge1"), P, pfx + ".FAM1.merge2", padding=1)
    z = processblock(z, P, pfx + ".Encoder.2")                                    # :228
    up = lambda t: t.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)      # nearest x2
    z12 = res1[:, :, ::2, ::2]                                                    # :230
    z21, z42 = up(res2), up(z)
    z41 = up(z42)
    res2 = fourier_fuse(z12, res2, z42, P, pfx + ".AFFs.1")                       # :235
    res1 = fourier_fuse(res1, z21, z41, P, pfx + ".AFFs.0")                       # :236
    z = processblock(z, P, pfx + ".Decoder.0")
    o4 = torch.sigmoid(_basic(z, P, pfx + ".ConvsOut.0", False, padding=1) + x_4) + 1e-8   # :239-241
    z = _basic_t(z, P, pfx + ".f2_up")
    z = _basic(torch.cat([z, res2], 1), P, pfx + ".Convs.0", True)
    z = processblock(z, P, pfx + ".Decoder.1")
    o2 = torch.sigmoid(_basic(z, P, pfx + ".ConvsOut.1", False, padding=1) + x_2) + 1e-8   # :246-248
    z = _basic_t(z, P, pfx + ".f3_up")
    z = _basic(torch.cat([z, res1], 1), P, pfx + ".Convs.1", True)
    z = processblock(z, P, pfx + ".Decoder.2")
    o1 = torch.sigmoid(_basic(z, P, pfx + ".out", False, padding=1) + x) + 1e-8            # :253-255
    return o4, o2, o1


def mar(x, ratio, P, pfx="net_a", cat=False):
    """MAR.forward, FDN_arch.py:269-286: gamma curve 1-(1-x_k)^(40*i_k) on a bilinear pyramid."""
    x1 = x
    x2 = bilinear_half(x1)
    x3 = bilinear_half(x2)
    i3, i2, i1 = mar_arch(x, ratio, P, pfx + ".net", cat)
    g = lambda xx, ii: 1.0 - torch.pow(1.0 - xx, ii * 40.0)
    return g(x3, i3), g(x2, i2), g(x1, i1)


# --------------------------------------------------------------------------------------
# FDN top level
# --------------------------------------------------------------------------------------
def fdn_guidance(inp, ratio_i, P, cat=False):
    """FDN.forward up to the FDformer call, FDN_arch.py:869-914.
    Returns (amps, phas, imgs) for levels 1..3 and the three MAR outputs."""
    r = ratio_i.view(-1, 1, 1, 1)                                                # :872
    p1 = inp
    p2 = bilinear_half(p1)
    p3 = bilinear_half(p2)
    pyr = [p1, p2, p3]
    norms = ["norm1", "norm2", "norm3"]
    phas = [replace_denormals(torch.fft.rfft2(_ln(pyr[i], P, norms[i]))).angle() for i in range(3)]  # :878-892
    q3, q2, q1 = mar(inp, r, P, "net_a", cat)                                    # :895
    imgs = [q1, q2, q3]
    amps = [torch.fft.rfft2(_ln(imgs[i], P, norms[i])).abs() for i in range(3)]  # :896-914
    return amps, phas, imgs


def fdn_forward(P, inp, ratio_i):
    """FDN.forward, FDN_arch.py:869-921.  P: flat state dict (already in the working dtype).
    Returns (result, x_high1q, x_high2q, x_high3q)."""
    amps, phas, imgs = fdn_guidance(inp, ratio_i, P)
    out = fdformer(inp, inp, amps, phas, imgs, P, "net_p")                       # :916-919
    return out, imgs[0], imgs[1], imgs[2]


def fdn_lolv1_forward(P, inp, ratio_i):
    """FDN_lolv1.forward, fdnlol24_arch.py:982-1033: the same graph at dim=24 (E = 28/57/115, Hd = 64/129/259;
    widths come from the weights), MAR with the live ProcessBlock.cat, and the result returned four times."""
    amps, phas, imgs = fdn_guidance(inp, ratio_i, P, cat=True)
    out = fdformer(inp, inp, amps, phas, imgs, P, "net_p")
    return out, out, out, out


def lolv1_ratio(padded, lp_ratio):
    """inference_fdn_lolv1.py:57-61: mean over pixels of Grayscale(padded input) divided by LPNet's prediction.
    Grayscale = 0.2989 R + 0.587 G + 0.114 B (torchvision.transforms.functional.rgb_to_grayscale)."""
    r, g, b = padded.unbind(dim=1)
    gray = (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(1)
    return gray.mean(dim=(2, 3)) / lp_ratio


# --------------------------------------------------------------------------------------
# LPNet (I_predict_net)
# --------------------------------------------------------------------------------------
def _bn(x, P, pfx, eps=1e-5):
    """BatchNorm2d in eval mode (running stats)."""
    sc = P[pfx + ".weight"] / torch.sqrt(P[pfx + ".running_var"] + eps)
    sh = P[pfx + ".bias"] - P[pfx + ".running_mean"] * sc
    return x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)


def _seblock(x, P, pfx, stride, proj):
    """SEBlock.forward, LPNet_arch.py:69-81."""
    y = F.relu(_bn(F.conv2d(x, P[pfx + ".conv1.0.weight"], stride=stride), P, pfx + ".conv1.1"))
    y = F.relu(_bn(F.conv2d(y, P[pfx + ".conv2.0.weight"], padding=1), P, pfx + ".conv2.1"))
    y = _bn(F.conv2d(y, P[pfx + ".conv3.0.weight"]), P, pfx + ".conv3.1")
    g = y.mean(dim=(2, 3), keepdim=True)
    g = F.relu(F.conv2d(g, P[pfx + ".se.1.weight"], P[pfx + ".se.1.bias"]))
    g = torch.sigmoid(F.conv2d(g, P[pfx + ".se.3.weight"], P[pfx + ".se.3.bias"]))
    y = y * g
    sc = x
    if proj:
        sc = _bn(F.conv2d(x, P[pfx + ".shortcut.0.weight"], stride=stride), P, pfx + ".shortcut.1")
    return F.relu(y + sc)


def lpnet_forward(P, x):
    """I_predict_net.forward(x, use_ori_i=False), LPNet_arch.py:114-134 -> (B,1)."""
    y = F.relu(_bn(F.conv2d(x, P["conv1.0.weight"], stride=2, padding=3), P, "conv1.1"))
    y = F.avg_pool2d(y, 3, 2, 1)   # count_include_pad=True (nn.AvgPool2d default), LPNet_arch.py:94
    for name, num, stride in (("conv2", 3, 1), ("conv3", 3, 2), ("conv4", 6, 6)):
        for i in range(num):
            y = _seblock(y, P, f"{name}.{i}", stride if i == 0 else 1, i == 0)
    y = y.mean(dim=(2, 3))          # GAP then "B C H W -> B (H W C)" with H=W=1
    y = F.linear(y, P["fc.0.weight"], P["fc.0.bias"])
    y = F.linear(y, P["fc2.0.weight"], P["fc2.0.bias"])
    return torch.sigmoid(y)


# --------------------------------------------------------------------------------------
# caller harness (inference_fdn_lolblur.py:47-75, basicsr/utils/img_util.py:9-98)
# --------------------------------------------------------------------------------------
def harness_pre(img_u8_bgr_hwc):
    """uint8 BGR HWC -> /255 fp32 -> RGB CHW -> batch -> reflect-pad bottom/right to x32.
    Returns (padded NCHW fp32, h, w)."""
    t = torch.as_tensor(img_u8_bgr_hwc).to(torch.float32) / 255.0
    t = t.flip(-1).permute(2, 0, 1).unsqueeze(0).contiguous()
    h, w = t.shape[-2:]
    hn, wn = (32 - h % 32) % 32, (32 - w % 32) % 32
    return F.pad(t, (0, wn, 0, hn), mode="reflect"), h, w


def harness_post(result, h, w):
    """crop -> clamp(0,1) -> *255 -> round (half-to-even, numpy) -> uint8, RGB->BGR HWC."""
    r = result[0, :, :h, :w].to(torch.float32).clamp(0, 1)
    a = (r.permute(1, 2, 0).flip(-1).contiguous().numpy() * 255.0).round()
    return a.astype("uint8")


def grids_indices(h, w, crop_h, crop_w, scale=1):
    """Tile origins of ImageRestorationModel.grids, image_restoration_model.py:261-312 (adaptive steps, the last
    tile of a row / column is pulled back inside the image).  Returns (crop_h, crop_w, [(i, j), ...])."""
    import math
    crop_h, crop_w = crop_h // scale * scale, crop_w // scale * scale                       # :276
    num_row, num_col = (h - 1) // crop_h + 1, (w - 1) // crop_w + 1                         # :278-279
    step_j = crop_w if num_col == 1 else math.ceil((w - crop_w) / (num_col - 1) - 1e-8)     # :282
    step_i = crop_h if num_row == 1 else math.ceil((h - crop_h) / (num_row - 1) - 1e-8)     # :283
    step_i, step_j = step_i // scale * scale, step_j // scale * scale                       # :286-287
    idx = []
    i, last_i = 0, False
    while i < h and not last_i:                                                             # :294-309
        j = 0
        if i + crop_h >= h:
            i, last_i = h - crop_h, True
        last_j = False
        while j < w and not last_j:
            if j + crop_w >= w:
                j, last_j = w - crop_w, True
            idx.append((i, j))
            j += step_j
        i += step_i
    return crop_h, crop_w, idx


def grids_split(x, crop_h, crop_w):
    """grids(): (1,C,h,w) -> (T,C,crop_h,crop_w) tiles + origins (scale = 1), image_restoration_model.py:303-312."""
    assert x.shape[0] == 1                                                                  # :265
    ch, cw, idx = grids_indices(x.shape[2], x.shape[3], crop_h, crop_w)
    return torch.cat([x[:, :, i:i + ch, j:j + cw] for i, j in idx], dim=0), idx


def grids_merge(outs, idx, h, w):
    """grids_inverse(): overlapping tiles are accumulated in order and divided by the coverage count,
    image_restoration_model.py:315-339."""
    T, C, ch, cw = outs.shape
    preds = torch.zeros((1, C, h, w), dtype=outs.dtype)
    count = torch.zeros((1, 1, h, w), dtype=outs.dtype)
    for t, (i, j) in enumerate(idx):
        preds[0, :, i:i + ch, j:j + cw] += outs[t]
        count[0, 0, i:i + ch, j:j + cw] += 1.0
    return preds / count


def gaussian_kernel_11():
    """cv2.getGaussianKernel(11, 1.5) (OpenCV, not vendored by the reference; unpinned version): for ksize > 7 or an
    explicit sigma it is exp(-(i - (ksize-1)/2)^2 / (2 sigma^2)) normalised to sum 1, in float64."""
    i = torch.arange(11, dtype=torch.float64) - 5.0
    k = torch.exp(-(i * i) / (2.0 * 1.5 * 1.5))
    return k / k.sum()


def calculate_psnr(img1, img2, crop_border=0):
    """basicsr/metrics/psnr_ssim.py:8-73 for (C,H,W) tensors, test_y_channel=False: float64 MSE, peak by img1.max()."""
    a, b = img1.to(torch.float64), img2.to(torch.float64)
    if crop_border:
        a, b = a[..., crop_border:-crop_border, crop_border:-crop_border], b[..., crop_border:-crop_border, crop_border:-crop_border]
    mse = torch.mean((a - b) ** 2).item()
    if mse == 0:
        return float("inf")
    peak = 1.0 if a.max().item() <= 1 else 255.0
    import math
    return 20.0 * math.log10(peak / math.sqrt(mse))


def ssim_3d(img1, img2, crop_border=0):
    """calculate_ssim(ssim3d=True) -> _ssim_3d, psnr_ssim.py:163-197: an 11x11x11 Gaussian window (outer product of three
    getGaussianKernel(11,1.5)) over the (H, W, C) volume with replicate padding, in float32; mean of the SSIM map."""
    a, b = img1.to(torch.float64), img2.to(torch.float64)
    if crop_border:
        a, b = a[..., crop_border:-crop_border, crop_border:-crop_border], b[..., crop_border:-crop_border, crop_border:-crop_border]
    max_value = 1 if a.max().item() <= 1 else 255
    C1, C2 = (0.01 * max_value) ** 2, (0.03 * max_value) ** 2
    k = gaussian_kernel_11()
    window = torch.outer(k, k)
    kern = torch.stack([window * kk for kk in k], dim=0).to(torch.float32)[None, None]        # (1,1,11,11,11): :151-156
    x = a.permute(1, 2, 0).to(torch.float32)[None, None]                                       # (H, W, C) volume
    y = b.permute(1, 2, 0).to(torch.float32)[None, None]
    conv = lambda t: F.conv3d(F.pad(t, (5, 5, 5, 5, 5, 5), mode="replicate"), kern)
    mu1, mu2 = conv(x), conv(y)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1, s2, s12 = conv(x * x) - mu1_sq, conv(y * y) - mu2_sq, conv(x * y) - mu1_mu2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return float(ssim_map.mean())


def to_y_channel(img_chw_bgr):
    """basicsr/metrics/metric_util.py:34-47 + utils/matlab_functions.py:207-238 (bgr2ycbcr, y_only) for a (3,H,W) BGR tensor in
    [0, 255]: float32 image / 255, float64 dot with (24.966, 128.553, 65.481) + 16, / 255 back to float32, * 255 in float32."""
    x = (img_chw_bgr.to(torch.float32) / 255.0).to(torch.float64)
    y = x[0] * 24.966 + x[1] * 128.553 + x[2] * 65.481 + 16.0
    return ((y / 255.0).to(torch.float32) * 255.0)[None]                       # (1,H,W), float32, range [16, 235]


def _filter2d(img, window, border):
    """cv2.filter2D(img, -1, window, borderType) on a float64 (H,W) plane (OpenCV is not vendored by the reference: correlation,
    anchor at the centre; default border BORDER_REFLECT_101 = torch 'reflect', BORDER_REPLICATE = 'replicate')."""
    r = window.shape[0] // 2
    t = F.pad(img[None, None], (r, r, r, r), mode=border)
    return F.conv2d(t, window[None, None])[0, 0]


def _ssim_planes(a, b, C1, C2, border, crop):
    k = gaussian_kernel_11()
    window = torch.outer(k, k)
    f = lambda t: _filter2d(t, window, border)
    cut = (lambda t: t[5:-5, 5:-5]) if crop else (lambda t: t)
    mu1, mu2 = cut(f(a)), cut(f(b))
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1, s2, s12 = cut(f(a * a)) - mu1_sq, cut(f(b * b)) - mu2_sq, cut(f(a * b)) - mu1_mu2
    return ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))


def ssim_2d(img1, img2, crop_border=0):
    """calculate_ssim(ssim3d=False) -> _ssim, psnr_ssim.py:84-116: per channel an 11x11 Gaussian filter in float64 (cv2.filter2D,
    default reflect-101 border), the valid region [5:-5, 5:-5] only, mean over (H-10, W-10, C)."""
    a, b = img1.to(torch.float64), img2.to(torch.float64)
    if crop_border:
        a, b = a[..., crop_border:-crop_border, crop_border:-crop_border], b[..., crop_border:-crop_border, crop_border:-crop_border]
    max_value = 1 if a.max().item() <= 1 else 255
    C1, C2 = (0.01 * max_value) ** 2, (0.03 * max_value) ** 2
    maps = [_ssim_planes(a[c], b[c], C1, C2, "reflect", True) for c in range(a.shape[0])]
    return float(torch.stack(maps).mean())


def ssim_y(img1_bgr, img2_bgr, crop_border=0):
    """calculate_ssim(test_y_channel=True) -> to_y_channel + _ssim_cly, psnr_ssim.py:199-240, :275-278: the Y plane, 11x11 Gaussian
    with BORDER_REPLICATE, no valid-region crop, constants for a 255 range."""
    a, b = img1_bgr.to(torch.float64), img2_bgr.to(torch.float64)
    if crop_border:
        a, b = a[..., crop_border:-crop_border, crop_border:-crop_border], b[..., crop_border:-crop_border, crop_border:-crop_border]
    ya, yb = to_y_channel(a)[0].to(torch.float64), to_y_channel(b)[0].to(torch.float64)
    return float(_ssim_planes(ya, yb, (0.01 * 255) ** 2, (0.03 * 255) ** 2, "replicate", False).mean())


def psnr_y(img1_bgr, img2_bgr, crop_border=0):
    """calculate_psnr(test_y_channel=True), psnr_ssim.py:52-61: MSE of the two Y planes (float32 values, float64 mean)."""
    a, b = img1_bgr.to(torch.float64), img2_bgr.to(torch.float64)
    if crop_border:
        a, b = a[..., crop_border:-crop_border, crop_border:-crop_border], b[..., crop_border:-crop_border, crop_border:-crop_border]
    ya, yb = to_y_channel(a), to_y_channel(b)
    d = (ya - yb)                                                                # float32 difference (numpy float32 arrays, :58)
    mse = float((d * d).mean(dtype=torch.float32))
    if mse == 0:
        return float("inf")
    peak = 1.0 if ya.max().item() <= 1 else 255.0
    return 20.0 * math.log10(peak / math.sqrt(mse))


def psnr(a, b, peak=1.0):
    """20*log10(peak/sqrt(mse)) (basicsr/metrics/psnr_ssim.py:59-63)."""
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else 20.0 * math.log10(peak / math.sqrt(mse))


def cast_params(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
