"""GPU parity: the HIP-backed modules (through the C ABI) against the oracle and the golden fixtures.

Tolerance policy: a block passes when its error against the fp64 oracle is at
most 4x the error the fp32 reference itself has against fp64, plus a 2e-6 relative floor.  End to
end parity is checked in the tamed-weights regime with a PSNR floor, next to the reference's own
self-noise (tests/golden/selfnoise.json)."""
import numpy as np
import pytest
import torch

import fdn_oracle as O
from common import assert_close_cond, fdn_weights, fixture, fixture_weights, lpnet_weights, rel_rms

pytestmark = pytest.mark.gpu
F64 = torch.float64


@pytest.fixture(scope="module")
def A():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import fdn_hip
    fdn_hip.lib()   # fail loudly if the HIP extension is not built
    from basicsr.models.archs import FDN_arch
    return FDN_arch


def dev(t):
    return t.to("cuda:0").contiguous()


def load(mod, sd):
    mod.load_state_dict(sd, strict=True)
    return mod.to("cuda:0").eval()


def truth(fn, sd, *xs):
    with torch.no_grad():
        return fn({"." + k: v.to(F64) for k, v in sd.items()}, *[x.to(F64) if x is not None else None for x in xs])


@pytest.mark.parametrize("name,c", [("fdsa_c32", 32), ("fdsa_c64", 64), ("fdsa_c128", 128)])
def test_fdsa(A, name, c):
    fx = fixture(name)
    sd = fixture_weights(name, fx["shapes"])
    m = load(A.FDSA(c), sd)
    with torch.no_grad():
        got = m(dev(fx["x"]))
    assert_close_cond(got, fx["y"], truth(lambda P, x: O.fdsa(x, P, ""), sd, fx["x"]), name)


@pytest.mark.parametrize("name,c", [("fdffn_c32", 32), ("fdffn_c64", 64), ("fdffn_c128", 128)])
def test_fdffn(A, name, c):
    fx = fixture(name)
    sd = fixture_weights(name, fx["shapes"])
    m = load(A.FDFFN(c), sd)
    with torch.no_grad():
        got = m(dev(fx["x"]))
    assert_close_cond(got, fx["y"], truth(lambda P, x: O.fdffn(x, P, ""), sd, fx["x"]), name)


@pytest.mark.parametrize("name,c", [("fcaffn_c32_32x32", 32), ("fcaffn_c32_24x40", 32), ("fcaffn_c64_46x40", 64),
                                    ("fcaffn_c128_16x16", 128)])
def test_fcaffn(A, name, c):
    fx = fixture(name)
    sd = fixture_weights(name, fx["shapes"])
    m = load(A.FCAFFN(c), sd)
    with torch.no_grad():
        got = m(dev(fx["x"]), dev(fx["amp"]), dev(fx["pha"]), dev(fx["img"]))
    t64 = truth(lambda P, x, a, p, i: O.fcaffn(x, a, p, i, P, ""), sd, fx["x"], fx["amp"], fx["pha"], fx["img"])
    assert_close_cond(got, fx["y"], t64, name)


@pytest.mark.parametrize("name,light", [("tblock_enc_c32", True), ("tblock_dec_c32", False)])
def test_tblock(A, name, light):
    fx = fixture(name)
    sd = fixture_weights(name, fx["shapes"], po_scale=float(fx["po_scale"]))
    m = load(A.TransformerBlock(dim=32, att=True, use_light=light, use_img=light), sd)
    with torch.no_grad():
        got = m((dev(fx["x"]), dev(fx["amp"]), dev(fx["pha"]), dev(fx["img"])))[0]
    t64 = truth(lambda P, x, a, p, i: O.tblock(x, a, p, i, P, "", True, light), sd, fx["x"], fx["amp"], fx["pha"], fx["img"])
    assert_close_cond(got, fx["y"], t64, name)


def test_fuse_resample_embed(A):
    fx = fixture("fuse_n32")
    sd = fixture_weights("fuse_n32", fx["shapes"])
    m = load(A.Fuse(32), sd)
    with torch.no_grad():
        got = m(dev(fx["enc"]), dev(fx["dnc"]))
    assert_close_cond(got, fx["y"], truth(lambda P, e, d: O.fuse(e, d, P, ""), sd, fx["enc"], fx["dnc"]), "fuse")
    for name, cls, fn, c in (("downsample_c32", A.Downsample, O.downsample, 32), ("upsample_c64", A.Upsample, O.upsample, 64)):
        fx = fixture(name)
        sd = fixture_weights(name, fx["shapes"])
        m = load(cls(c), sd)
        with torch.no_grad():
            got = m(dev(fx["x"]))
        assert_close_cond(got, fx["y"], truth(lambda P, x: fn(x, P, ""), sd, fx["x"]), name)
    fx = fixture("patch_embed_3_32")
    sd = fixture_weights("patch_embed_3_32", fx["shapes"])
    m = load(A.OverlapPatchEmbed(3, 32), sd)
    with torch.no_grad():
        got = m(dev(fx["x"]))
    assert rel_rms(got.cpu(), fx["y"]) < 2e-6


def test_mar_pieces(A):
    fx = fixture("freblock_c12")
    sd = fixture_weights("freblock_c12", fx["shapes"])
    m = load(A.FreBlock(12), sd)
    with torch.no_grad():
        got = m(dev(fx["x"]))
    assert_close_cond(got, fx["y"], truth(lambda P, x: O.freblock(x, P, ""), sd, fx["x"]), "freblock")
    fx = fixture("fourier_fuse_84_12")
    sd = fixture_weights("fourier_fuse_84_12", fx["shapes"])
    m = load(A.fourier_fuse(84, 12), sd)
    with torch.no_grad():
        got = m(dev(fx["x1"]), dev(fx["x2"]), dev(fx["x4"]))
    t64 = truth(lambda P, a, b, c: O.fourier_fuse(a, b, c, P, ""), sd, fx["x1"], fx["x2"], fx["x4"])
    assert_close_cond(got, fx["y"], t64, "fourier_fuse")


def test_mar_full(A):
    fx = fixture("mar_full")
    sd = fixture_weights("mar_full", fx["shapes"])
    m = load(A.MAR(True), sd)
    with torch.no_grad():
        y3, y2, y1 = m(dev(fx["x"]), dev(fx["ratio"]))
    for got, key in ((y3, "y3"), (y2, "y2"), (y1, "y1")):
        p = O.psnr(got.cpu(), fx[key])
        assert p > 100.0, f"mar {key}: PSNR {p:.1f} dB"


def test_lpnet_real_weights(A):
    from basicsr.models.archs.LPNet_arch import I_predict_net
    fx = fixture("lpnet_real")
    m = load(I_predict_net(), lpnet_weights())
    with torch.no_grad():
        y = m(dev(fx["x"]))
        g = torch.Generator().manual_seed(52)
        y2 = m(dev(torch.rand(1, 3, 736, 1280, generator=g)))
    assert torch.allclose(y.cpu(), fx["y"], rtol=0, atol=5e-6), (y.cpu() - fx["y"]).abs().max()
    assert torch.allclose(y2.cpu(), fx["y_736x1280_seed52"], rtol=0, atol=5e-6)


@pytest.mark.parametrize("name", ["fdn_tamed_64", "fdn_tamed_96x160_wc"])
def test_fdn_end_to_end_tamed(A, name):
    """Fixed 100 dB floor against the reference's own outputs - on frames WITHOUT ill-conditioned spots.  With the tamed synthetic weights nearly
    every random 96 x 160 frame has a spot where a spectrum bin of an FDSA block lies within fp32 rounding of zero, so that one rounding decides
    its phase and 1e-6 .. 1e-3 of a 16 x 16 window of y (tests/golden/wellcond_search.txt; DESIGN.md, parity policy): a fixed floor there pins the
    luck of one evaluation - `fdn_tamed_96x160` is such a frame and is held window by window in the test below.  `fdn_tamed_96x160_wc` is the
    first frame of a committed seed search (tests/golden/make_golden_wellcond.py: float64 oracle with rounding-sized noise on every
    TransformerBlock output / every forward FFT, fp32 oracle under one-ulp input changes) whose worst window stays at rounding level in every
    evaluation: 2.3e-6 over the 49 of the search, 3.1e-6 over the 145 of its conditioning file; on it a healthy path cannot fall below ~115 dB."""
    fx = fixture(name)
    m = load(A.FDN(), fdn_weights(tame=float(fx["tame"])))
    with torch.no_grad():
        out = m(dev(fx["x"]), ratio_i=dev(fx["ratio"]), device=torch.device("cuda:0"))
    for got, key, floor in zip(out, ("y", "q1", "q2", "q3"), (100.0, 100.0, 100.0, 100.0)):
        p = O.psnr(got.cpu(), fx[key])
        assert p > floor, f"{name}.{key}: PSNR {p:.1f} dB (reference self-noise: see tests/golden/selfnoise.json)"


@pytest.mark.parametrize("B,E,N,H,W", [(2, 76, 64, 48, 80), (1, 76, 64, 40, 36), (1, 57, 48, 16, 12), (2, 76, 64, 184, 320)])
def test_fdsa_out_level2_on_the_bf16_pipe(B, E, N, H, W):
    """The level-2 FDSA tail (three LayerNorms * v_value, project_out, residual, statistics; FDN_arch.py:633-639, :671) with project_out on
    v_mfma_f32_32x32x16_bf16 - one pixel per lane, eight waves around one packed operand image; the default since round 5.  Against float64 it
    has to be as good as the fp32-MFMA form it replaced (fdn_set_matrix_pipe(2), "bf16-narrow"; a ragged last tile, E < 2 * SH and N < 64
    included); the switch goes back to the default afterwards."""
    import ctypes
    import fdn_hip
    P = H * W
    g = torch.Generator().manual_seed(11)
    o, w = torch.randn(B, 4 * E, P, generator=g), torch.randn(N, 3 * E, generator=g) / (3 * E) ** .5
    g3, b3, res = torch.randn(3 * E, generator=g), torch.randn(3 * E, generator=g), torch.randn(B, N, P, generator=g)
    od, v = o.double(), o.double()[:, 3 * E:]
    parts = []
    for k in range(3):
        og = od[:, k * E:(k + 1) * E]
        mu, var = og.mean(1, keepdim=True), og.var(1, unbiased=False, keepdim=True)
        parts.append(((og - mu) / torch.sqrt(var + 1e-5) * g3[k * E:(k + 1) * E].double()[None, :, None] + b3[k * E:(k + 1) * E].double()[None, :, None]) * v)
    ref = torch.einsum("nk,bkp->bnp", w.double(), torch.cat(parts, 1)) + res.double()
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    errs = {}
    try:
        for mode in ("bf16-narrow", "bf16"):
            fdn_hip.set_matrix_pipe(mode)
            od_, wd, gd, bd, rd = dev(o), dev(w), dev(g3), dev(b3), dev(res)
            out = torch.full((B, N, P), float("nan"), device="cuda:0")
            st = torch.full((B, 2, P), float("nan"), device="cuda:0")
            rc = fdn_hip.lib().fdn_fdsa_out(ptr(od_), ptr(wd), ptr(gd), ptr(bd), ptr(rd), ptr(out), ptr(st), B, E, N, P, 0, fdn_hip.stream())
            assert rc == 0
            torch.cuda.synchronize()
            errs[mode] = rel_rms(out.cpu(), ref)
            assert rel_rms(st[:, 0].cpu(), ref.mean(1)) < 1e-5 and rel_rms(st[:, 1].cpu(), 1 / torch.sqrt(ref.var(1, unbiased=False) + 1e-5)) < 1e-5, mode
    finally:
        fdn_hip.set_matrix_pipe("bf16")
    assert errs["bf16"] < 2e-6 and errs["bf16"] < 1.5 * errs["bf16-narrow"] + 2e-8, errs


def _window_rms(d, size):
    B, C, H, W = d.shape
    return d.double().pow(2).reshape(B, C, H // size, size, W // size, size).mean((1, 3, 5)).sqrt().reshape(-1)


@pytest.mark.parametrize("name", ["fdn_tamed_64", "fdn_tamed_96x160", "fdn_tamed_96x160_wc"])
def test_fdn_end_to_end_tamed_conditioning(A, name):
    """The end-to-end forwards held per WINDOW to what fp32 can do there (the footing of test_config1_736x1280_frame_matches_reference) - THE gate
    for `fdn_tamed_96x160`.  Against the float64 truth a window may be off by 4 x the worst of: the reference's own error there,
    and the measured susceptibility of THAT window - five fp32 oracle evaluations (the frame and one-ulp perturbations), eight float64 evaluations
    with the rounding of an fp32 FFT emulated (make_golden_small_cond.py), and 96 float64 evaluations with rounding-sized white noise (2e-7 / 3e-7 of
    the rms) on every TransformerBlock output (make_golden_wellcond.py `blocknoise`) - plus a few ulp.  None of the measures involves this library.
    The 96 x 160 frame has three discrete states under such noise (windows 27 / 17 / 26 / 16 at 5.6e-5, 50 / 40 at 8.5e-6, 12 / 2 / 13 / 3 at 3.2e-4:
    one spectrum bin each whose sign rounding decides); an fp32 evaluation lands in some of them, which ones is a matter of its last bits.  Every
    other window - and every window of the other two frames - has to sit at rounding level."""
    fx, cond = fixture(name), fixture(name + "_cond")
    m = load(A.FDN(), fdn_weights(tame=float(fx["tame"])))
    with torch.no_grad():
        out = m(dev(fx["x"]), ratio_i=dev(fx["ratio"]), device=torch.device("cuda:0"))
    for got, key, size in zip(out, ("y", "q1", "q2", "q3"), (16, 16, 8, 4)):
        truth = cond[key + "_f64"]
        e_hip, e_ref = _window_rms(got.cpu().double() - truth, size), _window_rms(fx[key].double() - truth, size)
        susc = torch.maximum(cond[key + "_susc"].max(0).values, cond[key + "_noise"].max(0).values)
        if key + "_blocknoise" in cond:
            susc = torch.maximum(susc, cond[key + "_blocknoise"].max(0).values)
        allow = 4.0 * torch.maximum(e_ref, susc) + 5e-8
        bad = (e_hip > allow).nonzero().flatten().tolist()
        assert not bad, f"{name}.{key}: windows {bad[:8]}: HIP {e_hip[bad[:8]].tolist()} allowed {allow[bad[:8]].tolist()}"
        assert float(e_hip.median()) <= 2.0 * float(e_ref.median()) + 2e-8, (name, key, float(e_hip.median()), float(e_ref.median()))


ILL_CONDITIONED_96x160 = (27, 17, 26, 16, 50, 40, 12, 2, 13, 3)      # the frame's three discrete states (16 x 16 windows of y; docstring above)


def test_fdn_tamed_96x160_hard_regression(A):
    """ The frame that blocked the level-2 bf16-pipe default in round 4 keeps two HARD checks beside the per-window gate:
    * with fdn_set_matrix_pipe(2) ("bf16-narrow": the level-2 FDSA tail on its fp32-MFMA form - the arithmetic of rounds 3-4, which lands in none of
      the frame's discrete states) the fixed 100 dB floor against the reference's outputs still holds - a regression in anything else shows here;
    * on the default route every window OUTSIDE the documented ill-conditioned set is at rounding level (<= 3e-6 against the float64 truth) and the
      whole frame stays above 85 dB (round 5's default measured 87.9 dB: the (12, 2, 13, 3) state)."""
    import fdn_hip
    fx, cond = fixture("fdn_tamed_96x160"), fixture("fdn_tamed_96x160_cond")
    m = load(A.FDN(), fdn_weights(tame=float(fx["tame"])))
    try:
        fdn_hip.set_matrix_pipe("bf16-narrow")
        with torch.no_grad():
            out = m(dev(fx["x"]), ratio_i=dev(fx["ratio"]), device=torch.device("cuda:0"))
        for got, key in zip(out, ("y", "q1", "q2", "q3")):
            p = O.psnr(got.cpu(), fx[key])
            assert p > 100.0, f"fdn_tamed_96x160.{key} on the round-4 arithmetic (bf16-narrow): PSNR {p:.1f} dB"
    finally:
        fdn_hip.set_matrix_pipe("bf16")
    with torch.no_grad():
        y = m(dev(fx["x"]), ratio_i=dev(fx["ratio"]), device=torch.device("cuda:0"))[0].cpu()
    e = _window_rms(y.double() - cond["y_f64"], 16)
    outside = [w for w in range(e.numel()) if w not in ILL_CONDITIONED_96x160]
    worst = max(outside, key=lambda w: float(e[w]))
    p = O.psnr(y, fx["y"])
    print(f"fdn_tamed_96x160 default route: PSNR {p:.1f} dB; worst window outside the ill-conditioned set #{worst} {float(e[worst]):.2e}; "
          + "inside: " + ", ".join(f"#{w} {float(e[w]):.1e}" for w in ILL_CONDITIONED_96x160))
    assert float(e[worst]) <= 3e-6, (worst, float(e[worst]))
    assert p > 85.0, p


def test_harness_u8(A):
    """uint8 in -> uint8 out through the drop-in modules, mirroring inference_fdn_lolblur.py:47-75."""
    from basicsr.models.archs.LPNet_arch import I_predict_net
    fx = fixture("harness_u8")
    padded, h, w = O.harness_pre(fx["img"].numpy())
    net = load(A.FDN(), fdn_weights(tame=float(fx["tame"])))
    lp = load(I_predict_net(), lpnet_weights())
    with torch.no_grad():
        x = dev(padded)
        ratio = lp(x)
        res = net(x, ratio_i=ratio, device=x.device)[0]
    assert torch.allclose(ratio.cpu(), fx["ratio"], atol=5e-6)
    out = O.harness_post(res.cpu(), h, w)
    diff = out.astype(int) - fx["out_u8"].numpy().astype(int)
    # a 1e-5 float difference flips round() only next to a .5 boundary: off-by-one on <0.5 % of bytes
    assert O.psnr(res.cpu()[:, :, :h, :w].clamp(0, 1), fx["result"]) > 80.0   # fixture = clamped crop; white-noise uint8 input, measured 89 dB
    assert abs(diff).max() <= 1 and (diff != 0).mean() < 5e-3


def test_harness_pre_post_kernels_bit_exact(A):
    """fdn_pre_u8 / fdn_post_u8 against the oracle's host restatement (inference_fdn_lolblur.py:47-62,72-75): byte and
    float work, so bit-exact; batch of two, odd sizes, values on the .5 rounding boundary and outside [0,1]."""
    from fdn_hip import harness
    g = torch.Generator().manual_seed(5)
    for (h, w) in ((70, 90), (33, 64), (64, 64)):
        imgs = torch.randint(0, 256, (2, h, w, 3), generator=g, dtype=torch.uint8)
        x, hh, ww = harness.preprocess(imgs.cuda(), bgr=True)
        assert (hh, ww) == (h, w) and x.shape[-2] % 32 == 0 and x.shape[-1] % 32 == 0
        for b in range(2):
            ref, _, _ = O.harness_pre(imgs[b].numpy())
            assert torch.equal(x[b:b + 1].cpu(), ref)
        res = torch.randn(2, 3, x.shape[-2], x.shape[-1], generator=g) * 0.6 + 0.5
        res[0, :, 0, :8] = torch.tensor([0.5, 1.5, 2.5, 3.5, 126.5, 127.5, 254.5, 255.0]) / 255.0   # ties -> even
        out = harness.postprocess(res.cuda(), h, w, bgr=True).cpu().numpy()
        for b in range(2):
            assert (out[b] == O.harness_post(res[b:b + 1], h, w)).all()
        rgb = harness.postprocess(res.cuda(), h, w, bgr=False).cpu().numpy()
        assert (rgb[..., ::-1] == out).all()


def test_harness_enhance_u8_matches_fixture(A):
    """uint8 in -> uint8 out entirely on the GPU (fdn_hip.harness.enhance_u8) against the reference-generated fixture."""
    from basicsr.models.archs.LPNet_arch import I_predict_net
    from fdn_hip import harness
    fx = fixture("harness_u8")
    net = load(A.FDN(), fdn_weights(tame=float(fx["tame"])))
    lp = load(I_predict_net(), lpnet_weights())
    img = fx["img"].cuda().contiguous()
    out = harness.enhance_u8(net, lp, torch.stack([img, img]), bgr=True).cpu().numpy()
    assert (out[0] == out[1]).all()
    diff = out[0].astype(int) - fx["out_u8"].numpy().astype(int)
    assert abs(diff).max() <= 1 and (diff != 0).mean() < 5e-3


def test_batch_independence_and_determinism(A):
    """Samples never mix: out(batch)[i] == out(sample i), bit for bit; reruns identical."""
    m = load(A.FDN(), fdn_weights(tame=0.03))
    g = torch.Generator().manual_seed(5)
    x = dev(torch.rand(2, 3, 64, 96, generator=g))
    r = dev(torch.tensor([[0.4], [0.7]]))
    with torch.no_grad():
        full = m(x, ratio_i=r)[0]
        again = m(x, ratio_i=r)[0]
        one = m(x[1:2].contiguous(), ratio_i=r[1:2].contiguous())[0]
    assert torch.equal(full, again)
    assert torch.equal(full[1:2], one)


def test_no_cpu_fallback(A):
    m = A.FDN().eval()
    with pytest.raises(Exception):
        m(torch.rand(1, 3, 32, 32), ratio_i=torch.rand(1, 1))


# ---------------------------------------------------------------------------------------------------
# entry points that only the big shapes exercise, checked directly against fp64 math
# ---------------------------------------------------------------------------------------------------
def _rnd(*s, seed):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("K,N,H,W,pro", [(32, 152, 24, 40, "ln"), (64, 304, 16, 24, "ln"), (128, 612, 8, 24, "ln"),
                                        (96, 345, 8, 16, "none"), (86, 32, 46, 21, "none"), (459, 128, 8, 16, "ln3"),
                                        (114, 32, 24, 40, "ln3"), (32, 32, 16, 24, "muladd"), (300, 128, 9, 21, "ln3"),
                                        (200, 100, 9, 21, "none"), (345, 128, 23, 40, "none")])
def test_conv1x1_variants(A, K, N, H, W, pro):
    """Every GEMM kernel variant (small-K resident / streaming, generic resident / streaming, prologues,
    epilogues, fused statistics) against fp64."""
    from fdn_hip import ops
    B = 2
    x, w, res = _rnd(B, K, H, W, seed=1) * 1.5 + 0.3, _rnd(N, K, seed=2) / K ** 0.5, _rnd(B, N, H, W, seed=3)
    xd = x.double()
    kw = {}
    if pro == "ln":
        g, b = _rnd(K, seed=4), _rnd(K, seed=5)
        xin = O.ln_chan(xd, g.double(), b.double())
        kw["ln"] = (ops.chan_stats(dev(x)), dev(g), dev(b))
        xs = dev(x)
    elif pro == "ln3":
        E = K // 3
        g, b, vv = _rnd(K, seed=4), _rnd(K, seed=5), _rnd(B, E, H, W, seed=6)
        parts = [O.ln_chan(xd[:, i * E:(i + 1) * E], g[i * E:(i + 1) * E].double(), b[i * E:(i + 1) * E].double()) * vv.double()
                 for i in range(3)]
        xin = torch.cat(parts, 1)
        full = dev(torch.cat([x, vv], 1))
        xs = full[:, :K]
        kw["ln3_gate"] = (ops.chan_stats(xs, groups=3), dev(g), dev(b), full[:, K:])
    elif pro == "muladd":
        g, b, x1 = _rnd(K, seed=4), _rnd(K, seed=5), _rnd(B, K, H, W, seed=6)
        xin = O.ln_chan(xd, g.double(), b.double()) * x1.double() + x1.double()
        kw["ln_muladd"] = (ops.chan_stats(dev(x)), dev(g), dev(b), dev(x1))
        xs = dev(x)
    else:
        xin, xs = xd, dev(x)
    ref = torch.nn.functional.conv2d(xin, w.double().view(N, K, 1, 1)) + res.double()
    # without a weight cache: the fp32-MFMA kernels; with one: deep shapes (K, N >= 96) take the split-bf16 kernel (gemm_split.hip)
    for cache in (None, (ops.WeightCache(), "t")):
        got = ops.conv1x1(xs, dev(w), res=dev(res), want_stats=True, cache=cache, **kw)
        assert rel_rms(got.cpu(), ref) < 2e-6, cache
        mu, var = ref.mean(1), ref.var(1, unbiased=False)
        st = got._fdn_stats.cpu().view(B, 2, H, W)
        tol = 1e-5 if N <= 160 else 1e-4          # N > 160: statistics come from the one-pass fdn_chan_stats kernel
        assert rel_rms(st[:, 0], mu) < tol and rel_rms(st[:, 1], 1 / torch.sqrt(var + 1e-5)) < tol, cache


@pytest.mark.parametrize("K,N,H,W,pro,epi", [(128, 612, 23, 40, "ln", "none"), (128, 345, 184, 320, "ln", "none"), (459, 128, 23, 41, "ln3", "res"),
                                            (345, 128, 184, 320, "none", "res"), (128, 128, 23, 40, "muladd", "muladd"), (96, 96, 5, 7, "none", "bias"),
                                            (100, 130, 9, 13, "ln", "res"), (345, 128, 8, 17, "ln3", "none"), (128, 128, 184, 320, "muladd", "muladd"),
                                            (96, 460, 9, 13, "ln", "none"), (100, 300, 9, 13, "none", "bias"), (128, 612, 184, 320, "ln", "bias"),
                                            (32, 86, 23, 40, "ln", "none"), (64, 172, 46, 80, "ln", "none"), (48, 129, 9, 13, "ln", "bias"),
                                            (24, 64, 9, 13, "none", "none"), (64, 172, 368, 640, "ln", "none")])
def test_conv1x1_split_bf16_kernel(A, K, N, H, W, pro, epi):
    """gemm_split.hip (fp32 GEMM as six bf16 products of exactly split operands) against fp64, every prologue and epilogue,
    ragged K / N / pixel tails, the level-3 shapes of config 2: held to the bounds of the fp32-MFMA kernels (2e-6 relative RMS),
    and it must not be further from fp64 than those kernels by more than rounding noise."""
    import fdn_hip
    from fdn_hip import ops
    B = 2
    x, w = _rnd(B, K, H, W, seed=1) * 1.5 + 0.3, _rnd(N, K, seed=2) / K ** 0.5
    xd = x.double()
    kw, xs = {}, dev(x)
    if pro == "ln":
        g, b = _rnd(K, seed=4), _rnd(K, seed=5)
        xin = O.ln_chan(xd, g.double(), b.double())
        kw["ln"] = (ops.chan_stats(xs), dev(g), dev(b))
    elif pro == "ln3":
        E = K // 3
        g, b, vv = _rnd(K, seed=4), _rnd(K, seed=5), _rnd(B, E, H, W, seed=6)
        xin = torch.cat([O.ln_chan(xd[:, i * E:(i + 1) * E], g[i * E:(i + 1) * E].double(), b[i * E:(i + 1) * E].double()) * vv.double()
                         for i in range(3)], 1)
        full = dev(torch.cat([x, vv], 1))
        xs = full[:, :K]
        kw["ln3_gate"] = (ops.chan_stats(xs, groups=3), dev(g), dev(b), full[:, K:])
    elif pro == "muladd":
        g, b, x1 = _rnd(K, seed=4), _rnd(K, seed=5), _rnd(B, K, H, W, seed=6)
        xin = O.ln_chan(xd, g.double(), b.double()) * x1.double() + x1.double()
        kw["ln_muladd"] = (ops.chan_stats(xs), dev(g), dev(b), dev(x1))
    else:
        xin = xd
    ref = torch.nn.functional.conv2d(xin, w.double().view(N, K, 1, 1))
    bias = None
    if epi == "res":
        r = _rnd(B, N, H, W, seed=7)
        ref = ref + r.double()
        kw["res"] = dev(r)
    elif epi == "muladd":
        m, a = _rnd(B, N, H, W, seed=7), _rnd(B, N, H, W, seed=8)
        ref = ref * m.double() + a.double()
        kw["muladd"] = (dev(m), dev(a))
    elif epi == "bias":
        bias = _rnd(N, seed=9)
        ref = ref + bias.double().view(1, -1, 1, 1)
    want_stats = N <= 128 and K >= 96          # (the short-K strip form has no statistics epilogue: project_in convs do not need one)
    wc = ops.WeightCache()
    got = ops.conv1x1(xs, dev(w), None if bias is None else dev(bias), want_stats=want_stats, cache=(wc, "t"), **kw)
    assert any(k.endswith(":pk") for k in wc._store), "the packed-weight path was not taken"
    plain = ops.conv1x1(xs, dev(w), None if bias is None else dev(bias), want_stats=want_stats, **kw)
    e_split, e_f32 = rel_rms(got.cpu(), ref), rel_rms(plain.cpu(), ref)
    assert e_split < 2e-6 and e_split < 1.5 * e_f32 + 2e-8, (e_split, e_f32)
    if want_stats:
        st = got._fdn_stats.cpu().view(B, 2, H, W)
        assert rel_rms(st[:, 0], ref.mean(1)) < 1e-5 and rel_rms(st[:, 1], 1 / torch.sqrt(ref.var(1, unbiased=False) + 1e-5)) < 1e-5
    # bit-exact determinism and batch independence of the tiled launch
    again = ops.conv1x1(xs, dev(w), None if bias is None else dev(bias), cache=(wc, "t"), **kw)
    assert torch.equal(again, got)


@pytest.mark.parametrize("K0,K1,N,H,W", [(64, 64, 128, 23, 41), (96, 32, 96, 9, 13), (64, 64, 128, 368, 640), (32, 96, 130, 8, 17)])
def test_conv1x1_two_inputs_on_the_split_bf16_kernel(A, K0, K1, N, H, W):
    """Fuse.conv (FDN_arch.py:685: a 1x1 conv over cat([enc, dnc])) at level 2 on the K-streaming split-bf16 kernel (round 5: the descriptor is chosen
    per 32-deep chunk): against float64, with bias, the statistics epilogue, and against the fp32-MFMA route."""
    from fdn_hip import ops
    B = 2 if H < 100 else 1
    x0, x1 = _rnd(B, K0, H, W, seed=1), _rnd(B, K1, H, W, seed=2) * 0.7 + 0.1
    w, bias = _rnd(N, K0 + K1, seed=3) / (K0 + K1) ** 0.5, _rnd(N, seed=4)
    ref = torch.nn.functional.conv2d(torch.cat([x0, x1], 1).double(), w.double().view(N, -1, 1, 1), bias.double())
    wc = ops.WeightCache()
    want = N <= 128
    got = ops.conv1x1([dev(x0), dev(x1)], dev(w), dev(bias), want_stats=want, cache=(wc, "t"))
    assert any(k.endswith(":pk") for k in wc._store), "the packed-weight path was not taken"
    plain = ops.conv1x1([dev(x0), dev(x1)], dev(w), dev(bias), want_stats=want)
    e_split, e_f32 = rel_rms(got.cpu(), ref), rel_rms(plain.cpu(), ref)
    assert e_split < 2e-6 and e_split < 1.5 * e_f32 + 2e-8, (e_split, e_f32)
    if want:
        st = got._fdn_stats.cpu().view(B, 2, H, W)
        assert rel_rms(st[:, 0], ref.mean(1)) < 1e-5 and rel_rms(st[:, 1], 1 / torch.sqrt(ref.var(1, unbiased=False) + 1e-5)) < 1e-5


@pytest.mark.parametrize("H,W", [(32, 48), (96, 160)])
def test_fourier_fuse_per_source_resolution(A, H, W):
    """MAR's fourier_fuse (FDN_arch.py:120-147) fed from the sources at their own resolutions (round 5: the 1x1 conv fpre[0] commutes with the nearest
    replication of F.interpolate, :231-236) against the same module on the resized copies, both AFFs' channel layouts."""
    from fdn_hip import ops
    torch.manual_seed(5)
    c = 12
    for aff, same_c, up1_c in ((A.fourier_fuse(7 * c, c), [c], 2 * c), (A.fourier_fuse(7 * c, 2 * c), [c, 2 * c], 0)):
        aff = aff.to("cuda:0").eval()
        same = [dev(_rnd(2, k, H, W, seed=10 + i)) for i, k in enumerate(same_c)]
        up1 = dev(_rnd(2, up1_c, H // 2, W // 2, seed=20)) if up1_c else None
        lvl = 4 if up1_c else 2
        up2 = dev(_rnd(2, 4 * c, H // lvl, W // lvl, seed=21))
        with torch.no_grad():
            got = aff.forward_multires(same, up1, up2)
            big2 = ops.resample(up2, ops.RS_NEAREST_X2)
            if up1_c:
                ref = aff(same[0], ops.resample(up1, ops.RS_NEAREST_X2), ops.resample(big2, ops.RS_NEAREST_X2))
            else:
                ref = aff(same[0], same[1], big2)
        assert got.shape == ref.shape
        assert rel_rms(got.cpu(), ref.cpu().double()) < 2e-6


@pytest.mark.parametrize("C,Cout,h,w,B", [(64, 32, 23, 40, 2), (128, 64, 12, 20, 2), (48, 24, 7, 9, 1), (96, 48, 5, 3, 1), (10, 5, 1, 1, 2), (16, 8, 1, 6, 1),
                                          (16, 8, 5, 1, 1), (64, 32, 184, 320, 1)])
def test_upsample_conv_without_the_x2_image(A, C, Cout, h, w, B):
    """Upsample (FDN_arch.py:726-734: bilinear x2, align_corners=False, then Conv2d 3x3 padding 1 without bias) as nine per-tap 1x1 products at low
    resolution + fdn_upconv_gather (round 5): against float64 torch and against the route it replaces (fdn_resample x2 + fdn_conv2d), at the bound of
    that route.  Shapes: the two widths of FDN, FDN_lolv1's, one-pixel and one-row / one-column images (every edge blend at once), the level-2 frame."""
    from fdn_hip import ops
    F = torch.nn.functional
    x, wt = _rnd(B, C, h, w, seed=1), _rnd(Cout, C, 3, 3, seed=2) / (3 * C ** 0.5)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="bilinear", align_corners=False), wt.double(), padding=1)
    got = ops.upsample_conv3x3(dev(x), dev(wt), cache=(ops.WeightCache(), "up"))
    old = ops.conv2d(ops.resample(dev(x), ops.RS_BILINEAR_X2), dev(wt), pad=1)
    assert got.shape == ref.shape
    e_new, e_old = rel_rms(got.cpu(), ref), rel_rms(old.cpu(), ref)
    assert e_new < 2e-6 and e_new < 1.5 * e_old + 5e-8, (e_new, e_old)
    assert (got.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert rel_rms(ops.upsample_conv3x3(dev(x), dev(wt)).cpu(), ref) < 2e-6        # without a weight cache: the fp32-MFMA GEMMs


@pytest.mark.parametrize("K,N,H,W,pro", [(459, 128, 23, 41, "ln3"), (345, 128, 8, 17, "ln3"), (300, 128, 9, 21, "ln3"), (459, 128, 184, 320, "ln3"),
                                        (128, 128, 23, 40, "muladd"), (100, 130, 9, 13, "muladd"), (114, 32, 24, 40, "ln3")])
def test_gemm_takes_its_own_layernorm_statistics(A, K, N, H, W, pro):
    """stats=None (round 5): the K-streaming split-bf16 kernel takes the LayerNorm statistics of its pixel tile itself, in a pass before the
    product (no fdn_chan_stats launch in front of the level-3 project_out / FCAFFN GEMMs, FDN_arch.py:633-639, :420).  Held to the bound of the
    route with the launch, against float64, and to rounding-level agreement with it; a shape no such kernel covers ((114, 32): level 1 without
    the one-launch tail) falls back to the launch inside ops.conv1x1."""
    from fdn_hip import ops
    B = 2
    x, w = _rnd(B, K, H, W, seed=1) * 1.5 + 0.3, _rnd(N, K, seed=2) / K ** 0.5
    xd = x.double()
    g, b = _rnd(K, seed=4), _rnd(K, seed=5)
    if pro == "ln3":
        E = K // 3
        vv = _rnd(B, E, H, W, seed=6)
        xin = torch.cat([O.ln_chan(xd[:, i * E:(i + 1) * E], g[i * E:(i + 1) * E].double(), b[i * E:(i + 1) * E].double()) * vv.double()
                         for i in range(3)], 1)
        full = dev(torch.cat([x, vv], 1))
        xs = full[:, :K]
        mk = lambda st: {"ln3_gate": (st, dev(g), dev(b), full[:, K:])}
        given = ops.chan_stats(xs, groups=3)
    else:
        x1 = _rnd(B, K, H, W, seed=6)
        xin = O.ln_chan(xd, g.double(), b.double()) * x1.double() + x1.double()
        xs = dev(x)
        mk = lambda st: {"ln_muladd": (st, dev(g), dev(b), dev(x1))}
        given = ops.chan_stats(xs)
    r = _rnd(B, N, H, W, seed=7)
    ref = torch.nn.functional.conv2d(xin, w.double().view(N, K, 1, 1)) + r.double()
    wc = ops.WeightCache()
    own = ops.conv1x1(xs, dev(w), res=dev(r), cache=(wc, "t"), **mk(None))
    launched = ops.conv1x1(xs, dev(w), res=dev(r), cache=(wc, "t"), **mk(given))
    e_own, e_l = rel_rms(own.cpu(), ref), rel_rms(launched.cpu(), ref)
    assert e_own < 2e-6 and e_own < 1.5 * e_l + 2e-8, (e_own, e_l)
    assert rel_rms(own.cpu(), launched.cpu().double()) < 5e-7
    assert torch.equal(own, ops.conv1x1(xs, dev(w), res=dev(r), cache=(wc, "t"), **mk(None)))


@pytest.mark.parametrize("C,N,H,W", [(86, 32, 24, 40), (345, 128, 16, 24), (64, 64, 46, 40), (43, 16, 8, 35), (172, 64, 40, 72),
                                     (129, 48, 16, 136), (32, 32, 736, 1280)])
def test_ffn_tail_fused_equals_reference(A, C, N, H, W):
    """fdn_ffn_tail (gate + project_out + residual + statistics in one launch, both kernel forms) against fp64: odd widths,
    odd channel counts (the gate branch of a pair then reads two different planes), partial 64-column tiles, bf16-storage input."""
    import ctypes
    import fdn_hip
    from fdn_hip import ops
    B = 2 if H * W < 100000 else 1
    y, wd, w, res = _rnd(B, C, H, W, seed=1), _rnd(2 * C, 1, 3, 3, seed=2) * 0.3, _rnd(N, C, seed=3) / C ** 0.5, _rnd(B, N, H, W, seed=4)
    F = torch.nn.functional
    a, g = F.conv2d(y.double(), wd.double(), padding=1, groups=C).chunk(2, 1)
    ref = F.conv2d(F.gelu(a) * g, w.double().view(N, C, 1, 1)) + res.double()
    yd, wdd, wdv, rd = dev(y), dev(wd), dev(w), dev(res)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    forms = [0] + ([1] if N <= 64 and W % 4 == 0 else [])
    for form in forms:
        out = torch.empty(B, N, H, W, device="cuda:0")
        st = torch.empty(B, 1, 2, H * W, device="cuda:0")
        rc = fdn_hip.lib().fdn_ffn_tail(p(yd), p(wdd), p(wdv), p(rd), p(out), p(st), B, C, N, H, W, 0, form, fdn_hip.stream())
        assert rc == 0, (form, rc)
        assert rel_rms(out.cpu(), ref) < 3e-6, form
        assert rel_rms(st.cpu().view(B, 2, H, W)[:, 0], ref.mean(1)) < 1e-5, form
        assert rel_rms(st.cpu().view(B, 2, H, W)[:, 1], 1 / torch.sqrt(ref.var(1, unbiased=False) + 1e-5)) < 1e-5, form
    for mode in ("split", None):
        assert rel_rms(ops.ffn_tail(yd, wdd, wdv, res=rd, mode=mode).cpu(), ref) < 3e-6, mode
    if 1 in forms:                                             # bf16-storage input: reads exactly the stored values
        yb = yd.to(torch.bfloat16)
        got = ops.ffn_tail(yb, wdd, wdv, res=rd, want_stats=True, mode="sw")
        want = ops.ffn_tail(yb.float(), wdd, wdv, res=rd, want_stats=True, mode="sw")
        assert torch.equal(got, want) and torch.equal(got._fdn_stats, want._fdn_stats)
    if N > 64 or W % 4:
        with pytest.raises(fdn_hip.FdnHipError):
            ops.ffn_tail(yd, wdd, wdv, res=rd, mode="sw")


@pytest.mark.parametrize("Cin,Cout,H,W,stride", [(64, 32, 24, 40, 1), (3, 32, 16, 35, 1), (12, 12, 9, 21, 1), (32, 3, 16, 24, 1),
                                                 (128, 64, 8, 16, 1), (64, 128, 8, 24, 1), (12, 24, 18, 22, 2), (24, 48, 9, 21, 2),
                                                 (5, 7, 11, 13, 2)])
def test_conv3x3_paths(A, Cin, Cout, H, W, stride):
    """LDS-weight direct kernel (Cout <= 64, stride 1 and 2), MFMA implicit GEMM (wider / slice too big for LDS)."""
    from fdn_hip import ops
    x, w, b = _rnd(2, Cin, H, W, seed=1), _rnd(Cout, Cin, 3, 3, seed=2) / (3 * Cin ** 0.5), _rnd(Cout, seed=3)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1, stride=stride)
    assert rel_rms(ops.conv2d(dev(x), dev(w), dev(b), pad=1, stride=stride).cpu(), ref) < 2e-6


@pytest.mark.parametrize("Cin,Cout,H,W,act,res_before", [(16, 32, 8, 32, 0, False), (64, 32, 37, 70, 2, True), (128, 64, 19, 33, 0, False),
                                                          (24, 64, 5, 100, 2, False), (32, 64, 64, 64, 0, True), (64, 128, 9, 40, 1, True),
                                                          (24, 24, 21, 45, 1, False), (48, 48, 17, 33, 2, True), (16, 17, 9, 31, 0, False),
                                                          (32, 100, 20, 36, 1, True)])
def test_conv3x3_tiled_options(A, Cin, Cout, H, W, act, res_before):
    """LDS-tiled MFMA 3x3 (Cin % 8 == 0, Cout >= 16; three-way bf16 split of both operands on the bf16 matrix pipe, fp32-exact to
    rounding): partial tiles on both axes, channel groups that end inside a 32-channel tile (24, 48, 17, 100: nothing may be
    written past the tensor - the batch of 2 puts the next image's planes right behind), bias, activation, residual before /
    after it."""
    from fdn_hip import ops
    x, w, b = _rnd(2, Cin, H, W, seed=1), _rnd(Cout, Cin, 3, 3, seed=2) / (3 * Cin ** 0.5), _rnd(Cout, seed=3)
    res = _rnd(2, Cout, H, W, seed=4)
    y = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    if res_before:
        y = y + res.double()
    if act == 2:
        y = torch.relu(y)
    elif act == 1:
        y = torch.nn.functional.leaky_relu(y, 0.1)
    if not res_before:
        y = y + res.double()
    got = ops.conv2d(dev(x), dev(w), dev(b), pad=1, act=act, res=dev(res), res_before_act=res_before)
    assert rel_rms(got.cpu(), y) < 2e-6
    assert rel_rms(got.cpu()[1, :1], y[1, :1]) < 2e-6           # (the plane right behind image 0's last channel)
    plain = ops.conv2d(dev(x), dev(w), None, pad=1)
    assert rel_rms(plain.cpu(), torch.nn.functional.conv2d(x.double(), w.double(), padding=1)) < 2e-6


@pytest.mark.parametrize("Cin,Cout,H,W", [(24, 12, 9, 13), (48, 24, 8, 16), (6, 5, 7, 9), (24, 12, 46, 80), (24, 20, 5, 77), (48, 24, 23, 40)])
def test_conv_transpose4x4s2(A, Cin, Cout, H, W):
    from fdn_hip import ops
    x, w, b = _rnd(2, Cin, H, W, seed=1), _rnd(Cin, Cout, 4, 4, seed=2) / (4 * Cin ** 0.5), _rnd(Cout, seed=3)
    ref = torch.nn.functional.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    assert rel_rms(ops.conv_transpose4x4s2(dev(x), dev(w), dev(b), 0).cpu(), ref) < 2e-6


def test_fdsa_out_level1_equals_fallback(A):
    """The register-resident FDSA tail and the statistics + GEMM fallback agree to rounding."""
    from fdn_hip import ops
    E, C, H, W, B = 38, 32, 24, 40, 2
    o, w, g, b, res = _rnd(B, 4 * E, H, W, seed=1), _rnd(C, 3 * E, seed=2) / 10, _rnd(3 * E, seed=3), _rnd(3 * E, seed=4), _rnd(B, C, H, W, seed=5)
    od = dev(o)
    fast = ops.fdsa_out(od, dev(w), dev(g), dev(b), res=dev(res), want_stats=True)
    assert fast is not None
    slow = ops.conv1x1(od[:, :3 * E], dev(w), ln3_gate=(ops.chan_stats(od[:, :3 * E], groups=3), dev(g), dev(b), od[:, 3 * E:]),
                       res=dev(res), want_stats=True)
    assert rel_rms(fast.cpu(), slow.cpu()) < 1e-6
    assert rel_rms(fast._fdn_stats.cpu(), slow._fdn_stats.cpu()) < 1e-5


@pytest.mark.parametrize("H,W", [(738, 1282), (370, 642), (736, 1280), (1088, 1920), (18, 26)])
def test_full_image_fft_sizes(A, H, W):
    """rfft2 -> irfft2 round trip and forward parity with torch.fft (fp64) at the real sizes: 720p
    (2^5*23 x 2^8*5), 1080p (2^6*17 x 2^7*15), fourier_fuse's 738 x 1282 (641 prime) and 370 x 642."""
    from fdn_hip import ops
    x = _rnd(1, 2, H, W, seed=H + W)
    z = ops.rfft_rows(dev(x))
    mag, ang = ops.fft_cols_fwd(z, True, True, rd_before=False, fix_real=True)
    ref = torch.fft.rfft2(x.double())
    assert rel_rms(mag.cpu(), ref.abs()) < 3e-6
    zz = ops.fft_cols_inv_polar(mag, ang, H, W // 2 + 1)
    back = ops.irfft_rows(zz, H, W, 2.0 / (H * W))
    assert rel_rms(back.cpu(), x) < 1e-5


def test_baseline_size_blocks_vs_oracle(A):
    """BASELINE config-2 size (736 x 1280, one image): FDSA and FDFFN blocks against the CPU oracle."""
    for name, cls, fn in (("fdsa_c32", A.FDSA, O.fdsa), ("fdffn_c32", A.FDFFN, O.fdffn)):
        sd = fixture_weights(name, fixture(name)["shapes"])
        m = load(cls(32), sd)
        x = _rnd(1, 32, 736, 1280, seed=9)
        with torch.no_grad():
            got = m(dev(x)).cpu()
            ref = fn(x, {"." + k: v for k, v in sd.items()}, "")
        assert O.psnr(got, ref, peak=float(ref.abs().max())) > 110.0, name


def test_baseline_size_batch_independence(A):
    """Config-2 size: sharding the batch changes nothing, bit for bit (what the multi-GPU path relies on)."""
    m = load(A.FDN(), fdn_weights(tame=0.03))
    x = dev(torch.rand(2, 3, 736, 1280, generator=torch.Generator().manual_seed(11)))
    r = dev(torch.tensor([[0.45], [0.65]]))
    with torch.no_grad():
        full = m(x, ratio_i=r)[0]
        one = m(x[1:2].contiguous(), ratio_i=r[1:2].contiguous())[0]
    assert torch.isfinite(full).all()
    assert torch.equal(full[1:2], one)


MULTISTREAM = pytest.mark.xfail(strict=False, reason="MI355X / ROCm 7.2: a kernel issuing v_mfma_f32_32x32x16_bf16 corrupts kernels of "
                                "other HIP streams sharing the GPU (tools/cross_stream_probe.py, DESIGN.md 4.7); the product path runs one stream")


@MULTISTREAM
def test_forward_streams_bit_identical(A, monkeypatch):
    """Splitting the batch over HIP streams (fdn_hip.pipeline, experiments only: FDN_HIP_ALLOW_MULTISTREAM=1) returns exactly the
    single-stream result - when the platform issue named above does not strike (at this size the sub-batches rarely overlap)."""
    from basicsr.models.archs.LPNet_arch import I_predict_net
    from fdn_hip.pipeline import MULTISTREAM_ENV, forward_streams
    monkeypatch.setenv(MULTISTREAM_ENV, "1")
    net = load(A.FDN(), fdn_weights(tame=0.03))
    lp = load(I_predict_net(), lpnet_weights())
    x = dev(torch.rand(4, 3, 64, 96, generator=torch.Generator().manual_seed(21)))
    one = forward_streams(net, lp, x, 1)
    two = forward_streams(net, lp, x, 2)
    torch.cuda.synchronize()
    assert torch.equal(one, two)


def test_forward_streams_fp32_pipe_bit_identical(A, monkeypatch):
    """STRICT counterpart and A/B of the finding above: with fdn_hip.set_matrix_pipe("f32") no kernel issues
    v_mfma_f32_32x32x16_bf16, and the three-stream forward at a size where the sub-batches DO overlap (6 x 256 x 256: 7-8 of 8 runs
    wrong on the bf16 pipe) must equal the one-stream forward bit for bit, every time.  This is the strict test of the multi-stream
    host logic (stream pool, WeightCache events, record_stream), and the bisection that pins the corruption on the bf16 MFMA."""
    import fdn_hip
    from basicsr.models.archs.LPNet_arch import I_predict_net
    from fdn_hip.pipeline import MULTISTREAM_ENV, forward_streams
    monkeypatch.setenv(MULTISTREAM_ENV, "1")
    net = load(A.FDN(), fdn_weights(tame=0.03))
    lp = load(I_predict_net(), lpnet_weights())
    x = dev(torch.rand(6, 3, 256, 256, generator=torch.Generator().manual_seed(22)))
    fdn_hip.set_matrix_pipe("f32")
    try:
        one = forward_streams(net, lp, x, 1).clone()
        torch.cuda.synchronize()
        bad = 0
        for _ in range(6):
            three = forward_streams(net, lp, x, 3)
            torch.cuda.synchronize()
            bad += int(not torch.equal(one, three))
    finally:
        fdn_hip.set_matrix_pipe("bf16")
    bf = forward_streams(net, lp, x, 1)
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} of 6 three-stream forwards on the fp32 matrix pipe differ from the one-stream forward"
    # the two matrix pipes are two fp32 evaluations: they agree to rounding in every block test; end to end a rounding-sized difference
    # tips the few ill-conditioned spots of a frame either way (87.6 dB over these six frames, measured; tests/test_gpu_configs.py
    # holds the conditioning-aware bound), so only gross disagreement is an error here
    assert O.psnr(bf.cpu(), one.cpu()) > 70.0


def test_f32_matrix_pipe_launches_no_bf16_mfma_kernel(A):
    """What fdn_set_matrix_pipe(1) promises (include/fdn_hip.h): no kernel that issues a bf16 MFMA is launched - for EVERY conv
    shape of both models (Downsample 64 -> 128 used to slip through: a 3 x 3 conv with no direct-kernel form reached the split-bf16 kernel
    ungated).  Every launcher of such a kernel counts (fdn_bf16_mfma_launches); in 'f32' mode the count must stand still over whole
    forwards of FDN, FDN_lolv1 and LPNet and over direct calls of the wide 3 x 3 shapes; in the default mode it must move."""
    import fdn_hip
    from fdn_hip import ops
    from basicsr.models.archs.LPNet_arch import I_predict_net
    from basicsr.models.archs.fdnlol24_arch import FDN_lolv1
    from common import lolv1_weights
    net = load(A.FDN(), fdn_weights(tame=0.03))
    net24 = load(FDN_lolv1(), lolv1_weights(tame=0.03))
    lp = load(I_predict_net(), lpnet_weights())
    x = dev(torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(23)))
    g = torch.Generator().manual_seed(24)

    def everything():
        with torch.no_grad():
            r = lp(x)
            y = net(x, ratio_i=r)[0]
            y24 = net24(x, ratio_i=r)[0]
            for cin, cout in ((64, 128), (128, 64), (16, 96), (48, 96), (64, 32)):
                ops.conv2d(dev(torch.randn(1, cin, 24, 40, generator=g)), dev(torch.randn(cout, cin, 3, 3, generator=g)), pad=1)
        torch.cuda.synchronize()
        return y, y24
    n0 = fdn_hip.bf16_mfma_launches()
    y_bf, y24_bf = everything()
    n1 = fdn_hip.bf16_mfma_launches()
    assert n1 > n0 + 100, (n0, n1)
    fdn_hip.set_matrix_pipe("f32")
    try:
        y_f32, y24_f32 = everything()
        n2 = fdn_hip.bf16_mfma_launches()
    finally:
        fdn_hip.set_matrix_pipe("bf16")
    assert n2 == n1, f"{n2 - n1} bf16-MFMA kernels were launched in 'f32' mode"
    assert O.psnr(y_bf.cpu(), y_f32.cpu()) > 70.0 and O.psnr(y24_bf.cpu(), y24_f32.cpu()) > 70.0


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_tiling_kernels_bit_exact(A, case):
    """fdn_tiles_gather / fdn_tiles_merge against the reference-generated grids fixture: index work and an ordered fp32
    sum, so bit-exact."""
    import os
    from common import GOLDEN
    from fdn_hip import tiling
    z = np.load(os.path.join(GOLDEN, "grids.npz"))
    x, (ch, cw) = torch.from_numpy(z[case + "_x"]), z[case + "_crop"]
    assert tiling.tile_origins(x.shape[2], x.shape[3], int(ch), int(cw)) == [tuple(t) for t in z[case + "_idx"].tolist()]
    tiles, ij = tiling.split(dev(x), int(ch), int(cw))
    assert torch.equal(tiles.cpu(), torch.from_numpy(z[case + "_tiles"]))
    merged = tiling.merge(dev(torch.from_numpy(z[case + "_outs"])), ij, x.shape[2], x.shape[3])
    assert torch.equal(merged.cpu(), torch.from_numpy(z[case + "_merged"]))


def test_forward_tiled_matches_oracle(A):
    """LPNet -> FDN over overlapping 64x64 tiles of a 96x128 image, merged: HIP path against the oracle doing the same."""
    from basicsr.models.archs.LPNet_arch import I_predict_net
    from fdn_hip import tiling
    net = load(A.FDN(), fdn_weights(tame=0.03))
    lp = load(I_predict_net(), lpnet_weights())
    g = torch.Generator().manual_seed(9)
    x = torch.rand(1, 3, 96, 128, generator=g)
    got = tiling.forward_tiled(net, lp, dev(x), 64, 64)
    tiles, idx = O.grids_split(x, 64, 64)
    with torch.no_grad():
        outs = O.fdn_forward(fdn_weights(tame=0.03), tiles, O.lpnet_forward(lpnet_weights(), tiles))[0]
    ref = O.grids_merge(outs, idx, 96, 128)
    assert O.psnr(got.cpu(), ref) > 95.0


def test_metrics_kernels(A):
    """fdn_sse_max / fdn_ssim3d against the reference-generated values: PSNR is an fp64 reduction (1e-9 dB), the 3-D SSIM
    window is applied as three separable fp32 passes instead of one 1331-tap fp32 convolution (2e-5)."""
    from test_oracle_golden import _metric_cases
    from fdn_hip import metrics
    for name, x, y, ref in _metric_cases():
        assert abs(metrics.calculate_psnr(dev(x), dev(y), ref["crop_border"]) - ref["psnr"]) < 1e-9, name
        assert abs(metrics.calculate_ssim(dev(x), dev(y), ref["crop_border"]) - ref["ssim"]) < 2e-5, name
    g = torch.Generator().manual_seed(3)
    a = torch.rand(1, 3, 736, 1280, generator=g)
    b = (a + 0.02 * torch.randn(a.shape, generator=g)).clamp(0, 1)
    assert abs(metrics.calculate_psnr(dev(a), dev(b)) - O.calculate_psnr(a[0], b[0])) < 1e-9
    assert metrics.calculate_psnr(dev(a), dev(a)) == float("inf")
    # the remaining branches: Y channel (PSNR and _ssim_cly) and the 2-D SSIM, float64 on the device like the reference's numpy path
    for name, x, y, ref in _metric_cases("metrics2d"):
        b = ref["crop_border"]
        assert abs(metrics.calculate_ssim(dev(x), dev(y), b, ssim3d=False) - ref["ssim_2d"]) < 1e-10, name
        if "psnr_y" in ref:
            assert torch.equal(metrics.to_y_channel(dev(x)).cpu()[0], ref["ych"]), name
            assert abs(metrics.calculate_psnr(dev(x), dev(y), b, test_y_channel=True) - ref["psnr_y"]) < 1e-5, name
            assert abs(metrics.calculate_ssim(dev(x), dev(y), b, test_y_channel=True) - ref["ssim_y"]) < 1e-10, name


@pytest.mark.parametrize("K,N,H,W", [(32, 152, 736, 1280), (32, 86, 736, 1280), (64, 304, 368, 640), (64, 172, 368, 640),
                                     (128, 612, 184, 320), (128, 345, 184, 320), (86, 32, 736, 1280), (32, 32, 736, 1280),
                                     (172, 64, 368, 640), (345, 128, 184, 320), (12, 12, 736, 1280)])
def test_conv1x1_baseline_shapes_all_forms(A, K, N, H, W):
    """Every kernel of the conv1x1 family at the real level-1/2/3 shapes of config 2 (the persistent tile loops, the
    register-strip refill and the LDS tables only come into play at full size): plain, bias, folded LayerNorm
    prologue, residual + statistics epilogue, against fp64."""
    from fdn_hip import ops
    x, w = _rnd(1, K, H, W, seed=K + N), _rnd(N, K, seed=K * N) / K ** 0.5
    b, res, g, be = _rnd(N, seed=1), _rnd(1, N, H, W, seed=2), _rnd(K, seed=3), _rnd(K, seed=4)
    xd, wd = dev(x), dev(w)
    ref = torch.einsum("nk,bkhw->bnhw", w.double(), x.double())
    assert rel_rms(ops.conv1x1(xd, wd).cpu(), ref) < 1e-6
    assert rel_rms(ops.conv1x1(xd, wd, dev(b)).cpu(), ref + b.double().view(1, -1, 1, 1)) < 1e-6
    xn = (x.double() - x.double().mean(1, keepdim=True)) / torch.sqrt(x.double().var(1, unbiased=False, keepdim=True) + 1e-5)
    xn = xn * g.double().view(1, -1, 1, 1) + be.double().view(1, -1, 1, 1)
    ref_ln = torch.einsum("nk,bkhw->bnhw", w.double(), xn)
    assert rel_rms(ops.conv1x1(xd, wd, ln=(ops.chan_stats(xd), dev(g), dev(be))).cpu(), ref_ln) < 2e-6
    # the activation copies of the epilogues (resolved once per tile, separate code from the plain copy): act before res
    lk = torch.nn.functional.leaky_relu(ref + b.double().view(1, -1, 1, 1), 0.1)
    assert rel_rms(ops.conv1x1(xd, wd, dev(b), act=1).cpu(), lk) < 1e-6
    assert rel_rms(ops.conv1x1(xd, wd, dev(b), act=1, res=dev(res)).cpu(), lk + res.double()) < 1e-6
    assert rel_rms(ops.conv1x1(xd, wd, act=3, ln=(ops.chan_stats(xd), dev(g), dev(be))).cpu(), torch.sigmoid(ref_ln)) < 2e-6
    got = ops.conv1x1(xd, wd, res=dev(res), want_stats=N <= 160)
    tot = ref + res.double()
    assert rel_rms(got.cpu(), tot) < 1e-6
    if N <= 160:
        st = got._fdn_stats.cpu().view(2, H, W).double()
        assert (st[0] - tot.mean(1)[0]).abs().max() < 1e-5
        assert rel_rms(st[1], 1.0 / torch.sqrt(tot.var(1, unbiased=False)[0] + 1e-5)) < 1e-5


@pytest.mark.parametrize("H,W", [(16, 24), (9, 14), (23, 40), (8, 8)])
def test_resample_bilinear_matches_torch(A, H, W):
    """Both bilinear modes (quad-per-thread kernels where the width allows, scalar kernel otherwise) against
    F.interpolate(align_corners=False) in float64: x2 up (FDN_arch.py Upsample) and x0.5 down."""
    from fdn_hip import ops
    x = _rnd(2, 3, H, W, seed=H * W)
    up = ops.resample(dev(x), ops.RS_BILINEAR_X2)
    ref = torch.nn.functional.interpolate(x.double(), scale_factor=2, mode="bilinear", align_corners=False)
    assert rel_rms(up.cpu(), ref) < 1e-6 and (up.cpu().double() - ref).abs().max() < 1e-5
    nn_ = ops.resample(dev(x), ops.RS_NEAREST_X2)
    assert torch.equal(nn_.cpu(), torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest"))
    if H % 2 == 0 and W % 2 == 0:
        dn = ops.resample(dev(x), ops.RS_BILINEAR_HALF)
        refd = torch.nn.functional.interpolate(x.double(), scale_factor=0.5, mode="bilinear", align_corners=False)
        assert rel_rms(dn.cpu(), refd) < 1e-6


@pytest.mark.parametrize("C,H,Wf", [(12, 38, 21), (24, 17, 33), (48, 9, 161), (12, 738, 642)])
def test_spectral_mlp2_matches_float64_and_the_four_convs(C, H, Wf):
    """fdn_spectral_mlp2 (ABI 12): process1(mag), process2(pha) of a FreBlock / fourier_fuse (FDN_arch.py:93-94, :142-143) in one launch, in place,
    against float64 and against the four fdn_conv1x1 launches it replaces (the last shape: fourier_fuse's 738 x 642 level-1 map, ragged last block)."""
    from fdn_hip import ACT_LEAKY, ops
    B = 2
    g = torch.Generator().manual_seed(C + H)
    mag, pha = torch.rand(B, C, H, Wf, generator=g) * 3, (torch.rand(B, C, H, Wf, generator=g) * 2 - 1) * 3.14159
    W = [torch.randn(C, C, generator=g) / C ** 0.5 for _ in range(4)]
    b = [torch.randn(C, generator=g) * 0.1 for _ in range(4)]

    def ref(t, w1, b1, w2, b2):
        h = torch.einsum("jc,bchw->bjhw", w1.double(), t.double()) + b1.double()[None, :, None, None]
        h = torch.where(h > 0, h, 0.1 * h)
        return torch.einsum("kj,bjhw->bkhw", w2.double(), h) + b2.double()[None, :, None, None]
    rm, rp = ref(mag, W[0], b[0], W[1], b[1]), ref(pha, W[2], b[2], W[3], b[3])
    dm, dp = dev(mag).clone(), dev(pha).clone()
    ops.spectral_mlp2(dm, dp, dev(W[0]), dev(b[0]), dev(W[1]), dev(b[1]), dev(W[2]), dev(b[2]), dev(W[3]), dev(b[3]), slope=0.1)
    assert rel_rms(dm.cpu(), rm) < 5e-7 and rel_rms(dp.cpu(), rp) < 5e-7
    cm = ops.conv1x1(ops.conv1x1(dev(mag), dev(W[0]), dev(b[0]), act=ACT_LEAKY), dev(W[1]), dev(b[1]))
    cp = ops.conv1x1(ops.conv1x1(dev(pha), dev(W[2]), dev(b[2]), act=ACT_LEAKY), dev(W[3]), dev(b[3]))
    assert rel_rms(dm.cpu(), cm.cpu()) < 1e-6 and rel_rms(dp.cpu(), cp.cpu()) < 1e-6
