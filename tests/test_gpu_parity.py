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


@pytest.mark.parametrize("name", ["fdn_tamed_64", "fdn_tamed_96x160"])
def test_fdn_end_to_end_tamed(A, name):
    fx = fixture(name)
    m = load(A.FDN(), fdn_weights(tame=float(fx["tame"])))
    with torch.no_grad():
        out = m(dev(fx["x"]), ratio_i=dev(fx["ratio"]), device=torch.device("cuda:0"))
    for got, key, floor in zip(out, ("y", "q1", "q2", "q3"), (100.0, 100.0, 100.0, 100.0)):
        p = O.psnr(got.cpu(), fx[key])
        assert p > floor, f"{name}.{key}: PSNR {p:.1f} dB (reference self-noise: see tests/golden/selfnoise.json)"


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
