"""GPU parity of the LOL-v1 variant (FDN_lolv1, dim 24: SURVEY.md section 8 (f) rank 1) against the oracle and the
reference-generated fixtures of tests/golden/make_golden_lolv1.py.  Same tolerance policy as test_gpu_parity.py.
The widths exercise the odd shapes of the kernels: E = 28/57/115 (FDSA), Hd = 64/129/259 (FDFFN), K = 24/48/96."""
import os

import pytest
import torch

import fdn_oracle as O
from common import assert_close_cond, fixture, fixture_weights, lolv1_weights, lpnet_weights

pytestmark = pytest.mark.gpu
F64 = torch.float64


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import fdn_hip
    fdn_hip.lib()   # fail loudly if the HIP extension is not built
    from basicsr.models.archs import fdnlol24_arch
    return fdnlol24_arch


def dev(t):
    return t.to("cuda:0").contiguous()


def load(mod, sd):
    mod.load_state_dict(sd, strict=True)
    return mod.to("cuda:0").eval()


def dev_u8(arr):
    return torch.from_numpy(arr).to("cuda:0").contiguous()


def truth(fn, sd, *xs):
    with torch.no_grad():
        return fn({"." + k: v.to(F64) for k, v in sd.items()}, *[x.to(F64) for x in xs])


@pytest.mark.parametrize("c", [24, 48, 96])
def test_lolv1_fdsa_fdffn(L, c):
    from basicsr.models.archs import FDN_arch as A
    for cls, tag, fn in ((A.FDSA, "fdsa", O.fdsa), (A.FDFFN, "fdffn", O.fdffn)):
        name = f"lolv1_{tag}_c{c}"
        fx = fixture(name)
        sd = fixture_weights(name, fx["shapes"])
        m = load(cls(c), sd)
        with torch.no_grad():
            got = m(dev(fx["x"]))
        assert_close_cond(got, fx["y"], truth(lambda P, x: fn(x, P, ""), sd, fx["x"]), name)


def test_lolv1_processblock_and_mar(L):
    fx = fixture("lolv1_processblock_c12")
    sd = fixture_weights("lolv1_processblock_c12", fx["shapes"])
    m = load(L.ProcessBlock(12), sd)
    with torch.no_grad():
        got = m(dev(fx["x"]))
    assert_close_cond(got, fx["y"], truth(lambda P, x: O.processblock(x, P, "", cat=True), sd, fx["x"]), "lolv1 processblock")
    fx = fixture("lolv1_mar_full")
    sd = fixture_weights("lolv1_mar_full", fx["shapes"])
    m = load(L.MAR(True), sd)
    with torch.no_grad():
        y3, y2, y1 = m(dev(fx["x"]), dev(fx["ratio"]))
    for got, key in ((y3, "y3"), (y2, "y2"), (y1, "y1")):
        p = O.psnr(got.cpu(), fx[key])
        assert p > 100.0, f"lolv1 mar {key}: PSNR {p:.1f} dB"


@pytest.mark.parametrize("name", ["lolv1_tamed_64", "lolv1_tamed_96x160"])
def test_lolv1_end_to_end_tamed(L, name):
    fx = fixture(name)
    m = load(L.FDN_lolv1(), lolv1_weights(tame=float(fx["tame"])))
    with torch.no_grad():
        out = m(dev(fx["x"]), ratio_i=dev(fx["ratio"]), device=torch.device("cuda:0"))
    assert len(out) == 4 and all(o is out[0] for o in out)                      # fdnlol24_arch.py:1031
    p = O.psnr(out[0].cpu(), fx["y"])
    assert p > 95.0, f"{name}: PSNR {p:.1f} dB"       # reference self-noise: 150 / 131 dB (selfnoise_lolv1.json)


def test_lolv1_harness_u8(L):
    """uint8 in -> uint8 out with the LOL-v1 ratio convention (inference_fdn_lolv1.py:40-66), all on the GPU."""
    from basicsr.models.archs.LPNet_arch import I_predict_net
    from fdn_hip import harness
    fx = fixture("lolv1_harness_u8")
    net = load(L.FDN_lolv1(), lolv1_weights(tame=float(fx["tame"])))
    lp = load(I_predict_net(), lpnet_weights("lolv1"))
    img = fx["img"].cuda().contiguous()
    with torch.no_grad():
        x, h, w = harness.preprocess(img, bgr=True)
        assert torch.equal(x.cpu(), fx["padded"])
        lpr = lp(x)
        assert torch.allclose(lpr.cpu(), fx["lp_ratio"], atol=5e-6)
        assert torch.allclose(harness.lolv1_ratio(x, lpr).cpu(), fx["ratio"], rtol=3e-5)
    out = harness.enhance_u8(net, lp, img, bgr=True, ratio_mode="lolv1").cpu().numpy()[0]
    diff = out.astype(int) - fx["out_u8"].numpy().astype(int)
    assert abs(diff).max() <= 1 and (diff != 0).mean() < 1e-2


def test_drivers_lolv1_and_ratio_sweep(L, tmp_path, monkeypatch):
    """The two other callers of the path as command-line drivers (inference_fdn_lolv1.py:52-64, inference_fdn_multi_r.py:52-85): synthetic
    frames and checkpoints on disk -> PNGs, equal to what fdn_hip.harness.enhance_u8 returns for the same frames."""
    import sys
    import numpy as np
    from PIL import Image
    from basicsr.models.archs.FDN_arch import FDN
    from basicsr.models.archs.LPNet_arch import I_predict_net
    from common import fdn_weights, lpnet_weights
    from fdn_hip import harness
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fdn-tip2025_amd")
    monkeypatch.syspath_prepend(pkg)
    import inference_fdn_lolv1
    import inference_fdn_multi_r
    g = torch.Generator().manual_seed(5)
    frames = [(torch.rand(40, 72, 3, generator=g) * 255).to(torch.uint8).numpy() for _ in range(3)]
    for i, f in enumerate(frames):
        Image.fromarray(f, mode="RGB").save(tmp_path / f"low{i}.png")
    torch.save({"params": lolv1_weights(tame=0.03)}, tmp_path / "fdn_lolv1.pth")
    torch.save({"params": lpnet_weights("lolv1")}, tmp_path / "lp.pth")
    torch.save({"params": fdn_weights(tame=0.03)}, tmp_path / "fdn.pth")
    # LOL-v1 driver
    monkeypatch.setattr(sys, "argv", ["x", "--fdn", str(tmp_path / "fdn_lolv1.pth"), "--lpnet", str(tmp_path / "lp.pth"),
                                      "--input", str(tmp_path / "low*.png"), "--output", str(tmp_path / "out"), "--batch", "2"])
    inference_fdn_lolv1.main()
    net = load(L.FDN_lolv1(), lolv1_weights(tame=0.03))
    lp = load(I_predict_net(), lpnet_weights("lolv1"))
    want = harness.enhance_u8(net, lp, dev_u8(np.stack(frames)), bgr=False, ratio_mode="lolv1").cpu().numpy()
    for i in range(3):
        got = np.asarray(Image.open(tmp_path / "out" / f"low{i}.png"))
        assert np.array_equal(got, want[i])
    # ratio sweep driver: 4 grid values of one frame
    monkeypatch.setattr(sys, "argv", ["x", "--fdn", str(tmp_path / "fdn.pth"), "--input", str(tmp_path / "low0.png"), "--output",
                                      str(tmp_path / "multi_r"), "--start", "0.2", "--stop", "0.6", "--step", "0.1", "--batch", "3"])
    inference_fdn_multi_r.main()
    vals = inference_fdn_multi_r.sweep_values(0.2, 0.6, 0.1)
    net32 = load(FDN(), fdn_weights(tame=0.03))
    for v in vals:
        got = np.asarray(Image.open(tmp_path / "multi_r" / inference_fdn_multi_r.output_name(v)))
        want1 = harness.enhance_u8(net32, None, dev_u8(frames[0][None]), bgr=False, ratio_mode="fixed",
                                   ratio=torch.tensor([[float(v)]], dtype=torch.float32)).cpu().numpy()[0]
        assert np.array_equal(got, want1)
    outs = [np.asarray(Image.open(tmp_path / "multi_r" / inference_fdn_multi_r.output_name(v))).astype(np.int32) for v in vals]
    assert any(np.abs(outs[0] - o).max() > 0 for o in outs[1:])          # the ratio does steer the result
