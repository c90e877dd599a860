"""CPU: the oracle (oracle/fdn_oracle.py) against the fixtures produced by the reference itself.

This is what pins the oracle (the reference ships no tests/golden vectors of its own)."""
import pytest
import torch

import fdn_oracle as O
from common import assert_close_cond, fdn_weights, fixture, fixture_weights, lolv1_weights, lpnet_weights, rel_rms

F32, F64 = torch.float32, torch.float64


def _both(fn, sd, *xs):
    """Run an oracle function in fp32 and fp64."""
    with torch.no_grad():
        y32 = fn(O.cast_params(sd, F32), *[x.to(F32) if x is not None else None for x in xs])
        y64 = fn(O.cast_params(sd, F64), *[x.to(F64) if x is not None else None for x in xs])
    return y32, y64


@pytest.mark.parametrize("name", ["fdsa_c32", "fdsa_c64", "fdsa_c128"])
def test_fdsa(name):
    fx = fixture(name)
    sd = fixture_weights(name, fx["shapes"])
    y32, y64 = _both(lambda P, x: O.fdsa(x, P, ""), {"." + k: v for k, v in sd.items()}, fx["x"])
    assert_close_cond(y32, fx["y"], y64, name)


@pytest.mark.parametrize("name", ["fdffn_c32", "fdffn_c64", "fdffn_c128"])
def test_fdffn(name):
    fx = fixture(name)
    sd = fixture_weights(name, fx["shapes"])
    y32, y64 = _both(lambda P, x: O.fdffn(x, P, ""), {"." + k: v for k, v in sd.items()}, fx["x"])
    assert_close_cond(y32, fx["y"], y64, name)
    assert rel_rms(y32, fx["y"]) < 1e-5


@pytest.mark.parametrize("name", ["fcaffn_c32_32x32", "fcaffn_c32_24x40", "fcaffn_c64_46x40", "fcaffn_c128_16x16"])
def test_fcaffn(name):
    fx = fixture(name)
    sd = fixture_weights(name, fx["shapes"])
    y32, y64 = _both(lambda P, x, a, p, i: O.fcaffn(x, a, p, i, P, ""), {"." + k: v for k, v in sd.items()},
                     fx["x"], fx["amp"], fx["pha"], fx["img"])
    assert_close_cond(y32, fx["y"], y64, name)


@pytest.mark.parametrize("name,light", [("tblock_enc_c32", True), ("tblock_dec_c32", False)])
def test_tblock(name, light):
    fx = fixture(name)
    sd = fixture_weights(name, fx["shapes"], po_scale=float(fx["po_scale"]))
    y32, y64 = _both(lambda P, x, a, p, i: O.tblock(x, a, p, i, P, "", True, light), {"." + k: v for k, v in sd.items()},
                     fx["x"], fx["amp"], fx["pha"], fx["img"])
    assert_close_cond(y32, fx["y"], y64, name)


def test_fuse():
    fx = fixture("fuse_n32")
    sd = fixture_weights("fuse_n32", fx["shapes"])
    y32, y64 = _both(lambda P, e, d: O.fuse(e, d, P, ""), {"." + k: v for k, v in sd.items()}, fx["enc"], fx["dnc"])
    assert_close_cond(y32, fx["y"], y64, "fuse")


def test_resample_and_embed():
    for name, fn in (("downsample_c32", O.downsample), ("upsample_c64", O.upsample)):
        fx = fixture(name)
        sd = fixture_weights(name, fx["shapes"])
        y32, y64 = _both(lambda P, x: fn(x, P, ""), {"." + k: v for k, v in sd.items()}, fx["x"])
        assert_close_cond(y32, fx["y"], y64, name)
    fx = fixture("patch_embed_3_32")
    sd = fixture_weights("patch_embed_3_32", fx["shapes"])
    with torch.no_grad():
        y = torch.nn.functional.conv2d(fx["x"], sd["proj.weight"], padding=1)
    assert rel_rms(y, fx["y"]) < 1e-6


def test_mar_pieces():
    fx = fixture("freblock_c12")
    sd = fixture_weights("freblock_c12", fx["shapes"])
    y32, y64 = _both(lambda P, x: O.freblock(x, P, ""), {"." + k: v for k, v in sd.items()}, fx["x"])
    assert_close_cond(y32, fx["y"], y64, "freblock")
    fx = fixture("fourier_fuse_84_12")
    sd = fixture_weights("fourier_fuse_84_12", fx["shapes"])
    y32, y64 = _both(lambda P, a, b, c: O.fourier_fuse(a, b, c, P, ""), {"." + k: v for k, v in sd.items()},
                     fx["x1"], fx["x2"], fx["x4"])
    assert_close_cond(y32, fx["y"], y64, "fourier_fuse")


def test_mar_full():
    fx = fixture("mar_full")
    sd = fixture_weights("mar_full", fx["shapes"])
    with torch.no_grad():
        y3, y2, y1 = O.mar(fx["x"], fx["ratio"].view(-1, 1, 1, 1), {"net_a." + k: v for k, v in sd.items()}, "net_a")
    for got, key in ((y3, "y3"), (y2, "y2"), (y1, "y1")):
        assert O.psnr(got, fx[key]) > 110.0, key


def test_lpnet_real_weights():
    fx = fixture("lpnet_real")
    P = lpnet_weights()
    with torch.no_grad():
        y = O.lpnet_forward(P, fx["x"])
    assert torch.allclose(y, fx["y"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", ["fdn_tamed_64", "fdn_tamed_96x160", "fdn_tamed_96x160_wc"])
def test_fdn_end_to_end_tamed(name):
    fx = fixture(name)
    sd = fdn_weights(tame=float(fx["tame"]))
    with torch.no_grad():
        out = O.fdn_forward(sd, fx["x"], fx["ratio"])
    for got, key, floor in zip(out, ("y", "q1", "q2", "q3"), (100.0, 110.0, 110.0, 110.0)):
        p = O.psnr(got, fx[key])
        assert p > floor, f"{name}.{key}: PSNR {p:.1f} dB"


def test_config0_256_crop():
    """BASELINE.json configs[0]: the oracle on the 256 x 256 crop against the reference's own outputs (LPNet_lolblur.pth ratio,
    tamed FDN weights; tests/golden/make_golden_configs.py)."""
    from common import lpnet_weights
    fx = fixture("fdn_tamed_256")
    with torch.no_grad():
        ratio = O.lpnet_forward(lpnet_weights(), fx["x"])
        out = O.fdn_forward(fdn_weights(tame=float(fx["tame"])), fx["x"], ratio)
    assert torch.allclose(ratio, fx["ratio"], rtol=0, atol=2e-6)
    for got, key in zip(out, ("y", "q1", "q2", "q3")):
        p = O.psnr(got, fx[key])
        assert p > 100.0, f"fdn_tamed_256.{key}: PSNR {p:.1f} dB"


def test_harness_u8():
    fx = fixture("harness_u8")
    img = fx["img"].numpy()
    padded, h, w = O.harness_pre(img)
    assert torch.equal(padded, fx["padded"])
    with torch.no_grad():
        ratio = O.lpnet_forward(lpnet_weights(), padded)
        assert torch.allclose(ratio, fx["ratio"], atol=2e-6)
        res = O.fdn_forward(fdn_weights(tame=float(fx["tame"])), padded, ratio)[0]
    out = O.harness_post(res, h, w)
    diff = (out.astype(int) - fx["out_u8"].numpy().astype(int))
    assert abs(diff).max() <= 1 and (diff != 0).mean() < 1e-3


# ---- LOL-v1 variant (FDN_lolv1, dim 24) -----------------------------------------
@pytest.mark.parametrize("name", ["lolv1_fdsa_c24", "lolv1_fdsa_c48", "lolv1_fdsa_c96"])
def test_lolv1_fdsa(name):
    fx = fixture(name)
    sd = fixture_weights(name, fx["shapes"])
    y32, y64 = _both(lambda P, x: O.fdsa(x, P, ""), {"." + k: v for k, v in sd.items()}, fx["x"])
    assert_close_cond(y32, fx["y"], y64, name)


@pytest.mark.parametrize("name", ["lolv1_fdffn_c24", "lolv1_fdffn_c48", "lolv1_fdffn_c96"])
def test_lolv1_fdffn(name):
    fx = fixture(name)
    sd = fixture_weights(name, fx["shapes"])
    y32, y64 = _both(lambda P, x: O.fdffn(x, P, ""), {"." + k: v for k, v in sd.items()}, fx["x"])
    assert_close_cond(y32, fx["y"], y64, name)


def test_lolv1_processblock_and_mar():
    fx = fixture("lolv1_processblock_c12")
    sd = fixture_weights("lolv1_processblock_c12", fx["shapes"])
    y32, y64 = _both(lambda P, x: O.processblock(x, P, "", cat=True), {"." + k: v for k, v in sd.items()}, fx["x"])
    assert_close_cond(y32, fx["y"], y64, "lolv1 processblock")
    fx = fixture("lolv1_mar_full")
    sd = fixture_weights("lolv1_mar_full", fx["shapes"])
    with torch.no_grad():
        y3, y2, y1 = O.mar(fx["x"], fx["ratio"].view(-1, 1, 1, 1), {"net_a." + k: v for k, v in sd.items()}, "net_a", cat=True)
    for got, key in ((y3, "y3"), (y2, "y2"), (y1, "y1")):
        assert O.psnr(got, fx[key]) > 110.0, key


@pytest.mark.parametrize("name", ["lolv1_tamed_64", "lolv1_tamed_96x160"])
def test_lolv1_end_to_end_tamed(name):
    fx = fixture(name)
    with torch.no_grad():
        out = O.fdn_lolv1_forward(lolv1_weights(tame=float(fx["tame"])), fx["x"], fx["ratio"])
    assert all(o is out[0] for o in out)
    p = O.psnr(out[0], fx["y"])
    assert p > 100.0, f"{name}: PSNR {p:.1f} dB"


def test_lolv1_harness_u8():
    """inference_fdn_lolv1.py:40-66: ratio = mean(Grayscale(padded)) / LPNet_lolv1(padded)."""
    fx = fixture("lolv1_harness_u8")
    padded, h, w = O.harness_pre(fx["img"].numpy())
    assert torch.equal(padded, fx["padded"])
    with torch.no_grad():
        lpr = O.lpnet_forward(lpnet_weights("lolv1"), padded)
        assert torch.allclose(lpr, fx["lp_ratio"], atol=2e-6)
        ratio = O.lolv1_ratio(padded, lpr)
        assert torch.allclose(ratio, fx["ratio"], rtol=2e-5)
        res = O.fdn_lolv1_forward(lolv1_weights(tame=float(fx["tame"])), padded, ratio)[0]
    # ratio = 1.64 here (LOL-v1's mean/LPNet convention) against 0.3-0.8 in the tamed LOL-Blur fixtures: the net
    # is less well conditioned (84 dB against the reference), so a few more bytes sit next to a .5 boundary
    assert O.psnr(res[:, :, :h, :w].clamp(0, 1), fx["result"].clamp(0, 1)) > 75.0
    out = O.harness_post(res, h, w)
    diff = (out.astype(int) - fx["out_u8"].numpy().astype(int))
    assert abs(diff).max() <= 1 and (diff != 0).mean() < 1e-2


# ---- tiled inference (grids / grids_inverse) ------------------------------------------
@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_grids_against_reference_methods(case):
    """Fixture = the reference's own grids()/grids_inverse() code run on a stub model (tests/golden/make_golden_grids.py)."""
    import numpy as np
    z = np.load(__import__("os").path.join(__import__("common").GOLDEN, "grids.npz"))
    x, (ch, cw) = torch.from_numpy(z[case + "_x"]), z[case + "_crop"]
    tiles, idx = O.grids_split(x, int(ch), int(cw))
    assert [list(t) for t in idx] == z[case + "_idx"].tolist()
    assert torch.equal(tiles, torch.from_numpy(z[case + "_tiles"]))
    merged = O.grids_merge(torch.from_numpy(z[case + "_outs"]), idx, x.shape[2], x.shape[3])
    assert torch.equal(merged, torch.from_numpy(z[case + "_merged"]))


def _metric_cases(which="metrics"):
    import json
    import os
    import numpy as np
    from common import GOLDEN
    z = np.load(os.path.join(GOLDEN, which + ".npz"))
    cases = json.loads(bytes(z["cases_json"]).decode())
    return [(k, torch.from_numpy(z[k + "_x"]), torch.from_numpy(z[k + "_y"]), dict(v, ych=torch.from_numpy(z[k + "_ych"]) if k + "_ych" in z.files else None))
            for k, v in cases.items()]


def test_metrics_against_reference_functions():
    """calculate_psnr / _ssim_3d of the oracle against values produced by the reference's own functions
    (tests/golden/make_golden_metrics.py)."""
    for name, x, y, ref in _metric_cases():
        assert abs(O.calculate_psnr(x, y, ref["crop_border"]) - ref["psnr"]) < 1e-9, name
        assert abs(O.ssim_3d(x, y, ref["crop_border"]) - ref["ssim"]) < 1e-6, name


def test_metrics_2d_and_y_channel_against_reference_functions():
    """The other branches of the same functions - test_y_channel (to_y_channel, _ssim_cly) and ssim3d=False (_ssim) - against the
    values the reference's function bodies produce (tests/golden/make_golden_metrics2d.py)."""
    for name, x, y, ref in _metric_cases("metrics2d"):
        b = ref["crop_border"]
        assert abs(O.ssim_2d(x, y, b) - ref["ssim_2d"]) < 1e-12, name
        if "psnr_y" in ref:
            assert torch.equal(O.to_y_channel(x)[0], ref["ych"]), name
            assert abs(O.psnr_y(x, y, b) - ref["psnr_y"]) < 1e-5, name          # (the reference averages float32 squares: summation order)
            assert abs(O.ssim_y(x, y, b) - ref["ssim_y"]) < 1e-12, name


# --------------------------------------------------------------------------------------------------------------
# replace_denormals edge cases and dark inputs (fixtures of tests/golden/make_golden_edge.py: module outputs AND the
# reference's own intermediate tensors captured by forward hooks)
# --------------------------------------------------------------------------------------------------------------
def test_edge_fdsa_regions_and_wholesale_replaced_spectra():
    import edge_cases as EC
    fx, sd = EC.fdsa_edge()
    taps = {}
    with torch.no_grad():
        y = O.fdsa(fx["x"], {"." + k: v for k, v in sd.items()}, "", taps)
    for k in ("o1", "o2", "o3"):
        EC.assert_regions_close(taps[k], fx[k], "fdsa_edge." + k, 1e-6)
        EC.assert_channels_close(taps[k], fx[k], "fdsa_edge." + k, 1e-5)
    assert rel_rms(y, fx["y"]) < 1e-6
    # the zero / -0 patches really are the all-replaced case: every bin (1e-10, 1e-10) -> an impulse of sqrt(2) * 1e-10
    r0, r1, c0, c1 = EC.REGIONS["zero"]
    assert abs(fx["o1"][0, 0, r0, c0].item() - 2 ** 0.5 * 1e-10) < 1e-15 and fx["o1"][0, 0, r0 + 1:r1, c0:c1].abs().max() < 1e-16


def test_edge_fdsa_threshold_impulses():
    import edge_cases as EC
    fx, sd = EC.fdsa_kat()
    taps = {}
    with torch.no_grad():
        y = O.fdsa(fx["x"], {"." + k: v for k, v in sd.items()}, "", taps)
    for k in ("o1", "o2", "o3"):
        EC.assert_channels_close(taps[k], fx[k], "fdsa_kat." + k, 1e-6)
    assert rel_rms(y, fx["y"]) < 1e-6


def test_edge_fdffn_fcaffn_and_dark_e2e():
    import edge_cases as EC
    fx, sd = EC.fdffn_edge()
    taps = {}
    with torch.no_grad():
        y = O.fdffn(fx["x"], {"." + k: v for k, v in sd.items()}, "", taps)
    EC.assert_regions_close(taps["mid"], fx["mid"], "fdffn_edge.mid", 1e-6)
    EC.assert_channels_close(taps["mid"], fx["mid"], "fdffn_edge.mid", 1e-5)
    assert rel_rms(y, fx["y"]) < 1e-6
    fx, sd = EC.fcaffn_edge()
    taps = {}
    with torch.no_grad():
        y = O.fcaffn(fx["x"], fx["amp"], fx["pha"], fx["img"], {"." + k: v for k, v in sd.items()}, "", taps)
    EC.assert_channels_close(taps["xi"], fx["xi"], "fcaffn_edge.xi", 1e-5)
    assert rel_rms(y, fx["y"]) < 1e-5
    fx = fixture("fdn_tamed_64_dark")
    with torch.no_grad():
        outs = O.fdn_forward(fdn_weights(tame=float(fx["tame"])), fx["x"], fx["ratio"])
    for got, key in zip(outs, ("y", "q1", "q2", "q3")):
        assert O.psnr(got, fx[key]) > 110.0, key
