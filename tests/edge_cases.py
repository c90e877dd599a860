"""Shared description of the replace_denormals / dark-input fixtures (tests/golden/make_golden_edge.py): the weight edits
the generator applied to the reference modules and the image regions / channels that hold each special case."""
import torch

from common import fixture, fixture_weights

E32 = 38

# region name -> (row0, row1, col0, col1) of one 8x8 patch whose 1-pixel halo lies inside the special region
REGIONS = {"zero": (0, 8, 0, 8), "negzero": (24, 32, 8, 16), "const": (16, 24, 32, 40), "rand": (32, 40, 40, 48)}


def fdsa_edge():
    fx = fixture("fdsa_c32_edge")
    sd = fixture_weights("fdsa_c32_edge", fx["shapes"])
    w = sd["to_hidden.weight"]
    w[0:4] = 0.0
    w[E32 + 4:E32 + 8] = 0.0
    w[2 * E32 + 8:2 * E32 + 12] = 0.0
    return fx, sd


def fdsa_kat():
    fx = fixture("fdsa_c32_kat")
    sd = fixture_weights("fdsa_c32_kat", fx["shapes"])
    sd["to_hidden.weight"] = fx["to_hidden"]
    sd["to_hidden_dw.weight"] = fx["to_hidden_dw"]
    return fx, sd


def fdffn_edge():
    fx = fixture("fdffn_c32_edge")
    sd = fixture_weights("fdffn_c32_edge", fx["shapes"])
    sd["project_in.weight"][0:6] = 0.0
    return fx, sd


def fcaffn_edge():
    fx = fixture("fcaffn_c32_edge")
    return fx, fixture_weights("fcaffn_c32_edge", fx["shapes"])


def assert_regions_close(got, ref, what, rtol, regions=REGIONS, per_channel=False):
    """Every special region is judged against ITS OWN scale (values of order 1e-10 next to values of order 10)."""
    got, ref = got.double().cpu(), ref.double().cpu()
    for nm, (r0, r1, c0, c1) in regions.items():
        g, r = got[..., r0:r1, c0:c1], ref[..., r0:r1, c0:c1]
        if per_channel:
            scale = r.abs().amax(dim=(-1, -2), keepdim=True)
        else:
            scale = r.abs().max()
        err = ((g - r).abs() / (scale + 1e-300)).max().item()
        assert err <= rtol, f"{what}/{nm}: relative error {err:.3e} > {rtol:.1e} (scale {float(r.abs().max()):.3e})"


def assert_channels_close(got, ref, what, rtol):
    """Per (batch, channel) plane, relative to that plane's own scale."""
    got, ref = got.double().cpu(), ref.double().cpu()
    scale = ref.abs().amax(dim=(-1, -2), keepdim=True)
    err = ((got - ref).abs() / (scale + 1e-300)).amax(dim=(-1, -2))
    bad = torch.nonzero(err > rtol)
    assert bad.numel() == 0, f"{what}: channels {bad[:8].tolist()} off by {err.max().item():.3e} > {rtol:.1e}"
