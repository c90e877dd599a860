"""Shared helpers for the parity tests (fixtures, synthetic weights, tolerance policy)."""
import json
import os

import numpy as np
import torch

from weights import synth_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = 7  # tests/golden/make_golden.py


def fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        if k == "shapes_json":
            out["shapes"] = {kk: tuple(v) for kk, v in json.loads(bytes(z[k]).decode()).items()}
        else:
            out[k] = torch.from_numpy(z[k])
    return out


def fixture_weights(name, shapes, tame=None, po_scale=None):
    sd = synth_state_dict(shapes, SEED, prefix_key=name + "/", tame=tame)
    if po_scale is not None:
        for k in sd:
            if k.endswith("project_out.weight"):
                sd[k] = sd[k] * po_scale
    return sd


_fdn_shapes = None


def fdn_shapes():
    global _fdn_shapes
    if _fdn_shapes is None:
        _fdn_shapes = fixture("fdn_tamed_64")["shapes"]
    return _fdn_shapes


def fdn_weights(tame=0.03):
    return synth_state_dict(fdn_shapes(), SEED, prefix_key="fdn/", tame=tame)


def lpnet_weights(which="lolblur"):
    z = np.load(os.path.join(GOLDEN, f"lpnet_{which}_params.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


_lolv1_shapes = None


def lolv1_shapes():
    global _lolv1_shapes
    if _lolv1_shapes is None:
        _lolv1_shapes = fixture("lolv1_tamed_64")["shapes"]
    return _lolv1_shapes


def lolv1_weights(tame=0.03):
    """FDN_lolv1 (dim 24) synthetic state dict, the one tests/golden/make_golden_lolv1.py loaded into the reference."""
    return synth_state_dict(lolv1_shapes(), SEED, prefix_key="fdnlol/", tame=tame)


def rel_rms(a, b):
    a, b = a.double(), b.double()
    return (torch.sqrt(torch.mean((a - b) ** 2)) / (torch.sqrt(torch.mean(b ** 2)) + 1e-30)).item()


def assert_close_cond(got, ref32, truth64, what, factor=4.0, floor=2e-6):
    """Conditioning-aware tolerance: the candidate may be at most
    `factor` times as far from the fp64 truth as the fp32 reference itself is, plus a floor
    (relative RMS).  Also bounds the 99.9th percentile of |err| the same way."""
    got, ref32, truth64 = got.double().cpu(), ref32.double().cpu(), truth64.double().cpu()
    scale = torch.sqrt(torch.mean(truth64 ** 2)).item() + 1e-30
    e_got = torch.sqrt(torch.mean((got - truth64) ** 2)).item() / scale
    e_ref = torch.sqrt(torch.mean((ref32 - truth64) ** 2)).item() / scale
    q = lambda t: torch.quantile(t.abs().flatten()[:4_000_000], 0.999).item() / scale
    p_got, p_ref = q(got - truth64), q(ref32 - truth64)
    assert e_got <= factor * e_ref + floor, f"{what}: rel-RMS err {e_got:.3e} > {factor}*{e_ref:.3e}+{floor}"
    assert p_got <= factor * p_ref + 10 * floor, f"{what}: p99.9 err {p_got:.3e} > {factor}*{p_ref:.3e}+{10*floor}"
    return e_got, e_ref
