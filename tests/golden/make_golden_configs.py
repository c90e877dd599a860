"""End-to-end goldens at the BASELINE.json shapes, made by RUNNING THE REFERENCE (build container only).

    python tests/golden/make_golden_configs.py [256] [720p]

* fdn_tamed_256      configs[0]: `torch.manual_seed(0); x = torch.rand(1, 3, 256, 256)` through the reference's
                     inference_fdn_lolblur.py:60-71 plumbing (LPNet with the real LPNet_lolblur.pth -> ratio -> FDN), FDN
                     weights = the tamed synthetic state dict of the other end-to-end fixtures.  Full outputs.
* fdn_tamed_736x1280 configs[1]'s frame: rand(1, 3, 720, 1280) (seed 1) reflect-padded to 736 x 1280 exactly as the driver
                     does (:60-63), same weights.  The four outputs are 11 MB each, so the fixture keeps 64 seeded 32 x 32
                     windows of each (16 x 16 / 8 x 8 for the half / quarter scale maps) plus per-channel moments
                     (sum, sum of squares in float64) of the whole tensors: a wrong stage hook-up anywhere moves both.

The input is not stored: it is regenerated from its seed (CPU generator, platform independent) and pinned by a float64
checksum.  Nothing here is read by the product path.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from _refload import REF_ROOT, build_ref_fdn, quiet, ref_lpnet_module  # noqa: E402
from weights import shapes_of, synth_state_dict  # noqa: E402

SEED, TAME = 7, 0.03
NWIN = 64


def windows(H, W, size, seed):
    """NWIN seeded window origins (multiples of 8 so that every window holds whole 8 x 8 patches)."""
    g = torch.Generator().manual_seed(seed)
    ys = torch.randint(0, (H - size) // 8 + 1, (NWIN,), generator=g) * 8
    xs = torch.randint(0, (W - size) // 8 + 1, (NWIN,), generator=g) * 8
    return torch.stack([ys, xs], 1)


def crop(t, org, size):
    return torch.stack([t[0, :, y:y + size, x:x + size] for y, x in org.tolist()])


def moments(t):
    d = t.double()
    return torch.stack([d.sum((0, 2, 3)), (d * d).sum((0, 2, 3))])


def padded_input(h, w, seed):
    """inference_fdn_lolblur.py:60-63 on a synthetic frame."""
    x = torch.rand(1, 3, h, w, generator=torch.Generator().manual_seed(seed))
    hn, wn = (32 - h % 32) % 32, (32 - w % 32) % 32
    return torch.nn.functional.pad(x, (0, wn, 0, hn), mode="reflect")


def main(which):
    torch.set_num_threads(8)
    net = build_ref_fdn(0)
    net.load_state_dict(synth_state_dict(shapes_of(net), SEED, prefix_key="fdn/", tame=TAME), strict=True)
    lp = ref_lpnet_module().I_predict_net().eval()
    lp.load_state_dict(torch.load(os.path.join(REF_ROOT, "checkpoint", "LPNet_lolblur.pth"), map_location="cpu")["params"], strict=True)

    if "256" in which:
        torch.manual_seed(0)
        x = torch.rand(1, 3, 256, 256)
        with torch.no_grad(), quiet():
            ratio = lp(x)
            y, q1, q2, q3 = net(x, ratio_i=ratio)
        np.savez_compressed(os.path.join(HERE, "fdn_tamed_256.npz"), x=x.numpy(), ratio=ratio.numpy(), y=y.numpy(), q1=q1.numpy(),
                            q2=q2.numpy(), q3=q3.numpy(), tame=np.float32(TAME))
        print("wrote fdn_tamed_256", float(ratio))

    if "720p" in which:
        x = padded_input(720, 1280, 1)
        with torch.no_grad(), quiet():
            ratio = lp(x)
            outs = net(x, ratio_i=ratio)
        arrs = {"x_seed": np.int64(1), "x_sum64": np.float64(x.double().sum().item()), "ratio": ratio.numpy(), "tame": np.float32(TAME)}
        for key, t, size, ws in zip(("y", "q1", "q2", "q3"), outs, (32, 32, 16, 8), (101, 102, 103, 104)):
            org = windows(t.shape[2], t.shape[3], size, ws)
            arrs[key + "_org"], arrs[key + "_win"], arrs[key + "_mom"] = org.numpy(), crop(t, org, size).numpy(), moments(t).numpy()
        np.savez_compressed(os.path.join(HERE, "fdn_tamed_736x1280.npz"), **arrs)
        print("wrote fdn_tamed_736x1280", float(ratio))


if __name__ == "__main__":
    main(sys.argv[1:] or ["256", "720p"])
