"""float64 "truth" for the BASELINE.json configs[1] frame (build container, ~15 minutes of CPU).

    python tests/golden/make_golden_configs64.py            # the truth windows
    python tests/golden/make_golden_configs64.py sens       # + the per-window sensitivity (a second fp64 forward)

fdn_tamed_736x1280.npz holds the REFERENCE's fp32 outputs on the reflect-padded 736 x 1280 frame as 64 seeded windows per
output (make_golden_configs.py).  This script runs the float64 oracle (oracle/fdn_oracle.py, pinned by the other fixtures) on
the same frame with the same weights and the reference's own LPNet ratio, and stores the SAME windows in float64
(fdn_tamed_736x1280_f64.npz).  tests/test_gpu_configs.py then holds the HIP path to the conditioning-aware bound per window:
err(HIP, f64) <= 4 * err(reference fp32, f64) + floor - a window may only be far from the reference where the reference itself
is far from the truth.  `sens` adds `*_sens64`: the same windows of truth(x + d) - truth(x) for d = 6e-8 * randn (one fp32 ulp of
the input, seed 5) evaluated in float64 - how strongly each window amplifies a rounding-sized perturbation (the FDSA
recombination divides by |q| and |k|, SURVEY.md fact 9: a few spots of a frame amplify by 1e3 and more).  A window may be as far
from the truth as that amplification explains, and no further.  (Result: 6e-8 in every window - an INPUT perturbation is not what moves the
ill-conditioned windows, the rounding inside the blocks is; the committed fixture therefore carries make_golden_configs_susc.py's measures instead and
the `*_sens64` arrays were dropped from it.)  Nothing here is read by the product path.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import fdn_oracle as O  # noqa: E402
from common import fdn_weights  # noqa: E402
from make_golden_configs import crop, moments, padded_input  # noqa: E402  (window / moment helpers; build container only)


def main(sens=False):
    torch.set_num_threads(int(os.environ.get("FDN_GOLDEN_THREADS", "6")))
    z = np.load(os.path.join(HERE, "fdn_tamed_736x1280.npz"))
    x = padded_input(720, 1280, int(z["x_seed"]))
    assert abs(x.double().sum().item() - float(z["x_sum64"])) < 1e-6
    P = O.cast_params(fdn_weights(tame=float(z["tame"])), torch.float64)
    ratio = torch.from_numpy(z["ratio"]).double()          # the reference's LPNet output (fp32), as the reference's FDN received it
    out_path = os.path.join(HERE, "fdn_tamed_736x1280_f64.npz")
    if sens:
        have = dict(np.load(out_path))
        d = 6e-8 * torch.randn(x.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
        with torch.no_grad():
            outs = O.fdn_forward(P, x.double() + d, ratio)
        for key, t, size in zip(("y", "q1", "q2", "q3"), outs, (32, 32, 16, 8)):
            org = torch.from_numpy(z[key + "_org"])
            have[key + "_sens64"] = (crop(t, org, size) - torch.from_numpy(have[key + "_win64"])).numpy()
            rms = np.sqrt((have[key + "_sens64"] ** 2).mean((1, 2, 3)))
            print(key, "window RMS of truth(x + 1 ulp) - truth(x): max %.2e median %.2e" % (float(rms.max()), float(np.median(rms))))
        np.savez_compressed(out_path, **have)
        print("added *_sens64")
        return
    with torch.no_grad():
        outs = O.fdn_forward(P, x.double(), ratio)
    arrs = {}
    for key, t, size in zip(("y", "q1", "q2", "q3"), outs, (32, 32, 16, 8)):
        org = torch.from_numpy(z[key + "_org"])
        arrs[key + "_win64"] = crop(t, org, size).numpy()
        arrs[key + "_mom64"] = moments(t).numpy()
        ref = torch.from_numpy(z[key + "_win"]).double()
        print(key, "reference fp32 vs f64 truth on the windows: PSNR %.1f dB" % O.psnr(ref, torch.from_numpy(arrs[key + "_win64"])))
    np.savez_compressed(out_path, **arrs)
    print("wrote fdn_tamed_736x1280_f64")


if __name__ == "__main__":
    main("sens" in sys.argv[1:])
