"""float64 "truth" for the BASELINE.json configs[1] frame (build container, ~15 minutes of CPU).

    python tests/golden/make_golden_configs64.py

fdn_tamed_736x1280.npz holds the REFERENCE's fp32 outputs on the reflect-padded 736 x 1280 frame as 64 seeded windows per
output (make_golden_configs.py).  This script runs the float64 oracle (oracle/fdn_oracle.py, pinned by the other fixtures) on
the same frame with the same weights and the reference's own LPNet ratio, and stores the SAME windows in float64
(fdn_tamed_736x1280_f64.npz).  tests/test_gpu_configs.py then holds the HIP path to the conditioning-aware bound per window:
err(HIP, f64) <= 4 * err(reference fp32, f64) + floor - a window may only be far from the reference where the reference itself
is far from the truth.  Nothing here is read by the product path.
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


def main():
    torch.set_num_threads(int(os.environ.get("FDN_GOLDEN_THREADS", "6")))
    z = np.load(os.path.join(HERE, "fdn_tamed_736x1280.npz"))
    x = padded_input(720, 1280, int(z["x_seed"]))
    assert abs(x.double().sum().item() - float(z["x_sum64"])) < 1e-6
    P = O.cast_params(fdn_weights(tame=float(z["tame"])), torch.float64)
    ratio = torch.from_numpy(z["ratio"]).double()          # the reference's LPNet output (fp32), as the reference's FDN received it
    with torch.no_grad():
        outs = O.fdn_forward(P, x.double(), ratio)
    arrs = {}
    for key, t, size in zip(("y", "q1", "q2", "q3"), outs, (32, 32, 16, 8)):
        org = torch.from_numpy(z[key + "_org"])
        arrs[key + "_win64"] = crop(t, org, size).numpy()
        arrs[key + "_mom64"] = moments(t).numpy()
        ref = torch.from_numpy(z[key + "_win"]).double()
        print(key, "reference fp32 vs f64 truth on the windows: PSNR %.1f dB" % O.psnr(ref, torch.from_numpy(arrs[key + "_win64"])))
    np.savez_compressed(os.path.join(HERE, "fdn_tamed_736x1280_f64.npz"), **arrs)
    print("wrote fdn_tamed_736x1280_f64")


if __name__ == "__main__":
    main()
