"""Ratio sweep on the HIP path: the role of the reference's inference_fdn_multi_r.py:52-85 - ONE low-light frame enhanced with the
illumination ratio forced to every value of a grid (the reference: `for i in np.arange(0, 1, 0.01)`, `ratio = ratio / ratio * i`,
results written to ./multi_r/<i>.png), here with the grid values of the sweep as the BATCH of one forward per chunk: the frame is
repeated, ratio_i carries the grid, LPNet is not needed (its prediction only supplied the shape in the reference).

    python inference_fdn_multi_r.py --fdn FDN_lolblur.pth --input frame.png --output multi_r/ [--start 0 --stop 1 --step 0.01]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from inference_fdn_lolblur import load_params, read_rgb, write_rgb  # noqa: E402


def sweep_values(start, stop, step):
    """np.arange(start, stop, step) as the reference walks it (inference_fdn_multi_r.py:55)."""
    return np.arange(start, stop, step)


def output_name(v):
    """The reference formats the loop variable itself: './multi_r/{}.png'.format(i) (:84)."""
    return "{}.png".format(v)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--fdn", required=True, help="FDN checkpoint ({'params': state_dict}, 1503 keys)")
    ap.add_argument("--input", required=True, help="one input frame")
    ap.add_argument("--output", default="multi_r", help="output directory (the reference: ./multi_r)")
    ap.add_argument("--start", type=float, default=0.0)
    ap.add_argument("--stop", type=float, default=1.0)
    ap.add_argument("--step", type=float, default=0.01)
    ap.add_argument("--batch", type=int, default=8, help="grid values per forward")
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args()

    from basicsr.models.archs.FDN_arch import FDN
    from fdn_hip.harness import enhance_u8

    dev = torch.device(a.device)
    torch.cuda.set_device(dev)
    net = FDN().to(dev).eval()
    net.load_state_dict(load_params(a.fdn), strict=True)
    img = torch.from_numpy(read_rgb(a.input)).to(dev)
    vals = sweep_values(a.start, a.stop, a.step)
    for c0 in range(0, len(vals), a.batch):
        chunk = vals[c0:c0 + a.batch]
        ratio = torch.tensor(chunk, dtype=torch.float32, device=dev).view(-1, 1)
        out = enhance_u8(net, None, img.unsqueeze(0).expand(len(chunk), -1, -1, -1).contiguous(), bgr=False, ratio_mode="fixed", ratio=ratio)
        for v, o in zip(chunk, out.cpu().numpy()):
            write_rgb(os.path.join(a.output, output_name(v)), o)
    print(f"{len(vals)} ratios -> {a.output}")


if __name__ == "__main__":
    main()
