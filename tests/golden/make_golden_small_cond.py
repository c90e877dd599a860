"""Conditioning data of the small end-to-end fixtures (build container, about a minute):

    python tests/golden/make_golden_small_cond.py [fdn_tamed_96x160 fdn_tamed_64]

For each fixture (inputs and the REFERENCE's fp32 outputs are in <name>.npz, make_golden.py) this stores <name>_cond.npz:
  * <key>_f64      the float64 oracle's output (oracle/fdn_oracle.py, pinned by the other fixtures) - the truth;
  * <key>_susc     [n, windows] per-window RMS error against that truth of n fp32 ORACLE evaluations (the frame as it is, then
                   frame + 6e-8 * randn(seed)): where a window is ill-conditioned IN FP32;
  * <key>_noise    [m, windows] per-window RMS deviation of the float64 oracle with the rounding of an fp32 FFT emulated on every forward
                   transform (rfft2(t + 2e-7 * rms(t) * randn), m seeds): the conditioning itself, independent of any fp32 implementation.
Windows tile the whole output: 16 x 16 for y and q1, 8 x 8 for q2, 4 x 4 for q3 (the same frame regions).  Same method as
make_golden_configs64.py / make_golden_configs_susc.py for the 720p frame; tests/test_gpu_parity.py holds the HIP path per window to
4 x max(reference's own error, these measures) + a floor of a few ulp.  Nothing here is read by the product path.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import fdn_oracle as O  # noqa: E402
from common import fdn_weights, fixture  # noqa: E402

KEYS, WIN = ("y", "q1", "q2", "q3"), (16, 16, 8, 4)


def window_rms(d, size):
    """d: [B, C, H, W] difference -> RMS per size x size window (all channels), row-major over the frames of the batch"""
    B, C, H, W = d.shape
    t = d.double().pow(2).reshape(B, C, H // size, size, W // size, size).mean((1, 3, 5))
    return t.sqrt().reshape(-1)


def main(name, n_susc=5, n_noise=8):
    torch.set_num_threads(int(os.environ.get("FDN_GOLDEN_THREADS", "6")))
    fx = fixture(name)
    x, ratio, tame = fx["x"], fx["ratio"], float(fx["tame"])
    P32 = fdn_weights(tame=tame)
    P64 = O.cast_params(P32, torch.float64)
    with torch.no_grad():
        truth = O.fdn_forward(P64, x.double(), ratio.double())
    arrs = {}
    for key, t, size in zip(KEYS, truth, WIN):
        arrs[key + "_f64"] = t.numpy()
        e = window_rms(fx[key].double() - t, size)
        print(name, key, "reference fp32 against the truth: PSNR %.1f dB, worst window %.2e, median %.2e" % (O.psnr(fx[key].double(), t), float(e.max()), float(e.median())))
    rows = {k: [] for k in KEYS}
    for k in range(n_susc):
        xin = x if k == 0 else x + 6e-8 * torch.randn(x.shape, generator=torch.Generator().manual_seed(100 + k))
        with torch.no_grad():
            outs = O.fdn_forward(P32, xin, ratio)
        for key, t, tr, size in zip(KEYS, outs, truth, WIN):
            rows[key].append(window_rms(t.double() - tr, size).numpy())
        print(name, "fp32 oracle evaluation", k, "worst y windows:", [(int(i), float("%.2e" % rows["y"][-1][i])) for i in np.argsort(-rows["y"][-1])[:4]], flush=True)
    for key in KEYS:
        arrs[key + "_susc"] = np.stack(rows[key])
    real_rfft2 = torch.fft.rfft2
    rows = {k: [] for k in KEYS}
    for k in range(n_noise):
        gen = torch.Generator().manual_seed(500 + k)

        def noisy_rfft2(t, *a, **kw):
            rms = t.pow(2).mean(dim=(-2, -1), keepdim=True).sqrt()
            return real_rfft2(t + 2e-7 * rms * torch.randn(t.shape, generator=gen, dtype=torch.float64), *a, **kw)
        torch.fft.rfft2 = noisy_rfft2
        try:
            with torch.no_grad():
                outs = O.fdn_forward(P64, x.double(), ratio.double())
        finally:
            torch.fft.rfft2 = real_rfft2
        for key, t, tr, size in zip(KEYS, outs, truth, WIN):
            rows[key].append(window_rms(t - tr, size).numpy())
        print(name, "noise seed", k, "worst y windows:", [(int(i), float("%.2e" % rows["y"][-1][i])) for i in np.argsort(-rows["y"][-1])[:4]], flush=True)
    for key in KEYS:
        arrs[key + "_noise"] = np.stack(rows[key])
    np.savez_compressed(os.path.join(HERE, name + "_cond.npz"), **arrs)
    print("wrote", name + "_cond.npz")


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["fdn_tamed_96x160", "fdn_tamed_64"]):
        main(nm)
