"""Which windows of the configs[1] frame are ill-conditioned IN FP32 (build container, ~7 minutes of CPU per evaluation).

    python tests/golden/make_golden_configs_susc.py [n_evaluations, default 4]          # fp32 oracle evaluations  -> *_susc
    python tests/golden/make_golden_configs_susc.py noise64 [n, default 4] [first seed] # float64 + emulated FFT rounding -> *_susc_noise

The FDSA recombination divides by |q| and |k| and replaces spectrum bins below 1e-10 (FDN_arch.py:593-607): where a bin sits
near such a point, an fp32 evaluation lands on either side depending on its rounding, and the window around it moves by 1e-6 ..
1e-3 while its neighbours stay at 3e-8.  A float64 perturbation of the INPUT does not show this (`*_sens64`: 6e-8 everywhere),
because the rounding noise an fp32 evaluation accumulates inside 70 blocks is far larger than one input ulp.  So the fp32 ORACLE
(the restatement of the reference, pinned by the other fixtures) is evaluated n times - once on the frame as it is, then on
frame + 6e-8 * randn(seed k) - and the per-window RMS error of each evaluation against the float64 truth is stored
(`y_susc`: [n, 64]).  tests/test_gpu_configs.py: a window may be as far from the truth as 4 x the worst of the reference's own
error and these evaluations' errors at THAT window (+ a floor of a few ulp); nothing else is allowed.

Which bins an evaluation trips over depends on its arithmetic (the fp32 oracle trips at window 18 in every run, the reference never,
the HIP path at window 24 in four of five runs), so the windows of ONE implementation's evaluations are not the whole ill-conditioned
set.  `noise64` measures the conditioning itself, independent of any fp32 implementation: the float64 oracle is run with the rounding
of an fp32 FFT emulated on every forward transform of the path - rfft2(t + 2e-7 * rms(t) * randn), the backward-error form of an fp32
8x8 or full-image FFT - with n noise seeds, and the per-window RMS deviation from the
clean float64 truth is stored (`y_susc_noise`: [n, 64]).  A window that moves by 1e-6 under that noise is a window where ANY fp32
evaluation may land 1e-6 away from the truth.
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


def crop(t, org, size):
    return torch.stack([t[0, :, y:y + size, x:x + size] for y, x in org.tolist()])


def main_noise(n, first=0):
    """float64 oracle with emulated fp32 FFT rounding (see the module docstring); first > 0 appends seeds first .. first + n - 1."""
    torch.set_num_threads(int(os.environ.get("FDN_GOLDEN_THREADS", "6")))
    z = np.load(os.path.join(HERE, "fdn_tamed_736x1280.npz"))
    out_path = os.path.join(HERE, "fdn_tamed_736x1280_f64.npz")
    have = dict(np.load(out_path))
    x = torch.rand(1, 3, 720, 1280, generator=torch.Generator().manual_seed(int(z["x_seed"])))
    x = torch.nn.functional.pad(x, (0, 0, 0, 16), mode="reflect").double()
    P = O.cast_params(fdn_weights(tame=float(z["tame"])), torch.float64)
    ratio = torch.from_numpy(z["ratio"]).double()
    real_rfft2 = torch.fft.rfft2
    rows = {k: ([r for r in have[k + "_susc_noise"]] if first > 0 and k + "_susc_noise" in have else []) for k in ("y", "q1", "q2", "q3")}
    for k in range(first, first + n):
        gen = torch.Generator().manual_seed(500 + k)

        def noisy_rfft2(t, *a, **kw):
            # backward-error model of an fp32 FFT: the exact transform of a slightly perturbed REAL input (white noise of 2e-7 of the
            # input's rms per transformed patch / plane) - the noise then has the Hermitian structure of a real transform (a noise
            # term added to the spectrum itself puts imaginary parts on the self-conjugate bins and flips the +-pi phases MAR mixes)
            rms = t.pow(2).mean(dim=(-2, -1), keepdim=True).sqrt()
            return real_rfft2(t + 2e-7 * rms * torch.randn(t.shape, generator=gen, dtype=torch.float64), *a, **kw)
        torch.fft.rfft2 = noisy_rfft2
        try:
            with torch.no_grad():
                outs = O.fdn_forward(P, x, ratio)
        finally:
            torch.fft.rfft2 = real_rfft2
        for key, t, size in zip(("y", "q1", "q2", "q3"), outs, (32, 32, 16, 8)):
            w = crop(t, torch.from_numpy(z[key + "_org"]), size)
            rows[key].append(((w - torch.from_numpy(have[key + "_win64"])) ** 2).mean((1, 2, 3)).sqrt().numpy())
        print("noise seed", k, "y windows:", [(int(i), float("%.2e" % rows["y"][-1][i])) for i in np.argsort(-rows["y"][-1])[:8]],
              "median %.1e" % np.median(rows["y"][-1]), flush=True)
        for key in rows:
            have[key + "_susc_noise"] = np.stack(rows[key])
        np.savez_compressed(out_path, **have)
    print("stored *_susc_noise", have["y_susc_noise"].shape)


def main(n):
    torch.set_num_threads(int(os.environ.get("FDN_GOLDEN_THREADS", "6")))
    z = np.load(os.path.join(HERE, "fdn_tamed_736x1280.npz"))
    out_path = os.path.join(HERE, "fdn_tamed_736x1280_f64.npz")
    have = dict(np.load(out_path))
    x = torch.rand(1, 3, 720, 1280, generator=torch.Generator().manual_seed(int(z["x_seed"])))
    x = torch.nn.functional.pad(x, (0, 0, 0, 16), mode="reflect")
    P = fdn_weights(tame=float(z["tame"]))
    ratio = torch.from_numpy(z["ratio"])
    rows = {k: [] for k in ("y", "q1", "q2", "q3")}
    for k in range(n):
        xin = x if k == 0 else x + 6e-8 * torch.randn(x.shape, generator=torch.Generator().manual_seed(100 + k))
        with torch.no_grad():
            outs = O.fdn_forward(P, xin, ratio)
        for key, t, size in zip(("y", "q1", "q2", "q3"), outs, (32, 32, 16, 8)):
            w = crop(t, torch.from_numpy(z[key + "_org"]), size).double()
            e = ((w - torch.from_numpy(have[key + "_win64"])) ** 2).mean((1, 2, 3)).sqrt()
            rows[key].append(e.numpy())
        print("evaluation", k, "y windows above 1e-6:", [(int(i), float("%.2e" % rows["y"][-1][i])) for i in np.argsort(-rows["y"][-1])[:6]], flush=True)
        for key in rows:
            have[key + "_susc"] = np.stack(rows[key])
        np.savez_compressed(out_path, **have)
    print("stored *_susc", have["y_susc"].shape)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "noise64":
        main_noise(int(sys.argv[2]) if len(sys.argv) > 2 else 4, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    else:
        main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
