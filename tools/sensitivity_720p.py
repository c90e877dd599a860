"""How far do two equally valid fp32 evaluations of LPNet -> FDN on the 736 x 1280 fixture frame differ?  The same library is run on
the fixture input and on input + eps * randn (eps = 6e-8: one ulp of values in [0.5, 1)); reported per perturbation: frame PSNR,
the worst 32 x 32 window, how many of the 920 windows fall below 100 dB.  (tests/test_gpu_configs.py states its bounds from this.)
    python tools/sensitivity_720p.py [lib.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fdn_hip
if len(sys.argv) > 1 and sys.argv[1] != "default":
    fdn_hip._LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
import fdn_oracle as O
from common import GOLDEN, fdn_weights, lpnet_weights
import basicsr.models.archs.FDN_arch as A
from basicsr.models.archs.LPNet_arch import I_predict_net
z = np.load(os.path.join(GOLDEN, "fdn_tamed_736x1280.npz"))
net = A.FDN(); net.load_state_dict(fdn_weights(tame=float(z["tame"])), strict=True); net = net.to("cuda:0").eval()
lp = I_predict_net(); lp.load_state_dict(lpnet_weights(), strict=True); lp = lp.to("cuda:0").eval()
x = torch.rand(1, 3, 720, 1280, generator=torch.Generator().manual_seed(int(z["x_seed"])))
x = torch.nn.functional.pad(x, (0, 0, 0, 16), mode="reflect").to("cuda:0")
def run(t):
    with torch.no_grad():
        return net(t, ratio_i=lp(x), device=torch.device("cuda:0"))[0].cpu().double()
y0 = run(x)
def report(name, y1):
    d = (y1 - y0)
    mse = (d ** 2).mean().item()
    w = torch.nn.functional.avg_pool2d(d ** 2, 32).mean(1)[0]          # per 32x32 window
    wp = 10 * torch.log10(1.0 / w.clamp_min(1e-30))
    iy, ix = divmod(int(w.argmax()), w.shape[1])
    print(f"{name}: frame PSNR {10 * np.log10(1.0 / max(mse, 1e-30)):.1f} dB, worst 32x32 window {wp.min().item():.1f} dB at ({iy * 32},{ix * 32}), "
          f"windows below 100 dB: {(wp < 100).sum().item()} of {wp.numel()}, median {wp.median().item():.1f}, max|d| {d.abs().max().item():.2e}", flush=True)
for seed, eps in ((1, 6e-8), (2, 6e-8), (3, 6e-8), (4, 6e-9)):
    n = torch.randn(x.shape, generator=torch.Generator().manual_seed(seed)).to("cuda:0")
    report(f"input + {eps:g} * randn (seed {seed})", run((x + eps * n).clamp_(0.0, 1.0)))
