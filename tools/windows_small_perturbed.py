"""Which 16 x 16 windows of the 96 x 160 end-to-end fixture move when the HIP path is evaluated on input + 6e-8 * randn(seed) (one fp32 ulp)?
tools/windows_small_perturbed.py [n] [lib.so]: per evaluation the windows of y whose RMS error against the float64 truth exceeds 3e-7."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fdn_hip
args = sys.argv[1:]
n = int(args[0]) if args and args[0].isdigit() else 8
for a in args:
    if a.endswith(".so"): fdn_hip._LIB_PATH = os.path.abspath(a); print("library:", a)
from common import fixture, fdn_weights
from basicsr.models.archs.FDN_arch import FDN
dev = torch.device("cuda:0")
fx, cond = fixture("fdn_tamed_96x160"), fixture("fdn_tamed_96x160_cond")
m = FDN().to(dev).eval(); m.load_state_dict(fdn_weights(tame=float(fx["tame"])), strict=True)
truth = cond["y_f64"]
def wrms(d, size=16):
    B, C, H, W = d.shape
    return d.double().pow(2).reshape(B, C, H // size, size, W // size, size).mean((1, 3, 5)).sqrt().reshape(-1)
for k in range(n):
    x = fx["x"] if k == 0 else fx["x"] + 6e-8 * torch.randn(fx["x"].shape, generator=torch.Generator().manual_seed(100 + k))
    with torch.no_grad():
        y = m(x.to(dev), ratio_i=fx["ratio"].to(dev), device=dev)[0]
    e = wrms(y.cpu().double() - truth).numpy()
    print(k, [(int(i), float("%.1e" % e[i])) for i in np.argsort(-e)[:6] if e[i] > 3e-7], "median %.1e" % np.median(e), flush=True)
