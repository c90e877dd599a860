"""HIP path on the configs[1] frame and on frame + 6e-8 randn (k seeds): per-window RMS error against the float64 truth, and where inside
its worst window the error sits.  python tools/windows_720p_perturbed.py [k]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("fdn-tip2025_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from common import GOLDEN, fdn_weights, lpnet_weights
from basicsr.models.archs import FDN_arch as A
z = np.load(os.path.join(GOLDEN, "fdn_tamed_736x1280.npz")); z64 = np.load(os.path.join(GOLDEN, "fdn_tamed_736x1280_f64.npz"))
net = A.FDN(); net.load_state_dict(fdn_weights(tame=float(z["tame"])), strict=True); net = net.to("cuda:0").eval()
x0 = torch.rand(1, 3, 720, 1280, generator=torch.Generator().manual_seed(int(z["x_seed"])))
x0 = torch.nn.functional.pad(x0, (0, 0, 0, 16), mode="reflect")
ratio = torch.from_numpy(z["ratio"]).to("cuda:0")
org = z["y_org"]; t = torch.from_numpy(z64["y_win64"])
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    x = x0 if k == 0 else x0 + 6e-8 * torch.randn(x0.shape, generator=torch.Generator().manual_seed(100 + k))
    with torch.no_grad():
        y = net(x.to("cuda:0"), ratio_i=ratio, device=torch.device("cuda:0"))[0].cpu().double()
    mine = torch.stack([y[0, :, a:a + 32, b:b + 32] for a, b in org.tolist()])
    e = ((mine - t) ** 2).mean((1, 2, 3)).sqrt()
    print("run", k, [(int(i), f"{e[i]:.1e}") for i in torch.argsort(e, descending=True)[:6]])
    if k == 0:
        w = int(torch.argmax(e)); d = (mine[w] - t[w]).abs()
        pm = d.reshape(3, 4, 8, 4, 8).amax((0, 2, 4))
        print("  worst window", w, "origin", org[w].tolist(), "max |err| per 8x8 patch:\n", np.array2string(pm.numpy(), precision=1))
