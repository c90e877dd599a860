"""Per-window errors of the HIP path on the configs[1] frame against the float64 truth, beside the reference's own fp32 errors
(tests/golden/fdn_tamed_736x1280{,_f64}.npz).  python tools/windows_720p.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("fdn-tip2025_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from common import GOLDEN, fdn_weights, lpnet_weights
from basicsr.models.archs import FDN_arch as A
from basicsr.models.archs.LPNet_arch import I_predict_net
z = np.load(os.path.join(GOLDEN, "fdn_tamed_736x1280.npz")); z64 = np.load(os.path.join(GOLDEN, "fdn_tamed_736x1280_f64.npz"))
net = A.FDN(); net.load_state_dict(fdn_weights(tame=float(z["tame"])), strict=True); net = net.to("cuda:0").eval()
lp = I_predict_net(); lp.load_state_dict(lpnet_weights(), strict=True); lp = lp.to("cuda:0").eval()
x = torch.rand(1, 3, 720, 1280, generator=torch.Generator().manual_seed(int(z["x_seed"])))
x = torch.nn.functional.pad(x, (0, 0, 0, 16), mode="reflect").to("cuda:0")
with torch.no_grad():
    outs = net(x, ratio_i=lp(x), device=torch.device("cuda:0"))
for got, key, size in zip(outs, ("y", "q1", "q2", "q3"), (32, 32, 16, 8)):
    got = got.cpu().double()
    org = z[key + "_org"]
    mine = torch.stack([got[0, :, y0:y0 + size, x0:x0 + size] for y0, x0 in org.tolist()])
    t = torch.from_numpy(z64[key + "_win64"]); r = torch.from_numpy(z[key + "_win"]).double()
    eh = ((mine - t) ** 2).mean((1, 2, 3)).sqrt(); er = ((r - t) ** 2).mean((1, 2, 3)).sqrt()
    o = torch.argsort(eh, descending=True)
    print(key, "worst HIP windows (rms err vs f64):", [(int(i), f"{eh[i]:.2e}", f"ref {er[i]:.2e}") for i in o[:8]])
    print(key, "sorted HIP:", [f"{v:.1e}" for v in torch.sort(eh, descending=True)[0][:10]], " sorted ref:", [f"{v:.1e}" for v in torch.sort(er, descending=True)[0][:10]])
    print(key, "median HIP %.2e ref %.2e; windows with HIP > 4 ref + 1e-7: %d" % (eh.median(), er.median(), int((eh > 4 * er + 1e-7).sum())))
