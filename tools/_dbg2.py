import sys, os
sys.path.insert(0, "fdn-tip2025_amd")
import fdn_hip
fdn_hip._LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from fdn_hip import ops
dev = "cuda:0"
def _rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).to(dev)
C,H,W,B = 32,32,64,2
E = 38
x = _rnd(B, C, H, W, seed=1) * 1.5 + 0.3
w = _rnd(4 * E, C, seed=2) / C ** 0.5
g, b_ = _rnd(C, seed=3) * 0.2 + 1.0, _rnd(C, seed=4) * 0.1
dw, fw = _rnd(4 * E, 1, 3, 3, seed=5) / 3, _rnd(E, 1, 1, 8, 5, seed=6) * 0.2 + 1.0
st = ops.chan_stats(x)
hidden = ops.conv1x1(x, w, ln=(st, g, b_))
wpk = ops.fdsa_pack(w, g, b_)
got = ops.fdsa_fused(x, st, wpk, dw, fw)
torch.cuda.synchronize()
flat = got.flatten().cpu()
wks = flat[:384].view(32, 12)[:, :9]
exp = torch.stack([dw.cpu()[(m >> 3) * E + (m & 7)].flatten() for m in range(32)])
print("wks rows wrong:", [(m, (wks[m] - exp[m]).abs().max().item()) for m in range(32) if (wks[m] - exp[m]).abs().max() > 0])
hid = flat[1024:1024 + 32 * 350].view(32, 10, 35)[:, :, :34]
# expected: hidden at tile (0,0) of image 0: halo rows -1..8, cols -1..32
hp = torch.nn.functional.pad(hidden[0].cpu(), (1, 1, 1, 1))
exph = torch.stack([hp[(m >> 3) * E + (m & 7), 0:10, 0:34] for m in range(32)])
print("hid rows wrong:", [(m, (hid[m] - exph[m]).abs().max().item()) for m in range(32) if (hid[m] - exph[m]).abs().max() > 0])
