import sys, os
sys.path.insert(0, "fdn-tip2025_amd")
import fdn_hip
if len(sys.argv) > 1: fdn_hip._LIB_PATH = os.path.abspath(sys.argv[1])
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
ref = ops.fdsa_core(hidden, dw, fw)
got = ops.fdsa_fused(x, st, wpk, dw, fw)
torch.cuda.synchronize()
d = (got - ref).abs()
print("max", d.max().item(), "ref max", ref.abs().max().item())
for kind in range(4):
    dk = d[:, kind*E:(kind+1)*E]
    print("kind", kind, dk.max().item(), "per-e max:", [round(dk[:, e].max().item(), 4) for e in range(E)])
# spatial pattern
dd = d[0, 3*E:4*E].amax(dim=0)
print((dd > 1e-4).nonzero()[:20].tolist())
for kind, e in ((0, 3), (2, 6), (0, 6), (1, 3)):
    dk = d[:, kind * E + e]
    nz = (dk > 1e-4).nonzero()
    print("kind", kind, "e", e, "count", nz.shape[0], "of", dk.numel(), "first", nz[:6].tolist(), "rows", sorted(set(nz[:, 1].tolist()))[:20], "cols", sorted(set(nz[:, 2].tolist()))[:40])
