"""Level-3 LN3 / FCAFFN GEMMs (B = 8, 184 x 320): fdn_chan_stats + GEMM against the GEMM that takes its statistics itself (stats=None), interleaved."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
B, H, W, C, E = 8, 184, 320, 128, 153
r = lambda *s: torch.randn(*s, device=dev)
# (the tensors are rotated so that no call finds its input in the Infinity Cache from the call before)
NB = 6
os_ = [r(B, 4 * E, H, W) for _ in range(NB)]
xs_ = [r(B, C, H, W) for _ in range(NB)]
g3, b3, w = r(3 * E), r(3 * E), r(C, 3 * E) / (3 * E) ** .5
res = r(B, C, H, W)
cache = ops.WeightCache()
img = torch.rand(B, 3, H, W, device=dev)
wf = r(C, C) / C ** .5
g, b_ = r(C), r(C)
w1m, w3m, w1a, w3a = r(C, 3), r(C, 9) / 3, r(C, 3), r(C, 9) / 3
wpk = ops.fcaffn_in_pack(wf, w1m, w3m, w1a, w3a)
x1 = r(B, C, H, W)

def ln3(i, own):
    o = os_[i % NB]
    st = None if own else ops.chan_stats(o[:, :3 * E], groups=3)
    return ops.conv1x1(o[:, :3 * E], w, ln3_gate=(st, g3, b3, o[:, 3 * E:]), res=res, want_stats=True, cache=(cache, "po"))

def fc(i, own):
    xi = xs_[i % NB]
    return ops.fcaffn_in_packed(xi, None if own else ops.chan_stats(xi), x1, img, wpk, g, b_)

def timeit(fn, own, n=12):
    fn(0, own); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i, own)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for name, fn in (("459->128 LN3 * v_value + res", ln3), ("FCAFFN front half 128->128", fc)):
    rows = [(timeit(fn, False), timeit(fn, True)) for _ in range(5)]
    a = sorted(x[0] for x in rows)[2]; b = sorted(x[1] for x in rows)[2]
    print(f"{name:32s} stats launch + GEMM {a:.3f} ms   own statistics {b:.3f} ms   ({(b - a) * 1e3:+.0f} us)", flush=True)
