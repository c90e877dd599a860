"""Level-3 FCAFFN front half (B = 8, C = 128, 184 x 320): layernorm_chan + chan_stats + img_mod_maps + conv1x1 against chan_stats + fdn_fcaffn_in_packed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
B, C, H, W = 8, 128, 184, 320
r = lambda *s: torch.randn(*s, device=dev)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
xi, raw, img = r(B, C, H, W), r(B, C, H, W), torch.rand(B, 3, H, W, device=dev)
w = r(C, C) / C ** .5; g, b_, g1, b1 = r(C), r(C), r(C), r(C)
w1m, w3m, w1a, w3a = r(C, 3), r(C, 9) / 3, r(C, 3), r(C, 9) / 3
cache = ops.WeightCache()
st_raw = ops.chan_stats(raw)
def old():
    xn = ops.layernorm_chan(raw, g1, b1)
    mul, add = ops.img_mod_maps(img, w1m, w3m, w1a, w3a)
    return ops.conv1x1(xi, w, ln_muladd=(ops.chan_stats(xi), g, b_, xn), muladd=(mul, add), cache=(cache, "pi"))
wpk = ops.fcaffn_in_pack(w, w1m, w3m, w1a, w3a)
def new():
    return ops.fcaffn_in_packed(xi, ops.chan_stats(xi), raw, img, wpk, g, b_, x1_ln=(st_raw, g1, b1))
for name, fn in (("unfused (4 launches)", old), ("chan_stats + fdn_fcaffn_in_packed", new), ("unfused (4 launches)", old), ("chan_stats + fdn_fcaffn_in_packed", new)):
    print(f"{name:36s} {timeit(fn):.3f} ms", flush=True)
print("chan_stats alone %.3f ms" % timeit(lambda: ops.chan_stats(xi)))
