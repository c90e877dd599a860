"""FCAFFN front half at the bench shapes: chan_stats + img_mod_maps + conv1x1 (unfused) against fdn_fcaffn_in."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B = 8
for C, H, W in ((32, 736, 1280), (64, 368, 640)):
    r = lambda *s: torch.randn(*s, device=dev)
    xi, x1, img = r(B, C, H, W), r(B, C, H, W), torch.rand(B, 3, H, W, device=dev)
    w, g, b_ = r(C, C) / C ** .5, r(C), r(C)
    w1m, w3m, w1a, w3a = r(C, 3), r(C, 9), r(C, 3), r(C, 9)
    t_stats = timeit(lambda: ops.chan_stats(xi))
    t_maps = timeit(lambda: ops.img_mod_maps(img, w1m, w3m, w1a, w3a))
    st = ops.chan_stats(xi); mul, add = ops.img_mod_maps(img, w1m, w3m, w1a, w3a)
    t_gemm = timeit(lambda: ops.conv1x1(xi, w, ln_muladd=(st, g, b_, x1), muladd=(mul, add)))
    t_new = timeit(lambda: ops.fcaffn_in(xi, x1, img, w, g, b_, w1m, w3m, w1a, w3a))
    gb = 3 * xi.numel() * 4 / 1e9
    print(f"C={C} {H}x{W}: stats {t_stats:.3f} + maps {t_maps:.3f} + gemm {t_gemm:.3f} = {t_stats + t_maps + t_gemm:.3f} ms   fused {t_new:.3f} ms ({gb / t_new * 1e3:.0f} GB/s on 3 streams)", flush=True)
