"""Timing of the level-3 1x1 convs (B = 8, 184 x 320) per library build: tools/ab_gemm_l3.py [lib.so | default ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        print("==", lib, flush=True)
        subprocess.run([sys.executable, __file__, "--child", lib])
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import fdn_hip
if len(sys.argv) > 2 and sys.argv[2] != "default":
    fdn_hip._LIB_PATH = os.path.abspath(sys.argv[2])
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
B, H, W, C, E, Hd = 8, 184, 320, 128, 153, 345
r = lambda *s: torch.randn(*s, device=dev)

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

x = r(B, C, H, W); st = ops.chan_stats(x); g, b_ = r(C), r(C)
cache = ops.WeightCache()
cases = []
w1 = r(4 * E, C) / C ** .5; cases.append(("128->612 LN", lambda: ops.conv1x1(x, w1, ln=(st, g, b_), cache=(cache, "a"))))
w2 = r(Hd, C) / C ** .5; cases.append(("128->345 LN", lambda: ops.conv1x1(x, w2, ln=(st, g, b_), cache=(cache, "b"))))
h = r(B, Hd, H, W); w3 = r(C, Hd) / Hd ** .5; cases.append(("345->128 +res", lambda: ops.conv1x1(h, w3, res=x, want_stats=True, cache=(cache, "c"))))
o = r(B, 4 * E, H, W); st3 = ops.chan_stats(o[:, :3 * E], groups=3) if "groups" in ops.chan_stats.__code__.co_varnames else None
w5 = r(C, C) / C ** .5; x1 = r(B, C, H, W); mul, add = r(B, C, H, W), r(B, C, H, W)
cases.append(("128->128 LN*x1+x1, *mul+add", lambda: ops.conv1x1(x, w5, ln_muladd=(st, g, b_, x1), muladd=(mul, add), cache=(cache, "e"))))
cases.append(("128->128 +res", lambda: ops.conv1x1(x, w5, res=x1, want_stats=True, cache=(cache, "f"))))
for name, fn in cases:
    print(f"{name:32s} {timeit(fn):.3f} ms", flush=True)
