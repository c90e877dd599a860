import os, sys
sys.path.insert(0, "/root/repo/fdn-tip2025_amd")
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
r = lambda *s: torch.randn(*s, device=dev)
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (C, N, H, W) in ((172, 64, 368, 640), (172, 64, 736, 1280), (86, 32, 736, 1280)):
    y, wd, w, res = r(8, C, H, W), r(2 * C, 1, 3, 3) * 0.3, r(N, C) / C ** .5, r(8, N, H, W)
    for rep in range(2):
        print(C, N, H, W, {m: round(timeit(lambda: ops.ffn_tail(y, wd, w, res=res, want_stats=True, mode=m)), 3) for m in ("sw", "split")}, flush=True)
