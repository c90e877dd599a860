import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
B, C, H, W = 8, 128, 184, 320
x = torch.randn(B, C, H, W, device=dev); st = ops.chan_stats(x); g, b_ = torch.randn(C, device=dev), torch.randn(C, device=dev)
for N in (612, 345):
    w = torch.randn(N, C, device=dev) / C ** .5
    for _ in range(3): ops.conv1x1(x, w, ln=(st, g, b_))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv1x1(x, w, ln=(st, g, b_))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"FDN_DBG={os.environ.get('FDN_DBG','0')} N={N} {ms:.3f} ms  {2.0*B*C*N*H*W/ms*1e-9:.1f} TF/s")
