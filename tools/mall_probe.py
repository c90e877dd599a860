"""Does the Infinity Cache (256 MB) serve RE-READS of a recently read tensor?  fdn_chan_stats (a pure streaming read) is run repeatedly
on tensors of growing size: a working set that fits should read faster than HBM on the repeats.  python tools/mall_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (16, 32, 64, 128, 192, 256, 384, 512, 1024):
    C = 128
    P = mb * (1 << 20) // (4 * C)
    P = P // 256 * 256
    x = torch.randn(1, C, P // 256, 256, device=dev)
    t = timeit(lambda: ops.chan_stats(x))
    print(f"{x.numel() * 4 / 2**20:7.0f} MB read repeatedly: {t * 1e3:8.1f} us  {x.numel() * 4 / t / 1e9:6.2f} TB/s", flush=True)
