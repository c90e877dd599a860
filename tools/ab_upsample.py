"""Upsample (bilinear x2 + 3x3 conv) at the two shapes of the 720p bench: fdn_resample + fdn_conv2d against the per-tap 1x1 conv at low resolution +
fdn_upconv_gather, interleaved; per-launch times of the new route's two kernels beside it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import fdn_hip
if len(sys.argv) > 1:                      # tools/ab_upsample.py [lib.so]: another build of the library
    fdn_hip._LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
r = lambda *s: torch.randn(*s, device=dev)

def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for C, h, w in ((64, 368, 640), (128, 184, 320)):
    x, wt = r(8, C, h, w), r(C // 2, C, 3, 3) / (3 * C ** .5)
    cache = ops.WeightCache()
    old = lambda: ops.conv2d(ops.resample(x, ops.RS_BILINEAR_X2), wt, pad=1)
    new = lambda: ops.upsample_conv3x3(x, wt, cache=(cache, "up"))
    wr = wt.permute(2, 3, 0, 1).reshape(9 * (C // 2), C).contiguous()
    z = ops.conv1x1(x, wr, cache=(cache, "g"))
    gemm = lambda: ops.conv1x1(x, wr, cache=(cache, "g"))
    out = torch.empty(8, C // 2, 2 * h, 2 * w, device=dev)
    import ctypes, fdn_hip
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    gather = lambda: fdn_hip.lib().fdn_upconv_gather(P(z), P(out), 8, C // 2, h, w, fdn_hip.stream())
    rows = [(timeit(old), timeit(new)) for _ in range(5)]
    a, b = sorted(t[0] for t in rows)[2], sorted(t[1] for t in rows)[2]
    print(f"Upsample {C} -> {C // 2}, {h} x {w} -> {2 * h} x {2 * w}: resample + conv3x3 {a:.3f} ms   per-tap 1x1 + gather {b:.3f} ms"
          f"   (1x1 {C} -> {9 * C // 2}: {timeit(gemm):.3f}, gather: {timeit(gather):.3f})", flush=True)
