"""fdn_fdffn_mid at the level-1 / level-2 bench shapes with fp32 and bf16-storage operands (in / out / both)."""
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
for Hd, H, W in ((86, 736, 1280), (172, 368, 640)):
    r = lambda *s: torch.randn(*s, device=dev)
    h32 = r(B, Hd, H, W)
    h16 = h32.to(torch.bfloat16)
    w0, w2, fa, fp = r(Hd, 1, 3, 3), r(Hd, 1, 3, 3), r(Hd, 1, 1, 8, 5), r(Hd, 1, 1, 8, 5)
    for name, x, od in (("f32->f32", h32, torch.float32), ("bf16->f32", h16, torch.float32), ("f32->bf16", h32, torch.bfloat16), ("bf16->bf16", h16, torch.bfloat16)):
        ms = timeit(lambda: ops.fdffn_mid(x, w0, w2, fa, fp, out_dtype=od))
        print(f"Hd={Hd} {H}x{W} {name}: {ms:.3f} ms", flush=True)
