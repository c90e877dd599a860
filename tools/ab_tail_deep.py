"""The deep FFN tails (B = 8): level 2 172 -> 64 at 368 x 640, level 3 345 -> 128 at 184 x 320, Fuse 345 -> 128 at 368 x 640 / 172 -> 64 at
736 x 1280: gate + GEMM ("split") against the one-launch form with the projection on the bf16 matrix pipe ("gemm":
the shelved experiment tools/experiments/ffn_tail_gemm.hip - this script needs it restored as described in that file, plus the "gemm" mode of
ops.ffn_tail from commit 9aa8b58's successor in the git history; without it only "split" is timed)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
r = lambda *s: torch.randn(*s, device=dev)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (C, N, H, W) in ((172, 64, 368, 640), (345, 128, 184, 320), (345, 128, 368, 640), (172, 64, 736, 1280)):
    y, wd, w, res = r(8, C, H, W), r(2 * C, 1, 3, 3) * 0.3, r(N, C) / C ** .5, r(8, N, H, W)
    wc = ops.WeightCache()
    for rep in range(2):
        ts = {m: timeit(lambda: ops.ffn_tail(y, wd, w, res=res, want_stats=True, mode=m, cache=(wc, "po"))) for m in (("split", "gemm") if hasattr(ops.lib(), "fdn_ffn_tail_packed") else ("split",))}
        print(f"{C:4d} -> {N:3d} {H}x{W}: " + "   ".join(f"{m} {t:.3f} ms" for m, t in ts.items()), flush=True)
