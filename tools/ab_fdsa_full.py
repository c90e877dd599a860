"""The FDSA sub-block at the bench shapes: one launch (fdn_fdsa_full) against the two launches it replaces (fdn_fdsa_fused ->
fdn_fdsa_out), interleaved in one process.  python tools/ab_fdsa_full.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("fdn-tip2025_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402

if len(sys.argv) > 2:                         # another build of the same ABI (A/B): python tools/ab_fdsa_full.py <reps> <lib.so>
    import fdn_hip
    fdn_hip._LIB_PATH = os.path.abspath(sys.argv[2])
    print("library:", sys.argv[2])
from basicsr.models.archs import FDN_arch as A  # noqa: E402
from fdn_hip import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for C, H, W, B in ((32, 736, 1280, 8), (64, 368, 640, 8), (24, 736, 1280, 8), (48, 368, 640, 8)):
    m = A.FDSA(C).to("cuda:0").eval()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())).to("cuda:0") * (0.1 if p.dim() > 1 else 1.0) + (0.0 if p.dim() > 1 else 1.0))
    x = torch.randn(B, C, H, W, device="cuda:0")
    ln = (ops.chan_stats(x), torch.ones(C, device="cuda:0"), torch.zeros(C, device="cuda:0"))
    t = {True: [], False: []}
    with torch.no_grad():
        ops.FDSA_FULL_MAX_C = 64
        for full in (True, False):
            ops.FDSA_FULL = full
            m.fused(x, ln=ln, res=x)
        torch.cuda.synchronize()
        for _ in range(reps):
            for full in (True, False):
                ops.FDSA_FULL = full
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    m.fused(x, ln=ln, res=x)
                e1.record()
                torch.cuda.synchronize()
                t[full].append(e0.elapsed_time(e1) / 3)
    ops.FDSA_FULL = True
    med = lambda v: sorted(v)[len(v) // 2]
    px = B * H * W
    print(f"C={C} {H}x{W} B={B}: one launch {med(t[True]):.3f} ms (min {min(t[True]):.3f}), two launches {med(t[False]):.3f} ms (min {min(t[False]):.3f});"
          f" 3C*4 B/px = {3 * C * 4 * px / 1e9:.2f} GB -> {3 * C * 4 * px / med(t[True]) / 1e6:.0f} GB/s of the sub-block's compulsory bytes")
