"""A few launches of the FDSA sub-block at the level-1 / level-2 bench shapes for rocprofv3 --pmc passes (program directly after `--`):
one launch (fdn_fdsa_full) and the two launches it replaces.  python3 tools/pmc_fdsa_full.py [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("fdn-tip2025_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402

from basicsr.models.archs import FDN_arch as A  # noqa: E402
from fdn_hip import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for C, H, W, B in ((32, 736, 1280, 8), (64, 368, 640, 8)):
    m = A.FDSA(C).to("cuda:0").eval()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())).to("cuda:0") * (0.1 if p.dim() > 1 else 1.0) + (0.0 if p.dim() > 1 else 1.0))
    x = torch.randn(B, C, H, W, device="cuda:0")
    ln = (ops.chan_stats(x), torch.ones(C, device="cuda:0"), torch.zeros(C, device="cuda:0"))
    ops.FDSA_FULL_MAX_C = 64
    with torch.no_grad():
        for full in (True, False):
            ops.FDSA_FULL = full
            for _ in range(n):
                m.fused(x, ln=ln, res=x)
    torch.cuda.synchronize()
