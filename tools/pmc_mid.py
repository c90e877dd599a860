"""A few launches of fdn_fdffn_mid at the level-1 bench shape (fp32 in/out, then bf16 in/out) for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
B, Hd, H, W = 8, 86, 736, 1280
r = lambda *s: torch.randn(*s, device=dev)
h32 = r(B, Hd, H, W); h16 = h32.to(torch.bfloat16)
w0, w2, fa, fp = r(Hd, 1, 3, 3), r(Hd, 1, 3, 3), r(Hd, 1, 1, 8, 5), r(Hd, 1, 1, 8, 5)
for _ in range(3):
    ops.fdffn_mid(h32, w0, w2, fa, fp)
    ops.fdffn_mid(h16, w0, w2, fa, fp, out_dtype=torch.bfloat16)
torch.cuda.synchronize()
