"""A handful of conv1x1 launches at the real level-2/3 shapes, for rocprofv3 --pmc passes (tools/pmc_report.py reads the csv)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
B = 8
r = lambda *s: torch.randn(*s, device=dev)
for (C, H, W) in ((64, 368, 640), (128, 184, 320)):
    E, Hd = int(C * 1.2), int(C * 2.7)
    x = r(B, C, H, W); st = ops.chan_stats(x); g, b_ = r(C), r(C)
    wh = r(4 * E, C) / C ** .5
    o = r(B, 4 * E, H, W); st3 = ops.chan_stats(o[:, :3 * E], groups=3); g3, b3 = r(3 * E), r(3 * E); wo = r(C, 3 * E) / (3 * E) ** .5
    h = r(B, Hd, H, W); wo2 = r(C, Hd) / Hd ** .5; wi = r(Hd, C) / C ** .5
    for _ in range(2):
        ops.conv1x1(x, wh, ln=(st, g, b_))                                                   # to_hidden
        ops.conv1x1(o[:, :3 * E], wo, ln3_gate=(st3, g3, b3, o[:, 3 * E:]), res=x)           # attn_out
        ops.conv1x1(x, wi, ln=(st, g, b_))                                                   # ffn_in
        ops.conv1x1(h, wo2, res=x, want_stats=True)                                          # ffn_out
    torch.cuda.synchronize()
