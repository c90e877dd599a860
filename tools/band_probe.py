"""Does the Infinity Cache (256 MB) carry the FDSA hand-off (out1|out2|out3|v_value: 572 MB per 720p image in fp32) when the pair
fdn_fdsa_fused -> fdn_fdsa_out runs band by band?  Times the pair on one 736 x 1280 image in one piece and as 2 / 4 / 8 row bands
(emulated as separate images of the band's height, the hand-off tensor reused from band to band)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
C, E, W = 32, 38, 1280
r = lambda *s: torch.randn(*s, device=dev)
wh = r(4 * E, C) / C ** .5; g, b_ = r(C), r(C); dw, fw = r(4 * E, 1, 3, 3), r(E, 1, 1, 8, 5)
wpk = ops.fdsa_pack(wh, g, b_)
wo = r(C, 3 * E) / (3 * E) ** .5; g3, b3 = r(3 * E), r(3 * E)
NIMG = 4                                     # images in flight (a sub-batch): the bands of all of them are walked in turn
for nb, H in ((1, 736), (2, 368), (4, 184), (8, 96), (16, 48)):
    xs = [r(1, C, H, W) for _ in range(nb * NIMG)]
    sts = [ops.chan_stats(x) for x in xs]
    def run():
        for x, st in zip(xs, sts):
            o = ops.fdsa_fused(x, st, wpk, dw, fw)
            ops.fdsa_out(o, wo, g3, b3, res=x, want_stats=True)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{nb} band(s) of {H} rows: {e0.elapsed_time(e1) / 5 / NIMG * 736 / (H * nb):.3f} ms per 736-row image (hand-off {4 * E * H * W * 4 / 1e6:.0f} MB per band)", flush=True)
