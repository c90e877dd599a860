"""(round 6) The FDSA sub-block at the bench shapes: fdn_fdsa_fused + fdn_fdsa_out (two launches, 4E-plane hand-off through HBM) against
fdn_fdsa_fused_tail (one launch: the producing workgroup runs the tail on its own tile).  Interleaved timing + bit-for-bit comparison.
    python tools/ab_fdsa_tail.py [reps] [--small]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
import fdn_hip
for a_ in sys.argv[1:]:
    if a_.endswith(".so"):
        fdn_hip._LIB_PATH = os.path.abspath(a_)          # another build of the library (tools/ab_build.sh)
        print("library:", a_)
from fdn_hip import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 7
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
r = lambda *s: torch.randn(*s, device=dev, generator=g)


def case(B, C, H, W, time_it=True, edge=False):
    E = int(C * 1.2)
    x = r(B, C, H, W)
    if edge:
        x[:, :, :8, :32] = 0.0
        x[:, 1::5, 8:16, :] *= 1e-12
        x[:, :, 16:24, 8:40] = 0.5
    stats = ops.chan_stats(x)
    gm, bt = r(C), r(C)
    wh = r(4 * E, C) / C ** .5
    dw, fw = r(4 * E, 1, 3, 3), r(E, 1, 1, 8, 5)
    wp = r(C, 3 * E) / (3 * E) ** .5
    g3, b3 = r(3 * E), r(3 * E)
    wpk = ops.fdsa_pack(wh, gm, bt)
    Hd = int(C * 2.7)
    wi, g2, b2 = r(Hd, C) / C ** .5, r(C), r(C)           # the FDFFN behind the FDSA: project_in with its LayerNorm
    cache = ops.WeightCache()
    img = ops.fdsa_tail_pack(wp, g3, b3, C)
    if img is None:
        print(f"C={C}: no in-kernel tail"); return
    imgp = ops.fdsa_tail_pack(wp, g3, b3, C, pin=ops.fold_ln(wi, None, g2, b2))

    def pair(with_pin=False):
        o = ops.fdsa_fused(x, stats, wpk, dw, fw)
        y = ops.fdsa_out(o, wp, g3, b3, res=x, want_stats=True)
        if with_pin:
            y._fdn_pin = ops.conv1x1(y, wi, ln=(y._fdn_stats, g2, b2), cache=(cache, "pi"))
        return y

    def one(with_pin=False):
        if with_pin:
            return ops.fdsa_fused_tail(x, stats, wpk, dw, fw, imgp, res=x, want_stats=True, Hd=Hd)
        return ops.fdsa_fused_tail(x, stats, wpk, dw, fw, img, res=x, want_stats=True)

    ya, yb = pair(imgp is not None), one()
    torch.cuda.synchronize()
    same = torch.equal(ya, yb) and torch.equal(ya._fdn_stats, yb._fdn_stats)
    d = (ya - yb).abs().max().item()
    print(f"B={B} C={C} {H}x{W}{' edge' if edge else ''}: bit-identical {same} (max |diff| {d:.3e}, stats {(ya._fdn_stats - yb._fdn_stats).abs().max().item():.3e}, "
          f"nan {int(torch.isnan(yb).sum())})", flush=True)
    if imgp is not None:
        yc = one(True)
        torch.cuda.synchronize()
        samep = torch.equal(ya, yc) and torch.equal(ya._fdn_stats, yc._fdn_stats) and torch.equal(ya._fdn_pin, yc._fdn_pin)
        print(f"   with project_in {C}->{Hd}: bit-identical {samep} (h max |diff| {(ya._fdn_pin - yc._fdn_pin).abs().max().item():.3e}, nan {int(torch.isnan(yc._fdn_pin).sum())})", flush=True)
    if not time_it:
        return

    def timeit(f, n=6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    fused_only = lambda: ops.fdsa_fused(x, stats, wpk, dw, fw)
    res = {"pair": [], "fused_only": [], "one": []}
    fs = [("pair", pair), ("fused_only", fused_only), ("one", one)]
    if imgp is not None:
        res.update({"pair+project_in": [], "one+project_in": []})
        fs += [("pair+project_in", lambda: pair(True)), ("one+project_in", lambda: one(True))]
    for _, f in fs:
        timeit(f, 2)
    for _ in range(reps):
        for k, f in fs:
            res[k].append(timeit(f))
    print("   " + "   ".join(f"{k}: {statistics.median(v):.3f} ms (min {min(v):.3f})" for k, v in res.items()), flush=True)


small = "--small" in sys.argv
case(1, 32, 16, 40, time_it=False)
case(2, 32, 24, 72, time_it=False, edge=True)
case(1, 24, 32, 64, time_it=False)
case(2, 32, 64, 96, time_it=False, edge=True)
case(1, 64, 16, 40, time_it=False)
case(2, 64, 24, 72, time_it=False, edge=True)
case(2, 48, 32, 64, time_it=False, edge=True)
if not small:
    case(8, 32, 736, 1280)
    case(8, 24, 400 // 8 * 8, 608)
    case(8, 64, 368, 640)
    case(8, 48, 200, 304)
