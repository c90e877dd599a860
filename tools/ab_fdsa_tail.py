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
    img = ops.fdsa_tail_pack(wp, g3, b3, C)
    if img is None:
        print(f"C={C}: no in-kernel tail"); return

    def pair():
        o = ops.fdsa_fused(x, stats, wpk, dw, fw)
        return ops.fdsa_out(o, wp, g3, b3, res=x, want_stats=True)

    def one():
        return ops.fdsa_fused_tail(x, stats, wpk, dw, fw, img, res=x, want_stats=True)

    ya, yb = pair(), one()
    torch.cuda.synchronize()
    same = torch.equal(ya, yb) and torch.equal(ya._fdn_stats, yb._fdn_stats)
    d = (ya - yb).abs().max().item()
    print(f"B={B} C={C} {H}x{W}{' edge' if edge else ''}: bit-identical {same} (max |diff| {d:.3e}, stats {(ya._fdn_stats - yb._fdn_stats).abs().max().item():.3e}, "
          f"nan {int(torch.isnan(yb).sum())})", flush=True)
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
    for f in (pair, fused_only, one):
        timeit(f, 2)
    for _ in range(reps):
        res["pair"].append(timeit(pair)); res["fused_only"].append(timeit(fused_only)); res["one"].append(timeit(one))
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
