"""A/B timing of the fused patch kernels at the bench shapes: tools/ab_fused.py [lib.so ...] (default: the in-tree library).
Each library is timed in its own child process (the binding loads one library per process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        print("==", lib, flush=True)
        subprocess.run([sys.executable, __file__, "--child", lib])
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import fdn_hip
if len(sys.argv) > 2 and sys.argv[2] != "default":
    fdn_hip._LIB_PATH = os.path.abspath(sys.argv[2])
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
B, H0, W0 = 8, 736, 1280

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for lvl in (1, 2):
    C = 32 * 2 ** (lvl - 1); H, W = H0 >> (lvl - 1), W0 >> (lvl - 1)
    E, Hd = int(C * 1.2), int(C * 2.7)
    r = lambda *s: torch.randn(*s, device=dev)
    x = r(B, C, H, W); st = ops.chan_stats(x); g, b_ = r(C), r(C)
    wh = r(4 * E, C) / C ** .5; dw, fw = r(4 * E, 1, 3, 3), r(E, 1, 1, 8, 5)
    wpk = ops.fdsa_pack(wh, g, b_)
    print(f"L{lvl} fdsa_fused {timeit(lambda: ops.fdsa_fused(x, st, wpk, dw, fw)):.3f} ms", flush=True)
    h = r(B, Hd, H, W)
    w0, w2, fa, fp = r(Hd, 1, 3, 3), r(Hd, 1, 3, 3), r(Hd, 1, 1, 8, 5), r(Hd, 1, 1, 8, 5)
    print(f"L{lvl} fdffn_mid {timeit(lambda: ops.fdffn_mid(h, w0, w2, fa, fp)):.3f} ms", flush=True)
    if hasattr(ops, "fdffn_fused") and C in getattr(ops, "FDFFN_FUSED_C", ()):
        wi = r(Hd, C) / C ** .5
        pk = ops.fdffn_pack(wi, g, b_, fa, fp)
        print(f"L{lvl} conv1x1 ffn_in {timeit(lambda: ops.conv1x1(x, wi, ln=(st, g, b_))):.3f} ms", flush=True)
        print(f"L{lvl} fdffn_fused {timeit(lambda: ops.fdffn_fused(x, st, pk, w0, w2)):.3f} ms", flush=True)
    wg = r(2 * Hd, 1, 3, 3)
    print(f"L{lvl} dwconv_gate {timeit(lambda: ops.dwconv_gate(h, wg)):.3f} ms", flush=True)
    hid = r(B, 4 * E, H, W)
    print(f"L{lvl} fdsa_core {timeit(lambda: ops.fdsa_core(hid, dw, fw)):.3f} ms", flush=True)
