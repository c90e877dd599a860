"""Round timeline of the T form of fdn_fdsa_fused (12-wave persistent workgroups): needs the -DFDN_FUSED_TRACE build.
    python tools/fused_trace_t.py abx/lib_trace.so
Every wave of workgroups 0 .. 127 stamps s_memtime before and behind each barrier of its fourth tile (three rounds x barriers A, B, C).
Printed per wave role: time spent between barriers (work) and at barriers (waiting for the slowest wave)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import numpy as np, torch
import fdn_hip
fdn_hip._LIB_PATH = os.path.abspath(sys.argv[1])
from fdn_hip import ops
C, H, W, B = 32, 736, 1280, 8
E = int(1.2 * C)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(B, C, H, W, generator=g).to(dev)
w = (torch.randn(4 * E, C, generator=g) / C ** .5).to(dev)
dw = (torch.randn(4 * E, 9, generator=g) / 3).to(dev)
fw = torch.randn(E, 8, 5, generator=g).to(dev)
stats = ops.chan_stats(x)
wpk = ops.fdsa_pack(w, torch.ones(C, device=dev), torch.zeros(C, device=dev))
lib = ctypes.CDLL(fdn_hip._LIB_PATH)
buf = np.zeros(512 * 4 * 64, dtype=np.uint64)
for it in range(3):
    lib.fdn_debug_fused_trace_clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.fdsa_fused(x, stats, wpk, dw, fw); e1.record(); torch.cuda.synchronize()
print("kernel time %.3f ms" % e0.elapsed_time(e1))
assert lib.fdn_debug_fused_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes)) == 0
t = buf[:128 * 12 * 64].reshape(128, 12, 64).astype(np.int64)
xs = t[:, 8:12, 40:45]
if (xs > 0).all():
    dx = np.diff(xs, axis=2).mean(axis=(0, 1))
    print("producers, round before the last: window (B -> both chunks issued, C inside) %.0f, raw loads issued %.0f, -> A passed %.0f, convert %.0f" % tuple(dx))
n = 18                                             # 3 rounds x (A, B, C) x (arrive, leave)
ok = (t[:, :, :n] > 0).all(axis=(1, 2))
print("workgroups traced:", int(ok.sum()))
t = t[ok][:, :, :n]
t0 = t[:, :, :1].min(axis=1, keepdims=True)
rel = t - t0
names = []
for r in range(3):
    names += [f"r{r} wait A", f"r{r} A->B (rows | convert)", f"r{r} wait B", f"r{r} B->C (columns | MFMA)", f"r{r} wait C", f"r{r} C->A (inverse)"]
d = np.diff(t, axis=2)                              # 17 intervals: [wait A, A->B, wait B, B->C, wait C, C->A', ...]
roles = {"team 0 (waves 0-3)": slice(0, 4), "team 1 (waves 4-7)": slice(4, 8), "producers (8-11)": slice(8, 12)}
print("%-30s" % "interval (clocks)" + "".join("%22s" % k for k in roles))
for i in range(17):
    print("%-30s" % names[i] + "".join("%22.0f" % d[:, sl, i].mean() for sl in roles.values()))
print("tile (first A arrive -> last C leave): %.0f clocks" % (t[:, :, 17] - t[:, :, 0]).mean())
for k, sl in roles.items():
    wait = d[:, sl, 0::2].sum(axis=2).mean()
    work = d[:, sl, 1::2].sum(axis=2).mean()
    print(f"{k}: at barriers {wait:.0f}, between {work:.0f}")
