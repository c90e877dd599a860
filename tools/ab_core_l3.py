import ctypes, os, statistics, sys
sys.path.insert(0, "/root/repo/fdn-tip2025_amd")
import torch, fdn_hip
dev = torch.device("cuda:0")
B, E, H, W = 8, 153, 184, 320
hid = torch.randn(B, 4 * E, H, W, device=dev); dw = torch.randn(4 * E, 1, 3, 3, device=dev); fw = torch.randn(E, 1, 1, 8, 5, device=dev)
out = torch.empty_like(hid)
P = lambda t: ctypes.c_void_p(t.data_ptr())
libs = []
for p in sys.argv[1:]:
    l = ctypes.CDLL(fdn_hip.lib_path() if p == "default" else os.path.abspath(p)); fdn_hip._declare(l); libs.append((p, l))
def run(l, n):
    for _ in range(n): assert l.fdn_fdsa_core(P(hid), P(dw), P(fw), P(out), B, E, H, W, fdn_hip.stream()) == 0
res = {p: [] for p, _ in libs}
for rnd in range(8):
    for p, l in libs:
        run(l, 3); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(l, 20); b.record(); torch.cuda.synchronize()
        res[p].append(a.elapsed_time(b) / 20)
for p in res: print(f"{p}: {statistics.median(res[p]):.4f} ms (min {min(res[p]):.4f})")
