"""Per-step s_memtime trace of the generic conv1x1 kernel (needs the -DFDN_GEMM_TRACE build: FDN_HIP_LIB=abtest/libT.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
import fdn_hip
if os.environ.get('FDN_HIP_LIB'): fdn_hip._LIB_PATH = os.path.abspath(os.environ['FDN_HIP_LIB'])   # A/B build of the same ABI
from fdn_hip import Conv1x1Desc
dev = torch.device("cuda:0")
B, K, N, H, W = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 345, int(sys.argv[2]) if len(sys.argv) > 2 else 128, 184, 320
P = H * W
x = torch.randn(B, K, H, W, device=dev); w = torch.randn(N, K, device=dev) / K ** .5
res = torch.randn(B, N, H, W, device=dev); out = torch.empty_like(res)
trace = torch.zeros(2048, dtype=torch.int64, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
d = Conv1x1Desc()
d.x[0] = p(x); d.xbs[0] = K * P; d.kseg[0] = K
d.w = p(w); d.out = p(out); d.obs = N * P; d.B, d.K, d.N, d.P = B, K, N, P
d.pro = 0; d.epi = 1; d.res = p(res); d.rbs = N * P; d.act = 0
d.mul = p(trace); d.vec4 = 12345
if len(sys.argv) > 3 and sys.argv[3] == "ln3":          # LN3_GATE prologue: K = 3E, x = o[:, :3E], xb = o[:, 3E:]
    from fdn_hip import ops
    E = K // 3
    o = torch.randn(B, 4 * E, H, W, device=dev)
    st3 = ops.chan_stats(o[:, :3 * E], groups=3); g3 = torch.randn(3 * E, device=dev); b3 = torch.randn(3 * E, device=dev)
    d.x[0] = p(o); d.xbs[0] = 4 * E * P; d.pro = 2; d.ln_group = E
    d.stats = p(st3); d.gamma = p(g3); d.beta = p(b3); d.xb = ctypes.c_void_p(o.data_ptr() + 3 * E * P * 4); d.xbbs = 4 * E * P
for _ in range(2):
    rc = fdn_hip.lib().fdn_conv1x1(ctypes.byref(d), fdn_hip.stream()); assert rc == 0, rc
torch.cuda.synchronize()
t = trace.cpu().view(2, 128, 8)
for wv in (0, 1):
    print("wave", wv * 4)
    t0 = int(t[wv, 0, 0])
    for s in range(0, 40):
        r = [int(v) for v in t[wv, s, :6]]
        if r[0] == 0: break
        print(f" step {s:3d} begin {r[0]-t0:8d}  issue {r[1]-r[0]:6d}  mfma {r[2]-r[1]:6d}  epi {r[3]-r[2]:6d}  stash {r[4]-r[3]:6d}  barrier {r[5]-r[4]:6d}  total {r[5]-r[0]:6d}")
