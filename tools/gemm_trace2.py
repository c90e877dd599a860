"""s_memtime trace of conv1x1_smallk_stream_vec_kernel (level-3 to_hidden, K=128 N=612): -DFDN_GEMM_TRACE build via FDN_HIP_LIB."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
import fdn_hip
if os.environ.get('FDN_HIP_LIB'): fdn_hip._LIB_PATH = os.path.abspath(os.environ['FDN_HIP_LIB'])   # A/B build of the same ABI
from fdn_hip import Conv1x1Desc, ops
dev = torch.device("cuda:0")
B, K, N, H, W = 8, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 612, 184, 320
P = H * W
x = torch.randn(B, K, H, W, device=dev); w = torch.randn(N, K, device=dev) / K ** .5
out = torch.empty(B, N, H, W, device=dev); st = ops.chan_stats(x)
trace = torch.zeros(2048, dtype=torch.int64, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
d = Conv1x1Desc()
d.x[0] = p(x); d.xbs[0] = K * P; d.kseg[0] = K
d.w = p(w); d.out = p(out); d.obs = N * P; d.B, d.K, d.N, d.P = B, K, N, P
d.pro = 1; d.epi = 0; d.act = 0; d.stats = p(st)
d.mul = p(trace); d.vec4 = 12345
for _ in range(2):
    rc = fdn_hip.lib().fdn_conv1x1(ctypes.byref(d), fdn_hip.stream()); assert rc == 0, rc
torch.cuda.synchronize()
t = trace.cpu().view(2, 128, 8)
for wv in (0, 1):
    print("wave", wv * 4)
    t0 = int(t[wv, 0, 0])
    for s in range(0, 84):
        r = [int(v) for v in t[wv, s, :7]]
        if r[0] == 0: break
        top = f" tile-top {r[0]-r[6]:6d}" if r[6] else ""
        print(f" step {s:3d} begin {r[0]-t0:8d}  mfma {r[1]-r[0]:6d}  epi {r[2]-r[1]:6d}  stash {r[3]-r[2]:6d}  barrier {r[4]-r[3]:6d}  total {r[4]-r[0]:6d}{top}")
