"""Timing + fp64 check of the fourier_fuse column transforms ((H + 2)-row maps: 738 = 41 * 18 and 370 = 37 * 10 rows) and the row
transforms of the same maps (W + 2 = 1282, 642): tools/ab_fft_awkward.py [lib.so | default ...]"""
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

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (B, C, H, W) in ((8, 12, 738, 1282), (8, 24, 370, 642)):
    x = torch.randn(B, C, H, W, device=dev)
    z = ops.rfft_rows(x)
    ref = torch.fft.fft(torch.view_as_complex(z[:1].double().contiguous()), dim=-2)
    oa, og = ops.fft_cols_fwd(z.clone(), True, True, rd_before=False, fix_real=True)
    err = ((oa[:1].double() - ref.abs()).norm() / ref.abs().norm()).item()
    zr = torch.fft.rfft(x[:1].double(), dim=-1)
    rerr = ((torch.view_as_complex(z[:1].double().contiguous()) - zr).norm() / zr.norm()).item()
    zz = z.clone()
    print(f"{H}x{W}: rfft_rows {timeit(lambda: ops.rfft_rows(x)):.3f} ms (rel err {rerr:.2e})   "
          f"fft_cols_fwd {timeit(lambda: ops.fft_cols_fwd(zz, True, True, rd_before=False, fix_real=True)):.3f} ms (|.| rel err {err:.2e})", flush=True)
