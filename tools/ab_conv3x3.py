"""Timing of the dense 3x3 convs of the B = 8 720p forward for several library builds: tools/ab_conv3x3.py [lib.so | default ...]"""
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

def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (Cin, Cout, H, W) in ((64, 32, 736, 1280), (128, 64, 368, 640), (32, 64, 368, 640), (64, 128, 184, 320), (24, 24, 368, 640), (48, 48, 184, 320)):
    x = torch.randn(8, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) / (3 * Cin ** 0.5)
    t = timeit(lambda: ops.conv2d(x, w, None, pad=1))
    ref = torch.nn.functional.conv2d(x[:1, :, :64].double(), w.double(), padding=1)[:, :, 1:-1]
    got = ops.conv2d(x[:1, :, :64].contiguous(), w, None, pad=1)[:, :, 1:-1].double()
    print(f"{Cin:4d} -> {Cout:4d} {H}x{W}: {t:.3f} ms  {2 * 9 * Cin * Cout * H * W * 8 / t / 1e9:.0f} TFLOP/s  rel err {((got - ref).norm() / ref.norm()).item():.2e}", flush=True)
