"""Shader clock / power while fdn_fdffn_mid loops with fp32 and with bf16 operands (same instruction stream, half the bytes):
polls `rocm-smi` from a child process for a few seconds per case.  tools/clock_probe.py"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
B, Hd, H, W = 8, 86, 736, 1280
r = lambda *s: torch.randn(*s, device=dev)
h32 = r(B, Hd, H, W); h16 = h32.to(torch.bfloat16)
w0, w2, fa, fp = r(Hd, 1, 3, 3), r(Hd, 1, 3, 3), r(Hd, 1, 1, 8, 5), r(Hd, 1, 1, 8, 5)

def poll(tag, stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            lines = [l.strip() for l in t.splitlines() if ("sclk" in l or "mclk" in l or "Power" in l or "power" in l)]
            out.append(" | ".join(lines)[:400])
        except Exception as e:
            out.append(f"rocm-smi failed: {e}")
            break
        time.sleep(0.5)

for tag, x, od in (("fp32", h32, torch.float32), ("bf16", h16, torch.bfloat16), ("idle", None, None)):
    stop, out = threading.Event(), []
    th = threading.Thread(target=poll, args=(tag, stop, out)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 4.0:
        if x is not None:
            for _ in range(20): ops.fdffn_mid(x, w0, w2, fa, fp, out_dtype=od)
            torch.cuda.synchronize(); n += 20
        else:
            time.sleep(0.2)
    dt = time.time() - t0
    stop.set(); th.join()
    print(f"== {tag}: {dt / max(n, 1) * 1e3:.3f} ms per launch" if n else f"== {tag}")
    for l in out[-4:]: print("   ", l)
