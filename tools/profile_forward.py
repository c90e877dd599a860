"""Workload for the rocprofv3 passes of a round (run it under rocprofv3, once per pass, see tools/profile_passes.sh):
one warm-up and ONE logged single-stream LPNet -> FDN forward at the bench shape.  The ordered list of C-ABI calls of the logged
forward (bench.py's group keys with their algorithmic FLOPs / bytes) goes to $FDN_CALL_LOG; tools/profile_merge.py aligns it
with the tail of rocprofv3's dispatch list (every entry point launches exactly one kernel).

    python tools/profile_forward.py [--height 720 --width 1280 --batch 8 --dtype f32]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=720)
ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--dtype", default="f32")
a = ap.parse_args()
dev = torch.device("cuda:0")
import fdn_hip
fdn_hip.set_storage_dtype(a.dtype)
net, lp = bench.build_models(dev)
x = bench.make_input(a.batch, a.height, a.width, dev, seed=1000)
with torch.no_grad():
    net(x, ratio_i=lp(x), device=dev)                      # warm-up: FFT tables, packed / folded weights
torch.cuda.synchronize()
lib = fdn_hip.lib()
decl = [ln.split("(")[0].split()[-1] for ln in open(os.path.join(ROOT, "include", "fdn_hip.h")) if ln.startswith("int fdn_")]
calls, orig = [], {}
for n in decl:
    if n in ("fdn_abi_version", "fdn_fft_prepare"):
        continue
    f = getattr(lib, n)
    orig[n] = f

    def wrapped(*args, _f=f, _n=n):
        key, fl, by = bench.describe_call(_n, args)
        calls.append({"entry": _n, "group": key, "flops": fl, "bytes": by})
        return _f(*args)
    setattr(lib, n, wrapped)
with torch.no_grad():
    net(x, ratio_i=lp(x), device=dev)
torch.cuda.synchronize()
for n, f in orig.items():
    setattr(lib, n, f)
out = os.environ.get("FDN_CALL_LOG")
if out:
    json.dump({"shape": [a.batch, a.height, a.width], "dtype": a.dtype, "calls": calls}, open(out, "w"))
print(f"logged {len(calls)} C-ABI calls", flush=True)
