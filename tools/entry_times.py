"""Per-entry-point milliseconds of one instrumented forward at the bench shape (bench.py's KernelTimer), best of n: python tools/entry_times.py [n] [entry substrings ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
flt = [a for a in sys.argv[2:] if not a.endswith(".so") and "=" not in a]
for a in sys.argv[2:]:
    if "=" in a:                              # ops switch for an A/B run, e.g. AFF_MULTIRES=0
        sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
        from fdn_hip import ops as _ops
        k_, v_ = a.split("=")
        assert hasattr(_ops, k_), k_
        setattr(_ops, k_, bool(int(v_)))
        print("ops." + k_, "=", bool(int(v_)))
    if a.endswith(".so"):                     # another build of the same ABI (A/B)
        sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
        import fdn_hip
        fdn_hip._LIB_PATH = os.path.abspath(a)
        print("library:", a)
dev = torch.device("cuda:0")
net, lp = bench.build_models(dev)
x = bench.make_input(8, 720, 1280, dev, 1000)
best = {}
for _ in range(n + 1):
    with bench.KernelTimer() as kt, torch.no_grad():
        net(x, ratio_i=lp(x), device=dev)
    for k, v in kt.summary().items():
        if not flt or any(f in k for f in flt):
            best[k] = min(best.get(k, 1e9), v[2])
for k, v in sorted(best.items(), key=lambda kv: -kv[1]):
    print(f"{v:8.3f} ms  {k}")
print(f"{sum(best.values()):8.3f} ms  total of the listed groups")
