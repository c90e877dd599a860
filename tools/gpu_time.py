"""Per-entry-point timing of one LPNet->FDN forward (HIP events), progressively larger inputs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
net, lp = bench.build_models(dev)
shapes = [(1, 256, 256), (1, 720, 1280)] if len(sys.argv) < 2 else [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]]
for (B, h, w) in shapes:
    x = bench.make_input(B, h, w, dev, 1)
    for rep in range(2):
        with bench.KernelTimer() as kt:
            t0 = time.perf_counter()
            with torch.no_grad():
                r = lp(x)
                out = net(x, ratio_i=r)[0]
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        agg = kt.summary()
    tot = sum(v[1] for v in agg.values())
    print(f"=== B={B} {h}x{w}: wall {wall*1e3:.1f} ms, sum of kernels {tot:.1f} ms, launches {sum(v[0] for v in agg.values())}", flush=True)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        extra = f" {v[2]/(v[1]*1e-3)/1e12:6.1f} TF/s {v[3]/(v[1]*1e-3)/1e9:7.0f} GB/s(alg)" if v[2] else ""
        print(f"  {k:28s} n={v[0]:5d} {v[1]:10.2f} ms {100*v[1]/tot:5.1f}%{extra}", flush=True)
