"""Experiment: split the batch over 2 HIP streams so MFMA-bound and HBM-bound kernels of different halves overlap."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda:0")
net, lp = bench.build_models(dev)
x = bench.make_input(8, 720, 1280, dev, 1)
def fwd(xx):
    with torch.no_grad():
        return net(xx, ratio_i=lp(xx))[0]
def run_single():
    return fwd(x)
def run_split(ns):
    streams = run_split.streams[:ns]
    outs = []
    cur = torch.cuda.current_stream()
    for i, s in enumerate(streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(fwd(x[i * (8 // ns):(i + 1) * (8 // ns)].contiguous()))
    for s in streams: cur.wait_stream(s)
    return torch.cat(outs)
run_split.streams = [torch.cuda.Stream() for _ in range(4)]
ref = run_single()
for name, fn in (("1 stream B=8", run_single), ("2 streams B=4", lambda: run_split(2)), ("4 streams B=2", lambda: run_split(4))):
    out = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {dt*1e3:.1f} ms/step  {8/dt:.2f} img/s  equal={torch.equal(out, ref)}", flush=True)
