"""conv1x1 time by shape inside one real forward (B=8 720p)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench, fdn_hip
dev = torch.device("cuda:0")
net, lp = bench.build_models(dev)
x = bench.make_input(8, 720, 1280, dev, 1)
lib = fdn_hip.lib(); orig = lib.fdn_conv1x1; recs = []
def wrapped(d, s):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(d, s); e1.record()
    o = d._obj; recs.append(((o.K, o.N, o.P, o.pro, o.epi, o.B), e0, e1)); return r
for rep in range(2):
    recs.clear(); lib.fdn_conv1x1 = wrapped
    with torch.no_grad():
        r = lp(x); net(x, ratio_i=r)
    torch.cuda.synchronize(); lib.fdn_conv1x1 = orig
agg = collections.defaultdict(lambda: [0, 0.0])
for k, e0, e1 in recs:
    agg[k][0] += 1; agg[k][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print("total", tot)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    K, N, P, pro, epi, B = k
    fl = 2.0 * B * K * N * P * v[0]; by = 4.0 * B * P * (K + N) * v[0]
    print(f"K={K:4d} N={N:4d} P={P:7d} pro={pro} epi={epi} n={v[0]:3d} {v[1]:8.2f} ms {100*v[1]/tot:5.1f}%  {fl/(v[1]*1e-3)/1e12:6.1f} TF/s {by/(v[1]*1e-3)/1e9:6.0f} GB/s")
