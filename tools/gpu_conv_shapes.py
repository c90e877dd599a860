"""fdn_conv2d / convT / fft time by shape inside one real forward (B=8 720p)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench, fdn_hip
dev = torch.device("cuda:0")
net, lp = bench.build_models(dev)
x = bench.make_input(8, 720, 1280, dev, 1)
lib = fdn_hip.lib(); recs = []
names = {"fdn_conv2d": lambda a: ("conv2d Cin,H,W,Cout,K,s", a[6], a[7], a[8], a[9], a[10], a[12]),
         "fdn_conv_transpose4x4s2": lambda a: ("convT Cin,H,W,Cout", a[5], a[6], a[7], a[8]),
         "fdn_rfft_rows": lambda a: ("rfft rows,W", a[2].value, a[3]),
         "fdn_irfft_rows": lambda a: ("irfft planes,H,W", a[4].value, a[5], a[6]),
         "fdn_fft_cols_fwd": lambda a: ("cols_fwd planes,H,Wf", a[3].value, a[4], a[5]),
         "fdn_fft_cols_inv_polar": lambda a: ("inv_polar planes,H,Wf", a[5].value, a[6], a[7]),
         "fdn_resample": lambda a: ("resample planes,H,W,mode", a[2].value, a[3], a[4], a[5]),
         "fdn_layernorm_chan": lambda a: ("ln B,C,P", a[4], a[5], a[6])}
orig = {n: getattr(lib, n) for n in names}
def mk(n):
    def w(*a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig[n](*a); e1.record(); recs.append((names[n](a), e0, e1)); return r
    return w
for rep in range(2):
    recs.clear()
    for n in names: setattr(lib, n, mk(n))
    with torch.no_grad():
        r = lp(x); net(x, ratio_i=r)
    torch.cuda.synchronize()
    for n in names: setattr(lib, n, orig[n])
agg = collections.defaultdict(lambda: [0, 0.0])
for k, e0, e1 in recs:
    agg[k][0] += 1; agg[k][1] += e0.elapsed_time(e1)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1]:8.2f} ms n={v[0]:3d} {k}")
