"""GPU bring-up check #1: FDSA / FDFFN composed from the HIP ops vs the oracle and the fixtures."""
import os, sys, json, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "fdn-tip2025_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import fdn_oracle as O
from common import fixture, fixture_weights, rel_rms
import fdn_hip
from fdn_hip import ops

dev = torch.device("cuda:0")
print("lib abi", fdn_hip.lib().fdn_abi_version(), torch.cuda.get_device_name(0))

def report(name, got, ref32, t64):
    got = got.cpu()
    print(f"{name:28s} rel_rms(got,fp64)={rel_rms(got,t64):.3e} rel_rms(ref32,fp64)={rel_rms(ref32,t64):.3e} "
          f"rel_rms(got,ref)={rel_rms(got,ref32):.3e} maxabs={float((got-ref32).abs().max()):.3e}", flush=True)

def g(t): return t.to(dev).contiguous()

# ---- plain GEMM checks, odd shapes ----------------------------------------------------
for (B,K,N,H,W) in [(2,32,152,16,24),(1,114,32,8,40),(1,459,128,8,8),(2,3,32,5,7),(1,86,32,46,21),(1,128,612,16,16)]:
    torch.manual_seed(1)
    x = torch.randn(B,K,H,W); w = torch.randn(N,K)/K**0.5; bias = torch.randn(N); res = torch.randn(B,N,H,W)
    ref = torch.nn.functional.conv2d(x.double(), w.double().view(N,K,1,1), bias.double()) + res.double()
    got = ops.conv1x1(g(x), g(w), g(bias), res=g(res))
    print("gemm", (B,K,N,H,W), "rel", rel_rms(got.cpu(), ref), flush=True)

# LN prologue
x = torch.randn(2,64,16,24)*2+1; gam = torch.randn(64); bet = torch.randn(64); w = torch.randn(172,64)/8
ref = torch.nn.functional.conv2d(O.ln_chan(x.double(), gam.double(), bet.double()), w.double().view(172,64,1,1))
st = ops.chan_stats(g(x))
got = ops.conv1x1(g(x), g(w), ln=(st, g(gam), g(bet)))
print("gemm+LN rel", rel_rms(got.cpu(), ref))
got = ops.layernorm_chan(g(x), g(gam), g(bet))
print("layernorm rel", rel_rms(got.cpu(), O.ln_chan(x.double(), gam.double(), bet.double())))

def fdsa_hip(x, sd):
    E = sd["fft"].shape[0]
    hidden = ops.conv1x1(x, sd["to_hidden.weight"])
    o = ops.fdsa_core(hidden, sd["to_hidden_dw.weight"], sd["fft"])
    st = ops.chan_stats(o[:, :3*E], groups=3)
    gam = torch.cat([sd[f"norm{i}.body.weight"] for i in (1,2,3)]).contiguous()
    bet = torch.cat([sd[f"norm{i}.body.bias"] for i in (1,2,3)]).contiguous()
    return ops.conv1x1(o[:, :3*E], sd["project_out.weight"], ln3_gate=(st, gam, bet, o[:, 3*E:])), hidden, o

def fdffn_hip(x, sd):
    h = ops.conv1x1(x, sd["project_in.weight"])
    y = ops.fdffn_mid(h, sd["space.0.weight"], sd["space.2.weight"], sd["ffta"], sd["fftp"])
    z = ops.dwconv_gate(y, sd["dwconv.weight"])
    return ops.conv1x1(z, sd["project_out.weight"]), h, y, z

for name in ["fdsa_c32", "fdsa_c64", "fdsa_c128"]:
    try:
        fx = fixture(name); sd = fixture_weights(name, fx["shapes"])
        P64 = {"."+k: v.double() for k,v in sd.items()}
        with torch.no_grad(): t64 = O.fdsa(fx["x"].double(), P64, "")
        got, hidden, o = fdsa_hip(g(fx["x"]), {k: g(v) for k,v in sd.items()})
        torch.cuda.synchronize()
        report(name, got, fx["y"], t64)
    except Exception: traceback.print_exc()

for name in ["fdffn_c32", "fdffn_c64", "fdffn_c128"]:
    try:
        fx = fixture(name); sd = fixture_weights(name, fx["shapes"])
        P64 = {"."+k: v.double() for k,v in sd.items()}
        with torch.no_grad(): t64 = O.fdffn(fx["x"].double(), P64, "")
        got, h, y, z = fdffn_hip(g(fx["x"]), {k: g(v) for k,v in sd.items()})
        torch.cuda.synchronize()
        report(name, got, fx["y"], t64)
        # stage-level diagnostics
        import torch.nn.functional as F
        with torch.no_grad():
            xx = F.conv2d(fx["x"].double(), P64[".project_in.weight"]); hd = xx.shape[1]
            s = F.conv2d(F.gelu(F.conv2d(xx, P64[".space.0.weight"], padding=1, groups=hd)), P64[".space.2.weight"], padding=1, groups=hd)
            zf = O.replace_denormals(torch.fft.rfft2(O.to_patches(xx)))
            zf = O.polar(zf.abs()*P64[".ffta"], zf.angle()-P64[".fftp"])
            yy = O.from_patches(torch.fft.irfft2(zf, s=(8,8))) + s
            a,b = F.conv2d(yy, P64[".dwconv.weight"], padding=1, groups=hd).chunk(2,1); zz = F.gelu(a)*b
        print("   stages: proj_in", rel_rms(h.cpu(), xx), "mid", rel_rms(y.cpu(), yy), "gate", rel_rms(z.cpu(), zz), flush=True)
    except Exception: traceback.print_exc()

# dwconv plain + img maps
x = torch.randn(2,5,37,70); w = torch.randn(5,1,3,3)
ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1, groups=5)
print("dw3x3 rel", rel_rms(ops.dwconv3x3(g(x), g(w)).cpu(), ref))
img = torch.rand(2,3,37,70); w1m=torch.randn(32,3,1,1); w3m=torch.randn(32,1,3,3); w1a=torch.randn(32,3,1,1); w3a=torch.randn(32,1,3,3)
F = torch.nn.functional
rm = F.conv2d(F.conv2d(img.double(), w1m.double()), w3m.double(), padding=1, groups=32)
ra = F.conv2d(F.conv2d(img.double(), w1a.double()), w3a.double(), padding=1, groups=32)
m,a = ops.img_mod_maps(g(img), g(w1m), g(w3m), g(w1a), g(w3a))
print("img maps rel", rel_rms(m.cpu(), rm), rel_rms(a.cpu(), ra))
print("CHECK1 DONE")
