"""Inside the FDSA of the block tools/flip_trace.py points at: the same input with the forward's LayerNorm statistics and with fresh ones (they differ in the last
bits) through to_hidden -> fdsa_core, and where (channel kind, channel, 8 x 8 patch) the core's outputs part; then the float64 spectra of q, k, v of that patch.
tools/flip_inside.py [seed, default 103] [block, default net_p.decoder_level3.5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fdn_hip
from fdn_hip import ops
from common import fixture, fdn_weights
from basicsr.models.archs import FDN_arch as A
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 103
bname = sys.argv[2] if len(sys.argv) > 2 else "net_p.decoder_level3.5"
dev = torch.device("cuda:0")
fx = fixture("fdn_tamed_96x160")
m = A.FDN().to(dev).eval(); m.load_state_dict(fdn_weights(tame=float(fx["tame"])), strict=True)
blk = dict(m.named_modules())[bname]
saved = {}
def hook(mod, inp, out):
    saved["x"] = inp[0][0].detach().clone(); saved["st"] = inp[0][0]._fdn_stats.detach().clone()
h = blk.register_forward_hook(hook)
with torch.no_grad():
    m((fx["x"] + 6e-8 * torch.randn(fx["x"].shape, generator=torch.Generator().manual_seed(seed))).to(dev), ratio_i=fx["ratio"].to(dev), device=dev)
x, st_fwd = saved["x"], saved["st"]
st_fresh = ops.chan_stats(x)
att = blk.attn; e = att.expand_dim
W = lambda p: p.detach()
outs = []
with torch.no_grad():
    for st in (st_fwd, st_fresh.view_as(st_fwd)):
        hidden = ops.conv1x1(x, W(att.to_hidden.weight).flatten(1), ln=(st,) + blk.norm1.params())
        o = ops.fdsa_core(hidden, W(att.to_hidden_dw.weight), W(att.fft))
        outs.append((hidden.clone(), o.clone()))
rel = lambda u, v: float((u - v).double().pow(2).mean().sqrt() / v.double().pow(2).mean().sqrt())
print("hidden:", rel(outs[0][0], outs[1][0]), " core out:", rel(outs[0][1], outs[1][1]))
d = (outs[0][1] - outs[1][1]).abs()
B, C4, H, Wd = d.shape
dp = d.reshape(B, 4, e, H // 8, 8, Wd // 8, 8).amax((4, 6))          # [B, kind, e, py, px]
top = torch.topk(dp.flatten(), 6)
for v, i in zip(top.values.tolist(), top.indices.tolist()):
    idx = list(torch.unravel_index(torch.tensor(i), dp.shape))
    print("  max |d| %.2e at (b, kind, e, py, px) =" % v, [int(t) for t in idx])
b_, kind, ee, py, px = [int(t) for t in torch.unravel_index(top.indices[0], dp.shape)]
# float64 look at that patch: dw conv of hidden (both variants), rfft2, magnitudes of q, k, v bins
import torch.nn.functional as F
for tag, (hidden, o) in zip(("forward statistics", "fresh statistics"), outs):
    hd = F.conv2d(hidden.double().cpu(), W(att.to_hidden_dw.weight).double().cpu(), padding=1, groups=4 * e)
    sp = []
    for k3 in range(3):
        patch = hd[b_, k3 * e + ee, py * 8:py * 8 + 8, px * 8:px * 8 + 8]
        sp.append(torch.fft.rfft2(patch))
    q, k, v = sp
    qk = q * k
    mags = lambda z: (float(z.abs().min()), float(z.abs().max()))
    print(tag, "channel", ee, "patch", (py, px), ": |q| min/max %.2e %.2e  |k| %.2e %.2e  |v| %.2e %.2e  |q k| %.2e %.2e" % (mags(q) + mags(k) + mags(v) + mags(qk)))
    small = [(int(i // 5), int(i % 5), float("%.2e" % q.abs().flatten()[i]), float("%.2e" % k.abs().flatten()[i])) for i in torch.argsort((q.abs() * k.abs()).flatten())[:4]]
    print("    smallest |q||k| bins (ky, kx, |q|, |k|):", small, " smallest |re|, |im| parts of q: %.2e %.2e, of k: %.2e %.2e" % (float(q.real.abs().min()), float(q.imag.abs().min()), float(k.real.abs().min()), float(k.imag.abs().min())))
# ---- the tail: LayerNorm statistics of out1|out2|out3 and the projection, both variants
with torch.no_grad():
    ys, sts = [], []
    gam = torch.cat([n.body.weight.detach() for n in (att.norm1, att.norm2, att.norm3)]); bet = torch.cat([n.body.bias.detach() for n in (att.norm1, att.norm2, att.norm3)])
    for hidden, o in outs:
        st3 = ops.chan_stats(o[:, :3 * e], groups=3)
        y = ops.conv1x1(o[:, :3 * e], W(att.project_out.weight).flatten(1), ln3_gate=(st3, gam, bet, o[:, 3 * e:]), res=x)
        ys.append(y.clone()); sts.append(st3.clone())
print("LN3 statistics: rel diff of mean %.2e, of rstd %.2e; rstd range %.2e .. %.2e" % (rel(sts[0][:, :, 0], sts[1][:, :, 0]), rel(sts[0][:, :, 1], sts[1][:, :, 1]), float(sts[1][:, :, 1].min()), float(sts[1][:, :, 1].max())))
print("FDSA output (x + project_out): rel diff %.2e, max |d| %.2e, output rms %.2e, project_out part rms %.2e" % (rel(ys[0], ys[1]), float((ys[0] - ys[1]).abs().max()), float(ys[1].pow(2).mean().sqrt()), float((ys[1] - x).pow(2).mean().sqrt())))
dd = (ys[0] - ys[1]).abs()
i = torch.nonzero(dd == dd.max())[0].tolist()
print("largest output difference at (b, c, y, x) =", i, "values", float(ys[0][tuple(i)]), float(ys[1][tuple(i)]), " project_out part there %.3e" % float((ys[1] - x)[tuple(i)]))
py_, px_ = i[2], i[3]
oo = outs[1][1][0, :, py_, px_].double().cpu()
for g in range(3):
    og = oo[g * e:(g + 1) * e]
    print("  group", g, "at that pixel: mean %.3e  std %.3e  max |o| %.3e (channel %d)  vv range %.2e" % (float(og.mean()), float(og.std(unbiased=False)), float(og.abs().max()), int(og.abs().argmax()), float(oo[3 * e:].abs().max())))
od = (outs[0][1] - outs[1][1])[0, :, py_, px_].abs()
print("  largest |d o| at that pixel: %.2e in channel %d (kind %d)" % (float(od.max()), int(od.argmax()) % e, int(od.argmax()) // e))
# ---- the whole block, product route, forward statistics against fresh ones
with torch.no_grad():
    r = []
    for st in (st_fwd, st_fresh.view_as(st_fwd)):
        x1 = blk.attn.fused(x, ln=(st,) + blk.norm1.params(), res=x)
        s1 = x1._fdn_stats.clone()
        x2 = blk.ffn.fused(x1, ln=(ops.stats_of(x1),) + blk.norm2.params(), res=x1)
        x2f = blk.ffn.fused(x1, ln=(ops.chan_stats(x1),) + blk.norm2.params(), res=x1)
        r.append((x1.clone(), s1, x2.clone(), x2f.clone()))
print("block, product route: after FDSA %.2e (its statistics: mean %.2e rstd %.2e), after FDFFN %.2e; FDFFN with fresh statistics of its input %.2e" % (
    rel(r[0][0], r[1][0]), rel(r[0][1][:, :, 0], r[1][1][:, :, 0]), rel(r[0][1][:, :, 1], r[1][1][:, :, 1]), rel(r[0][2], r[1][2]), rel(r[0][3], r[1][3])))
print("per variant: FDFFN with the FDSA epilogue's statistics against FDFFN with fdn_chan_stats: %.2e, %.2e" % (rel(r[0][2], r[0][3]), rel(r[1][2], r[1][3])))
# ---- the same stages on the packed (split-bf16) GEMMs, as the product route runs them
with torch.no_grad():
    wc = ops.WeightCache()
    pk = []
    for st in (st_fwd, st_fresh.view_as(st_fwd)):
        hidden = ops.conv1x1(x, W(att.to_hidden.weight).flatten(1), ln=(st,) + blk.norm1.params(), cache=(wc, "th"))
        o = ops.fdsa_core(hidden, W(att.to_hidden_dw.weight), W(att.fft))
        st3 = ops.chan_stats(o[:, :3 * e], groups=3)
        y = ops.conv1x1(o[:, :3 * e], W(att.project_out.weight).flatten(1), ln3_gate=(st3, gam, bet, o[:, 3 * e:]), res=x, cache=(wc, "po"))
        pk.append((hidden.clone(), o.clone(), st3.clone(), y.clone()))
print("packed route: hidden %.2e  core out %.2e  LN3 mean %.2e rstd %.2e  output %.2e" % (rel(pk[0][0], pk[1][0]), rel(pk[0][1], pk[1][1]), rel(pk[0][2][:, :, 0], pk[1][2][:, :, 0]),
      rel(pk[0][2][:, :, 1], pk[1][2][:, :, 1]), rel(pk[0][3], pk[1][3])))
print("packed against fp32-MFMA route (fresh statistics): hidden %.2e  core out %.2e  output %.2e" % (rel(pk[1][0], outs[1][0]), rel(pk[1][1], outs[1][1]), rel(pk[1][3], ys[1])))
dh = (pk[0][0] - pk[1][0]).abs()
j = torch.nonzero(dh == dh.max())[0].tolist()
print("largest hidden difference %.2e at (b, c, y, x) =" % float(dh.max()), j, "values", float(pk[0][0][tuple(j)]), float(pk[1][0][tuple(j)]), " fp32 route there", float(outs[1][0][tuple(j)]))
# ---- where the core's outputs of the two packed variants part, and the float64 spectra there
print("core out: packed(forward stats) against fp32 route %.2e; packed(fresh) against fp32 route %.2e" % (rel(pk[0][1], outs[1][1]), rel(pk[1][1], outs[1][1])))
d = (pk[0][1] - pk[1][1]).abs()
dp = d.reshape(B, 4, e, H // 8, 8, Wd // 8, 8).amax((4, 6))
top = torch.topk(dp.flatten(), 5)
for v_, i_ in zip(top.values.tolist(), top.indices.tolist()):
    print("  max |d| %.2e at (b, kind, e, py, px) =" % v_, [int(t) for t in torch.unravel_index(torch.tensor(i_), dp.shape)])
b_, kind, ee, py, px = [int(t) for t in torch.unravel_index(top.indices[0], dp.shape)]
for tag, hidden in (("packed, forward statistics", pk[0][0]), ("packed, fresh statistics", pk[1][0])):
    hd = F.conv2d(hidden.double().cpu(), W(att.to_hidden_dw.weight).double().cpu(), padding=1, groups=4 * e)
    q, k, v = [torch.fft.rfft2(hd[b_, k3 * e + ee, py * 8:py * 8 + 8, px * 8:px * 8 + 8]) for k3 in range(3)]
    vf = v * W(att.fft).double().cpu()[ee, 0, 0]
    tiny = lambda z: sorted([float("%.2e" % t) for t in torch.cat([z.real.abs().flatten(), z.imag.abs().flatten()]).tolist()])[:6]
    print(tag, "channel", ee, "patch", (py, px), ": smallest |re| / |im| parts of q", tiny(q), " k", tiny(k), " v*fft", tiny(vf), " q*k", tiny(q * k))
# ---- float64 evaluation of the core's first output (|v fft| e^{i(ang q - ang k)}) at that patch for both hidden tensors, against what the kernel returned
def rd(z, thr=1e-10):
    re, im = z.real, z.imag
    re = torch.where((re < thr) & (re > -thr), torch.full_like(re, thr), re)
    im = torch.where((im < thr) & (im > -thr), torch.full_like(im, thr), im)
    return torch.complex(re, im)
for tag, (hidden, o, _, _) in (("packed, forward statistics", pk[0]), ("packed, fresh statistics", pk[1])):
    hd = F.conv2d(hidden.double().cpu(), W(att.to_hidden_dw.weight).double().cpu(), padding=1, groups=4 * e)
    pq, pkk, pv = [hd[b_, k3 * e + ee, py * 8:py * 8 + 8, px * 8:px * 8 + 8] for k3 in range(3)]
    q, k, v = torch.fft.rfft2(pq), torch.fft.rfft2(pkk), torch.fft.rfft2(pv)
    vf = rd(v * W(att.fft).double().cpu()[ee, 0, 0])
    ph = rd(q).angle() - rd(k).angle()
    o1 = torch.fft.irfft2(torch.polar(vf.abs(), ph), s=(8, 8))
    got = o[b_, ee, py * 8:py * 8 + 8, px * 8:px * 8 + 8].double().cpu()
    kb = k.abs().flatten(); order = torch.argsort(kb)[:3]
    print(tag, ": kernel against float64 of ITS hidden: max |d| %.2e (out1 patch max %.2e); k patch max |value| %.2e; smallest |k| bins" % (float((got - o1).abs().max()), float(o1.abs().max()), float(pkk.abs().max())),
          [(int(i // 5), int(i % 5), float("%.2e" % kb[i]), complex(k.flatten()[i])) for i in order])
