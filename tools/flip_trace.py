"""Where does a discrete flip of the 96 x 160 end-to-end fixture start?  Two evaluations of the HIP path - the frame and frame + 6e-8 * randn(seed) - are
compared block by block (TransformerBlock outputs, in execution order): relative RMS difference per block, and for the first block where it jumps, the
difference after each of its sub-blocks.  tools/flip_trace.py [seed, default 103]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fdn_hip
from fdn_hip import ops
from common import fixture, fdn_weights
from basicsr.models.archs import FDN_arch as A
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 103
dev = torch.device("cuda:0")
fx = fixture("fdn_tamed_96x160")
m = A.FDN().to(dev).eval(); m.load_state_dict(fdn_weights(tame=float(fx["tame"])), strict=True)
names = {mod: n for n, mod in m.named_modules()}
log = []
def hook(mod, inp, out):
    st = getattr(inp[0][0], "_fdn_stats", None)
    log.append((names[mod], out[0].detach().clone(), inp[0][0].detach().clone(), None if st is None else st.detach().clone()))
hs = [mod.register_forward_hook(hook) for mod in m.modules() if isinstance(mod, A.TransformerBlock)]
def run(x):
    log.clear()
    with torch.no_grad():
        y = m(x.to(dev), ratio_i=fx["ratio"].to(dev), device=dev)
    return [y[0].clone()] + list(log)
a = run(fx["x"])
b = run(fx["x"] + 6e-8 * torch.randn(fx["x"].shape, generator=torch.Generator().manual_seed(seed)))
rel = lambda u, v: float((u - v).double().pow(2).mean().sqrt() / v.double().pow(2).mean().sqrt())
print("y:", rel(a[0], b[0]))
first = None
for (n, oa, ia, sa), (_, ob, ib, sb) in zip(a[1:], b[1:]):
    r_in, r_out = rel(ia, ib), rel(oa, ob)
    mx = float((oa - ob).abs().max())
    flag = ""
    if first is None and r_out > 30 * max(r_in, 1e-7):
        first = n; flag = "   <-- jump"
    print(f"{n:40s} in {r_in:.2e} out {r_out:.2e} max|d| {mx:.2e}{flag}")
if first:
    blk = dict(m.named_modules())[first]
    (n, oa, ia, sa) = [t for t in a[1:] if t[0] == first][0]
    (_, ob, ib, sb) = [t for t in b[1:] if t[0] == first][0]
    for tag, xin, st in (("run a", ia, sa), ("run b", ib, sb)):
        fresh = ops.chan_stats(xin)
        if st is None:
            print("  ", tag, "no producer statistics")
        else:
            d = (st.view_as(fresh) - fresh).abs()
            print("  ", tag, "producer statistics against fdn_chan_stats of the same tensor: max |d mean| %.2e  max |d rstd| %.2e (rstd up to %.2e)" % (float(d[:, :, 0].max()), float(d[:, :, 1].max()), float(fresh[:, :, 1].max())))
    print("inside", first, "input shape", tuple(ia.shape))
    outs = []
    for xin in (ia, ib):
        with torch.no_grad():
            x1 = blk.attn.fused(xin, ln=(ops.stats_of(xin),) + blk.norm1.params(), res=xin) if blk.att else xin      # (fresh statistics: the clone carries none)
            x2 = blk.ffn.fused(x1, ln=(ops.stats_of(x1),) + blk.norm2.params(), res=x1)
        outs.append((x1, x2))
    print("  after FDSA :", rel(outs[0][0], outs[1][0]), " after FDFFN:", rel(outs[0][1], outs[1][1]))
    print("  recomputed block output against the one of the forward: run a %.2e, run b %.2e" % (rel(outs[0][1], oa), rel(outs[1][1], ob)))
    # the same with the statistics the forward used
    for tag, xin, st, o_fwd in (("run a", ia, sa, oa), ("run b", ib, sb, ob)):
        if st is None: continue
        with torch.no_grad():
            x1 = blk.attn.fused(xin, ln=(st,) + blk.norm1.params(), res=xin)
            x2 = blk.ffn.fused(x1, ln=(ops.stats_of(x1),) + blk.norm2.params(), res=x1)
        print("  ", tag, "recomputed with the forward's statistics against the forward: %.2e" % rel(x2, o_fwd))
    d = (outs[0][0] - outs[1][0]).abs()
    idx = torch.nonzero(d == d.max())[0].tolist()
    print("  largest FDSA difference", float(d.max()), "at (b, c, y, x) =", idx, " values", float(outs[0][0][tuple(idx)]), float(outs[1][0][tuple(idx)]))
