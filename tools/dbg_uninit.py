"""Debug aid: every torch.empty of fdn_hip.ops returns a NaN-filled tensor, so a kernel that reads memory nobody wrote shows up as the first
TransformerBlock whose output is not finite.  python tools/dbg_uninit.py [fixture]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("fdn-tip2025_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from common import fixture, fdn_weights
from fdn_hip import ops
from basicsr.models.archs import FDN_arch as A
real_empty = torch.empty
def nan_empty(*a, **k):
    t = real_empty(*a, **k)
    if t.is_floating_point():
        t.fill_(float("nan"))
    elif t.dtype == torch.uint8:
        t.fill_(0xFF)
    return t
class T:                       # torch proxy for the ops module only
    def __getattr__(self, n):
        return nan_empty if n == "empty" else getattr(torch, n)
ops.torch = T()
dev = lambda t: t.to("cuda:0").contiguous()
fx = fixture(sys.argv[1] if len(sys.argv) > 1 else "fdn_tamed_96x160")
m = A.FDN(); m.load_state_dict(fdn_weights(tame=float(fx["tame"])), strict=True); m = m.to("cuda:0").eval()
for cfg in ((False, 32), (True, 32), (True, 64)):
    ops.FDSA_FULL, ops.FDSA_FULL_MAX_C = cfg
    bad = []
    def mk(name):
        def hook(mod, inp, out):
            o = out[0] if isinstance(out, tuple) else out
            if not torch.isfinite(o).all():
                bad.append((name, int((~torch.isfinite(o)).sum())))
        return hook
    hs = [mod.register_forward_hook(mk(n)) for n, mod in m.named_modules() if isinstance(mod, (A.TransformerBlock, A.FDSA, A.FDFFN, A.FCAFFN))]
    with torch.no_grad():
        y = m(dev(fx["x"]), ratio_i=dev(fx["ratio"]), device=torch.device("cuda:0"))[0]
    torch.cuda.synchronize()
    for h in hs: h.remove()
    print(cfg, "non-finite outputs:", bad[:5], "final finite:", bool(torch.isfinite(y).all()))
