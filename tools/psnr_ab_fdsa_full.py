import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
for p in ("fdn-tip2025_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import fdn_oracle as O
from common import fixture, fdn_weights, lolv1_weights, GOLDEN
from fdn_hip import ops
from basicsr.models.archs import FDN_arch as A, fdnlol24_arch as L
dev = lambda t: t.to("cuda:0").contiguous()
def run(cls, w, fx):
    m = cls(); m.load_state_dict(w, strict=True); m = m.to("cuda:0").eval()
    with torch.no_grad():
        return m(dev(fx["x"]), ratio_i=dev(fx["ratio"]), device=torch.device("cuda:0"))[0].cpu()
for name, cls, wf in (("lolv1_tamed_64", L.FDN_lolv1, lolv1_weights), ("lolv1_tamed_96x160", L.FDN_lolv1, lolv1_weights),
                      ("fdn_tamed_64", A.FDN, fdn_weights), ("fdn_tamed_96x160", A.FDN, fdn_weights), ("fdn_tamed_64_dark", A.FDN, fdn_weights)):
    fx = fixture(name)
    w = wf(tame=float(fx["tame"]))
    res = {}
    for full, mc in ((True, 32), (True, 64), (False, 32)):
        ops.FDSA_FULL, ops.FDSA_FULL_MAX_C = full, mc
        y = run(cls, w, fx)
        res[(full, mc)] = y
        print(name, "full" if full else "pair", mc, "PSNR vs reference %.1f dB" % O.psnr(y, fx["y"]))
    print("   full(32) vs pair: %.1f dB" % O.psnr(res[(True, 32)], res[(False, 32)]))
