"""Debug aid: record the output of every fdn_hip.ops call at one spatial size (default the level-3 size of the 96x160 fixture) for two
routings of the FDSA blocks and print the first call whose result differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("fdn-tip2025_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from common import fixture, fdn_weights
from fdn_hip import ops
from basicsr.models.archs import FDN_arch as A
dev = lambda t: t.to("cuda:0").contiguous()
fx = fixture("fdn_tamed_96x160")
m = A.FDN(); m.load_state_dict(fdn_weights(tame=float(fx["tame"])), strict=True); m = m.to("cuda:0").eval()
HW = (24, 40)
names = ["conv1x1", "fdsa_core", "chan_stats", "fdffn_mid", "ffn_tail", "dwconv_gate", "fcaffn_in_packed", "rfft_rows_ln", "irfft_rows", "fft_cols_fcaffn"]
orig = {n: getattr(ops, n) for n in names}
logs = {}
CFGS = [(False, 32, 0), (False, 32, 1), (False, 32, 2), (True, 32, 3), (True, 32, 4), (True, 64, 5), (False, 32, 6)]
for cfg in CFGS:
    ops.FDSA_FULL, ops.FDSA_FULL_MAX_C = cfg[:2]
    log = []
    def wrap(n):
        def f(*a, **k):
            y = orig[n](*a, **k)
            t = y if torch.is_tensor(y) else None
            if t is not None and t.dim() >= 4 and tuple(t.shape[2:4]) == HW:
                st = getattr(t, "_fdn_stats", None)
                log.append((n, tuple(t.shape), t.detach().clone(), None if st is None else st.detach().clone()))
            elif t is not None and n == "chan_stats" and t.shape[-1] == HW[0] * HW[1]:
                log.append((n, tuple(t.shape), t.detach().clone(), None))
            return y
        return f
    for n in names:
        setattr(ops, n, wrap(n))
    with torch.no_grad():
        m(dev(fx["x"]), ratio_i=dev(fx["ratio"]), device=torch.device("cuda:0"))
    torch.cuda.synchronize()
    for n in names:
        setattr(ops, n, orig[n])
    logs[cfg] = log
def first_diff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0] and x[1] == y[1], (i, x[0], y[0])
        d = (x[2] - y[2]).abs().max().item()
        ds = 0.0 if x[3] is None else (x[3] - y[3]).abs().max().item()
        if d > 0 or ds > 0:
            return i, x[0], x[1], d, ds, int((x[2] != y[2]).sum())
    return None
base = logs[CFGS[1]]
for c in CFGS:
    print(c, "vs run 1 (pair, warm):", first_diff(base, logs[c]))
