"""fdn_fdsa_out of two library builds on the same random input, against a float64 evaluation: tools/cmp_fdsa_out.py libA.so libB.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fdn-tip2025_amd"))
import torch, fdn_hip
dev = torch.device("cuda:0")
P_ = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
libs = {}
for p in sys.argv[1:]:
    l = ctypes.CDLL(fdn_hip.lib_path() if p == "default" else os.path.abspath(p)); fdn_hip._declare(l); libs[p] = l
torch.manual_seed(0)
for (B, E, N, H, W) in ((2, 76, 64, 48, 80), (1, 76, 64, 40, 36), (8, 76, 64, 368, 640), (1, 57, 48, 16, 12)):
    P = H * W
    o = torch.randn(B, 4 * E, P, device=dev); w = torch.randn(N, 3 * E, device=dev) / (3 * E) ** .5
    g3, b3 = torch.randn(3 * E, device=dev), torch.randn(3 * E, device=dev); res = torch.randn(B, N, P, device=dev)
    od = o.double(); v = od[:, 3 * E:]
    parts = []
    for g in range(3):
        og = od[:, g * E:(g + 1) * E]
        mu = og.mean(1, keepdim=True); var = og.var(1, unbiased=False, keepdim=True)
        parts.append(((og - mu) / torch.sqrt(var + 1e-5) * g3[g * E:(g + 1) * E].double()[None, :, None] + b3[g * E:(g + 1) * E].double()[None, :, None]) * v)
    ref = torch.einsum("nk,bkp->bnp", w.double(), torch.cat(parts, 1)) + res.double()
    for p, l in libs.items():
        out = torch.full((B, N, P), float("nan"), device=dev); st = torch.full((B, 2, P), float("nan"), device=dev)
        rc = l.fdn_fdsa_out(P_(o), P_(w), P_(g3), P_(b3), P_(res), P_(out), P_(st), B, E, N, P, 0, fdn_hip.stream())
        torch.cuda.synchronize()
        err = (out.double() - ref).abs()
        bad = (err > 1e-4).nonzero()
        print((B, E, N, H, W), p, "rc", rc, "max err %.2e rms %.2e" % (err.max().item(), err.pow(2).mean().sqrt().item()), "nan", torch.isnan(out).sum().item(),
              "bad entries", len(bad), bad[:4].tolist(), "stats mean err %.2e" % (st[:, 0].double() - ref.mean(1)).abs().max().item())
