"""Micro-bench of the hot kernels at the real level-1/2/3 shapes (B=8, 736x1280) vs their own rooflines."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "8"))
H0, W0 = 736, 1280
only = sys.argv[1:] 

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def rep(name, ms, elems, flops=0):
    gb = elems * 4 / 1e9
    print(f"{name:44s} {ms:8.3f} ms  {gb/ (ms*1e-3):8.0f} GB/s(alg)  {flops/(ms*1e-3)/1e12:6.1f} TF/s", flush=True)

def want(n): return not only or any(o in n for o in only)

for lvl in (1, 2, 3):
    C = 32 * 2 ** (lvl - 1); H, W = H0 >> (lvl - 1), W0 >> (lvl - 1); P = H * W
    E, Hd = int(C * 1.2), int(C * 2.7)
    r = lambda *s: torch.randn(*s, device=dev)
    x = r(B, C, H, W)
    if want("stats"):
        rep(f"L{lvl} chan_stats C={C}", timeit(lambda: ops.chan_stats(x)), B * P * C)
    st = ops.chan_stats(x); g, b_ = r(C), r(C)
    wh = r(4 * E, C) / C ** .5
    if want("to_hidden"):
        rep(f"L{lvl} conv1x1 to_hidden LN {C}->{4*E}", timeit(lambda: ops.conv1x1(x, wh, ln=(st, g, b_))), B * P * (C + 4 * E), 2. * B * P * C * 4 * E)
    hidden = ops.conv1x1(x, wh, ln=(st, g, b_))
    dw, fw = r(4 * E, 1, 3, 3), r(E, 1, 1, 8, 5)
    if want("fdsa_fused") and C in ops.FDSA_FUSED_C:
        wpk = ops.fdsa_pack(wh, g, b_)
        rep(f"L{lvl} fdsa_fused (LN+to_hidden+core) C={C} E={E}", timeit(lambda: ops.fdsa_fused(x, st, wpk, dw, fw)), B * P * (C + 4 * E), 2. * B * P * C * 4 * E)
    if want("fdsa_core"):
        rep(f"L{lvl} fdsa_core E={E}", timeit(lambda: ops.fdsa_core(hidden, dw, fw)), B * P * 8 * E)
    o = ops.fdsa_core(hidden, dw, fw)
    if want("stats3"):
        rep(f"L{lvl} chan_stats3 E={E}", timeit(lambda: ops.chan_stats(o[:, :3 * E], groups=3)), B * P * 3 * E)
    st3 = ops.chan_stats(o[:, :3 * E], groups=3); g3, b3 = r(3 * E), r(3 * E); wo = r(C, 3 * E) / (3 * E) ** .5
    if want("attn_out"):
        rep(f"L{lvl} conv1x1 attn_out ln3gate+res {3*E}->{C}", timeit(lambda: ops.conv1x1(o[:, :3 * E], wo, ln3_gate=(st3, g3, b3, o[:, 3 * E:]), res=x)), B * P * (4 * E + 2 * C), 2. * B * P * C * 3 * E)
    del hidden, o
    wi = r(Hd, C) / C ** .5
    if want("ffn_in"):
        rep(f"L{lvl} conv1x1 ffn_in LN {C}->{Hd}", timeit(lambda: ops.conv1x1(x, wi, ln=(st, g, b_))), B * P * (C + Hd), 2. * B * P * C * Hd)
    h = ops.conv1x1(x, wi, ln=(st, g, b_))
    w0, w2, fa, fp = r(Hd, 1, 3, 3), r(Hd, 1, 3, 3), r(Hd, 1, 1, 8, 5), r(Hd, 1, 1, 8, 5)
    if want("fdffn_mid"):
        rep(f"L{lvl} fdffn_mid Hd={Hd}", timeit(lambda: ops.fdffn_mid(h, w0, w2, fa, fp)), B * P * 2 * Hd)
    wg = r(2 * Hd, 1, 3, 3)
    if want("gate"):
        rep(f"L{lvl} dwconv_gate Hd={Hd}", timeit(lambda: ops.dwconv_gate(h, wg)), B * P * 2 * Hd)
    wo2 = r(C, Hd) / Hd ** .5
    if want("ffn_out"):
        rep(f"L{lvl} conv1x1 ffn_out +res {Hd}->{C}", timeit(lambda: ops.conv1x1(h, wo2, res=x)), B * P * (Hd + 2 * C), 2. * B * P * C * Hd)
    del h
    if want("fcaffn"):
        amp, pha = torch.rand(B, 3, H, W // 2 + 1, device=dev), torch.rand(B, 3, H, W // 2 + 1, device=dev)
        wxa, wxp = r(C, 3), r(C, 3)
        rep(f"L{lvl} rfft_rows C={C}", timeit(lambda: ops.rfft_rows(x)), B * P * 2 * C)
        z = ops.rfft_rows(x)
        rep(f"L{lvl} fft_cols_fcaffn C={C}", timeit(lambda: ops.fft_cols_fcaffn(z, amp, pha, wxa, wxp)), B * P * 2 * C)
        rep(f"L{lvl} irfft_rows C={C}", timeit(lambda: ops.irfft_rows(z, H, W, 1.0)), B * P * 2 * C)
        del z
    if want("conv2d") and lvl < 3:
        w3 = r(2 * C, C, 3, 3)
        xs = ops.resample(x, ops.RS_BILINEAR_HALF)
        rep(f"L{lvl} conv2d down {C}->{2*C} @L{lvl+1}", timeit(lambda: ops.conv2d(xs, w3, pad=1)), B * P // 4 * 3 * C, 2. * B * P // 4 * C * 2 * C * 9)
        w3u = r(C, 2 * C, 3, 3); xu = r(B, 2 * C, H, W)
        rep(f"L{lvl} conv2d up {2*C}->{C} @L{lvl}", timeit(lambda: ops.conv2d(xu, w3u, pad=1)), B * P * 3 * C, 2. * B * P * C * 2 * C * 9)
        del xu, xs
    del x
    torch.cuda.empty_cache()

# fused tail vs gate + GEMM
for lvl in (1, 2, 3):
    C = 32 * 2 ** (lvl - 1); H, W = H0 >> (lvl - 1), W0 >> (lvl - 1); P = H * W
    Hd = int(C * 2.7)
    if not want("tail"): break
    for (cc, nn, tag) in ((Hd, C, "fdffn"), (C, C, "fcaffn")):
        y = torch.randn(B, cc, H, W, device=dev); wg = torch.randn(2 * cc, 1, 3, 3, device=dev); wo = torch.randn(nn, cc, device=dev) / cc ** .5
        res = torch.randn(B, nn, H, W, device=dev)
        rep(f"L{lvl} {tag} tail FUSED {cc}->{nn}", timeit(lambda: ops.ffn_tail(y, wg, wo, res=res, want_stats=True, mode="fused")), B * P * (cc + 2 * nn))
        if nn <= 64:
            rep(f"L{lvl} {tag} tail SLIDING {cc}->{nn}", timeit(lambda: ops.ffn_tail(y, wg, wo, res=res, want_stats=True, mode="sw")), B * P * (cc + 2 * nn), 2. * B * P * cc * nn)
        rep(f"L{lvl} {tag} tail gate+gemm {cc}->{nn}", timeit(lambda: ops.ffn_tail(y, wg, wo, res=res, want_stats=True, mode="split")), B * P * (cc + 2 * nn))
        del y, res
