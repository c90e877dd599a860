"""Interleaved A/B timing of several builds of libfdn_hip.so in ONE process: tools/ab_libs.py [--edge] [--l2] [kernel,...] lib.so [lib.so ...]
Sequential runs of two builds differ by up to 15 % on this part (clock / temperature), so the builds take turns: R rounds of
(lib A x n, lib B x n, ...), median per build.  Kernels: mid, fused, gate, tail, core, out (level-1 shapes, B = 8; --l2: level 2)."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
import fdn_hip
from fdn_hip import ops

args = sys.argv[1:]
lvl = 1
edge = False
if args and args[0] == "--edge":          # zero patches, tiny values and an exact constant in the inputs: the replace_denormals paths are exercised
    edge, args = True, args[1:]
if args and args[0] == "--l2":
    lvl, args = 2, args[1:]
kernels = args[0].split(",") if args and not args[0].endswith(".so") and args[0] != "default" else ["mid", "fused", "gate", "tail", "core", "out"]
paths = [a for a in args if a.endswith(".so") or a == "default"] or ["default"]
libs = []
for p in paths:
    l = ctypes.CDLL(fdn_hip.lib_path() if p == "default" else os.path.abspath(p))
    fdn_hip._declare(l)
    libs.append((p, l))
dev = torch.device("cuda:0")
B = 8
C = 32 * lvl
H, W = 736 // lvl, 1280 // lvl
E, Hd = int(C * 1.2), int(C * 2.7)
r = lambda *s: torch.randn(*s, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st_ = fdn_hip.stream
x = r(B, C, H, W); stats = ops.chan_stats(x); g, b_ = r(C), r(C)
h = r(B, Hd, H, W); w0, w2, fa, fp = r(Hd, 1, 3, 3), r(Hd, 1, 3, 3), r(Hd, 1, 1, 8, 5), r(Hd, 1, 1, 8, 5)
wg = r(2 * Hd, 1, 3, 3); wo = r(C, Hd) / Hd ** .5
wh = r(4 * E, C) / C ** .5; dw, fw = r(4 * E, 1, 3, 3), r(E, 1, 1, 8, 5)
hid = r(B, 4 * E, H, W); wp = r(C, 3 * E) / (3 * E) ** .5; g3, b3 = r(3 * E), r(3 * E)
if edge:
    for t in (x, h, hid):
        t[:, :, :16, :64] = 0.0
        t[:, 3::7, 16:24, :] *= 1e-12
        t[:, 1::5, 32:40, 64:128] = 0.5
        t[:, :, 48:56, :32] = -0.0
    stats = ops.chan_stats(x)
out_h, out_c, out_4e = torch.empty_like(h), torch.empty_like(x), torch.empty_like(hid)
st_out = torch.empty(B, 1, 2, H * W, device=dev)


def call(l, k):
    if k == "mid":
        return l.fdn_fdffn_mid(P(h), P(w0), P(w2), P(fa), P(fp), P(out_h), B, Hd, H, W, 0, 0, st_())
    if k == "gate":
        return l.fdn_dwconv_gate(P(h), P(wg), P(out_h), B, Hd, H, W, 0, 0, st_())
    if k == "tail":
        return l.fdn_ffn_tail(P(h), P(wg), P(wo), P(x), P(out_c), P(st_out), B, Hd, C, H, W, 0, 1, st_())
    if k == "core":
        return l.fdn_fdsa_core(P(hid), P(dw), P(fw), P(out_4e), B, E, H, W, st_())
    if k == "out":
        return l.fdn_fdsa_out(P(hid), P(wp), P(g3), P(b3), P(x), P(out_c), P(st_out), B, E, C, H * W, 0, st_())
    if k == "fused":
        return l.fdn_fdsa_fused(P(x), ctypes.c_long(C * H * W), P(stats), P(wpks[id(l)]), P(dw), P(fw), P(out_4e), B, C, E, H, W, 0, st_())
    raise SystemExit("unknown kernel " + k)


wpks = {}
for _, l in libs:                       # (the packed operand layout belongs to the build)
    nb = (E + 7) // 8 * (3 * ((C + 15) // 16) + 1) * 64 * 4
    wpk = torch.empty(max(nb, (E + 7) // 8 * (C // 2 + 1) * 64), device=dev)
    assert l.fdn_fdsa_pack(P(wh), P(g), P(b_), P(wpk), C, E, st_()) == 0
    wpks[id(l)] = wpk


def timeit(l, k, n=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        rc = call(l, k)
    e1.record(); torch.cuda.synchronize()
    assert rc == 0, (k, rc)
    return e0.elapsed_time(e1) / n


for k in kernels:
    for _, l in libs:
        timeit(l, k, 2)
    res = {p: [] for p, _ in libs}
    for rnd in range(7):
        for p, l in libs:
            res[p].append(timeit(l, k))
    print(f"L{lvl} {k:6s} " + "   ".join(f"{p}: {statistics.median(v):.3f} ms (min {min(v):.3f})" for p, v in res.items()), flush=True)
    if len(libs) > 1:                  # do the builds agree?  (same inputs: the output buffer of the kernel after each build's call)
        outs = []
        for p, l in libs:
            ob = {"mid": out_h, "gate": out_h, "tail": out_c, "out": out_c, "core": out_4e, "fused": out_4e}[k]
            ob.fill_(float("nan"))
            call(l, k)
            torch.cuda.synchronize()
            outs.append(ob.clone())
        for (p, _), o in zip(libs[1:], outs[1:]):
            same = torch.equal(outs[0], o)
            print(f"         {paths[0]} vs {p}: bit-identical {same}" + ("" if same else f", max |diff| {(outs[0] - o).abs().max().item():.3e}, nan {int(torch.isnan(o).sum())}"), flush=True)
            if not same and k in ("core", "fused"):         # which of out1 | out2 | out3 | v_value differ
                print("           per output: " + " ".join(f"{n}={torch.equal(a_, b_)}" for n, a_, b_ in zip(("out1", "out2", "out3", "vv"), outs[0].chunk(4, 1), o.chunk(4, 1))), flush=True)
