"""Phase timeline of fdn_fdsa_fused from s_memtime stamps (needs the -DFDN_FUSED_TRACE build: tools/ab_build.sh trace patchfft "-DFDN_FUSED_TRACE").

    python tools/fused_trace.py abx/lib_trace.so [C] [H] [W] [B]

Every wave of 512 workgroups from the middle of the grid stamps s_memtime at eight points per chunk (0 chunk start, 1 MFMA phase issued, 2 behind
barrier 1, 3 row phase done, 4 behind barrier 2, 5 column phase done, 6 behind barrier 3, 7 inverse rows issued) and records HW_ID / XCC_ID.
Printed: the mean duration of each interval per wave index (where a wave's life goes), the share of it spent waiting at barriers, and - per SIMD,
reconstructed from HW_ID - how the two resident workgroups' phases interleave (fraction of time BOTH waves of a SIMD sit at a barrier).
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import numpy as np, torch
import fdn_hip
fdn_hip._LIB_PATH = os.path.abspath(sys.argv[1])
from fdn_hip import ops
TAIL = "--tail" in sys.argv          # (round 6) fdn_fdsa_fused_tail: the tail's stamps 40..46 (loop exit, stores drained, barrier, group 0 normalised = its data arrived,
PIN = "--pin" in sys.argv            # ... and with the following FDFFN's project_in in the same launch (stamp 47: its last store issued)
if PIN:
    sys.argv.remove("--pin")
if TAIL:                             #  group 0's MFMAs issued, group 2's MFMAs issued, epilogue stores issued); E <= 40 only (five chunks)
    sys.argv.remove("--tail")
C = int(sys.argv[2]) if len(sys.argv) > 2 else 32
H = int(sys.argv[3]) if len(sys.argv) > 3 else 736
W = int(sys.argv[4]) if len(sys.argv) > 4 else 1280
B = int(sys.argv[5]) if len(sys.argv) > 5 else 8
E = int(1.2 * C)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(B, C, H, W, generator=g).to(dev)
w = (torch.randn(4 * E, C, generator=g) / C ** .5).to(dev)
dw = (torch.randn(4 * E, 9, generator=g) / 3).to(dev)
fw = torch.randn(E, 8, 5, generator=g).to(dev)
gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
stats = ops.chan_stats(x)
wpk = ops.fdsa_pack(w, gam, bet)
lib = ctypes.CDLL(fdn_hip._LIB_PATH)          # (the debug entry points are not part of the ABI table)
NWG = 512
buf = np.zeros(NWG * 4 * 64, dtype=np.uint64)
for it in range(3):
    lib.fdn_debug_fused_trace_clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if TAIL:
        gt = torch.Generator().manual_seed(2)
        wo = (torch.randn(C, 3 * E, generator=gt) / (3 * E) ** .5).to(dev)
        Hd = int(2.7 * C)
        wi = (torch.randn(Hd, C, generator=gt) / C ** .5).to(dev)
        img = ops.fdsa_tail_pack(wo, torch.ones(3 * E, device=dev), torch.zeros(3 * E, device=dev), C,
                                 pin=ops.fold_ln(wi, None, torch.ones(C, device=dev), torch.zeros(C, device=dev)) if PIN else None)
        out = ops.fdsa_fused_tail(x, stats, wpk, dw.reshape(4 * E, 1, 3, 3), fw, img, res=x, want_stats=True, Hd=Hd if PIN else 0)
    else:
        out = ops.fdsa_fused(x, stats, wpk, dw, fw)
    e1.record()
    torch.cuda.synchronize()
print("kernel time %.3f ms" % e0.elapsed_time(e1))
assert lib.fdn_debug_fused_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes)) == 0
t = buf.reshape(NWG, 4, 64).astype(np.int64)
nch = min((E + 7) // 8, 7)          # (a wave has 56 stamp slots: the first seven chunks)
st = t[:, :, :nch * 8].reshape(NWG, 4, nch, 8)
ok = st[:, :, :, 0].min(axis=(1, 2)) > 0
print("workgroups traced:", int(ok.sum()), "chunks", nch)
st, meta = st[ok], t[ok][:, :, 60:64]
names = ["mfma phase (issue + acc -> LDS)", "barrier 1", "row phase", "barrier 2", "column phase", "barrier 3", "inverse rows + stores"]
d = np.diff(st, axis=3)                                             # [wg, wave, chunk, 7]
gap = st[:, :, 1:, 0] - st[:, :, :-1, 7]                            # end of a chunk -> start of the next (stage_store)
life = st[:, :, -1, 7] - meta[:, :, 3]
print("mean wave life (entry stamp -> last stamp): %.0f clocks; prologue (strips) %.0f" % (life.mean(), (st[:, :, 0, 0] - meta[:, :, 3]).mean()))
print("%-34s %10s %10s %10s %10s" % ("interval (clocks, mean per chunk)", "wave 0", "wave 1", "wave 2", "wave 3"))
for i, n in enumerate(names):
    print("%-34s %10.0f %10.0f %10.0f %10.0f" % ((n,) + tuple(d[:, wv, :, i].mean() for wv in range(4))))
print("%-34s %10.0f %10.0f %10.0f %10.0f" % (("chunk turnaround",) + tuple(gap[:, wv].mean() for wv in range(4))))
if TAIL:
    tl = t[ok][:, :, 40:48 if PIN else 47]
    tn = ["loop exit -> stores drained (vmcnt 0)", "barrier", "loads -> group 0 normalised", "group 0 MFMAs issued", "groups 1-2 (loads hidden?) issued", "epilogue (residual, stores, stats)"] \
        + (["project_in (LN, swaps, cuts, 72 MFMAs, 48 stores)"] if PIN else [])
    dt = np.diff(tl, axis=2)
    print("tail: last chunk stamp -> loop exit %.0f clocks" % (tl[:, :, 0] - st[:, :, -1, 7]).mean())
    for i, n in enumerate(tn):
        print("%-40s %10.0f %10.0f %10.0f %10.0f" % ((n,) + tuple(dt[:, wv, i].mean() for wv in range(4))))
    print("tail total %.0f clocks of a wave life of %.0f" % (dt.sum(axis=2).mean(), (tl[:, :, -1] - meta[:, :, 3]).mean()))
tot = d.sum(axis=3).mean(axis=(0, 2))
bar = d[:, :, :, [1, 3, 5]].sum(axis=3).mean(axis=(0, 2))
print("per chunk total", np.round(tot), " at barriers", np.round(bar), " share", np.round(bar / tot, 3))
# per-SIMD view: which waves share a SIMD (HW_ID: simd_id bits 5:4, cu_id 11:8, sh 12, se 15:13; XCC_ID low bits)
hw, xcc = meta[:, :, 0], meta[:, :, 1] & 0xF
simd_key = ((xcc << 20) | (((hw >> 13) & 7) << 16) | (((hw >> 12) & 1) << 12) | (((hw >> 8) & 0xF) << 4) | ((hw >> 4) & 3))
print("SIMD of waves 0..3 of the first traced workgroups:", [[int((h >> 4) & 3) for h in hw[i]] for i in range(6)])
# time both residents of a SIMD are inside a barrier interval simultaneously
ev = {}
for wg in range(st.shape[0]):
    for wv in range(4):
        ev.setdefault(int(simd_key[wg, wv]), []).append((wg, wv))
both = alone = 0
pairs = 0
for k, lst in ev.items():
    for i in range(len(lst)):
        for j in range(i + 1, len(lst)):
            a_, b_ = lst[i], lst[j]
            sa, sb = st[a_[0], a_[1]], st[b_[0], b_[1]]
            lo, hi = max(sa[0, 0], sb[0, 0]), min(sa[-1, 7], sb[-1, 7])
            if hi - lo < 2000:
                continue
            pairs += 1
            def bar_iv(s):
                return [(s[c, p], s[c, p + 1]) for c in range(s.shape[0]) for p in (1, 3, 5)]
            ia, ib = bar_iv(sa), bar_iv(sb)
            ov = 0
            for (a0, a1) in ia:
                for (b0, b1) in ib:
                    ov += max(0, min(a1, b1, hi) - max(a0, b0, lo))
            both += ov
            alone += hi - lo
print("SIMD pairs overlapping in time: %d; fraction of their common time with BOTH waves waiting at a barrier: %.3f" % (pairs, both / max(alone, 1)))
