"""(round 6) Phase timelines of fdn_ffn_tail (sliding-window form) and fdn_fdffn_mid from s_memtime sums, and their knock-out builds.

    python tools/tail_trace.py                     # needs abx/lib_ttrace.so (tools/ab_build.sh ttrace ffn_tail "-DFDN_TAILSW_TRACE"; mid: patchfft "-DFDN_MID_TRACE")

Every wave of 512 workgroups from the middle of the grid sums the clocks it spends in each phase of its inner loop.  ffn_tail_sw_kernel, per channel
pair: (0) request pair m + 2 + read the taps + first three window rows, (1) window reads + packed stencils + GELU + gate + MFMAs (issue), (2) park pair
m + 1 in LDS, (3) barrier.  fdffn_mid_kernel, per channel: A ring conv + GELU + forward rows, barrier, B park + columns, barrier, C inverse rows + second
conv + stores, barrier.  Knock-out libraries (abx/lib_kot_*.so, abx/lib_kom_*.so) are timed interleaved with the default build (tools/ab_libs.py's way).
"""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import numpy as np, torch
import fdn_hip
from fdn_hip import ops

dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
st_ = fdn_hip.stream
g = torch.Generator(device=dev).manual_seed(3)
r = lambda *s: torch.randn(*s, device=dev, generator=g)


def load(path):
    l = ctypes.CDLL(os.path.abspath(path))
    fdn_hip._declare(l)
    return l


def shapes(kind):
    B = 8
    if kind == "tail[86->32]":
        return B, 86, 32, 736, 1280
    if kind == "tail[172->64]@L2":
        return B, 172, 64, 368, 640
    if kind == "tail[172->64]@L1(fuse1)":
        return B, 172, 64, 736, 1280
    if kind == "mid[86]":
        return B, 86, 0, 736, 1280
    if kind == "mid[172]@L2":
        return B, 172, 0, 368, 640
    raise SystemExit(kind)


def make(kind):
    B, C, N, H, W = shapes(kind)
    if kind.startswith("tail"):
        h, wg, wo, x = r(B, C, H, W), r(2 * C, 1, 3, 3), r(N, C) / C ** .5, r(B, N, H, W)
        out, so = torch.empty_like(x), torch.empty(B, 1, 2, H * W, device=dev)
        return lambda l: l.fdn_ffn_tail(P(h), P(wg), P(wo), P(x), P(out), P(so), B, C, N, H, W, 0, 1, st_())
    h, w0, w2, fa, fp = r(B, C, H, W), r(C, 1, 3, 3), r(C, 1, 3, 3), r(C, 1, 1, 8, 5), r(C, 1, 1, 8, 5)
    out = torch.empty_like(h)
    return lambda l: l.fdn_fdffn_mid(P(h), P(w0), P(w2), P(fa), P(fp), P(out), B, C, H, W, 0, 0, st_())


def timeit(call, l, n=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        rc = call(l)
    e1.record(); torch.cuda.synchronize()
    assert rc == 0, rc
    return e0.elapsed_time(e1) / n


def trace_tail(lib, kind):
    call = make(kind)
    for _ in range(2):
        call(lib)
    lib.fdn_debug_tail_trace(None, 0, 1)
    ms = timeit(call, lib, 1)
    buf = np.zeros(512 * 4 * 16, dtype=np.uint64)
    assert lib.fdn_debug_tail_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes), 0) == 0
    t = buf.reshape(512, 4, 16).astype(np.float64)
    t = t[t[:, 0, 8] > 0]
    npairs = t[0, 0, 8]
    life = t[:, :, 0] + t[:, :, 6] + t[:, :, 5]
    print(f"{kind}: kernel {ms:.3f} ms (trace build), {len(t)} workgroups traced, {int(npairs)} channel pairs per tile, wave life {life.mean():.0f} clocks")
    names = ["request pair m+2, taps, first window rows", "windows + stencils + GELU + gate + MFMA issue", "park pair m+1 in LDS", "barrier"]
    loop = t[:, :, 1:5].sum(axis=2).mean()
    print(f"  prologue {t[:, :, 0].mean():.0f} ({t[:, :, 0].mean() / life.mean():.1%} of the life)   loop {loop:.0f}   epilogue {t[:, :, 5].mean():.0f} ({t[:, :, 5].mean() / life.mean():.1%})")
    for i, n in enumerate(names):
        v = t[:, :, 1 + i].mean()
        print(f"  per pair: {n:48s} {v / npairs:7.0f} clocks  {v / loop:6.1%} of the loop")


def trace_mid(lib, kind):
    call = make(kind)
    for _ in range(2):
        call(lib)
    lib.fdn_debug_mid_trace(None, 0, 1)
    ms = timeit(call, lib, 1)
    buf = np.zeros(512 * 4 * 16, dtype=np.uint64)
    assert lib.fdn_debug_mid_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes), 0) == 0
    t = buf.reshape(512, 4, 16).astype(np.float64)
    t = t[t[:, 0, 6] > 0]
    nch = t[:, :, 6].mean()
    print(f"{kind}: kernel {ms:.3f} ms (trace build), {len(t)} workgroups traced, {nch:.1f} channels per workgroup, wave life {t[:, :, 7].mean():.0f} clocks "
          f"(outside the channel loop {1 - t[:, :, :6].sum(axis=2).mean() / t[:, :, 7].mean():.1%})")
    names = ["A ring conv + GELU + forward rows", "barrier 1", "B park next halo + columns (160 threads)", "barrier 2", "C inverse rows + second conv + stores", "barrier 3"]
    loop = t[:, :, :6].sum(axis=2).mean()
    for i, n in enumerate(names):
        per_wave = " ".join(f"{t[:, wv, i].mean() / nch:7.0f}" for wv in range(4))
        print(f"  per channel: {n:42s} waves 0-3: {per_wave}   {t[:, :, i].mean() / loop:6.1%} of the loop")


def knockouts(kind, libs):
    call = make(kind)
    res = {n: [] for n, _ in libs}
    for _, l in libs:
        timeit(call, l, 2)
    for _ in range(7):
        for n, l in libs:
            res[n].append(timeit(call, l))
    base = statistics.median(res[libs[0][0]])
    print(f"{kind} knock-outs (interleaved medians): " + "   ".join(f"{n}: {statistics.median(v):.3f} ms ({statistics.median(v) / base - 1:+.1%})" for n, v in res.items()))


if __name__ == "__main__":
    abx = os.path.join(ROOT, "abx")
    have = lambda n: os.path.isfile(os.path.join(abx, n))
    if have("lib_ttrace.so"):
        lt = load(os.path.join(abx, "lib_ttrace.so"))
        for k in ("tail[86->32]", "tail[172->64]@L2", "tail[172->64]@L1(fuse1)"):
            trace_tail(lt, k)
    if have("lib_mtrace.so"):
        lm = load(os.path.join(abx, "lib_mtrace.so"))
        for k in ("mid[86]", "mid[172]@L2"):
            trace_mid(lm, k)
    base = ("default", load(fdn_hip.lib_path()))
    kot = [(n, load(os.path.join(abx, f"lib_kot_{n}.so"))) for n in ("gelu", "mfma", "loads", "stencil") if have(f"lib_kot_{n}.so")]
    if kot:
        for k in ("tail[86->32]", "tail[172->64]@L2"):
            knockouts(k, [base] + kot)
    kom = [(n, load(os.path.join(abx, f"lib_kom_{n}.so"))) for n in ("gelu", "conv2", "cols", "store", "loads") if have(f"lib_kom_{n}.so")]
    if kom:
        for k in ("mid[86]", "mid[172]@L2"):
            knockouts(k, [base] + kom)
