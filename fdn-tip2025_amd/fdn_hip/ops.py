"""Tensor-level wrappers around the C ABI (one Python function per entry point of fdn_hip.h).

Inputs are fp32 ROCm tensors in NCHW layout.  A tensor argument may be a channel slice
`t[:, a:b]` of a contiguous NCHW tensor (inner three dims dense, arbitrary batch stride).
"""
import ctypes

import torch

from . import (ACT_NONE, EPI_MULADD, EPI_NONE, EPI_RES, PRO_LN, PRO_LN3_GATE, PRO_LN_MULADD, PRO_NONE,
               Conv1x1Desc, FdnHipError, check, lib, storage_dtype, stream)

BF16 = torch.bfloat16
ERR_UNSUPPORTED = 4          # FDN_ERR_UNSUPPORTED of include/fdn_hip.h: the library has no form for this shape / mode (the caller takes another route)


def _planes(t, what, bf16_ok=False):
    """(device pointer, batch stride in elements) of an NCHW tensor whose C,H,W dims are dense."""
    if not t.is_cuda:
        raise FdnHipError(f"{what} must live on a ROCm device (got {t.device}); the FDN path has no CPU fallback")
    if t.dtype != torch.float32 and not (bf16_ok and t.dtype == BF16):
        raise FdnHipError(f"{what} must be float32{' or bfloat16 storage' if bf16_ok else ''} (got {t.dtype})")
    assert t.dim() == 4, what
    _, C, H, W = t.shape
    st = t.stride()
    if not (st[3] == 1 and st[2] == W and st[1] == H * W) and t.numel() > 0:
        raise FdnHipError(f"{what}: channel/row/column dims must be dense (strides {st})")
    return ctypes.c_void_p(t.data_ptr()), (st[0] if t.shape[0] > 1 else C * H * W)


def _flat(t, what, bf16_ok=False):
    if t is None:
        return None
    if not t.is_cuda or not (t.dtype == torch.float32 or (bf16_ok and t.dtype == BF16)) or not t.is_contiguous():
        raise FdnHipError(f"{what} must be a contiguous float32 ROCm tensor")
    return ctypes.c_void_p(t.data_ptr())


def block_storage(C, P, hidden=None):
    """Storage dtype of the block-internal activations of an FDSA / FDFFN block of input width C and P pixels:
    bf16 in bf16-storage mode for the blocks whose kernels carry the bf16 load / store forms, fp32 otherwise.  The predicate
    mirrors the library's: levels 1-2 (C <= 64), pixel pairs (P % 4 == 0), and for the FDFFN hidden tensor (width `hidden`)
    the project_in form that writes bf16 (hidden >= 64 and >= 2 C; a non-stock width such as dim 16 -> 43 stays fp32)."""
    if storage_dtype() != "bf16" or C > 64 or P % 4:
        return torch.float32
    if hidden is not None and (hidden < 64 or hidden < 2 * C):
        return torch.float32
    return BF16


class WeightCache:
    """Derived weight tensors of ONE module (concatenated / folded / packed operands), rebuilt when a source
    parameter changes (data pointer or version).  Entries are stream-safe: the building stream records an event and
    any other stream that later hits the entry waits on it once, so a cold model may be driven from several HIP
    streams at once (pipeline.forward_streams).  The cache lives on its module: it dies with it."""

    def __init__(self):
        self._store = {}

    def get(self, name, srcs, build):
        key = tuple(None if p is None else (p.data_ptr(), p._version, str(p.device)) for p in srcs)
        hit = self._store.get(name)
        cuda = any(p is not None and p.is_cuda for p in srcs)
        cur = torch.cuda.current_stream() if cuda else None
        # an entry holds references to its source tensors (parameters, or detached aliases of their storage): while it lives
        # that storage cannot be freed and handed to a NEW parameter at the same address with the same version, so
        # (pointer, version, device) identifies the source
        if hit is None or hit[0] != key:
            with torch.no_grad():
                val = build()
            ev = None
            if cuda:
                ev = torch.cuda.Event()
                ev.record(cur)
                if hit is not None:                 # other streams may still be reading the operands being replaced
                    for t in (hit[1] if isinstance(hit[1], (tuple, list)) else (hit[1],)):
                        if torch.is_tensor(t) and t.is_cuda:
                            for sid in hit[3]:
                                t.record_stream(torch.cuda.ExternalStream(sid, device=t.device))
            hit = (key, val, ev, {cur.cuda_stream} if cuda else set(), tuple(srcs))
            self._store[name] = hit
        elif cuda and cur.cuda_stream not in hit[3]:
            if not torch.cuda.is_current_stream_capturing():       # (a capture is preceded by a synchronising warm-up)
                cur.wait_event(hit[2])
                hit[3].add(cur.cuda_stream)
        return hit[1]

    def versions(self):
        return tuple(h[0] for h in self._store.values())

    def __deepcopy__(self, memo):          # HIP events do not copy; a copied module rebuilds its derived weights
        return WeightCache()


def fold_ln(w, bias, gamma, beta):
    """w' = w * diag(gamma), bias' = w @ beta (+ bias): a LayerNorm's affine part in front of a 1x1 conv is linear in the
    conv, so it is folded into the GEMM operands (FDN_PRO_LN then only normalises)."""
    with torch.no_grad():
        w2 = w.reshape(w.shape[0], -1)
        wf = (w2 * gamma.reshape(1, -1)).contiguous()
        bf = torch.mv(w2.double(), beta.double()).float()
        if bias is not None:
            bf = bf + bias
    return wf, bf.contiguous()


def conv1x1_pack(w, ln3_E=0):
    """w [N, K(,1,1)] -> the split-bf16 operand image of fdn_conv1x1_pack (gemm_split.hip): three exact bf16 parts per weight."""
    N = w.shape[0]
    K = w.numel() // N
    wpk = torch.empty(lib().fdn_conv1x1_pack_bytes(N, K, ln3_E), device=w.device, dtype=torch.uint8)
    check(lib().fdn_conv1x1_pack(_flat(w.reshape(N, K), "w"), N, K, ln3_E, ctypes.c_void_p(wpk.data_ptr()), stream()), "fdn_conv1x1_pack")
    return wpk


def conv1x1(xs, w, bias=None, *, out=None, act=ACT_NONE, ln=None, ln3_gate=None, ln_muladd=None, res=None,
            muladd=None, want_stats=False, cache=None, out_dtype=torch.float32):
    """1x1 conv with fused prologue/epilogue (fdn_conv1x1).

    want_stats: also produce the channel-LayerNorm statistics of the output in the epilogue and attach
    them to the returned tensor as `._fdn_stats` (consumed by `stats_of`).

    xs: tensor or list of <=3 tensors concatenated along channels.  w: [N, K] or [N, K, 1, 1].
    ln=(stats, gamma, beta) | ln3_gate=(stats, gamma[3E], beta[3E], vv) | ln_muladd=(stats, gamma, beta, x1)
    (ln3_gate / ln_muladd: stats=None lets the kernel take the statistics itself - the K-streaming split-bf16 kernel does, in a pass over
    its pixel tile; for every other shape the fdn_chan_stats launch happens here)
    res: residual added after act | muladd=(mul, add).
    cache=(WeightCache, name): where the derived operands are kept - the LayerNorm-folded weights of `ln` (else they are rebuilt
    per call) and, for the deep shapes (K, N >= 96: level 3), the packed split-bf16 weights that put the GEMM on the bf16
    matrix pipe (without a cache those shapes run the fp32-MFMA kernels).
    out_dtype=torch.bfloat16 stores the result as bf16 (FDFFN project_in); a bf16 `xs` is read as bf16 storage
    (FDFFN project_out).  The library refuses forms it has no bf16 kernel for.
    """
    if torch.is_tensor(xs):
        xs = [xs]
    B, _, H, W = xs[0].shape
    P = H * W
    N = w.shape[0]
    K = sum(x.shape[1] for x in xs)
    assert w.numel() == N * K, (w.shape, K)
    if out is None:
        out = torch.empty((B, N, H, W), device=xs[0].device, dtype=out_dtype)
    d = Conv1x1Desc()
    for i, x in enumerate(xs):
        d.x[i], d.xbs[i] = _planes(x, f"x[{i}]", bf16_ok=(i == 0 and len(xs) == 1))
        d.kseg[i] = x.shape[1]
    d.x_bf16 = int(xs[0].dtype == BF16)
    d.out_bf16 = int(out.dtype == BF16)
    d.w = _flat(w, "w")
    d.bias = _flat(bias, "bias")
    d.out, d.obs = _planes(out, "out", bf16_ok=True)
    d.B, d.K, d.N, d.P = B, K, N, P
    d.pro, d.ln_group = PRO_NONE, K
    w0, bias0 = w, bias
    if ln is not None:
        d.pro = PRO_LN                        # the kernel normalises only; the affine part rides in the weights
        if cache is not None:
            w, bias = cache[0].get(cache[1], [w, bias, ln[1], ln[2]], lambda w=w, bias=bias: fold_ln(w, bias, ln[1], ln[2]))
        else:
            w, bias = fold_ln(w, bias, ln[1], ln[2])
        d.w, d.bias = _flat(w, "w"), _flat(bias, "bias")
        d.stats = _flat(ln[0], "stats")
    elif ln3_gate is not None:
        d.pro, d.ln_group = PRO_LN3_GATE, K // 3
        d.stats, d.gamma, d.beta = _flat(ln3_gate[0], "stats"), _flat(ln3_gate[1], "gamma"), _flat(ln3_gate[2], "beta")
        d.xb, d.xbbs = _planes(ln3_gate[3], "v_value")
    elif ln_muladd is not None:
        d.pro = PRO_LN_MULADD
        d.stats, d.gamma, d.beta = _flat(ln_muladd[0], "stats"), _flat(ln_muladd[1], "gamma"), _flat(ln_muladd[2], "beta")
        d.xb, d.xbbs = _planes(ln_muladd[3], "x1")
    if (cache is not None and ((K >= 96 and N >= 96) or (16 < K <= 64 and 2 * N >= 5 * K and res is None and muladd is None and not want_stats)) and (len(xs) == 1 or (len(xs) == 2 and K >= 96 and N >= 96 and xs[0].shape[1] % 32 == 0 and d.pro == PRO_NONE)) and act == ACT_NONE and xs[0].dtype == torch.float32
            and out.dtype == torch.float32):
        srcs = [w0, bias0] + (list(ln[1:3]) if ln is not None else [])
        d.wpk = ctypes.c_void_p(cache[0].get(cache[1] + ":pk", srcs, lambda w=w: conv1x1_pack(
            w, K // 3 if ln3_gate is not None else 0)).data_ptr())
    d.act = act
    d.epi = EPI_NONE
    if res is not None:
        d.epi = EPI_RES
        d.res, d.rbs = _planes(res, "res")
    elif muladd is not None:
        d.epi = EPI_MULADD
        d.mul, d.mbs = _planes(muladd[0], "mul")
        d.add, mbs2 = _planes(muladd[1], "add")
        assert mbs2 == d.mbs
    if d.pro in (PRO_LN3_GATE, PRO_LN_MULADD) and not d.stats and not d.wpk:
        # only the K-streaming split-bf16 kernel (packed operands: the predicate above, bf16 matrix pipe) takes its tile's LayerNorm statistics itself;
        # every other shape / mode gets the fdn_chan_stats launch here instead of a refused call first
        auto = chan_stats(xs[0], groups=3) if ln3_gate is not None else chan_stats(xs[0])
        d.stats = _flat(auto, "stats")
    stats = None
    if want_stats and N <= 160:
        stats = torch.empty((B, 1, 2, P), device=out.device, dtype=torch.float32)
        d.stats_out = _flat(stats, "stats_out")
    rc = lib().fdn_conv1x1(ctypes.byref(d), stream())
    if rc == ERR_UNSUPPORTED and d.pro in (PRO_LN3_GATE, PRO_LN_MULADD) and not d.stats:      # safety net: no kernel of this shape takes the statistics itself
        x0 = xs[0]
        auto = chan_stats(x0, groups=3) if ln3_gate is not None else chan_stats(x0)
        d.stats = _flat(auto, "stats")
        rc = lib().fdn_conv1x1(ctypes.byref(d), stream())
    check(rc, "fdn_conv1x1")
    if want_stats:
        out._fdn_stats = stats if stats is not None else chan_stats(out)   # LayerNorm statistics travel with the tensor
    return out


GEMM_OWN_STATS = True       # the level-3 LN3 / FCAFFN GEMMs take their LayerNorm statistics in-kernel (False: an fdn_chan_stats launch in front; A/B runs)


def stats_of(x):
    """LayerNorm statistics of x: reuse the ones its producer attached, else compute them."""
    st = getattr(x, "_fdn_stats", None)
    return st if st is not None else chan_stats(x)


def chan_stats(x, groups=1):
    """(mean, rstd) over each of `groups` equal channel groups -> [B, G, 2, H*W] (fdn_chan_stats)."""
    B, C, H, W = x.shape
    E = C // groups
    ptr, xbs = _planes(x, "x")
    stats = torch.empty((B, groups, 2, H * W), device=x.device, dtype=torch.float32)
    check(lib().fdn_chan_stats(ptr, ctypes.c_long(xbs), _flat(stats, "stats"), B, groups, E, H * W, stream()),
          "fdn_chan_stats")
    return stats


def layernorm_chan(x, gamma, beta):
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    check(lib().fdn_layernorm_chan(_flat(x, "x"), _flat(gamma, "gamma"), _flat(beta, "beta"), _flat(out, "out"),
                                   B, C, H * W, stream()), "fdn_layernorm_chan")
    return out


def fdsa_core(hidden, dw_w, fft_w):
    B, C4, H, W = hidden.shape
    out = torch.empty_like(hidden)
    check(lib().fdn_fdsa_core(_flat(hidden, "hidden"), _flat(dw_w, "dw_w"), _flat(fft_w, "fft_w"), _flat(out, "out"),
                              B, C4 // 4, H, W, stream()), "fdn_fdsa_core")
    return out


FDSA_FUSED_C = (24, 32, 48, 64)          # input widths fdn_fdsa_fused is instantiated for


def fdsa_pack(w, gamma, beta):
    """to_hidden weight [4E, C(,1,1)] (+ the LayerNorm in front) -> MFMA operands of fdn_fdsa_fused (fdn_fdsa_pack)."""
    E4, C = w.shape[0], w.shape[1]
    E = E4 // 4
    nch = (E + 7) // 8
    wpk = torch.empty((nch, 3 * ((C + 15) // 16) + 1, 64, 4), device=w.device, dtype=torch.float32)      # 16-byte bf16 operand fragments
    check(lib().fdn_fdsa_pack(_flat(w.reshape(E4, C), "w"), _flat(gamma, "gamma"), _flat(beta, "beta"), _flat(wpk, "wpk"),
                              C, E, stream()), "fdn_fdsa_pack")
    return wpk


def fdsa_fused(x, stats, wpk, dw_w, fft_w, out_dtype=torch.float32):
    """LayerNorm + to_hidden + fdsa_core in one launch (fdn_fdsa_fused): x [B,C,H,W] -> (out1|out2|out3|v_value) [B,4E,H,W]."""
    B, C, H, W = x.shape
    E = fft_w.shape[0]
    assert C in FDSA_FUSED_C, C
    out = torch.empty((B, 4 * E, H, W), device=x.device, dtype=out_dtype)
    ptr, xbs = _planes(x, "x")
    check(lib().fdn_fdsa_fused(ptr, ctypes.c_long(xbs), _flat(stats, "stats"), _flat(wpk, "wpk"), _flat(dw_w, "dw_w"),
                               _flat(fft_w, "fft_w"), _flat(out, "out", True), B, C, E, H, W, int(out_dtype == BF16), stream()),
          "fdn_fdsa_fused")
    return out


FDSA_TAIL = True            # (round 6) levels 1-2: fdn_fdsa_fused_tail - the producing workgroup runs fdn_fdsa_out's arithmetic on its own tile (one launch, bit-identical)
_fdsa_scratch = {}          # (device, floats) -> scratch tensor of fdn_fdsa_fused_tail: one per device and size, shared by every block (one stream per GPU)


def fdsa_tail_pack(w_out, gamma3, beta3, C, pin=None):
    """project_out [N, 3E(,1,1)] + norm1..3 -> the LDS operand image of fdn_fdsa_fused_tail's in-kernel tail; None = no form for this width.
    pin = (wf [Hd, C], bf [Hd]): the LayerNorm-folded project_in of the FDFFN behind the FDSA (fold_ln), run by the same tail (level 1)."""
    N = w_out.shape[0]
    E = w_out.numel() // N // 3
    Hd = pin[0].shape[0] if pin is not None else 0
    n = lib().fdn_fdsa_tail_pack_floats(C, E, N, Hd)
    if n <= 0:
        return None
    img = torch.empty(n, device=w_out.device, dtype=torch.float32)
    check(lib().fdn_fdsa_tail_pack(_flat(w_out.reshape(N, 3 * E), "w_out"), _flat(gamma3, "gamma3"), _flat(beta3, "beta3"),
                                   _flat(pin[0].reshape(Hd, C), "pin_w") if pin is not None else None, _flat(pin[1], "pin_b") if pin is not None else None,
                                   _flat(img, "img"), C, E, N, Hd, stream()), "fdn_fdsa_tail_pack")
    return img


def fdsa_fused_tail(x, stats, wpk, dw_w, fft_w, tail_img, res=None, want_stats=False, Hd=0, h_dtype=torch.float32):
    """x [B,C,H,W] -> res + project_out(norm1..3(FDSA core(LN(x))) * v_value) in ONE launch (fdn_fdsa_fused_tail), bit-identical to
    fdsa_fused + fdsa_out.  Hd > 0 (tail_img packed with `pin`): also h = project_in(LN(out)) [B,Hd,H,W] of the FDFFN that follows, attached to
    the result as `._fdn_pin`.  Returns None when the library has no form for the shape."""
    B, C, H, W = x.shape
    E = dw_w.shape[0] // 4
    ptr, xbs = _planes(x, "x")
    n = lib().fdn_fdsa_scratch_floats(B, E, H, W)
    key = (str(x.device), n)
    scr = _fdsa_scratch.get(key)
    if scr is None:          # (a ring of per-resident-workgroup blocks that starts with the slot flags: zero once, every launch leaves them zero)
        scr = _fdsa_scratch[key] = torch.zeros(n, device=x.device, dtype=torch.float32)
    out = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
    st = torch.empty((B, 1, 2, H * W), device=x.device, dtype=torch.float32) if want_stats else None
    h = torch.empty((B, Hd, H, W), device=x.device, dtype=h_dtype) if Hd else None
    rc = lib().fdn_fdsa_fused_tail(ptr, ctypes.c_long(xbs), _flat(stats, "stats"), _flat(wpk, "wpk"), _flat(dw_w, "dw_w"), _flat(fft_w, "fft_w"),
                                   _flat(tail_img, "tail_img"), _flat(res, "res"), _flat(out, "out"), _flat(st, "stats_out"), _flat(scr, "scratch"),
                                   _flat(h, "h_out", True), B, C, E, H, W, Hd, int(h_dtype == BF16), stream())
    if rc == ERR_UNSUPPORTED:
        return None
    check(rc, "fdn_fdsa_fused_tail")
    if want_stats:
        out._fdn_stats = st
    if Hd:
        out._fdn_pin = h
    return out


FDSA_TAIL_PIN_MAX_C = 32    # widths up to which the following FDFFN's project_in rides in the FDSA launch.  64 (level 2 too: the kernel exists, bit-identical) measured
                            # 278.75 against 277.66 ms per step (three alternating runs, profiles/r06_h_bench_*.json): the 72 KB of operands come from L2 per tile
FDSA_TAIL_PIN = True        # (round 6) level 1: that launch also runs the following FDFFN's project_in (bit-identical to fdn_conv1x1's kernel for the shape)
FDSA_FULL = False           # True: the whole FDSA sub-block in one launch (fdn_fdsa_full) for C <= FDSA_FULL_MAX_C; False: fdn_fdsa_fused + fdn_fdsa_out
FDSA_FULL_MAX_C = 32        # measured (tools/ab_fdsa_full.py, B = 8 720p shapes): one launch 3.52 against 3.77 ms at C = 32 and 2.89 against 2.93 at
                            # C = 24 (8 x 16 tiles, 8-channel chunks); at C = 48 / 64 (8 x 8 tiles, 16-channel chunks) it loses, 2.63 against 2.25 ms


def fdsa_full_pack(w_hidden, gamma, beta, w_out, gamma3, beta3):
    """Operand image of fdn_fdsa_full: to_hidden (+ the LayerNorm in front) and project_out (+ norm1..3) as split-bf16 MFMA operands.
    None when the library has no form for this width."""
    E4, C = w_hidden.shape[0], w_hidden.shape[1]
    E = E4 // 4
    nbytes = lib().fdn_fdsa_full_pack_bytes(C, E)
    if nbytes <= 0:
        return None
    wpk = torch.empty(nbytes, device=w_hidden.device, dtype=torch.uint8)
    check(lib().fdn_fdsa_full_pack(_flat(w_hidden.reshape(E4, C), "w_hidden"), _flat(gamma, "gamma"), _flat(beta, "beta"),
                                   _flat(w_out.reshape(w_out.shape[0], 3 * E), "w_out"), _flat(gamma3, "gamma3"), _flat(beta3, "beta3"),
                                   ctypes.c_void_p(wpk.data_ptr()), C, E, stream()), "fdn_fdsa_full_pack")
    return wpk


def fdsa_full(x, stats, wpk, dw_w, fft_w, res=None, want_stats=False):
    """x [B,C,H,W] -> res + project_out(norm1..3(FDSA core(LN(x))) * v_value) in ONE launch (fdn_fdsa_full).  Returns None when the
    library has no form for the shape (the caller takes fdsa_fused + fdsa_out)."""
    B, C, H, W = x.shape
    E = fft_w.shape[0]
    out = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
    st = torch.empty((B, 1, 2, H * W), device=x.device, dtype=torch.float32) if want_stats else None
    ptr, xbs = _planes(x, "x")
    rc = lib().fdn_fdsa_full(ptr, ctypes.c_long(xbs), _flat(stats, "stats"), ctypes.c_void_p(wpk.data_ptr()), _flat(dw_w, "dw_w"),
                             _flat(fft_w, "fft_w"), _flat(res, "res"), _flat(out, "out"), _flat(st, "stats_out"), B, C, E, H, W, stream())
    if rc == ERR_UNSUPPORTED:
        return None
    check(rc, "fdn_fdsa_full")
    if want_stats:
        out._fdn_stats = st
    return out


def fdsa_out(o, w, gamma3, beta3, res=None, want_stats=False):
    """Fused FDSA tail (fdn_fdsa_out).  Returns None when the size is not covered (E > 76 or N > 64)."""
    B, C4, H, W = o.shape
    E, N, P = C4 // 4, w.shape[0], H * W
    if E > 76 or N > 64:          # register-resident form: levels 1 and 2 (level 3 needs 612 values per pixel)
        return None
    out = torch.empty((B, N, H, W), device=o.device, dtype=torch.float32)
    stats = torch.empty((B, 1, 2, P), device=o.device, dtype=torch.float32) if want_stats else None
    rc = lib().fdn_fdsa_out(_flat(o, "o", True), _flat(w, "w"), _flat(gamma3, "gamma3"), _flat(beta3, "beta3"), _flat(res, "res"),
                            _flat(out, "out"), _flat(stats, "stats_out"), B, E, N, P, int(o.dtype == BF16), stream())
    if rc == ERR_UNSUPPORTED and o.dtype != BF16:          # the caller takes the statistics + GEMM route (fp32 only)
        return None
    check(rc, "fdn_fdsa_out")
    if want_stats:
        out._fdn_stats = stats
    return out


def fdffn_mid(x, w0, w2, ffta, fftp, out_dtype=None):
    """x may be bf16 storage; out_dtype defaults to x's."""
    B, Hd, H, W = x.shape
    out = torch.empty(x.shape, device=x.device, dtype=out_dtype or x.dtype)
    check(lib().fdn_fdffn_mid(_flat(x, "x", True), _flat(w0, "w0"), _flat(w2, "w2"), _flat(ffta, "ffta"), _flat(fftp, "fftp"),
                              _flat(out, "out", True), B, Hd, H, W, int(x.dtype == BF16), int(out.dtype == BF16), stream()),
          "fdn_fdffn_mid")
    return out


FFN_TAIL_MODE = None          # None = per-shape choice below; "sw" | "fused" | "split" forces one (A/B runs, tests)


def ffn_tail(y, dw_w, w, res=None, want_stats=False, mode=None, cache=None):
    """gate + project_out + residual (+ next LayerNorm statistics).
      "sw"    one launch, sliding-window kernel (fdn_ffn_tail form 1): N <= 64, W % 4 == 0, fp32 or bf16-storage y;
      "fused" one launch, the chunked kernel of round 1 (fdn_ffn_tail form 0), fp32 y;
      "split" fdn_dwconv_gate then the project_out GEMM (the gated tensor makes a round trip through HBM)."""
    B, C, H, W = y.shape
    N = w.shape[0]
    mode = mode or FFN_TAIL_MODE
    if mode is None:
        # measured on MI355X (tools/bench_kernels.py tail, B = 8 720p): one launch wins for the 32-wide outputs (level 1: FDFFN 86 -> 32
        # 1.60 vs 1.94 ms, FCAFFN 32 -> 32 0.75 vs 0.97 ms) and for the 64 -> 64 FCAFFN tail of level 2 (0.56 vs 0.76 ms); the deep
        # 172 -> 64 FDFFN tail of level 2 lost in round 3 (1.24 vs 1.06 ms) and wins since the packed (A, B) stencil of round 4 (1.01-1.08
        # vs 1.05-1.11 ms; the Fuse shape 172 -> 64 at 736 x 1280: 3.70 vs 4.18 ms, tools/tail_modes.py); level 3 (N = 128) stays on gate + GEMM
        mode = "sw" if (W % 4 == 0 and N <= 64) else "split"
    if mode == "split":
        g = dwconv_gate(y, dw_w)                 # (bf16 storage in -> bf16 storage out -> the project_out conv reads bf16)
        return conv1x1(g, w, res=res, want_stats=want_stats, cache=cache)
    out = torch.empty((B, N, H, W), device=y.device, dtype=torch.float32)
    stats = torch.empty((B, 1, 2, H * W), device=y.device, dtype=torch.float32) if want_stats else None
    check(lib().fdn_ffn_tail(_flat(y, "y", True), _flat(dw_w, "dw_w"), _flat(w, "w"), _flat(res, "res"), _flat(out, "out"),
                             _flat(stats, "stats_out"), B, C, N, H, W, int(y.dtype == BF16), 1 if mode == "sw" else 0, stream()),
          "fdn_ffn_tail")
    if want_stats:
        out._fdn_stats = stats
    return out


def dwconv_gate(x, w, out_dtype=None):
    """x may be bf16 storage; out_dtype defaults to x's."""
    B, C, H, W = x.shape
    out = torch.empty(x.shape, device=x.device, dtype=out_dtype or x.dtype)
    check(lib().fdn_dwconv_gate(_flat(x, "x", True), _flat(w, "w"), _flat(out, "out", True), B, C, H, W, int(x.dtype == BF16),
                                int(out.dtype == BF16), stream()), "fdn_dwconv_gate")
    return out


def dwconv3x3(x, w, act=ACT_NONE):
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    check(lib().fdn_dwconv3x3(_flat(x, "x"), _flat(w, "w"), _flat(out, "out"), B, C, H, W, act, stream()), "fdn_dwconv3x3")
    return out


def img_mod_maps(img, w1_mul, w3_mul, w1_add, w3_add):
    B, _, H, W = img.shape
    C = w1_mul.shape[0]
    mul = torch.empty((B, C, H, W), device=img.device, dtype=torch.float32)
    add = torch.empty_like(mul)
    check(lib().fdn_img_mod_maps(_flat(img, "img"), _flat(w1_mul, "w1_mul"), _flat(w3_mul, "w3_mul"),
                                 _flat(w1_add, "w1_add"), _flat(w3_add, "w3_add"), _flat(mul, "mul"), _flat(add, "add"),
                                 B, C, H, W, stream()), "fdn_img_mod_maps")
    return mul, add


FCAFFN_IN_C = (32, 64)     # widths fdn_fcaffn_in has a form for


def fcaffn_in(xi, x1, img, w, gamma, beta, w1_mul, w3_mul, w1_add, w3_add, x1_ln=None):
    """project_in(norm(xi) * x1 + x1) * conv3_mul(conv1_mul(img)) + conv3_add(conv1_add(img)) in one launch (FDN_arch.py:419-423);
    C in FCAFFN_IN_C, W even.  x1_ln = (stats, gamma, beta): x1 is given un-normalised and its LayerNorm is applied on load."""
    B, C, H, W = xi.shape
    out = torch.empty_like(xi)
    st1, g1, b1 = x1_ln if x1_ln is not None else (None, None, None)
    check(lib().fdn_fcaffn_in(_flat(xi, "xi"), _flat(x1, "x1"), _flat(st1, "stats1"), _flat(g1, "gamma1"), _flat(b1, "beta1"),
                              _flat(img, "img"), _flat(w, "w"), _flat(gamma, "gamma"),
                              _flat(beta, "beta"), _flat(w1_mul, "w1_mul"), _flat(w3_mul, "w3_mul"), _flat(w1_add, "w1_add"),
                              _flat(w3_add, "w3_add"), _flat(out, "out"), B, C, H, W, stream()), "fdn_fcaffn_in")
    return out


FCAFFN_PACKED_MIN_C = 96          # widths from which the split-bf16 GEMM form (fdn_fcaffn_in_packed) takes the sub-block


def fcaffn_in_pack(w, w1_mul, w3_mul, w1_add, w3_add):
    """project_in [C, C] and the two folded image maps as split-bf16 MFMA operands (fdn_fcaffn_in_pack), once per weight set."""
    C = w.shape[0]
    wpk = torch.empty(lib().fdn_fcaffn_in_pack_bytes(C), device=w.device, dtype=torch.uint8)
    check(lib().fdn_fcaffn_in_pack(_flat(w.reshape(C, C), "w"), _flat(w1_mul, "w1_mul"), _flat(w3_mul, "w3_mul"), _flat(w1_add, "w1_add"),
                                   _flat(w3_add, "w3_add"), C, ctypes.c_void_p(wpk.data_ptr()), stream()), "fdn_fcaffn_in_pack")
    return wpk


def fcaffn_in_packed(xi, stats_xi, x1, img, wpk, gamma, beta, x1_ln=None):
    """fcaffn_in for C >= FCAFFN_PACKED_MIN_C (level 3): one launch on the split-bf16 GEMM; stats_xi = chan_stats(xi), or None: the kernel
    takes the statistics of xi itself."""
    B, C, H, W = xi.shape
    out = torch.empty_like(xi)
    st1, g1, b1 = x1_ln if x1_ln is not None else (None, None, None)
    check(lib().fdn_fcaffn_in_packed(_flat(xi, "xi"), _flat(stats_xi, "stats_xi"), _flat(x1, "x1"), _flat(st1, "stats1"), _flat(g1, "gamma1"),
                                     _flat(b1, "beta1"), _flat(img, "img"), ctypes.c_void_p(wpk.data_ptr()), _flat(gamma, "gamma"),
                                     _flat(beta, "beta"), _flat(out, "out"), B, C, H, W, stream()), "fdn_fcaffn_in_packed")
    return out


# ---------------------------------------------------------------------------------------------
# full-image FFT pipeline
# ---------------------------------------------------------------------------------------------
RS_BILINEAR_HALF, RS_BILINEAR_X2, RS_NEAREST_HALF, RS_NEAREST_X2, RS_PIXEL_UNSHUFFLE = 0, 1, 2, 3, 4


def spec_pitch(Wf):
    """Row pitch (bins) that starts every spectrum row on a 128-byte line: the column pass then runs on the padded width."""
    return (Wf + 15) // 16 * 16


def rfft_rows(x, pitch=None):
    """real [..., H, W] -> interleaved complex [..., H, pitch or W//2+1, 2] along the last axis (bins past W//2 are zeros)."""
    W = x.shape[-1]
    rows = x.numel() // W
    out = torch.empty(x.shape[:-1] + (pitch or W // 2 + 1, 2), device=x.device, dtype=torch.float32)
    check(lib().fdn_rfft_rows(_flat(x, "x"), _flat(out, "out"), ctypes.c_long(rows), W, ctypes.c_long(pitch or 0), stream()), "fdn_rfft_rows")
    return out


ROWS_PLANNED_W = tuple(2 * r * p for r in (20, 30) for p in (32, 16, 8)) + (608, 304) + (1120, 560, 280)     # widths fdn_rfft_rows_ln has a form for (19 x 16, 19 x 8: LOL-v1 padded; 35 x 16 / 8 / 4: LOL-Blur frames)


def rows_ln_ok(x):
    """fdn_rfft_rows_ln has a form for this tensor: a planned width, and the statistics of the whole batch behind one 2 GB descriptor
    (fft2d.hip; a larger batch takes fdn_layernorm_chan + fdn_rfft_rows, as include/fdn_hip.h says)"""
    B, _, H, W = x.shape
    return W in ROWS_PLANNED_W and B * 2 * H * W * 4 <= 0x7FFFFFFF


def rfft_rows_ln(x, stats, gamma, beta, pitch=None):
    """rfft along rows of the channel LayerNorm of x [B, C, H, W], normalised on load (fdn_rfft_rows_ln); W in ROWS_PLANNED_W."""
    B, C, H, W = x.shape
    out = torch.empty((B, C, H, pitch or W // 2 + 1, 2), device=x.device, dtype=torch.float32)
    check(lib().fdn_rfft_rows_ln(_flat(x, "x"), _flat(stats, "stats"), _flat(gamma, "gamma"), _flat(beta, "beta"), _flat(out, "out"),
                                 B, C, H, W, ctypes.c_long(pitch or 0), stream()), "fdn_rfft_rows_ln")
    return out


def irfft_rows(z, H, W, scale, res=None, alpha=0.0, out=None):
    """complex planes [B, C, Hin>=H, Wfin>=W//2+1, 2] -> real [B, C, H, W] = scale*c2r + alpha*res."""
    B, C, Hin, Wfin, _ = z.shape
    if out is None:
        out = torch.empty((B, C, H, W), device=z.device, dtype=torch.float32)
    check(lib().fdn_irfft_rows(_flat(z, "z"), ctypes.c_long(Wfin), ctypes.c_long(Hin * Wfin), _flat(out, "out"),
                               ctypes.c_long(B * C), H, W, ctypes.c_float(scale), _flat(res, "res"),
                               ctypes.c_float(alpha), stream()), "fdn_irfft_rows")
    return out


def sincos(x):
    """(sin x, cos x) as the polar <-> complex steps of the column kernels evaluate them (fdn_sincos_f32; test hook)."""
    sn, cs = torch.empty_like(x), torch.empty_like(x)
    check(lib().fdn_sincos_f32(_flat(x, "x"), _flat(sn, "sn"), _flat(cs, "cs"), ctypes.c_long(x.numel()), stream()), "fdn_sincos_f32")
    return sn, cs


def pack_guidance(amp, pha, pitch=None):
    """(amp, pha) [B,3,H,Wf] -> one 32-byte record per bin [B,H,pitch or Wf,8] (fdn_pack_guidance).  The guidance of a
    level is shared by all its encoder blocks, so the packed copy is memoised on the amp tensor object."""
    B, _, H, Wf = amp.shape
    pitch = pitch or Wf
    cached = getattr(amp, "_fdn_packed", None)
    if cached is not None and cached[0] is pha and cached[1].shape[2] == pitch:
        return cached[1]
    out = torch.empty((B, H, pitch, 8), device=amp.device, dtype=torch.float32)
    check(lib().fdn_pack_guidance(_flat(amp, "amp"), _flat(pha, "pha"), _flat(out, "packed"), B, H, Wf, ctypes.c_long(pitch), stream()),
          "fdn_pack_guidance")
    amp._fdn_packed = (pha, out)
    return out


def fft_cols_fcaffn(z, amp, pha, wxa, wxp):
    """z [B, C, H, Wz, 2] with Wz >= the guidance width (a padded spectrum: the guidance is packed with the same pitch)."""
    B, C, H, Wf, _ = z.shape
    guide = pack_guidance(amp, pha, pitch=Wf)
    check(lib().fdn_fft_cols_fcaffn(_flat(z, "z"), _flat(guide, "guide"), _flat(wxa, "wxa"), _flat(wxp, "wxp"), B, C, H, Wf,
                                    stream()), "fdn_fft_cols_fcaffn")
    return z


def fft_cols_fwd(z, want_abs, want_ang, rd_before=False, fix_real=True):
    B, C, H, Wf, _ = z.shape
    oa = torch.empty((B, C, H, Wf), device=z.device, dtype=torch.float32) if want_abs else None
    og = torch.empty((B, C, H, Wf), device=z.device, dtype=torch.float32) if want_ang else None
    check(lib().fdn_fft_cols_fwd(_flat(z, "z"), _flat(oa, "abs"), _flat(og, "ang"), ctypes.c_long(B * C), H, Wf,
                                 int(rd_before), int(fix_real), stream()), "fdn_fft_cols_fwd")
    return oa, og


def fft_cols_inv_polar(mag, pha, H, Wf):
    B, C, Hin, Wfin = mag.shape
    z = torch.empty((B, C, H, Wf, 2), device=mag.device, dtype=torch.float32)
    check(lib().fdn_fft_cols_inv_polar(_flat(mag, "mag"), _flat(pha, "pha"), Hin, Wfin, _flat(z, "z"),
                                       ctypes.c_long(B * C), H, Wf, stream()), "fdn_fft_cols_inv_polar")
    return z


# ---------------------------------------------------------------------------------------------
# dense convs, resampling, small helpers
# ---------------------------------------------------------------------------------------------
def conv2d(x, w, bias=None, stride=1, pad=0, act=ACT_NONE, res=None, res_before_act=False, post_add=0.0):
    B, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    out = torch.empty((B, Cout, OH, OW), device=x.device, dtype=torch.float32)
    check(lib().fdn_conv2d(_flat(x, "x"), _flat(w, "w"), _flat(bias, "bias"), _flat(res, "res"), _flat(out, "out"),
                           B, Cin, H, W, Cout, KH, KW, stride, pad, act, int(res_before_act), ctypes.c_float(post_add),
                           stream()), "fdn_conv2d")
    return out


def conv_transpose4x4s2(x, w, bias, act):
    B, Cin, H, W = x.shape
    Cout = w.shape[1]
    out = torch.empty((B, Cout, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    check(lib().fdn_conv_transpose4x4s2(_flat(x, "x"), _flat(w, "w"), _flat(bias, "bias"), _flat(out, "out"),
                                        B, Cin, H, W, Cout, act, stream()), "fdn_conv_transpose4x4s2")
    return out


AFF_MULTIRES = True         # MAR's fourier_fuse 1x1 convs per source resolution (False: nearest-resized copies + one 84-channel conv; A/B runs)
UPCONV_GATHER = True        # Upsample as a low-resolution 1x1 conv per tap + fdn_upconv_gather (False: fdn_resample x2 + the 3x3 conv; A/B runs)


def upsample_conv3x3(x, w, cache=None):
    """Conv2d(C, Cout, 3, padding=1, bias=False) of the bilinear x2 image of x (FDN_arch.py:726-734) without that image: the nine per-tap
    1x1 products at low resolution (one fdn_conv1x1, weight rearranged to [9 Cout, C]: a quarter of the conv's matrix work), then
    fdn_upconv_gather sums the taps' bilinear samples per output pixel."""
    B, C, h, w_ = x.shape
    Cout = w.shape[0]
    build = lambda: w.detach().permute(2, 3, 0, 1).reshape(9 * Cout, C).contiguous()        # row (3 dy + dx) Cout + co
    wr = cache[0].get(cache[1] + ":taps", [w], build) if cache is not None else build()
    z = conv1x1(x, wr, cache=None if cache is None else (cache[0], cache[1] + ":z"))
    out = torch.empty((B, Cout, 2 * h, 2 * w_), device=x.device, dtype=torch.float32)
    check(lib().fdn_upconv_gather(_flat(z, "z"), _flat(out, "out"), B, Cout, h, w_, stream()), "fdn_upconv_gather")
    return out


def resample(x, mode, r=1):
    B, C, H, W = x.shape
    if mode in (RS_BILINEAR_HALF, RS_NEAREST_HALF):
        shp = (B, C, H // 2, W // 2)
    elif mode in (RS_BILINEAR_X2, RS_NEAREST_X2):
        shp = (B, C, 2 * H, 2 * W)
    else:
        shp = (B, C * r * r, H // r, W // r)
    out = torch.empty(shp, device=x.device, dtype=torch.float32)
    check(lib().fdn_resample(_flat(x, "x"), _flat(out, "out"), ctypes.c_long(B * C), H, W, mode, r, stream()),
          "fdn_resample")
    return out


def dw1x1_pad1(x, w, bias):
    B, C, H, W = x.shape
    out = torch.empty((B, C, H + 2, W + 2), device=x.device, dtype=torch.float32)
    check(lib().fdn_dw1x1_pad1(_flat(x, "x"), _flat(w, "w"), _flat(bias, "bias"), _flat(out, "out"), B, C, H, W,
                               stream()), "fdn_dw1x1_pad1")
    return out


def avgpool3s2(x):
    B, C, H, W = x.shape
    out = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=x.device, dtype=torch.float32)
    check(lib().fdn_avgpool3s2(_flat(x, "x"), _flat(out, "out"), ctypes.c_long(B * C), H, W, stream()), "fdn_avgpool3s2")
    return out


def global_avgpool(x):
    B, C, H, W = x.shape
    out = torch.empty((B, C, 1, 1), device=x.device, dtype=torch.float32)
    check(lib().fdn_global_avgpool(_flat(x, "x"), _flat(out, "out"), ctypes.c_long(B * C), ctypes.c_long(H * W),
                                   stream()), "fdn_global_avgpool")
    return out


def se_apply(y, gate, shortcut):
    B, C, H, W = y.shape
    out = torch.empty_like(y)
    check(lib().fdn_se_apply(_flat(y, "y"), _flat(gate, "gate"), _flat(shortcut, "shortcut"), _flat(out, "out"),
                             ctypes.c_long(B * C), ctypes.c_long(H * W), stream()), "fdn_se_apply")
    return out


def scale_batch_(x, ratio):
    B = x.shape[0]
    check(lib().fdn_scale_batch(_flat(x, "x"), _flat(ratio, "ratio"), B, ctypes.c_long(x.numel() // B), stream()),
          "fdn_scale_batch")
    return x


SPECTRAL_MLP_C = (12, 24)          # widths the mirror sends to fdn_spectral_mlp2 (MAR's levels 1-2).  C = 48 exists in the library but is NOT used:
                                   # its 9,216 weight reads per bin make it 0.34 ms per block against 0.17 ms for the four MFMA convs (profiles/r05_f_summary.txt)
SPECTRAL_MLP_FUSED = True           # False: the four fdn_conv1x1 launches per block instead (A/B runs: bench.py --unfused-mlps)


def spectral_mlp2(mag, pha, w1m, b1m, w2m, b2m, w1p, b1p, w2p, b2p, slope=0.1):
    """mag <- process1(mag), pha <- process2(pha) IN PLACE: the two per-bin Conv1x1 -> LeakyReLU -> Conv1x1 MLPs of a FreBlock / fourier_fuse in one
    launch (fdn_spectral_mlp2; FDN_arch.py:93-94, :142-143).  mag, pha [B, C, H, Wf] contiguous."""
    B, C = mag.shape[0], mag.shape[1]
    P = mag[0, 0].numel()
    check(lib().fdn_spectral_mlp2(_flat(mag, "mag"), _flat(pha, "pha"), _flat(w1m.reshape(C, C), "w1m"), _flat(b1m, "b1m"), _flat(w2m.reshape(C, C), "w2m"),
                                  _flat(b2m, "b2m"), _flat(w1p.reshape(C, C), "w1p"), _flat(b1p, "b1p"), _flat(w2p.reshape(C, C), "w2p"), _flat(b2p, "b2p"),
                                  B, C, ctypes.c_long(P), ctypes.c_float(slope), stream()), "fdn_spectral_mlp2")
    return mag, pha


def gamma_curve(x, i_map, scale=40.0):
    out = torch.empty_like(x)
    check(lib().fdn_gamma_curve(_flat(x, "x"), _flat(i_map, "i_map"), _flat(out, "out"), ctypes.c_float(scale),
                                ctypes.c_long(x.numel()), stream()), "fdn_gamma_curve")
    return out
