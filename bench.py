#!/usr/bin/env python3
"""FDN inference bench on MI355X:  images/s of the whole LPNet -> FDN forward (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: this process spawns one rank per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             (the driver's launch; WORLD_SIZE must equal --gpus)

A step = one pass of the hot path over one batch: config 2 of BASELINE.json, 8 synthetic images of
1280x720 reflect-padded to 736x1280, fp32, already resident in HBM.  N GPUs = N shards of 8 images
(the batch shards embarrassingly) -> "scaling": "weak".  With N > 1 the global batch lives
on rank 0 and the timed region contains the RCCL scatter of the inputs and gather of the outputs that
north_star names (BASELINE.json configs[3] at N = 8: 64 images, 8 x 8); the same K steps without the
collectives are reported beside it ("without_collectives").  --dtype bf16 --height 1080 --width 1920
--batch 4 is configs[2].  Weights are deterministic synthetic (the trained FDN checkpoint is absent from
the reference checkout); LPNet uses the real LPNet_lolblur weights when tests/golden has them.

Extra JSON objects: "roofline" = the dominant INDIVIDUAL kernel (entry point at one shape = one rocprof
kernel) against its own bound, timed live with HIP events on the launch stream in an instrumented
single-stream forward; "top_kernels" = the three largest, each against its own bound; "cpu_baseline" =
the CPU oracle on a bounded sample.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in ("fdn-tip2025_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))

import torch  # noqa: E402

F_ALG_PER_PX = 1_981_721          # conv FLOPs per padded pixel
B_ALG_ELEMS_PER_PX = 7_256        # compulsory fp32 elements per padded pixel
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_F32_MFMA_TF = 157.3          # MI355X_MICROARCH.md: fp32 MFMA dense peak


def build_models(dev, variant="lolblur"):
    from basicsr.models.archs.LPNet_arch import I_predict_net
    from weights import shapes_of, synth_state_dict
    if variant == "lolv1":                                       # dim-24 model, not the headline config
        from basicsr.models.archs.fdnlol24_arch import FDN_lolv1
        net = FDN_lolv1().eval()
        net.load_state_dict(synth_state_dict(shapes_of(net), seed=7, prefix_key="fdnlol/", tame=0.03), strict=True)
    else:
        from basicsr.models.archs.FDN_arch import FDN
        net = FDN().eval()
        net.load_state_dict(synth_state_dict(shapes_of(net), seed=7, prefix_key="fdn/", tame=0.03), strict=True)
    lp = I_predict_net().eval()
    gold = os.path.join(ROOT, "tests", "golden", f"lpnet_{variant}_params.npz")
    if os.path.isfile(gold):
        import numpy as np
        z = np.load(gold)
        lp.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)
    else:
        lp.load_state_dict(synth_state_dict(shapes_of(lp), seed=7, prefix_key="lpnet/"), strict=True)
    return net.to(dev), lp.to(dev)


def make_input(batch, h, w, dev, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 3, h, w, generator=g)
    hn, wn = (32 - h % 32) % 32, (32 - w % 32) % 32
    return torch.nn.functional.pad(x, (0, wn, 0, hn), mode="reflect").to(dev).contiguous()   # inference_fdn_lolblur.py:60-62


def _iv(v):
    """ctypes argument -> Python int (plain ints pass through, c_long / c_int carry .value)."""
    return v.value if hasattr(v, "value") else v


class _Call:
    """The arguments of one C-ABI call BY NAME (fdn_hip/_abi.py ARG_NAMES, generated from include/fdn_hip.h): a reordered or inserted
    parameter of a later ABI version cannot silently shift a shape figure - a missing name raises KeyError instead."""

    def __init__(self, name, args):
        from fdn_hip._abi import ARG_NAMES
        names = ARG_NAMES[name]
        if len(names) != len(args):
            raise TypeError(f"{name}: {len(args)} arguments against {len(names)} declared in include/fdn_hip.h")
        self._kw = dict(zip(names, args))

    def i(self, *keys):
        """the named integer arguments as Python ints (one value for one key, a tuple otherwise)"""
        vals = tuple(int(_iv(self._kw[k])) for k in keys)
        return vals[0] if len(vals) == 1 else vals

    def given(self, key):
        """a pointer argument that may be NULL: True when it is set"""
        v = self._kw[key]
        return v is not None and bool(_iv(v))

    def obj(self, key):
        return self._kw[key]


def _describe_conv1x1(c):
    d = c.obj("d")._obj
    f = 2.0 * d.B * d.K * d.N * d.P
    b = d.B * d.P * ((2.0 if d.x_bf16 else 4.0) * d.K + (2.0 if d.out_bf16 else 4.0) * d.N
                     + 4.0 * ((d.N if d.epi == 1 else 0) + (2 * d.N if d.epi == 2 else 0)
                              + (d.K // 3 if d.pro == 2 else 0) + (d.K if d.pro == 3 else 0)))
    return f"fdn_conv1x1[{d.K}->{d.N},pro{d.pro},epi{d.epi},P={d.P}{',xbf16' if d.x_bf16 else ''}{',obf16' if d.out_bf16 else ''}]", f, b


def _describe_fdsa_fused(c):
    B, C, E, H, W = c.i("B", "C", "E", "H", "W")
    ob = 2.0 if c.i("out_bf16") else 4.0
    return f"fdn_fdsa_fused[C={C},E={E},{H}x{W}{',obf16' if ob == 2.0 else ''}]", 2.0 * B * H * W * C * 4 * E, B * H * W * (4.0 * C + ob * 4 * E)


def _describe_fdsa_fused_tail(c):
    # the whole sub-block: x in, res in, out + statistics out (the tile-local scratch is not algorithmic traffic); both 1x1 convs' flops;
    # Hd > 0: plus the following FDFFN's project_in (C -> Hd) and its Hd output planes
    B, C, E, H, W, Hd = c.i("B", "C", "E", "H", "W", "Hd")
    return (f"fdn_fdsa_fused_tail[C={C},E={E},{H}x{W},Hd={Hd}]", 2.0 * B * H * W * (C * 4 * E + 3 * E * C + C * Hd),
            4.0 * B * H * W * (3 * C + 4 + Hd))


def _describe_fdsa_full(c):
    B, C, E, H, W = c.i("B", "C", "E", "H", "W")
    return f"fdn_fdsa_full[C={C},E={E},{H}x{W}]", 2.0 * B * H * W * (C * 4 * E + 3 * E * C), 4.0 * B * H * W * (3 * C + 2)


def _describe_fdsa_core(c):
    B, E, H, W = c.i("B", "E", "H", "W")
    return f"fdn_fdsa_core[E={E},{H}x{W}]", 0.0, 4.0 * B * H * W * 8 * E


def _describe_fdsa_out(c):
    B, E, N, P = c.i("B", "E", "N", "P")
    ib = 2.0 if c.i("o_bf16") else 4.0
    return f"fdn_fdsa_out[E={E},N={N},P={P}{',ibf16' if ib == 2.0 else ''}]", 2.0 * B * P * 3 * E * N, B * P * (ib * 4 * E + 4.0 * 2 * N)


def _describe_fdffn_mid(c):
    B, Hd, H, W = c.i("B", "Hd", "H", "W")
    ib, ob = (2.0 if c.i("x_bf16") else 4.0), (2.0 if c.i("out_bf16") else 4.0)
    return f"fdn_fdffn_mid[Hd={Hd},{H}x{W}{',bf16' if ib + ob < 8 else ''}]", 0.0, B * H * W * Hd * (ib + ob)


def _describe_dwconv_gate(c):
    B, C, H, W = c.i("B", "C", "H", "W")
    ib, ob = (2.0 if c.i("x_bf16") else 4.0), (2.0 if c.i("out_bf16") else 4.0)
    return f"fdn_dwconv_gate[C={C},{H}x{W}{',bf16' if ib + ob < 8 else ''}]", 0.0, B * H * W * C * (ib + ob)


def _describe_ffn_tail(c):
    B, C, N, H, W = c.i("B", "C", "N", "H", "W")
    ib = 2.0 if c.i("y_bf16") else 4.0
    return (f"fdn_ffn_tail[{C}->{N},{H}x{W},form{c.i('form')}{',ibf16' if ib == 2.0 else ''}]", 2.0 * B * H * W * C * N,
            B * H * W * (ib * C + 4.0 * 2 * N))


def _describe_rfft_rows(c):
    rows, W = c.i("rows", "W")
    return f"fdn_rfft_rows[W={W},rows={rows}]", 0.0, 4.0 * rows * (W + 2 * (W // 2 + 1))


def _describe_irfft_rows(c):
    planes, H, W = c.i("planes", "H", "W")
    return f"fdn_irfft_rows[{H}x{W},planes={planes}]", 0.0, 4.0 * planes * H * (2 * (W // 2 + 1) + W + (W if c.given("res") else 0))


def _describe_fft_cols_fcaffn(c):
    B, C, H, Wf = c.i("B", "C", "H", "Wf")
    return f"fdn_fft_cols_fcaffn[C={C},{H}x{Wf}]", 0.0, 4.0 * B * H * Wf * (4 * C + 8)


def _describe_rfft_rows_ln(c):
    B, C, H, W = c.i("B", "C", "H", "W")
    return f"fdn_rfft_rows_ln[W={W},rows={B * C * H}]", 0.0, 4.0 * B * H * (C * (W + 2 * (W // 2 + 1)) + 2 * W)


def _describe_fcaffn_in(c):
    B, C, H, W = c.i("B", "C", "H", "W")
    return f"fdn_fcaffn_in[C={C},{H}x{W}]", 2.0 * B * H * W * C * (C + 2 * 27), 4.0 * B * H * W * (3 * C + 3)


def _describe_fcaffn_in_packed(c):
    B, C, H, W = c.i("B", "C", "H", "W")
    return f"fdn_fcaffn_in_packed[C={C},{H}x{W}]", 2.0 * B * H * W * C * (C + 2 * 27), 4.0 * B * H * W * (3 * C + 3 + 4)   # xi, x1, out, the image, two statistics pairs


def _describe_conv2d(c):
    B, Cin, H, W, Cout, KH, KW, st, pad = c.i("B", "Cin", "H", "W", "Cout", "KH", "KW", "stride", "pad")
    OH, OW = (H + 2 * pad - KH) // st + 1, (W + 2 * pad - KW) // st + 1
    return (f"fdn_conv2d[{Cin}->{Cout},k{KH}s{st},{H}x{W}]", 2.0 * B * OH * OW * Cin * Cout * KH * KW,
            4.0 * B * (Cin * H * W + Cout * OH * OW * (2 if c.given("res") else 1)))


def _describe_upconv_gather(c):
    B, Cout, h, w = c.i("B", "Cout", "h", "w")
    return f"fdn_upconv_gather[Cout={Cout},{h}x{w}]", 0.0, 4.0 * B * h * w * (9 * Cout + 4 * Cout)


def _describe_chan_stats(c):
    B, G, E, P = c.i("B", "G", "E", "P")
    return f"fdn_chan_stats[G={G},E={E},P={P}]", 0.0, 4.0 * B * P * (G * E + 2 * G)


def _describe_layernorm_chan(c):
    B, C, P = c.i("B", "C", "P")
    return f"fdn_layernorm_chan[C={C},P={P}]", 0.0, 4.0 * B * P * 2 * C


def _describe_img_mod_maps(c):
    B, C, H, W = c.i("B", "C", "H", "W")
    return f"fdn_img_mod_maps[C={C},{H}x{W}]", 0.0, 4.0 * B * H * W * (3 + 2 * C)


# entry point -> parser of its call.  tests/test_host_cpu.py feeds every parser a call built from the header's own prototype and checks that
# the key carries the values under their NAMES; entry points without a parser are grouped under their bare name with no roofline figure.
DESCRIBERS = {
    "fdn_conv1x1": _describe_conv1x1, "fdn_fdsa_fused": _describe_fdsa_fused, "fdn_fdsa_fused_tail": _describe_fdsa_fused_tail, "fdn_fdsa_full": _describe_fdsa_full,
    "fdn_fdsa_core": _describe_fdsa_core, "fdn_fdsa_out": _describe_fdsa_out, "fdn_fdffn_mid": _describe_fdffn_mid,
    "fdn_dwconv_gate": _describe_dwconv_gate, "fdn_ffn_tail": _describe_ffn_tail, "fdn_rfft_rows": _describe_rfft_rows,
    "fdn_irfft_rows": _describe_irfft_rows, "fdn_fft_cols_fcaffn": _describe_fft_cols_fcaffn, "fdn_rfft_rows_ln": _describe_rfft_rows_ln,
    "fdn_fcaffn_in": _describe_fcaffn_in, "fdn_fcaffn_in_packed": _describe_fcaffn_in_packed, "fdn_conv2d": _describe_conv2d,
    "fdn_chan_stats": _describe_chan_stats, "fdn_layernorm_chan": _describe_layernorm_chan, "fdn_img_mod_maps": _describe_img_mod_maps,
    "fdn_upconv_gather": _describe_upconv_gather,
}


def describe_call(name, a):
    """(group key, algorithmic FLOPs, algorithmic HBM bytes) of one C-ABI call: the figures of DESIGN.md (each operand read once, each
    result written once, fp32 unless a *_bf16 flag is set), conv FLOPs only (FFT / pointwise work is not counted)."""
    fn = DESCRIBERS.get(name)
    if fn is None:
        return name, 0.0, 0.0
    return fn(_Call(name, a))


class KernelTimer:
    """Wrap every C-ABI entry point with HIP events on the launch stream (one instrumented single-stream forward).
    Calls are grouped per entry point AND shape: at a fixed shape an entry point always launches the same kernel
    instantiation, so a group is one individual rocprof kernel."""

    def __init__(self):
        import fdn_hip
        self.lib = fdn_hip.lib()
        self.records = []
        decl = [ln.split("(")[0].split()[-1] for ln in open(os.path.join(ROOT, "include", "fdn_hip.h")) if ln.startswith("int fdn_")]
        self.names = [n for n in decl if n not in ("fdn_abi_version", "fdn_fft_prepare")]
        self.orig = {}

    def __enter__(self):
        for n in self.names:
            f = getattr(self.lib, n)
            self.orig[n] = f

            def wrapped(*a, _f=f, _n=n):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = _f(*a)
                e1.record()
                self.records.append((_n, e0, e1) + describe_call(_n, a))
                return r
            setattr(self.lib, n, wrapped)
        return self

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(self.lib, n, f)

    def summary(self):
        """{group key: [entry point, launches, ms, flops, bytes]}"""
        torch.cuda.synchronize()
        agg = {}
        for n, e0, e1, key, fl, by in self.records:
            t = agg.setdefault(key, [n, 0, 0.0, 0.0, 0.0])
            t[1] += 1
            t[2] += e0.elapsed_time(e1)
            t[3] += fl
            t[4] += by
        return agg


def kernel_roofline(key, rec, traffic):
    """One individual kernel against its OWN bound: the larger of (algorithmic bytes / 8 TB/s) and (conv FLOPs /
    157.3 TFLOP/s) decides whether it is priced against HBM or the fp32 matrix cores."""
    n, cnt, ms, fl, by = rec
    t = ms * 1e-3
    t_hbm, t_mfma = by / (PEAK_HBM_GBS * 1e9), fl / (PEAK_F32_MFMA_TF * 1e12)
    out = {"kernel": key, "launches": cnt, "avg_ms": ms / cnt}
    if t_mfma > t_hbm:
        ach = fl / t / 1e12
        out.update({"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TF})
    elif by > 0:
        ach = by / t / 1e9
        out.update({"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS})
    else:
        out.update({"bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None})
    out["algorithmic_bytes_per_launch"] = by / cnt if by else None
    out["traffic"] = traffic.get(key) if traffic else None
    return out


def matching_profile(shape, dtype):
    """The newest committed profiles/*traffic_groups.json whose [batch, height, width] and storage dtype are the benchmarked ones
    (tools/profile_merge.py output), or None."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic_groups.json")), reverse=True):
        d = json.load(open(f))
        if list(d.get("shape", [])) == list(shape) and d.get("dtype", "f32") == dtype:
            d["file"] = os.path.relpath(f, ROOT)
            return d
    return None


def pmc_traffic_by_group(shape, dtype):
    """HBM bytes per launch of each kernel group from the committed rocprofv3 PMC passes: the newest
    profiles/*traffic_groups.json (written by tools/profile_merge.py: the memory-side request counters by size, FETCH_SIZE and
    WRITE_SIZE beside them, each in its own --pmc run over one logged forward) whose [batch, height, width] and storage dtype
    are the ones being benchmarked - bytes per launch depend on all of them.  {} when no matching profile is committed."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic_groups.json")), reverse=True):
        d = json.load(open(f))
        if list(d.get("shape", [])) == list(shape) and d.get("dtype", "f32") == dtype:
            return {k: v["hbm_bytes_per_launch"] for k, v in d["groups"].items()}
    return {}


def rank_cpus(local_rank, world):
    """Host cores for this rank (one process per GPU; ~2,400 launches per step are issued from Python, so eight ranks must not pile
    onto the same cores or sit across the socket from their GPU).  GPU r's NUMA-local cores from sysfs (`local_cpulist` of the
    PCI function that HIP device r is - every PCI domain, device order taken from the KFD topology and the *_VISIBLE_DEVICES
    index lists), cut evenly between the ranks that share them; an even cut of the allowed cores when the order of the devices
    cannot be established or sysfs has nothing to say.  Read-only, before any GPU call, no exec."""
    import glob
    allowed = sorted(os.sched_getaffinity(0))

    def cpulist(txt):
        out = set()
        for part in txt.strip().split(","):
            if part:
                lo, _, hi = part.partition("-")
                out.update(range(int(lo), int(hi or lo) + 1))
        return out

    def gpu_pci_ids():
        """PCI ids (dddd:bb:dd.f) of the GPUs in HIP device order, without touching the GPU runtime: the KFD topology lists the
        agents in the order ROCr enumerates them (nodes with SIMDs are GPUs; `domain` + `location_id` = PCI address), and
        ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES (index lists) select from it.  None when the order cannot be established."""
        nodes = []
        for nd in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda q: int(os.path.basename(q))):
            try:
                props = dict(ln.split()[:2] for ln in open(os.path.join(nd, "properties")) if len(ln.split()) >= 2)
            except OSError:
                return None
            if int(props.get("simd_count", "0")) > 0:
                loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
                nodes.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            v = os.environ.get(var)
            if v:
                try:
                    nodes = [nodes[int(i)] for i in v.split(",")]
                except (ValueError, IndexError):
                    return None                 # UUID lists or stale indices: the order is unknown
        return nodes or None

    local = []
    ids = gpu_pci_ids()
    if ids is not None:
        for pid in ids:
            try:
                local.append(frozenset(cpulist(open(f"/sys/bus/pci/devices/{pid}/local_cpulist").read()) & set(allowed)))
            except OSError:
                local = []
                break
    if len(local) >= world and local[local_rank]:
        mine = local[local_rank]
        sharers = [r for r in range(world) if local[r] == mine]
        cores = sorted(mine)
        k, n = sharers.index(local_rank), len(sharers)
        cut = cores[k * len(cores) // n:(k + 1) * len(cores) // n]
        if cut:
            return cut, "numa-local (sysfs local_cpulist)"
    cut = allowed[local_rank * len(allowed) // world:(local_rank + 1) * len(allowed) // world]
    return (cut or allowed), "even cut of the allowed cores"


def usable_cores():
    """Host cores this process may actually use: min(cpu_count, affinity mask, cgroup cpu.max quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_baseline(h=256, w=256, full_720p=False):
    """The CPU oracle (fp32, PyTorch CPU ops = a port of the reference's algorithm) on ONE h x w image - BASELINE.json
    configs[0] as it stands - timed the way SURVEY.md 8(d) asks: one warm-up, then the median of three runs, at n = 8 threads
    (comparable with the build container's probe numbers) and at n = all usable host cores.  `value` is the all-cores figure in
    the unit of the metric, i.e. SCALED BY PIXEL COUNT to a 736 x 1280 image: that scaling FLATTERS the CPU - the reference
    itself measured 0.0028 images/s at 736 x 1280 on 8 threads because its working set leaves the caches, which a
    bounded sample cannot show.  `value_config0` is the unscaled measurement."""
    import statistics
    import fdn_oracle as O
    from weights import synth_state_dict
    from common import fdn_shapes, lpnet_weights
    P = synth_state_dict(fdn_shapes(), 7, prefix_key="fdn/", tame=0.03)
    PL = lpnet_weights()
    x = torch.rand(1, 3, h, w, generator=torch.Generator().manual_seed(0))

    def one():
        with torch.no_grad():
            t0 = time.perf_counter()
            r = O.lpnet_forward(PL, x)
            O.fdn_forward(P, x, r)
            return time.perf_counter() - t0

    cores = usable_cores()
    runs = {}
    for n in sorted({min(8, cores), cores}):
        torch.set_num_threads(n)
        one()                                                  # warm-up (allocator, oneDNN / MKL plans)
        runs[n] = sorted(one() for _ in range(3))
    dt = statistics.median(runs[cores])
    px_ratio = (h * w) / (736.0 * 1280.0)
    out = {"value": px_ratio / dt, "unit": "images/s (736x1280-equivalent: the sample SCALED BY PIXEL COUNT, which flatters the CPU)",
           "cores": cores, "kind": "port", "value_config0": 1.0 / dt,
           "unit_config0": f"images/s at {h}x{w} (BASELINE.json configs[0], unscaled)",
           "protocol": "1 warm-up + median of 3",
           "by_threads": {str(n): {"images_per_s_config0": 1.0 / statistics.median(v), "seconds": [round(t, 3) for t in v]} for n, v in runs.items()},
           "reference_720p_note": "the reference itself: 0.0028 images/s at 736x1280 on 8 threads in the build container (SURVEY.md 6)",
           "sample": f"1 image {h}x{w} fp32, LPNet+FDN oracle forward, median {dt:.2f} s wall at {cores} threads"}
    if full_720p:                                                  # ONE real run at the metric's own size, all cores (no warm-up at this size:
        x = torch.rand(1, 3, 720, 1280, generator=torch.Generator().manual_seed(0))        # the 256 x 256 runs above warmed the libraries)
        x = torch.nn.functional.pad(x, (0, 0, 0, 16), mode="reflect")
        torch.set_num_threads(cores)
        t720 = one()
        out["measured_736x1280"] = {"images_per_s": 1.0 / t720, "seconds": round(t720, 2), "threads": cores,
                                    "note": "one timed oracle forward of ONE padded 736x1280 image (no scaling): the CPU figure in the metric's own unit"}
        out["measured_720p"] = out["measured_736x1280"]            # (the name VERDICT r5 item 4 uses)
    return out


def parity_against_reference(forward, x):
    """(BASELINE.json metric "...; PSNR vs ref") The graph that was just timed, replayed ONCE with the configs[1] fixture frame in
    batch slot 0 (the seeded 720 x 1280 frame of tests/golden/fdn_tamed_736x1280.npz, reflect-padded; the other slots keep the bench input: samples do
    not interact) and its `result` judged by the rule of tests/test_gpu_configs.py::test_config1_736x1280_frame_matches_reference: 64 seeded 32 x 32
    windows against the REFERENCE's own fp32 forward on that frame (generated in the build container by tests/golden/make_golden_configs.py) and
    against the float64 truth, every window within 4 x what fp32 is known to do at that window + 5e-8.  Nothing here reads /root/reference."""
    import numpy as np
    gold = os.path.join(ROOT, "tests", "golden")
    z, z64 = np.load(os.path.join(gold, "fdn_tamed_736x1280.npz")), np.load(os.path.join(gold, "fdn_tamed_736x1280_f64.npz"))
    if abs(float(z["tame"]) - 0.03) > 1e-6:
        return {"skipped": "fixture weights differ from the bench's"}
    f = torch.rand(1, 3, 720, 1280, generator=torch.Generator().manual_seed(int(z["x_seed"])))
    f = torch.nn.functional.pad(f, (0, 0, 0, 16), mode="reflect")
    if abs(f.double().sum().item() - float(z["x_sum64"])) > 1e-6:
        return {"skipped": "the seeded frame is not the one the fixture was made from"}
    xb = x.clone()
    xb[0] = f[0].to(x.device)
    y = forward(xb)[0:1].float().cpu()
    ratio = getattr(forward, "ratio", None)
    rerr = None if ratio is None else float((ratio[0:1].cpu() - torch.from_numpy(z["ratio"])).abs().max())
    org, win, t64 = z["y_org"], torch.from_numpy(z["y_win"]).double(), torch.from_numpy(z64["y_win64"])
    mine = torch.stack([y[0, :, y0:y0 + 32, x0:x0 + 32] for y0, x0 in org.tolist()]).double()
    rms = lambda t: (t ** 2).mean((1, 2, 3)).sqrt()
    e_hip, e_ref = rms(mine - t64), rms(win - t64)
    cap = torch.maximum(torch.maximum(e_ref, torch.from_numpy(z64["y_susc"]).max(0)[0]), torch.from_numpy(z64["y_susc_noise"]).max(0)[0])
    bad = (e_hip > 4.0 * cap + 5e-8).nonzero().flatten().tolist()
    mse = float(((mine - win) ** 2).mean())
    d = y.double()
    mom = torch.from_numpy(z["y_mom"])
    n = y.shape[2] * y.shape[3]
    return {"frame": "tests/golden/fdn_tamed_736x1280.npz (seeded 720x1280 frame, reflect-padded; the reference's own forward, 64 windows of `result`)",
            "graph": "the captured graph of the timed steps, fixture frame in batch slot 0",
            "psnr_vs_reference_windows": (float("inf") if mse == 0 else 10.0 * math.log10(1.0 / mse)),
            "psnr_vs_float64_windows": 10.0 * math.log10(1.0 / float(((mine - t64) ** 2).mean())),
            "reference_psnr_vs_float64_windows": 10.0 * math.log10(1.0 / float(((win - t64) ** 2).mean())),
            "windows": 64, "windows_failing": len(bad), "failing": bad[:8], "rule": "err(HIP, f64)[w] <= 4 max(err(ref fp32, f64)[w], susceptibility[w]) + 5e-8",
            "median_window_err_vs_f64": {"hip": float(e_hip.median()), "reference": float(e_ref.median())},
            "mean_abs_err_per_channel": float(((d.sum((0, 2, 3)) - mom[0]).abs() / n).max()),
            "ratio_abs_err": rerr, "ok": (not bad) and (rerr is None or rerr <= 5e-6)}


class GpuSensors:
    """Shader clock and socket power of the benchmarked GPU while the timed steps run, sampled from sysfs by a host thread every 50 ms (hwmon
    `freq1_input` in Hz, `power1_average` / `power1_input` in microwatts; amdgpu's pp_dpm_sclk as a fall-back for the clock).  Read-only, no tool
    is started, nothing is set.  The package runs at its power cap under these kernels: box-to-box and loop-vs-step
    differences show up here."""

    def __init__(self, device_index=0):
        import glob
        self.freq = self.power = self.dpm = None
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        bus = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        pick = [c for c in cards if bus and bus in os.path.realpath(c)] or cards
        for c in pick:
            hw = sorted(glob.glob(os.path.join(c, "hwmon", "hwmon*")))
            for h in hw:
                f = os.path.join(h, "freq1_input")
                pw = [q for q in (os.path.join(h, "power1_average"), os.path.join(h, "power1_input")) if os.path.isfile(q)]
                if os.path.isfile(f) or pw:
                    self.freq = f if os.path.isfile(f) else None
                    self.power = pw[0] if pw else None
                    self.dpm = os.path.join(c, "pp_dpm_sclk") if os.path.isfile(os.path.join(c, "pp_dpm_sclk")) else None
                    self.card = c
                    break
            if self.freq or self.power:
                break
        self.samples = []
        self._stop = None

    @staticmethod
    def _read(path):
        try:
            return open(path).read()
        except Exception:
            return None

    def _one(self):
        mhz = w = None
        t = self._read(self.freq) if self.freq else None
        if t and t.strip().isdigit():
            mhz = int(t) / 1e6
        elif self.dpm:
            t = self._read(self.dpm)
            if t:
                cur = [ln for ln in t.splitlines() if ln.rstrip().endswith("*")]
                if cur:
                    mhz = float("".join(ch for ch in cur[0].split(":")[1] if ch.isdigit() or ch == "."))
        t = self._read(self.power) if self.power else None
        if t and t.strip().isdigit():
            w = int(t) / 1e6
        return mhz, w

    def __enter__(self):
        import threading
        self.samples = []
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                self.samples.append(self._one())
                self._stop.wait(0.05)
        self._th = threading.Thread(target=loop, daemon=True)
        if self.freq or self.power or self.dpm:
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th.is_alive():
            self._th.join()

    def summary(self):
        f = [a for a, _ in self.samples if a]
        p = [b for _, b in self.samples if b]
        if not f and not p:
            return {"available": False, "note": "no readable hwmon clock / power file for this GPU"}
        out = {"available": True, "samples": len(self.samples), "source": "sysfs hwmon, 50 ms period, host thread, over the timed steps"}
        if f:
            out.update({"clock_mhz_mean": sum(f) / len(f), "clock_mhz_min": min(f), "clock_mhz_max": max(f)})
        if p:
            out.update({"power_w_mean": sum(p) / len(p), "power_w_max": max(p)})
        return out


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: this process never touches a GPU; it starts one child per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, exactly what torch.distributed.run would set) and
    returns the worst exit code.  Rank 0's JSON line goes to our stdout."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e))
    rc = 0
    try:
        live = list(procs)
        while live and not rc:                  # poll ALL ranks: whichever fails first ends the job (a rank blocked in a collective
            for p in list(live):                # behind a dead peer would otherwise hold us until the RCCL watchdog fires)
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    rc = max(rc, abs(code))
            if live and not rc:
                time.sleep(0.05)
    finally:
        for p in procs:                        # a failed rank must not leave the others waiting in a collective
            if p.poll() is None:
                p.kill()
                p.wait()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the per-GPU batch is split over; > 1 is for experiments only: kernels of different "
                    "streams that overlap a bf16-MFMA kernel return wrong rows on MI355X / ROCm 7.2 (DESIGN.md 4.7), 1 is the product path")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32 = BASELINE.json configs[1] (the reference's arithmetic); bf16 = configs[2]: bf16 STORAGE of the "
                         "block-internal activations between kernels, fp32 math / FFT / residual stream (DESIGN.md)")
    ap.add_argument("--scatter-gather", dest="sg", action="store_true", default=None,
                    help="put the RCCL scatter of the inputs / gather of the outputs in the timed region (default: on when N > 1)")
    ap.add_argument("--no-scatter-gather", dest="sg", action="store_false")
    ap.add_argument("--variant", choices=("lolblur", "lolv1"), default="lolblur",
                    help="lolblur = FDN (BASELINE.json's metric); lolv1 = FDN_lolv1, dim 24 (SURVEY.md 8(f) rank 1)")
    ap.add_argument("--no-affinity", action="store_true", help="do not pin the ranks of a multi-GPU run to NUMA-local cores")
    ap.add_argument("--graph", dest="graph", action="store_true", default=True,
                    help="(default) replay the per-rank step from ONE captured HIP graph instead of ~2,400 eager launches: same kernels, "
                         "same results bit for bit, and the host issues one hipGraphLaunch per step (eight ranks share one host's cores). "
                         "The capture happens in this process before the warm-up steps and is not timed")
    ap.add_argument("--eager", dest="graph", action="store_false", help="issue every launch from Python instead of replaying the captured graph")
    ap.add_argument("--config", choices=("fdn", "lpnet"), default="fdn",
                    help="fdn = LPNet -> FDN (the metric); lpnet = I_predict_net alone on the same batch (BASELINE.json configs[4] / SURVEY 8d C5)")
    ap.add_argument("--narrow-pipe", action="store_true", help="A/B: fdn_set_matrix_pipe(2) - the level-2 FDSA tail on its fp32-MFMA form, the default of ABI 10 "
                    "(since round 5 the default runs it on the bf16 matrix pipe)")
    ap.add_argument("--stats-launches", action="store_true", help="A/B: fdn_chan_stats launches in front of the level-3 LN3 / FCAFFN GEMMs instead of the in-kernel statistics pass")
    ap.add_argument("--resample-upsample", action="store_true", help="A/B: Upsample as fdn_resample x2 + the 3x3 conv instead of the low-resolution 1x1 conv per tap + fdn_upconv_gather")
    ap.add_argument("--aff-resized", action="store_true", help="A/B: MAR's fourier_fuse on nearest-resized copies (one 84-channel 1x1 conv) instead of one 1x1 conv per source resolution")
    ap.add_argument("--unfused-mlps", action="store_true", help="A/B: MAR's per-bin MLPs as four fdn_conv1x1 launches per block instead of one fdn_spectral_mlp2")
    ap.add_argument("--cpu-720p", action="store_true", help="cpu_baseline also times ONE real 736 x 1280 oracle forward (BASELINE.md 4.3 'single timed run at "
                    "720p': minutes of host time, off by default; the committed line is profiles/r05_cpu_720p.json)")
    ap.add_argument("--fdsa-full", action="store_true", help="A/B: route the level-1 FDSA sub-blocks through fdn_fdsa_full (one launch) instead of "
                    "fdn_fdsa_fused + fdn_fdsa_out (DESIGN.md section 4: built, correct, not the default)")
    ap.add_argument("--fdsa-pair", action="store_true", help="A/B: the FDSA sub-blocks of levels 1-2 as fdn_fdsa_fused + fdn_fdsa_out (two launches, the 4E-plane hand-off "
                    "through HBM: the round-5 route) instead of fdn_fdsa_fused_tail")
    ap.add_argument("--tail-pin-l2", action="store_true", help="A/B: the level-2 FDFFN project_in (64 -> 172) inside fdn_fdsa_fused_tail as well (default: level 1 only; measured slower at step level)")
    ap.add_argument("--no-tail-pin", action="store_true", help="A/B: the FDFFN project_in of level 1 as its own fdn_conv1x1 launch instead of inside fdn_fdsa_fused_tail")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short side measurements of BASELINE.json configs[2] (1080p B = 4 bf16 storage) "
                    "and configs[4] (LPNet alone) that the default headline run appends as `other_configs`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the replay of the timed graph on the configs[1] fixture frame (`parity` in the JSON line)")
    ap.add_argument("--no-cpu-720p", action="store_true", help="cpu_baseline without the ONE real 736 x 1280 oracle forward (~3 minutes of host time) that the "
                    "default headline run includes")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU / gloo rehearsal of the launcher, sharding, scatter / gather and timing protocol with a stand-in "
                         "forward (tests/test_multirank_cpu.py); measures nothing")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))       # parent: no GPU call before or after this line
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # stdout carries ONE line, the JSON record of rank 0: everything else a rank or a library writes to file descriptor 1 (RCCL prints a version
    # banner through C stdio, flushed at exit - i.e. BEHIND a line printed earlier) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world != a.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {a.gpus}: the two must agree")

    if a.dry_run and os.environ.get("FDN_BENCH_DRY_FAIL_RANK") == str(rank):
        sys.exit(3)                                        # rehearsal of a crashed rank (tests/test_multirank_cpu.py)
    affinity = None
    if world > 1 and not a.no_affinity:
        cpus, how = rank_cpus(local_rank, world)
        os.sched_setaffinity(0, cpus)                      # before the first GPU call of this process
        affinity = {"cores": len(cpus), "first": cpus[0], "last": cpus[-1], "how": how}

    from fdn_hip import sharding
    dist = None
    if a.dry_run:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU: the FDN path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("FDN_BENCH_FORCE_DIST") == "1":      # (the env switch exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.dry_run:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
    sg = (world > 1) if a.sg is None else a.sg
    sg = sg and dist is not None

    if a.dry_run:
        x = torch.rand(a.batch, 3, 16, 24, generator=torch.Generator().manual_seed(1000 + rank))
        mk = lambda r: torch.rand(a.batch, 3, 16, 24, generator=torch.Generator().manual_seed(1000 + r))
        forward = lambda t: t * 2.0 + 1.0
        sync = lambda: None
    else:
        from fdn_hip.pipeline import forward_streams
        import fdn_hip
        fdn_hip.set_storage_dtype(a.dtype)
        fdn_hip.ops.FDSA_FULL = bool(a.fdsa_full)
        fdn_hip.ops.FDSA_TAIL = not a.fdsa_pair
        fdn_hip.ops.FDSA_TAIL_PIN = not a.no_tail_pin
        fdn_hip.ops.FDSA_TAIL_PIN_MAX_C = 64 if a.tail_pin_l2 else 32
        if a.narrow_pipe:
            fdn_hip.set_matrix_pipe("bf16-narrow")
        fdn_hip.ops.SPECTRAL_MLP_FUSED = not a.unfused_mlps
        fdn_hip.ops.GEMM_OWN_STATS = not a.stats_launches
        fdn_hip.ops.UPCONV_GATHER = not a.resample_upsample
        fdn_hip.ops.AFF_MULTIRES = not a.aff_resized
        net, lp = build_models(dev, a.variant)
        x = make_input(a.batch, a.height, a.width, dev, seed=1000 + rank)
        mk = lambda r: make_input(a.batch, a.height, a.width, dev, seed=1000 + r)
        forward = lambda t: forward_streams(net, lp, t, a.streams)     # LPNet -> FDN on one stream (--streams > 1: experiments, see above)
        if a.config == "lpnet":
            def forward(t):                                            # C5: the second arch alone
                with torch.no_grad():
                    return lp(t)
        if a.graph:
            if a.config == "lpnet":
                eager_lp = forward
                g_lp = {}

                def forward(t):
                    if "g" not in g_lp:
                        g_lp["x"] = t.clone()
                        for _ in range(2):
                            eager_lp(g_lp["x"])
                        torch.cuda.synchronize()
                        g_lp["g"] = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g_lp["g"], capture_error_mode="thread_local"):
                            g_lp["out"] = eager_lp(g_lp["x"])
                    g_lp["x"].copy_(t)
                    g_lp["g"].replay()
                    return g_lp["out"]
            else:
                from fdn_hip.pipeline import GraphedStep
                forward = GraphedStep(net, lp, a.streams)
            forward(x)                                                 # capture now: same process, before the warm-up, untimed
            torch.cuda.synchronize()
        sync = torch.cuda.synchronize
    B, _, H, W = x.shape
    root_in = root_out = None
    if sg and rank == 0:
        root_in = [mk(r) for r in range(world)]                        # the global batch lives on the root: N x B images
        root_out = [torch.empty_like(x) for _ in range(world)]

    def step(with_sg):
        if with_sg:
            return sharding.sharded_step(dist, forward, x, root_in, root_out)
        return forward(x)

    def barrier():
        if dist is not None:
            dist.barrier()
        sync()

    host_s = [0.0]

    def timed(with_sg, steps):
        barrier()
        t0 = time.perf_counter()
        if not a.dry_run:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for _ in range(steps):
            step(with_sg)
        if not a.dry_run:
            e1.record()
        host_s[0] = time.perf_counter() - t0               # host time to ISSUE the K steps (the GPU is still running)
        barrier()
        dt = time.perf_counter() - t0
        if not a.dry_run:
            dt = max(dt, e0.elapsed_time(e1) / 1e3)
        return sharding.max_over_ranks(dist, dt, dev) if dist is not None else dt

    for _ in range(a.warmup):
        step(sg)
    sensors = None
    if rank == 0 and not a.dry_run:
        sensors = GpuSensors(local_rank)
        with sensors:
            dt = timed(sg, a.steps)                                    # THE measurement: exactly K steps
    else:
        dt = timed(sg, a.steps)
    host_issue_ms = host_s[0] / a.steps * 1e3
    dt_nosg = None
    if sg:                                                             # beside it: the same K steps without the collectives
        dt_nosg = timed(False, a.steps)
    if a.dry_run and sg and rank == 0:                                 # the rehearsal also checks the plumbing
        ok = all(torch.equal(o, forward(i)) for o, i in zip(root_out, root_in))
        assert ok, "dry run: gathered outputs differ from the root's own forward"

    parity = None
    if (rank == 0 and not a.dry_run and a.graph and a.config == "fdn" and a.variant == "lolblur" and a.dtype == "f32" and a.streams == 1
            and (a.height, a.width) == (720, 1280) and not a.no_parity):
        parity = parity_against_reference(forward, x)

    default_routing = not (a.fdsa_full or a.fdsa_pair or a.no_tail_pin or a.tail_pin_l2 or a.narrow_pipe or a.unfused_mlps or a.stats_launches or a.resample_upsample or a.aff_resized)

    def roofline_of(xin, dtype, hw, lpnet_only=False):
        """(roofline of the dominant individual kernel, top 3, the matching committed PMC profile or None) from two instrumented single-stream
        forwards of `xin` in the current storage mode."""
        agg = None
        for _ in range(2):                                          # the first instrumented pass also pays one-off host costs
            with KernelTimer() as kt:
                with torch.no_grad():
                    if lpnet_only:
                        lp(xin)
                    else:
                        net(xin, ratio_i=lp(xin), device=dev)       # single stream: events bracket each launch
            cur = kt.summary()
            agg = cur if agg is None else {k: (v if v[2] <= agg.get(k, v)[2] else agg[k]) for k, v in cur.items()}
        total_ms = sum(v[2] for v in agg.values())
        # figures taken from the committed PMC passes describe the DEFAULT routing of the code they were recorded on: with an A/B route
        # switched on (--fdsa-pair, --narrow-pipe) the kernels differ, so nothing is borrowed from them
        prof_ = matching_profile([xin.shape[0], hw[0], hw[1]], dtype) if (not lpnet_only and a.variant == "lolblur" and default_routing) else None
        traffic = {k: v["hbm_bytes_per_launch"] for k, v in prof_["groups"].items()} if prof_ else {}
        ranked = sorted(agg.items(), key=lambda kv: -kv[1][2])
        top_ = []
        for key, rec in ranked[:3]:
            r_ = kernel_roofline(key, rec, traffic)
            r_["share_of_step"] = rec[2] / total_ms
            g_ = prof_["groups"].get(key) if prof_ else None
            if g_ and "simd_time_frac" in g_:                       # what actually bounds it: issue time of the two pipes (committed PMC passes)
                r_["simd_busy_from_profile"] = {"valu": round(g_["simd_time_frac"]["valu"], 3), "mfma": round(g_["simd_time_frac"]["mfma"], 3),
                                                "profile": prof_["file"], "profile_avg_ms": g_["avg_ms"]}
            top_.append(r_)
        roof_ = dict(top_[0])                                       # the dominant INDIVIDUAL kernel
        ent = {}
        for key, rec in agg.items():
            ent[rec[0]] = ent.get(rec[0], 0.0) + rec[2]
        roof_["single_stream_forward_ms"] = total_ms
        roof_["by_entry_point_ms"] = {k: round(v, 3) for k, v in sorted(ent.items(), key=lambda kv: -kv[1])}
        return roof_, top_, prof_

    def whole_path_of(dtype, P_, batch, ips_per_gpu, step_s, prof_):
        """B_alg = 7,256 elements per padded pixel x sizeof(elem) - 4 bytes in fp32, 2 in the bf16-storage configuration"""
        eb = 4.0 if dtype == "f32" else 2.0
        wp = {"hbm_algorithmic_frac": B_ALG_ELEMS_PER_PX * eb * P_ * ips_per_gpu / (PEAK_HBM_GBS * 1e9),
              "mfma_f32_frac": F_ALG_PER_PX * P_ * ips_per_gpu / (PEAK_F32_MFMA_TF * 1e12)}
        if prof_ is not None:
            # SURVEY 8(d) `roofline.measured`: the bytes a step moves ACCORDING TO THE COMMITTED PMC PASSES of this shape (memory-side read
            # requests by size + write requests, one forward of B images, recorded on the code of that profile - named in `profile` /
            # `profile_commit`, NOT measured in this run) over THIS run's step time, against 8 TB/s
            moved = (prof_["total"]["read_GB"] + prof_["total"]["write_GB"]) * 1e9
            wp.update({"hbm_from_committed_profile_frac": moved / step_s / (PEAK_HBM_GBS * 1e9),
                       "hbm_from_committed_profile_GB_per_step": moved / 1e9,
                       "hbm_from_committed_profile_over_algorithmic": moved / (B_ALG_ELEMS_PER_PX * eb * P_ * batch),
                       "profile": prof_["file"], "profile_commit": prof_.get("commit")})
        return wp

    roof = top = None
    prof = None
    if rank == 0 and not a.no_roofline and not a.dry_run:
        roof, top, prof = roofline_of(x, a.dtype, (a.height, a.width), lpnet_only=a.config == "lpnet")

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.dry_run:
        headline_default = (a.config == "fdn" and a.variant == "lolblur" and a.dtype == "f32" and (a.height, a.width, a.batch) == (720, 1280, 8))
        cpu = cpu_baseline(full_720p=a.cpu_720p or (headline_default and not a.no_cpu_720p))

    # BASELINE.json names two more single-GPU configurations: they are measured here, briefly, in the same process and by the same protocol (captured
    # graph, untimed warm-up, K steps between synchronisations), so that every run of the headline command records them too (VERDICT r4, row g)
    other = None
    if (rank == 0 and world == 1 and not a.dry_run and not a.no_other_configs and a.config == "fdn" and a.variant == "lolblur" and a.graph
            and a.dtype == "f32" and (a.height, a.width, a.batch) == (720, 1280, 8) and not (a.fdsa_full or a.fdsa_pair or a.no_tail_pin or a.tail_pin_l2 or a.narrow_pipe or a.unfused_mlps or a.stats_launches or a.resample_upsample or a.aff_resized)):
        from fdn_hip.pipeline import GraphedStep

        def side(fn, xin, steps=3):
            fn(xin)                                                    # capture + warm-up, untimed
            fn(xin)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                fn(xin)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / steps
        other = {}
        try:
            fdn_hip.set_storage_dtype("bf16")
            x2 = make_input(4, 1080, 1920, dev, seed=2000)
            ms2 = side(GraphedStep(net, lp, 1), x2, steps=10)
            other["configs[2]"] = {"workload": "FDN 1920x1080 (padded 1920x1088) batch=4, bf16 storage of block-internal activations, fp32 math", "value": 4e3 / ms2,
                                   "unit": "images/s", "ms_per_step": ms2, "steps": 10, "dtype": "bf16"}
            if not a.no_roofline:                                  # its own roofline / whole_path blocks (row g, "thin (measurement)")
                r2, t2, p2 = roofline_of(x2, "bf16", (1080, 1920))
                other["configs[2]"].update({"roofline": r2, "top_kernels": t2,
                                            "whole_path": whole_path_of("bf16", x2.shape[2] * x2.shape[3], 4, 4e3 / ms2, ms2 * 1e-3, p2)})
            del x2
        finally:
            fdn_hip.set_storage_dtype("f32")
        g4 = {}

        def lp_graph(t):
            if "g" not in g4:
                g4["x"] = t.clone()
                with torch.no_grad():
                    for _ in range(2):
                        lp(g4["x"])
                    torch.cuda.synchronize()
                    g4["g"] = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g4["g"], capture_error_mode="thread_local"):
                        g4["out"] = lp(g4["x"])
            g4["x"].copy_(t)
            g4["g"].replay()
            return g4["out"]
        ms4 = side(lp_graph, x, steps=20)
        other["configs[4]"] = {"workload": f"LPNet_lolblur forward {a.width}x{a.height} (padded {W}x{H}) batch={B}, real weights", "value": B * 1e3 / ms4,
                               "unit": "images/s", "ms_per_step": ms4, "steps": 20, "dtype": "f32"}
        torch.cuda.empty_cache()

    if rank == 0:
        imgs = world * B * a.steps
        ips = imgs / dt
        P = H * W
        headline = a.variant == "lolblur" and (a.height, a.width, B) == (720, 1280, 8) and a.dtype == "f32"
        cfg3 = a.variant == "lolblur" and (a.height, a.width, B) == (1080, 1920, 4) and a.dtype == "bf16"
        if a.dry_run:
            metric, workload = "DRY RUN (CPU/gloo rehearsal, measures nothing)", "dry run"
        elif a.config == "lpnet":
            metric = f"images/sec, I_predict_net (LPNet alone) {a.width}x{a.height} bs={B} fp32 [BASELINE.json configs[4], not the headline metric]"
            workload = f"BASELINE.json configs[4]: LPNet_lolblur forward {a.width}x{a.height} (padded {W}x{H}) batch={B} per GPU fp32, real LPNet_lolblur weights"
        elif a.variant == "lolblur":
            metric = f"images/sec, FDN (LPNet->FDN forward) {a.width}x{a.height} bs={B} {'fp32' if a.dtype == 'f32' else 'bf16-storage'}"
            workload = (("BASELINE.json configs[1]: " if headline else "BASELINE.json configs[2]: " if cfg3 else "")
                        + f"FDN {a.width}x{a.height} (padded {W}x{H}) batch={B} per GPU "
                        + ("fp32" if a.dtype == "f32" else "bf16 storage of block-internal activations, fp32 math"))
            if world > 1 and headline:
                workload += f"; BASELINE.json configs[3] shape: global batch {world * B} sharded {world} ways"
        else:
            metric = f"images/sec, FDN_lolv1 (LPNet->FDN_lolv1 forward) {a.width}x{a.height} bs={B} fp32 [not the headline metric]"
            workload = f"FDN_lolv1 (dim 24) {a.width}x{a.height} (padded {W}x{H}) batch={B} per GPU fp32"
        line = {
            "metric": metric, "value": ips, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "ms_per_image": 1e3 / ips * world, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": workload, "global_batch": world * B, "parallelism": f"batch-shard x{world}",
                       "weights": "synthetic (tamed 0.03) FDN + real LPNet", "scatter_gather_timed": bool(sg), "hip_streams": a.streams,
                       "rccl_ranks": dist.get_world_size() if dist is not None else 1, "hip_graph": bool(a.graph), "fdsa_full": bool(a.fdsa_full), "fdsa_pair": bool(a.fdsa_pair), "no_tail_pin": bool(a.no_tail_pin), "tail_pin_l2": bool(a.tail_pin_l2), "narrow_pipe": bool(a.narrow_pipe), "unfused_mlps": bool(a.unfused_mlps), "stats_launches": bool(a.stats_launches), "resample_upsample": bool(a.resample_upsample), "aff_resized": bool(a.aff_resized),
                       "host_issue_ms_per_step": host_issue_ms, "cpu_affinity_rank0": affinity},
            "whole_path": whole_path_of(a.dtype, P, B, ips / world, dt / a.steps, prof if a.config != "lpnet" else None),
            "roofline": roof, "top_kernels": top, "cpu_baseline": cpu, "other_configs": other,
            "parity": parity, "gpu_sensors": sensors.summary() if sensors is not None else None,
        }
        if a.config == "lpnet":
            line["whole_path"] = None                               # SURVEY 8(d)'s F_alg / B_alg are the LPNet -> FDN path's
        if dt_nosg is not None:
            line["without_collectives"] = {"value": imgs / dt_nosg, "ms_per_step": dt_nosg / a.steps * 1e3,
                                           "note": "the same K steps with every rank's shard already resident (no scatter / gather)"}
        if a.dry_run:
            line["dry_run"] = True
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
