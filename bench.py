#!/usr/bin/env python3
"""FDN inference bench on MI355X:  images/s of the whole LPNet -> FDN forward (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: config 2 of BASELINE.json, 8 synthetic images of
1280x720 reflect-padded to 736x1280, fp32, already resident in HBM.  N GPUs = N independent shards
of 8 images (the batch shards embarrassingly) -> "scaling": "weak"; no data-path
collective unless --scatter-gather puts the RCCL scatter/gather of north_star in the timed region.
Weights are deterministic synthetic (the trained FDN checkpoint is absent from the reference
checkout); LPNet uses the real LPNet_lolblur weights when tests/golden has them.

Extra JSON objects: "roofline" for the dominant kernel family (timed live with HIP events on the
launch stream in one instrumented forward), "cpu_baseline" (the CPU oracle on a bounded sample).
"""
import argparse
import ctypes
import json
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


class KernelTimer:
    """Wrap every C-ABI entry point with HIP events on the launch stream (one instrumented forward)."""

    def __init__(self):
        import fdn_hip
        self.lib = fdn_hip.lib()
        self.records = []
        self.names = [n for n in ("fdn_conv1x1", "fdn_fdsa_out", "fdn_ffn_tail", "fdn_chan_stats", "fdn_layernorm_chan", "fdn_fdsa_core", "fdn_fdffn_mid",
                                  "fdn_dwconv_gate", "fdn_dwconv3x3", "fdn_img_mod_maps", "fdn_rfft_rows", "fdn_irfft_rows",
                                  "fdn_fft_cols_fcaffn", "fdn_fft_cols_fwd", "fdn_fft_cols_inv_polar", "fdn_conv2d",
                                  "fdn_conv_transpose4x4s2", "fdn_resample", "fdn_dw1x1_pad1", "fdn_avgpool3s2",
                                  "fdn_global_avgpool", "fdn_se_apply", "fdn_scale_batch", "fdn_gamma_curve")]
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
                flops = byts = 0
                if _n == "fdn_conv1x1":
                    d = a[0]._obj
                    flops = 2.0 * d.B * d.K * d.N * d.P
                    byts = 4.0 * d.B * d.P * (d.K + d.N + (d.N if d.epi == 1 else 0) + (2 * d.N if d.epi == 2 else 0)
                                              + (d.K // 3 if d.pro == 2 else 0) + (d.K if d.pro == 3 else 0))
                self.records.append((_n, e0, e1, flops, byts))
                return r
            setattr(self.lib, n, wrapped)
        return self

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(self.lib, n, f)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for n, e0, e1, fl, by in self.records:
            t = agg.setdefault(n, [0, 0.0, 0.0, 0.0])
            t[0] += 1
            t[1] += e0.elapsed_time(e1)
            t[2] += fl
            t[3] += by
        return agg


def pmc_traffic_per_launch(prefix):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes (profiles/*traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, KB * 1024; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for coalesced streaming reads on gfx950).  None when no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))["kernels"]
    n = sum(v["launches"] for k, v in d.items() if k.startswith(prefix))
    if n == 0:
        return None
    gb = sum(2.0 * v["fetch_GB_raw"] + v["write_GB"] for k, v in d.items() if k.startswith(prefix))
    return gb * 1e9 / n


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


def cpu_baseline(h=256, w=256):
    """Oracle (fp32, PyTorch CPU ops) on ONE h x w image; reported in 736x1280-equivalent images/s."""
    import fdn_oracle as O
    from weights import synth_state_dict
    from common import fdn_shapes, lpnet_weights
    cores = usable_cores()
    torch.set_num_threads(cores)
    P = synth_state_dict(fdn_shapes(), 7, prefix_key="fdn/", tame=0.03)
    PL = lpnet_weights()
    x = torch.rand(1, 3, h, w, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        t0 = time.perf_counter()
        r = O.lpnet_forward(PL, x)
        O.fdn_forward(P, x, r)
        dt = time.perf_counter() - t0
    px_ratio = (h * w) / (736.0 * 1280.0)
    return {"value": px_ratio / dt, "unit": "images/s (736x1280-equivalent)", "cores": cores, "kind": "port",
            "sample": f"1 image {h}x{w} fp32, LPNet+FDN oracle forward, {dt:.2f} s wall, scaled by pixel count"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the per-GPU batch is split over (measured best: 3)")
    ap.add_argument("--scatter-gather", action="store_true", help="time RCCL scatter of inputs / gather of outputs too")
    ap.add_argument("--variant", choices=("lolblur", "lolv1"), default="lolblur",
                    help="lolblur = FDN (BASELINE.json's metric); lolv1 = FDN_lolv1, dim 24 (SURVEY.md 8(f) rank 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the FDN path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("FDN_BENCH_FORCE_DIST") == "1":      # (the env switch exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)   # RCCL over xGMI

    from fdn_hip.pipeline import forward_streams
    net, lp = build_models(dev, a.variant)
    x = make_input(a.batch, a.height, a.width, dev, seed=1000 + rank)
    B, _, H, W = x.shape
    root_in = root_out = None
    if a.scatter_gather and dist is not None and rank == 0:
        root_in = [make_input(a.batch, a.height, a.width, dev, seed=1000 + r) for r in range(world)]
        root_out = [torch.empty_like(x) for _ in range(world)]

    def step():
        xin = x
        if a.scatter_gather and dist is not None:
            xin = torch.empty_like(x)
            dist.scatter(xin, root_in if rank == 0 else None, src=0)
        out = forward_streams(net, lp, xin, a.streams)          # LPNet -> FDN, batch halves on separate HIP streams
        if a.scatter_gather and dist is not None:
            dist.gather(out, root_out if rank == 0 else None, dst=0)
        return out

    for _ in range(a.warmup):
        step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        step()
    e1.record()
    barrier()
    wall = time.perf_counter() - t0
    dt = max(wall, e0.elapsed_time(e1) / 1e3)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    roof = None
    if rank == 0 and not a.no_roofline:
        agg = None
        for _ in range(2):                                          # the first instrumented pass also pays one-off host costs
            with KernelTimer() as kt:
                with torch.no_grad():
                    net(x, ratio_i=lp(x), device=dev)               # single stream: events bracket each launch
            cur = kt.summary()
            agg = cur if agg is None else {k: (v if v[1] <= agg.get(k, v)[1] else agg[k]) for k, v in cur.items()}
        total_ms = sum(v[1] for v in agg.values())
        dom = max(agg.items(), key=lambda kv: kv[1][1])
        name, (cnt, ms, fl, by) = dom
        if name == "fdn_conv1x1":
            ach = fl / (ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TF,
                    "traffic": pmc_traffic_per_launch("conv1x1"), "kernel": "conv1x1_kernel (fdn_conv1x1)", "launches": cnt, "avg_ms": ms / cnt,
                    "hbm_gbs_algorithmic": by / (ms * 1e-3) / 1e9, "share_of_step": ms / total_ms}
        else:
            roof = {"bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                    "kernel": name, "launches": cnt, "avg_ms": ms / cnt, "share_of_step": ms / total_ms}
        roof["by_kernel_ms"] = {k: round(v[1], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        imgs = world * B * a.steps
        ips = imgs / dt
        P = H * W
        line = {
            "metric": ("images/sec, FDN (LPNet->FDN forward) 1280x720 bs=8 fp32" if a.variant == "lolblur" else
                       f"images/sec, FDN_lolv1 (LPNet->FDN_lolv1 forward) {a.width}x{a.height} bs={B} fp32 [not the headline metric]"),
            "value": ips, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "ms_per_image": 1e3 / ips * world, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE.json configs[1]: FDN {a.width}x{a.height} (padded {W}x{H}) batch={B} per GPU fp32" if a.variant == "lolblur"
                                    else f"FDN_lolv1 (dim 24) {a.width}x{a.height} (padded {W}x{H}) batch={B} per GPU fp32"),
                       "global_batch": world * B, "parallelism": f"batch-shard x{world}", "weights": "synthetic (tamed 0.03) FDN + real LPNet",
                       "scatter_gather_timed": bool(a.scatter_gather), "hip_streams": a.streams},
            "whole_path": {"hbm_algorithmic_frac": B_ALG_ELEMS_PER_PX * 4.0 * P * (ips / world) / (PEAK_HBM_GBS * 1e9),
                           "mfma_f32_frac": F_ALG_PER_PX * P * (ips / world) / (PEAK_F32_MFMA_TF * 1e12)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
