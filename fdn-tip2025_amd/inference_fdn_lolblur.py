"""LOL-Blur inference driver on the HIP path: the role of the reference's inference_fdn_lolblur.py:1-75
(load LPNet + FDN checkpoints, walk a directory of low-light blurry frames, write the enhanced frames), with
the paths as arguments instead of constants and the per-image host work moved to the GPU:

    decode (PIL, worker threads)  ->  uint8 HWC on the GPU  ->  fdn_pre_u8  ->  LPNet -> FDN  ->  fdn_post_u8  ->  encode

Images of equal size are batched (the reference runs batch 1; every op of the path is per-sample).
Needs a ROCm GPU and the built libfdn_hip.so; there is no CPU fallback.

    python inference_fdn_lolblur.py --fdn FDN_lolblur.pth --lpnet LPNet_lolblur.pth --input 'frames/*.png' --output out/
"""
import argparse
import glob
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def read_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("RGB"), dtype=np.uint8)           # (a copy: PIL hands out a read-only buffer)


def write_rgb(path, arr):
    from PIL import Image
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    Image.fromarray(arr, mode="RGB").save(path)


def load_params(path):
    sd = torch.load(path, map_location="cpu")
    return sd["params"] if isinstance(sd, dict) and "params" in sd else sd     # inference_fdn_lolblur.py:28,31


def run_driver(doc, build_models, ratio_mode="lolblur", fdn_keys="FDN checkpoint ({'params': state_dict}, 1503 keys)"):
    """The directory walk shared by the LOL-Blur and LOL-v1 drivers: build_models() -> (FDN-like module, LPNet module) on the CPU,
    ratio_mode as fdn_hip.harness.enhance_u8 takes it."""
    ap = argparse.ArgumentParser(description=doc, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--fdn", required=True, help=fdn_keys)
    ap.add_argument("--lpnet", required=True, help="LPNet checkpoint (292 keys)")
    ap.add_argument("--input", required=True, help="glob of input frames")
    ap.add_argument("--output", required=True, help="output directory")
    ap.add_argument("--input-root", default=None,
                    help="directory the output tree mirrors: a frame <input-root>/0256/0089.png is written to <output>/0256/0089.png "
                         "(the reference keeps the LOL-Blur sequence folders the same way, inference_fdn_lolblur.py:44-45,73); "
                         "default: the common parent directory of all input frames")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args()

    from fdn_hip.harness import enhance_u8

    dev = torch.device(a.device)
    torch.cuda.set_device(dev)
    net, lp = build_models()
    net = net.to(dev).eval()
    net.load_state_dict(load_params(a.fdn), strict=True)
    lp = lp.to(dev).eval()
    lp.load_state_dict(load_params(a.lpnet), strict=True)

    paths = sorted(glob.glob(a.input))
    if not paths:
        raise SystemExit(f"no input frames match {a.input}")
    root = a.input_root or os.path.commonpath([os.path.dirname(os.path.abspath(p)) for p in paths])
    dest = {p: os.path.join(a.output, os.path.relpath(os.path.abspath(p), root)) for p in paths}
    if any(d.startswith("..") for d in (os.path.relpath(v, a.output) for v in dest.values())):
        raise SystemExit(f"--input-root {root} does not contain every input frame")
    if len(set(dest.values())) != len(paths):                              # never let two frames race for one output file
        raise SystemExit("two input frames map to the same output path; pass an --input-root above both")
    with ThreadPoolExecutor(max_workers=4) as pool:
        decoded = pool.map(read_rgb, paths)                               # decode runs ahead of the GPU
        pending, writers = [], []

        def flush():
            if not pending:
                return
            batch = torch.from_numpy(np.stack([im for _, im in pending])).to(dev, non_blocking=True)
            out = enhance_u8(net, lp, batch, bgr=False, ratio_mode=ratio_mode).cpu().numpy()
            for (p, _), o in zip(pending, out):
                writers.append(pool.submit(write_rgb, dest[p], o))
            pending.clear()

        for p, im in zip(paths, decoded):
            if pending and (pending[0][1].shape != im.shape or len(pending) == a.batch):
                flush()
            pending.append((p, im))
        flush()
        for w in writers:
            w.result()
    print(f"{len(paths)} frames -> {a.output}")


def main():
    def build():
        from basicsr.models.archs.FDN_arch import FDN
        from basicsr.models.archs.LPNet_arch import I_predict_net
        return FDN(), I_predict_net()
    run_driver(__doc__, build)


if __name__ == "__main__":
    main()
