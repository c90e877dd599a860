"""Does a level-3 stage run faster per image when the batch goes through it in slices small enough for the 256 MB memory-side cache?
tools/batch_split_probe.py: the decoder_level3 stage (10 blocks of FDSA + FDFFN at 128 x 184 x 320) at B = 8 as one batch and as
slices of 4 / 2 / 1, and the level-2 decoder stage the same way; ms per 8 images, median of 5."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
from basicsr.models.archs.FDN_arch import FDN

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = FDN().to(dev).eval()
fd = net.fdformer if hasattr(net, "fdformer") else [m for m in net.modules() if m.__class__.__name__ == "FDformer"][0]


def timed(stage, x, sl):
    outs = []
    def run():
        outs.clear()
        for i in range(0, x.shape[0], sl):
            outs.append(stage((x[i:i + sl].contiguous() if sl < x.shape[0] else x, None, None, None))[0])
    with torch.no_grad():
        run(); run()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
    return statistics.median(ts), torch.cat(outs)


for name, stage, shape in (("decoder_level3", fd.decoder_level3, (8, 128, 184, 320)), ("decoder_level2", fd.decoder_level2, (8, 64, 368, 640)),
                           ("decoder_level1", fd.decoder_level1, (8, 32, 736, 1280))):
    x = torch.randn(*shape, device=dev)
    ref = None
    for sl in (8, 4, 2, 1):
        t, y = timed(stage, x, sl)
        ref = y if ref is None else ref
        print(f"{name} slices of {sl}: {t:8.2f} ms per 8 images   bit-equal to one batch: {bool(torch.equal(y, ref))}", flush=True)
