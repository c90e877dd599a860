"""Column pass time against the row pitch of the spectrum: Wf = 641 (rows start 8 bytes further along each line) vs 640 / 656 (128-byte aligned rows)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import fdn_hip
if len(sys.argv) > 1 and sys.argv[1] != "default":          # another build of the library
    fdn_hip._LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from fdn_hip import ops
dev = torch.device("cuda:0")

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B = 8
for C, H, Wfs in ((32, 736, (641, 640, 656)), (64, 368, (321, 320, 336)), (128, 184, (161, 160, 176))):
    for Wf in Wfs:
        z = torch.randn(B, C, H, Wf, 2, device=dev)
        amp, pha = torch.rand(B, 3, H, Wf, device=dev), torch.rand(B, 3, H, Wf, device=dev)
        wxa, wxp = torch.randn(C, 3, device=dev), torch.randn(C, 3, device=dev)
        guide = ops.pack_guidance(amp, pha)
        import ctypes
        from fdn_hip import lib, check, stream
        def run():
            check(lib().fdn_fft_cols_fcaffn(ops._flat(z, "z"), ops._flat(guide, "g"), ops._flat(wxa, "a"), ops._flat(wxp, "p"), B, C, H, Wf, stream()), "cols")
        ms = timeit(run)
        print(f"C={C} H={H} Wf={Wf}: {ms:.3f} ms  {ms / Wf * 1e3:.3f} us/column-set  {2 * z.numel() * 4 / ms / 1e6:.0f} GB/s(alg)", flush=True)
