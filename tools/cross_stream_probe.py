"""The cross-stream corruption on MI355X / ROCm 7.2, reproduced without this library's matrix kernels:

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/victims.hip -o abx/libvictims.so      # (abx/ travels to the GPU box)
    python tools/cross_stream_probe.py [--net]

"Victims" - kernels whose result for a fixed input is known (this library's row FFT, channel LayerNorm, statistics, the FDFFN
patch kernel; torch.fft.rfft = rocFFT; a torch elementwise op) - run on one HIP stream while a "noise" kernel runs on another:
a loop of vector FMAs, of fp32 MFMAs, of bf16 MFMAs (compiler-generated, tools/micro/victims.hip: no LDS, no memory traffic in the
loop), or fdn_fdsa_fused.  Counted: victim launches whose output differs from the quiet run.  Measured
(profiles/r03_cross_stream_probe.txt): vector ALU / fp32 MFMA noise 0 of 90 everywhere; v_mfma_f32_32x32x16_bf16 noise 21-45 of 90
for the row FFT, rocFFT and LayerNorm hit as well; the differing outputs are whole rows (one wrong intermediate value of one row).
--net: the whole LPNet -> FDN forward with sub-batches on 3 streams against 1 stream (forward_streams)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
import torch
import fdn_hip
from fdn_hip import ops
if "--net" in sys.argv:
    sys.path.insert(0, ROOT)
    import bench
    from fdn_hip.pipeline import forward_streams
    net, lp = bench.build_models(torch.device("cuda:0"))
    for (B, H, W) in ((6, 256, 256), (3, 352, 640)):
        x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(1)).to("cuda:0")
        ref = forward_streams(net, lp, x, 1).clone(); torch.cuda.synchronize()
        for n in (1, 3):
            bad = 0
            for rep in range(8):
                o = forward_streams(net, lp, x, n); torch.cuda.synchronize()
                bad += 0 if torch.equal(o, ref) else 1
            print(f"LPNet -> FDN {(B, H, W)}: {n} stream(s) against the first 1-stream run: {bad} of 8 runs differ", flush=True)
    sys.exit(0)
V = ctypes.CDLL(os.path.join(ROOT, "abx", "libvictims.so"))
V.noise.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
r = lambda *s: torch.randn(*s, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
nout = torch.empty(2048 * 16 * 256, device=dev)
x2 = r(2, 32, 256, 256); g, b_ = r(32), r(32); z = r(2, 12, 256, 256)
zc = torch.view_as_real(torch.fft.rfft(z)).contiguous()
h32 = r(2, 86, 256, 256); w0, w2, fa, fp = r(86, 1, 3, 3), r(86, 1, 3, 3), r(86, 1, 1, 8, 5), r(86, 1, 1, 8, 5)
victims = [("layernorm_chan", lambda: ops.layernorm_chan(x2, g, b_)), ("rfft_rows", lambda: ops.rfft_rows(z)),
           ("chan_stats", lambda: ops.chan_stats(x2)), ("fdffn_mid", lambda: ops.fdffn_mid(h32, w0, w2, fa, fp)),
           ("torch.fft.rfft", lambda: torch.view_as_real(torch.fft.rfft(z))), ("torch add", lambda: x2 * 1.5 + 2.0)]
def nz(mode):
    def f():
        assert V.noise(mode, ctypes.c_void_p(nout.data_ptr()), 2048, 40, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    return f
x64 = r(2, 64, 128, 128); st64 = ops.chan_stats(x64); g64, b64 = r(64), r(64); wh64 = r(304, 64) / 8; dw64, fw64 = r(304, 1, 3, 3), r(76, 1, 1, 8, 5)
wpk64 = ops.fdsa_pack(wh64, g64, b64)
noises = [("vector ALU only", nz(4)), ("fp32 mfma", nz(5)), ("bf16 mfma 16x16x32", nz(6)), ("mfma independent", nz(0)), ("mfma chain -> LDS", nz(1)), ("mfma chain -> memory", nz(2)), ("mfma chain only", nz(3)),
          ("fdsa_fused<64>", lambda: ops.fdsa_fused(x64, st64, wpk64, dw64, fw64))]
for nname, noise in noises:
    for vname, fn in victims:
        ref = fn(); torch.cuda.synchronize()
        bad = 0
        for rep in range(15):
            with torch.cuda.stream(s2):
                for _ in range(4): noise()
            with torch.cuda.stream(s1):
                outs = [fn() for _ in range(6)]
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
        print(f"{vname} beside {nname}: mismatches {bad} of 90", flush=True)
