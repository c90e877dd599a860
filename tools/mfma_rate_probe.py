"""Issue rate of the bf16 matrix instructions (independent accumulators, no memory traffic): tools/micro/victims.hip noise kernels.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/victims.hip -o abx/libvictims.so; python tools/mfma_rate_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
V = ctypes.CDLL(os.path.join(ROOT, "abx", "libvictims.so"))
V.noise.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.empty(4096 * 16 * 256, device="cuda:0")
def run(mode, blocks, iters):
    assert V.noise(mode, ctypes.c_void_p(out.data_ptr()), blocks, iters, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
def timeit(mode, blocks, iters):
    run(mode, blocks, iters); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(mode, blocks, iters); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3
cus = torch.cuda.get_device_properties(0).multi_processor_count
for name, mode, per_iter, flops in (("v_mfma_f32_32x32x16_bf16, 4 accumulators", 0, 4, 32 * 32 * 16 * 2), ("v_mfma_f32_16x16x32_bf16, 4 accumulators", 6, 8, 16 * 16 * 32 * 2),
                                    ("v_mfma_f32_32x32x16_bf16, one dependent chain of 6", 3, 6, 32 * 32 * 16 * 2), ("v_mfma_f32_32x32x2_f32, 2 accumulators", 5, 4, 32 * 32 * 2 * 2)):
    for wg_per_cu in (1, 2):
        blocks, iters = cus * wg_per_cu, 20000
        t = timeit(mode, blocks, iters)
        n = blocks * 4 * iters * per_iter                       # wave-level instructions
        print(f"{name}, {wg_per_cu * 4} waves per CU: {n * flops / t / 1e12:.0f} TFLOP/s, {t / (iters * per_iter) * 2.4e9 * (1 if wg_per_cu == 1 else 0.5):.1f} cycles (2.4 GHz) per instruction and SIMD", flush=True)
