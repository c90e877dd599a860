"""The platform finding that shapes the execution model: a kernel of ANOTHER HIP stream that shares the GPU with
a loop of v_mfma_f32_32x32x16_bf16 returns wrong rows.  Kept as a test so that the day the platform stops doing it shows up as XPASS
(and sub-batch streams can come back); the product path runs one stream per GPU and does not depend on the outcome."""
import ctypes
import os
import shutil
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))


@pytest.fixture(scope="module")
def noise_lib(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    so = str(tmp_path_factory.mktemp("victims") / "libvictims.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(ROOT, "tools", "micro", "victims.hip"), "-o", so],
                   check=True)
    lib = ctypes.CDLL(so)
    lib.noise.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


def _mismatches(noise_lib, mode):
    from fdn_hip import ops
    dev = torch.device("cuda:0")
    z = torch.randn(2, 12, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    out = torch.empty(2048 * 16 * 256, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ref = ops.rfft_rows(z)
    torch.cuda.synchronize()
    bad = 0
    for _ in range(10):
        with torch.cuda.stream(s2):
            for _ in range(4):
                assert noise_lib.noise(mode, ctypes.c_void_p(out.data_ptr()), 2048, 40, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        with torch.cuda.stream(s1):
            outs = [ops.rfft_rows(z) for _ in range(6)]
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
    return bad


def test_row_fft_beside_vector_alu_and_fp32_mfma_loops(noise_lib):
    """Control: neighbours made of vector FMAs or fp32 MFMAs on another stream leave the row FFT's 60 launches bit-identical."""
    bad = _mismatches(noise_lib, 4), _mismatches(noise_lib, 5)
    if bad != (0, 0):       # never seen (0 of 540 launches per kind in tools/cross_stream_probe.py); reported, not fatal: the product runs one stream
        pytest.xfail(f"co-resident vector-ALU / fp32-MFMA loops disturbed the row FFT in {bad} of 60 launches each")


@pytest.mark.xfail(strict=False, reason="MI355X / ROCm 7.2: 20-45 of 90 launches of a kernel on another stream come back with a wrong row while a "
                                        "compiler-generated loop of v_mfma_f32_32x32x16_bf16 shares the GPU (profiles/r03_cross_stream_probe.txt)")
def test_row_fft_beside_bf16_mfma_loop(noise_lib):
    assert _mismatches(noise_lib, 0) == 0
