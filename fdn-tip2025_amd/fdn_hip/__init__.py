"""ctypes binding of libfdn_hip.so (the C ABI declared in include/fdn_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every compute step
of the FDN path is a call into the library.  There is NO CPU or eager fallback: if the shared
library is missing, or a tensor is not a contiguous fp32 ROCm tensor, the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfdn_hip.so")      # (measurement tools that A/B another build of the same ABI set this before lib())
_lib = None

ACT_NONE, ACT_LEAKY, ACT_RELU, ACT_SIGMOID, ACT_GELU = 0, 1, 2, 3, 4
PRO_NONE, PRO_LN, PRO_LN3_GATE, PRO_LN_MULADD = 0, 1, 2, 3
EPI_NONE, EPI_RES, EPI_MULADD = 0, 1, 2

c_fp = ctypes.c_void_p


class Conv1x1Desc(ctypes.Structure):
    _fields_ = [
        ("x", c_fp * 3), ("xbs", ctypes.c_long * 3), ("kseg", ctypes.c_int * 3),
        ("w", c_fp), ("bias", c_fp), ("out", c_fp), ("obs", ctypes.c_long),
        ("B", ctypes.c_int), ("K", ctypes.c_int), ("N", ctypes.c_int), ("P", ctypes.c_int),
        ("pro", ctypes.c_int), ("ln_group", ctypes.c_int),
        ("stats", c_fp), ("gamma", c_fp), ("beta", c_fp), ("xb", c_fp), ("xbbs", ctypes.c_long),
        ("act", ctypes.c_int), ("epi", ctypes.c_int),
        ("res", c_fp), ("rbs", ctypes.c_long), ("mul", c_fp), ("add", c_fp), ("mbs", ctypes.c_long),
        ("vec4", ctypes.c_int), ("stats_out", c_fp), ("x_bf16", ctypes.c_int), ("out_bf16", ctypes.c_int), ("wpk", c_fp),
    ]


class FdnHipError(RuntimeError):
    pass


ABI_VERSION = 15         # include/fdn_hip.h: bumped on any signature change


def lib_path():
    return _LIB_PATH


def lib():
    """Load libfdn_hip.so once.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(_LIB_PATH):
            raise ImportError(
                f"{_LIB_PATH} not found: build it with fdn-tip2025_amd/build.sh (hipcc --offload-arch=gfx950). "
                "The FDN path has no CPU fallback.")
        _lib = ctypes.CDLL(_LIB_PATH)
        _declare(_lib)
        if _lib.fdn_abi_version() != ABI_VERSION:
            v = _lib.fdn_abi_version()
            _lib = None
            raise ImportError(f"{_LIB_PATH} has ABI version {v}, this binding needs {ABI_VERSION}: rebuild with build.sh")
    return _lib


def _declare(l):
    """argtypes / restype of every entry point (a wrong argument count or kind raises here instead of corrupting the call)."""
    from ._abi import PROTOTYPES
    kinds = {"P": ctypes.c_void_p, "I": ctypes.c_int, "L": ctypes.c_long, "F": ctypes.c_float, "DESC": ctypes.POINTER(Conv1x1Desc)}
    for name, (ret, sig) in PROTOTYPES.items():
        f = getattr(l, name)                       # AttributeError: the library lacks a symbol the header declares
        f.restype = ctypes.c_char_p if ret == "S" else ctypes.c_long if ret == "L" else ctypes.c_int
        f.argtypes = [kinds[k] for k in sig]


def check(code, what):
    if code != 0:
        raise FdnHipError(f"{what}: {lib().fdn_error_string(code).decode()} (code {code})")


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Storage format of the block-internal activations that travel between kernels:
#   "f32"  - everything fp32, the reference's arithmetic (BASELINE.json configs[1]);
#   "bf16" - the wide intermediates inside FDSA / FDFFN (out1|out2|out3|v_value, the FDFFN hidden tensors) are STORED as
#            bf16; every product, accumulation, FFT, LayerNorm statistic and the residual stream stay fp32 (configs[2]).
_storage = "f32"


def set_storage_dtype(name):
    global _storage
    if name not in ("f32", "bf16"):
        raise ValueError(f"storage dtype must be 'f32' or 'bf16', got {name!r}")
    _storage = name


def storage_dtype():
    return _storage


_matrix_pipe = "bf16"


def set_matrix_pipe(name):
    """"bf16" (default): matrix products as six bf16 products of exactly split fp32 operands on v_mfma_f32_32x32x16_bf16;
    "f32": the fp32-MFMA forms only (diagnostic: bisecting the cross-stream finding; same results to rounding) - what it
    promises is checkable: `bf16_mfma_launches()` stands still;
    "bf16-narrow": the default of ABI 10 - as "bf16" but the level-2 FDSA tail keeps its fp32-MFMA form (kept for A/B runs)."""
    global _matrix_pipe
    if name not in ("bf16", "f32", "bf16-narrow"):
        raise ValueError(f"matrix pipe must be 'bf16', 'f32' or 'bf16-narrow', got {name!r}")
    check(lib().fdn_set_matrix_pipe({"bf16": 0, "f32": 1, "bf16-narrow": 2}[name]), "fdn_set_matrix_pipe")
    _matrix_pipe = name


def matrix_pipe():
    """"bf16" or "f32" ("bf16-narrow" reports as "bf16": the routing of the host mirror is the same)."""
    return "bf16" if _matrix_pipe == "bf16-narrow" else _matrix_pipe


def matrix_pipe_mode():
    """the name last given to set_matrix_pipe (part of the capture key of pipeline.GraphedForward / GraphedStep)"""
    return _matrix_pipe


def bf16_mfma_launches():
    """number of bf16-MFMA kernel launches this process has enqueued (fdn_bf16_mfma_launches, ABI 11)"""
    return int(lib().fdn_bf16_mfma_launches())


def dev(t, what="tensor"):
    """Validate a tensor handed to the library and return its device pointer."""
    if t is None:
        return None
    if not t.is_cuda:
        raise FdnHipError(f"{what} must live on a ROCm device (got {t.device}); the FDN path has no CPU fallback")
    if t.dtype != torch.float32:
        raise FdnHipError(f"{what} must be float32 (got {t.dtype})")
    if not t.is_contiguous():
        raise FdnHipError(f"{what} must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


from . import ops  # noqa: E402,F401
