"""Validation metrics on the GPU: the reference's defaults, basicsr/metrics/psnr_ssim.py `calculate_psnr` (:8-73) and the
`ssim3d=True` path of `calculate_ssim` (:163-197, :215-290), for (C,H,W) or (1,C,H,W) float32 ROCm tensors in RGB/any
channel order (both metrics are symmetric in the channels except for the 3-D window, which sees them in the given order,
as the reference does).  `test_y_channel` is not covered (off in the LOL-Blur options).  No CPU fallback."""
import ctypes
import math

import torch

from . import FdnHipError, check, lib, stream


def _prep(img1, img2, crop_border):
    if img1.shape != img2.shape:
        raise FdnHipError(f"Image shapes are different: {tuple(img1.shape)}, {tuple(img2.shape)}.")      # psnr_ssim.py:30
    out = []
    for t in (img1, img2):
        if t.dim() == 4:
            t = t.squeeze(0)
        if t.dim() != 3 or not t.is_cuda or t.dtype != torch.float32:
            raise FdnHipError("metrics take float32 ROCm tensors of shape (C,H,W) or (1,C,H,W)")
        if crop_border:
            t = t[:, crop_border:-crop_border, crop_border:-crop_border]
        out.append(t.contiguous())
    return out


def _sse_max(a, b):
    acc = torch.zeros(2, dtype=torch.float64, device=a.device)
    check(lib().fdn_sse_max(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_long(a.numel()),
                            ctypes.c_void_p(acc.data_ptr()), stream()), "fdn_sse_max")
    sse, mx = acc.tolist()
    return sse, mx


def calculate_psnr(img1, img2, crop_border=0):
    a, b = _prep(img1, img2, crop_border)
    sse, mx = _sse_max(a, b)
    mse = sse / a.numel()
    if mse == 0:
        return float("inf")
    max_value = 1.0 if mx <= 1 else 255.0                                                              # :60
    return 20.0 * math.log10(max_value / math.sqrt(mse))


def calculate_ssim(img1, img2, crop_border=0):
    a, b = _prep(img1, img2, crop_border)
    C, H, W = a.shape
    mx = float(a.max().item())
    max_value = 1.0 if mx <= 1 else 255.0                                                              # :268
    ws = torch.empty(10 * a.numel(), dtype=torch.float32, device=a.device)
    acc = torch.zeros(1, dtype=torch.float64, device=a.device)
    check(lib().fdn_ssim3d(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), C, H, W, ctypes.c_float(max_value),
                           ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(acc.data_ptr()), stream()), "fdn_ssim3d")
    return float(acc.item()) / a.numel()
