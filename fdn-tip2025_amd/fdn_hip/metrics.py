"""Validation metrics on the GPU: the reference's basicsr/metrics/psnr_ssim.py `calculate_psnr` (:8-73) and `calculate_ssim`
(:243-328) with all their branches - the default 3-D Gaussian SSIM (`ssim3d=True`, :163-197), the 2-D one (`ssim3d=False`, `_ssim`
:84-116) and the Y-channel variants (`test_y_channel=True`: `to_y_channel` + `_ssim_cly` :199-240) - for (C,H,W) or (1,C,H,W) float32
ROCm tensors.  With `test_y_channel` the channels must be in B, G, R order and the range [0, 255], as the reference's callers pass
them (`tensor2img(..., rgb2bgr=True)`).  No CPU fallback."""
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


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _sse_max(a, b):
    acc = torch.zeros(2, dtype=torch.float64, device=a.device)
    check(lib().fdn_sse_max(_ptr(a), _ptr(b), ctypes.c_long(a.numel()), _ptr(acc), stream()), "fdn_sse_max")
    sse, mx = acc.tolist()
    return sse, mx


def to_y_channel(img_bgr):
    """(3,H,W) B,G,R in [0, 255] -> (1,H,W) Y in [16, 235] (metric_util.py:34-47)."""
    if img_bgr.shape[0] != 3:
        raise FdnHipError("to_y_channel needs a 3-channel (B, G, R) image")
    _, H, W = img_bgr.shape
    out = torch.empty((1, H, W), dtype=torch.float32, device=img_bgr.device)
    check(lib().fdn_y_channel(_ptr(img_bgr), _ptr(out), H, W, stream()), "fdn_y_channel")
    return out


def calculate_psnr(img1, img2, crop_border=0, test_y_channel=False):
    a, b = _prep(img1, img2, crop_border)
    if test_y_channel:                                                                                 # :55-57
        a, b = to_y_channel(a), to_y_channel(b)
    sse, mx = _sse_max(a, b)
    mse = sse / a.numel()
    if mse == 0:
        return float("inf")
    max_value = 1.0 if mx <= 1 else 255.0                                                              # :60
    return 20.0 * math.log10(max_value / math.sqrt(mse))


def _ssim2d(a, b, max_value, replicate_no_crop):
    C, H, W = a.shape
    ws = torch.empty(5 * a.numel(), dtype=torch.float64, device=a.device)
    acc = torch.zeros(1, dtype=torch.float64, device=a.device)
    check(lib().fdn_ssim2d(_ptr(a), _ptr(b), C, H, W, ctypes.c_float(max_value), int(replicate_no_crop), _ptr(ws), _ptr(acc), stream()),
          "fdn_ssim2d")
    count = C * H * W if replicate_no_crop else C * (H - 10) * (W - 10)
    return float(acc.item()) / count


def calculate_ssim(img1, img2, crop_border=0, test_y_channel=False, ssim3d=True):
    a, b = _prep(img1, img2, crop_border)
    if test_y_channel:                                                                                 # :275-278: Y plane, _ssim_cly
        return _ssim2d(to_y_channel(a), to_y_channel(b), 255.0, True)
    C, H, W = a.shape
    mx = float(a.max().item())
    max_value = 1.0 if mx <= 1 else 255.0                                                              # :286
    if not ssim3d:
        return _ssim2d(a, b, max_value, False)                                                         # _ssim, :84-116
    ws = torch.empty(10 * a.numel(), dtype=torch.float32, device=a.device)
    acc = torch.zeros(1, dtype=torch.float64, device=a.device)
    check(lib().fdn_ssim3d(_ptr(a), _ptr(b), C, H, W, ctypes.c_float(max_value), _ptr(ws), _ptr(acc), stream()), "fdn_ssim3d")
    return float(acc.item()) / a.numel()
