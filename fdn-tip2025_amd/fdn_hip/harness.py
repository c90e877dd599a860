"""The step either side of the LPNet -> FDN forward, on the GPU.

Mirrors what inference_fdn_lolblur.py:47-75 does on the host with cv2 / numpy:

    img = cv2.imread(p).astype(np.float32) / 255.          # uint8 BGR HWC -> fp32
    img = img2tensor(img, bgr2rgb=True)[None]               # RGB CHW, batch 1        (img_util.py:9-33)
    img = F.pad(img, (0, w_n, 0, h_n), 'reflect')           # bottom/right to the x32 grid
    ratio = LPNet(img); result = FDN(img, ratio_i=ratio)[0]
    out = tensor2img(result[:, :, :h, :w], rgb2bgr=True)    # clamp, *255, round, uint8 BGR HWC (img_util.py:36-98)

here as two HIP kernels (fdn_pre_u8 / fdn_post_u8) around the drop-in modules, batched: B images of one size go
through one forward.  No CPU fallback: the uint8 tensors must live on the ROCm device.
"""
import ctypes
import torch

from . import lib, check, stream, FdnHipError


def _u8(t, what):
    if not t.is_cuda or t.dtype != torch.uint8 or not t.is_contiguous():
        raise FdnHipError(f"{what} must be a contiguous uint8 ROCm tensor")
    return ctypes.c_void_p(t.data_ptr())


def padded_size(h, w, multiple=32):
    """inference_fdn_lolblur.py:57-59: pad bottom/right up to the next multiple of 32."""
    return h + (multiple - h % multiple) % multiple, w + (multiple - w % multiple) % multiple


def preprocess(img_u8, bgr=True):
    """uint8 [B,h,w,3] (or [h,w,3]) on the GPU -> (fp32 [B,3,H,W] reflect-padded RGB in [0,1], h, w)."""
    if img_u8.dim() == 3:
        img_u8 = img_u8.unsqueeze(0)
    if img_u8.dim() != 4 or img_u8.shape[-1] != 3:
        raise FdnHipError(f"expected uint8 images [B,h,w,3], got {tuple(img_u8.shape)}")
    B, h, w, _ = img_u8.shape
    H, W = padded_size(h, w)
    if H - h >= h or W - w >= w:
        raise FdnHipError(f"reflect padding {h}x{w} -> {H}x{W} needs pad < size (F.pad raises the same way)")
    out = torch.empty((B, 3, H, W), device=img_u8.device, dtype=torch.float32)
    check(lib().fdn_pre_u8(_u8(img_u8, "img"), ctypes.c_void_p(out.data_ptr()), B, h, w, H, W, int(bool(bgr)), stream()),
          "fdn_pre_u8")
    return out, h, w


def postprocess(result, h, w, bgr=True):
    """fp32 [B,3,H,W] -> uint8 [B,h,w,3]: crop, clamp(0,1), *255, round half-to-even (numpy .round())."""
    if not result.is_cuda or result.dtype != torch.float32 or not result.is_contiguous() or result.dim() != 4:
        raise FdnHipError("result must be a contiguous float32 ROCm tensor [B,3,H,W]")
    B, C, H, W = result.shape
    if C != 3 or h > H or w > W:
        raise FdnHipError(f"cannot crop {h}x{w} out of {tuple(result.shape)}")
    out = torch.empty((B, h, w, 3), device=result.device, dtype=torch.uint8)
    check(lib().fdn_post_u8(ctypes.c_void_p(result.data_ptr()), _u8(out, "out"), B, h, w, H, W, int(bool(bgr)), stream()),
          "fdn_post_u8")
    return out


def lolv1_ratio(x, lp_ratio):
    """inference_fdn_lolv1.py:57-61: ratio_i = mean(Grayscale(padded input)) / LPNet(padded input).  The plane means
    come from fdn_global_avgpool; Grayscale is linear (0.2989 R + 0.587 G + 0.114 B), so its mean is the same
    combination of the three plane means ([B,3] values, combined on the device)."""
    from . import ops
    m = ops.global_avgpool(x).reshape(x.shape[0], 3)
    gray = 0.2989 * m[:, 0:1] + 0.587 * m[:, 1:2] + 0.114 * m[:, 2:3]
    return gray / lp_ratio


@torch.no_grad()
def enhance_u8(net, lpnet, img_u8, bgr=True, ratio_mode="lolblur", ratio=None):
    """uint8 in -> uint8 out through LPNet -> FDN (the body of the reference's per-image loop, batched).
    ratio_mode: "lolblur" feeds LPNet's prediction (inference_fdn_lolblur.py:69-71), "lolv1" feeds
    mean(gray)/prediction (inference_fdn_lolv1.py:57-62), "fixed" feeds the caller's `ratio` [B,1] and skips LPNet - the
    ratio sweep of inference_fdn_multi_r.py:78-84 (`ratio = ratio / ratio * i`)."""
    if ratio_mode not in ("lolblur", "lolv1", "fixed"):
        raise ValueError(f"ratio_mode {ratio_mode!r}")
    x, h, w = preprocess(img_u8, bgr=bgr)
    if ratio_mode == "fixed":
        if ratio is None or tuple(ratio.shape) != (x.shape[0], 1):
            raise FdnHipError(f"ratio_mode 'fixed' needs ratio [B,1] for B = {x.shape[0]}")
        result = net(x, ratio_i=ratio.to(device=x.device, dtype=torch.float32).contiguous(), device=x.device)[0]
    elif ratio_mode == "lolblur":
        from .pipeline import run
        result = run(net, lpnet, x)                # hipGraph replay for small frames, the eager forward otherwise
    else:
        ratio = lolv1_ratio(x, lpnet(x))
        result = net(x, ratio_i=ratio, device=x.device)[0]
    return postprocess(result.contiguous(), h, w, bgr=bgr)
