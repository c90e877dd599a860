"""Tiled inference on the GPU: the reference's mechanism for images larger than a forward can take
(ImageRestorationModel.grids / grids_inverse, basicsr/models/image_restoration_model.py:261-339, enabled by
`val.grids` with `crop_size_h/w`, :737-743).  Tiles overlap by an adaptive step, run through the network as a batch
and are averaged where they overlap - the result differs from an untiled forward (FDN's FFTs are global), so this is a
feature of the reference being mirrored, not an optimisation.  scale = 1 (restoration, `opt['scale']`).
"""
import ctypes
import math

import torch

from . import FdnHipError, check, lib, stream


def tile_origins(h, w, crop_h, crop_w):
    """Origins (i, j) of the tiles, image_restoration_model.py:278-309."""
    if crop_h > h or crop_w > w or crop_h <= 0 or crop_w <= 0:
        raise FdnHipError(f"crop {crop_h}x{crop_w} does not fit the image {h}x{w}")
    num_row, num_col = (h - 1) // crop_h + 1, (w - 1) // crop_w + 1
    step_j = crop_w if num_col == 1 else math.ceil((w - crop_w) / (num_col - 1) - 1e-8)
    step_i = crop_h if num_row == 1 else math.ceil((h - crop_h) / (num_row - 1) - 1e-8)
    idx = []
    i, last_i = 0, False
    while i < h and not last_i:
        j = 0
        if i + crop_h >= h:
            i, last_i = h - crop_h, True
        last_j = False
        while j < w and not last_j:
            if j + crop_w >= w:
                j, last_j = w - crop_w, True
            idx.append((i, j))
            j += step_j
        i += step_i
    return idx


def _f32(t, what):
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise FdnHipError(f"{what} must be a contiguous float32 ROCm tensor")
    return ctypes.c_void_p(t.data_ptr())


def split(x, crop_h, crop_w):
    """grids(): x (1,C,h,w) -> (tiles (T,C,crop_h,crop_w), origins tensor int32 [T,2] on the device)."""
    if x.dim() != 4 or x.shape[0] != 1:
        raise FdnHipError("tiled inference takes one image at a time (the reference asserts b == 1, :265)")
    _, C, h, w = x.shape
    idx = tile_origins(h, w, crop_h, crop_w)
    ij = torch.tensor(idx, dtype=torch.int32, device=x.device)
    tiles = torch.empty((len(idx), C, crop_h, crop_w), device=x.device, dtype=torch.float32)
    check(lib().fdn_tiles_gather(_f32(x, "x"), _f32(tiles, "tiles"), ctypes.c_void_p(ij.data_ptr()), len(idx), C, h, w, crop_h,
                                 crop_w, stream()), "fdn_tiles_gather")
    return tiles, ij


def merge(outs, ij, h, w):
    """grids_inverse(): tiles (T,C,ch,cw) + origins -> (1,C,h,w), overlaps averaged."""
    T, C, ch, cw = outs.shape
    out = torch.empty((1, C, h, w), device=outs.device, dtype=torch.float32)
    check(lib().fdn_tiles_merge(_f32(outs, "outs"), _f32(out, "out"), ctypes.c_void_p(ij.data_ptr()), T, C, h, w, ch, cw, stream()),
          "fdn_tiles_merge")
    return out


@torch.no_grad()
def forward_tiled(net, lpnet, x, crop_h, crop_w, batch=8):
    """LPNet -> FDN on overlapping tiles of one padded image (crop sizes multiples of 32), merged like the reference."""
    if crop_h % 32 or crop_w % 32:
        raise FdnHipError("tile sizes must be multiples of 32 (three levels x 8x8 patches)")
    tiles, ij = split(x.contiguous(), crop_h, crop_w)
    outs = torch.empty_like(tiles)
    for s in range(0, tiles.shape[0], batch):
        t = tiles[s:s + batch]
        outs[s:s + batch] = net(t, ratio_i=lpnet(t), device=t.device)[0]
    return merge(outs, ij, x.shape[2], x.shape[3])
