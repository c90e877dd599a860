"""MI355X-native FDN_lolv1 (drop-in for the reference's basicsr/models/archs/fdnlol24_arch.py:951-1033).

The LOL-v1 model is the LOL-Blur FDN at dim = 24 (FDformer widths 24/48/96, FDSA E = 28/57/115, FDFFN
Hd = 64/129/259, fdnlol24_arch.py:963-969) with a MAR whose ProcessBlock applies its `.cat` 1x1 conv
(:769-776) and gates the ratio multiplication on `use_ratio` (:160-169).  `forward` returns the restored
image four times (:1031).  Same parameter tree as the reference (1503 keys), same call convention as
inference_fdn_lolv1.py:62; every compute step runs in libfdn_hip.so - the kernels are shape-generic.

Not re-created: the hard-coded `mar_lol.pth` load of the reference constructor (:976-978; a full
checkpoint overwrites net_a anyway) and the classes the model never instantiates (Se, PPM, SpatialAttention,
Img_merge, SpaBlock).
"""
import numbers  # noqa: F401

import numpy as np  # noqa: F401
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401
from einops import rearrange  # noqa: F401

from fdn_hip import ops
from basicsr.models.archs.FDN_arch import (FDN, FDformer, FreBlock, LayerNorm, MAR_archa, _w)   # noqa: F401
from basicsr.models.archs import FDN_arch as _blur


class ProcessBlock(nn.Module):
    """cat(FreBlock(x)) + x  (fdnlol24_arch.py:760-776, spatial=False)."""

    def __init__(self, in_nc, spatial=False):
        super().__init__()
        if spatial:
            raise NotImplementedError("spatial=True is never instantiated by FDN_lolv1 (fdnlol24_arch.py:104-137)")
        self.frequency_process = FreBlock(in_nc)
        self.cat = nn.Conv2d(in_nc, in_nc, 1, 1, 0)

    def forward(self, x):
        y = self.frequency_process(x, skip_gain=1.0)                               # irfft2(...) + x, :758
        return ops.conv1x1(y, _w(self.cat.weight), _w(self.cat.bias), res=x)      # cat(x_freq) + xori, :774-776


class fourier_multi_scale_gamma2(MAR_archa):
    """fdnlol24_arch.py:97-209: MAR_archa's wiring with the live-cat ProcessBlock."""

    def __init__(self, use_ratio):
        super().__init__(use_ratio=use_ratio, block=ProcessBlock, apply_ratio=bool(use_ratio))


class MAR(_blur.MAR):
    """fdnlol24_arch.py:211-248."""

    def __init__(self, use_ratio=True):
        super().__init__(use_ratio=use_ratio)
        self.net = fourier_multi_scale_gamma2(use_ratio=use_ratio)


class FDN_lolv1(FDN):
    """FDN_lolv1 = MAR + FDformer(dim=24) + 3 LayerNorm(3) (fdnlol24_arch.py:951-1033)."""

    def __init__(self):
        nn.Module.__init__(self)
        self.net_a = MAR(use_ratio=True)
        self.net_p = FDformer(inp_channels=3, out_channels=3, dim=24, num_blocks=[6, 6, 10], num_refinement_blocks=4,
                              ffn_expansion_factor=3, bias=False)
        for p in self.net_a.parameters():
            p.requires_grad = False
        self.norm1 = LayerNorm(3)
        self.norm2 = LayerNorm(3)
        self.norm3 = LayerNorm(3)

    def forward(self, inp_img, ori=None, device=None, ratio_i=None, mode=1):
        out = super().forward(inp_img, ori=ori, device=device, ratio_i=ratio_i, mode=mode)[0]
        return out, out, out, out
