"""MI355X-native I_predict_net (drop-in for the reference's basicsr/models/archs/LPNet_arch.py).

Same class name, ctor `I_predict_net(c=16)`, `forward(x, use_ori_i=False)` -> (B, 1) and the same
292-key checkpoint layout (incl. BatchNorm running stats), reference LPNet_arch.py:42-134.  The
network is conv + BN(eval) + SE only (no FFT); BatchNorm is folded into the preceding conv once per
weight version and every op runs in libfdn_hip.so.  No CPU fallback.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401
from einops import rearrange  # noqa: F401

from fdn_hip import ACT_NONE, ACT_RELU, ACT_SIGMOID, ops

from .FDN_arch import _Cache, _w

try:  # the reference driver uses `transforms.Grayscale` from this module's namespace (inference_fdn_lolblur.py:35)
    from torchvision import transforms
except Exception:  # torchvision absent: minimal stand-in with the ITU-R 601 weights torchvision uses
    class _Grayscale:
        def __init__(self, num_output_channels=1):
            self.n = num_output_channels

        def __call__(self, x):
            r, g, b = x.unbind(dim=-3)
            y = (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)
            return y if self.n == 1 else y.expand(*x.shape)

    class _Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class transforms:  # noqa: N801
        Grayscale = _Grayscale
        Compose = _Compose


def _fold(conv, bn, cache, name):
    """conv (no bias) followed by eval-mode BatchNorm -> (weight', bias')."""
    srcs = [conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]

    def scale():
        return bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)

    w = cache.get(name + ".w", srcs, lambda: conv.weight.detach() * scale().view(-1, 1, 1, 1))
    b = cache.get(name + ".b", srcs, lambda: bn.bias.detach() - bn.running_mean * scale())
    return w, b


class SEBlock(nn.Module):
    """SE bottleneck (reference LPNet_arch.py:42-81)."""

    def __init__(self, in_channels, filters, stride=1, is_1x1conv=False):
        super().__init__()
        f1, f2, f3 = filters
        self.stride = stride
        self.is_1x1conv = is_1x1conv
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, f1, 1, stride=stride, bias=False), nn.BatchNorm2d(f1), nn.ReLU())
        self.conv2 = nn.Sequential(nn.Conv2d(f1, f2, 3, stride=1, padding=1, bias=False), nn.BatchNorm2d(f2), nn.ReLU())
        self.conv3 = nn.Sequential(nn.Conv2d(f2, f3, 1, stride=1, bias=False), nn.BatchNorm2d(f3))
        if is_1x1conv:
            self.shortcut = nn.Sequential(nn.Conv2d(in_channels, f3, 1, stride=stride, bias=False), nn.BatchNorm2d(f3))
        self.se = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(f3, f3 // 16, 1), nn.ReLU(),
                                nn.Conv2d(f3 // 16, f3, 1), nn.Sigmoid())
        self._c = _Cache()

    def _pw(self, x, seq, name, act, stride):
        w, b = _fold(seq[0], seq[1], self._c, name)
        if stride == 1:
            return ops.conv1x1(x, w, b, act=act)
        return ops.conv2d(x, w, b, stride=stride, pad=0, act=act)

    def forward(self, x):
        y = self._pw(x, self.conv1, "c1", ACT_RELU, self.stride)
        w2, b2 = _fold(self.conv2[0], self.conv2[1], self._c, "c2")
        y = ops.conv2d(y, w2, b2, stride=1, pad=1, act=ACT_RELU)
        y = self._pw(y, self.conv3, "c3", ACT_NONE, 1)
        g = ops.global_avgpool(y)
        g = ops.conv1x1(g, _w(self.se[1].weight), _w(self.se[1].bias), act=ACT_RELU)
        g = ops.conv1x1(g, _w(self.se[3].weight), _w(self.se[3].bias), act=ACT_SIGMOID)
        sc = self._pw(x, self.shortcut, "sc", ACT_NONE, self.stride) if self.is_1x1conv else x
        return ops.se_apply(y, g, sc)


class I_predict_net(nn.Module):
    """Illumination predictor: one sigmoid scalar per image (reference LPNet_arch.py:86-134)."""

    def __init__(self, c=16):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(3, c, kernel_size=7, stride=2, padding=3, bias=False), nn.BatchNorm2d(c),
                                   nn.ReLU(), nn.AvgPool2d(kernel_size=3, stride=2, padding=1))
        self.conv2 = self._make_layer(c, (c, c, 2 * c), 3, 1)
        self.conv3 = self._make_layer(2 * c, (2 * c, 2 * c, 4 * c), 3, 2)
        self.conv4 = self._make_layer(4 * c, (4 * c, 4 * c, 8 * c), 6, 6)
        self.global_average_pool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Sequential(nn.Linear(8 * c, 8 * c))
        self.fc2 = nn.Sequential(nn.Linear(8 * c, 1))
        self.sigmoid = nn.Sigmoid()
        self.trans_gray = transforms.Compose([transforms.Grayscale(num_output_channels=1)])
        self._c = _Cache()

    @staticmethod
    def _make_layer(in_channels, filters, num, stride=1):
        layers = [SEBlock(in_channels, filters, stride=stride, is_1x1conv=True)]
        layers += [SEBlock(filters[2], filters, stride=1, is_1x1conv=False) for _ in range(1, num)]
        return nn.Sequential(*layers)

    def forward(self, x, use_ori_i=False):
        x = x.contiguous()
        w, b = _fold(self.conv1[0], self.conv1[1], self._c, "stem")
        y = ops.conv2d(x, w, b, stride=2, pad=3, act=ACT_RELU)
        y = ops.avgpool3s2(y)
        for stage in (self.conv2, self.conv3, self.conv4):
            for blk in stage:
                y = blk(y)
        g = ops.global_avgpool(y)                       # "B C H W -> B (H W C)" with H = W = 1
        g = ops.conv1x1(g, _w(self.fc[0].weight), _w(self.fc[0].bias))
        g = ops.conv1x1(g, _w(self.fc2[0].weight), _w(self.fc2[0].bias), act=ACT_SIGMOID)
        out = g.view(x.shape[0], 1)
        if use_ori_i:                                   # LPNet_arch.py:131-132 (unused by the FDN drivers)
            m = ops.global_avgpool(x).view(x.shape[0], 3)
            gray = (0.2989 * m[:, 0] + 0.587 * m[:, 1] + 0.114 * m[:, 2]).view(-1, 1)
            out = gray / out
        return out
