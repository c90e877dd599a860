"""Import the reference arch files BY PATH (test tooling; runs only in the build container).

The reference package route dies on `import cv2` (basicsr/utils/img_util.py:1) and
LPNet_arch.py:84 needs torchvision, so we (1) load the two arch files with
importlib.util.spec_from_file_location, (2) put a tiny torchvision.transforms stub into
sys.modules (Grayscale = 0.2989 R + 0.587 G + 0.114 B), (3) wrap torch.load so the
hard-coded fourier_gamma.pth request (FDN_arch.py:860-862) is answered with the MAR's own
default-initialised state dict.  Nothing from /root/reference is copied anywhere.
"""
import contextlib
import importlib.util
import io
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("FDN_REFERENCE_ROOT", "/root/reference")
ARCH_DIR = os.path.join(REF_ROOT, "basicsr", "models", "archs")


def reference_available():
    return os.path.isfile(os.path.join(ARCH_DIR, "FDN_arch.py"))


def _install_torchvision_stub():
    if "torchvision" in sys.modules:
        return
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class Grayscale:
        def __init__(self, num_output_channels=1):
            self.n = num_output_channels

        def __call__(self, x):
            r, g, b = x.unbind(dim=-3)
            y = (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)
            return y if self.n == 1 else y.expand(*x.shape)

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    tr.Grayscale = Grayscale
    tr.Compose = Compose
    tv.transforms = tr
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tr


def _load_by_path(name, fname):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ARCH_DIR, fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_cache = {}


def ref_fdn_module():
    if "fdn" not in _cache:
        _cache["fdn"] = _load_by_path("_ref_FDN_arch", "FDN_arch.py")
    return _cache["fdn"]


def ref_lpnet_module():
    if "lp" not in _cache:
        _install_torchvision_stub()
        _cache["lp"] = _load_by_path("_ref_LPNet_arch", "LPNet_arch.py")
    return _cache["lp"]


def build_ref_fdn(seed=0):
    """Construct the reference FDN() with seeded default init (trained weights are absent)."""
    m = ref_fdn_module()
    real_load = torch.load

    def fake_load(path, *a, **k):
        if str(path).endswith("fourier_gamma.pth"):
            return {"params": m.MAR(True).state_dict()}
        return real_load(path, *a, **k)

    torch.manual_seed(seed)
    torch.load = fake_load
    try:
        net = m.FDN()
    finally:
        torch.load = real_load
    return net.eval()


@contextlib.contextmanager
def quiet():
    """Swallow the stray print(ratio.shape) at FDN_arch.py:211."""
    old = sys.stdout
    sys.stdout = io.StringIO()
    try:
        yield
    finally:
        sys.stdout = old
