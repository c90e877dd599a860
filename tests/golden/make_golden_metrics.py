"""Golden values for the validation metrics: calculate_psnr and the ssim3d path of
calculate_ssim are lifted out of the reference's basicsr/metrics/psnr_ssim.py by AST (the module imports cv2 and skimage,
absent here) and executed with (1) a stand-in for cv2.getGaussianKernel restating OpenCV's published formula (cv2 is a
third-party dependency the reference does not vendor or pin) and (2) `.cuda()` turned into a no-op (no GPU in the build
container).  Run:  python tests/golden/make_golden_metrics.py"""
import ast
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _refload import REF_ROOT  # noqa: E402


def get_gaussian_kernel(ksize, sigma):
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    return (k / k.sum()).reshape(-1, 1)


def reference_functions():
    path = os.path.join(REF_ROOT, "basicsr", "metrics", "psnr_ssim.py")
    tree = ast.parse(open(path).read())
    want = ("calculate_psnr", "calculate_ssim", "_ssim_3d", "_generate_3d_gaussian_kernel", "_3d_gaussian_calculator")
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    ns = {"torch": torch, "np": np, "cv2": types.SimpleNamespace(getGaussianKernel=get_gaussian_kernel),
          "reorder_image": lambda img, input_order="HWC": img if input_order == "HWC" else img.transpose(1, 2, 0),
          "to_y_channel": None}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    return ns


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    R = reference_functions()
    out, cases = {}, {}
    for name, (h, w), scale, noise, border in (("a", (48, 64), 1.0, 0.05, 0), ("b", (37, 53), 1.0, 0.2, 4), ("c", (40, 40), 255.0, 12.0, 0)):
        g = torch.Generator().manual_seed(h * 100 + w)
        x = torch.rand(3, h, w, generator=g) * scale
        y = (x + noise * torch.randn(3, h, w, generator=g)).clamp(0, scale)
        psnr = R["calculate_psnr"](x.clone(), y.clone(), border, input_order="HWC")       # tensors are transposed to HWC inside (:37-44)
        ssim = R["calculate_ssim"](x.clone(), y.clone(), border, input_order="HWC", ssim3d=True)
        out[name + "_x"], out[name + "_y"] = x.numpy(), y.numpy()
        cases[name] = {"crop_border": border, "psnr": float(psnr), "ssim": float(ssim)}
        print(name, cases[name])
    out["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)


if __name__ == "__main__":
    main()
