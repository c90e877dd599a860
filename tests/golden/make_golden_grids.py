"""Golden vectors for the tiled-inference helpers: the reference's own
`grids` / `grids_inverse` methods (basicsr/models/image_restoration_model.py:261-339) are lifted out of the class
by AST at generation time (the module itself cannot be imported here: cv2, lmdb, ...) and executed on a stub `self`,
so the fixture holds what the reference code computes.  Run:  python tests/golden/make_golden_grids.py"""
import ast
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _refload import REF_ROOT  # noqa: E402


def reference_methods():
    path = os.path.join(REF_ROOT, "basicsr", "models", "image_restoration_model.py")
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ImageRestorationModel"][0]
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("grids", "grids_inverse")]
    mod = ast.Module(body=fns, type_ignores=[])
    ns = {"torch": torch}
    exec(compile(mod, path, "exec"), ns)
    return ns["grids"], ns["grids_inverse"]


def main():
    grids, grids_inverse = reference_methods()
    out = {}
    cases = [("a", 50, 75, 32, 32), ("b", 48, 48, 48, 48), ("c", 65, 35, 32, 24), ("d", 33, 100, 16, 48)]
    for name, h, w, ch, cw in cases:
        g = torch.Generator().manual_seed(h * 1000 + w)
        x = torch.rand(1, 3, h, w, generator=g)
        me = types.SimpleNamespace(gt=x, lq=x, scale=1, device="cpu", opt={"val": {"crop_size_h": ch, "crop_size_w": cw}})
        grids(me)
        tiles = me.lq
        idx = np.array([[d["i"], d["j"]] for d in me.idxes], dtype=np.int32)
        outs = torch.rand(tiles.shape, generator=g)                                  # stand-in for the network outputs
        me.outs = outs
        grids_inverse(me)
        out.update({f"{name}_x": x.numpy(), f"{name}_crop": np.array([ch, cw], dtype=np.int32), f"{name}_tiles": tiles.numpy(),
                    f"{name}_idx": idx, f"{name}_outs": outs.numpy(), f"{name}_merged": me.output.numpy()})
        print(name, (h, w), (ch, cw), "tiles", tuple(tiles.shape), idx.tolist())
    np.savez_compressed(os.path.join(HERE, "grids.npz"), **out)


if __name__ == "__main__":
    main()
