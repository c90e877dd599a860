"""Deterministic synthetic weights for parity tests.

The trained FDN checkpoint is absent from the reference checkout, so
every FDN-side fixture uses weights produced by this generator: a value depends only on
(seed, state-dict key, shape), never on module construction order, so the identical state dict
can be loaded into the reference (in the build container, tests/golden/make_golden.py) and into
the HIP-backed modules on the GPU box without shipping 32 MB of floats.
"""
import math
import zlib

import torch


def _gen(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(key, shape, seed=0):
    g = _gen(key, seed)
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if leaf in ("fft", "ffta"):                      # spectral gains, init 1 in the reference
        return 1.0 + 0.2 * torch.randn(shape, generator=g)
    if leaf == "fftp":                               # spectral phase offsets, init 0
        return 0.5 * torch.randn(shape, generator=g)
    if leaf == "weight" and len(shape) == 1:         # LayerNorm / BatchNorm scale
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf == "bias":
        return 0.05 * torch.randn(shape, generator=g)
    if leaf == "weight":
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        bound = 1.0 / math.sqrt(max(fan_in, 1))      # same scale as PyTorch's default conv init
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    return 0.1 * torch.randn(shape, generator=g)


def synth_state_dict(shapes, seed=0, prefix_key="", tame=None):
    """shapes: {key: shape}.  prefix_key is prepended when hashing (lets a sub-module fixture
    use names independent of where it sits).  tame: multiply every FDformer
    '*project_out.weight' under net_p by this factor."""
    sd = {}
    for k, shp in shapes.items():
        t = synth_tensor(prefix_key + k, shp, seed)
        if tame is not None and k.startswith("net_p.") and k.endswith("project_out.weight"):
            t = t * tame
        sd[k] = t
    return sd


def shapes_of(module_or_sd):
    sd = module_or_sd if isinstance(module_or_sd, dict) else module_or_sd.state_dict()
    return {k: tuple(v.shape) for k, v in sd.items()}
