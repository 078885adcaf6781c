"""Key-seeded synthetic weights and inputs for benchmarks and parity runs.

Every state-dict tensor is drawn from a generator seeded by crc32(key), so the reference (in the build
container), the CPU oracle and this HIP build hold identical parameters without shipping a weight file.
Inputs follow /root/reference/tests/conftest.py:19-55 (x ~ U[0,1), bdist ~ U[0,1), y in {0,1,2} or
{-1,..,2}) but are seeded. tests/test_synthetic.py checks these stay bit-identical to the oracle's copies.
"""
from __future__ import annotations

import math
import typing as T
import zlib

import torch

_CONV_BIAS_MARKERS = ("skip.", "up_conv.", "qkv.", "proj.", "conv.1.bias", "final_dist.0", "final_edge.0",
                      "final_crop.0")


def seeded_state_dict(template: T.Dict[str, torch.Tensor], salt: int = 0) -> T.Dict[str, torch.Tensor]:
    out = {}
    for key, t in template.items():
        k = key.replace("cultionet_TowerUNet.mask_model.", "").replace("_orig_mod.", "")
        g = torch.Generator().manual_seed((zlib.crc32(k.encode()) + salt) & 0x7FFFFFFF)
        shape = tuple(t.shape)
        if k.endswith("num_batches_tracked"):
            v = torch.zeros(shape, dtype=t.dtype)
        elif k.endswith("running_mean"):
            v = torch.randn(shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            v = torch.rand(shape, generator=g) + 0.5
        elif "gamma" in k:
            v = torch.rand(shape, generator=g) * 0.45 + 0.8
        elif t.dim() >= 2:
            v = torch.randn(shape, generator=g) * math.sqrt(2.0 / math.prod(shape[1:]))
        elif k.endswith("weight"):
            v = 1.0 + torch.randn(shape, generator=g) * 0.1
        else:
            v = torch.randn(shape, generator=g) * (0.5 if any(s in k for s in _CONV_BIAS_MARKERS) else 0.1)
        out[key] = v.to(t.dtype)
    return out


def seeded_batch(batch: int, channels: int = 3, time: int = 12, height: int = 100, width: int = 100, seed: int = 7,
                 with_mask: bool = False):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, channels, time, height, width, generator=g)
    bdist = torch.rand(batch, height, width, generator=g)
    y = torch.randint(-1 if with_mask else 0, 3, (batch, height, width), generator=g)
    return x, y, bdist
