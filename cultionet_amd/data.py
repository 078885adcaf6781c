"""In-memory ``Data`` batch container (the contract of /root/reference/src/cultionet/data/data.py:51-139).

Only what the hot path reads: ``x [B,C,T,H,W]``, ``y [B,H,W]`` (int, -1 = unlabeled), ``bdist [B,H,W]``,
``lon``/``lat`` [B], plus arbitrary tensor/array/list attributes; ``.to()`` returns a new Data and is how
Lightning moves a batch to the device. File IO / CRS / plotting are out of scope (GIS stack).
"""
from __future__ import annotations

from copy import deepcopy
from typing import Optional

import numpy as np
import torch


class Data:
    def __init__(self, x: torch.Tensor, y: Optional[torch.Tensor] = None, **kwargs):
        self.x = x
        self.y = y
        for k, v in kwargs.items():
            if v is not None:
                assert isinstance(v, (torch.Tensor, np.ndarray, list)), \
                    "Only tensors, arrays, and lists are supported."
            setattr(self, k, v)

    def _get_attrs(self) -> set:
        return set(self.__dict__.keys())

    def to_dict(self, device: Optional[str] = None, dtype: Optional[str] = None) -> dict:
        out = {}
        for key in self._get_attrs():
            value = getattr(self, key)
            if isinstance(value, torch.Tensor):
                out[key] = value.clone()
                if device is not None:
                    out[key] = out[key].to(device=device, dtype=dtype)
            elif isinstance(value, np.ndarray):
                out[key] = value.copy()
            else:
                out[key] = None if value is None else deepcopy(value)
        return out

    def to(self, device: Optional[str] = None, dtype: Optional[str] = None) -> "Data":
        return Data(**self.to_dict(device=device, dtype=dtype))

    def copy(self) -> "Data":
        return Data(**self.to_dict())

    def __add__(self, other: "Data") -> "Data":
        return Data(**{k: v + getattr(other, k) for k, v in self.to_dict().items() if isinstance(v, torch.Tensor)})

    @property
    def num_samples(self) -> int:
        return self.x.shape[0]

    @property
    def num_channels(self) -> int:
        return self.x.shape[1]

    @property
    def num_time(self) -> int:
        return self.x.shape[2]

    @property
    def height(self) -> int:
        return self.x.shape[3]

    @property
    def width(self) -> int:
        return self.x.shape[4]


def collate_fn(data_list: "list[Data]") -> Data:
    """Concatenate samples into one batch (the DataLoader collate of /root/reference/src/cultionet/data/utils.py:
    55-68): tensors and arrays are concatenated along their first axis, lists are chained, None stays None."""
    first = data_list[0]
    out = {}
    for key in first.to_dict().keys():
        v0 = getattr(first, key)
        if v0 is None:
            out[key] = None
        elif isinstance(v0, torch.Tensor):
            out[key] = torch.cat([getattr(d, key) for d in data_list])
        elif isinstance(v0, np.ndarray):
            out[key] = np.concatenate([getattr(d, key) for d in data_list])
        elif isinstance(v0, list):
            merged = []
            for d in data_list:
                merged = merged + getattr(d, key)
            out[key] = merged
        else:
            raise TypeError(f"collate_fn: unsupported attribute type {type(v0)} for {key!r}")
    return Data(**out)
