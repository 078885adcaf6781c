"""Device-side input / output edges of the hot path.

``prepare_chips`` fuses what ``EdgeDataset.get`` + ``NormValues`` do on the CPU per sample
(/root/reference/src/cultionet/data/datasets.py:443-446, utils/normalize.py:63-82) into one HBM pass over the
collated batch: raw (integer) reflectances -> x/10000 -> clip(1e-9, 1) -> z-score, fp32.
``predictions_to_uint16`` is the arithmetic of ``LightningGTiffWriter.write_on_batch_end``
(callbacks.py:176-227): drop the window padding, x10000, clip, uint16 (the file IO stays with the caller).
"""
from __future__ import annotations

import typing as T

import torch

from . import _lib
from .engine import _stream

SCALE_FACTOR = 10_000.0
_DTYPES = {torch.float32: 0, torch.int32: 1, torch.int16: 2, torch.uint16: 3}


def prepare_chips(x_raw: torch.Tensor, mean: T.Optional[torch.Tensor] = None, std: T.Optional[torch.Tensor] = None,
                  scale: float = 1.0 / SCALE_FACTOR, lo: float = 1e-9, hi: float = 1.0) -> torch.Tensor:
    """x_raw: [B, C, T, H, W] on the GPU (f32 / i32 / i16 / u16); mean/std: per-channel [C] (or broadcastable)."""
    if not x_raw.is_cuda:
        raise RuntimeError("prepare_chips needs a device tensor")
    if x_raw.dtype not in _DTYPES:
        raise TypeError(f"unsupported raw dtype {x_raw.dtype}")
    x_raw = x_raw.contiguous()
    B, C = x_raw.shape[0], x_raw.shape[1]
    L = int(x_raw[0, 0].numel())
    out = torch.empty(x_raw.shape, dtype=torch.float32, device=x_raw.device)
    m = mean.to(device=x_raw.device, dtype=torch.float32).reshape(-1).contiguous() if mean is not None else None
    s = std.to(device=x_raw.device, dtype=torch.float32).reshape(-1).contiguous() if std is not None else None
    if m is not None and m.numel() != C:
        raise ValueError("mean must hold one value per channel")
    _lib.call("cn_prepare_chips_f32", x_raw.data_ptr(), _DTYPES[x_raw.dtype], out.data_ptr(),
              m.data_ptr() if m is not None else None, s.data_ptr() if s is not None else None, B, C, L, float(scale),
              float(lo), float(hi), _stream())
    return out


def predictions_to_uint16(prediction: T.Dict[str, torch.Tensor], padding: int, height: int, width: int,
                          scale: float = SCALE_FACTOR) -> torch.Tensor:
    """{distance, edge, crop}: [B,1,H,W] -> uint16 [B,3,height,width] with the window padding removed."""
    d, e, c = (prediction[k].contiguous() for k in ("distance", "edge", "crop"))
    B, _, H, W = d.shape
    out = torch.empty((B, 3, height, width), dtype=torch.uint16, device=d.device)
    _lib.call("cn_predictions_to_u16", d.data_ptr(), e.data_ptr(), c.data_ptr(), out.data_ptr(), B, H, W, int(padding),
              int(padding), int(height), int(width), float(scale), _stream())
    return out
