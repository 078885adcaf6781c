"""Sliding-window predict -- CPU restatement, TEST INFRASTRUCTURE ONLY.

The reference's predict path needs dask / xarray / rasterio / geowombat (absent here), so its arithmetic is restated
step by step with numpy / torch, window by window, the way the reference does it:

  * tiling: chunks of ``window_size``; every chunk grown by ``padding`` with zeros outside the scene
    (``map_overlap(depth=padding, boundary=0, trim=False)``, /root/reference/src/cultionet/data/create.py:176-212)
    and zero-filled at the bottom / right to ``window_size + 2*padding`` (``BatchStore.write_batch``,
    data/store.py:69-100);
  * ``EdgeDataset.get``: ``x / 10000`` -> ``clip(1e-9, 1)`` (data/datasets.py:443-446), then the z-score
    ``(x - mean) / std`` per channel (utils/normalize.py:63-82);
  * eval forward;
  * ``LightningGTiffWriter.write_on_batch_end``: slice ``[padding : padding + h]``, ``x 10000``, ``clip(0, 10000)``,
    write as uint16 (callbacks.py:176-227; rasterio casts by truncation).
Parity status: restatement-checked (no reference fixture can exist without the GIS stack).
"""
from __future__ import annotations

import numpy as np
import torch


def predict_scene(model, scene: np.ndarray, window_size: int, padding: int, mean=None, std=None) -> np.ndarray:
    C, Tn, H, W = scene.shape
    S = window_size + 2 * padding
    out = np.zeros((3, H, W), dtype=np.uint16)
    model.eval()
    for r0 in range(0, H, window_size):
        for c0 in range(0, W, window_size):
            h, w = min(window_size, H - r0), min(window_size, W - c0)
            tile = np.zeros((C, Tn, S, S), dtype=np.float64)
            ys, xs = max(r0 - padding, 0), max(c0 - padding, 0)
            ye, xe = min(r0 + h + padding, H), min(c0 + w + padding, W)
            # real neighbours where the scene has them, zeros elsewhere; note upstream fills BEYOND the grown chunk
            # (h + 2*padding < S for end chunks) with zeros too, while the scene may still hold pixels there: the grown
            # chunk ends at r0 + h + padding, which for end chunks is the scene edge anyway
            tile[:, :, ys - (r0 - padding):ye - (r0 - padding), xs - (c0 - padding):xe - (c0 - padding)] = \
                scene[:, :, ys:ye, xs:xe]
            x = np.clip(tile.astype(np.float32) / np.float32(10000.0), np.float32(1e-9), np.float32(1.0))
            if mean is not None:
                x = (x - np.asarray(mean, np.float32).reshape(C, 1, 1, 1)) / np.asarray(std, np.float32).reshape(C, 1, 1, 1)
            with torch.no_grad():
                pred = model(torch.from_numpy(x.astype(np.float32))[None])
            stack = np.concatenate([pred[k][0].numpy()[:, padding:padding + h, padding:padding + w]
                                    for k in ("distance", "edge", "crop")], axis=0)
            out[:, r0:r0 + h, c0:c0 + w] = np.clip(stack * 10000.0, 0, 10000.0).astype(np.uint16)
    return out
