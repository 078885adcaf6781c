"""Neighborhood attention (NA2D) CPU restatement -- TEST INFRASTRUCTURE ONLY.

The reference calls ``natten.NeighborhoodAttention2D`` (natten==0.17.1, pinned
at /root/reference/.github/workflows/ci.yml:53 and README.md:247) from
/root/reference/src/cultionet/nn/modules/convolution.py:341-350. natten is a
third-party C++/CUDA dependency that is not vendored in /root/reference and is
not installed here, so its published algorithm is restated below, twice and
independently (a vectorised gather and scalar loops); tests/test_na2d_oracle.py
requires the two to agree and checks window properties. The reference's own
tests only assert output *shapes* at this boundary
(/root/reference/tests/test_tower_unet.py:7-38) => NA parity is UNPINNED against
real natten; everything else in the oracle is pinned against the imported
reference.

Semantics (natten 0.17.1, unfused path, no rel-pos bias, non-causal):
  * per axis, the K keys of query i are start(i) + j*d, j = 0..K-1, with the
    window start pulled inward at the borders (no zero padding);
  * logits are laid out [B, heads, H, W, K*K] row-major over the window and
    soft-maxed over the last axis;
  * module: qkv = Linear(C, 3C); q *= head_dim**-0.5; proj = Linear(C, C).
"""
from __future__ import annotations

import torch
import torch.nn as nn


def window_start(i: int, length: int, kernel_size: int, dilation: int) -> int:
    """First key index of query ``i`` along one axis (natten ``get_window_start``)."""
    n = kernel_size // 2
    if dilation <= 1:
        return max(i - n, 0) + ((length - i - n - 1) if (i + n >= length) else 0)
    ni = i - n * dilation
    if ni < 0:
        return i % dilation
    if i + n * dilation >= length:
        imodd = i % dilation
        a = (length // dilation) * dilation
        b = length - a
        if imodd < b:
            return length - b + imodd - 2 * n * dilation
        return a + imodd - kernel_size * dilation
    return ni


def window_index(length: int, kernel_size: int, dilation: int) -> torch.Tensor:
    """[length, K] key indices for every query position along one axis."""
    idx = torch.empty(length, kernel_size, dtype=torch.long)
    for i in range(length):
        s = window_start(i, length, kernel_size, dilation)
        for j in range(kernel_size):
            idx[i, j] = s + j * dilation
    return idx


def _gather_windows(t: torch.Tensor, ih: torch.Tensor, iw: torch.Tensor) -> torch.Tensor:
    """t: [B, h, H, W, D] -> [B, h, H, W, K, K, D] of neighbourhood rows."""
    # rows then columns (advanced indexing on dims 2 and 3)
    g = t[:, :, ih]  # [B, h, H, K, W, D]
    g = g[:, :, :, :, iw]  # [B, h, H, K, W, K, D]
    return g.permute(0, 1, 2, 4, 3, 5, 6)


def na2d_qk(q: torch.Tensor, k: torch.Tensor, kernel_size: int, dilation: int = 1) -> torch.Tensor:
    """q, k: [B, heads, H, W, D] -> logits [B, heads, H, W, K*K]."""
    B, h, H, W, D = q.shape
    ih = window_index(H, kernel_size, dilation).to(q.device)
    iw = window_index(W, kernel_size, dilation).to(q.device)
    kw = _gather_windows(k, ih, iw)  # [B,h,H,W,K,K,D]
    attn = torch.einsum("bhxyd,bhxyijd->bhxyij", q, kw)
    return attn.reshape(B, h, H, W, kernel_size * kernel_size)


def na2d_av(attn: torch.Tensor, v: torch.Tensor, kernel_size: int, dilation: int = 1) -> torch.Tensor:
    """attn: [B, heads, H, W, K*K], v: [B, heads, H, W, D] -> [B, heads, H, W, D]."""
    B, h, H, W, D = v.shape
    ih = window_index(H, kernel_size, dilation).to(v.device)
    iw = window_index(W, kernel_size, dilation).to(v.device)
    vw = _gather_windows(v, ih, iw)
    a = attn.reshape(B, h, H, W, kernel_size, kernel_size)
    return torch.einsum("bhxyij,bhxyijd->bhxyd", a, vw)


def na2d(q, k, v, kernel_size: int, dilation: int = 1, scale=None):
    """Fused-call signature of natten.functional.na2d on [B, H, W, heads, D] tensors."""
    if scale is None:
        scale = q.shape[-1] ** -0.5
    qh, kh, vh = (t.permute(0, 3, 1, 2, 4) for t in (q, k, v))
    attn = na2d_qk(qh * scale, kh, kernel_size, dilation).softmax(dim=-1)
    out = na2d_av(attn, vh, kernel_size, dilation)
    return out.permute(0, 2, 3, 1, 4)


def na2d_scalar(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, kernel_size: int, dilation: int) -> torch.Tensor:
    """Second, independent restatement: brute-force loops (small cases only).

    q is already scaled. q, k, v: [B, heads, H, W, D] -> [B, heads, H, W, D].
    """
    B, h, H, W, D = q.shape
    out = torch.zeros_like(q)
    for y in range(H):
        sy = window_start(y, H, kernel_size, dilation)
        for x in range(W):
            sx = window_start(x, W, kernel_size, dilation)
            logits = []
            vals = []
            for a in range(kernel_size):
                for b in range(kernel_size):
                    ky, kx = sy + a * dilation, sx + b * dilation
                    logits.append((q[:, :, y, x] * k[:, :, ky, kx]).sum(-1))
                    vals.append(v[:, :, ky, kx])
            lg = torch.stack(logits, dim=-1)  # [B,h,KK]
            p = torch.softmax(lg, dim=-1)
            vv = torch.stack(vals, dim=-2)  # [B,h,KK,D]
            out[:, :, y, x] = (p.unsqueeze(-1) * vv).sum(-2)
    return out


class NeighborhoodAttention2D(nn.Module):
    """Module-level restatement of natten.NeighborhoodAttention2D (NHWC input).

    Parameter names follow natten so reference checkpoints load:
    ``qkv.{weight,bias}``, ``proj.{weight,bias}``.
    """

    def __init__(
        self,
        dim: int,
        num_heads: int,
        kernel_size: int,
        dilation: int = 1,
        is_causal: bool = False,
        rel_pos_bias: bool = False,
        qkv_bias: bool = True,
        qk_scale=None,
        attn_drop: float = 0.0,
        proj_drop: float = 0.0,
    ):
        super().__init__()
        assert not rel_pos_bias and not is_causal
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = qk_scale or self.head_dim**-0.5
        self.kernel_size = kernel_size
        self.dilation = dilation
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, H, W, C = x.shape
        qkv = (
            self.qkv(x)
            .reshape(B, H, W, 3, self.num_heads, self.head_dim)
            .permute(3, 0, 4, 1, 2, 5)
        )
        q, k, v = qkv[0], qkv[1], qkv[2]
        q = q * self.scale
        attn = na2d_qk(q, k, self.kernel_size, self.dilation)
        attn = attn.softmax(dim=-1)
        attn = self.attn_drop(attn)
        x = na2d_av(attn, v, self.kernel_size, self.dilation)
        x = x.permute(0, 2, 3, 1, 4).reshape(B, H, W, C)
        return self.proj_drop(self.proj(x))
