"""Tanimoto loss family on HIP kernels.

Host-side mirror of /root/reference/src/cultionet/losses/losses.py (TanimotoComplementLoss,
TanimotoDistLoss, CombinedLoss): same class names, constructor arguments and call signature
``loss(inputs, targets, mask=None)``. The arithmetic (five masked sums per sample, the depth-5 complement
formula, batch mean and the elementwise gradient) runs in cn_tanimoto_{fwd,bwd}_f32; LossPreprocessing's
one-hot / mask handling is folded into the kernel's target / mask modes.
"""
from __future__ import annotations

import typing as T

import torch
import torch.nn as nn

from . import engine as E
from .autograd_bridge import tanimoto_autograd


def _kernel_args(inputs: torch.Tensor, targets: torch.Tensor, mask: T.Optional[torch.Tensor], one_hot_targets: bool):
    C = inputs.shape[1]
    kw: T.Dict[str, T.Any] = {}
    if targets.dtype == torch.int64:
        if one_hot_targets and C > 1:
            kw.update(labels=targets.contiguous(), target_mode=E.TGT_ONEHOT)
        else:  # class index == value for a single channel: t = (y == 1) would drop other classes; use float
            kw.update(target_f=targets.to(torch.float32).reshape(inputs.shape).contiguous(), target_mode=E.TGT_FLOAT)
    else:
        kw.update(target_f=targets.to(torch.float32).reshape(inputs.shape).contiguous(), target_mode=E.TGT_FLOAT)
    if mask is None:
        kw.update(mask=None, mask_mode=E.MSK_NONE)
    elif mask.dtype == torch.int64:
        kw.update(mask=mask.contiguous(), mask_mode=E.MSK_I64)
    else:
        kw.update(mask=mask.to(torch.float32).contiguous(), mask_mode=E.MSK_F32)
    return kw


class _TanimotoBase(nn.Module):
    kind = "TanimotoComplementLoss"

    def __init__(self, smooth: float = 1e-5, depth: int = 5, transform_logits: bool = False,
                 one_hot_targets: bool = True):
        super().__init__()
        if transform_logits:
            raise NotImplementedError("transform_logits=True is unused by cultionet's LOSS_DICT")
        self.smooth, self.depth, self.one_hot_targets = smooth, depth, one_hot_targets

    def forward(self, inputs: torch.Tensor, targets: torch.Tensor, mask: T.Optional[torch.Tensor] = None):
        kw = _kernel_args(inputs, targets, mask, self.one_hot_targets)
        return tanimoto_autograd(inputs, loss_kind=E.LOSS_KINDS[self.kind], smooth=self.smooth, depth=self.depth, **kw)


class TanimotoComplementLoss(_TanimotoBase):
    """losses.py:134-218."""

    kind = "TanimotoComplementLoss"


class TanimotoDistLoss(_TanimotoBase):
    """losses.py:251-340."""

    kind = "TanimotoDistLoss"

    def __init__(self, smooth: float = 1e-5, transform_logits: bool = False, one_hot_targets: bool = True):
        super().__init__(smooth=smooth, depth=5, transform_logits=transform_logits, one_hot_targets=one_hot_targets)


class CombinedLoss(nn.Module):
    """losses.py:62-100 restricted to the pair used by LOSS_DICT[TanimotoCombined] (one fused kernel)."""

    def __init__(self, losses: T.List[T.Callable]):
        super().__init__()
        kinds = sorted(type(l).__name__ for l in losses)
        if kinds != ["TanimotoComplementLoss", "TanimotoDistLoss"]:
            raise NotImplementedError("CombinedLoss supports [TanimotoDistLoss, TanimotoComplementLoss]")
        self.losses = losses
        self.one_hot_targets = losses[0].one_hot_targets

    def forward(self, inputs, targets, mask=None):
        kw = _kernel_args(inputs, targets, mask, self.one_hot_targets)
        return tanimoto_autograd(inputs, loss_kind=E.LOSS_KINDS["TanimotoCombined"], **kw)
