"""Learning-rate schedules of the native training step (``HipTrainer``).

The reference's default is ``torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=learning_rate, epochs=...,
steps_per_epoch=...)`` stepped once per optimizer step (/root/reference/src/cultionet/models/lightning.py:657-664,
``interval="step"``). With its defaults (``pct_start=0.3``, cosine annealing, ``div_factor=25``,
``final_div_factor=1e4``, ``cycle_momentum=True``, ``base_momentum=0.85``, ``max_momentum=0.95``) it drives BOTH the
learning rate and AdamW's beta1 (torch cycles ``betas[0]`` for Adam-type optimizers), so the configured
``betas=(0.9, 0.98)`` never reaches the update: beta1 starts at 0.95, dips to 0.85 at the LR peak and returns to 0.95.
The fused AdamW kernel takes (lr, beta1) per launch, so the schedule is two scalars computed on the host.
"""
from __future__ import annotations

import math
import typing as T


class ConstantLR:
    def __init__(self, lr: float, beta1: float = 0.9):
        self.lr, self.beta1 = float(lr), float(beta1)

    def __call__(self, step: int) -> T.Tuple[float, float]:
        return self.lr, self.beta1


class OneCycleLR:
    """(lr, beta1) of torch's OneCycleLR at optimizer step ``step`` (1-based: the k-th ``optimizer.step()`` runs with
    the values the scheduler set after k-1 ``scheduler.step()`` calls)."""

    def __init__(self, max_lr: float, total_steps: int, pct_start: float = 0.3, div_factor: float = 25.0,
                 final_div_factor: float = 1e4, base_momentum: float = 0.85, max_momentum: float = 0.95):
        if total_steps <= 0:
            raise ValueError("Expected positive integer total_steps")
        if not 0.0 <= pct_start <= 1.0:
            raise ValueError("Expected float between 0 and 1 pct_start")
        self.total_steps = int(total_steps)
        self.max_lr = float(max_lr)
        self.initial_lr = self.max_lr / div_factor
        self.min_lr = self.initial_lr / final_div_factor
        self.base_momentum, self.max_momentum = float(base_momentum), float(max_momentum)
        self.phase1_end = float(pct_start * self.total_steps) - 1.0
        self.phase2_end = float(self.total_steps) - 1.0

    @staticmethod
    def _cos(start: float, end: float, pct: float) -> float:
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    def __call__(self, step: int) -> T.Tuple[float, float]:
        n = step - 1  # scheduler.step() calls made before this optimizer step
        if n > self.total_steps:
            raise ValueError(f"Tried to step {n} times. The specified number of total steps is {self.total_steps}")
        if n <= self.phase1_end or self.phase1_end == self.phase2_end:
            pct = n / self.phase1_end if self.phase1_end > 0 else 1.0
            return (self._cos(self.initial_lr, self.max_lr, pct),
                    self._cos(self.max_momentum, self.base_momentum, pct))
        pct = (n - self.phase1_end) / (self.phase2_end - self.phase1_end)
        return (self._cos(self.max_lr, self.min_lr, pct), self._cos(self.base_momentum, self.max_momentum, pct))
