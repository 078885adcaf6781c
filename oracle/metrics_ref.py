"""torchmetrics scorers used by the reference's _shared_eval_step -- CPU restatement, TEST INFRASTRUCTURE ONLY.

/root/reference/src/cultionet/models/lightning.py:562-577 builds four torchmetrics objects; torchmetrics is a
third-party dependency that is not installed here (nor vendored under /root/reference), so their published
semantics (torchmetrics 1.x) are restated with plain torch and used (a) as the ``torchmetrics`` stub when the
reference is imported to generate fixtures and (b) as the checker of the HIP metrics kernel:

  MeanAbsoluteError / MeanSquaredError   mean |p - t| / mean (p - t)^2 over all elements;
  FBetaScore(task="multiclass", num_classes=2, beta=2.0)
      the task wrapper's default ``average="micro"``: tp = sum of the confusion-matrix diagonal, fp = fn = N - tp,
      F_beta = (1+b^2) tp / ((1+b^2) tp + b^2 fn + fp) = tp / N, i.e. plain accuracy whatever beta is;
  MatthewsCorrCoef(task="multiclass", num_classes=2)
      from the confusion matrix C: (c*s - sum_k p_k t_k) / sqrt((s^2 - sum p_k^2) (s^2 - sum t_k^2)) with
      c = trace, s = total, p_k / t_k = predicted / true counts. Degenerate 2x2 cases as torchmetrics >= 1.0
      (`_matthews_corrcoef_reduce`; setup.cfg pins torchmetrics>=1.3): every prediction right -> 1, every prediction
      wrong -> -1, and with an empty marginal (denominator 0) the eps-regularised ratio
      sqrt(eps) * ((tp + tn) - (fp + fn)) / sqrt((tp+fp+eps)(tp+fn+eps)(tn+fp+eps)(tn+fn+eps)), eps = float32 epsilon.
      (Rounds 1-2 returned 0 there, which is torchmetrics < 1.0.) tests/test_metrics_ref.py holds hand-computed answers.

``forward`` returns the value of the current batch (what ``scorer(preds, target)`` returns in the reference).
Parity status: restatement-checked (a real torchmetrics is used instead whenever it is importable).
"""
from __future__ import annotations

import torch
import torch.nn as nn


class MeanAbsoluteError(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, preds, target):
        return (preds.float() - target.float()).abs().mean()


class MeanSquaredError(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, preds, target):
        return ((preds.float() - target.float()) ** 2).mean()


def _confmat(preds, target, num_classes):
    idx = target.long() * num_classes + preds.long()
    return torch.bincount(idx.flatten(), minlength=num_classes * num_classes).reshape(num_classes, num_classes).double()


class FBetaScore(nn.Module):
    def __init__(self, task="multiclass", num_classes=2, beta=1.0, average="micro", **k):
        super().__init__()
        assert task == "multiclass" and average == "micro"
        self.num_classes, self.beta = num_classes, beta

    def forward(self, preds, target):
        c = _confmat(preds, target, self.num_classes)
        tp = c.diag().sum()
        fp = fn = c.sum() - tp
        b2 = self.beta ** 2
        return ((1 + b2) * tp / ((1 + b2) * tp + b2 * fn + fp)).float()


class MatthewsCorrCoef(nn.Module):
    def __init__(self, task="multiclass", num_classes=2, **k):
        super().__init__()
        assert task == "multiclass"
        self.num_classes = num_classes

    def forward(self, preds, target):
        c = _confmat(preds, target, self.num_classes)
        if self.num_classes == 2:
            tn, fp, fn, tp = (float(v) for v in c.reshape(-1))
            if tp + tn != 0 and fp + fn == 0:
                return torch.ones((), dtype=torch.float32)
            if tp + tn == 0 and fp + fn != 0:
                return -torch.ones((), dtype=torch.float32)
        tk, pk = c.sum(1), c.sum(0)
        cc, s = c.diag().sum(), c.sum()
        num = cc * s - (tk * pk).sum()
        den = (s ** 2 - (pk ** 2).sum()) * (s ** 2 - (tk ** 2).sum())
        if float(den) == 0:
            if self.num_classes != 2:
                return torch.zeros((), dtype=torch.float32)
            eps = float(torch.finfo(torch.float32).eps)
            num = torch.tensor(eps ** 0.5 * ((tp + tn) - (fp + fn)), dtype=torch.float64)
            den = torch.tensor((tp + fp + eps) * (tp + fn + eps) * (tn + fp + eps) * (tn + fn + eps), dtype=torch.float64)
        return (num / torch.sqrt(den)).float()
