"""Known answers for oracle/metrics_ref.MatthewsCorrCoef (the checker of the HIP metrics kernel): the regular formula
and the degenerate 2x2 branches of torchmetrics >= 1.0 (`_matthews_corrcoef_reduce`), computed by hand."""
import math

import torch


def _mcc(tp, fp, fn, tn):
    from oracle import metrics_ref as M

    preds = torch.tensor([1] * tp + [1] * fp + [0] * fn + [0] * tn)
    target = torch.tensor([1] * tp + [0] * fp + [1] * fn + [0] * tn)
    return float(M.MatthewsCorrCoef()(preds, target))


def test_regular_case():
    # tp 6, fp 1, fn 2, tn 3: (18 - 2) / sqrt(7 * 8 * 4 * 5)
    assert abs(_mcc(6, 1, 2, 3) - 16 / math.sqrt(1120)) < 1e-6


def test_all_right_and_all_wrong():
    assert _mcc(4, 0, 0, 5) == 1.0
    assert _mcc(0, 0, 0, 7) == 1.0   # a background chip predicted as background: no positives anywhere
    assert _mcc(0, 3, 2, 0) == -1.0


def test_empty_marginal_uses_the_eps_ratio():
    # no positive prediction, truth mixed: tp 0, fp 0, fn 3, tn 5 -> sqrt(eps) * (5 - 3) / sqrt(eps * 3 * 5 * 8)
    assert abs(_mcc(0, 0, 3, 5) - 2 / math.sqrt(120)) < 1e-5
    # no true positive, some predicted: tp 0, fn 0, fp 2, tn 6 -> sqrt(eps) * (6 - 2) / sqrt(2 * eps * 8 * 6)
    assert abs(_mcc(0, 2, 0, 6) - 4 / math.sqrt(96)) < 1e-5
    # more wrong than right with an empty marginal: negative
    assert _mcc(0, 0, 5, 1) < 0
