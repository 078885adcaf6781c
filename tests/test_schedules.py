"""The native OneCycle schedule against torch.optim.lr_scheduler.OneCycleLR (the reference's default scheduler,
models/lightning.py:657-664), including the beta1 cycling torch applies to AdamW."""
import pytest
import torch

from cultionet_amd.schedules import ConstantLR, OneCycleLR


@pytest.mark.parametrize("total,max_lr", [(10, 0.01), (37, 0.003), (200, 0.01)])
def test_onecycle_matches_torch(total, max_lr):
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.AdamW([p], lr=max_lr, betas=(0.9, 0.98))
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=max_lr, total_steps=total)
    mine = OneCycleLR(max_lr, total)
    for k in range(1, total + 1):
        lr, b1 = mine(k)
        g = opt.param_groups[0]
        assert abs(lr - g["lr"]) <= 1e-12 + 1e-9 * abs(g["lr"]), (k, lr, g["lr"])
        assert abs(b1 - g["betas"][0]) <= 1e-12, (k, b1, g["betas"])
        p.grad = torch.ones(3)
        opt.step()
        sch.step()


def test_constant():
    assert ConstantLR(0.01)(5) == (0.01, 0.9)
