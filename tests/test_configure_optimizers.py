"""configure_optimizers of the Lightning surface (reference: models/lightning.py:611-683): every optimizer / scheduler choice
the reference offers builds the same torch object with the same hyper-parameters, OneCycleLR alone steps per batch, and the
two refusals raise the reference's NameErrors. CPU only (parameter containers, no kernels)."""
import types

import pytest
import torch

EXPECT_OPT = {
    "Adam": (torch.optim.Adam, {"betas": (0.9, 0.999)}),
    "AdamW": (torch.optim.AdamW, {"betas": (0.9, 0.98)}),
    "RAdam": (torch.optim.RAdam, {"betas": (0.9, 0.99), "decoupled_weight_decay": True}),
    "SGD": (torch.optim.SGD, {"momentum": 0.9}),
}
EXPECT_SCHED = {
    "CosineAnnealingLR": (torch.optim.lr_scheduler.CosineAnnealingLR, {"T_max": 20, "eta_min": 1e-5}, "epoch"),
    "ExponentialLR": (torch.optim.lr_scheduler.ExponentialLR, {"gamma": 0.5}, "epoch"),
    "StepLR": (torch.optim.lr_scheduler.StepLR, {"gamma": 0.5, "step_size": 7}, "epoch"),
    "OneCycleLR": (torch.optim.lr_scheduler.OneCycleLR, {}, "step"),
}


def _lit(**kw):
    from cultionet_amd.lightning import CultionetLitModel

    return CultionetLitModel(in_channels=3, in_time=12, hidden_channels=8, **kw)


@pytest.mark.parametrize("opt", list(EXPECT_OPT))
@pytest.mark.parametrize("sched", list(EXPECT_SCHED))
def test_every_choice_of_the_reference(opt, sched):
    lit = _lit(optimizer=opt, lr_scheduler=sched, learning_rate=0.02, weight_decay=3e-3, eps=1e-4, steplr_step_size=7)
    lit.__dict__["trainer"] = types.SimpleNamespace(max_epochs=3, estimated_stepping_batches=11)  # what OneCycleLR reads
    try:
        out = lit.configure_optimizers()
    except (AttributeError, RuntimeError):  # a LightningModule base that guards `.trainer`: bind the property's backing field
        lit._trainer = lit.__dict__.pop("trainer")
        out = lit.configure_optimizers()
    o, cfg = out["optimizer"], out["lr_scheduler"]
    cls, fixed = EXPECT_OPT[opt]
    assert type(o) is cls
    d = o.defaults
    assert d["lr"] == pytest.approx(0.02) or sched == "OneCycleLR"  # (OneCycleLR rewrites lr to its initial value)
    if opt != "Adam":
        assert d["weight_decay"] == pytest.approx(3e-3)
    if opt != "SGD":
        assert d["eps"] == pytest.approx(1e-4)
    for k, v in fixed.items():
        assert d[k] == v, (k, d[k])
    assert sum(p.numel() for g in o.param_groups for p in g["params"]) == sum(p.numel() for p in lit.cultionet_model.parameters())
    scls, sfixed, interval = EXPECT_SCHED[sched]
    s = cfg["scheduler"]
    assert type(s) is scls and cfg["interval"] == interval
    assert (cfg["name"], cfg["monitor"], cfg["frequency"]) == ("lr_sch", "val_score", 1)
    for k, v in sfixed.items():
        assert getattr(s, k) == pytest.approx(v), k
    if sched == "OneCycleLR":
        assert s.total_steps == 3 * 11


def test_refusals_are_the_references_name_errors():
    with pytest.raises(NameError, match="AdamW"):
        _lit(optimizer="Lion").configure_optimizers()
    with pytest.raises(NameError, match="not implemented"):
        _lit(lr_scheduler="Plateau").configure_optimizers()
