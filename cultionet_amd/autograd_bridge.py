"""torch.autograd <-> engine bridge: lets ``loss.backward()`` of a Lightning fit loop drive the HIP tape.

The engine has its own reverse-mode tape (cultionet_amd.engine). For callers that live in torch.autograd
(lightning.Trainer.fit -> training_step -> loss.backward(), torch DDP hooks, torch optimizers) the whole
TowerUNet is presented as ONE autograd.Function: forward runs the engine forward and keeps the tape,
backward seeds the three output gradients, replays the tape and hands the parameter gradients back to
autograd (views of the flat gradient buffer, which is handed over whole, so DDP hooks fire as usual).
"""
from __future__ import annotations

import typing as T

import torch

from . import engine as E
from .enums import InferenceNames

_KEYS = (InferenceNames.DISTANCE, InferenceNames.EDGE, InferenceNames.CROP)


class _TowerUNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        store = model.param_store()
        with E.using_store(store), E.recording(True) as tape, E.mixed_precision(_autocast_bf16()):
            outs = model.forward_vars(model.input_var(x))
        ctx.model, ctx.tape, ctx.outs, ctx.store = model, tape, outs, store
        ctx.n_params = len(params)
        return tuple(outs[k].t for k in _KEYS)

    @staticmethod
    def backward(ctx, *grads):
        store = ctx.store
        store.zero_grad()
        for k, g in zip(_KEYS, grads):
            ctx.outs[k].grad = g.contiguous() if g is not None else None
        with E.using_store(store):
            ctx.tape.backward()
        # hand the flat gradient buffer itself to autograd (AccumulateGrad keeps the views as p.grad) and give the store
        # a fresh one: no 42 MB copy per backward; the next backward's zero_grad() fills the new buffer
        flat = store.flat_grad
        store.flat_grad = torch.empty_like(flat)
        pg = tuple(flat[o:o + p.numel()].view(p.shape) for p, o in zip(store.params, store.offsets))
        ctx.tape = ctx.outs = None
        return (None, None) + pg


def _autocast_bf16() -> bool:
    """lightning.Trainer(precision="16-mixed" / "bf16-mixed") runs the step under torch.autocast: that selects the
    bf16 MFMA path here (MI355X has no reason to prefer fp16; the GradScaler of "16-mixed" is harmless on it)."""
    try:
        return bool(torch.is_autocast_enabled("cuda")) and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16)
    except Exception:  # pragma: no cover
        return False


def run_towerunet(model, x: torch.Tensor) -> T.Dict[str, torch.Tensor]:
    store = model.param_store()
    if torch.is_grad_enabled() and any(p.requires_grad for p in store.params):
        d, e, c = _TowerUNetFn.apply(model, x, *store.params)
        return {_KEYS[0]: d, _KEYS[1]: e, _KEYS[2]: c}
    with E.using_store(store), E.recording(False), E.mixed_precision(_autocast_bf16()):
        outs = model.forward_vars(model.input_var(x))
    return {k: outs[k].t for k in _KEYS}


class _TanimotoFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, kw):
        with E.recording(True) as tape:
            pv = E.Var(pred.contiguous(), True)
            loss = E.tanimoto_loss(pv, **kw)
        ctx.tape, ctx.pv = tape, pv
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        ctx.tape.backward()
        dp = ctx.pv.grad
        ctx.tape = ctx.pv = None
        return dp * g, None


def tanimoto_autograd(pred: torch.Tensor, **kw) -> torch.Tensor:
    """Tanimoto loss (HIP kernels) as a differentiable torch scalar."""
    if torch.is_grad_enabled() and pred.requires_grad:
        return _TanimotoFn.apply(pred, kw)
    with E.recording(False):
        return E.tanimoto_loss(E.Var(pred.contiguous(), False), **kw).view(())
