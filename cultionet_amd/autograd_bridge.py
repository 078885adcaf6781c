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
        with E.using_store(store), E.recording(True) as tape, E.mixed_precision(_autocast_bf16(model)):
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


# torch >= 2.4 takes the device type; older releases only have the CUDA-specific spellings. Probed ONCE at import, no
# blanket try/except around the query itself: a failure to read the autocast state must not silently mean "fp32".
try:
    torch.is_autocast_enabled("cuda")
    _AUTOCAST_NEW_API = True
except TypeError:  # pragma: no cover - older torch
    _AUTOCAST_NEW_API = False
_warned: T.Set[str] = set()


def _warn_once(key: str, msg: str) -> None:
    if key not in _warned:
        _warned.add(key)
        import warnings

        warnings.warn(msg, stacklevel=3)


def _ambient_autocast() -> T.Optional[torch.dtype]:
    """dtype of the ambient CUDA autocast region, or None outside one."""
    if _AUTOCAST_NEW_API:
        return torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None
    return torch.get_autocast_gpu_dtype() if torch.is_autocast_enabled() else None  # pragma: no cover


def _autocast_bf16(model=None) -> bool:
    """Does this forward run the mixed-precision (bf16 NHWC, MFMA) path?

    1. An explicit ``model.precision`` ("32-true" / "bf16-mixed" / "16-mixed"; CultionetLitModel.precision forwards to
       it) wins -- the choice is then independent of whatever autocast region the caller happens to be in.
    2. Otherwise lightning.Trainer(precision="16-mixed" / "bf16-mixed") is recognised by its torch.autocast region
       (model.py:168-186). fp16 autocast is served by the bf16 path (MI355X has no reason to prefer fp16, and the
       GradScaler of "16-mixed" is harmless on it); that substitution and the implicit selection are announced once.
    """
    if model is not None and not getattr(model, "mixed_precision_ok", True):
        explicit = getattr(model, "precision", None)
        if explicit in ("bf16-mixed", "16-mixed") or (explicit is None and _ambient_autocast() is not None):
            _warn_once("width", "cultionet_amd: the mixed-precision path needs channel counts that are multiples of 8 "
                                "(hidden_channels % 8 == 0); this model runs in fp32")
        return False
    explicit = getattr(model, "precision", None) if model is not None else None
    if explicit is not None:
        if explicit in ("32-true", "32"):
            return False
        if explicit in ("bf16-mixed", "16-mixed"):
            return True
        raise ValueError(f"unsupported precision {explicit!r} (32-true | bf16-mixed | 16-mixed)")
    dt = _ambient_autocast()
    if dt is None:
        return False
    if dt == torch.bfloat16:
        _warn_once("bf16", "cultionet_amd: torch.autocast(bfloat16) is active -> TowerUNet runs its bf16 mixed-precision "
                           "HIP path (set model.precision = '32-true' to force fp32)")
        return True
    if dt == torch.float16:
        _warn_once("fp16", "cultionet_amd: fp16 autocast ('16-mixed') is served by the bf16 mixed-precision HIP path on "
                           "MI355X (same exponent range as fp32: the GradScaler is a no-op in effect)")
        return True
    _warn_once("other", f"cultionet_amd: autocast dtype {dt} has no mixed-precision path; running fp32")
    return False


def run_towerunet(model, x: torch.Tensor) -> T.Dict[str, torch.Tensor]:
    store = model.param_store()
    if torch.is_grad_enabled() and any(p.requires_grad for p in store.params):
        d, e, c = _TowerUNetFn.apply(model, x, *store.params)
        return {_KEYS[0]: d, _KEYS[1]: e, _KEYS[2]: c}
    bf16 = _autocast_bf16(model)

    def eager(xx: torch.Tensor) -> T.Dict[str, torch.Tensor]:
        with E.using_store(store), E.recording(False), E.mixed_precision(bf16):
            outs = model.forward_vars(model.input_var(xx))
        return {k: outs[k].t for k in _KEYS}

    if getattr(model, "replay", False) and not model.training and not torch.is_grad_enabled():
        from . import replay  # inference through a recorded launch plan (cultionet_amd/replay.py)

        return replay.forward(model, x, bf16, eager)
    return eager(x)


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
