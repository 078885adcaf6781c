"""Launch-plan replay of the inference forward.

The eval forward of a [n,C,T,S,S] window batch is ~120 C-ABI calls whose arguments never change from one batch to the
next (static shapes, running statistics, same stream). Issued eagerly they cost ~2.5 ms of Python per forward -- more
than the GPU needs for the reference CLI's default predict batch of 4 windows (scripts/args.yml:248-254), so the
sliding-window predictor was bounded by the host's launch rate. A hipGraph is the textbook answer; on this stack it is
the wrong one (a captured step REPLAYS SLOWER than eager launches -- every node pays a
barrier packet). What is replayed here instead is the list of C-ABI calls itself:

  * RECORD: the forward runs once, normally, inside a private torch memory pool (``torch.cuda.use_mem_pool``); every
    ``_lib.call`` is appended to the plan as (foreign function, argument tuple) -- device pointers included, which
    stay valid because the pool (and with it every intermediate buffer, at its address) is kept alive by the plan;
  * REPLAY: ``for fn, args in plan: fn(*args)`` -- no Python module tree, no tensor allocation, no shape logic.

A plan is valid for one (input shape, precision, stream, parameter version, BatchNorm-statistics version); anything
else records a new one. The input is copied into the plan's own input buffer (one copy kernel) unless the caller
already writes there (``plan_input``). Outputs are the plan's buffers: consume them before the next replay on the same
stream (the predictor's stitch kernel does). Inference only -- a training step is a tape, not a list.
"""
from __future__ import annotations

import typing as T

import torch

from . import _lib
from . import engine as E


class ForwardPlan:
    __slots__ = ("calls", "pool", "x", "outputs", "key", "keep")

    def __init__(self):
        self.calls: T.List[T.Tuple[T.Any, tuple]] = []
        self.pool = None
        self.x: T.Optional[torch.Tensor] = None
        self.outputs: T.Optional[T.Dict[str, torch.Tensor]] = None
        self.key = None
        self.keep: T.List[T.Any] = []


def _bn_signature(model) -> int:
    sig = model.__dict__.get("_cn_bn_buffers")
    if sig is None:
        sig = [b for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)
               for b in (m.running_mean, m.running_var) if b is not None]
        model.__dict__["_cn_bn_buffers"] = sig
    return sum(b._version for b in sig)


def _key(model, store, x: torch.Tensor, bf16: bool):
    store.refresh()
    return (tuple(x.shape), x.dtype, bool(bf16), id(store), store.version, E._bn_stats_epoch, _bn_signature(model),
            torch.cuda.current_stream(x.device).cuda_stream, E._EVAL_FUSION)


def plan_input(model, shape: T.Sequence[int], bf16: bool, device) -> T.Optional[torch.Tensor]:
    """The input buffer of the current plan for ``shape`` (None before the first forward): a producer that writes its
    batch there (cn_window_chips_f32 in the sliding-window predictor) saves the copy."""
    plans = model.__dict__.get("_cn_plans") or {}
    p = plans.get((tuple(shape), bool(bf16)))
    return p.x if p is not None else None


def forward(model, x: torch.Tensor, bf16: bool, run: T.Callable[[torch.Tensor], T.Dict[str, torch.Tensor]]):
    """Eval forward of ``model`` on ``x`` through a recorded launch plan. ``run(x)`` is the eager forward (used once per
    plan, under the recorder)."""
    store = model.param_store()
    plans = model.__dict__.setdefault("_cn_plans", {})
    slot = (tuple(x.shape), bool(bf16))
    key = _key(model, store, x, bf16)
    plan = plans.get(slot)
    if plan is not None and plan.key == key:
        if x.data_ptr() != plan.x.data_ptr():
            plan.x.copy_(x)
        for fn, args in plan.calls:
            rc = fn(*args)
            if rc != 0:
                raise _lib.HipKernelError(f"replayed launch failed: {_lib.ERRORS.get(rc, rc)}")
        return plan.outputs
    # ---- record ----
    run(x)  # warm every lazily created workspace / packed weight OUTSIDE the pool (they outlive the plan)
    plan = ForwardPlan()
    plan.pool = torch.cuda.MemPool()
    lib = _lib.load()
    orig = _lib.call

    def recording(name: str, *args):
        fn = getattr(lib, name)
        rc = fn(*args)
        if rc != 0:
            raise _lib.HipKernelError(f"{name} failed: {_lib.ERRORS.get(rc, rc)}")
        plan.calls.append((fn, args))
        return rc

    with torch.cuda.use_mem_pool(plan.pool):
        plan.x = torch.empty_like(x)
        plan.x.copy_(x)
        _lib.call = recording
        try:
            outs = run(plan.x)
        finally:
            _lib.call = orig
        plan.outputs = dict(outs)
    plan.key = _key(model, store, x, bf16)  # (recording may have refreshed packed weights: the key after it)
    plans[slot] = plan
    return plan.outputs
