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


MAX_PLANS = 4  # per model: (input shape, precision) slots kept alive


class ForwardPlan:
    __slots__ = ("calls", "pool", "x", "outputs", "key", "keep")

    def __init__(self):
        self.calls: T.List[T.Tuple[int, T.Any, tuple]] = []  # (0, C entry point, args) | (1, python stream op, args)
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
    return (tuple(x.shape), x.dtype, bool(bf16), store.uid, store.version, E._bn_stats_epoch, _bn_signature(model),
            torch.cuda.current_stream(x.device).cuda_stream, E._EVAL_FUSION, E.workspace_epoch(),
            E.branch_streams_allowed())


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
        if len(plans) > 1:  # most recently used last (the cache is a small LRU, see below)
            plans[slot] = plans.pop(slot)
        if x.data_ptr() != plan.x.data_ptr():
            plan.x.copy_(x)
        for kind, fn, args in plan.calls:
            if kind:  # a stream operation of the recorded forward (event record / wait of engine.spawn / join)
                fn(*args)
                continue
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
        plan.calls.append((0, fn, args))
        return rc

    epoch0 = E.workspace_epoch()
    with torch.cuda.use_mem_pool(plan.pool):
        plan.x = torch.empty_like(x)
        plan.x.copy_(x)
        _lib.call = recording
        prev_rec = E._recorder
        E._recorder = plan.calls  # engine._py_op appends (1, fn, args): stream operations in program order
        try:
            outs = run(plan.x)
        finally:
            _lib.call = orig
            E._recorder = prev_rec
        plan.outputs = dict(outs)
    if E.workspace_epoch() != epoch0:  # a scratch buffer moved while recording: stale pointers, do not keep the plan
        return plan.outputs
    plan.key = _key(model, store, x, bf16)  # (recording may have refreshed packed weights: the key after it)
    plans.pop(slot, None)
    plans[slot] = plan
    # a plan owns a private memory pool with the whole activation set of its batch (GBs for a packed window batch):
    # keep the MAX_PLANS most recently used shapes (full batch + ragged tail, both precisions), drop the rest
    while len(plans) > MAX_PLANS:
        plans.pop(next(iter(plans)))
    return plan.outputs


# ---------------------------------------------------------------------------------------------------------------------
# The native TRAINING step (forward + loss + backward of HipTrainer) as a launch plan.
#
# At the reference's defaults -- batch 4, precision "16-mixed" (scripts/args.yml:248-254, model.py:168-186) -- the step
# is ~2.5-4 ms of GPU work behind ~9.8 ms of Python (the tape, ~470 C-ABI calls, ~2400 small torch calls): host-bound.
# The same recording trick as above, with what a training step adds:
#   * two streams: the weight-gradient side stream's event record / wait calls and the final join are part of the plan
#     (engine._py_op), in program order;
#   * NO buffer is reused inside a recorded step. The caching allocator makes cross-stream reuse safe EAGERLY by looking
#     at events when a block is freed; a replay has no such check, so a block that held a weight gradient's operand and
#     was later handed to the compute stream would be a race. While recording, every tensor torch hands out is kept alive
#     (the plan's ``keep`` list is an allocation sink of the engine allocator: engine.holding_allocations), so each kernel of the step has buffers of its own; the price
#     is memory (the sum of a step's allocations instead of its peak), which is what 288 GB are for;
#   * the optimizer (clip + AdamW: two launches with per-step scalars) stays outside the plan.
# Valid for one (batch shapes, precision, stream, store, loss kind, scratch-buffer epoch); a communicator (per-bucket
# collectives) falls back to the eager step. Dropout > 0 replays: the per-step part of every mask seed is a device word
# that the plan's first launch bumps (engine.begin_rng_step), the recorded seed arguments are the per-call constants. Bit-exact w.r.t. the eager step except where the eager step itself
# is not (float-atomic parameter-gradient sums): tests/test_replay_train_gpu.py.
# ---------------------------------------------------------------------------------------------------------------------
class StepPlan:
    __slots__ = ("ops", "pool", "keep", "inputs", "outputs", "key", "n_calls", "sums")

    def __init__(self):
        self.ops: T.List[T.Tuple[int, T.Any, tuple]] = []
        self.pool = None
        self.keep: T.List[torch.Tensor] = []
        self.inputs: T.Optional[T.Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None
        self.outputs: T.Optional[T.Dict[str, torch.Tensor]] = None
        self.key = None
        self.n_calls = 0
        self.sums = None  # the store's deferred-slice-sum state (engine._SliceSums) whose host table this plan rewrites


def step_key(trainer, batch) -> tuple:
    x, y, bd = batch.x, batch.y, batch.bdist
    return (tuple(x.shape), x.dtype, tuple(y.shape), y.dtype, tuple(bd.shape), bd.dtype, trainer.bf16, trainer.store.uid,
            str(trainer.lit.loss_name), torch.cuda.current_stream(x.device).cuda_stream, E._OVERLAP_WGRAD,
            E.workspace_epoch(), E.branch_streams_allowed())


def record_step(trainer, batch, eager: T.Callable) -> StepPlan:
    """Record ``eager(plan_batch)`` (HipTrainer's forward + loss + backward) into a StepPlan. The caller has run at least
    two eager steps before (every lazily created workspace, packed weight and stream exists)."""
    from .data import Data

    plan = StepPlan()
    plan.pool = torch.cuda.MemPool()
    lib = _lib.load()
    orig_call = _lib.call
    ops = plan.ops
    epoch0 = E.workspace_epoch()

    def recording(name: str, *args):
        fn = getattr(lib, name)
        rc = fn(*args)
        if rc != 0:
            raise _lib.HipKernelError(f"{name} failed: {_lib.ERRORS.get(rc, rc)}")
        ops.append((0, fn, args))
        return rc

    with torch.cuda.use_mem_pool(plan.pool):
        # the plan's inputs in the CANONICAL form the loss / forward kernels consume (x fp32, labels int64, distances
        # fp32, all dense): `dst.copy_(src)` of every replayed step then does any cast / re-striding on the device, and no
        # torch conversion kernel (y.long(), .contiguous() -- which a launch plan cannot see) sits between the plan's
        # input buffers and the recorded launches
        dev = batch.x.device
        plan.inputs = (torch.empty(tuple(batch.x.shape), dtype=torch.float32, device=dev),
                       torch.empty(tuple(batch.y.shape), dtype=torch.int64, device=dev),
                       torch.empty(tuple(batch.bdist.shape), dtype=torch.float32, device=dev))
        for dst, src in zip(plan.inputs, (batch.x, batch.y, batch.bdist)):
            dst.copy_(src)
        pb = Data(x=plan.inputs[0], y=plan.inputs[1], bdist=plan.inputs[2])
        trainer.store.bump()  # the batched weight re-pack must be PART of the plan even if nothing changed since the last pack
        _lib.call = recording
        E._recorder = ops
        try:
            with E.holding_allocations(plan.keep):  # every buffer the engine hands out lives as long as the plan
                eager(pb)
        finally:
            E._recorder = None
            _lib.call = orig_call
    plan.outputs = dict(trainer.last_outputs)
    plan.sums = getattr(trainer.store, "_slice_sums", None)
    if plan.sums is not None:
        plan.sums.writer = plan
    # a scratch buffer that was (re)allocated WHILE recording leaves stale pointers in the earlier entries: no key, so
    # the caller drops this plan and records again once the buffers have settled
    plan.key = step_key(trainer, batch) if E.workspace_epoch() == epoch0 else None
    plan.n_calls = sum(1 for o in ops if o[0] == 0)
    return plan


def replay_step(plan: StepPlan, batch) -> None:
    for dst, src in zip(plan.inputs, (batch.x, batch.y, batch.bdist)):
        if dst.data_ptr() != src.data_ptr():
            dst.copy_(src)
    st = plan.sums
    if st is not None and st.writer is not plan:
        # ANOTHER plan (or an eager pass) wrote the slice-sum table last and its upload may still be in flight: this plan's
        # weight gradients are about to rewrite the host table with different records. (The same plan replayed back to
        # back rewrites identical bytes -- no wait, the host keeps running ahead.)
        if not st.upload_ev.query():
            st.upload_ev.synchronize()
        st.writer = plan
    if st is not None:
        # the plan's recorded cn_slice_sums_run calls upload their ranges themselves: whatever an eager pass believed the
        # device table to hold is void from here on
        st.uploaded.clear()
    for kind, fn, args in plan.ops:
        if kind == 0:
            rc = fn(*args)
            if rc != 0:
                raise _lib.HipKernelError(f"replayed launch failed: {_lib.ERRORS.get(rc, rc)}")
        else:
            fn(*args)
    E._note_bn_update(True)  # the recorded BatchNorm launches updated running statistics: invalidate eval-mode folds
