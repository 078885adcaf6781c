"""Execution engine of cultionet_amd: a reverse-mode tape whose every node is a HIP kernel launch.

PyTorch is used for device memory (caching allocator), streams and parameter containers only;
all arithmetic runs in libcultionet_hip.so through the C ABI (cultionet_amd._lib). There is no
CPU path: calling any op without the HIP library or with CPU tensors raises.

Design (MI355X-first, not torch.autograd):
  * ``Var`` = a device buffer (+ its gradient buffer). Buffers are NCHW fp32; a Var may be a channel
    slice of a bigger buffer (batch stride != C*H*W), which every kernel supports natively.
  * forward ops append a closure to the tape; ``Tape.backward()`` walks it in reverse. Gradient
    buffers are written with beta=0 by the first producer and accumulated in-kernel (beta=1) by
    later ones, or aliased when an op is the identity on its gradient (residual adds, concat
    slices): no separate add/copy kernels.
  * parameters live in ONE flat fp32 buffer and so do their gradients (``ParamStore``): the optimizer
    is one fused launch and data-parallel all-reduce buckets are plain slices of the flat gradient.
"""
from __future__ import annotations

import threading
import contextlib
import os
import typing as T

import torch

from . import _lib

_state = threading.local()


# ---- launch-plan recording (cultionet_amd/replay.py) ----------------------------------------------------------------
# While a plan is being recorded, every stream / event operation and every host-side torch call that launches work is
# routed through _py_op so that the plan can repeat it; C-ABI calls are captured by a recording _lib.call.
_recorder: T.Optional[T.List] = None


def _py_op(fn: T.Callable, *args) -> None:
    """Run ``fn(*args)`` now and, while a launch plan is being recorded, append it to the plan."""
    fn(*args)
    if _recorder is not None:
        _recorder.append((1, fn, args))


def _main_stream() -> "torch.cuda.Stream":
    """torch's current stream, looked up once per ``using_store`` scope (a forward, a backward or a whole native step:
    the stream cannot change inside one) -- ``torch.cuda.current_stream()`` costs ~11 us, and at ~420 launches per step
    that was a third of the host's enqueue time."""
    ms = getattr(_state, "main_stream", None)
    return ms if ms is not None else torch.cuda.current_stream()


def _stream() -> int:
    """HIP stream handle the next launch goes to: the side stream inside a ``side_stream`` block, else torch's
    current stream."""
    ov = getattr(_state, "stream_override", None)
    if ov is not None:
        return ov
    h = getattr(_state, "main_handle", None)
    return h if h is not None else torch.cuda.current_stream().cuda_stream


def _check(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("cultionet_amd ops need device tensors (MI355X); there is no CPU fallback")
    if t.dtype == torch.bfloat16:
        # mixed-precision path: bf16 activations are NHWC (logical shape [B,C,H,W], channel stride 1)
        if t.dim() != 4 or (t.shape[1] > 1 and t.stride(1) != 1) or t.stride(3) % 8 or t.data_ptr() % 16:
            raise RuntimeError("bf16 activations must be NHWC views (channels innermost, 16-byte aligned slices)")
        return t
    if t.dtype != torch.float32:
        raise RuntimeError(f"fp32 (or bf16 NHWC) tensor expected, got {t.dtype}")
    return t


def is16(t: torch.Tensor) -> bool:
    return t.dtype == torch.bfloat16


def ld(t: torch.Tensor) -> int:
    """Pixel stride (elements) of a bf16 NHWC activation held as a logical [B,C,H,W] view."""
    return t.stride(3)


def _dense16(t: torch.Tensor) -> bool:
    """Pixels of all images form ONE run of rows with stride ld (kernels take P = B*H*W rows)."""
    B, _, H, W = t.shape
    l = t.stride(3)
    return (W == 1 or True) and (H == 1 or t.stride(2) == W * l) and (B == 1 or t.stride(0) == H * W * l)


class mixed_precision:
    """Context manager: run the TowerUNet body (encoder .. tower heads) in bf16 NHWC with fp32 accumulation,
    statistics and parameters -- the reference's precision="16-mixed" (model.py:168-186) on MI355X MFMA."""

    def __init__(self, enabled: bool = True):
        self.enabled = bool(enabled)

    def __enter__(self):
        self.prev = getattr(_state, "bf16", False)
        _state.bf16 = self.enabled
        return self

    def __exit__(self, *exc):
        _state.bf16 = self.prev
        return False


def bf16_enabled() -> bool:
    return bool(getattr(_state, "bf16", False))


def bstride(t: torch.Tensor) -> int:
    """Batch stride of an NCHW(-like) buffer whose inner dims are dense."""
    return t.stride(0) if t.shape[0] > 1 else int(t[0].numel())


def _dense_inner(t: torch.Tensor) -> bool:
    exp = 1
    for d in range(t.dim() - 1, 0, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            return False
        exp *= t.shape[d]
    return True


class Var:
    """A device buffer and (during training) its gradient buffer."""

    __slots__ = ("t", "grad", "req", "parent", "stats", "bnfin", "valid")

    def __init__(self, t: torch.Tensor, req: bool = False):
        self.t = t
        self.grad: T.Optional[torch.Tensor] = None
        self.req = req
        self.parent: T.Optional[T.Tuple["Var", int, int]] = None  # channel slice [c0, c1) of another Var
        self.stats: T.Optional[torch.Tensor] = None  # bf16 conv outputs: per-channel {sum, sumsq} from the epilogue
        self.bnfin = None  # (bn module, mean, rstd): BatchNorm statistics already finished by the producing conv launch
        self.valid: T.Optional[T.Tuple[int, int]] = None  # (H, W) of the image inside a larger stored grid (conv_transpose2d)

    @property
    def shape(self):
        return self.t.shape


class Tape:
    def __init__(self, enabled: bool):
        self.enabled = enabled
        self.nodes: T.List[T.Callable[[], None]] = []
        self.marks: T.Dict[int, int] = {}  # flat offset of a parameter -> index of the node that uses it

    def add(self, fn: T.Callable[[], None], params: T.Sequence[T.Optional[torch.Tensor]] = ()) -> None:
        if self.enabled:
            br = getattr(_state, "branch", None)
            if br is not None:  # recorded inside spawn(): its backward runs on the branch's stream
                fn = br.wrap(fn)
            self.nodes.append(fn)
            if params:
                base = current_store()._base
                idx = len(self.nodes) - 1
                for p in params:
                    if p is not None:  # a gradient is final once backward has passed the FIRST node that used it
                        k = (p.data_ptr() - base) // 4
                        self.marks[k] = min(self.marks.get(k, idx), idx)

    def backward(self) -> None:
        nodes, self.nodes = self.nodes, []
        hold = begin_branch_backward(self)
        store = getattr(_state, "store", None)
        try:
            with deferring_slice_sums(store) if store is not None else contextlib.nullcontext():
                while nodes:
                    fn = nodes.pop()
                    fn()
                    if hold is not None:
                        hold.append(fn)
        finally:
            release_branches()
        join_side_stream()


# ---------------------------------------------------------------------------
# weight gradients on a side stream
# ---------------------------------------------------------------------------
# In backward only dx feeds the next node; dW (and bias gradients) are needed by the optimizer alone. Their launches
# go to a second HIP stream that waits for the node's dy, so the launch-shaped layers of the coarse levels (25x25,
# 13x13: a fraction of the 256 CUs each) run beside the dx chain instead of in front of it. The main stream joins
# the side stream once, before the gradient exchange / optimizer. CN_OVERLAP_WGRAD=0 keeps everything on one stream.
_OVERLAP_WGRAD = os.environ.get("CN_OVERLAP_WGRAD", "1") == "1"
# diagnostic: weight gradients of more than this many multiply-accumulates stay on the compute stream (0 = no limit)
_SIDE_MAX_WORK = float(os.environ.get("CN_SIDE_MAX_WORK", "0"))
_side_streams: T.Dict[T.Any, T.Dict[str, T.Any]] = {}


def overlap_wgrad(enabled: bool) -> bool:
    """Switch the weight-gradient side stream on / off at run time (bench.py's isolated per-kernel pass); returns the
    previous setting. Pending side-stream work is joined first."""
    global _OVERLAP_WGRAD
    prev = _OVERLAP_WGRAD
    join_side_stream()
    _OVERLAP_WGRAD = bool(enabled)
    return prev


def _make_side_stream(dev) -> "torch.cuda.Stream":
    """The weight-gradient stream. Its kernels (one wave per SIMD, hundreds of microseconds each) must not take CUs
    from the data-gradient chain on the compute stream, which is the critical path: it is created with the LOWEST
    priority the device offers through the C ABI (torch.cuda.Stream can only ask for normal or higher) and wrapped as
    a torch ExternalStream for events / record_stream. CN_SIDE_STREAM = "low" (default) | "normal" |
    "mask:<n>" (hipExtStreamCreateWithCUMask over the first n CUs of every XCD-interleaved group)."""
    import ctypes

    mode = os.environ.get("CN_SIDE_STREAM", "low")
    if mode == "normal":
        return torch.cuda.Stream(device=dev)
    handle = ctypes.c_void_p()
    with torch.cuda.device(dev):
        if mode.startswith("mask:"):
            n = int(mode.split(":")[1])
            ncu = torch.cuda.get_device_properties(dev).multi_processor_count
            words = (ncu + 31) // 32
            mask = (ctypes.c_uint * words)()
            for i in range(min(n, ncu)):
                mask[i // 32] |= 1 << (i % 32)
            _lib.call("cn_stream_create", 0, mask, words, ctypes.byref(handle))
        else:
            rng = (ctypes.c_int * 2)()
            _lib.call("cn_stream_priority_range", rng)
            _lib.call("cn_stream_create", int(rng[0]), None, 0, ctypes.byref(handle))
    return torch.cuda.ExternalStream(handle.value, device=dev)


def _side_state(dev) -> T.Dict[str, T.Any]:
    st = _side_streams.get(dev)
    if st is None:
        st = {"stream": _make_side_stream(dev), "event": torch.cuda.Event(), "dirty": False}
        _side_streams[dev] = st
    return st


class side_stream:
    """Context manager: enqueue the enclosed launches on the side stream, ordered after everything enqueued so far on
    the current stream. ``tensors`` are the buffers those launches read (kept from being recycled by the caching
    allocator until the side stream has passed them)."""

    def __init__(self, *tensors: torch.Tensor, work: float = 0.0):
        self.tensors = tensors
        self.ctx = None
        self.work = work  # multiply-accumulates of the enclosed launches (0: unknown / small)

    def __enter__(self):
        if not _OVERLAP_WGRAD or (_SIDE_MAX_WORK > 0 and self.work > _SIDE_MAX_WORK):
            return self
        main = _main_stream()
        st = _side_state(main.device)
        cur = getattr(_state, "stream_obj_override", None) or main  # (inside spawn(): the branch's stream)
        if cur is main:
            _py_op(st["event"].record, main)
            _py_op(st["stream"].wait_event, st["event"])
        else:  # a fresh event: st["event"] may still be pending on the compute stream's behalf
            ev = torch.cuda.Event()
            _py_op(ev.record, cur)
            _py_op(st["stream"].wait_event, ev)
        st["dirty"] = True
        self.st = st
        self.prev = (getattr(_state, "stream_override", None), getattr(_state, "stream_obj_override", None))
        # launches go through the C ABI with an explicit stream handle: redirect _stream() instead of switching torch's
        # current stream (a torch.cuda.stream() context costs ~15 us of host time per layer). torch allocations made
        # inside the block still belong to the main stream: nothing inside may rely on a torch op (fills go through
        # cn_fill_f32 on the side stream).
        self.ctx = True
        _state.stream_override = st["stream"].cuda_stream
        _state.stream_obj_override = st["stream"]
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            _state.stream_override, _state.stream_obj_override = self.prev
            for t in self.tensors:
                if t is not None:
                    t.record_stream(self.st["stream"])
        return False


def join_side_stream() -> None:
    """Make the current stream wait for every weight-gradient launch issued so far."""
    if not torch.cuda.is_available():
        return
    main = _main_stream()
    st = _side_streams.get(main.device)
    if st is not None and st["dirty"]:
        _py_op(main.wait_stream, st["stream"])
        st["dirty"] = False


# ---------------------------------------------------------------------------
# small sub-graphs beside the big kernels: spawn() / join()
# ---------------------------------------------------------------------------
# final_c and final_b (TowerUNetFinal, nunet.py:197-202 of the reference) are chains of ~15 small launches forward and
# ~25 backward on 9- / 3- / 1-channel tensors. They depend only on x_tower_c / x_tower_b, which exist long before the
# forward reaches the heads, and in backward nothing needs their result until tower_b / tower_c run. spawn() puts such
# a chain on an auxiliary stream the moment its input exists, so that it executes BESIDE the MFMA-bound tower
# convolutions of the compute stream (the regime in which overlap pays on this chip: a bandwidth- / latency-bound kernel
# next to a matrix-pipe-bound one); its tape nodes run on the same stream in backward. CN_HEAD_STREAMS=0: off.
_HEAD_STREAMS = os.environ.get("CN_HEAD_STREAMS", "1") != "0"
_aux_streams: T.Dict[T.Any, T.List["torch.cuda.Stream"]] = {}


# Hardware queues of one process (round 6, tools/one_aux_try.sh + profiles/r06_hw_queues.txt). The HIP runtime keeps a
# pool of up to GPU_MAX_HW_QUEUES hardware queues PER STREAM PRIORITY, and this chip runs a process's queues concurrently
# only while there are at most SEVEN of them: with an eighth the queues are time-sliced and the whole step runs 1.5-1.9x
# slower (that is the "auxiliary streams + bucket collectives" slowdown of rounds 4-5: a torch.distributed process group
# brings six normal-priority streams with it -- 6 + the weight-gradient stream's lowest-priority queue + two
# highest-priority auxiliary queues = 9). The engine therefore owns exactly one queue per extra priority -- the
# weight-gradient stream (lowest) and ONE auxiliary stream for every spawn() (highest) -- and the process is expected to
# cap the normal pool at 5 (cultionet_amd.configure_runtime): 5 + 1 + 1 = 7 whatever the process group creates.
_HW_QUEUE_BUDGET = 7


def _hw_queue_cap() -> int:
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))  # (unset: the runtime's default of 4)
    except ValueError:
        return 4


def branch_streams_allowed() -> bool:
    """spawn() runs its sub-graph on the auxiliary stream unless that could push the process over the hardware-queue
    budget: with a torch.distributed process group alive (six normal-priority streams of its own) the auxiliary stream
    needs GPU_MAX_HW_QUEUES <= 5 (cap + weight-gradient queue + auxiliary queue <= 7). Checked LIVE at every spawn()
    (ADVICE r4), so it follows the group's lifetime and covers the drop-in mode under Lightning's own DDP;
    CN_KEEP_BRANCH_STREAMS=1 overrides it (to profile the interaction); ``engine.branch_streams(False)`` switches
    spawn() off for a scope. One-rank RCCL group, same box: bf16 2249 chips/s with the auxiliary stream at 5 queues
    against 2162 without it at 8 (single process 2279-2285); at 6 queues 1396."""
    if not _HEAD_STREAMS or getattr(_state, "no_branches", False):
        return False
    if _KEEP_BRANCHES:
        return True
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return _hw_queue_cap() + 2 <= _HW_QUEUE_BUDGET
    return True


_KEEP_BRANCHES = os.environ.get("CN_KEEP_BRANCH_STREAMS") == "1"


class branch_streams:
    """Context manager: allow (default) / forbid spawn() to use auxiliary streams on this thread for a scope."""

    def __init__(self, enabled: bool):
        self.enabled = bool(enabled)

    def __enter__(self):
        self.prev = getattr(_state, "no_branches", False)
        _state.no_branches = not self.enabled
        return self

    def __exit__(self, *exc):
        _state.no_branches = self.prev
        return False


def _aux_stream(dev, k: int = 0) -> "torch.cuda.Stream":
    """THE auxiliary stream of ``dev`` (``k`` names the sub-graph for the reader: every spawn() shares one stream since
    round 6 -- a second highest-priority stream is a second hardware queue, see _HW_QUEUE_BUDGET; same box, single
    process: bf16 2279 / fp32 387.5 chips/s with one stream against 2285 / 385.8 with two)."""
    if _OVERLAP_WGRAD:
        _side_state(dev)  # HIP deals streams to hardware queues in creation order: the weight-gradient stream first
    lst = _aux_streams.setdefault(dev, [])
    if not lst:
        # HIGHEST priority, through the C ABI like the weight-gradient stream: a priority is a property of the hardware
        # queue, so this stream can never be dealt the queue of the compute stream, of the (lowest-priority)
        # weight-gradient stream or of RCCL's streams -- with torch.cuda.Stream() (normal priority) a live process group
        # put an auxiliary stream on a queue shared with the compute stream: 2115 -> 1255 chips/s (bf16), 377 -> 241 (fp32).
        import ctypes

        handle = ctypes.c_void_p()
        rng = (ctypes.c_int * 2)()
        with torch.cuda.device(dev):
            _lib.call("cn_stream_priority_range", rng)
            _lib.call("cn_stream_create", int(rng[1]), None, 0, ctypes.byref(handle))
        lst.append(torch.cuda.ExternalStream(handle.value, device=dev))
    return lst[0]


# ---- the engine's allocator -----------------------------------------------------------------------------------------
# Every per-step buffer of the engine (activations, gradients, statistics rows, masks, scratch) comes from _alloc().
# torch's caching allocator only knows the compute stream, while launches carry explicit stream handles: whoever needs
# buffers to outlive a host-side free -- the branches of spawn() until they are joined, a launch plan while it is being
# recorded -- registers a list as an allocation SINK of the calling thread (``holding_allocations``), and _alloc() files
# every tensor it hands out there. Nothing in ``torch``'s namespace is touched (rounds 3-5 wrapped
# torch.empty / zeros / full process-wide instead).
def _alloc(shape, dtype: torch.dtype, device) -> torch.Tensor:
    t = torch.empty(shape, dtype=dtype, device=device)
    sinks = getattr(_state, "alloc_sinks", None)
    if sinks:
        for sink in sinks:
            sink.append(t)
    return t


def _alloc_like(t: torch.Tensor) -> torch.Tensor:
    return _alloc(tuple(t.shape), t.dtype, t.device)


def alloc(shape, dtype: torch.dtype, device) -> torch.Tensor:
    """Public form of the engine allocator for the host mirror (unet_parts / lightning): a buffer whose lifetime follows
    the calling thread's open branches / plan recording."""
    return _alloc(shape, dtype, device)


class holding_allocations:
    """Context manager: every buffer _alloc() hands out on this thread is also appended to ``sink`` (kept alive by it)."""

    def __init__(self, sink: T.List[T.Any]):
        self.sink = sink

    def __enter__(self):
        sinks = getattr(_state, "alloc_sinks", None)
        if sinks is None:
            sinks = _state.alloc_sinks = []
        sinks.append(self.sink)
        return self.sink

    def __exit__(self, *exc):
        sinks = _state.alloc_sinks
        for i in range(len(sinks) - 1, -1, -1):
            if sinks[i] is self.sink:
                del sinks[i]
                break
        return False


class _KeepAlive:
    """Frees are deferred while any branch is open. torch's caching allocator only knows the compute stream (launches
    go through the C ABI with explicit stream handles): a block freed on the host while an auxiliary stream's kernel
    still uses it could be handed to a compute-stream allocation at once. Reference-counted per thread: buffers the
    engine allocates (``keep`` is an allocation sink of _alloc while depth > 0), finished backward closures and incoming
    gradients are held until the thread's last open branch has been joined."""

    def __init__(self):
        self.depth = 0
        self.keep: T.List[T.Any] = []
        self._hold: T.Optional[holding_allocations] = None

    def acquire(self) -> None:
        self.depth += 1
        if self.depth == 1:
            self._hold = holding_allocations(self.keep)
            self._hold.__enter__()

    def release(self, force: bool = False) -> None:
        if self.depth == 0:
            return
        self.depth = 0 if force else self.depth - 1
        if self.depth == 0:
            if self._hold is not None:
                self._hold.__exit__(None, None, None)
                self._hold = None
            self.keep = []


def _keepalive() -> _KeepAlive:
    ka = getattr(_state, "keepalive", None)
    if ka is None:
        ka = _state.keepalive = _KeepAlive()
    return ka


def begin_branch_backward(tape) -> T.Optional[T.List[T.Any]]:
    """A tape with spawned branches: NOTHING is freed during its backward (returns the list that holds the finished
    nodes' closures; the engine's allocations are filed there by _alloc). Between a branch's start and its fork the compute stream
    runs arbitrary nodes of its own; a block one of them frees on the host while its kernel is still queued must not be
    handed to an allocation whose kernel runs on an auxiliary stream."""
    if not getattr(tape, "has_branches", False):
        return None
    ka = _keepalive()
    ka.acquire()
    return ka.keep


def release_branches() -> None:
    """After a backward pass (also one that raised): nothing may be left holding tensors or torch wrappers."""
    _keepalive().release(force=True)
    _state.open_branches = []


def aux_stream_events() -> T.List["torch.cuda.Event"]:
    """Events recorded now on the auxiliary streams with backward work of open branches (for the RCCL bucket stream,
    like side_stream_event())."""
    out = []
    for br in getattr(_state, "open_branches", ()):
        ev = torch.cuda.Event()
        ev.record(br.stream)
        out.append(ev)
    return out


class _Branch:
    def __init__(self, stream: "torch.cuda.Stream", main: "torch.cuda.Stream", inputs, aliases):
        self.stream, self.main = stream, main
        self.inputs, self.aliases = inputs, aliases
        self.fork_ev, self.done_ev = torch.cuda.Event(), torch.cuda.Event()
        self.left = 0
        self.result = None

    def enter(self):
        prev = (getattr(_state, "stream_override", None), getattr(_state, "stream_obj_override", None))
        _state.stream_override, _state.stream_obj_override = self.stream.cuda_stream, self.stream
        _bind_conv_workspace(self.main.device)
        return prev

    @staticmethod
    def leave(prev) -> None:
        _state.stream_override, _state.stream_obj_override = prev

    def wrap(self, fn: T.Callable[[], None]) -> T.Callable[[], None]:
        self.left += 1

        def node():
            prev = self.enter()
            try:
                fn()
            finally:
                self.leave(prev)
            _keepalive().keep.append(fn)  # its closure (saved activations) lives until the branches are joined
            self.left -= 1
            if self.left == 0:
                _py_op(self.done_ev.record, self.stream)

        return node

    def bwd_fork(self) -> None:
        """In backward this runs AFTER the branch's nodes (it was recorded before them): the compute stream waits for
        the branch, takes over the gradients of its inputs, and the deferred frees of this branch are released."""
        _py_op(self.main.wait_event, self.done_ev)
        for x, xa in zip(self.inputs, self.aliases):
            if xa.grad is not None:
                give_grad(x, xa.grad)
                xa.grad = None
        ob = getattr(_state, "open_branches", None)
        if ob and self in ob:
            ob.remove(self)
        _keepalive().release()


def spawn(fn: T.Callable[..., T.Any], inputs: T.Sequence[Var], k: int) -> T.Tuple[T.Optional[_Branch], T.Any]:
    """Run ``fn(*aliases of inputs)`` -- engine ops that read nothing but ``inputs`` and the parameters -- on auxiliary
    stream k, concurrently with whatever the compute stream does next. Returns (branch, result); join([branch, ...])
    must be called before the result is used. The branch sees ALIASES of its inputs (same tensors, own gradient slots):
    two streams never accumulate into one gradient buffer; the compute stream adds the alias gradients in bwd_fork."""
    tape = current_tape()
    if (not _OVERLAP_WGRAD or not torch.cuda.is_available() or getattr(_state, "stream_override", None) is not None
            or getattr(_state, "branch", None) is not None or not branch_streams_allowed()):
        return None, fn(*inputs)
    main = _main_stream()
    aliases = [Var(x.t, x.req) for x in inputs]
    br = _Branch(_aux_stream(main.device, k), main, list(inputs), aliases)
    if tape.enabled:
        tape.has_branches = True
        tape.add(br.bwd_fork)
    _keepalive().acquire()
    _py_op(br.fork_ev.record, main)
    _py_op(br.stream.wait_event, br.fork_ev)
    prev = br.enter()
    _state.branch = br
    try:
        br.result = fn(*aliases)
    finally:
        _state.branch = None
        br.leave(prev)
    _py_op(br.done_ev.record, br.stream)
    return br, br.result


def join(branches: T.Sequence[T.Optional[_Branch]]) -> None:
    """The compute stream waits for the spawned branches (forward); in backward this is where they may start."""
    brs = [b for b in branches if b is not None]
    if not brs:
        return
    tape = current_tape()
    for br in brs:
        _py_op(br.main.wait_event, br.done_ev)
        _keepalive().release()
    if not tape.enabled:
        return

    def bwd_join():
        ka = _keepalive()
        ob = getattr(_state, "open_branches", None)
        if ob is None:
            ob = _state.open_branches = []
        ev = torch.cuda.Event()
        for br in brs:
            ka.acquire()
            ob.append(br)
            r = br.result
            for v in (r if isinstance(r, (tuple, list)) else (r,)):
                if isinstance(v, Var) and v.grad is not None:
                    ka.keep.append(v.grad)  # allocated before the deferral began, freed inside the branch
        _py_op(ev.record, brs[0].main)
        for br in brs:
            _py_op(br.stream.wait_event, ev)
            if br.left == 0:  # no backward nodes: nothing to wait for at the fork
                _py_op(br.done_ev.record, br.stream)

    tape.add(bwd_join)


def side_stream_event() -> T.Optional["torch.cuda.Event"]:
    """An event recorded on the weight-gradient side stream now (None when nothing is pending there): lets another
    stream -- the RCCL bucket stream -- wait for the gradients issued so far without stalling the compute stream."""
    if not torch.cuda.is_available():
        return None
    st = _side_streams.get(_main_stream().device)
    if st is None or not st["dirty"]:
        return None
    ev = torch.cuda.Event()
    ev.record(st["stream"])
    return ev


def current_tape() -> Tape:
    tp = getattr(_state, "tape", None)
    if tp is None:
        tp = Tape(False)
        _state.tape = tp
    return tp


class recording:
    """Context manager: run forward ops under a fresh tape (enabled or not)."""

    def __init__(self, enabled: bool = True):
        self.tape = Tape(enabled)

    def __enter__(self) -> Tape:
        self.prev = getattr(_state, "tape", None)
        _state.tape = self.tape
        return self.tape

    def __exit__(self, *exc):
        _state.tape = self.prev
        return False


# ---------------------------------------------------------------------------
# gradient plumbing
# ---------------------------------------------------------------------------

def grad_buffer(v: Var) -> T.Tuple[torch.Tensor, int]:
    """(buffer, accumulate flag) to write/accumulate v's gradient into."""
    if v.grad is None:
        if v.parent is not None:
            # a channel slice writes straight into its parent's gradient (zero-filled once, then accumulated)
            pv, c0, c1 = v.parent
            if pv.grad is None:
                pv.grad = _new(tuple(pv.t.shape), pv.t)
                if is16(pv.t):
                    _lib.call("cn_zero_bf16", pv.grad.data_ptr(), ld(pv.grad), _rows(pv.grad), pv.grad.shape[1], _stream())
                else:
                    _lib.call("cn_fill_f32", pv.grad.data_ptr(), pv.grad.numel(), 0.0, _stream())
            v.grad = pv.grad[:, c0:c1]
            return v.grad, 1
        v.grad = _new(tuple(v.t.shape), v.t)
        return v.grad, 0
    return v.grad, 1


def give_grad(v: Var, g: torch.Tensor) -> None:
    """v.grad (+)= g where the op is the identity on the gradient: alias when first, else add in place."""
    if not v.req:
        return
    if v.grad is None and v.parent is None:
        v.grad = g
        return
    if v.grad is None:
        grad_buffer(v)
    if is16(g):
        _lib.call("cn_copy_bf16", g.data_ptr(), ld(g), v.grad.data_ptr(), ld(v.grad), _rows(g), g.shape[1], 1, _stream())
        return
    B = g.shape[0]
    n = g[0].numel()
    _lib.call("cn_copy_f32", g.data_ptr(), bstride(g), v.grad.data_ptr(), bstride(v.grad), B, n, 1, _stream())


# ---------------------------------------------------------------------------
# flat parameter store
# ---------------------------------------------------------------------------

class ParamStore:
    """All parameters of a module in one flat fp32 buffer; gradients likewise.

    ``param.data`` becomes a view of ``flat`` (checkpoints / state_dict keep working); gradients are
    accumulated by the kernels into ``flat_grad``, on which the fused AdamW step and the RCCL
    all-reduce operate directly (``attach_grads`` exposes them as ``param.grad`` views).
    """

    _serial = __import__("itertools").count(1)

    def __init__(self, module: torch.nn.Module):
        params = [p for p in module.parameters()]
        if not params:
            raise ValueError("module has no parameters")
        params = self._with_contiguous_groups(module, params)
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("ParamStore needs the module on the GPU (call .to('cuda') first)")
        self.params = params
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4  # 16-byte aligned slices
        self.numel = n
        self.offsets = offs
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self.version = 0  # bumped whenever parameter values change (invalidates packed weights)
        # process-unique serial: caches key their validity on it (CPython may hand a rebuilt store the id() of a dead one,
        # and a cached VIEW of the old flat buffer would then be used silently -- ADVICE r4)
        self.uid = next(ParamStore._serial)
        self._sig = self._signature()
        self._base = self.flat.data_ptr()
        self._packs: T.List[T.Tuple] = []      # (PackedWeight, attr, dst tensor, w ptr, T, K, N, sk, sn, st)
        self._pack_table: T.Optional[torch.Tensor] = None
        self._packs16: T.List[T.Tuple] = []    # the same for the bf16 MFMA-fragment copies
        self._pack_table16: T.Optional[torch.Tensor] = None

    @staticmethod
    def _with_contiguous_groups(module: torch.nn.Module, params: T.List[torch.nn.Parameter]) -> T.List[torch.nn.Parameter]:
        """Registration order with the members of every declared group made adjacent (at the place of the group's first
        member). A submodule declares groups through ``cn_contiguous_params()`` -> lists of parameters that one kernel
        reads / writes as ONE tensor (the three 128 -> 3 head convolutions of a tower: one 128 -> 9 convolution on the
        views, no concatenated copies of weights or gradients). Members must be multiples of 4 elements (the store pads
        every slice to 16 bytes: a padded member would break adjacency)."""
        first_of, member = {}, set()
        for m in module.modules():
            fn = getattr(m, "cn_contiguous_params", None)
            if fn is None:
                continue
            for grp in fn():
                grp = list(grp)
                if len(grp) < 2 or any(q.numel() % 4 for q in grp) or any(id(q) in member for q in grp):
                    continue
                first_of[id(grp[0])] = grp
                member.update(id(q) for q in grp)
        if not first_of:
            return params
        out = []
        for q in params:
            if id(q) in first_of:
                out.extend(first_of[id(q)])
            elif id(q) not in member:
                out.append(q)
        assert len(out) == len(params)
        return out

    def owns(self, module: torch.nn.Module) -> bool:
        lo, hi = self._base, self._base + self.numel * 4
        first, last = self.params[0], self.params[-1]
        return lo <= first.data_ptr() < hi and lo <= last.data_ptr() < hi

    def grad_of(self, p: torch.Tensor) -> torch.Tensor:
        """Gradient slice matching a parameter (or any dense view of one) inside the flat buffer."""
        o = (p.data_ptr() - self._base) // 4
        if o < 0 or o + p.numel() > self.numel:
            raise RuntimeError("tensor is not part of this ParamStore (was the module moved or re-created?)")
        return self.flat_grad[o:o + p.numel()].view(p.shape)

    def attach_grads(self) -> None:
        """Point every ``param.grad`` at its slice of the flat gradient (for torch optimizers driving the NATIVE step).
        Not for drop-in mode: autograd_bridge hands the flat buffer itself to autograd at every backward (p.grad then
        IS a view of that step's buffer) and gives the store a fresh one, so views attached here would go stale."""
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)

    def zero_grad(self) -> None:
        _lib.call("cn_fill_f32", self.flat_grad.data_ptr(), self.numel, 0.0, _stream())

    def bump(self) -> None:
        self.version += 1

    def _signature(self) -> int:
        """Sum of the parameters' torch version counters: moves whenever anything outside the engine writes a
        parameter in place (torch optimizers in drop-in mode, ``load_state_dict`` / ``load_from_checkpoint`` through
        nn.Module's recursion, ``param.data.copy_``). The engine's own kernels (fused AdamW) do not touch it and call
        ``bump()`` themselves."""
        return sum(p._version for p in self.params)

    def refresh(self) -> None:
        """Invalidate the packed weight copies if the parameters were modified behind the engine's back."""
        sig = self._signature()
        if sig != self._sig:
            self._sig = sig
            self.bump()

    def register_pack(self, pw, attr: str, dst: torch.Tensor, w: torch.Tensor, T_: int, K: int, N: int, sk: int,
                      sn: int, st: int) -> None:
        self._packs.append((pw, attr, dst, w.data_ptr(), T_, K, N, sk, sn, st))
        self._pack_table = None

    def register_pack16(self, pw, attr: str, dst: torch.Tensor, w: torch.Tensor, T_: int, K: int, N: int, sk: int,
                        sn: int, st: int) -> None:
        self._packs16.append((pw, attr, dst, w.data_ptr(), T_, K, N, sk, sn, st))
        self._pack_table16 = None

    def repack_all(self) -> None:
        """Refresh every registered packed weight copy with ONE launch per precision (after the parameters changed)."""
        if self._packs16:
            if self._pack_table16 is None:
                import struct

                buf = bytearray()
                for (_pw, _attr, dst, wptr, T_, K, N, sk, sn, st) in self._packs16:
                    buf += struct.pack("<QQiiiiiiqqqQ", wptr, dst.data_ptr(), T_, K, N, (K + 15) // 16, (N + 31) // 32, 0,
                                       sk, sn, st, 0)  # 72-byte CnBPackDesc records (nscale = NULL)
                self._pack_table16 = torch.frombuffer(buf, dtype=torch.uint8).clone().to(self.flat.device)
            _lib.call("cn_pack_weights_batched_bf16", self._pack_table16.data_ptr(), len(self._packs16), _stream())
            for (pw, _attr, _dst, *_rest) in self._packs16:
                pw.version = self.version
        if not self._packs:
            return
        if self._pack_table is None:
            import struct

            buf = bytearray()
            for (_pw, _attr, dst, wptr, T_, K, N, sk, sn, st) in self._packs:
                kp = _lib.query("cn_conv_kpad", K)
                np_ = _lib.query("cn_conv_npad", N)
                buf += struct.pack("<QQiiiiiiqqq", wptr, dst.data_ptr(), T_, K, N, kp, np_, 0, sk, sn, st)
            host = torch.frombuffer(buf, dtype=torch.uint8).clone()
            self._pack_table = host.to(self.flat.device)
        _lib.call("cn_pack_weights_batched_f32", self._pack_table.data_ptr(), len(self._packs), _stream())
        for (pw, _attr, _dst, *_rest) in self._packs:
            pw.version = self.version


# Launch plans (cultionet_amd/replay.py) bake the raw pointers of the process-wide scratch buffers below into their
# recorded arguments. Every (re)allocation of one of them bumps this epoch; the plans' keys hold it, so a plan recorded
# against an older buffer is never replayed (it is re-recorded against the current one).
_ws_epoch = 0


def workspace_epoch() -> int:
    return _ws_epoch


def _bump_ws_epoch() -> None:
    global _ws_epoch
    _ws_epoch += 1


_CONV_WS_FLOATS = 16 << 20  # split-K partial slices of the implicit-GEMM launches (64 MB, one per process)
_conv_ws: T.Dict[T.Tuple[str, int], torch.Tensor] = {}


def _bind_conv_workspace(dev: torch.device) -> None:
    """Register a split-K scratch buffer for the CURRENT stream (the library keys its scratch by stream, so model
    instances running on different streams / autograd worker threads never share one)."""
    s = _stream()
    key = (str(dev), s)
    if key in _conv_ws:
        return
    ws = torch.empty(_CONV_WS_FLOATS, dtype=torch.float32, device=dev)
    _conv_ws[key] = ws
    _bump_ws_epoch()
    _lib.call("cn_conv_set_workspace", s, ws.data_ptr(), ws.numel())
    # CN_AUTOTUNE=1: measure the (tile, K split) candidates per conv shape during the first steps instead of
    # trusting the launch-cost model (+0.5 % at batch 8; off by default so that runs are reproducible)
    _lib.call("cn_conv_set_autotune", 1 if os.environ.get("CN_AUTOTUNE", "0") == "1" else 0)


def current_store() -> ParamStore:
    st = getattr(_state, "store", None)
    if st is None:
        raise RuntimeError("no active ParamStore (ops on parameters must run inside a model forward)")
    return st


class using_store:
    def __init__(self, store: ParamStore):
        self.store = store

    def __enter__(self):
        self.prev = getattr(_state, "store", None)
        _state.store = self.store
        self.prev_stream = (getattr(_state, "main_stream", None), getattr(_state, "main_handle", None))
        if self.store.flat.is_cuda:
            ms = torch.cuda.current_stream()
            _state.main_stream, _state.main_handle = ms, ms.cuda_stream
        _bind_conv_workspace(self.store.flat.device)
        return self.store

    def __exit__(self, *exc):
        _state.store = self.prev
        _state.main_stream, _state.main_handle = self.prev_stream
        return False


def pgrad(p: torch.nn.Parameter) -> torch.Tensor:
    return current_store().grad_of(p)


# ---------------------------------------------------------------------------
# packed weights cache (re-packed when the store version changes)
# ---------------------------------------------------------------------------

class PackedWeight:
    """Packed copies of one weight tensor for the implicit-GEMM kernels (forward / bwd-data)."""

    __slots__ = ("fwd", "bwd", "fwd16", "bwd16", "version", "store_id")

    def __init__(self):
        self.fwd = None
        self.bwd = None
        self.fwd16 = None  # bf16 MFMA-fragment copies (mixed-precision path)
        self.bwd16 = None
        self.version = -1
        self.store_id = 0


def _pack(pw: "PackedWeight", attr: str, w: torch.Tensor, T_: int, K: int, N: int, sk: int, sn: int,
          st: int) -> torch.Tensor:
    """Pack now (first use) into a persistent buffer and register it for the batched per-step repack."""
    kp = _lib.query("cn_conv_kpad", K)
    np_ = _lib.query("cn_conv_npad", N)
    out = _alloc(T_ * kp * np_, torch.float32, w.device)
    _lib.call("cn_pack_weights_f32", w.data_ptr(), out.data_ptr(), T_, K, N, sk, sn, st, _stream())
    current_store().register_pack(pw, attr, out, w, T_, K, N, sk, sn, st)
    return out


def _sync_packs(pw: "PackedWeight") -> None:
    """Bring all registered packed weights up to date if the parameters changed since the last pack."""
    st = current_store()
    if pw.store_id != st.uid:  # parameters were re-flattened into a new store: drop copies of the old one
        pw.fwd = pw.bwd = pw.fwd16 = pw.bwd16 = None
        pw.store_id = st.uid
    if pw.version != st.version:
        if pw.fwd is None and pw.bwd is None and pw.fwd16 is None and pw.bwd16 is None:
            pw.version = st.version
        else:
            st.repack_all()


def _pack16(pw: "PackedWeight", attr: str, w: torch.Tensor, T_: int, K: int, N: int, sk: int, sn: int,
            st: int) -> torch.Tensor:
    out = _alloc(_lib.query("cn_bconv_packed_elems", T_, K, N), torch.bfloat16, w.device)
    _lib.call("cn_pack_weights_bf16", w.data_ptr(), out.data_ptr(), T_, K, N, sk, sn, st, _stream())
    current_store().register_pack16(pw, attr, out, w, T_, K, N, sk, sn, st)
    return out


def packed_conv(mod, need_bwd: bool, bf16: bool = False) -> PackedWeight:
    """Packed weights of an nn.Conv2d / nn.Linear-like module (weight [Cout][Cin][KH][KW])."""
    pw = mod.__dict__.get("_cn_packed")
    if pw is None:
        pw = PackedWeight()
        mod.__dict__["_cn_packed"] = pw
    _sync_packs(pw)
    w = mod.weight
    cout, cin = w.shape[0], w.shape[1]
    taps = int(w[0, 0].numel()) if w.dim() > 2 else 1
    if bf16:
        if pw.fwd16 is None:
            pw.fwd16 = _pack16(pw, "fwd16", w, taps, cin, cout, taps, cin * taps, 1)
        if need_bwd and pw.bwd16 is None:
            pw.bwd16 = _pack16(pw, "bwd16", w, taps, cout, cin, cin * taps, taps, 1)
        return pw
    if pw.fwd is None:
        pw.fwd = _pack(pw, "fwd", w, taps, cin, cout, taps, cin * taps, 1)
    if need_bwd and pw.bwd is None:
        pw.bwd = _pack(pw, "bwd", w, taps, cout, cin, cin * taps, taps, 1)
    return pw


def packed_convT(mod, need_bwd: bool, bf16: bool = False, taps_as_channels: bool = False) -> PackedWeight:
    """Packed weights of an nn.ConvTranspose2d (weight [Cin][Cout][KH][KW]). ``taps_as_channels`` (fp32, stride >=
    kernel size, _conv_transpose2d_taps): the tensor as the [Cin][Cout*KH*KW] matrix of a 1x1 transposed convolution."""
    pw = mod.__dict__.get("_cn_packed")
    if pw is None:
        pw = PackedWeight()
        mod.__dict__["_cn_packed"] = pw
    _sync_packs(pw)
    w = mod.weight
    cin, cout = w.shape[0], w.shape[1]
    taps = int(w[0, 0].numel())
    if taps_as_channels and not bf16:
        n = cout * taps
        if pw.fwd is None:
            pw.fwd = _pack(pw, "fwd", w, 1, cin, n, n, 1, 0)
        if need_bwd and pw.bwd is None:
            pw.bwd = _pack(pw, "bwd", w, 1, n, cin, 1, n, 0)
        return pw
    if bf16:
        if pw.fwd16 is None:
            pw.fwd16 = _pack16(pw, "fwd16", w, taps, cin, cout, cout * taps, taps, 1)
        if need_bwd and pw.bwd16 is None:
            pw.bwd16 = _pack16(pw, "bwd16", w, taps, cout, cin, taps, cout * taps, 1)
        return pw
    if pw.fwd is None:
        pw.fwd = _pack(pw, "fwd", w, taps, cin, cout, cout * taps, taps, 1)
    if need_bwd and pw.bwd is None:
        pw.bwd = _pack(pw, "bwd", w, taps, cout, cin, taps, cout * taps, 1)
    return pw


# ---------------------------------------------------------------------------
# ops
# ---------------------------------------------------------------------------

def _new(shape, like: torch.Tensor) -> torch.Tensor:
    if like.dtype == torch.bfloat16 and len(shape) == 4:
        B, C, H, W = shape
        return _alloc((B, H, W, C), torch.bfloat16, like.device).permute(0, 3, 1, 2)
    return _alloc(shape, torch.float32, like.device)


def new_buffer(shape, like: torch.Tensor) -> torch.Tensor:
    """An activation buffer of logical shape [B,C,H,W] in ``like``'s precision / layout (fp32 NCHW or bf16 NHWC)."""
    return _new(tuple(shape), like)


def _rows(t: torch.Tensor) -> int:
    return t.shape[0] * t.shape[2] * t.shape[3]


_WS_FLOATS = 16 << 20  # persistent weight-gradient scratch (64 MB): aligned operand copies + partial dW slices


# ---------------------------------------------------------------------------
# deferred weight-gradient slice sums (csrc/cn_slicesum.h)
# ---------------------------------------------------------------------------
# A many-split weight gradient leaves partial dW slices in scratch; `dW += sum(slices)` is needed by the optimizer (or by
# the gradient bucket's all-reduce), not by the next backward node. During a backward pass the library appends those
# sums to a table instead of launching them (34 fp32 / 68 bf16 dependent launches per step), and ONE batched launch per
# flush -- per ready bucket under data parallelism, else one per step -- adds them all. Since the slices must survive
# until then, every weight-gradient call of the pass takes its scratch from an arena that only moves forward.
# CN_DEFER_SUMS=0: the sums run right behind their contraction as in rounds 1-4.
_DEFER_SUMS = os.environ.get("CN_DEFER_SUMS", "1") != "0"
_SS_CAP = 512            # records per backward pass (64 bytes each)
_SS_BLOCK = 64 << 20     # floats per arena block (256 MB)
# Pending slices are summed as soon as this many floats are waiting (CN_SUM_FLUSH_MB, default 128 MB): they are then
# still in the 256 MB Infinity Cache, and the pass does not end with ONE 0.6 GB reduction behind the last weight gradient
# on the stream that finishes last (fp32, same box: all-at-the-end 386.4 chips/s against 387.5 with the 34 immediate
# sums). 0 = only the final flush.
_SS_FLUSH_FLOATS = int(float(os.environ.get("CN_SUM_FLUSH_MB", "128")) * (1 << 20) / 4)


class _SliceSums:
    """Per-device state of the deferred sums: pinned host table the library writes, its device copy, the arena."""

    def __init__(self, dev: torch.device):
        self.dev = dev
        self.host = torch.zeros(_SS_CAP * 64, dtype=torch.uint8).pin_memory()
        self.table = torch.zeros(_SS_CAP * 64, dtype=torch.uint8, device=dev)
        self.uploaded: T.Dict[T.Tuple[int, int], bytes] = {}  # record range -> bytes the device copy holds for it
        self.upload_ev = torch.cuda.Event()                    # (re-)recorded behind every table upload
        self.writer: T.Any = None    # the launch plan that rewrote the host table last (None: an eager pass)
        self.blocks: T.List[torch.Tensor] = []
        self.active = False
        self.reset()

    def reset(self) -> None:
        self.cur, self.off, self.seen, self.flushed, self.pending = 0, 0, 0, 0, 0

    def _commit(self) -> None:
        """Move the arena past the slices of the records appended since the last call (a call whose sum was not
        deferred leaves nothing behind: its scratch is reused by the next call, as stream order allows)."""
        n = _lib.query("cn_slice_sums_count")
        if n <= self.seen or not self.blocks:
            return
        import struct

        raw = bytes(self.host[self.seen * 64:n * 64].numpy())
        base = self.blocks[self.cur].data_ptr()
        for i in range(n - self.seen):
            part, _dw, stride, _n, nslices = struct.unpack_from("<QQqqi", raw, i * 64)
            end = (part - base) // 4 + stride * nslices
            if 0 <= end <= self.blocks[self.cur].numel():
                self.off = max(self.off, (end + 63) // 64 * 64)
            self.pending += stride * nslices
        self.seen = n

    def take(self, need: int) -> T.Tuple[int, int]:
        self._commit()
        if 0 < _SS_FLUSH_FLOATS <= self.pending:
            flush_slice_sums()
        need = int(need)
        while True:
            if self.cur >= len(self.blocks):
                # (no workspace-epoch bump: a block that is ADDED moves nothing a recorded plan points at)
                self.blocks.append(torch.empty(max(_SS_BLOCK, need), dtype=torch.float32, device=self.dev))
            blk = self.blocks[self.cur]
            if blk.numel() - self.off >= need:
                return blk.data_ptr() + 4 * self.off, need
            if self.off == 0:  # an oversized request in front of a standard block
                torch.cuda.synchronize(self.dev)
                self.blocks[self.cur] = torch.empty(need, dtype=torch.float32, device=self.dev)
                _bump_ws_epoch()
                continue
            self.cur, self.off = self.cur + 1, 0


def _ss_state() -> T.Optional[_SliceSums]:
    """The active deferral state of this thread (None: sums run immediately)."""
    return getattr(_state, "slice_sums", None)


def _sum_stream() -> int:
    """Handle of the stream every weight gradient of the pass runs on (the side stream, or the compute stream when the
    overlap is off): the batched sums are stream-ordered behind the contractions that wrote their slices."""
    if _OVERLAP_WGRAD:
        return _side_state(_main_stream().device)["stream"].cuda_stream
    return _stream()


class deferring_slice_sums:
    """Context manager around one backward pass: weight-gradient slice sums into ``store.flat_grad`` are collected and
    run by flush_slice_sums() (called on exit, and by the data-parallel bucket logic before a bucket is reduced)."""

    def __init__(self, store: "ParamStore"):
        self.store = store
        self.on = False

    def __enter__(self):
        if (not _DEFER_SUMS or _SIDE_MAX_WORK > 0 or _ss_state() is not None or not self.store.flat_grad.is_cuda):
            return self
        # One state PER STORE (tables, arena): a launch plan recorded for one trainer bakes these addresses in, and a
        # second trainer of the process (tests run an eager and a replayed one side by side) must not rewrite them.
        st = getattr(self.store, "_slice_sums", None)
        if st is None:
            st = self.store._slice_sums = _SliceSums(self.store.flat_grad.device)
        if st.active:  # a pass over this store is already collecting (another thread): that one keeps the arena
            return self
        st.reset()
        st.active = True
        # the library is about to rewrite the host table: an upload still in flight (enqueued by the previous eager pass,
        # or by a replayed plan, which re-records the event) must have read it first
        if not st.upload_ev.query():
            st.upload_ev.synchronize()
        if _recorder is None:
            st.writer = None
        _state.slice_sums = st
        _lib.call("cn_slice_sums_begin", st.host.data_ptr(), _SS_CAP, self.store.flat_grad.data_ptr(),
                  self.store.numel)
        self.on = True
        return self

    def __exit__(self, *exc):
        if self.on:
            try:
                if exc[0] is None:
                    flush_slice_sums()
            finally:
                _lib.call("cn_slice_sums_end")
                _state.slice_sums.active = False
                _state.slice_sums = None
        return False


def flush_slice_sums() -> None:
    """ONE launch adding every pending slice sum into the flat gradient, on the weight gradients' stream."""
    st = _ss_state()
    if st is None:
        return
    st._commit()
    n = _lib.query("cn_slice_sums_count")
    first = st.flushed
    if n <= first:
        return
    now = st.host.numpy().reshape(-1, 64)[first:n, :60].tobytes()  # (without the block dealing the run call writes)
    # a recorded plan always refreshes the device copy: eager steps in between (other shapes) may have changed it
    upload = 0 if (st.uploaded.get((first, n)) == now and _recorder is None) else 1
    sobj = _side_state(st.dev)["stream"] if _OVERLAP_WGRAD else _main_stream()
    stream = _sum_stream()
    _lib.call("cn_slice_sums_run", st.host.data_ptr(), st.table.data_ptr(), first, n - first, upload, stream)
    if upload:  # the device copy of every overlapping range is no longer what `uploaded` says
        for k in [k for k in st.uploaded if k[0] < n and first < k[1]]:
            del st.uploaded[k]
        st.uploaded[(first, n)] = now
        _py_op(st.upload_ev.record, sobj)  # (a plan entry too: the next eager pass waits for a replay's upload)
    st.pending = 0
    st.flushed = n
    # every slice handed out so far has been consumed by a launch that is now enqueued on the weight gradients' stream,
    # and whatever takes scratch next runs behind it on the same stream: the arena starts over (it used to
    # grow to the whole pass's slices, ~0.6 GB per fp32 step, and never shrank)
    st.cur, st.off = 0, 0


def _pad_ws(*tensors: torch.Tensor) -> T.Tuple[T.Optional[int], int]:
    """(pointer, floats) of the weight-gradient scratch: room for aligned copies of odd-sized operands plus the
    partial-dW slices of many-split launches. One persistent buffer per device (launches are stream-ordered)."""
    need = _WS_FLOATS
    for t in tensors:
        H, W = t.shape[-2], t.shape[-1]
        if (H * W) % 4 or (W & 1):
            need += t.shape[0] * t.shape[1] * (H * (W + 1) + 3)
    dev = tensors[0].device
    ss = _ss_state()
    if ss is not None:  # deferred slice sums: this call's slices must outlive the next call
        return ss.take(need)
    pool = getattr(_state, "ws_pool", None)
    if pool is None:
        pool = _state.ws_pool = {}
    ws = pool.get(dev)
    if ws is None or ws.numel() < need:
        if ws is not None:  # growing: the old buffer may still be read by launches in flight on the side stream
            torch.cuda.synchronize(dev)
        ws = pool[dev] = torch.empty(need, dtype=torch.float32, device=dev)
        _bump_ws_epoch()
    return ws.data_ptr(), ws.numel()


def conv2d(x: Var, mod, stride: int = 1, padding: int = 0, dilation: int = 1, out: T.Optional[torch.Tensor] = None,
           want_stats: bool = False, bn=None) -> Var:
    """nn.Conv2d forward (+ tape node for bwd-data, bwd-weight, bias grad). ``bn``: the BatchNorm2d that follows (training
    mode, mixed precision): the conv launch finishes its batch statistics when it can (see _bnstats_launch)."""
    tape = current_tape()
    xt = _check(x.t)
    if is16(xt):
        return _conv2d_bf16(x, mod, stride, padding, dilation, out, want_stats, bn)
    B, Cin, H, W = xt.shape
    w = mod.weight
    Cout = w.shape[0]
    KH, KW = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
    Ho = (H + 2 * padding - dilation * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (KW - 1) - 1) // stride + 1
    pw = packed_conv(mod, tape.enabled and x.req)
    y = out if out is not None else _new((B, Cout, Ho, Wo), xt)
    bias = mod.bias
    _lib.call("cn_conv2d_fwd_f32", xt.data_ptr(), bstride(xt), pw.fwd.data_ptr(),
              bias.data_ptr() if bias is not None else None, y.data_ptr(), bstride(y), B, Cin, H, W, Cout, KH, KW,
              stride, padding, dilation, 0, _stream())
    yv = Var(y, tape.enabled)
    if tape.enabled:
        store = current_store()

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            with side_stream(xt, dy, work=float(B) * Ho * Wo * Cin * Cout * KH * KW):
                s = _stream()
                wsp, wsn = _pad_ws(xt, dy)
                _lib.call("cn_conv2d_bwd_weight_f32", xt.data_ptr(), bstride(xt), dy.data_ptr(), bstride(dy),
                          store.grad_of(w).data_ptr(), B, Cin, H, W, Cout, KH, KW, stride, padding, dilation, wsp, wsn,
                          s)
                if bias is not None:
                    _lib.call("cn_channel_sum_f32", dy.data_ptr(), bstride(dy), B, Cout, Ho * Wo,
                              store.grad_of(bias).data_ptr(), 1, s)
            if x.req:
                dx, acc = grad_buffer(x)
                _lib.call("cn_conv2d_bwd_data_f32", dy.data_ptr(), bstride(dy), pw.bwd.data_ptr(), dx.data_ptr(),
                          bstride(dx), B, Cin, H, W, Cout, KH, KW, stride, padding, dilation, acc, _stream())
            yv.grad = None

        tape.add(bwd, (w, bias))
    return yv


def conv2d_group(xs: T.Sequence[Var], mods: T.Sequence, paddings: T.Sequence[int], dilations: T.Sequence[int],
                 stride: int = 1, bns: T.Optional[T.Sequence] = None) -> T.List[Var]:
    """G (<= 4) nn.Conv2d of identical shapes (per-conv padding / dilation) in ONE implicit-GEMM launch -- the
    dilation branches of ResidualAConv. ``xs`` may repeat one Var (branches sharing their input: backward sums the
    G input gradients in the same launch). Weight gradients stay one launch per conv."""
    import ctypes

    tape = current_tape()
    G = len(mods)
    xts = [_check(x.t) for x in xs]
    if is16(xts[0]):  # mixed precision: one grouped launch, each conv with BatchNorm statistics rows from its epilogue
        return _conv2d_group_bf16(xs, mods, paddings, dilations, stride, bns)
    B, Cin, H, W = xts[0].shape
    w0 = mods[0].weight
    Cout, KH, KW = w0.shape[0], w0.shape[2], w0.shape[3]
    for xt, m in zip(xts, mods):
        if tuple(xt.shape) != (B, Cin, H, W) or tuple(m.weight.shape) != tuple(w0.shape) or bstride(xt) != bstride(xts[0]):
            raise ValueError("conv2d_group: all inputs / weights must have the same shape")
    Ho = (H + 2 * paddings[0] - dilations[0] * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * paddings[0] - dilations[0] * (KW - 1) - 1) // stride + 1
    shared_in = all(x is xs[0] for x in xs)
    need_bwd = tape.enabled and any(x.req for x in xs)
    pws = [packed_conv(m, need_bwd) for m in mods]
    ys = [_new((B, Cout, Ho, Wo), xts[0]) for _ in range(G)]
    biases = [m.bias for m in mods]
    has_bias = biases[0] is not None
    tab = lambda ptrs: (ctypes.c_void_p * G)(*ptrs)
    pads_c = (ctypes.c_int * G)(*paddings)
    dils_c = (ctypes.c_int * G)(*dilations)
    _lib.call("cn_conv2d_fwd_grouped_f32", G, tab([t.data_ptr() for t in xts]), bstride(xts[0]),
              tab([p.fwd.data_ptr() for p in pws]), tab([b.data_ptr() for b in biases]) if has_bias else None,
              tab([y.data_ptr() for y in ys]), bstride(ys[0]), B, Cin, H, W, Cout, KH, KW, stride, pads_c, dils_c, 0,
              _stream())
    yvs = [Var(y, tape.enabled) for y in ys]
    if tape.enabled:
        store = current_store()

        def bwd():
            live = [i for i in range(G) if yvs[i].grad is not None]
            grouped_w = (len(live) == G and len(set(paddings)) == 1 and len(set(dilations)) == 1
                         and len({bstride(yvs[i].grad) for i in live}) == 1)
            with side_stream(*(list(xts) + [yvs[i].grad for i in live]),
                             work=float(G) * B * Ho * Wo * Cin * Cout * KH * KW):
                s = _stream()
                if grouped_w:  # one launch for the G weight gradients
                    wsp, wsn = _pad_ws(*([xts[i] for i in range(G)] + [yvs[i].grad for i in range(G)]))
                    _lib.call("cn_conv2d_bwd_weight_grouped_f32", G, tab([t.data_ptr() for t in xts]), bstride(xts[0]),
                              tab([yvs[i].grad.data_ptr() for i in range(G)]), bstride(yvs[0].grad),
                              tab([store.grad_of(m.weight).data_ptr() for m in mods]), B, Cin, H, W, Cout, KH, KW,
                              stride, paddings[0], dilations[0], wsp, wsn, s)
                for i in live:
                    dy, m, xt = yvs[i].grad, mods[i], xts[i]
                    if grouped_w:
                        if has_bias:
                            _lib.call("cn_channel_sum_f32", dy.data_ptr(), bstride(dy), B, Cout, Ho * Wo,
                                      store.grad_of(m.bias).data_ptr(), 1, s)
                        continue
                    wsp, wsn = _pad_ws(xt, dy)
                    _lib.call("cn_conv2d_bwd_weight_f32", xt.data_ptr(), bstride(xt), dy.data_ptr(), bstride(dy),
                              store.grad_of(m.weight).data_ptr(), B, Cin, H, W, Cout, KH, KW, stride, paddings[i],
                              dilations[i], wsp, wsn, s)
                    if has_bias:
                        _lib.call("cn_channel_sum_f32", dy.data_ptr(), bstride(dy), B, Cout, Ho * Wo,
                                  store.grad_of(m.bias).data_ptr(), 1, s)
            s = _stream()
            todo = [i for i in live if xs[i].req]
            if todo:
                bufs = []
                if shared_in:
                    dx, acc = grad_buffer(xs[0])
                    bufs = [(dx, acc)] * len(todo)
                else:
                    bufs = [grad_buffer(xs[i]) for i in todo]
                accs = {a for _, a in bufs}
                dybs = {bstride(yvs[i].grad) for i in todo}
                dxbs = {bstride(d) for d, _ in bufs}
                if len(todo) == G and len(accs) == 1 and len(dybs) == 1 and len(dxbs) == 1:
                    dyp = [yvs[i].grad.data_ptr() for i in todo]
                    wpp = [p.bwd.data_ptr() for p in pws]
                    dxp = [d.data_ptr() for d, _ in bufs]
                    khs, kws, pds, dls = [KH] * G, [KW] * G, list(paddings), list(dilations)
                    n = len(dyp)
                    ptab = lambda ptrs: (ctypes.c_void_p * n)(*ptrs)
                    itab = lambda v: (ctypes.c_int * n)(*v)
                    _lib.call("cn_conv2d_bwd_data_grouped_f32", n, ptab(dyp), dybs.pop(), ptab(wpp), ptab(dxp),
                              dxbs.pop(), B, Cin, H, W, Cout, itab(khs), itab(kws), stride, itab(pds), itab(dls),
                              accs.pop(), s)
                else:  # mixed accumulate flags / strides: conv by conv
                    first = True
                    for i, (d, a) in zip(todo, bufs):
                        dy = yvs[i].grad
                        _lib.call("cn_conv2d_bwd_data_f32", dy.data_ptr(), bstride(dy), pws[i].bwd.data_ptr(),
                                  d.data_ptr(), bstride(d), B, Cin, H, W, Cout, KH, KW, stride, paddings[i],
                                  dilations[i], a if (first or not shared_in) else 1, s)
                        first = False
            for v in yvs:
                v.grad = None

        tape.add(bwd, tuple(m.weight for m in mods) + tuple(b for b in biases if b is not None))
    return yvs


# CN_CONVT_TAPS=0: a ConvTranspose2d with stride >= kernel size (final_c's 3 x 3, stride 4) runs through the parity-class
# launches like every other one instead of the dense 1x1 contraction + pointwise scatter / gather (A/B switch).
_CONVT_TAPS = os.environ.get("CN_CONVT_TAPS", "1") != "0"


def _conv_transpose2d_taps(x: Var, mod, stride: int, padding: int, size, out: T.Optional[torch.Tensor]) -> Var:
    """nn.ConvTranspose2d(k, stride s >= k, padding) FOLLOWED BY check_upsample to ``size`` (None: no resize), fp32: no
    two taps of an input pixel meet, so the contraction is a dense 1x1 GEMM on the small grid into
    P [B, Cout*k*k, H, W] (the weight tensor viewed as [Cin][Cout*k*k]) and one pointwise pass writes
    resize(bias + scatter(P)); backward: dP = gather(resize^T(dz)) in one pass, then the 1x1 data / weight gradients (the
    weight gradient lands in the parameter's own [Cin][Cout][k][k] layout). Returns the RESIZED tensor (csrc/cn_pointwise.hip,
    cn_convt_taps_*; convolution.py:45-68 + unet_parts.py:227-309 of the reference)."""
    tape = current_tape()
    xt = x.t
    B, Cin, H, W = xt.shape
    w = mod.weight
    Cout, K = w.shape[1], w.shape[2]
    KK = K * K
    Hy = (H - 1) * stride - 2 * padding + K
    Wy = (W - 1) * stride - 2 * padding + K
    Ho, Wo = (int(size[0]), int(size[1])) if size is not None else (Hy, Wy)
    pw = packed_convT(mod, tape.enabled and x.req, taps_as_channels=True)
    P = _new((B, Cout * KK, H, W), xt)
    _lib.call("cn_conv_transpose2d_fwd_f32", xt.data_ptr(), bstride(xt), pw.fwd.data_ptr(), None, P.data_ptr(),
              bstride(P), B, Cin, H, W, Cout * KK, 1, 1, 1, 0, 0, 0, _stream())
    z = out if (out is not None and tuple(out.shape) == (B, Cout, Ho, Wo)) else _new((B, Cout, Ho, Wo), xt)
    bias = mod.bias
    _lib.call("cn_convt_taps_fwd_f32", P.data_ptr(), bstride(P), bias.data_ptr() if bias is not None else None,
              z.data_ptr(), bstride(z), B, Cout, H, W, K, stride, padding, Ho, Wo, _stream())
    del P  # (not needed by the backward pass)
    zv = Var(z, tape.enabled)
    if tape.enabled:
        store = current_store()

        def bwd():
            dz = zv.grad
            if dz is None:
                return
            dP = _new((B, Cout * KK, H, W), xt)
            _lib.call("cn_convt_taps_bwd_f32", dz.data_ptr(), bstride(dz), dP.data_ptr(), bstride(dP), B, Cout, H, W, K,
                      stride, padding, Ho, Wo, _stream())
            with side_stream(xt, dP, dz):
                s = _stream()
                wsp, wsn = _pad_ws(xt, dP)
                _lib.call("cn_conv_transpose2d_bwd_weight_f32", xt.data_ptr(), bstride(xt), dP.data_ptr(), bstride(dP),
                          store.grad_of(w).data_ptr(), B, Cin, H, W, Cout * KK, 1, 1, 1, 0, 0, wsp, wsn, s)
                if bias is not None:  # the resize weights of every output pixel sum to one: sum of dz
                    _lib.call("cn_channel_sum_f32", dz.data_ptr(), bstride(dz), B, Cout, Ho * Wo,
                              store.grad_of(bias).data_ptr(), 1, s)
            if x.req:
                dx, acc = grad_buffer(x)
                _lib.call("cn_conv_transpose2d_bwd_data_f32", dP.data_ptr(), bstride(dP), pw.bwd.data_ptr(),
                          dx.data_ptr(), bstride(dx), B, Cin, H, W, Cout * KK, 1, 1, 1, 0, 0, acc, _stream())
            zv.grad = None

        tape.add(bwd, (w, bias))
    return zv


# CN_CONVT_OUTPAD=0: ConvTranspose2d writes the reference's (2n-1)^2 tensor (rounds 1-5) instead of the 2n x 2n
# output_padding grid of the resize that follows (A/B switch).
_CONVT_OUTPAD = os.environ.get("CN_CONVT_OUTPAD", "1") != "0"


def conv_transpose2d(x: Var, mod, stride: int, padding: int, size: T.Optional[T.Tuple[int, int]] = None,
                     out: T.Optional[torch.Tensor] = None) -> Var:
    """nn.ConvTranspose2d forward (k x k, stride s, padding p, with bias).

    ``size``: the size check_upsample will resize the result to (convolution.py:45-68). When it exceeds the natural
    (2n-1)-style output by less than the stride on both axes (every site of TowerUNet), the fp32 result is computed on
    THAT grid -- the transposed convolution with output_padding = size - natural, whose top-left natural-size block is
    exactly the reference's tensor -- and returned as a Var whose ``valid`` attribute names the image inside it:
    planes of 100 x 100 / 50 x 50 are 16-byte aligned where 99 x 99 / 49 x 49 / 97 x 97 are not, so the weight gradient
    needs no aligned copy of dy (cn_pad_planes), the data gradient stages 16 bytes per lane instead of 4, and
    resize_bilinear reads / writes the stored grid in place."""
    tape = current_tape()
    xt = _check(x.t)
    if is16(xt):
        return _conv_transpose2d_bf16(x, mod, stride, padding)
    B, Cin, H, W = xt.shape
    w = mod.weight
    Cout, KH, KW = w.shape[1], w.shape[2], w.shape[3]
    Ho = (H - 1) * stride - 2 * padding + KH
    Wo = (W - 1) * stride - 2 * padding + KW
    if _CONVT_TAPS and KH == KW and stride >= KH and 0 <= padding < stride and \
            (size is None or (2 * Ho > int(size[0]) and 2 * Wo > int(size[1]))):
        return _conv_transpose2d_taps(x, mod, stride, padding, size, out)
    op = 0
    if _CONVT_OUTPAD and size is not None and stride > 1:
        dh, dw_ = int(size[0]) - Ho, int(size[1]) - Wo
        if dh == dw_ and 0 < dh < stride:
            op = dh
    Hs, Ws = Ho + op, Wo + op  # the stored grid
    pw = packed_convT(mod, tape.enabled and x.req)
    y = _new((B, Cout, Hs, Ws), xt)
    bias = mod.bias
    _lib.call("cn_conv_transpose2d_fwd_f32", xt.data_ptr(), bstride(xt), pw.fwd.data_ptr(),
              bias.data_ptr() if bias is not None else None, y.data_ptr(), bstride(y), B, Cin, H, W, Cout, KH, KW,
              stride, padding, op, 0, _stream())
    yv = Var(y, tape.enabled)
    if op:
        yv.valid = (Ho, Wo)
    if tape.enabled:
        store = current_store()

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            with side_stream(xt, dy):
                s = _stream()
                wsp, wsn = _pad_ws(xt, dy)
                _lib.call("cn_conv_transpose2d_bwd_weight_f32", xt.data_ptr(), bstride(xt), dy.data_ptr(), bstride(dy),
                          store.grad_of(w).data_ptr(), B, Cin, H, W, Cout, KH, KW, stride, padding, op, wsp, wsn, s)
                if bias is not None:  # (the padding of dy is zero: the resize adjoint writes it)
                    _lib.call("cn_channel_sum_f32", dy.data_ptr(), bstride(dy), B, Cout, Hs * Ws,
                              store.grad_of(bias).data_ptr(), 1, s)
            if x.req:
                dx, acc = grad_buffer(x)
                _lib.call("cn_conv_transpose2d_bwd_data_f32", dy.data_ptr(), bstride(dy), pw.bwd.data_ptr(),
                          dx.data_ptr(), bstride(dx), B, Cin, H, W, Cout, KH, KW, stride, padding, op, acc, _stream())
            yv.grad = None

        tape.add(bwd, (w, bias))
    return yv


def time_conv(x: Var, mod, tin: int) -> Var:
    """nn.Conv3d(kernel (k,1,1), bias=False) on x viewed as [B, Cin*Tin, H, W] -> [B, Cout*Tout, H, W]."""
    tape = current_tape()
    xt = _check(x.t)
    B, CT, H, W = xt.shape
    w = mod.weight  # [Cout, Cin, k, 1, 1]
    Cout, Cin, k = w.shape[0], w.shape[1], w.shape[2]
    assert CT == Cin * tin
    tout = tin - k + 1
    ver = current_store().version
    pw = mod.__dict__.get("_cn_packed")
    if pw is None or pw.version != ver or pw.store_id != current_store().uid:
        pw = PackedWeight()
        pw.version = ver
        pw.store_id = current_store().uid
        mod.__dict__["_cn_packed"] = pw
    if pw.fwd is None:
        n = _lib.query("cn_conv_kpad", Cin * tin) * _lib.query("cn_conv_npad", Cout * tout)
        pw.fwd = _new((n,), xt)
        _lib.call("cn_pack_timeconv_f32", w.data_ptr(), pw.fwd.data_ptr(), Cout, Cin, tin, k, 0, _stream())
    if tape.enabled and x.req and pw.bwd is None:
        n = _lib.query("cn_conv_kpad", Cout * tout) * _lib.query("cn_conv_npad", Cin * tin)
        pw.bwd = _new((n,), xt)
        _lib.call("cn_pack_timeconv_f32", w.data_ptr(), pw.bwd.data_ptr(), Cout, Cin, tin, k, 1, _stream())
    y = _new((B, Cout * tout, H, W), xt)
    _lib.call("cn_conv2d_fwd_f32", xt.data_ptr(), bstride(xt), pw.fwd.data_ptr(), None, y.data_ptr(), bstride(y), B,
              CT, H, W, Cout * tout, 1, 1, 1, 0, 1, 0, _stream())
    yv = Var(y, tape.enabled)
    if tape.enabled:
        store = current_store()

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            with side_stream(xt, dy):
                s = _stream()
                dwexp = _alloc(Cout * tout * CT, torch.float32, xt.device)
                _lib.call("cn_fill_f32", dwexp.data_ptr(), dwexp.numel(), 0.0, s)
                _lib.call("cn_conv2d_bwd_weight_f32", xt.data_ptr(), bstride(xt), dy.data_ptr(), bstride(dy),
                          dwexp.data_ptr(), B, CT, H, W, Cout * tout, 1, 1, 1, 0, 1, None, 0, s)
                _lib.call("cn_fold_timeconv_grad_f32", dwexp.data_ptr(), store.grad_of(w).data_ptr(), Cout, Cin, tin, k,
                          s)
                if _OVERLAP_WGRAD:
                    dwexp.record_stream(_side_state(xt.device)["stream"])
            s = _stream()
            if x.req:
                dx, acc = grad_buffer(x)
                _lib.call("cn_conv2d_bwd_data_f32", dy.data_ptr(), bstride(dy), pw.bwd.data_ptr(), dx.data_ptr(),
                          bstride(dx), B, CT, H, W, Cout * tout, 1, 1, 1, 0, 1, acc, s)
            yv.grad = None

        tape.add(bwd, (w,))
    return yv


_pt_ws: T.Dict[T.Tuple, torch.Tensor] = {}
# The fused PreTimeReduction family (csrc/cn_pretime.hip). CN_PRETIME_FUSED = "1" (default): training
# and inference (forward 3 launches / 1 in eval mode instead of ~20, backward 3 instead of ~20; same-box A/B in round 4:
# bf16 batch 32 at parity with the op-by-op path, fp32 batch 8 +1 %, 32 / 29 launches fewer per step); "infer": the
# inference forward only; "0": never. CN_PRETIME_BWD = "main" (default) / "side": the stream of the three backward
# launches -- queued on the weight-gradient stream, behind the encoder's 100 x 100 weight gradients, they cost 1 %.
_PRETIME_FUSED = os.environ.get("CN_PRETIME_FUSED", "1")
_PRETIME_BWD = os.environ.get("CN_PRETIME_BWD", "main")


def _zeroed_scratch(n: int, dev: torch.device) -> torch.Tensor:
    """A scratch buffer whose ticket-counter head must be zero before its first launch, zeroed ON THE STREAM THAT WILL
    USE IT. ``torch.zeros`` fills on torch's current (compute) stream, but these buffers are keyed to the
    weight-gradient side stream or an auxiliary stream, which had only waited for an event recorded BEFORE the
    allocation: the fill raced the first ticketed kernel. ``cn_fill_f32`` on ``_stream()`` is ordered by construction;
    a recycled block is safe because that stream already waits for everything the compute stream enqueued earlier."""
    ws = torch.empty(n, dtype=torch.float32, device=dev)
    _lib.call("cn_fill_f32", ws.data_ptr(), n, 0.0, _stream())
    return ws


def _pretime_ws(need: int, dev: torch.device) -> torch.Tensor:
    """Scratch of the fused PreTimeReduction calls, one per (device, stream): zero-filled once (ticket counters at its
    head; every launch leaves them zero)."""
    key = (dev, _stream())
    ws = _pt_ws.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            torch.cuda.synchronize(dev)
        ws = _pt_ws[key] = _zeroed_scratch(need, dev)
        _bump_ws_epoch()
    return ws


def pretime_reduction(x: Var, pre, in_channels: int, in_time: int) -> T.Optional[Var]:
    """PreTimeReduction (models/nunet.py:60-105) through the fused kernel family cn_pretime_*: both Conv3d stacks, their
    BatchNorm3d / BatchNorm2d, the branch sum and the LayerNorm, recomputed from x in every pass (3 launches forward in
    training, 1 in inference, 3 backward -- on the weight-gradient side stream: the stage has parameter gradients only).
    Writes fp32 NCHW, or bf16 NHWC directly when the mixed-precision region is on. Returns None for shapes / settings the
    fused kernels do not cover (the caller keeps the generic op-by-op path)."""
    import ctypes

    tape = current_tape()
    if _PRETIME_FUSED == "0" or (_PRETIME_FUSED != "1" and (tape.enabled or pre.training)):
        return None
    xt = _check(x.t)
    if is16(xt) or x.req or not xt.is_contiguous():
        return None
    B, CT, H, W = xt.shape
    C, Tn = in_channels, in_time
    br = (pre.conv3, pre.conv5)
    ln = pre.layer_norm[1]
    Cout = ln.weight.shape[0]
    HW = H * W
    training = bool(pre.training)
    bn3 = [b.seq[1] for b in br]
    bn2 = [b.seq[5] for b in br]
    if CT != C * Tn or any(b.momentum is None for b in bn3 + bn2) or \
            bn3[0].eps != bn3[1].eps or bn2[0].eps != bn2[1].eps or bn3[0].momentum != bn3[1].momentum or \
            bn2[0].momentum != bn2[1].momentum or any(b.running_mean is None for b in bn3 + bn2) or \
            any(b.training != training for b in bn3 + bn2):
        return None
    with_bwd = 1 if tape.enabled else 0
    need = int(_lib.query("cn_pretime_workspace_floats", B, C, Tn, HW, Cout, with_bwd))
    if need < 0:
        return None
    dev = xt.device
    bf16 = bf16_enabled()
    if bf16:
        y = _alloc((B, H, W, Cout), torch.bfloat16, dev).permute(0, 3, 1, 2)
        ystride, kind = Cout, 1
    else:
        y = _alloc((B, Cout, H, W), torch.float32, dev)
        ystride, kind = Cout * HW, 0
    plist = []
    for b, b3, b2 in zip(br, bn3, bn2):
        plist += [b.seq[0].weight, b.seq[3].weight, b3.weight, b3.bias, b3.running_mean, b3.running_var, b2.weight,
                  b2.bias, b2.running_mean, b2.running_var]
    plist += [ln.weight, ln.bias]
    params = (ctypes.c_void_p * 22)(*[t.data_ptr() for t in plist])
    stats_t = _alloc(2 * (2 * C + 2 * Cout), torch.float32, dev)
    offs, o = [], 0
    for _ in range(2):
        for n in (C, C, Cout, Cout):
            offs.append(o)
            o += n
    stats = (ctypes.c_void_p * 8)(*[stats_t[i:].data_ptr() for i in offs])
    bnc = (ctypes.c_float * 4)(float(bn3[0].eps), float(bn3[0].momentum), float(bn2[0].eps), float(bn2[0].momentum))
    _note_bn_update(training)
    ws = _pretime_ws(need, dev)
    _lib.call("cn_pretime_fwd_f32", xt.data_ptr(), bstride(xt), params, stats, y.data_ptr(), ystride, kind, B, C, Tn, HW,
              Cout, 1 if training else 0, bnc, float(ln.eps), ws.data_ptr(), ws.numel(), _stream())
    yv = Var(y, tape.enabled)
    if tape.enabled:
        store = current_store()
        glist = []
        for b, b3, b2 in zip(br, bn3, bn2):
            glist += [b.seq[0].weight, b.seq[3].weight, b3.weight, b3.bias, b2.weight, b2.bias]
        glist += [ln.weight, ln.bias]

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            if bf16:
                dstride = ld(dy)
                if not _dense16(dy):
                    raise RuntimeError("pretime_reduction: the output gradient must be a dense NHWC buffer")
            else:
                if not _dense_inner(dy):  # (a channel slice of a concat gradient: batch stride > Cout * HW is fine)
                    raise RuntimeError("pretime_reduction: the output gradient must be NCHW with dense planes")
                dstride = bstride(dy)
            grads = (ctypes.c_void_p * 14)(*[store.grad_of(p).data_ptr() for p in glist])
            # parameter gradients only, and the LAST node of the backward. On the compute stream: the weight-gradient
            # stream still holds the encoder's 100 x 100 weight gradients at this point, behind which these three
            # launches would queue (CN_PRETIME_BWD=side: measured 1 % slower end to end)
            ctx = side_stream(xt, dy, stats_t) if _PRETIME_BWD == "side" else contextlib.nullcontext()
            with ctx:
                wsb = _pretime_ws(need, dev)
                _lib.call("cn_pretime_bwd_f32", xt.data_ptr(), bstride(xt), params, stats, dy.data_ptr(), dstride, kind,
                          grads, B, C, Tn, HW, Cout, 1 if training else 0, bnc, float(ln.eps), wsb.data_ptr(),
                          wsb.numel(), _stream())
            yv.grad = None

        tape.add(bwd, tuple(glist))
    return yv


ACT_NONE, ACT_SILU = 0, 1

# bumped by every train-mode BatchNorm forward (the kernels update running_mean / running_var in place, which torch's
# version counters do not see): invalidates the eval-mode folded weights below
_bn_stats_epoch = 0


def _note_bn_update(training: bool) -> None:
    global _bn_stats_epoch
    if training:
        _bn_stats_epoch += 1


def _bn_momentum(bn) -> float:
    """torch's momentum=None means a cumulative moving average (factor 1/num_batches_tracked, a device counter);
    the reference never configures it and the fused kernel takes a host scalar: refuse instead of guessing."""
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm(momentum=None) (cumulative average) has no HIP kernel; the reference "
                                  "uses the default momentum=0.1 everywhere")
    return float(bn.momentum)


def bn_act(x: Var, bn, act: int, residual: T.Optional[Var] = None, channels: T.Optional[int] = None,
           training: bool = True, out: T.Optional[torch.Tensor] = None) -> Var:
    """y = act(batch_norm(x)) (+ residual). x viewed as [B][C][L] with C = ``channels`` (BatchNorm3d: L = T*H*W)."""
    tape = current_tape()
    xt = _check(x.t)
    if is16(xt):
        if channels is not None and channels != xt.shape[1]:
            raise NotImplementedError("BatchNorm3d views are fp32-only (PreTimeReduction runs in fp32)")
        return _bn_act_bf16(x, bn, act, residual, training, out)
    B = xt.shape[0]
    C = channels if channels is not None else xt.shape[1]
    L = int(xt[0].numel()) // C
    dev = xt.device
    y = out if out is not None else _new(xt.shape, xt)
    mean = _new((C,), xt)
    rstd = _new((C,), xt)
    ws = _alloc(_lib.query("cn_bn_workspace_doubles", C), torch.float64, dev)
    rt = residual.t if residual is not None else None
    mom = _bn_momentum(bn)
    use_batch = training or (bn.running_mean is None)
    _note_bn_update(training)
    _lib.call("cn_bn_act_fwd_f32", xt.data_ptr(), bstride(xt), bn.weight.data_ptr(), bn.bias.data_ptr(),
              bn.running_mean.data_ptr() if bn.running_mean is not None else None,
              bn.running_var.data_ptr() if bn.running_var is not None else None,
              rt.data_ptr() if rt is not None else None, bstride(rt) if rt is not None else 0, y.data_ptr(),
              bstride(y), mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), B, C, L, 1 if use_batch else 0, float(mom),
              float(bn.eps), act, _stream())
    req = tape.enabled
    yv = Var(y, req)
    if tape.enabled:
        store = current_store()
        gamma, beta = bn.weight, bn.bias

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            s = _stream()
            if residual is not None:
                give_grad(residual, dy)
            coef = _new((2 * C,), xt)
            if x.req:
                dx, acc = grad_buffer(x)
                dxp, dxbs = dx.data_ptr(), bstride(dx)
            else:
                dxp, dxbs, acc = None, 0, 0
            _lib.call("cn_bn_act_bwd_f32", xt.data_ptr(), bstride(xt), dy.data_ptr(), bstride(dy), mean.data_ptr(),
                      rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dxp, dxbs, store.grad_of(gamma).data_ptr(),
                      store.grad_of(beta).data_ptr(), coef.data_ptr(), ws.data_ptr(), B, C, L, 1 if use_batch else 0,
                      act, acc, 1, s)
            yv.grad = None

        tape.add(bwd, (gamma, beta))
    return yv


def bn_act_group(xs: T.Sequence[Var], bns: T.Sequence, act: int, residual: T.Optional[Var] = None,
                 sum_outputs: bool = False, training: bool = True,
                 outs: T.Optional[T.Sequence[torch.Tensor]] = None) -> T.Union[Var, T.List[Var]]:
    """G BatchNorm2d(+act) over G same-shaped tensors in one launch pair. ``sum_outputs``: returns the single Var
    ``residual + sum_g act(bn_g(x_g))`` (the ResUNet-a sum, accumulated in the order of the sequential form);
    otherwise the list of G activations."""
    import ctypes

    tape = current_tape()
    G = len(xs)
    xts = [_check(x.t) for x in xs]
    if residual is not None and not sum_outputs:
        raise ValueError("bn_act_group: a residual needs sum_outputs=True")
    if is16(xts[0]):  # mixed precision: the G branches in one launch per pass (cn_bn_act_group_*_bf16)
        return _bn_act_group_bf16(xs, bns, act, residual, sum_outputs, training, outs)
    B, C = xts[0].shape[0], xts[0].shape[1]
    L = int(xts[0][0].numel()) // C
    for t in xts:
        if tuple(t.shape) != tuple(xts[0].shape) or bstride(t) != bstride(xts[0]):
            raise ValueError("bn_act_group: inputs must have the same shape and strides")
    dev = xts[0].device
    use_batch = training or any(bn.running_mean is None for bn in bns)
    _note_bn_update(training)
    tab = lambda ptrs: (ctypes.c_void_p * G)(*ptrs)
    if outs is not None:  # caller-provided outputs (e.g. channel slices of one buffer, all with the same strides)
        ys = list(outs)
        if len(ys) != (1 if sum_outputs else G) or any(bstride(y) != bstride(ys[0]) for y in ys):
            raise ValueError("bn_act_group: outs must match the outputs and share their batch stride")
    else:
        ys = [_new(xts[0].shape, xts[0])] if sum_outputs else [_new(xts[0].shape, xts[0]) for _ in range(G)]
    means = [_new((C,), xts[0]) for _ in range(G)]
    rstds = [_new((C,), xts[0]) for _ in range(G)]
    ws = _alloc(G * _lib.query("cn_bn_workspace_doubles", C), torch.float64, dev)
    rt = residual.t if residual is not None else None
    mom = _bn_momentum(bns[0])
    has_running = all(bn.running_mean is not None for bn in bns)
    _lib.call("cn_bn_act_group_fwd_f32", G, tab([t.data_ptr() for t in xts]), bstride(xts[0]),
              tab([bn.weight.data_ptr() for bn in bns]), tab([bn.bias.data_ptr() for bn in bns]),
              tab([bn.running_mean.data_ptr() for bn in bns]) if has_running else None,
              tab([bn.running_var.data_ptr() for bn in bns]) if has_running else None,
              rt.data_ptr() if rt is not None else None, bstride(rt) if rt is not None else 0,
              (ctypes.c_void_p * len(ys))(*[y.data_ptr() for y in ys]) if not sum_outputs else tab([ys[0].data_ptr()] * G),
              bstride(ys[0]), tab([m.data_ptr() for m in means]), tab([r.data_ptr() for r in rstds]), ws.data_ptr(),
              B, C, L, 1 if use_batch else 0, float(mom), float(bns[0].eps), act, 1 if sum_outputs else 0, _stream())
    yvs = [Var(y, tape.enabled) for y in ys]
    if tape.enabled:
        store = current_store()

        def bwd():
            s = _stream()
            if sum_outputs:
                dy = yvs[0].grad
                if dy is None:
                    return
                if residual is not None:
                    give_grad(residual, dy)
                dys = [dy] * G
            else:
                dys = [v.grad for v in yvs]
                if any(d is None for d in dys):
                    raise RuntimeError("bn_act_group: every output needs a gradient")
            bufs = [grad_buffer(x) if x.req else (None, 0) for x in xs]
            dxbs = {bstride(d) for d, _ in bufs if d is not None}
            dybs = {bstride(d) for d in dys}
            if len(dxbs) > 1 or len(dybs) > 1:
                raise RuntimeError("bn_act_group: gradient buffers must share their strides")
            _lib.call("cn_bn_act_group_bwd_f32", G, tab([t.data_ptr() for t in xts]), bstride(xts[0]),
                      tab([d.data_ptr() for d in dys]), dybs.pop(), tab([m.data_ptr() for m in means]),
                      tab([r.data_ptr() for r in rstds]), tab([bn.weight.data_ptr() for bn in bns]),
                      tab([bn.bias.data_ptr() for bn in bns]),
                      tab([d.data_ptr() if d is not None else None for d, _ in bufs]), dxbs.pop() if dxbs else 0,
                      (ctypes.c_int * G)(*[a for _, a in bufs]),
                      tab([store.grad_of(bn.weight).data_ptr() for bn in bns]),
                      tab([store.grad_of(bn.bias).data_ptr() for bn in bns]), ws.data_ptr(), B, C, L,
                      1 if use_batch else 0, act, 1, s)
            for v in yvs:
                v.grad = None

        tape.add(bwd, tuple(bn.weight for bn in bns) + tuple(bn.bias for bn in bns))
    return yvs[0] if sum_outputs else yvs


def _out_ok(out: T.Optional[torch.Tensor], like: torch.Tensor) -> bool:
    """May ``out`` (a channel slice of a concat buffer handed down by TowerUNet) receive a result shaped like ``like``?"""
    return (out is not None and tuple(out.shape) == tuple(like.shape) and out.dtype == like.dtype
            and out.device == like.device and (not is16(like) or (_dense16(out) and out.stride(1) == 1)))


def layer_norm_c(x: Var, ln, residual: T.Optional[Var] = None, out: T.Optional[torch.Tensor] = None) -> Var:
    """nn.LayerNorm over the channel axis of an NCHW buffer (+ residual)."""
    tape = current_tape()
    xt = _check(x.t)
    if is16(xt):
        return _layer_norm_c_bf16(x, ln, residual, out)
    B, C = xt.shape[0], xt.shape[1]
    L = int(xt[0, 0].numel())
    y = out if _out_ok(out, xt) and _dense_inner(out) else _new(xt.shape, xt)
    mu = _new((B, L), xt)
    rstd = _new((B, L), xt)
    rt = residual.t if residual is not None else None
    _lib.call("cn_layernorm_c_fwd_f32", xt.data_ptr(), bstride(xt), ln.weight.data_ptr(), ln.bias.data_ptr(),
              rt.data_ptr() if rt is not None else None, bstride(rt) if rt is not None else 0, y.data_ptr(),
              bstride(y), mu.data_ptr(), rstd.data_ptr(), B, C, L, float(ln.eps), _stream())
    yv = Var(y, tape.enabled)
    if tape.enabled:
        store = current_store()

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            if residual is not None:
                give_grad(residual, dy)
            dx, acc = grad_buffer(x)
            nws = _lib.query("cn_layernorm_c_workspace_floats", B, C, L)
            ws = _alloc(max(nws, 1), torch.float32, xt.device)
            _lib.call("cn_layernorm_c_bwd_f32", xt.data_ptr(), bstride(xt), dy.data_ptr(), bstride(dy),
                      ln.weight.data_ptr(), mu.data_ptr(), rstd.data_ptr(), dx.data_ptr(), bstride(dx),
                      store.grad_of(ln.weight).data_ptr(), store.grad_of(ln.bias).data_ptr(), B, C, L, acc,
                      ws.data_ptr(), nws, _stream())
            yv.grad = None

        tape.add(bwd, (ln.weight, ln.bias))
    return yv


def na2d(qkv: Var, heads: int, kernel_size: int, dilation: int, attn_drop: float = 0.0) -> Var:
    """Neighborhood attention core on a [B, 3C, H, W] qkv buffer -> [B, C, H, W]."""
    tape = current_tape()
    qt = _check(qkv.t)
    if is16(qt):
        return _na2d_bf16(qkv, heads, kernel_size, dilation, attn_drop)
    B, C3, H, W = qt.shape
    C = C3 // 3
    out = _new((B, C, H, W), qt)
    attn = _new((B, heads, kernel_size * kernel_size, H, W), qt)
    seed = _next_seed() if attn_drop > 0.0 else 0
    stepw = _step_word(qt.device).data_ptr() if attn_drop > 0.0 else None
    step_fwd = _rng["step"]
    _lib.call("cn_na2d_fwd_f32", qt.data_ptr(), bstride(qt), out.data_ptr(), bstride(out), attn.data_ptr(), B, C,
              heads, H, W, kernel_size, dilation, float(attn_drop), seed, stepw, _stream())
    ov = Var(out, tape.enabled)
    if tape.enabled:

        def bwd():
            do = ov.grad
            if do is None:
                return
            dattn = _alloc_like(attn)
            if qkv.grad is None:
                dq = _new(qt.shape, qt)
                _lib.call("cn_na2d_bwd_f32", qt.data_ptr(), bstride(qt), do.data_ptr(), bstride(do), attn.data_ptr(),
                          dattn.data_ptr(), dq.data_ptr(), bstride(dq), B, C, heads, H, W, kernel_size, dilation,
                          float(attn_drop), _bwd_seed(seed, step_fwd), stepw, _stream())
                qkv.grad = dq
            else:  # pragma: no cover - qkv has a single consumer in TowerUNet
                dq = _new(qt.shape, qt)
                _lib.call("cn_na2d_bwd_f32", qt.data_ptr(), bstride(qt), do.data_ptr(), bstride(do), attn.data_ptr(),
                          dattn.data_ptr(), dq.data_ptr(), bstride(dq), B, C, heads, H, W, kernel_size, dilation,
                          float(attn_drop), _bwd_seed(seed, step_fwd), stepw, _stream())
                give_grad(qkv, dq)
            ov.grad = None

        tape.add(bwd)
    return ov


def spatial_channel_attention(skip: Var, out: Var, mod) -> Var:
    """out * (1 + gamma * 0.5 * (channel_attention(skip) + spatial_attention(skip))) -- SpatialChannelAttention
    (nn/modules/attention.py:89-126) applied the way ResidualAConv does (convolution.py:388-393).
    ``mod``: the host mirror with .channel_attention.fc1/.fc2 (Conv2d 1x1 pairs), .spatial_attention.conv, .gamma."""
    tape = current_tape()
    st, ot = _check(skip.t), _check(out.t)
    if is16(st):
        raise NotImplementedError("attention_weights='spatial_channel' has no bf16 kernel (fp32 only)")
    B, C, H, W = st.shape
    L = H * W
    Ch = C // 2
    dev = st.device
    fc1, fc2 = mod.channel_attention.fc1, mod.channel_attention.fc2
    w1a, w2a, w1m, w2m = fc1[0].weight, fc1[2].weight, fc2[0].weight, fc2[2].weight
    gamma = mod.gamma
    f = lambda *shape: _alloc(shape, torch.float32, dev)
    avg, mx, ca = f(B, C), f(B, C), f(B, C)
    hpre_a, hpre_m = f(B, Ch), f(B, Ch)
    idx = _alloc((B, C), torch.int32, dev)
    cidx = _alloc((B, L), torch.int32, dev)
    pooled = f(B, 2, H, W)
    s = _stream()
    _lib.call("cn_sca_pool_fwd_f32", st.data_ptr(), bstride(st), B, C, L, avg.data_ptr(), mx.data_ptr(), idx.data_ptr(),
              pooled.data_ptr(), cidx.data_ptr(), s)
    _lib.call("cn_sca_mlp_fwd_f32", avg.data_ptr(), mx.data_ptr(), w1a.data_ptr(), w2a.data_ptr(), w1m.data_ptr(),
              w2m.data_ptr(), hpre_a.data_ptr(), hpre_m.data_ptr(), ca.data_ptr(), B, C, Ch, s)
    pv = Var(pooled, tape.enabled)
    shared = {}  # d ca handed from the apply node to the pooling node
    if tape.enabled:
        store = current_store()

        def bwd_pool():  # recorded first => runs last: after the apply node and the 3x3 conv's backward
            dca = shared.pop("dca", None)
            if dca is None:
                return
            s2 = _stream()
            davg, dmx = f(B, C), f(B, C)
            _lib.call("cn_sca_mlp_bwd_f32", avg.data_ptr(), mx.data_ptr(), w1a.data_ptr(), w2a.data_ptr(),
                      w1m.data_ptr(), w2m.data_ptr(), hpre_a.data_ptr(), hpre_m.data_ptr(), ca.data_ptr(),
                      dca.data_ptr(), store.grad_of(w1a).data_ptr(), store.grad_of(w2a).data_ptr(),
                      store.grad_of(w1m).data_ptr(), store.grad_of(w2m).data_ptr(), davg.data_ptr(), dmx.data_ptr(),
                      B, C, Ch, s2)
            if skip.req:
                dpool = pv.grad
                if dpool is None:
                    dpool = _alloc_like(pooled)
                    _lib.call("cn_fill_f32", dpool.data_ptr(), dpool.numel(), 0.0, s2)
                dx, acc = grad_buffer(skip)
                _lib.call("cn_sca_pool_bwd_f32", davg.data_ptr(), dmx.data_ptr(), idx.data_ptr(), dpool.data_ptr(),
                          cidx.data_ptr(), dx.data_ptr(), bstride(dx), B, C, L, acc, s2)
            pv.grad = None

        tape.add(bwd_pool, (w1a, w2a, w1m, w2m))
    sconv = thin_conv3x3(pv, [mod.spatial_attention.conv], grouped=False)  # [B,1,H,W]
    y = _new(ot.shape, ot)
    _lib.call("cn_sca_apply_fwd_f32", ot.data_ptr(), bstride(ot), ca.data_ptr(), sconv.t.data_ptr(), gamma.data_ptr(),
              y.data_ptr(), bstride(y), B, C, L, s)
    yv = Var(y, tape.enabled)
    if tape.enabled:

        def bwd_apply():
            dy = yv.grad
            if dy is None:
                return
            s2 = _stream()
            dca, dsconv, scratch = f(B, C), f(B, 1, H, W), f(B * C)
            if out.req:
                do, acc = grad_buffer(out)
                dop, dobs = do.data_ptr(), bstride(do)
            else:
                dop, dobs, acc = None, 0, 0
            _lib.call("cn_sca_apply_bwd_f32", dy.data_ptr(), bstride(dy), ot.data_ptr(), bstride(ot), ca.data_ptr(),
                      sconv.t.data_ptr(), gamma.data_ptr(), dop, dobs, acc, dca.data_ptr(), dsconv.data_ptr(),
                      store.grad_of(gamma).data_ptr(), scratch.data_ptr(), B, C, L, s2)
            shared["dca"] = dca
            give_grad(sconv, dsconv)
            yv.grad = None

        tape.add(bwd_apply, (gamma,))
    return yv


def resize_bilinear(x: Var, size: T.Tuple[int, int], out: T.Optional[torch.Tensor] = None) -> Var:
    """F.interpolate(mode='bilinear', align_corners=True); identity when the size already matches. A Var with a
    ``valid`` attribute (conv_transpose2d on the output_padding grid) is resized from the image INSIDE its stored grid."""
    tape = current_tape()
    xt = _check(x.t)
    B, C, Hp, Wp = xt.shape
    Hi, Wi = x.valid if x.valid is not None else (Hp, Wp)
    Ho, Wo = int(size[0]), int(size[1])
    if (Hi, Wi) == (Ho, Wo) and out is None:
        return x
    if is16(xt):
        return _resize_bilinear_bf16(x, (Ho, Wo), out)
    y = out if out is not None else _new((B, C, Ho, Wo), xt)
    _lib.call("cn_bilinear_fwd_f32", xt.data_ptr(), bstride(xt), y.data_ptr(), bstride(y), B, C, Hi, Wi, Ho, Wo,
              Hp, Wp, _stream())
    yv = Var(y, tape.enabled and x.req)
    if tape.enabled and x.req:

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            dx, acc = grad_buffer(x)
            _lib.call("cn_bilinear_bwd_f32", dy.data_ptr(), bstride(dy), dx.data_ptr(), bstride(dx), B, C, Hi, Wi, Ho,
                      Wo, Hp, Wp, acc, _stream())
            yv.grad = None

        tape.add(bwd)
    return yv


def split_channels(v: Var, sizes: T.Sequence[int]) -> T.List[Var]:
    """Channel slices of ``v`` as Vars (views, no copies); their gradients land in v's gradient buffer."""
    outs, c0 = [], 0
    for c in sizes:
        sv = Var(v.t[:, c0:c0 + c], v.req)
        sv.parent = (v, c0, c0 + c)
        outs.append(sv)
        c0 += c
    return outs


def join_channels(parts: T.Sequence[Var], buf: torch.Tensor) -> Var:
    """The parts ARE the consecutive channel slices of ``buf`` (written in place by ops with ``out=``):
    torch.cat without copies. Backward hands each part its slice of the gradient."""
    tape = current_tape()
    c0 = 0
    for p in parts:
        c = p.t.shape[1]
        if p.t.data_ptr() != buf[:, c0:c0 + c].data_ptr() or \
                (not is16(buf) and buf.shape[0] > 1 and bstride(p.t) != bstride(buf)):
            raise RuntimeError("join_channels: parts must be the channel slices of buf, in order")
        c0 += c
    yv = Var(buf, tape.enabled)
    if tape.enabled:

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            o = 0
            for p in parts:
                c = p.t.shape[1]
                give_grad(p, dy[:, o:o + c])
                o += c
            yv.grad = None

        tape.add(bwd)
    return yv


def thin_conv3x3(x: Var, mods: T.Sequence, grouped: bool, dilation: int = 1,
                 out: T.Optional[torch.Tensor] = None) -> Var:
    """Several thin 3x3 'same' nn.Conv2d (weights [CP][Cin][3][3], padding == dilation) in one direct-kernel pass:
    ``grouped=False``: all read x; ``grouped=True``: conv g reads channels [g*Cin, (g+1)*Cin) of x.
    Output [B, len(mods)*CP, H, W] (channel = g*CP + c)."""
    tape = current_tape()
    xt = _check(x.t)
    if is16(xt):
        return _thin_conv3x3_bf16(x, mods, grouped, dilation, out)
    B, Cx, H, W = xt.shape
    n = len(mods)
    w0 = mods[0].weight
    CP, Cin = w0.shape[0], w0.shape[1]
    for m in mods:
        if tuple(m.weight.shape) != (CP, Cin, 3, 3):
            raise ValueError("thin_conv3x3: all weight sets must be [CP][Cin][3][3]")
    if Cx != (n * Cin if grouped else Cin):
        raise ValueError("thin_conv3x3: input channels do not match the weights")
    ws = [m.weight for m in mods]
    bs = [m.bias for m in mods]
    has_bias = bs[0] is not None
    wtab = _ptr_table([w.data_ptr() for w in ws])
    btab = _ptr_table([b.data_ptr() for b in bs]) if has_bias else None
    y = out if out is not None else _new((B, n * CP, H, W), xt)
    g = 1 if grouped else 0
    wpk = _new((Cin * 84,), xt) if (n, CP, g) == (3, 3, 0) and Cin >= 16 else None  # per-call packed weights
    _lib.call("cn_thin_conv3x3_fwd_f32", xt.data_ptr(), bstride(xt), wtab, btab, y.data_ptr(), bstride(y), B, Cin, H, W,
              n, CP, g, dilation, wpk.data_ptr() if wpk is not None else None, _stream())
    yv = Var(y, tape.enabled)
    if tape.enabled:
        store = current_store()

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            with side_stream(xt, dy):
                s = _stream()
                dwtab = _ptr_table([store.grad_of(w).data_ptr() for w in ws])
                _lib.call("cn_thin_conv3x3_bwd_weight_f32", xt.data_ptr(), bstride(xt), dy.data_ptr(), bstride(dy),
                          dwtab, B, Cin, H, W, n, CP, g, dilation, s)
                if has_bias:
                    for i, b in enumerate(bs):
                        dyi = dy[:, i * CP:(i + 1) * CP]
                        _lib.call("cn_channel_sum_f32", dyi.data_ptr(), bstride(dy), B, CP, H * W,
                                  store.grad_of(b).data_ptr(), 1, s)
            s = _stream()
            if x.req:
                dx, acc = grad_buffer(x)
                _lib.call("cn_thin_conv3x3_bwd_data_f32", dy.data_ptr(), bstride(dy), wtab, dx.data_ptr(), bstride(dx),
                          B, Cin, H, W, n, CP, g, dilation, acc, wpk.data_ptr() if wpk is not None else None, s)
            yv.grad = None

        tape.add(bwd, tuple(ws) + tuple(b for b in bs if b is not None))
    return yv


def cat_channels(parts: T.Sequence[Var], buf: T.Optional[torch.Tensor] = None) -> Var:
    """torch.cat(dim=1). Backward hands each input a channel-slice VIEW of the gradient (no copies).
    ``buf``: a preallocated [B, sum C, H, W] buffer; parts that were already produced in place in their slice of it
    (ops with ``out=``) are not copied."""
    tape = current_tape()
    t0 = _check(parts[0].t)
    B, H, W = t0.shape[0], t0.shape[2], t0.shape[3]
    Ctot = sum(p.t.shape[1] for p in parts)
    y = buf if buf is not None else _new((B, Ctot, H, W), t0)
    if tuple(y.shape) != (B, Ctot, H, W):
        raise ValueError("cat_channels: buffer shape does not match the parts")
    off = 0
    for p in parts:
        c = p.t.shape[1]
        dst = y[:, off:off + c]
        if is16(y):
            if p.t.data_ptr() != dst.data_ptr():
                _lib.call("cn_copy_bf16", _check(p.t).data_ptr(), ld(p.t), dst.data_ptr(), ld(y), B * H * W, c, 0,
                          _stream())
            off += c
            continue
        in_place = p.t.data_ptr() == dst.data_ptr() and (B == 1 or bstride(p.t) == bstride(y))
        if not in_place:
            _lib.call("cn_copy_f32", p.t.data_ptr(), bstride(p.t), dst.data_ptr(), bstride(y), B, c * H * W, 0,
                      _stream())
        off += c
    yv = Var(y, tape.enabled)
    if tape.enabled:

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            o = 0
            for p in parts:
                c = p.t.shape[1]
                give_grad(p, dy[:, o:o + c])
                o += c
            yv.grad = None

        tape.add(bwd)
    return yv


def add(a: Var, b: Var) -> Var:
    tape = current_tape()
    at, bt = _check(a.t), _check(b.t)
    B = at.shape[0]
    n = int(at[0].numel())
    y = _new(at.shape, at)
    if is16(at):
        _lib.call("cn_add_bf16", at.data_ptr(), ld(at), bt.data_ptr(), ld(bt), y.data_ptr(), ld(y), _rows(at),
                  at.shape[1], _stream())
    else:
        _lib.call("cn_add_f32", at.data_ptr(), bstride(at), bt.data_ptr(), bstride(bt), y.data_ptr(), bstride(y), B, n,
                  _stream())
    yv = Var(y, tape.enabled)
    if tape.enabled:

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            give_grad(a, dy)
            if b.req:
                if b.grad is None:  # never let two Vars alias one gradient buffer
                    cp = _new(dy.shape, dy)
                    if is16(dy):
                        _lib.call("cn_copy_bf16", dy.data_ptr(), ld(dy), cp.data_ptr(), ld(cp), _rows(dy), dy.shape[1],
                                  0, _stream())
                    else:
                        _lib.call("cn_copy_f32", dy.data_ptr(), bstride(dy), cp.data_ptr(), bstride(cp), B, n, 0,
                                  _stream())
                    b.grad = cp
                else:
                    give_grad(b, dy)
            yv.grad = None

        tape.add(bwd)
    return yv


def _ptr_table(ptrs: T.Sequence[int]):
    import ctypes

    return (ctypes.c_void_p * len(ptrs))(*ptrs)


def final_combine(ha: Var, hb: Var, hc: Var, params: T.Sequence[torch.nn.Parameter], smooth: float) -> T.Tuple[Var, Var, Var]:
    """Fused TowerUNetFinalCombine. ``params``: 16 scalar parameters in the layout of cn_final_combine_fwd_f32."""
    tape = current_tape()
    a, b, c = _check(ha.t), _check(hb.t), _check(hc.t)
    B, _, H, W = a.shape
    HW = H * W
    for t in (a, b, c):
        if not t.is_contiguous():
            raise RuntimeError("final_combine expects dense [B,3,H,W] tower outputs")
    ptab = _ptr_table([p.data_ptr() for p in params])
    dist, edge, crop = (_new((B, 1, H, W), a) for _ in range(3))
    _lib.call("cn_final_combine_fwd_f32", a.data_ptr(), b.data_ptr(), c.data_ptr(), ptab, dist.data_ptr(),
              edge.data_ptr(), crop.data_ptr(), B, HW, float(smooth), _stream())
    outs = tuple(Var(t, tape.enabled) for t in (dist, edge, crop))
    if tape.enabled:
        store = current_store()

        def bwd():
            zeros = None
            gs = []
            for v in outs:
                if v.grad is None:
                    if zeros is None:
                        zeros = _new(dist.shape, dist)
                        _lib.call("cn_fill_f32", zeros.data_ptr(), zeros.numel(), 0.0, _stream())
                    gs.append(zeros)
                else:
                    gs.append(v.grad)
            das = []
            for v in (ha, hb, hc):
                if v.grad is not None:
                    raise RuntimeError("final_combine inputs must have a single consumer")
                v.grad = _new(v.t.shape, v.t)
                das.append(v.grad)
            dtab = _ptr_table([store.grad_of(p).data_ptr() for p in params])
            _lib.call("cn_final_combine_bwd_f32", a.data_ptr(), b.data_ptr(), c.data_ptr(), ptab, dist.data_ptr(),
                      edge.data_ptr(), crop.data_ptr(), gs[0].data_ptr(), gs[1].data_ptr(), gs[2].data_ptr(),
                      das[0].data_ptr(), das[1].data_ptr(), das[2].data_ptr(), dtab, B, HW, float(smooth), _stream())
            for v in outs:
                v.grad = None

        tape.add(bwd, tuple(params))
    return outs


# loss kinds / target modes of cn_tanimoto_*
LOSS_KINDS = {"TanimotoComplementLoss": 0, "TanimotoDistLoss": 1, "TanimotoCombined": 2}
TGT_FLOAT, TGT_EQ, TGT_RANGE, TGT_ONEHOT = 0, 1, 2, 3
MSK_NONE, MSK_LABEL, MSK_I64, MSK_F32 = 0, 1, 2, 3


def tanimoto_loss(pred: Var, *, target_f: T.Optional[torch.Tensor] = None, labels: T.Optional[torch.Tensor] = None,
                  mask: T.Optional[torch.Tensor] = None, target_mode: int, mask_mode: int, klass: int = 0,
                  loss_kind: int = 0, weight: float = 1.0, smooth: float = 1e-5, depth: int = 5,
                  total: T.Optional[torch.Tensor] = None) -> torch.Tensor:
    """Batch-mean Tanimoto loss of ``pred`` [B,C,H,W]; returns a 1-element device tensor.

    The tape node writes ``weight * dL/dpred`` into pred's gradient; ``total`` (1-element device
    tensor), when given, is incremented by ``weight * loss`` on the device.
    """
    tape = current_tape()
    pt = _check(pred.t)
    B, C = pt.shape[0], pt.shape[1]
    HW = int(pt[0, 0].numel())
    dev = pt.device
    if labels is not None and labels.dtype != torch.int64:
        raise RuntimeError("labels must be int64")
    if target_f is not None:
        _check(target_f)
        if not target_f.is_contiguous():
            raise RuntimeError("float target must be contiguous")
    sums = _alloc(5 * B, torch.float64, dev)
    coef = _alloc(4 * B, torch.float32, dev)
    loss = _alloc(1, torch.float32, dev)
    tf = target_f.data_ptr() if target_f is not None else None
    lb = labels.data_ptr() if labels is not None else None
    mk = mask.data_ptr() if mask is not None else None
    _lib.call("cn_tanimoto_fwd_f32", pt.data_ptr(), bstride(pt), tf, lb, mk, target_mode, mask_mode, klass, B, C, HW,
              loss_kind, smooth, depth, sums.data_ptr(), coef.data_ptr(), loss.data_ptr(), float(weight),
              total.data_ptr() if total is not None else None, _stream())
    if tape.enabled and pred.req:

        def bwd():
            dp, acc = grad_buffer(pred)
            _lib.call("cn_tanimoto_bwd_f32", pt.data_ptr(), bstride(pt), tf, lb, mk, target_mode, mask_mode, klass, B,
                      C, HW, coef.data_ptr(), float(weight), dp.data_ptr(), bstride(dp), acc, _stream())
            _keep = (target_f, labels, mask)  # noqa: F841  keep inputs alive until backward has run

        tape.add(bwd)
    return loss


def tanimoto_loss_multi(preds: T.Sequence[Var], terms: T.Sequence[T.Dict[str, T.Any]], *, loss_kind: int = 0,
                        weights: T.Optional[T.Sequence[float]] = None, smooth: float = 1e-5, depth: int = 5,
                        total: T.Optional[torch.Tensor] = None) -> torch.Tensor:
    """The n (<= 4) Tanimoto losses of calc_loss in ONE launch per pass (cn_tanimoto_multi_*_f32): ``terms[h]`` holds
    tanimoto_loss()'s keyword arguments for head h (target_f / labels / mask / target_mode / mask_mode / klass). Returns
    the n per-head batch means (device tensor); ``total`` (1 element) is WRITTEN with sum_h weights[h] * loss_h. The one
    tape node writes weights[h] * dL_h/dpred_h into every head's gradient."""
    import struct

    tape = current_tape()
    n = len(preds)
    weights = [1.0] * n if weights is None else [float(w) for w in weights]
    pts = [_check(p.t) for p in preds]
    B = pts[0].shape[0]
    HW = int(pts[0][0, 0].numel())
    dev = pts[0].device
    for pt in pts:
        if pt.shape[0] != B or int(pt[0, 0].numel()) != HW:
            raise ValueError("tanimoto_loss_multi: the heads must share batch size and H*W")
    for kw in terms:
        lab, tf = kw.get("labels"), kw.get("target_f")
        if lab is not None and lab.dtype != torch.int64:
            raise RuntimeError("labels must be int64")
        if tf is not None and not _check(tf).is_contiguous():
            raise RuntimeError("float target must be contiguous")
    sums = _alloc(5 * B * n, torch.float64, dev)
    coef = _alloc(4 * B * n, torch.float32, dev)
    loss = _alloc(n, torch.float32, dev)

    def records(grads):
        buf = bytearray()
        for h, (pt, kw) in enumerate(zip(pts, terms)):
            ptr = lambda t: t.data_ptr() if t is not None else 0
            dp, dbs, acc = grads[h] if grads is not None else (0, 0, 0)
            buf += struct.pack("<QqQQQQqiiiifi", pt.data_ptr(), bstride(pt), ptr(kw.get("target_f")), ptr(kw.get("labels")),
                               ptr(kw.get("mask")), dp, dbs, int(kw["target_mode"]), int(kw["mask_mode"]),
                               int(kw.get("klass", 0)), pt.shape[1], weights[h], acc)
        return bytes(buf)

    _lib.call("cn_tanimoto_multi_fwd_f32", n, records(None), B, HW, loss_kind, smooth, depth, sums.data_ptr(),
              coef.data_ptr(), loss.data_ptr(), total.data_ptr() if total is not None else None, _stream())
    if tape.enabled and any(p.req for p in preds):

        def bwd():
            grads = []
            for p in preds:
                dp, acc = grad_buffer(p)
                grads.append((dp.data_ptr(), bstride(dp), acc))
            _lib.call("cn_tanimoto_multi_bwd_f32", n, records(grads), B, HW, coef.data_ptr(), _stream())
            _keep = terms  # noqa: F841  keep targets / labels / masks alive until backward has run

        tape.add(bwd)
    return loss


# ---------------------------------------------------------------------------
# dropout / pooling
# ---------------------------------------------------------------------------
_rng = {"seed": 0x5EED, "calls": 0, "step": 0, "words": {}}


def manual_seed(seed: int) -> None:
    """Seed of the counter-based dropout masks. A mask's seed has a HOST part -- (seed, index of the dropout call
    inside the step), a launch argument and therefore constant inside a recorded launch plan -- and a DEVICE part, the
    step word (one 64-bit word per device, bumped once per training forward by `begin_rng_step`), which the kernels add
    to it. Eager and replayed steps bump the same word the same way, so they draw identical masks."""
    _rng["seed"] = int(seed) & 0xFFFFFFFFFFFF
    _rng["calls"] = 0
    _rng["step"] = 0
    for w in _rng["words"].values():
        w.zero_()


def _step_word(dev) -> torch.Tensor:
    w = _rng["words"].get(dev)
    if w is None:
        w = _rng["words"][dev] = torch.zeros(1, dtype=torch.int64, device=dev)
    return w


def _host_step_inc() -> None:
    _rng["step"] += 1


def begin_rng_step(dev) -> None:
    """Start of a training forward with dropout: the per-step call counter restarts and the device step word advances
    (one 1-thread launch on the compute stream; part of a recorded plan like any other launch, as is the host mirror
    of the word's value)."""
    _rng["calls"] = 0
    _py_op(_host_step_inc)
    _lib.call("cn_rng_advance_u64", _step_word(dev).data_ptr(), 1, 0, _stream())


_STEP_MULT = 0xD1B54A32D192ED03  # cn_step_seed (cn_common.h)


def _bwd_seed(seed: int, step_fwd: int) -> int:
    """Seed for a backward mask launch: the kernels add the CURRENT step word; if another training forward has bumped
    it since this node's forward (two forwards before a backward in drop-in mode), compensate on the host."""
    d = step_fwd - _rng["step"]
    return seed if d == 0 else (seed + d * _STEP_MULT) & 0xFFFFFFFFFFFFFFFF


def _next_seed() -> int:
    _rng["calls"] += 1
    return ((_rng["seed"] << 16) ^ (_rng["calls"] * 0x9E3779B1)) & 0xFFFFFFFFFFFFFFFF


def dropout(x: Var, p: float, channelwise: bool, training: bool, out: T.Optional[torch.Tensor] = None) -> Var:
    """nn.Dropout2d (channelwise) / nn.Dropout; identity in eval mode or for p == 0 (``out`` is then ignored: the caller
    hands it to the producer of x instead)."""
    if not training or p <= 0.0:
        return x
    tape = current_tape()
    xt = _check(x.t)
    B, C = xt.shape[0], xt.shape[1]
    L = int(xt.shape[2] * xt.shape[3]) if is16(xt) else int(xt[0, 0].numel())
    seed = _next_seed()
    stepw = _step_word(xt.device).data_ptr()
    step_fwd = _rng["step"]
    y = out if _out_ok(out, xt) and (is16(xt) or _dense_inner(out)) else _new(xt.shape, xt)
    cw = 1 if channelwise else 0
    if is16(xt):  # mixed precision: the same counter-based masks on the NHWC buffer
        _lib.call("cn_dropout_bf16", xt.data_ptr(), ld(xt), y.data_ptr(), ld(y), B, C, L, float(p), seed, stepw, cw, 0,
                  _stream())
    else:
        _lib.call("cn_dropout_f32", xt.data_ptr(), bstride(xt), y.data_ptr(), bstride(y), B, C, L, float(p), seed, stepw,
                  cw, 0, _stream())
    yv = Var(y, tape.enabled and x.req)
    if tape.enabled and x.req:

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            dx, acc = grad_buffer(x)
            if is16(dy):
                _lib.call("cn_dropout_bf16", dy.data_ptr(), ld(dy), dx.data_ptr(), ld(dx), B, C, L, float(p),
                          _bwd_seed(seed, step_fwd), stepw, cw, acc, _stream())
            else:
                _lib.call("cn_dropout_f32", dy.data_ptr(), bstride(dy), dx.data_ptr(), bstride(dx), B, C, L, float(p),
                          _bwd_seed(seed, step_fwd), stepw, cw, acc, _stream())
            yv.grad = None

        tape.add(bwd)
    return yv


def adaptive_max_pool2d(x: Var, size: T.Tuple[int, int]) -> Var:
    """F.adaptive_max_pool2d(x, output_size=size)."""
    tape = current_tape()
    xt = _check(x.t)
    if is16(xt):
        raise NotImplementedError("pool_by_max has no bf16 kernel (fp32 only)")
    B, C, Hi, Wi = xt.shape
    Ho, Wo = int(size[0]), int(size[1])
    y = _new((B, C, Ho, Wo), xt)
    idx = _alloc((B, C, Ho, Wo), torch.int32, xt.device)
    _lib.call("cn_adaptive_maxpool_fwd_f32", xt.data_ptr(), bstride(xt), y.data_ptr(), bstride(y), idx.data_ptr(), B, C,
              Hi, Wi, Ho, Wo, _stream())
    yv = Var(y, tape.enabled and x.req)
    if tape.enabled and x.req:

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            dx, acc = grad_buffer(x)
            _lib.call("cn_adaptive_maxpool_bwd_f32", dy.data_ptr(), bstride(dy), idx.data_ptr(), dx.data_ptr(),
                      bstride(dx), B, C, Hi, Wi, Ho, Wo, acc, _stream())
            yv.grad = None

        tape.add(bwd)
    return yv


# ---------------------------------------------------------------------------
# mixed-precision (bf16 NHWC) op bodies: activations / activation gradients bf16, parameters + statistics fp32
# ---------------------------------------------------------------------------
_WS16_MAX_FLOATS = 96 << 20  # cap of the weight-gradient scratch (384 MB); the pixel split shrinks to fit


def _ws16(need: int, dev: torch.device, pool_name: str = "wgrad") -> T.Tuple[int, int]:
    """(pointer, floats) of a persistent fp32 scratch of the bf16 kernels. One buffer per (device, purpose): the
    weight-gradient slices live on the side stream, BatchNorm partial sums on the main stream."""
    need = int(min(max(need, 1 << 20), _WS16_MAX_FLOATS))
    if pool_name == "wgrad":
        ss = _ss_state()
        if ss is not None:  # deferred slice sums: this call's slices must outlive the next call
            return ss.take(need)
    pool = getattr(_state, "ws16_pool", None)
    if pool is None:
        pool = _state.ws16_pool = {}
    # ("wgrad" / "side" live on the ONE weight-gradient stream; every other pool belongs to the stream that launches)
    key = (dev, pool_name) if pool_name in ("wgrad", "side") else (dev, pool_name, getattr(_state, "stream_override", None))
    ws = pool.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:  # growing: the old buffer may still be in use by launches in flight on either stream
            torch.cuda.synchronize(dev)
        # zero-filled: the single-launch reductions keep their ticket counters in the first words of these buffers
        # (zero on entry, left zero on exit)
        ws = pool[key] = _zeroed_scratch(need, dev)
        _bump_ws_epoch()
    return ws.data_ptr(), ws.numel()


def _bn_ws16(C: int, dev: torch.device, pool_name: str = "bn") -> int:
    return _ws16(_lib.query("cn_bn_workspace_floats_bf16", C), dev, pool_name)[0]


def to_bf16(x: Var) -> Var:
    """fp32 NCHW -> bf16 NHWC at the entry of the mixed-precision region (gradient converted back)."""
    tape = current_tape()
    xt = _check(x.t)
    if is16(xt):
        return x
    B, C, H, W = xt.shape
    y = _alloc((B, H, W, C), torch.bfloat16, xt.device).permute(0, 3, 1, 2)
    if C % 8:
        raise RuntimeError("the bf16 region needs channel counts that are multiples of 8")
    _lib.call("cn_convert_f32nchw_to_bf16nhwc", xt.data_ptr(), bstride(xt), y.data_ptr(), ld(y), B, C, C, H * W,
              _stream())
    yv = Var(y, tape.enabled and x.req)
    if tape.enabled and x.req:

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            dx, acc = grad_buffer(x)
            _lib.call("cn_convert_bf16nhwc_to_f32nchw", dy.data_ptr(), ld(dy), dx.data_ptr(), bstride(dx), B, C, H * W,
                      acc, _stream())
            yv.grad = None

        tape.add(bwd)
    return yv


def to_f32(x: Var) -> Var:
    """bf16 NHWC -> fp32 NCHW inside the mixed-precision region: the way out for an op whose bf16 kernel does not cover a
    shape (LayerNorm over a channel count that is not 8 x a power of two, attention head dimensions outside 4..64 --
    widths like hidden 24 / 40 / 48 / 96): the op runs through its fp32 kernel and to_bf16() brings the result back."""
    tape = current_tape()
    xt = _check(x.t)
    if not is16(xt):
        return x
    B, C, H, W = xt.shape
    y = _alloc((B, C, H, W), torch.float32, xt.device)
    _lib.call("cn_convert_bf16nhwc_to_f32nchw", xt.data_ptr(), ld(xt), y.data_ptr(), bstride(y), B, C, H * W, 0, _stream())
    yv = Var(y, tape.enabled and x.req)
    if tape.enabled and x.req:

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            dx, acc = grad_buffer(x)
            if acc:
                tmp = _new(tuple(xt.shape), xt)
                _lib.call("cn_convert_f32nchw_to_bf16nhwc", dy.data_ptr(), bstride(dy), tmp.data_ptr(), ld(tmp), B, C, C,
                          H * W, _stream())
                _lib.call("cn_copy_bf16", tmp.data_ptr(), ld(tmp), dx.data_ptr(), ld(dx), _rows(tmp), C, 1, _stream())
            else:
                _lib.call("cn_convert_f32nchw_to_bf16nhwc", dy.data_ptr(), bstride(dy), dx.data_ptr(), ld(dx), B, C, C,
                          H * W, _stream())
            yv.grad = None

        tape.add(bwd)
    return yv


def _pow2(n: int) -> bool:
    return n >= 1 and (n & (n - 1)) == 0


_CONV_BNFIN = os.environ.get("CN_CONV_BNFIN", "1") != "0"


def _bnstats_launch(xts, pws, ys, B, Cin, H, W, Cout, KH, KW, stride, paddings, dilations, stats, bns):
    """A training-mode ConvBlock2d convolution (bias-free, G <= 4 branches) through
    cn_conv2d_fwd_grouped_bnstats_bf16: the launch writes the per-tile statistics rows AND, for launches of <= 1008 pixel
    tiles, finishes them (mean / rstd / running statistics) by a last-block ticket -- the dependent finalize launch between
    the convolution and the BatchNorm apply disappears. Returns per-output (bn, mean, rstd) or None when the launch only
    wrote the rows."""
    import ctypes

    G = len(xts)
    dev = xts[0].device
    tab = lambda ptrs: (ctypes.c_void_p * G)(*ptrs)
    means = _alloc((G, Cout), torch.float32, dev)
    rstds = _alloc((G, Cout), torch.float32, dev)
    has_running = bns[0].running_mean is not None
    need = int(_lib.query("cn_bn_group_workspace_floats_bf16", G, Cout))
    ws = _bn_group_ws16(G, Cout, dev)
    fin = ctypes.c_int(0)
    _lib.call("cn_conv2d_fwd_grouped_bnstats_bf16", G, tab([t.data_ptr() for t in xts]), ld(xts[0]),
              tab([p.fwd16.data_ptr() for p in pws]), tab([y.data_ptr() for y in ys]), ld(ys[0]), B, Cin, H, W, Cout, KH,
              KW, stride, (ctypes.c_int * G)(*paddings), (ctypes.c_int * G)(*dilations),
              tab([t.data_ptr() for t in stats]), tab([means[g].data_ptr() for g in range(G)]),
              tab([rstds[g].data_ptr() for g in range(G)]),
              tab([bn.running_mean.data_ptr() for bn in bns]) if has_running else None,
              tab([bn.running_var.data_ptr() for bn in bns]) if has_running else None, _bn_momentum(bns[0]),
              float(bns[0].eps), ws, max(need, 1 << 20), ctypes.byref(fin), _stream())
    if not fin.value:
        return None
    return [(bns[g], means[g], rstds[g]) for g in range(G)]


def _bnfin_ok(bns, mods) -> bool:
    return (_CONV_BNFIN and _GROUP_BF16 and bns is not None and len(bns) == len(mods)
            and all(m.bias is None for m in mods) and all(bn.training for bn in bns)
            and all(bn.momentum == bns[0].momentum and bn.eps == bns[0].eps for bn in bns)
            and all((bn.running_mean is None) == (bns[0].running_mean is None) for bn in bns)
            and mods[0].weight.shape[0] % 8 == 0)


def _conv2d_bf16(x: Var, mod, stride: int, padding: int, dilation: int, out: T.Optional[torch.Tensor],
                 want_stats: bool, bn=None) -> Var:
    tape = current_tape()
    xt = x.t
    B, Cin, H, W = xt.shape
    w = mod.weight
    Cout = w.shape[0]
    KH, KW = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
    Ho = (H + 2 * padding - dilation * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (KW - 1) - 1) // stride + 1
    pw = packed_conv(mod, tape.enabled and x.req, bf16=True)
    y = _check(out) if out is not None else _new((B, Cout, Ho, Wo), xt)
    bias = mod.bias
    stats = None
    if want_stats:  # one row of {sum, sumsq}[Cout] per pixel tile, written by the conv epilogue (no zero-fill)
        rows = _lib.query("cn_conv2d_stats_rows_bf16", B, H, W, Cout, KH, KW, stride, padding, dilation)
        stats = _alloc((rows, 2, Cout), torch.float32, xt.device)
    bnfin = None
    if stats is not None and w.dim() == 4 and _bnfin_ok([bn] if bn is not None else None, [mod]):
        bnfin = _bnstats_launch([xt], [pw], [y], B, Cin, H, W, Cout, KH, KW, stride, [padding], [dilation], [stats], [bn])
        bnfin = bnfin[0] if bnfin is not None else False  # False: launched (rows only)
    if bnfin is None:
        _lib.call("cn_conv2d_fwd_bf16", xt.data_ptr(), ld(xt), pw.fwd16.data_ptr(),
                  bias.data_ptr() if bias is not None else None, y.data_ptr(), ld(y), 0, B, Cin, H, W, Cout, KH, KW,
                  stride, padding, dilation, 0, 0, stats.data_ptr() if stats is not None else None, _stream())
    yv = Var(y, tape.enabled)
    yv.stats = stats
    yv.bnfin = bnfin or None
    if tape.enabled:
        store = current_store()

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            with side_stream(xt, dy, work=float(B) * Ho * Wo * Cin * Cout * KH * KW):
                s = _stream()
                need = _lib.query("cn_bwgrad_workspace_floats", B, Cin, H, W, Cout, KH, KW, stride, padding, dilation, 0)
                wsp, wsn = _ws16(need, xt.device)
                _lib.call("cn_conv2d_bwd_weight_bf16", xt.data_ptr(), ld(xt), dy.data_ptr(), ld(dy),
                          store.grad_of(w).data_ptr(), B, Cin, H, W, Cout, KH, KW, stride, padding, dilation, wsp, wsn,
                          s)
                if bias is not None:
                    _lib.call("cn_channel_sum_bf16", dy.data_ptr(), ld(dy), B * Ho * Wo, Cout,
                              store.grad_of(bias).data_ptr(), 1, _bn_ws16(Cout, xt.device, "side"), s)
            if x.req:
                dx, acc = grad_buffer(x)
                _lib.call("cn_conv2d_bwd_data_bf16", dy.data_ptr(), ld(dy), pw.bwd16.data_ptr(), dx.data_ptr(), ld(dx),
                          B, Cin, H, W, Cout, KH, KW, stride, padding, dilation, acc, _stream())
            yv.grad = None

        tape.add(bwd, (w, bias))
    return yv


_GROUP_BF16 = os.environ.get("CN_BF16_GROUPED", "1") == "1"  # diagnostic: 0 = branch-by-branch launches (round 3)


def _conv2d_group_bf16(xs: T.Sequence[Var], mods: T.Sequence, paddings: T.Sequence[int], dilations: T.Sequence[int],
                       stride: int, bns: T.Optional[T.Sequence] = None) -> T.List[Var]:
    """The G dilation branches of a ResidualAConv level as ONE bf16 implicit-GEMM launch (G classes of one launch:
    twice the blocks on the small planes, half the launches), every conv with its own BatchNorm statistics rows.
    Backward: weight gradients conv by conv on the side stream; data gradients in one grouped launch when the G inputs
    are distinct tensors (second level), conv by conv when they share their input (first level: the second launch
    accumulates -- a grouped launch would race on the shared dx)."""
    import ctypes

    tape = current_tape()
    G = len(mods)
    xts = [x.t for x in xs]
    B, Cin, H, W = xts[0].shape
    w0 = mods[0].weight
    same = all(tuple(t.shape) == (B, Cin, H, W) and ld(t) == ld(xts[0]) for t in xts) and \
        all(tuple(m.weight.shape) == tuple(w0.shape) and (m.bias is None) == (mods[0].bias is None) for m in mods)
    if not _GROUP_BF16 or G < 2 or G > 4 or not same or w0.dim() != 4:
        return [_conv2d_bf16(x, m, stride, p, d, None, tape.enabled, bns[i] if bns is not None else None)
                for i, (x, m, p, d) in enumerate(zip(xs, mods, paddings, dilations))]
    Cout, KH, KW = w0.shape[0], w0.shape[2], w0.shape[3]
    Ho = (H + 2 * paddings[0] - dilations[0] * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * paddings[0] - dilations[0] * (KW - 1) - 1) // stride + 1
    need_bwd = tape.enabled and any(x.req for x in xs)
    pws = [packed_conv(m, need_bwd, bf16=True) for m in mods]
    ys = [_new((B, Cout, Ho, Wo), xts[0]) for _ in range(G)]
    biases = [m.bias for m in mods]
    has_bias = biases[0] is not None
    tab = lambda ptrs: (ctypes.c_void_p * G)(*ptrs)
    pads_c = (ctypes.c_int * G)(*paddings)
    dils_c = (ctypes.c_int * G)(*dilations)
    stats = None
    if tape.enabled:  # (training forward: the tape is on; rows of {sum, sumsq}[Cout] per pixel tile and conv)
        rows = _lib.query("cn_conv2d_stats_rows_bf16", B, H, W, Cout, KH, KW, stride, max(paddings), max(dilations))
        stats = [_alloc((rows, 2, Cout), torch.float32, xts[0].device) for _ in range(G)]
    bnfin = None
    if stats is not None and _bnfin_ok(bns, mods):
        bnfin = _bnstats_launch(xts, pws, ys, B, Cin, H, W, Cout, KH, KW, stride, list(paddings), list(dilations), stats,
                                list(bns))
        if bnfin is None:
            bnfin = False  # launched, rows only
    if bnfin is None:
        _lib.call("cn_conv2d_fwd_grouped_bf16", G, tab([t.data_ptr() for t in xts]), ld(xts[0]),
                  tab([p.fwd16.data_ptr() for p in pws]), tab([b.data_ptr() for b in biases]) if has_bias else None,
                  tab([y.data_ptr() for y in ys]), ld(ys[0]), B, Cin, H, W, Cout, KH, KW, stride, pads_c, dils_c, 0,
                  tab([t.data_ptr() for t in stats]) if stats is not None else None, _stream())
    yvs = [Var(y, tape.enabled) for y in ys]
    for i, v in enumerate(yvs):
        v.stats = stats[i] if stats is not None else None
        v.bnfin = bnfin[i] if bnfin else None
    if tape.enabled:
        store = current_store()
        shared_in = any(xs[i] is xs[j] for i in range(G) for j in range(i))

        def bwd():
            live = [i for i in range(G) if yvs[i].grad is not None]
            if not live:
                return
            with side_stream(*(list(xts) + [yvs[i].grad for i in live]),
                             work=float(len(live)) * B * Ho * Wo * Cin * Cout * KH * KW):
                s = _stream()
                for i in live:
                    dy, m = yvs[i].grad, mods[i]
                    need = _lib.query("cn_bwgrad_workspace_floats", B, Cin, H, W, Cout, KH, KW, stride, paddings[i],
                                      dilations[i], 0)
                    wsp, wsn = _ws16(need, xts[i].device)
                    _lib.call("cn_conv2d_bwd_weight_bf16", xts[i].data_ptr(), ld(xts[i]), dy.data_ptr(), ld(dy),
                              store.grad_of(m.weight).data_ptr(), B, Cin, H, W, Cout, KH, KW, stride, paddings[i],
                              dilations[i], wsp, wsn, s)
                    if has_bias:
                        _lib.call("cn_channel_sum_bf16", dy.data_ptr(), ld(dy), B * Ho * Wo, Cout,
                                  store.grad_of(m.bias).data_ptr(), 1, _bn_ws16(Cout, xts[i].device, "side"), s)
            todo = [i for i in live if xs[i].req]
            bufs = [grad_buffer(xs[i]) for i in todo] if not shared_in else None
            if todo and not shared_in and len(todo) == G and len({a for _, a in bufs}) == 1 \
                    and len({ld(d) for d, _ in bufs}) == 1 and len({ld(yvs[i].grad) for i in todo}) == 1:
                _lib.call("cn_conv2d_bwd_data_grouped_bf16", G, tab([yvs[i].grad.data_ptr() for i in todo]),
                          ld(yvs[0].grad), tab([p.bwd16.data_ptr() for p in pws]), tab([d.data_ptr() for d, _ in bufs]),
                          ld(bufs[0][0]), B, Cin, H, W, Cout, KH, KW, stride, pads_c, dils_c, bufs[0][1], _stream())
            else:
                for n, i in enumerate(todo):
                    dx, acc = bufs[n] if bufs is not None else grad_buffer(xs[i])
                    dy = yvs[i].grad
                    _lib.call("cn_conv2d_bwd_data_bf16", dy.data_ptr(), ld(dy), pws[i].bwd16.data_ptr(), dx.data_ptr(),
                              ld(dx), B, Cin, H, W, Cout, KH, KW, stride, paddings[i], dilations[i], acc, _stream())
            for v in yvs:
                v.grad = None

        tape.add(bwd, tuple(m.weight for m in mods) + tuple(b for b in biases if b is not None))
    return yvs


class _Fold16:
    """Eval-mode BatchNorm folded into the bf16 packed weights of the convolution in front of it."""

    __slots__ = ("wp", "scale", "shift", "key")

    def __init__(self):
        self.wp = self.scale = self.shift = None
        self.key = None


_EVAL_FUSION = os.environ.get("CN_EVAL_FUSION", "1") == "1"


def eval_fusion(enabled: bool) -> bool:
    """Switch the fused inference ConvBlock2d on / off (tests compare both forms); returns the previous setting."""
    global _EVAL_FUSION
    prev, _EVAL_FUSION = _EVAL_FUSION, bool(enabled)
    return prev


def can_fuse_eval(x: Var, bn, training: bool) -> bool:
    """ConvBlock2d as ONE launch: inference mode, mixed-precision (bf16 NHWC) input, nothing recorded for backward,
    running statistics present and a channel count the 16-byte NHWC stores cover."""
    return (_EVAL_FUSION and not training and not current_tape().enabled and is16(x.t)
            and bn.running_mean is not None and bn.weight.shape[0] % 8 == 0)


def conv_bn_act_eval(x: Var, conv, bn, act: int, stride: int, padding: int, dilation: int,
                     residual: T.Optional[Var] = None) -> Var:
    """y = residual + act(BatchNorm_eval(conv(x))) in ONE launch (cn_conv2d_fwd_fused_bf16): the running statistics
    are folded into a scaled copy of the packed weights (W' = W * gamma / sigma per cout) and a bias
    (beta - mu * gamma / sigma); the activation and the ResUNet-a running sum ride in the conv epilogue. The folded
    copy is refreshed when the parameters change (ParamStore version), when a train-mode forward has updated running
    statistics anywhere (engine epoch), or when the buffers were written through torch (their version counters)."""
    xt = x.t
    B, Cin, H, W = xt.shape
    w = conv.weight
    Cout = w.shape[0]
    KH, KW = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
    taps = KH * KW
    Ho = (H + 2 * padding - dilation * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (KW - 1) - 1) // stride + 1
    store = current_store()
    fd = conv.__dict__.get("_cn_fold16")
    if fd is None:
        fd = conv.__dict__["_cn_fold16"] = _Fold16()
    key = (store.uid, store.version, _bn_stats_epoch, bn.running_mean._version, bn.running_var._version)
    s = _stream()
    if fd.key != key:
        if fd.wp is None or fd.key is None or fd.key[0] != key[0]:
            fd.wp = _alloc(_lib.query("cn_bconv_packed_elems", taps, Cin, Cout), torch.bfloat16, xt.device)
            fd.scale = _alloc(Cout, torch.float32, xt.device)
            fd.shift = _alloc(Cout, torch.float32, xt.device)
        cb = conv.bias
        _lib.call("cn_bn_fold_f32", bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                  bn.running_var.data_ptr(), cb.data_ptr() if cb is not None else None, float(bn.eps), Cout,
                  fd.scale.data_ptr(), fd.shift.data_ptr(), s)
        _lib.call("cn_pack_weights_scaled_bf16", w.data_ptr(), fd.scale.data_ptr(), fd.wp.data_ptr(), taps, Cin, Cout,
                  taps, Cin * taps, 1, s)
        fd.key = key
    y = _new((B, Cout, Ho, Wo), xt)
    rt = _check(residual.t) if residual is not None else None
    if rt is not None and tuple(rt.shape) != (B, Cout, Ho, Wo):
        raise ValueError("conv_bn_act_eval: the residual must have the output's shape")
    _lib.call("cn_conv2d_fwd_fused_bf16", xt.data_ptr(), ld(xt), fd.wp.data_ptr(), fd.shift.data_ptr(),
              rt.data_ptr() if rt is not None else None, ld(rt) if rt is not None else 0, y.data_ptr(), ld(y), B, Cin, H, W,
              Cout, KH, KW, stride, padding, dilation, 1 if act == ACT_SILU else 0, s)
    return Var(y, False)


def _conv_transpose2d_bf16(x: Var, mod, stride: int, padding: int) -> Var:
    tape = current_tape()
    xt = x.t
    B, Cin, H, W = xt.shape
    w = mod.weight
    Cout, KH, KW = w.shape[1], w.shape[2], w.shape[3]
    Ho = (H - 1) * stride - 2 * padding + KH
    Wo = (W - 1) * stride - 2 * padding + KW
    pw = packed_convT(mod, tape.enabled and x.req, bf16=True)
    y = _new((B, Cout, Ho, Wo), xt)
    bias = mod.bias
    _lib.call("cn_conv_transpose2d_fwd_bf16", xt.data_ptr(), ld(xt), pw.fwd16.data_ptr(),
              bias.data_ptr() if bias is not None else None, y.data_ptr(), ld(y), B, Cin, H, W, Cout, KH, KW, stride,
              padding, 0, _stream())
    yv = Var(y, tape.enabled)
    if tape.enabled:
        store = current_store()

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            with side_stream(xt, dy):
                s = _stream()
                need = _lib.query("cn_bwgrad_workspace_floats", B, Cin, H, W, Cout, KH, KW, stride, padding, 1, 1)
                wsp, wsn = _ws16(need, xt.device)
                _lib.call("cn_conv_transpose2d_bwd_weight_bf16", xt.data_ptr(), ld(xt), dy.data_ptr(), ld(dy),
                          store.grad_of(w).data_ptr(), B, Cin, H, W, Cout, KH, KW, stride, padding, wsp, wsn, s)
                if bias is not None:
                    _lib.call("cn_channel_sum_bf16", dy.data_ptr(), ld(dy), B * Ho * Wo, Cout,
                              store.grad_of(bias).data_ptr(), 1, _bn_ws16(Cout, xt.device, "side"), s)
            if x.req:
                dx, acc = grad_buffer(x)
                _lib.call("cn_conv_transpose2d_bwd_data_bf16", dy.data_ptr(), ld(dy), pw.bwd16.data_ptr(),
                          dx.data_ptr(), ld(dx), B, Cin, H, W, Cout, KH, KW, stride, padding, acc, _stream())
            yv.grad = None

        tape.add(bwd, (w, bias))
    return yv


def _bn_act_bf16(x: Var, bn, act: int, residual: T.Optional[Var], training: bool,
                 out: T.Optional[torch.Tensor]) -> Var:
    if _GROUP_BF16 and _dense16(x.t) and (out is None or _dense16(out)):
        # G = 1 of the grouped entry point: same arithmetic, the finalize is ONE coalesced, ticketed launch
        return _bn_act_group_bf16([x], [bn], act, residual, True, training, [out] if out is not None else None)
    tape = current_tape()
    xt = x.t
    B, C, H, W = xt.shape
    P = B * H * W
    dev = xt.device
    y = _check(out) if out is not None else _new(xt.shape, xt)
    mean = _alloc(C, torch.float32, dev)
    rstd = _alloc(C, torch.float32, dev)
    rt = _check(residual.t) if residual is not None else None
    use_batch = training or (bn.running_mean is None)
    _note_bn_update(training)
    sums = x.stats if use_batch else None
    _lib.call("cn_bn_act_fwd_bf16", xt.data_ptr(), ld(xt), bn.weight.data_ptr(), bn.bias.data_ptr(),
              bn.running_mean.data_ptr() if bn.running_mean is not None else None,
              bn.running_var.data_ptr() if bn.running_var is not None else None,
              rt.data_ptr() if rt is not None else None, ld(rt) if rt is not None else 0, y.data_ptr(), ld(y),
              mean.data_ptr(), rstd.data_ptr(), _bn_ws16(C, dev), P, C, 1 if use_batch else 0, _bn_momentum(bn),
              float(bn.eps), act, sums.data_ptr() if sums is not None else None,
              sums.shape[0] if sums is not None else 0, _stream())
    yv = Var(y, tape.enabled)
    if tape.enabled:
        store = current_store()
        gamma, beta = bn.weight, bn.bias

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            if residual is not None:
                give_grad(residual, dy)
            if x.req:
                dx, acc = grad_buffer(x)
                dxp, dxl = dx.data_ptr(), ld(dx)
            else:
                dxp, dxl, acc = None, 0, 0
            _lib.call("cn_bn_act_bwd_bf16", xt.data_ptr(), ld(xt), dy.data_ptr(), ld(dy), mean.data_ptr(),
                      rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dxp, dxl, store.grad_of(gamma).data_ptr(),
                      store.grad_of(beta).data_ptr(), _bn_ws16(C, dev), P, C, 1 if use_batch else 0, act, acc, _stream())
            yv.grad = None

        tape.add(bwd, (gamma, beta))
    return yv


_bng_ws: T.Dict[T.Tuple, torch.Tensor] = {}


def _bn_group_ws16(G: int, C: int, dev: torch.device) -> int:
    """Scratch of the grouped BatchNorm calls on the COMPUTE stream (launches are stream-ordered): ticket counters
    (zeroed ONCE here; every launch leaves them zero), finalize slices, backward coefficients, partial rows."""
    need = int(_lib.query("cn_bn_group_workspace_floats_bf16", G, C))
    key = (dev, _stream())
    ws = _bng_ws.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            torch.cuda.synchronize(dev)
        ws = _bng_ws[key] = _zeroed_scratch(max(need, 1 << 20), dev)
        _bump_ws_epoch()
    return ws.data_ptr()


def _bn_act_group_bf16(xs: T.Sequence[Var], bns: T.Sequence, act: int, residual: T.Optional[Var], sum_outputs: bool,
                       training: bool, outs: T.Optional[T.Sequence[torch.Tensor]]) -> T.Union[Var, T.List[Var]]:
    import ctypes

    tape = current_tape()
    G = len(xs)
    xts = [x.t for x in xs]
    B, C, H, W = xts[0].shape
    P = B * H * W
    dev = xts[0].device
    ok = 1 <= G <= 4 and all(tuple(t.shape) == (B, C, H, W) and ld(t) == ld(xts[0]) and _dense16(t) for t in xts) \
        and all(bn.momentum == bns[0].momentum and bn.eps == bns[0].eps for bn in bns) \
        and all((bn.running_mean is None) == (bns[0].running_mean is None) for bn in bns)
    use_batch = training or (bns[0].running_mean is None)
    has_sums = all(x.stats is not None for x in xs) and len({x.stats.shape[0] for x in xs}) == 1
    if not _GROUP_BF16 or not ok:
        if not sum_outputs:
            return [_bn_act_bf16(x, bn, act, None, training, outs[i] if outs is not None else None)
                    for i, (x, bn) in enumerate(zip(xs, bns))]
        acc = residual
        for x, bn in zip(xs, bns):
            acc = _bn_act_bf16(x, bn, act, acc, training, None)
        return acc
    _note_bn_update(training)
    tab = lambda ptrs: (ctypes.c_void_p * G)(*ptrs)
    if outs is not None:
        ys = [_check(o) for o in outs]
        if len(ys) != (1 if sum_outputs else G) or any(ld(y) != ld(ys[0]) for y in ys):
            raise ValueError("bn_act_group: outs must match the outputs and share their pixel stride")
    else:
        ys = [_new(xts[0].shape, xts[0]) for _ in range(1 if sum_outputs else G)]
    # statistics already finished by the convolution launch that produced xs (_bnstats_launch)?
    prefin = training and all(x.bnfin is not None and x.bnfin[0] is bn for x, bn in zip(xs, bns))
    if prefin:
        means = [x.bnfin[1] for x in xs]
        rstds = [x.bnfin[2] for x in xs]
    else:
        means = _alloc((G, C), torch.float32, dev)
        rstds = _alloc((G, C), torch.float32, dev)
    rt = _check(residual.t) if residual is not None else None
    has_running = bns[0].running_mean is not None
    sums = [x.stats for x in xs] if (use_batch and has_sums and not prefin) else None
    ws = _bn_group_ws16(G, C, dev)
    gam, bet = tab([bn.weight.data_ptr() for bn in bns]), tab([bn.bias.data_ptr() for bn in bns])
    mtab, rtab = tab([means[g].data_ptr() for g in range(G)]), tab([rstds[g].data_ptr() for g in range(G)])
    _lib.call("cn_bn_act_group_fwd_bf16", G, tab([t.data_ptr() for t in xts]), ld(xts[0]), gam, bet,
              tab([bn.running_mean.data_ptr() for bn in bns]) if has_running else None,
              tab([bn.running_var.data_ptr() for bn in bns]) if has_running else None,
              rt.data_ptr() if rt is not None else None, ld(rt) if rt is not None else 0,
              tab([(ys[0] if sum_outputs else ys[g]).data_ptr() for g in range(G)]), ld(ys[0]), mtab, rtab, ws, P, C,
              1 if use_batch else 0, _bn_momentum(bns[0]), float(bns[0].eps), act, 1 if sum_outputs else 0,
              tab([t.data_ptr() for t in sums]) if sums is not None else None,
              -1 if prefin else (sums[0].shape[0] if sums is not None else 0), _stream())
    yvs = [Var(y, tape.enabled) for y in ys]
    if tape.enabled:
        store = current_store()

        def bwd():
            if sum_outputs:
                dy = yvs[0].grad
                if dy is None:
                    return
                if residual is not None:
                    give_grad(residual, dy)
                dys = [dy] * G
            else:
                dys = [v.grad for v in yvs]
                if any(d is None for d in dys):
                    raise RuntimeError("bn_act_group: every output needs a gradient")
            bufs = [grad_buffer(x) if x.req else (None, 0) for x in xs]
            dxl = {ld(d) for d, _ in bufs if d is not None}
            dyl = {ld(d) for d in dys}
            if len(dxl) > 1 or len(dyl) > 1:
                raise RuntimeError("bn_act_group: gradient buffers must share their pixel strides")
            _lib.call("cn_bn_act_group_bwd_bf16", G, tab([t.data_ptr() for t in xts]), ld(xts[0]),
                      tab([d.data_ptr() for d in dys]), dyl.pop(), mtab, rtab, gam, bet,
                      tab([d.data_ptr() if d is not None else None for d, _ in bufs]), dxl.pop() if dxl else 0,
                      (ctypes.c_int * G)(*[a for _, a in bufs]),
                      tab([store.grad_of(bn.weight).data_ptr() for bn in bns]),
                      tab([store.grad_of(bn.bias).data_ptr() for bn in bns]), _bn_group_ws16(G, C, dev), P, C,
                      1 if use_batch else 0, act, _stream())
            _keep = (means, rstds)  # noqa: F841
            for v in yvs:
                v.grad = None

        tape.add(bwd, tuple(bn.weight for bn in bns) + tuple(bn.bias for bn in bns))
    return yvs[0] if sum_outputs else yvs


def _layer_norm_c_bf16(x: Var, ln, residual: T.Optional[Var], out: T.Optional[torch.Tensor] = None) -> Var:
    tape = current_tape()
    xt = x.t
    B, C, H, W = xt.shape
    if C % 8 or not _pow2(C >> 3) or (C >> 3) > 64:
        # the bf16 kernels reduce over C / 8 lanes with shuffles (8, 16, ... 512 channels): other widths through fp32
        r32 = to_f32(residual) if residual is not None else None
        return to_bf16(layer_norm_c(to_f32(x), ln, residual=r32))
    P = B * H * W
    y = out if _out_ok(out, xt) else _new(xt.shape, xt)
    rt = _check(residual.t) if residual is not None else None
    _lib.call("cn_layernorm_c_fwd_bf16", xt.data_ptr(), ld(xt), ln.weight.data_ptr(), ln.bias.data_ptr(),
              rt.data_ptr() if rt is not None else None, ld(rt) if rt is not None else 0, y.data_ptr(), ld(y), P, C,
              float(ln.eps), _stream())
    yv = Var(y, tape.enabled)
    if tape.enabled:
        store = current_store()

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            if residual is not None:
                give_grad(residual, dy)
            dx, acc = grad_buffer(x)
            _lib.call("cn_layernorm_c_bwd_bf16", xt.data_ptr(), ld(xt), dy.data_ptr(), ld(dy), ln.weight.data_ptr(),
                      dx.data_ptr(), ld(dx), store.grad_of(ln.weight).data_ptr(), store.grad_of(ln.bias).data_ptr(), P, C,
                      float(ln.eps), acc, _stream())
            yv.grad = None

        tape.add(bwd, (ln.weight, ln.bias))
    return yv


def _na2d_bf16(qkv: Var, heads: int, kernel_size: int, dilation: int, attn_drop: float = 0.0) -> Var:
    tape = current_tape()
    qt = qkv.t
    B, C3, H, W = qt.shape
    C = C3 // 3
    if C % heads or (C // heads) not in (4, 8, 16, 32, 64):
        # head dimensions the bf16 kernels are not compiled for (hidden 24 / 40 / 48 / 96 ...): the fp32 kernels take any
        return to_bf16(na2d(to_f32(qkv), heads, kernel_size, dilation, attn_drop))
    seed = _next_seed() if attn_drop > 0.0 else 0
    stepw = _step_word(qkv.t.device).data_ptr() if attn_drop > 0.0 else None
    step_fwd = _rng["step"]
    out = _new((B, C, H, W), qt)
    attn = _alloc((B, heads, kernel_size * kernel_size, H, W), torch.float32, qt.device)
    _lib.call("cn_na2d_fwd_bf16", qt.data_ptr(), ld(qt), out.data_ptr(), ld(out), attn.data_ptr(), B, C, heads, H, W,
              kernel_size, dilation, float(attn_drop), seed, stepw, _stream())
    ov = Var(out, tape.enabled)
    if tape.enabled:

        def bwd():
            do = ov.grad
            if do is None:
                return
            dattn = _alloc_like(attn)
            dq = _new(qt.shape, qt)
            _lib.call("cn_na2d_bwd_bf16", qt.data_ptr(), ld(qt), do.data_ptr(), ld(do), attn.data_ptr(),
                      dattn.data_ptr(), dq.data_ptr(), ld(dq), B, C, heads, H, W, kernel_size, dilation,
                      float(attn_drop), _bwd_seed(seed, step_fwd), stepw, _stream())
            if qkv.grad is None and qkv.parent is None:
                qkv.grad = dq
            else:  # pragma: no cover - qkv has a single consumer in TowerUNet
                give_grad(qkv, dq)
            ov.grad = None

        tape.add(bwd)
    return ov


def _resize_bilinear_bf16(x: Var, size: T.Tuple[int, int], out: T.Optional[torch.Tensor]) -> Var:
    tape = current_tape()
    xt = x.t
    B, C, Hi, Wi = xt.shape
    Ho, Wo = size
    y = _check(out) if out is not None else _new((B, C, Ho, Wo), xt)
    if (Hi, Wi) == (Ho, Wo):
        _lib.call("cn_copy_bf16", xt.data_ptr(), ld(xt), y.data_ptr(), ld(y), B * Ho * Wo, C, 0, _stream())
    else:
        _lib.call("cn_bilinear_fwd_bf16", xt.data_ptr(), ld(xt), y.data_ptr(), ld(y), B, C, Hi, Wi, Ho, Wo, _stream())
    yv = Var(y, tape.enabled and x.req)
    if tape.enabled and x.req:

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            if (Hi, Wi) == (Ho, Wo):
                give_grad(x, dy)
            else:
                dx, acc = grad_buffer(x)
                _lib.call("cn_bilinear_bwd_bf16", dy.data_ptr(), ld(dy), dx.data_ptr(), ld(dx), B, C, Hi, Wi, Ho, Wo, acc,
                          _stream())
            yv.grad = None

        tape.add(bwd)
    return yv


class _ThinPack16:
    """Concatenated copy [n*CP][Cin][3][3] of the n thin head weights + its bf16 fragment packs (refreshed when the
    parameters change): the three head streams of a tower run as ONE 128 -> 9 convolution, one 9 -> 128 backward-data
    and one weight-gradient launch instead of three each."""

    __slots__ = ("wcat", "dwcat", "fwd16", "bwd16", "version", "store_id", "view")

    def __init__(self):
        self.wcat = self.dwcat = self.fwd16 = self.bwd16 = None
        self.version = -1
        self.store_id = 0
        self.view = False  # wcat is a VIEW of the store's flat buffer (the n weights are adjacent there)


def _thin_conv3x3_bf16(x: Var, mods: T.Sequence, grouped: bool, dilation: int, out: T.Optional[torch.Tensor]) -> Var:
    """The first head convolutions (128 -> 3, three streams) on a bf16 tower output: the MFMA kernel with an fp32
    NCHW epilogue, so everything downstream of it (9 / 3 / 1-channel head tensors) stays on the fp32 head kernels."""
    if grouped:
        raise NotImplementedError("grouped thin convolutions read the fp32 head tensors, never bf16")
    if any(m.bias is not None for m in mods):
        raise NotImplementedError("thin head convolutions on the bf16 path are bias-free (ConvBlock2d)")
    tape = current_tape()
    xt = x.t
    B, Cin, H, W = xt.shape
    n = len(mods)
    CP = mods[0].weight.shape[0]
    CPt = n * CP
    y = out if out is not None else _alloc((B, CPt, H, W), torch.float32, xt.device)
    if y.dtype != torch.float32 or not y.is_contiguous():
        raise RuntimeError("thin_conv3x3 (bf16 input) writes a dense fp32 NCHW tensor")
    HW = H * W
    store = current_store()
    store.refresh()
    tw = mods[0].__dict__.get("_cn_thin16")
    if tw is None or tw.store_id != store.uid:
        tw = _ThinPack16()
        tw.store_id = store.uid
        per = CP * Cin * 9
        w0 = mods[0].weight
        # the store keeps declared groups adjacent (ParamStore._with_contiguous_groups): the n weights ARE one
        # [n*CP][Cin][3][3] tensor, and so are their gradients -- no concatenated copies, no per-step copy / pack / zero /
        # copy-back launches (27 per step for the three towers)
        tw.view = all(m.weight.is_contiguous() and m.weight.data_ptr() == w0.data_ptr() + 4 * per * i
                      for i, m in enumerate(mods))
        tw.fwd16 = _alloc(_lib.query("cn_bconv_packed_elems", 9, Cin, CPt), torch.bfloat16, xt.device)
        tw.bwd16 = _alloc(_lib.query("cn_bconv_packed_elems", 9, CPt, Cin), torch.bfloat16, xt.device)
        if tw.view:
            o = (w0.data_ptr() - store._base) // 4
            tw.wcat = store.flat[o:o + n * per].view(CPt, Cin, 3, 3)
            tw.dwcat = None  # the flat gradient slice of the moment (the bridge swaps flat_grad): looked up at use
            s = _stream()
            _lib.call("cn_pack_weights_bf16", tw.wcat.data_ptr(), tw.fwd16.data_ptr(), 9, Cin, CPt, 9, Cin * 9, 1, s)
            _lib.call("cn_pack_weights_bf16", tw.wcat.data_ptr(), tw.bwd16.data_ptr(), 9, CPt, Cin, Cin * 9, 9, 1, s)
            store.register_pack16(tw, "fwd16", tw.fwd16, tw.wcat, 9, Cin, CPt, 9, Cin * 9, 1)
            store.register_pack16(tw, "bwd16", tw.bwd16, tw.wcat, 9, CPt, Cin, Cin * 9, 9, 1)
            tw.version = store.version
        else:
            tw.wcat = _alloc((CPt, Cin, 3, 3), torch.float32, xt.device)
            tw.dwcat = _alloc_like(tw.wcat)
        mods[0].__dict__["_cn_thin16"] = tw
    if tw.version != store.version:
        if tw.view:  # registered with the batched per-step repack: one launch for every layer of the model
            store.repack_all()
        else:
            s = _stream()
            per = CP * Cin * 9
            for i, m in enumerate(mods):
                _lib.call("cn_copy_f32", m.weight.data_ptr(), per, tw.wcat[i * CP].data_ptr(), per, 1, per, 0, s)
            _lib.call("cn_pack_weights_bf16", tw.wcat.data_ptr(), tw.fwd16.data_ptr(), 9, Cin, CPt, 9, Cin * 9, 1, s)
            _lib.call("cn_pack_weights_bf16", tw.wcat.data_ptr(), tw.bwd16.data_ptr(), 9, CPt, Cin, Cin * 9, 9, 1, s)
        tw.version = store.version
    _lib.call("cn_conv2d_fwd_bf16", xt.data_ptr(), ld(xt), tw.fwd16.data_ptr(), None, y.data_ptr(), 0, CPt * HW, B, Cin,
              H, W, CPt, 3, 3, 1, dilation, dilation, 0, 1, None, _stream())
    yv = Var(y, tape.enabled)
    if tape.enabled:

        def bwd():
            dy = yv.grad
            if dy is None:
                return
            s = _stream()
            cp8 = (CPt + 7) // 8 * 8
            d16 = _alloc((B, H, W, cp8), torch.bfloat16, xt.device)
            _lib.call("cn_convert_f32nchw_to_bf16nhwc", dy.data_ptr(), bstride(dy), d16.data_ptr(), cp8, B, CPt, cp8, HW, s)
            with side_stream(xt, d16):
                ss = _stream()
                need = _lib.query("cn_bwgrad_workspace_floats", B, Cin, H, W, CPt, 3, 3, 1, dilation, dilation, 0)
                wsp, wsn = _ws16(need, xt.device)
                if tw.view:  # accumulates straight into the flat gradient (zeroed once per step)
                    _lib.call("cn_conv2d_bwd_weight_bf16", xt.data_ptr(), ld(xt), d16.data_ptr(), cp8,
                              store.grad_of(tw.wcat).data_ptr(), B, Cin, H, W, CPt, 3, 3, 1, dilation, dilation, wsp,
                              wsn, ss)
                else:
                    _lib.call("cn_fill_f32", tw.dwcat.data_ptr(), tw.dwcat.numel(), 0.0, ss)
                    _lib.call("cn_conv2d_bwd_weight_bf16", xt.data_ptr(), ld(xt), d16.data_ptr(), cp8,
                              tw.dwcat.data_ptr(), B, Cin, H, W, CPt, 3, 3, 1, dilation, dilation, wsp, wsn, ss)
                    per = CP * Cin * 9
                    for i, m in enumerate(mods):
                        _lib.call("cn_copy_f32", tw.dwcat[i * CP].data_ptr(), per, store.grad_of(m.weight).data_ptr(),
                                  per, 1, per, 1, ss)
            if x.req:
                dx, acc = grad_buffer(x)
                _lib.call("cn_conv2d_bwd_data_bf16", d16.data_ptr(), cp8, tw.bwd16.data_ptr(), dx.data_ptr(), ld(dx), B,
                          Cin, H, W, CPt, 3, 3, 1, dilation, dilation, acc, s)
            yv.grad = None

        tape.add(bwd, tuple(m.weight for m in mods))
    return yv
