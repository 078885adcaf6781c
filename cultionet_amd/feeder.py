"""Host -> device input feeding for the training step.

The reference feeds the step through ``DataLoader(..., pin_memory=..., collate_fn=collate_fn)``
(/root/reference/src/cultionet/data/modules.py:44-56, data/utils.py:55-68) and scales / clips / z-scores every sample
on the CPU in ``EdgeDataset.get`` (data/datasets.py:443-446, utils/normalize.py:63-82). At GPU step times of ~20 ms the
host arithmetic and a synchronous copy would be what limits data-parallel scaling, so here:

  * collated batches stay RAW (int16 / uint16 / int32 reflectances, as stored) in pinned host memory;
  * ``DeviceFeeder`` copies batch i+1 to the device on a COPY STREAM while the step of batch i runs, and does the
    reference's per-sample arithmetic there as ONE pass over the batch (``cn_prepare_chips_f32``);
  * the compute stream only waits for the copy stream's event (no host synchronisation), and the staging buffers are
    handed to the compute stream with ``record_stream`` so the caching allocator cannot recycle them early.

Float batches (already prepared upstream) are copied and passed through untouched.
"""
from __future__ import annotations

import typing as T

import torch

from .data import Data
from .edges import prepare_chips


def pin_batch(batch: Data) -> Data:
    """A copy of a host batch whose tensors live in pinned memory (what DataLoader(pin_memory=True) does)."""
    out = {}
    for k, v in batch.to_dict().items():
        out[k] = v.pin_memory() if isinstance(v, torch.Tensor) and not v.is_cuda else v
    return Data(**out)


class DeviceFeeder:
    """``for batch in feeder.iterate(host_batches): step(batch)`` -- double-buffered host -> HBM feeding."""

    def __init__(self, device: T.Union[str, torch.device], mean: T.Optional[torch.Tensor] = None,
                 std: T.Optional[torch.Tensor] = None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceFeeder feeds a GPU (there is no CPU training path)")
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.mean, self.std = mean, std

    def _stage(self, host: Data) -> T.Tuple[Data, "torch.cuda.Event"]:
        """Enqueue copy + prologue of one batch on the copy stream; returns (device batch, ready event)."""
        dev = self.device
        with torch.cuda.stream(self.copy_stream):
            kw = {}
            for k, v in host.__dict__.items():
                kw[k] = v.to(dev, non_blocking=True) if isinstance(v, torch.Tensor) else v
            x = kw["x"]
            if x.dtype != torch.float32:  # raw reflectances: x/10000 -> clip -> z-score, one pass on the device
                kw["x"] = prepare_chips(x, self.mean, self.std)
            bd = kw.get("bdist")
            if bd is not None and bd.dtype != torch.float32:
                B = bd.shape[0]
                kw["bdist"] = prepare_chips(bd.reshape(B, 1, 1, *bd.shape[1:])).reshape(bd.shape)
            y = kw.get("y")
            if y is not None and y.dtype != torch.int64:
                kw["y"] = y.long()
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return Data(**kw), ev

    def iterate(self, host_batches: T.Iterable[Data]) -> T.Iterator[Data]:
        it = iter(host_batches)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            batch, ev = nxt
            try:
                nxt = self._stage(next(it))  # batch i+1 is in flight while the caller steps on batch i
            except StopIteration:
                nxt = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for v in batch.__dict__.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
            yield batch
