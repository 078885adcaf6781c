"""Deferred weight-gradient slice sums (csrc/cn_slicesum.h, engine.deferring_slice_sums): the ONE batched launch per
flush must give what the per-layer reductions gave, and it must really replace them.

Reference semantics: plain autograd accumulation of conv weight gradients
(/root/reference/src/cultionet/nn/modules/convolution.py:71-120); the oracle-level parity of the step itself is
tests/test_model_gpu.py (which runs with the deferral on, the default)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _table():
    host = torch.zeros(64 * 64, dtype=torch.uint8).pin_memory()
    dev = torch.zeros(64 * 64, dtype=torch.uint8, device="cuda:0")
    return host, dev


@pytest.mark.parametrize("shape", [(8, 128, 128, 100, 100), (8, 32, 32, 100, 100), (2, 128, 640, 26, 26)])
def test_fp32_weight_gradient_deferred_equals_immediate(shape):
    """cn_conv2d_bwd_weight_f32 with a sink registered: no reduction launch, slices left in the scratch; after
    cn_slice_sums_run dW is BITWISE what the immediate path produces (same threads, same order) whenever that path is
    deterministic (<= 128 slices; above it the immediate kernel uses 16-way float atomics)."""
    from cultionet_amd import _lib

    B, Cout, Cin, H, W = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    dy = torch.randn(B, Cout, H, W, generator=g).cuda()
    grads = torch.zeros(2, Cout * Cin * 9, device="cuda:0")
    ws = torch.empty(2, 16 << 20, device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    args = lambda k: (x.data_ptr(), Cin * H * W, dy.data_ptr(), Cout * H * W, grads[k].data_ptr(), B, Cin, H, W, Cout,
                      3, 3, 1, 1, 1, ws[k].data_ptr(), ws[k].numel(), s)
    n0 = _lib.query("cn_launch_count", 1)
    _lib.call("cn_conv2d_bwd_weight_f32", *args(0))
    imm = _lib.query("cn_launch_count", 1)
    host, dev = _table()
    _lib.call("cn_slice_sums_begin", host.data_ptr(), 64, grads.data_ptr(), grads.numel())
    try:
        _lib.call("cn_conv2d_bwd_weight_f32", *args(1))
        dfr = _lib.query("cn_launch_count", 1)
        n = _lib.query("cn_slice_sums_count")
        torch.cuda.synchronize()
        if n == 0:  # few splits: atomics straight into dW, nothing to defer
            assert dfr == imm
        else:
            assert dfr == imm - 1  # the reduction launch is gone ...
            assert float(grads[1].abs().max()) == 0.0  # ... and dW untouched until the sums run
            _lib.call("cn_slice_sums_run", host.data_ptr(), dev.data_ptr(), 0, n, 1, s)
    finally:
        _lib.call("cn_slice_sums_end")
    torch.cuda.synchronize()
    assert _lib.query("cn_slice_sums_count") == -1
    a, b = grads[0].cpu(), grads[1].cpu()
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    rec = np.frombuffer(bytes(host[:64].numpy()), dtype=np.int32)
    if n and rec[8] <= 128:  # nslices
        assert torch.equal(a, b)


def test_temporary_dw_is_not_deferred():
    """A dW outside the registered gradient buffer (the thin heads' concatenated temporary, the time convolution's
    expanded gradient) is consumed right after the call: the sink must refuse it and the call must reduce at once."""
    from cultionet_amd import _lib

    B, C, H, W = 8, 128, 100, 100
    x = torch.randn(B, C, H, W, device="cuda:0")
    dy = torch.randn(B, C, H, W, device="cuda:0")
    dw = torch.zeros(C * C * 9, device="cuda:0")
    other = torch.zeros(1024, device="cuda:0")
    ws = torch.empty(16 << 20, device="cuda:0")
    host, _ = _table()
    s = torch.cuda.current_stream().cuda_stream
    _lib.call("cn_slice_sums_begin", host.data_ptr(), 64, other.data_ptr(), other.numel())
    try:
        _lib.call("cn_conv2d_bwd_weight_f32", x.data_ptr(), C * H * W, dy.data_ptr(), C * H * W, dw.data_ptr(), B, C, H,
                  W, C, 3, 3, 1, 1, 1, ws.data_ptr(), ws.numel(), s)
        assert _lib.query("cn_slice_sums_count") == 0
    finally:
        _lib.call("cn_slice_sums_end")
    torch.cuda.synchronize()
    assert float(dw.abs().max()) > 0.0


@pytest.mark.parametrize("precision", ["32-true", "bf16-mixed"])
def test_step_gradients_with_and_without_deferral(precision, monkeypatch):
    """The whole training step on the same weights and batch, sums deferred (default) vs immediate: the flat gradient
    identical up to the float atomics of the few-split launches (1e-5 of the gradient's scale), and the reduction
    launches really gone."""
    from cultionet_amd import _lib
    from cultionet_amd import engine as E
    from cultionet_amd.data import Data
    from cultionet_amd.lightning import HipTrainer
    from oracle import towerunet_oracle as O
    from oracle.selfcheck import build_pair

    lit, _ = build_pair(hidden=32, device="cuda:0")
    lit.train()
    B = 8
    x, y, bdist = O.seeded_batch(B, seed=3, with_mask=True)
    batch = Data(x=x.cuda(), y=y.cuda(), bdist=bdist.cuda())
    trainer = HipTrainer(lit, precision=precision)
    out = {}
    for mode in (False, True, False, True):  # (twice each: the second pass of a mode reuses its arena / table)
        monkeypatch.setattr(E, "_DEFER_SUMS", mode)
        _lib.query("cn_launch_count", 1)
        loss = trainer.forward_backward(batch)
        torch.cuda.synchronize()
        out[mode] = (float(loss.item()), trainer.store.flat_grad.clone(), _lib.query("cn_launch_count", 1))
    (l0, g0, n0), (l1, g1, n1) = out[False], out[True]
    assert l0 == l1
    assert n1 <= n0 - 25, (n0, n1)  # >= 34 (fp32) / 68 (bf16) reductions became 1
    # the batched kernel sums every output with the threads and in the order of the immediate kernels, so the only
    # differences left are the float atomics of the step's few-split launches (present in both modes)
    print(precision, "bitwise identical:", bool(torch.equal(g0, g1)), "launches", n0, "->", n1)
    assert float((g0 - g1).abs().max()) <= 1e-5 * float(g0.abs().max())
