"""The fused PreTimeReduction kernel family (cn_pretime_*; reference models/nunet.py:18-105) through the C ABI against
the CPU oracle's PreTimeReduction module (plain torch fp32, pinned bit-exact to the imported reference by
tests/test_oracle_vs_reference.py): training forward (batch statistics, running-statistics update), inference forward,
and every parameter gradient of the backward. fp32 NCHW output: 2e-5 * scale (outputs), 1e-4 * scale (gradients), as the
other fp32 kernels; bf16 NHWC output: 6e-3 * scale."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ptrs(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts])


def _setup(C, T, Cout, seed=0):
    from oracle import towerunet_oracle as O

    torch.manual_seed(seed)
    ref = O.PreTimeReduction(C, T, Cout)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.weight.copy_(1 + 0.2 * torch.randn_like(m.weight))
                m.bias.copy_(0.2 * torch.randn_like(m.bias))
                m.running_mean.copy_(0.1 * torch.randn_like(m.running_mean))
                m.running_var.copy_(1 + 0.2 * torch.rand_like(m.running_var))
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.copy_(1 + 0.2 * torch.randn_like(m.weight))
                m.bias.copy_(0.2 * torch.randn_like(m.bias))
    return ref


def _plist(mod):
    out = []
    for b in (mod.conv3, mod.conv5):
        s = b.seq
        out += [s[0].weight, s[3].weight, s[1].weight, s[1].bias, s[1].running_mean, s[1].running_var, s[5].weight,
                s[5].bias, s[5].running_mean, s[5].running_var]
    ln = mod.layer_norm[1]
    return out + [ln.weight, ln.bias]


def _glist(mod):
    out = []
    for b in (mod.conv3, mod.conv5):
        s = b.seq
        out += [s[0].weight, s[3].weight, s[1].weight, s[1].bias, s[5].weight, s[5].bias]
    ln = mod.layer_norm[1]
    return out + [ln.weight, ln.bias]


def _close(a, b, rel, what, abs_=1e-7):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = float(b.abs().max())
    err = float((a - b).abs().max())
    assert err <= rel * scale + abs_, f"{what}: max err {err:.3e} > {rel * scale + abs_:.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("B,C,T,H,W,Cout,kind", [
    (2, 3, 12, 28, 28, 8, 0), (2, 3, 12, 20, 23, 32, 0), (1, 4, 25, 19, 17, 16, 0), (3, 3, 12, 50, 50, 32, 1),
    (2, 3, 12, 33, 31, 64, 0), (2, 1, 7, 16, 16, 8, 1), (8, 3, 12, 100, 100, 32, 0), (1, 4, 25, 40, 40, 32, 0)])
def test_pretime_train_fwd_bwd(B, C, T, H, W, Cout, kind):
    from cultionet_amd import _lib

    dev = torch.device("cuda:0")
    ref = _setup(C, T, Cout).train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, T, H, W, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    if kind == 1:
        dy = dy.to(torch.bfloat16).float()
    rm0 = [p.clone() for p in _plist(ref)]
    y = ref(x)
    y.backward(dy)
    HW = H * W
    need = _lib.query("cn_pretime_workspace_floats", B, C, T, HW, Cout, 1)
    if need < 0:
        pytest.skip("outside the fused kernel (LDS image of the gradient pass): the engine keeps its generic path")
    ws = torch.zeros(need, device=dev)
    xd = x.to(dev).view(B, C * T, H, W)
    pd = [p.detach().clone().to(dev) for p in rm0]  # parameters and the ORIGINAL running statistics
    stats_t = torch.empty(2 * (2 * C + 2 * Cout), device=dev)
    offs, o = [], 0
    for _ in range(2):
        for n in (C, C, Cout, Cout):
            offs.append(o)
            o += n
    stats = (ctypes.c_void_p * 8)(*[stats_t[i:].data_ptr() for i in offs])
    bn = (ctypes.c_float * 4)(1e-5, 0.1, 1e-5, 0.1)
    if kind == 0:
        yd = torch.full((B, Cout, H, W), float("nan"), device=dev)
        ystride = Cout * HW
    else:
        yd = torch.full((B, H, W, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
        ystride = Cout
    s = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        if rep == 1:  # a second call on the same workspace (tickets back to zero), running statistics reset
            for t, t0 in zip(pd, rm0):
                t.copy_(t0)
        _lib.call("cn_pretime_fwd_f32", xd.data_ptr(), C * T * HW, _ptrs(pd), stats, yd.data_ptr(), ystride, kind, B, C,
                  T, HW, Cout, 1, bn, 1e-5, ws.data_ptr(), ws.numel(), s)
        got = yd if kind == 0 else yd.permute(0, 3, 1, 2).float()
        _close(got, y, 2e-5 if kind == 0 else 6e-3, "y", abs_=1e-6)
    for t, r_ in zip(pd, _plist(ref)):  # running statistics after ONE training forward
        _close(t, r_, 1e-5, "params / running statistics", abs_=1e-6)
    assert int(ws[:64].view(torch.int32).abs().sum()) == 0
    gd = [torch.zeros_like(p, device=dev) for p in _glist(ref)]
    dyd = dy.to(dev) if kind == 0 else dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    _lib.call("cn_pretime_bwd_f32", xd.data_ptr(), C * T * HW, _ptrs(pd), stats, dyd.data_ptr(), ystride, kind,
              _ptrs(gd), B, C, T, HW, Cout, 1, bn, 1e-5, ws.data_ptr(), ws.numel(), s)
    names = ["wa3", "wb3", "g3_3", "b3_3", "g2_3", "b2_3", "wa5", "wb5", "g3_5", "b3_5", "g2_5", "b2_5", "gL", "bL"]
    for n, got, p in zip(names, gd, _glist(ref)):
        # (the conv-a weight gradients of a BatchNorm'd stack are sums that cancel to ~1e-6 of their terms: absolute floor)
        _close(got, p.grad, 2e-4, n, abs_=2e-5 * float(dy.abs().max()) * (B * HW) ** 0.5)
    assert int(ws[:64].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("B,C,T,H,W,Cout,kind", [(2, 3, 12, 28, 28, 8, 0), (1, 4, 25, 110, 110, 32, 1),
                                                 (4, 4, 25, 50, 47, 64, 0)])
def test_pretime_inference_forward(B, C, T, H, W, Cout, kind):
    from cultionet_amd import _lib

    dev = torch.device("cuda:0")
    ref = _setup(C, T, Cout, seed=3).eval()
    x = torch.randn(B, C, T, H, W, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        y = ref(x)
    HW = H * W
    need = _lib.query("cn_pretime_workspace_floats", B, C, T, HW, Cout, 0)
    ws = torch.zeros(need, device=dev)
    pd = [p.detach().clone().to(dev) for p in _plist(ref)]
    stats = (ctypes.c_void_p * 8)(*([None] * 8))
    bn = (ctypes.c_float * 4)(1e-5, 0.1, 1e-5, 0.1)
    xd = x.to(dev).view(B, C * T, H, W)
    if kind == 0:
        yd = torch.empty((B, Cout, H, W), device=dev)
        ystride = Cout * HW
    else:
        yd = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=dev)
        ystride = Cout
    _lib.call("cn_pretime_fwd_f32", xd.data_ptr(), C * T * HW, _ptrs(pd), stats, yd.data_ptr(), ystride, kind, B, C, T,
              HW, Cout, 0, bn, 1e-5, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    got = yd if kind == 0 else yd.permute(0, 3, 1, 2).float()
    _close(got, y, 2e-5 if kind == 0 else 6e-3, "y", abs_=1e-6)


def test_pretime_unsupported_shapes_keep_the_generic_path():
    from cultionet_amd import _lib

    assert _lib.query("cn_pretime_workspace_floats", 1, 3, 12, 100, 128, 1) == -1  # Cout > 64
    assert _lib.query("cn_pretime_workspace_floats", 1, 3, 4, 100, 32, 1) == -1    # T < 5
    assert _lib.query("cn_pretime_workspace_floats", 1, 9, 12, 100, 32, 1) == -1   # C > 8
    assert _lib.query("cn_pretime_workspace_floats", 1, 8, 60, 100, 64, 1) == -1   # LDS image beyond 160 KiB


def test_generic_kernel_still_serves_the_default_cubes():
    """The register variant takes (C, T) = (3, 12) and the (4, 25) inference pass by default; CN_PRETIME_REG=0 (read once
    per process, hence the child) sends the same shapes through the generic kernel, which every other cube still uses."""
    import os
    import subprocess
    import sys

    if os.environ.get("CN_PRETIME_REG") == "0":
        pytest.skip("already the generic-kernel child")
    env = dict(os.environ, CN_PRETIME_REG="0")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-k",
                        "3-12-28-28-8 or 3-12-50-50-32 or 8-3-12-100-100-32 or 4-25-110-110-32"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "passed" in r.stdout
