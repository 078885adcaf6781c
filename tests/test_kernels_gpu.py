"""Per-kernel parity: every HIP entry point (through the C ABI, via cultionet_amd.engine) against the
same op in plain PyTorch fp32 on the CPU, forward and backward, on seeded inputs.

Tolerances (fp32): outputs 2e-5 * scale where scale = max|ref| (the MFMA path is an fp32 fma chain
with a different summation order than oneDNN); gradients 1e-4 * scale.
"""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _close(a, b, tol, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol * scale:.3e} (scale {scale:.3e})"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _engine_run(mod, fn, inputs, dy, req=None):
    """Run fn(*Vars) under the tape with module `mod` on the GPU; returns y, input grads, param grads."""
    from cultionet_amd import engine as E

    dev = _dev()
    mod = mod.to(dev)
    store = E.ParamStore(mod)
    store.zero_grad()
    with E.using_store(store), E.recording(True) as tape:
        xs = [E.Var(t.to(dev).contiguous(), True if req is None else req[i]) for i, t in enumerate(inputs)]
        y = fn(*xs)
        y.grad = dy.to(dev).contiguous()
        tape.backward()
    torch.cuda.synchronize()
    pg = {n: store.grad_of(p).cpu() for n, p in mod.named_parameters()}
    return y.t.cpu(), [x.grad.cpu() if x.grad is not None else None for x in xs], pg


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, dil, bias
    (2, 16, 20, 20, 32, 3, 1, 1, 1, False),
    (2, 32, 25, 25, 64, 3, 1, 1, 1, False),
    (1, 72, 13, 13, 128, 3, 1, 1, 1, False),
    (2, 8, 28, 28, 16, 3, 2, 1, 1, False),    # pool conv (stride 2)
    (1, 16, 25, 25, 32, 3, 2, 1, 1, False),   # odd size stride 2 -> 13
    (2, 40, 14, 14, 128, 1, 1, 0, 1, True),   # 1x1 skip with bias
    (2, 128, 13, 13, 256, 1, 1, 0, 1, False),
    (2, 32, 28, 28, 3, 3, 1, 1, 1, False),    # head 128->3 style
    (2, 3, 28, 28, 1, 3, 1, 1, 1, True),      # head 3->1 with bias
    (2, 3, 28, 28, 3, 3, 1, 1, 1, False),     # fuse conv
    (2, 16, 28, 28, 16, 3, 1, 2, 2, False),   # true dilated conv
    (1, 130, 9, 11, 140, 3, 1, 1, 1, True),   # ragged channels, non-square
    (3, 24, 100, 100, 32, 3, 1, 1, 1, False), # BASELINE spatial size
    (4, 48, 100, 100, 256, 1, 1, 0, 1, True),   # large 1x1: the dedicated GEMM kernel (fwd and bwd-data)
    (8, 130, 50, 50, 200, 1, 1, 0, 1, False),   # same, ragged channels
    # the shapes that carry the step time at BASELINE configs[1] (batch 8, hidden 32)
    (8, 128, 100, 100, 128, 3, 1, 1, 1, False),  # up_au / tower_a block 1
    (8, 480, 100, 100, 128, 3, 1, 1, 1, False),  # tower_a block 0
    (8, 576, 50, 50, 128, 3, 1, 1, 1, False),    # tower_b block 0
    (8, 640, 25, 25, 128, 3, 1, 1, 1, False),    # tower_c block 0
    (8, 480, 100, 100, 128, 1, 1, 0, 1, True),   # tower_a skip
    # odd planes (H*W % 4 == 1) through the 16-byte staging kernel's odd-plane variant: per-channel alignment shifts,
    # the two partial pieces at a plane's ends, ragged channel chunks, stride 2 from an odd plane, dilation
    (8, 128, 25, 25, 128, 3, 1, 1, 1, False),    # up_cu / tower_c blocks
    (8, 256, 13, 13, 256, 3, 1, 1, 1, True),
    (2, 128, 99, 99, 128, 3, 2, 1, 1, False),    # pool conv from the 99x99 ConvTranspose output
    (3, 12, 49, 49, 40, 3, 2, 1, 1, True),
    (2, 20, 25, 25, 24, 3, 1, 2, 2, False),      # dilated, Cin % 8 != 0
    (1, 9, 5, 5, 7, 3, 1, 1, 1, True),           # plane smaller than one tile, one image
    (4, 64, 9, 9, 64, 1, 1, 0, 1, True),
]


def _random_conv_cases(n, seed):
    """Seeded random shapes across the kernels' paths: 16-byte (H*W % 4 == 0) and dword staging, ragged channel chunks
    (Cin % 8 != 0), ragged couts, stride 2, dilation, 1x1, planes smaller than a tile."""
    import random

    rng = random.Random(seed)
    out = []
    for _ in range(n):
        k = rng.choice([1, 3, 3, 3])
        d = rng.choice([1, 1, 2, 3]) if k == 3 else 1
        s = rng.choice([1, 1, 1, 2]) if d == 1 else 1
        p = d * (k // 2)
        cin = rng.choice([3, 8, 12, 30, 36, 64, 70, 128, 132])
        cout = rng.choice([1, 3, 8, 24, 32, 64, 100, 128, 136])
        h = rng.choice([7, 9, 12, 13, 25, 26, 28, 50])
        w = rng.choice([7, 10, 12, 13, 25, 26, 28, 50])
        if k == 3 and (h <= 2 * d or w <= 2 * d):
            h, w = h + 2 * d, w + 2 * d
        out.append((rng.choice([1, 2, 3]), cin, h, w, cout, k, s, p, d, rng.random() < 0.3))
    return out


@pytest.mark.parametrize("case", CONV_CASES + _random_conv_cases(14, seed=20261003))
def test_conv2d(case):
    from cultionet_amd import engine as E

    B, Cin, H, W, Cout, k, s, p, d, bias = case
    conv = nn.Conv2d(Cin, Cout, k, stride=s, padding=p, dilation=d, bias=bias)
    x = _rand(B, Cin, H, W, seed=1)
    xr = x.clone().requires_grad_(True)
    yr = conv(xr)
    dy = _rand(*yr.shape, seed=2)
    yr.backward(dy)
    ref_w, ref_b = conv.weight.grad.clone(), (conv.bias.grad.clone() if bias else None)
    y, (dx,), pg = _engine_run(conv, lambda v: E.conv2d(v, conv, s, p, d), [x], dy)
    _close(y, yr, 2e-5, "y")
    _close(dx, xr.grad, 1e-4, "dx")
    _close(pg["weight"], ref_w, 1e-4, "dw")
    if bias:
        _close(pg["bias"], ref_b, 1e-4, "db")


CONVT_CASES = [
    # B, Cin, H, W, Cout, stride
    (2, 16, 13, 13, 16, 2),
    (2, 32, 25, 25, 32, 2),
    (1, 24, 14, 14, 24, 2),
    (2, 16, 7, 7, 16, 4),     # final_c: stride 4 (some outputs receive only bias)
    (1, 128, 25, 25, 128, 2),
    (2, 40, 50, 50, 40, 2),
]


@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv_transpose2d(case):
    from cultionet_amd import engine as E

    B, Cin, H, W, Cout, s = case
    conv = nn.ConvTranspose2d(Cin, Cout, 3, stride=s, padding=1)
    x = _rand(B, Cin, H, W, seed=3)
    xr = x.clone().requires_grad_(True)
    yr = conv(xr)
    dy = _rand(*yr.shape, seed=4)
    yr.backward(dy)
    ref_w, ref_b = conv.weight.grad.clone(), conv.bias.grad.clone()
    y, (dx,), pg = _engine_run(conv, lambda v: E.conv_transpose2d(v, conv, s, 1), [x], dy)
    _close(y, yr, 2e-5, "y")
    _close(dx, xr.grad, 1e-4, "dx")
    _close(pg["weight"], ref_w, 1e-4, "dw")
    _close(pg["bias"], ref_b, 1e-4, "db")


@pytest.mark.parametrize("case", [
    # B, C, H, W, stride, size: the ConvTranspose2d + check_upsample sites of TowerUNet (convolution.py:45-68)
    (2, 16, 25, 25, 2, 50),    # 49 -> 50
    (2, 24, 50, 50, 2, 100),   # 99 -> 100 (the five 100 x 100 sites)
    (1, 16, 25, 25, 4, 100),   # final_c: stride 4, 97 -> 100
    (2, 16, 13, 13, 2, 25),    # 25 already: no resize, no output padding
    (2, 8, 14, 14, 2, 30),     # 27 -> 30: a gap of 3 >= stride -- the unpadded path + a plain resize
])
def test_conv_transpose2d_then_resize(case):
    """The reference's ConvTranspose2d module (up_conv, then check_upsample's align_corners bilinear resize) against
    torch: the engine computes the transposed convolution on the resize's own grid (output_padding, 16-byte aligned
    planes) and resizes from the image inside it; values and all gradients must not notice."""
    from cultionet_amd import engine as E
    from cultionet_amd.convolution import ConvTranspose2d

    B, C, H, W, s, size = case
    torch.manual_seed(11)
    mod = ConvTranspose2d(C, C, 3, s, 1)
    x = _rand(B, C, H, W, seed=31)
    xr = x.clone().requires_grad_(True)
    yr = mod.up_conv(xr)
    natural = yr.shape[-1]
    if natural != size:
        yr = F.interpolate(yr, size=(size, size), mode="bilinear", align_corners=True)
    dy = _rand(*yr.shape, seed=32)
    yr.backward(dy)
    ref_w, ref_b = mod.up_conv.weight.grad.clone(), mod.up_conv.bias.grad.clone()
    seen = {}

    def run(v):
        out = mod(v, size=(size, size))
        seen["stored"] = tuple(out.t.shape[-2:])
        return out

    y, (dx,), pg = _engine_run(mod, run, [x], dy)
    assert tuple(y.shape[-2:]) == (size, size)
    _close(y, yr, 2e-5, "y")
    _close(dx, xr.grad, 1e-4, "dx")
    _close(pg["up_conv.weight"], ref_w, 1e-4, "dw")
    _close(pg["up_conv.bias"], ref_b, 1e-4, "db")


@pytest.mark.parametrize("k", [3, 5])
def test_time_conv(k):
    """nn.Conv3d(kernel (k,1,1)) of PreTimeReduction as a banded 1x1 contraction."""
    from cultionet_amd import engine as E

    B, C, Tn, H, W, Cout = 2, 3, 12, 20, 20, 3
    conv = nn.Conv3d(C, Cout, (k, 1, 1), bias=False)
    x = _rand(B, C, Tn, H, W, seed=5)
    xr = x.clone().requires_grad_(True)
    yr = conv(xr)
    dy = _rand(*yr.shape, seed=6)
    yr.backward(dy)
    ref_w = conv.weight.grad.clone()
    y, (dx,), pg = _engine_run(conv, lambda v: E.time_conv(v, conv, Tn), [x.reshape(B, C * Tn, H, W)],
                               dy.reshape(B, -1, H, W))
    _close(y.reshape(yr.shape), yr, 2e-5, "y")
    _close(dx.reshape(x.shape), xr.grad, 1e-4, "dx")
    _close(pg["weight"], ref_w, 1e-4, "dw")


@pytest.mark.parametrize("shape,act,res,train", [
    ((4, 32, 25, 25), 1, True, True),
    ((2, 128, 13, 13), 1, False, True),
    ((3, 16, 50, 50), 0, False, True),
    ((2, 3, 28, 28), 1, True, True),
    ((2, 32, 25, 25), 1, True, False),
    ((2, 64, 100, 100), 1, True, True),
])
def test_bn_act(shape, act, res, train):
    from cultionet_amd import engine as E

    C = shape[1]
    bn = nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * _rand(C, seed=7))
        bn.bias.copy_(0.1 * _rand(C, seed=8))
        bn.running_mean.copy_(0.1 * _rand(C, seed=9))
        bn.running_var.copy_(torch.rand(C, generator=torch.Generator().manual_seed(10)) + 0.5)
    bn.train(train)
    x = _rand(*shape, seed=11) * 2 + 0.5
    r = _rand(*shape, seed=12)
    xr, rr = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    import copy

    bn_ref = copy.deepcopy(bn)
    z = bn_ref(xr)
    if act:
        z = F.silu(z)
    yr = z + rr if res else z
    dy = _rand(*shape, seed=13)
    yr.backward(dy)

    def fn(xv, rv):
        return E.bn_act(xv, bn, act, residual=rv if res else None, training=train)

    y, (dx, dr), pg = _engine_run(bn, fn, [x, r], dy)
    _close(y, yr, 2e-5, "y")
    _close(dx, xr.grad, 1e-4, "dx")
    if res:
        _close(dr, rr.grad, 1e-6, "dres")
    _close(pg["weight"], bn_ref.weight.grad, 1e-4, "dgamma")
    _close(pg["bias"], bn_ref.bias.grad, 1e-4, "dbeta")
    _close(bn.running_mean, bn_ref.running_mean, 1e-5, "running_mean")
    _close(bn.running_var, bn_ref.running_var, 1e-5, "running_var")


def test_bn3d_view():
    """BatchNorm3d over [B, C, T, H, W] run on the [B, C*T, H, W] view (channels=C)."""
    from cultionet_amd import engine as E

    B, C, Tn, H, W = 2, 3, 10, 20, 20
    bn = nn.BatchNorm3d(C)
    x = _rand(B, C, Tn, H, W, seed=14)
    xr = x.clone().requires_grad_(True)
    import copy

    bn_ref = copy.deepcopy(bn)
    yr = F.silu(bn_ref(xr))
    dy = _rand(*yr.shape, seed=15)
    yr.backward(dy)
    y, (dx,), pg = _engine_run(bn, lambda v: E.bn_act(v, bn, 1, channels=C), [x.reshape(B, C * Tn, H, W)],
                               dy.reshape(B, C * Tn, H, W))
    _close(y.reshape(yr.shape), yr, 2e-5, "y")
    _close(dx.reshape(x.shape), xr.grad, 1e-4, "dx")
    _close(pg["weight"], bn_ref.weight.grad, 1e-4, "dgamma")


@pytest.mark.parametrize("shape,res", [((2, 32, 20, 20), False), ((2, 128, 25, 25), True), ((1, 8, 28, 28), True)])
def test_layer_norm_c(shape, res):
    from cultionet_amd import engine as E

    C = shape[1]
    ln = nn.LayerNorm(C)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * _rand(C, seed=16))
        ln.bias.copy_(0.1 * _rand(C, seed=17))
    x, r = _rand(*shape, seed=18) * 3, _rand(*shape, seed=19)
    xr, rr = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    import copy

    ln_ref = copy.deepcopy(ln)
    yr = ln_ref(xr.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
    if res:
        yr = yr + rr
    dy = _rand(*shape, seed=20)
    yr.backward(dy)
    y, (dx, dr), pg = _engine_run(ln, lambda xv, rv: E.layer_norm_c(xv, ln, rv if res else None), [x, r], dy)
    _close(y, yr, 2e-5, "y")
    _close(dx, xr.grad, 1e-4, "dx")
    _close(pg["weight"], ln_ref.weight.grad, 1e-4, "dw")
    _close(pg["bias"], ln_ref.bias.grad, 1e-4, "db")


@pytest.mark.parametrize("B,C,heads,H,W,dil", [(2, 32, 4, 14, 14, 1), (1, 32, 4, 28, 28, 2), (2, 128, 8, 25, 25, 1),
                                               (1, 128, 4, 50, 50, 2), (1, 16, 8, 7, 9, 2),
                                               (8, 128, 4, 100, 100, 2),   # BASELINE configs[1]: tower a / up_au
                                               (8, 128, 4, 50, 50, 1), (8, 128, 8, 25, 25, 1)])
def test_na2d(B, C, heads, H, W, dil):
    from cultionet_amd import engine as E
    from oracle import na2d_ref as N

    D = C // heads
    qkv = _rand(B, 3 * C, H, W, seed=21)
    qr = qkv.clone().requires_grad_(True)
    t = qr.reshape(B, 3, heads, D, H, W).permute(1, 0, 2, 4, 5, 3)  # [3,B,h,H,W,D]
    q, k, v = t[0] * (D ** -0.5), t[1], t[2]
    o = N.na2d_av(N.na2d_qk(q, k, 3, dil).softmax(-1), v, 3, dil)  # [B,h,H,W,D]
    yr = o.permute(0, 1, 4, 2, 3).reshape(B, C, H, W)
    dy = _rand(B, C, H, W, seed=22)
    yr.backward(dy)
    y, (dq,), _ = _engine_run(nn.Linear(1, 1), lambda v_: E.na2d(v_, heads, 3, dil), [qkv], dy)
    _close(y, yr, 2e-5, "out")
    _close(dq, qr.grad, 1e-4, "dqkv")


@pytest.mark.parametrize("B,C,Hi,Wi,Ho,Wo", [(2, 8, 13, 13, 14, 14), (1, 16, 49, 49, 50, 50), (2, 4, 99, 99, 100, 100),
                                             (2, 4, 97, 97, 100, 100), (1, 3, 27, 25, 28, 28),
                                             (1, 20, 99, 99, 100, 100), (1, 8, 100, 100, 60, 60),
                                             (1, 8, 100, 100, 100, 100), (1, 40, 25, 25, 100, 100),
                                             (1, 32, 51, 51, 100, 100), (1, 8, 100, 100, 51, 51)])
def test_bilinear(B, C, Hi, Wi, Ho, Wo):
    from cultionet_amd import engine as E

    x = _rand(B, C, Hi, Wi, seed=23)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=True)
    dy = _rand(*yr.shape, seed=24)
    yr.backward(dy)
    y, (dx,), _ = _engine_run(nn.Linear(1, 1), lambda v: E.resize_bilinear(v, (Ho, Wo)), [x], dy)
    _close(y, yr, 1e-6, "y")
    _close(dx, xr.grad, 1e-5, "dx")


def test_cat_and_slices():
    from cultionet_amd import engine as E

    a, b, c = _rand(2, 5, 9, 9, seed=25), _rand(2, 7, 9, 9, seed=26), _rand(2, 3, 9, 9, seed=27)
    dy = _rand(2, 15, 9, 9, seed=28)
    y, grads, _ = _engine_run(nn.Linear(1, 1), lambda x, y_, z: E.cat_channels([x, y_, z]), [a, b, c], dy)
    _close(y, torch.cat([a, b, c], 1), 0, "cat")
    _close(grads[0], dy[:, :5], 0, "da")
    _close(grads[1], dy[:, 5:12], 0, "db")
    _close(grads[2], dy[:, 12:], 0, "dc")


def test_final_combine():
    from cultionet_amd import engine as E
    from oracle import towerunet_oracle as O

    B, H, W = 2, 20, 20
    fc = O.TowerUNetFinalCombine()
    with torch.no_grad():
        for i, p in enumerate(fc.parameters()):
            p.copy_(torch.rand(p.shape, generator=torch.Generator().manual_seed(30 + i)) * 0.5 + 0.75)
    hs = [_rand(B, 3, H, W, seed=40 + i) for i in range(3)]
    hr = [h.clone().requires_grad_(True) for h in hs]
    outs = fc(*[torch.chunk(h, 3, dim=1) for h in hr])
    dys = [_rand(B, 1, H, W, seed=50 + i) for i in range(3)]
    loss = sum((outs[k] * d).sum() for k, d in zip(("distance", "edge", "crop"), dys))
    loss.backward()

    from cultionet_amd.unet_parts import final_combine_params

    dev = _dev()
    fc_dev = fc.to(dev)
    store = E.ParamStore(fc_dev)
    store.zero_grad()
    with E.using_store(store), E.recording(True) as tape:
        vs = [E.Var(h.to(dev), True) for h in hs]
        o = E.final_combine(vs[0], vs[1], vs[2], final_combine_params(fc_dev), 1e-2)
        for v, d in zip(o, dys):
            v.grad = d.to(dev)
        tape.backward()
    torch.cuda.synchronize()
    for v, k in zip(o, ("distance", "edge", "crop")):
        _close(v.t, outs[k], 2e-6, k)
    for v, h in zip(vs, hr):
        _close(v.grad, h.grad, 1e-5, "dh")
    fc_cpu = O.TowerUNetFinalCombine()
    for (n, p), (_, pr) in zip(fc_dev.named_parameters(), fc.named_parameters()):
        pass
    # parameter grads (fc was moved in place: its .grad tensors hold the CPU reference values)
    ref = {n: p.grad for n, p in fc.named_parameters()}
    for n, p in fc_dev.named_parameters():
        _close(store.grad_of(p), ref[n] if ref[n] is not None else torch.zeros_like(p), 1e-4, n)


def _loss_inputs():
    import numpy as np

    rng = np.random.default_rng(100)
    B, H, W = 2, 20, 20
    rng.uniform(low=-3, high=3, size=(B, 2, H, W))
    crop_prob = torch.from_numpy(rng.dirichlet((0.5, 0.5), size=(B * H * W))).float()
    crop_prob = crop_prob.reshape(B, H, W, 2).permute(0, 3, 1, 2).contiguous()
    rng.random((B, 1, H, W))
    dist = torch.from_numpy(rng.random((B, 1, H, W))).float()
    targets = torch.from_numpy(rng.integers(low=0, high=2, size=(B, H, W))).long()
    rng.integers(low=0, high=1, size=(B, H, W))
    dist_t = torch.from_numpy(rng.random((B, H, W))).float()
    mask = torch.from_numpy(rng.integers(low=0, high=2, size=(B, 1, H, W))).long()
    return crop_prob, dist, targets, dist_t, mask


def test_tanimoto_known_answers_and_grads():
    """/root/reference/tests/test_loss.py:109-145 known answers (3 decimals) + gradients vs the oracle."""
    from cultionet_amd import engine as E
    from oracle import towerunet_oracle as O

    dev = _dev()
    crop_prob, dist, targets, dist_t, mask = _loss_inputs()
    cases = [
        ("TanimotoDistLoss", crop_prob, dict(labels=targets, target_mode=E.TGT_ONEHOT), None, 0.611, O.tanimoto_dist_loss, True),
        ("TanimotoDistLoss", crop_prob, dict(labels=targets, target_mode=E.TGT_ONEHOT), mask, 0.431, O.tanimoto_dist_loss, True),
        ("TanimotoComplementLoss", crop_prob, dict(labels=targets, target_mode=E.TGT_ONEHOT), None, 0.824, O.tanimoto_complement_loss, True),
        ("TanimotoComplementLoss", crop_prob, dict(labels=targets, target_mode=E.TGT_ONEHOT), mask, 0.692, O.tanimoto_complement_loss, True),
        ("TanimotoCombined", crop_prob, dict(labels=targets, target_mode=E.TGT_ONEHOT), None, 0.717, O.tanimoto_combined_loss, True),
        ("TanimotoCombined", crop_prob, dict(labels=targets, target_mode=E.TGT_ONEHOT), mask, 0.561, O.tanimoto_combined_loss, True),
        ("TanimotoDistLoss", dist, dict(target_f=dist_t, target_mode=E.TGT_FLOAT), None, 0.417, O.tanimoto_dist_loss, False),
        ("TanimotoComplementLoss", dist, dict(target_f=dist_t, target_mode=E.TGT_FLOAT), None, 0.704, O.tanimoto_complement_loss, False),
    ]
    for kind, pred, tk, mk, expect, ofn, onehot in cases:
        pr = pred.clone().requires_grad_(True)
        tgt = tk.get("labels") if "labels" in tk else tk["target_f"]
        lr = ofn(pr, tgt, mk, one_hot_targets=onehot)
        lr.backward()
        with E.recording(True) as tape:
            pv = E.Var(pred.to(dev), True)
            kw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in tk.items()}
            loss = E.tanimoto_loss(pv, mask=mk.to(dev) if mk is not None else None,
                                   mask_mode=E.MSK_I64 if mk is not None else E.MSK_NONE,
                                   loss_kind=E.LOSS_KINDS[kind], **kw)
            tape.backward()
        torch.cuda.synchronize()
        assert round(float(loss.item()), 3) == expect, (kind, float(loss.item()), expect)
        assert abs(float(loss.item()) - float(lr.item())) < 2e-6
        _close(pv.grad, pr.grad, 1e-5, kind + " grad")


def test_label_modes():
    """get_true_labels fused into the loss kernel (lightning.py:161-207)."""
    from cultionet_amd import engine as E
    from oracle import towerunet_oracle as O

    dev = _dev()
    g = torch.Generator().manual_seed(60)
    B, H, W = 3, 25, 25
    y = torch.randint(-1, 3, (B, H, W), generator=g)
    pred = torch.rand(B, 1, H, W, generator=g)
    te, tc, mask = O.true_labels(y, 2)
    for mode, tgt in ((E.TGT_EQ, te), (E.TGT_RANGE, tc)):
        pr = pred.clone().requires_grad_(True)
        lr = O.tanimoto_complement_loss(pr, tgt, mask)
        lr.backward()
        with E.recording(True) as tape:
            pv = E.Var(pred.to(dev), True)
            loss = E.tanimoto_loss(pv, labels=y.to(dev), target_mode=mode, mask_mode=E.MSK_LABEL, klass=2)
            tape.backward()
        assert abs(float(loss.item()) - float(lr.item())) < 2e-6
        _close(pv.grad, pr.grad, 1e-5, "grad")


def test_adamw_and_clip():
    from cultionet_amd import _lib
    from cultionet_amd import engine as E

    dev = _dev()
    n = 10007
    p0, g0 = _rand(n, seed=70), _rand(n, seed=71) * 3
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=0.01, weight_decay=1e-3, eps=1e-4, betas=(0.9, 0.98))
    p, m, v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
    for step in range(1, 4):
        g = g0 * step
        pr.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        gd = g.to(dev)
        _lib.call("cn_grad_sumsq_f32", gd.data_ptr(), n, sumsq.data_ptr(), E._stream())
        _lib.call("cn_adamw_step_f32", p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 0.01, 0.9, 0.98,
                  1e-4, 1e-3, step, 1.0, sumsq.data_ptr(), 1.0, E._stream())
        torch.cuda.synchronize()
        _close(p, pr, 1e-6, f"step {step}")


@pytest.mark.parametrize("shape,out", [((2, 8, 28, 28), (14, 14)), ((2, 16, 25, 25), (12, 12)), ((1, 4, 13, 9), (6, 4))])
def test_adaptive_max_pool(shape, out):
    from cultionet_amd import engine as E

    x = _rand(*shape, seed=80)
    xr = x.clone().requires_grad_(True)
    yr = F.adaptive_max_pool2d(xr, out)
    dy = _rand(*yr.shape, seed=81)
    yr.backward(dy)
    y, (dx,), _ = _engine_run(nn.Linear(1, 1), lambda v: E.adaptive_max_pool2d(v, out), [x], dy)
    _close(y, yr, 0, "y")
    _close(dx, xr.grad, 1e-6, "dx")


@pytest.mark.parametrize("channelwise", [True, False])
def test_dropout_statistics_and_backward(channelwise):
    """Masks cannot match torch's RNG; check the contract instead: kept values are x/(1-p), the kept fraction
    is ~1-p, the backward pass uses the same mask, and the mask is a pure function of the seed."""
    from cultionet_amd import engine as E

    dev = _dev()
    p = 0.3
    x = (_rand(8, 64, 20, 20, seed=82).abs() + 0.5).to(dev)
    dy = _rand(8, 64, 20, 20, seed=83).to(dev)
    outs = []
    for _ in range(2):
        E.manual_seed(1234)
        with E.recording(True) as tape:
            xv = E.Var(x, True)
            yv = E.dropout(xv, p, channelwise, training=True)
            yv.grad = dy.clone()
            tape.backward()
        torch.cuda.synchronize()
        outs.append((yv.t.cpu(), xv.grad.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    y, dx = outs[0]
    keep = y != 0
    assert torch.allclose(y[keep], (x.cpu() / (1 - p))[keep], rtol=1e-6)
    assert torch.allclose(dx, torch.where(keep, dy.cpu() / (1 - p), torch.zeros(())), rtol=1e-6)
    frac = keep.float().mean().item()
    assert abs(frac - (1 - p)) < (0.08 if channelwise else 0.01), frac
    if channelwise:  # whole (b, c) planes are kept or dropped together
        plane = keep.flatten(2)
        assert torch.equal(plane.all(-1), plane.any(-1))


def test_train_step_with_dropout_runs():
    """dropout=0.2 (the LitModel default): Dropout2d after encoder blocks + NA attn/proj dropout."""
    from cultionet_amd import engine as E
    from cultionet_amd.data import Data
    from cultionet_amd.lightning import CultionetLitModel, HipTrainer
    from cultionet_amd import synthetic as S

    dev = _dev()
    lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=8, dropout=0.2)
    m = lit.cultionet_model.mask_model
    m.load_state_dict(S.seeded_state_dict(m.state_dict()))
    lit = lit.to(dev).train()
    x, y, bdist = S.seeded_batch(2, height=28, width=28, with_mask=True)
    batch = Data(x=x.to(dev), y=y.to(dev), bdist=bdist.to(dev))
    tr = HipTrainer(lit)
    E.manual_seed(7)
    l1 = float(tr.forward_backward(batch).item())
    g1 = tr.store.flat_grad.clone()
    E.manual_seed(7)
    l2 = float(tr.forward_backward(batch).item())
    assert l1 == l2 and torch.isfinite(g1).all()
    assert 0.3 < l1 < 1.0
    lit.eval()
    with torch.no_grad():
        out = m(batch.x)
    assert torch.isfinite(out["crop"]).all()


@pytest.mark.parametrize("dtype", [torch.int16, torch.int32, torch.float32])
def test_prepare_chips(dtype):
    """datasets.py:443-446 + normalize.py:63-82 fused."""
    from cultionet_amd.edges import prepare_chips

    g = torch.Generator().manual_seed(90)
    raw = torch.randint(-50, 12000, (2, 4, 5, 17, 19), generator=g).to(dtype)
    mean = torch.rand(4, generator=g) * 0.3
    std = torch.rand(4, generator=g) * 0.2 + 0.05
    ref = (raw.float() / 10_000.0).clip(1e-9, 1)
    ref = (ref - mean.view(1, 4, 1, 1, 1)) / std.view(1, 4, 1, 1, 1)
    out = prepare_chips(raw.to(_dev()), mean, std).cpu()
    _close(out, ref, 1e-6, "prepared")


def test_predictions_to_uint16():
    """callbacks.py:176-227: slice padding, x10000, clip, uint16."""
    from cultionet_amd.edges import predictions_to_uint16

    g = torch.Generator().manual_seed(91)
    pred = {k: torch.rand(2, 1, 40, 44, generator=g) * 1.2 - 0.1 for k in ("distance", "edge", "crop")}
    pad, h, w = 4, 32, 36
    ref = torch.cat([pred[k][:, :, pad:pad + h, pad:pad + w] for k in ("distance", "edge", "crop")], 1)
    ref = (ref * 10_000.0).clip(0, 10_000.0).numpy().astype("uint16")
    out = predictions_to_uint16({k: v.to(_dev()) for k, v in pred.items()}, pad, h, w).cpu().numpy()
    assert (out.astype(int) - ref.astype(int)).__abs__().max() <= 1  # fp32 product rounding at integer boundaries
    assert (out == ref).mean() > 0.999


# ---------------------------------------------------------------------------
# thin (direct) 3x3 convolutions of the TowerUNetFinal head streams + grouped implicit-GEMM launches
# ---------------------------------------------------------------------------

class _Sets(nn.Module):
    def __init__(self, n, cin, cp, bias, dil=1):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(cin, cp, 3, padding=dil, dilation=dil, bias=bias) for _ in range(n)])


THIN_CASES = [
    # n sets, Cin per set, cout per set, grouped, bias, B, H, W, dil
    (3, 40, 3, False, False, 2, 28, 28, 1),   # 128 -> 3 x3 streams on one input
    (3, 3, 1, True, True, 2, 28, 28, 1),      # 3 -> 1 x3 streams on their own inputs (with bias)
    (1, 3, 3, False, False, 2, 28, 28, 1),    # fuse conv
    (1, 5, 1, False, True, 1, 9, 11, 1),      # single C -> 1 stream, ragged size
    (3, 128, 3, False, False, 2, 100, 100, 1),  # BASELINE head size
    (3, 7, 3, False, True, 3, 13, 17, 2),     # dilated
]


@pytest.mark.parametrize("case", THIN_CASES)
def test_thin_conv3x3(case):
    from cultionet_amd import engine as E

    n, cin, cp, grouped, bias, B, H, W, dil = case
    torch.manual_seed(5)
    mod = _Sets(n, cin, cp, bias, dil)
    x = _rand(B, n * cin if grouped else cin, H, W, seed=6)
    xr = x.clone().requires_grad_(True)
    if grouped:
        yr = torch.cat([c(xr[:, i * cin:(i + 1) * cin]) for i, c in enumerate(mod.convs)], dim=1)
    else:
        yr = torch.cat([c(xr) for c in mod.convs], dim=1)
    dy = _rand(*yr.shape, seed=7)
    yr.backward(dy)
    ref = {k: p.grad.clone() for k, p in mod.named_parameters()}
    y, (dx,), pg = _engine_run(mod, lambda v: E.thin_conv3x3(v, list(mod.convs), grouped, dil), [x], dy)
    _close(y, yr, 2e-5, "y")
    _close(dx, xr.grad, 1e-4, "dx")
    for k in ref:
        _close(pg[k], ref[k], 1e-4, k)


def test_thin_conv_unsupported_config_is_an_error():
    from cultionet_amd import engine as E

    mod = _Sets(2, 4, 2, False)
    x = _rand(1, 4, 8, 8)
    with pytest.raises(RuntimeError):
        _engine_run(mod, lambda v: E.thin_conv3x3(v, list(mod.convs), False), [x], _rand(1, 4, 8, 8))


def test_split_join_channels_plumbing():
    """conv -> split -> per-slice BN+SiLU written in place into one buffer -> join -> conv, against torch."""
    from cultionet_amd import engine as E

    torch.manual_seed(8)

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = nn.Conv2d(8, 9, 3, padding=1, bias=False)
            self.bns = nn.ModuleList([nn.BatchNorm2d(3) for _ in range(3)])
            self.c2 = nn.Conv2d(9, 4, 3, padding=1, bias=False)

    m = M()
    x = _rand(2, 8, 12, 12, seed=9)
    xr = x.clone().requires_grad_(True)
    h = m.c1(xr)
    a = torch.cat([F.silu(bn(h[:, 3 * i:3 * i + 3])) for i, bn in enumerate(m.bns)], dim=1)
    yr = m.c2(a)
    dy = _rand(*yr.shape, seed=10)
    yr.backward(dy)
    ref = {k: p.grad.clone() for k, p in m.named_parameters()}

    def fn(v):
        h9 = E.conv2d(v, m.c1, 1, 1, 1)
        buf = torch.empty_like(h9.t)
        acts = [E.bn_act(p, bn, E.ACT_SILU, out=buf[:, 3 * i:3 * i + 3])
                for i, (p, bn) in enumerate(zip(E.split_channels(h9, [3, 3, 3]), m.bns))]
        return E.conv2d(E.join_channels(acts, buf), m.c2, 1, 1, 1)

    y, (dx,), pg = _engine_run(m, fn, [x], dy)
    _close(y, yr, 2e-5, "y")
    _close(dx, xr.grad, 1e-4, "dx")
    for k in ref:
        _close(pg[k], ref[k], 1e-4, k)


GROUPED_CASES = [
    # G, B, Cin, H, W, Cout, dils, shared input, shared output (dgrad sums)
    (2, 2, 32, 28, 28, 32, (1, 2), True),
    (2, 2, 32, 25, 25, 48, (1, 3), False),
    (3, 1, 16, 13, 13, 16, (1, 1, 1), True),
    (2, 8, 128, 25, 25, 128, (1, 2), True),
]


@pytest.mark.parametrize("case", GROUPED_CASES)
def test_grouped_conv_launch(case):
    """cn_conv2d_fwd_grouped_f32 / cn_conv2d_bwd_data_grouped_f32 against per-branch torch convs."""
    import ctypes
    from cultionet_amd import engine as E, _lib

    G, B, Cin, H, W, Cout, dils, shared = case
    dev = _dev()
    torch.manual_seed(11)
    convs = nn.ModuleList([nn.Conv2d(Cin, Cout, 3, padding=d, dilation=d, bias=False) for d in dils]).to(dev)
    store = E.ParamStore(convs)
    xs = [_rand(B, Cin, H, W, seed=20 + (0 if shared else i)).to(dev) for i in range(G)]
    dys = [_rand(B, Cout, H, W, seed=30 + i).to(dev) for i in range(G)]
    ys = [torch.empty(B, Cout, H, W, device=dev) for _ in range(G)]
    with E.using_store(store):
        pws = [E.packed_conv(c, True) for c in convs]
    tab = lambda ptrs: (ctypes.c_void_p * G)(*ptrs)
    ints = (ctypes.c_int * G)(*dils)
    s = E._stream()
    _lib.call("cn_conv2d_fwd_grouped_f32", G, tab([x.data_ptr() for x in xs]), E.bstride(xs[0]),
              tab([p.fwd.data_ptr() for p in pws]), None, tab([y.data_ptr() for y in ys]), E.bstride(ys[0]), B, Cin, H,
              W, Cout, 3, 3, 1, ints, ints, 0, s)
    for i in range(G):
        _close(ys[i], convs[i](xs[i]), 2e-5, f"y{i}")
    # bwd-data: distinct outputs, then all branches summed into one buffer
    dxs = [torch.empty(B, Cin, H, W, device=dev) for _ in range(G)]
    k3 = (ctypes.c_int * G)(*([3] * G))
    _lib.call("cn_conv2d_bwd_data_grouped_f32", G, tab([d.data_ptr() for d in dys]), E.bstride(dys[0]),
              tab([p.bwd.data_ptr() for p in pws]), tab([d.data_ptr() for d in dxs]), E.bstride(dxs[0]), B, Cin, H, W,
              Cout, k3, k3, 1, ints, ints, 0, s)
    refs = []
    for i in range(G):
        xr = xs[i].clone().requires_grad_(True)
        convs[i](xr).backward(dys[i])
        refs.append(xr.grad)
        _close(dxs[i], xr.grad, 1e-4, f"dx{i}")
    dsum = torch.empty(B, Cin, H, W, device=dev)
    _lib.call("cn_conv2d_bwd_data_grouped_f32", G, tab([d.data_ptr() for d in dys]), E.bstride(dys[0]),
              tab([p.bwd.data_ptr() for p in pws]), tab([dsum.data_ptr()] * G), E.bstride(dsum), B, Cin, H, W, Cout, k3,
              k3, 1, ints, ints, 0, s)
    _close(dsum, sum(refs), 1e-4, "dx sum")
    if G < 4:  # a 1x1 conv of the same input joins the 3x3 branches (the RESA skip)
        skip = nn.Conv2d(Cin, Cout, 1, bias=False).to(dev)
        store2 = E.ParamStore(skip)
        with E.using_store(store2):
            spw = E.packed_conv(skip, True)
        dys_s = _rand(B, Cout, H, W, seed=44).to(dev)
        xr = xs[0].clone().requires_grad_(True)
        skip(xr).backward(dys_s)
        n = G + 1
        tabn = lambda ptrs: (ctypes.c_void_p * n)(*ptrs)
        _lib.call("cn_conv2d_bwd_data_grouped_f32", n, tabn([d.data_ptr() for d in dys] + [dys_s.data_ptr()]),
                  E.bstride(dys[0]), tabn([p.bwd.data_ptr() for p in pws] + [spw.bwd.data_ptr()]),
                  tabn([dsum.data_ptr()] * n), E.bstride(dsum), B, Cin, H, W, Cout, (ctypes.c_int * n)(*([3] * G + [1])),
                  (ctypes.c_int * n)(*([3] * G + [1])), 1, (ctypes.c_int * n)(*(list(dils) + [0])),
                  (ctypes.c_int * n)(*(list(dils) + [1])), 0, s)
        _close(dsum, sum(refs) + xr.grad, 1e-4, "dx sum + 1x1 skip")


def test_conv2d_splitk_without_workspace_uses_atomics():
    """K-split launches (small spatial size, deep K) on the float-atomic path: no workspace bound."""
    from cultionet_amd import engine as E, _lib

    B, Cin, H, W, Cout = 2, 128, 13, 13, 128
    conv = nn.Conv2d(Cin, Cout, 3, padding=1, bias=True)
    x = _rand(B, Cin, H, W, seed=41)
    xr = x.clone().requires_grad_(True)
    yr = conv(xr)
    dy = _rand(*yr.shape, seed=42)
    yr.backward(dy)
    dev = _dev()
    conv = conv.to(dev)
    store = E.ParamStore(conv)
    store.zero_grad()
    try:
        with E.using_store(store), E.recording(True) as tape:
            _lib.call("cn_conv_set_workspace", E._stream(), None, 0)   # unregister this stream's scratch: atomics
            xv = E.Var(x.to(dev), True)
            y = E.conv2d(xv, conv, 1, 1, 1)
            y.grad = dy.to(dev)
            tape.backward()
        torch.cuda.synchronize()
        _close(y.t, yr, 2e-5, "y")
        _close(xv.grad, xr.grad, 1e-4, "dx")
    finally:
        E._conv_ws.clear()  # the next forward binds the workspace again


@pytest.mark.parametrize("G,shape,summed,train", [(2, (2, 16, 20, 20), False, True), (2, (2, 16, 20, 20), True, True),
                                                   (3, (1, 40, 13, 13), True, True), (2, (2, 8, 28, 28), True, False),
                                                   (4, (2, 32, 25, 25), False, True)])
def test_bn_act_group(G, shape, summed, train):
    """cn_bn_act_group_*: G BatchNorm+SiLU in one launch pair, separately or as res + sum_g f_g(x_g)."""
    import copy
    from cultionet_amd import engine as E

    C = shape[1]
    torch.manual_seed(50)
    bns = nn.ModuleList([nn.BatchNorm2d(C) for _ in range(G)])
    with torch.no_grad():
        for i, bn in enumerate(bns):
            bn.weight.copy_(1 + 0.1 * _rand(C, seed=60 + i))
            bn.bias.copy_(0.1 * _rand(C, seed=70 + i))
            bn.running_mean.copy_(0.1 * _rand(C, seed=80 + i))
            bn.running_var.copy_(torch.rand(C, generator=torch.Generator().manual_seed(90 + i)) + 0.5)
    bns.train(train)
    ref = copy.deepcopy(bns)
    xs = [_rand(*shape, seed=100 + i) * 1.5 + 0.2 for i in range(G)]
    r = _rand(*shape, seed=120)
    xrs = [x.clone().requires_grad_(True) for x in xs]
    rr = r.clone().requires_grad_(True)
    fs = [F.silu(bn(x)) for bn, x in zip(ref, xrs)]
    if summed:
        yr = rr
        for f in fs:
            yr = yr + f
    else:
        yr = torch.cat(fs, dim=1)
    dy = _rand(*yr.shape, seed=121)
    yr.backward(dy)

    def fn(*vs):
        if summed:
            return E.bn_act_group(list(vs[:G]), list(bns), E.ACT_SILU, residual=vs[G], sum_outputs=True, training=train)
        outs = E.bn_act_group(list(vs[:G]), list(bns), E.ACT_SILU, training=train)
        return E.cat_channels(outs)

    y, grads, pg = _engine_run(bns, fn, xs + [r], dy)
    _close(y, yr, 2e-5, "y")
    for i in range(G):
        _close(grads[i], xrs[i].grad, 1e-4, f"dx{i}")
        _close(pg[f"{i}.weight"], ref[i].weight.grad, 1e-4, f"dgamma{i}")
        _close(pg[f"{i}.bias"], ref[i].bias.grad, 1e-4, f"dbeta{i}")
        if train:
            _close(bns[i].running_var.cpu(), ref[i].running_var, 1e-5, f"running_var{i}")
    if summed:
        _close(grads[G], rr.grad, 1e-6, "dres")


@pytest.mark.parametrize("G,B,Cin,H,W,Cout,shared", [(2, 2, 32, 28, 28, 32, True), (2, 2, 40, 25, 25, 48, False),
                                                      (3, 1, 16, 13, 13, 16, True), (2, 8, 128, 25, 25, 128, False),
                                                      (2, 3, 32, 50, 50, 32, True)])
def test_grouped_weight_gradient(G, B, Cin, H, W, Cout, shared):
    """cn_conv2d_bwd_weight_grouped_f32 (aligned and odd sizes, shared / own inputs) against torch."""
    import ctypes
    from cultionet_amd import engine as E, _lib

    dev = _dev()
    torch.manual_seed(12)
    convs = nn.ModuleList([nn.Conv2d(Cin, Cout, 3, padding=1, bias=False) for _ in range(G)]).to(dev)
    store = E.ParamStore(convs)
    store.zero_grad()
    xs = [_rand(B, Cin, H, W, seed=200 + (0 if shared else i)).to(dev) for i in range(G)]
    if shared:
        xs = [xs[0]] * G
    dys = [_rand(B, Cout, H, W, seed=210 + i).to(dev) for i in range(G)]
    tab = lambda ptrs: (ctypes.c_void_p * G)(*ptrs)
    with E.using_store(store):
        wsp, wsn = E._pad_ws(*(xs + dys))
        _lib.call("cn_conv2d_bwd_weight_grouped_f32", G, tab([x.data_ptr() for x in xs]), E.bstride(xs[0]),
                  tab([d.data_ptr() for d in dys]), E.bstride(dys[0]),
                  tab([store.grad_of(c.weight).data_ptr() for c in convs]), B, Cin, H, W, Cout, 3, 3, 1, 1, 1, wsp, wsn,
                  E._stream())
    torch.cuda.synchronize()
    for i, c in enumerate(convs):
        xr = xs[i].clone().requires_grad_(True)
        w = c.weight.detach().clone().requires_grad_(True)
        F.conv2d(xr, w, padding=1).backward(dys[i])
        _close(store.grad_of(c.weight), w.grad, 1e-4, f"dw{i}")


@pytest.mark.parametrize("B,C,H,W", [(2, 16, 12, 12), (1, 40, 9, 11), (3, 128, 25, 25)])
def test_spatial_channel_attention(B, C, H, W):
    """cn_sca_*: out * (1 + gamma*0.5*(channel + spatial attention of skip)) against the oracle's restatement."""
    from cultionet_amd import engine as E
    from cultionet_amd.convolution import SpatialChannelAttention
    from oracle import towerunet_oracle as O

    torch.manual_seed(61)
    mod = SpatialChannelAttention(C, "SiLU")
    with torch.no_grad():
        mod.gamma.fill_(0.7)
    ref = O.SpatialChannelAttention(C, "SiLU")
    ref.load_state_dict(mod.state_dict())
    skip, out = _rand(B, C, H, W, seed=62), _rand(B, C, H, W, seed=63)
    sr, orr = skip.clone().requires_grad_(True), out.clone().requires_grad_(True)
    yr = orr * ref(sr)
    dy = _rand(B, C, H, W, seed=64)
    yr.backward(dy)
    refg = {k: p.grad.clone() for k, p in ref.named_parameters()}
    y, (ds, do), pg = _engine_run(mod, lambda s, o: E.spatial_channel_attention(s, o, mod), [skip, out], dy)
    _close(y, yr, 2e-5, "y")
    _close(do, orr.grad, 1e-4, "dout")
    _close(ds, sr.grad, 1e-4, "dskip")
    for k in refg:
        _close(pg[k], refg[k], 1e-4, k)


# ---------------------------------------------------------------------------
# edge cases at the C ABI: empty batches, invalid arguments, tiny planes, run-to-run determinism of the forward
# ---------------------------------------------------------------------------

def test_empty_batch_and_invalid_arguments():
    import ctypes
    from cultionet_amd import engine as E, _lib

    dev = _dev()
    conv = nn.Conv2d(8, 8, 3, padding=1, bias=False).to(dev)
    store = E.ParamStore(conv)
    with E.using_store(store):
        pw = E.packed_conv(conv, True)
    x = torch.zeros(1, 8, 6, 6, device=dev)
    y = torch.zeros(1, 8, 6, 6, device=dev)
    s = E._stream()
    lib = _lib.load()
    # B == 0: nothing to do, success
    assert lib.cn_conv2d_fwd_f32(x.data_ptr(), 288, pw.fwd.data_ptr(), None, y.data_ptr(), 288, 0, 8, 6, 6, 8, 3, 3, 1, 1,
                                 1, 0, s) == 0
    assert lib.cn_bn_act_fwd_f32(x.data_ptr(), 288, None, None, None, None, None, 0, y.data_ptr(), 288, None, None, None,
                                 0, 8, 36, 1, ctypes.c_float(0.1), ctypes.c_float(1e-5), 1, s) == 0
    # kernels larger than 3x3 / zero stride / too many groups are argument errors, not launches
    assert lib.cn_conv2d_fwd_f32(x.data_ptr(), 288, pw.fwd.data_ptr(), None, y.data_ptr(), 288, 1, 8, 6, 6, 8, 5, 5, 1, 2,
                                 1, 0, s) == -1
    assert lib.cn_conv2d_fwd_f32(x.data_ptr(), 288, pw.fwd.data_ptr(), None, y.data_ptr(), 288, 1, 8, 6, 6, 8, 3, 3, 0, 1,
                                 1, 0, s) == -1
    tab = (ctypes.c_void_p * 5)(*([x.data_ptr()] * 5))
    ints = (ctypes.c_int * 5)(*([1] * 5))
    assert lib.cn_conv2d_fwd_grouped_f32(5, tab, 288, tab, None, tab, 288, 1, 8, 6, 6, 8, 3, 3, 1, ints, ints, 0, s) == -1
    with pytest.raises(_lib.HipKernelError):
        _lib.call("cn_thin_conv3x3_fwd_f32", x.data_ptr(), 288, tab, None, y.data_ptr(), 288, 1, 8, 6, 6, 2, 2, 0, 1,
                  None, s)
    torch.cuda.synchronize()


@pytest.mark.parametrize("H,W", [(4, 4), (5, 7), (3, 16)])
def test_tiny_planes(H, W):
    """Planes smaller than one tile / one wave: conv (+ split-K), BN, LayerNorm, bilinear."""
    from cultionet_amd import engine as E

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.c = nn.Conv2d(40, 24, 3, padding=1, bias=True)
            self.bn = nn.BatchNorm2d(24)
            self.ln = nn.LayerNorm(24)

    torch.manual_seed(71)
    m = M()
    x = _rand(2, 40, H, W, seed=72)
    xr = x.clone().requires_grad_(True)
    h = F.silu(m.bn(m.c(xr)))
    h = m.ln(h.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
    yr = F.interpolate(h, size=(H + 1, W + 2), mode="bilinear", align_corners=True)
    dy = _rand(*yr.shape, seed=73)
    yr.backward(dy)
    ref = {k: p.grad.clone() for k, p in m.named_parameters()}

    def fn(v):
        h = E.bn_act(E.conv2d(v, m.c, 1, 1, 1), m.bn, E.ACT_SILU)
        return E.resize_bilinear(E.layer_norm_c(h, m.ln), (H + 1, W + 2))

    y, (dx,), pg = _engine_run(m, fn, [x], dy)
    _close(y, yr, 2e-5, "y")
    _close(dx, xr.grad, 2e-4, "dx")
    for k in ref:
        _close(pg[k], ref[k], 2e-4, k)


def test_forward_is_bit_reproducible(golden_dir):
    """Two eval forwards of the same model and batch give bit-identical outputs (no atomics on the forward path
    with the split-K workspace bound), hence identical > 0.5 masks."""
    import numpy as np
    import os
    from cultionet_amd import synthetic as O
    from cultionet_amd.data import Data
    from cultionet_amd.lightning import CultionetLitModel

    dev = _dev()
    lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=8, dropout=0.0)
    model = lit.cultionet_model.mask_model
    model.load_state_dict(O.seeded_state_dict(model.state_dict()))
    lit = lit.to(dev).eval()
    x, y, bdist = O.seeded_batch(2, height=28, width=28, seed=5)
    batch = Data(x=x.to(dev), y=y.to(dev), bdist=bdist.to(dev), lon=torch.zeros(2, device=dev),
                 lat=torch.zeros(2, device=dev))
    with torch.no_grad():
        a = {k: v.clone() for k, v in lit(batch).items() if v is not None}
        b = {k: v.clone() for k, v in lit(batch).items() if v is not None}
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_conv_autotune_gives_the_same_results():
    """cn_conv_set_autotune(1): the first launch of a shape times the candidates (output rewritten each time), later
    launches -- accumulating ones included -- reuse the winner; results stay within the usual tolerance."""
    from cultionet_amd import engine as E, _lib

    B, Cin, H, W, Cout = 2, 96, 25, 25, 128
    conv = nn.Conv2d(Cin, Cout, 3, padding=1, bias=True)
    x = _rand(B, Cin, H, W, seed=81)
    xr = x.clone().requires_grad_(True)
    yr = conv(xr)
    dy = _rand(*yr.shape, seed=82)
    yr.backward(dy)
    ref_w = conv.weight.grad.clone()
    try:
        def fn(v):
            _lib.call("cn_conv_set_autotune", 1)   # after using_store bound the workspace
            a = E.conv2d(v, conv, 1, 1, 1)
            return E.conv2d(v, conv, 1, 1, 1) if a is not None else a  # second launch: cached choice

        y, (dx,), pg = _engine_run(conv, fn, [x], dy)
        _close(y, yr, 2e-5, "y")
        _close(dx, xr.grad, 1e-4, "dx")           # one node had no gradient: dx comes from the second conv only
        _close(pg["weight"], ref_w, 1e-4, "dw")
    finally:
        _lib.call("cn_conv_set_autotune", 0)


@pytest.mark.parametrize("shape", [(128, 128, 3), (130, 72, 3), (128, 480, 1), (3, 128, 3), (256, 40, 1)])
def test_pack_weights_batched_matches_single(shape):
    """The tile-transposed batched repack (all layers in one launch) against the plain gather pack, for the four
    stride patterns of include/cultionet_hip.h (Conv2d fwd / bwd-data, ConvTranspose2d fwd / bwd-data)."""
    import struct

    from cultionet_amd import _lib

    cout, cin, k = shape
    taps = k * k
    dev = _dev()
    w = _rand(cout, cin, k, k, seed=5).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    pats = [(cin, cout, taps, cin * taps), (cout, cin, cin * taps, taps),  # conv fwd, conv bwd-data
            (cout, cin, cin * taps, taps), (cin, cout, taps, cin * taps)]  # convT fwd (w as [Cin=cout][Cout=cin]), bwd
    singles, outs, buf = [], [], bytearray()
    for (K, N, sk, sn) in pats:
        kp, np_ = _lib.query("cn_conv_kpad", K), _lib.query("cn_conv_npad", N)
        a = torch.full((taps * kp * np_,), float("nan"), device=dev)
        b = torch.full((taps * kp * np_,), float("nan"), device=dev)
        _lib.call("cn_pack_weights_f32", w.data_ptr(), a.data_ptr(), taps, K, N, sk, sn, 1, s)
        buf += struct.pack("<QQiiiiiiqqq", w.data_ptr(), b.data_ptr(), taps, K, N, kp, np_, 0, sk, sn, 1)
        singles.append(a)
        outs.append(b)
    table = torch.frombuffer(buf, dtype=torch.uint8).clone().to(dev)
    _lib.call("cn_pack_weights_batched_f32", table.data_ptr(), len(pats), s)
    torch.cuda.synchronize()
    for a, b in zip(singles, outs):
        assert torch.equal(a.cpu(), b.cpu())
