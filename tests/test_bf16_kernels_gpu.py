"""Per-kernel parity of the bf16 NHWC mixed-precision entry points (through the C ABI) against plain PyTorch fp32 on
the CPU evaluated on the SAME bf16-rounded inputs.

Tolerances: the kernels accumulate in fp32 from bf16 operands, so against an fp32 evaluation of the rounded operands
the only differences are the summation order and the final rounding of a bf16 OUTPUT (half an ulp = 2^-9 relative).
bf16 outputs: |err| <= 6e-3 * max|ref| (+ tiny absolute); fp32 outputs (weight / parameter gradients, statistics):
|err| <= 2e-3 * max|ref| (inputs are exact, only the order of ~1e5-term fp32 sums differs).
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _s():
    return torch.cuda.current_stream().cuda_stream


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _r(t):  # round to bf16 and back
    return t.to(BF).float()


def _nhwc(t, ld=None):
    """[B,C,H,W] fp32 CPU -> (logical NCHW view of a bf16 NHWC buffer on the GPU with pixel stride ld)."""
    B, C, H, W = t.shape
    ld = ld or C
    buf = torch.zeros((B, H, W, ld), dtype=BF, device=_dev())
    buf[..., :C] = t.permute(0, 2, 3, 1).to(BF).to(_dev())
    return buf[..., :C].permute(0, 3, 1, 2)


def _empty_nhwc(B, C, H, W, ld=None, fill=float("nan")):
    ld = ld or C
    buf = torch.full((B, H, W, ld), fill, dtype=BF, device=_dev())
    return buf[..., :C].permute(0, 3, 1, 2)


def _ld(t):
    return t.stride(3)


def _close(a, b, rel, what="", abs_=1e-6):
    a = a.detach().float().cpu().double()
    b = b.detach().float().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = float(b.abs().max())
    err = float((a - b).abs().max())
    assert err <= rel * scale + abs_, f"{what}: max err {err:.3e} > {rel * scale + abs_:.3e} (scale {scale:.3e})"


def _pack(w, T, K, N, sk, sn, st):
    from cultionet_amd import _lib

    n = _lib.query("cn_bconv_packed_elems", T, K, N)
    wp = torch.empty(n, dtype=BF, device=_dev())
    _lib.call("cn_pack_weights_bf16", w.data_ptr(), wp.data_ptr(), T, K, N, sk, sn, st, _s())
    return wp


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, dil, bias
    (2, 16, 20, 20, 32, 3, 1, 1, 1, False),
    (2, 32, 25, 25, 64, 3, 1, 1, 1, True),
    (1, 72, 13, 13, 128, 3, 1, 1, 1, False),
    (2, 8, 28, 28, 16, 3, 2, 1, 1, False),     # pool conv (stride 2)
    (1, 16, 25, 25, 32, 3, 2, 1, 1, False),    # odd size stride 2 -> 13
    (2, 40, 14, 14, 128, 1, 1, 0, 1, True),    # 1x1 skip with bias
    (2, 128, 13, 13, 256, 1, 1, 0, 1, False),
    (2, 16, 28, 28, 16, 3, 1, 2, 2, False),    # true dilated conv
    (1, 136, 9, 11, 140, 3, 1, 1, 1, True),    # ragged couts, non-square
    (3, 24, 100, 100, 32, 3, 1, 1, 1, False),  # BASELINE spatial size
    (2, 128, 50, 50, 128, 3, 1, 1, 1, False),
    (2, 128, 100, 100, 384, 1, 1, 0, 1, True),  # qkv projection
    (2, 64, 50, 50, 64, 3, 1, 3, 3, False),     # dilations 3 / 4 / 5 at tower sizes: the 16-piece register-prefetched
    (2, 32, 100, 100, 32, 3, 1, 4, 4, False),   # halo of the weight gradient (3, 4) and its synchronous fallback (5)
    (1, 32, 50, 50, 32, 3, 1, 5, 5, False),
    # the shapes that carry the step time at batch 8 (the batch-32 launches only add pixel tiles)
    (8, 128, 100, 100, 128, 3, 1, 1, 1, False),
    (8, 480, 100, 100, 128, 3, 1, 1, 1, False),
    (8, 576, 50, 50, 128, 3, 1, 1, 1, False),
    (8, 640, 25, 25, 128, 3, 1, 1, 1, False),
]


def _random_conv_cases(n, seed):
    """Seeded random shapes around the tile / chunk boundaries of the kernels: channels across the 16 / 64 / 128 steps
    (ragged k-steps and couts), sizes around the 5x25 and 128-pixel tiles, all kernel / stride / dilation modes."""
    import random

    rng = random.Random(seed)
    out = []
    for _ in range(n):
        k = rng.choice([1, 3, 3, 3])
        d = rng.choice([1, 1, 2, 3]) if k == 3 else 1
        s = rng.choice([1, 1, 1, 2]) if d == 1 else 1
        p = d * (k // 2)
        cin = rng.choice([8, 16, 24, 40, 64, 72, 128, 136, 200])
        cout = rng.choice([8, 16, 24, 40, 64, 96, 128, 136, 264])
        h = rng.choice([5, 9, 13, 24, 25, 26, 31, 50, 51])
        w = rng.choice([6, 11, 13, 24, 25, 26, 33, 50, 52])
        if k == 3 and (h <= 2 * d or w <= 2 * d):
            h, w = h + 2 * d, w + 2 * d
        out.append((rng.choice([1, 2, 3]), cin, h, w, cout, k, s, p, d, rng.random() < 0.3))
    return out


@pytest.mark.parametrize("case", CONV_CASES + _random_conv_cases(14, seed=20261002))
def test_conv2d_bf16(case):
    from cultionet_amd import _lib

    B, Cin, H, W, Cout, k, s, p, d, bias = case
    T = k * k
    w = _r(_rand(Cout, Cin, k, k, seed=2, scale=(Cin * T) ** -0.5))
    b = _rand(Cout, seed=3) if bias else None
    x = _r(_rand(B, Cin, H, W, seed=1))
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, stride=s, padding=p, dilation=d)
    dy = _r(_rand(*yr.shape, seed=4))
    yr.backward(dy)
    Ho, Wo = yr.shape[-2:]
    dev = _dev()
    wd = w.to(dev)
    xg = _nhwc(x, ld=Cin + 8)
    wp = _pack(wd, T, Cin, Cout, T, Cin * T, 1)
    y = _empty_nhwc(B, Cout, Ho, Wo, ld=Cout + 16)
    bd = b.to(dev) if bias else None
    rows = _lib.query("cn_conv2d_stats_rows_bf16", B, H, W, Cout, k, k, s, p, d)
    stats = torch.full((rows, 2, Cout), float("nan"), device=dev)
    _lib.call("cn_conv2d_fwd_bf16", xg.data_ptr(), _ld(xg), wp.data_ptr(), bd.data_ptr() if bias else None,
              y.data_ptr(), _ld(y), 0, B, Cin, H, W, Cout, k, k, s, p, d, 0, 0, stats.data_ptr(), _s())
    torch.cuda.synchronize()
    _close(y, yr, 6e-3, "y")
    _close(stats[:, 0].sum(0), yr.detach().sum(dim=(0, 2, 3)), 2e-3, "stats sum", abs_=2e-3 * float(yr.abs().max()) * 10)
    _close(stats[:, 1].sum(0), (yr.detach() ** 2).sum(dim=(0, 2, 3)), 2e-3, "stats sumsq")
    # f32 NCHW output (thin heads) and accumulate
    y32 = torch.full((B, Cout, Ho, Wo), 1.0, device=dev)
    _lib.call("cn_conv2d_fwd_bf16", xg.data_ptr(), _ld(xg), wp.data_ptr(), bd.data_ptr() if bias else None,
              y32.data_ptr(), 0, Cout * Ho * Wo, B, Cin, H, W, Cout, k, k, s, p, d, 1, 1, None, _s())
    _close(y32, yr + 1.0, 2e-3, "y f32 nchw (+=)")
    # backward data
    dyg = _nhwc(dy, ld=Cout + 8)
    wpt = _pack(wd, T, Cout, Cin, Cin * T, T, 1)
    dx = _empty_nhwc(B, Cin, H, W)
    _lib.call("cn_conv2d_bwd_data_bf16", dyg.data_ptr(), _ld(dyg), wpt.data_ptr(), dx.data_ptr(), _ld(dx), B, Cin, H, W,
              Cout, k, k, s, p, d, 0, _s())
    _close(dx, xr.grad, 6e-3, "dx")
    _lib.call("cn_conv2d_bwd_data_bf16", dyg.data_ptr(), _ld(dyg), wpt.data_ptr(), dx.data_ptr(), _ld(dx), B, Cin, H, W,
              Cout, k, k, s, p, d, 1, _s())
    _close(dx, 2 * _r(xr.grad), 1e-2, "dx (+=)")
    # backward weight
    nws = _lib.query("cn_bwgrad_workspace_floats", B, Cin, H, W, Cout, k, k, s, p, d, 0)
    assert nws > 0
    ws = torch.empty(nws, device=dev)
    dw = torch.full((Cout, Cin, k, k), 0.5, device=dev)
    _lib.call("cn_conv2d_bwd_weight_bf16", xg.data_ptr(), _ld(xg), dyg.data_ptr(), _ld(dyg), dw.data_ptr(), B, Cin, H, W,
              Cout, k, k, s, p, d, ws.data_ptr(), nws, _s())
    torch.cuda.synchronize()
    _close(dw, wr.grad + 0.5, 2e-3, "dw")
    # a much smaller workspace: the pixel split shrinks to fit
    small = max(1, nws // 7)
    need1 = (((Cout + 63) // 64) * 64) * (((Cin + 63) // 64) * 64) * T
    small = max(small, need1)
    ws2 = torch.empty(small, device=dev)
    dw2 = torch.zeros((Cout, Cin, k, k), device=dev)
    _lib.call("cn_conv2d_bwd_weight_bf16", xg.data_ptr(), _ld(xg), dyg.data_ptr(), _ld(dyg), dw2.data_ptr(), B, Cin, H,
              W, Cout, k, k, s, p, d, ws2.data_ptr(), small, _s())
    _close(dw2, wr.grad, 2e-3, "dw small ws")


CONVT_CASES = [
    # B, Cin, H, W, Cout, stride
    (2, 16, 13, 13, 16, 2),
    (2, 32, 25, 25, 32, 2),
    (1, 128, 50, 50, 128, 2),   # BASELINE 50 -> 99
    (2, 32, 7, 7, 24, 4),       # final_c style stride 4
    (1, 128, 25, 25, 128, 4),   # 25 -> 97
]


@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv_transpose2d_bf16(case):
    from cultionet_amd import _lib

    B, Cin, H, W, Cout, s = case
    k, p, T = 3, 1, 9
    w = _r(_rand(Cin, Cout, k, k, seed=2, scale=(Cin * T) ** -0.5))
    b = _rand(Cout, seed=3)
    x = _r(_rand(B, Cin, H, W, seed=1))
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, b, stride=s, padding=p)
    dy = _r(_rand(*yr.shape, seed=4))
    yr.backward(dy)
    Ho, Wo = yr.shape[-2:]
    dev = _dev()
    wd, bd = w.to(dev), b.to(dev)
    xg = _nhwc(x)
    wp = _pack(wd, T, Cin, Cout, Cout * T, T, 1)
    y = _empty_nhwc(B, Cout, Ho, Wo)
    _lib.call("cn_conv_transpose2d_fwd_bf16", xg.data_ptr(), _ld(xg), wp.data_ptr(), bd.data_ptr(), y.data_ptr(), _ld(y),
              B, Cin, H, W, Cout, k, k, s, p, 0, _s())
    _close(y, yr, 6e-3, "y")
    dyg = _nhwc(dy)
    wpt = _pack(wd, T, Cout, Cin, T, Cout * T, 1)
    dx = _empty_nhwc(B, Cin, H, W)
    _lib.call("cn_conv_transpose2d_bwd_data_bf16", dyg.data_ptr(), _ld(dyg), wpt.data_ptr(), dx.data_ptr(), _ld(dx), B,
              Cin, H, W, Cout, k, k, s, p, 0, _s())
    _close(dx, xr.grad, 6e-3, "dx")
    nws = _lib.query("cn_bwgrad_workspace_floats", B, Cin, H, W, Cout, k, k, s, p, 1, 1)
    ws = torch.empty(nws, device=dev)
    dw = torch.zeros((Cin, Cout, k, k), device=dev)
    _lib.call("cn_conv_transpose2d_bwd_weight_bf16", xg.data_ptr(), _ld(xg), dyg.data_ptr(), _ld(dyg), dw.data_ptr(), B,
              Cin, H, W, Cout, k, k, s, p, ws.data_ptr(), nws, _s())
    _close(dw, wr.grad, 2e-3, "dw")


@pytest.mark.parametrize("shape,act,res,train", [((2, 16, 20, 20), 1, False, True), ((2, 128, 25, 25), 1, True, True),
                                                  ((1, 480, 13, 13), 0, False, True), ((2, 32, 28, 28), 1, True, False),
                                                  ((3, 64, 50, 50), 1, False, True), ((2, 8, 28, 28), 1, True, True)])
def test_bn_act_bf16(shape, act, res, train):
    from cultionet_amd import _lib

    B, C, H, W = shape
    P = B * H * W
    x = _r(_rand(*shape, seed=1) * 1.5 + 0.3)
    r = _r(_rand(*shape, seed=2)) if res else None
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * _rand(C, seed=3))
        bn.bias.copy_(0.1 * _rand(C, seed=4))
        bn.running_mean.copy_(0.2 * _rand(C, seed=5))
        bn.running_var.copy_(1 + 0.1 * _rand(C, seed=6).abs())
    bn.train(train)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    xr = x.clone().requires_grad_(True)
    z = bn(xr)
    yr = F.silu(z) if act else z
    if res:
        yr = yr + r
    dy = _r(_rand(*shape, seed=7))
    yr.backward(dy)
    dev = _dev()
    xg = _nhwc(x, ld=C + 8)
    rg = _nhwc(r, ld=C + 16) if res else None
    y = _empty_nhwc(B, C, H, W)
    gamma, beta = bn.weight.detach().to(dev), bn.bias.detach().to(dev)
    rm, rv = rm0.to(dev), rv0.to(dev)
    mean, rstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ws = torch.empty(_lib.query("cn_bn_workspace_floats_bf16", C), device=dev)
    _lib.call("cn_bn_act_fwd_bf16", xg.data_ptr(), _ld(xg), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(),
              rv.data_ptr(), rg.data_ptr() if res else None, _ld(rg) if res else 0, y.data_ptr(), _ld(y),
              mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), P, C, 1 if train else 0, 0.1, bn.eps, act, None, 0, _s())
    _close(y, yr, 6e-3, "y")
    if train:
        _close(rm, bn.running_mean, 1e-4, "running_mean", abs_=1e-5)
        _close(rv, bn.running_var, 1e-4, "running_var", abs_=1e-5)
    dyg = _nhwc(dy)
    dx = _empty_nhwc(B, C, H, W)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    _lib.call("cn_bn_act_bwd_bf16", xg.data_ptr(), _ld(xg), dyg.data_ptr(), _ld(dyg), mean.data_ptr(), rstd.data_ptr(),
              gamma.data_ptr(), beta.data_ptr(), dx.data_ptr(), _ld(dx), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), P,
              C, 1 if train else 0, act, 0, _s())
    _close(dx, xr.grad, 8e-3, "dx")
    _close(dg, bn.weight.grad, 2e-3, "dgamma", abs_=1e-3)
    _close(db, bn.bias.grad, 2e-3, "dbeta", abs_=1e-3)


def _tab(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


_SHARED_WS = {}


def _shared_group_ws(need):
    """ONE zero-initialised workspace shared by every grouped-BatchNorm case of this module, as the engine shares one
    buffer between calls of every (G, C): the ticket counters must sit where no other shape ever writes row data."""
    ws = _SHARED_WS.get("ws")
    if ws is None or ws.numel() < need:
        ws = _SHARED_WS["ws"] = torch.zeros(max(need, 8 << 20), device=_dev())
    return ws


@pytest.mark.parametrize("shape,G,summed,act,res,train",
                         [((2, 16, 20, 20), 2, True, 1, True, True), ((2, 128, 25, 25), 2, True, 1, True, True),
                          ((3, 64, 50, 50), 2, False, 1, False, True), ((1, 480, 13, 13), 1, True, 0, False, True),
                          ((2, 32, 28, 28), 2, True, 1, True, False), ((2, 8, 28, 28), 3, True, 1, True, True),
                          ((2, 24, 17, 19), 4, False, 1, False, True), ((8, 128, 100, 100), 2, True, 1, True, True),
                          ((2, 40, 30, 30), 3, False, 0, False, True)])
def test_bn_act_group_bf16(shape, G, summed, act, res, train):
    """cn_bn_act_group_{fwd,bwd}_bf16: G BatchNorm(+SiLU) layers in one launch per pass, ONE finalize launch (coalesced
    column sums + last-block ticket). Summed mode = the ResUNet-a level `res + sum_g SiLU(BN_g(x_g))` (convolution.py:
    376-395) with its shared dy in backward; plain mode = G independent outputs / gradients. Against torch fp32 on the
    same bf16-rounded operands; two calls back to back also check that the ticket counters return to zero."""
    from cultionet_amd import _lib

    B, C, H, W = shape
    P = B * H * W
    dev = _dev()
    xs = [_r(_rand(*shape, seed=10 + g) * (1.0 + 0.3 * g) + 0.2 * g) for g in range(G)]
    r = _r(_rand(*shape, seed=2)) if res else None
    bns = []
    for g in range(G):
        bn = torch.nn.BatchNorm2d(C)
        with torch.no_grad():
            bn.weight.copy_(1 + 0.1 * _rand(C, seed=30 + g))
            bn.bias.copy_(0.1 * _rand(C, seed=40 + g))
            bn.running_mean.copy_(0.2 * _rand(C, seed=50 + g))
            bn.running_var.copy_(1 + 0.1 * _rand(C, seed=60 + g).abs())
        bn.train(train)
        bns.append(bn)
    rm0 = [bn.running_mean.clone() for bn in bns]
    rv0 = [bn.running_var.clone() for bn in bns]
    xr = [x.clone().requires_grad_(True) for x in xs]
    outs = []
    for g in range(G):
        z = bns[g](xr[g])
        outs.append(F.silu(z) if act else z)
    if summed:
        yr = sum(outs) + (r if res else 0.0)
        dy = _r(_rand(*shape, seed=7))
        yr.backward(dy)
        dys = [dy] * G
    else:
        dys = [_r(_rand(*shape, seed=70 + g)) for g in range(G)]
        torch.autograd.backward(outs, dys)
    xg = [_nhwc(x, ld=C + 8) for x in xs]
    rg = _nhwc(r, ld=C + 16) if res else None
    ys = [_empty_nhwc(B, C, H, W) for _ in range(1 if summed else G)]
    gam = [bn.weight.detach().to(dev) for bn in bns]
    bet = [bn.bias.detach().to(dev) for bn in bns]
    rm, rv = [t.to(dev) for t in rm0], [t.to(dev) for t in rv0]
    mean, rstd = torch.empty((G, C), device=dev), torch.empty((G, C), device=dev)
    ws = _shared_group_ws(_lib.query("cn_bn_group_workspace_floats_bf16", G, C))
    for rep in range(2):  # twice: the second call finds the counters as the first left them (zero)
        if rep == 1:
            for g in range(G):
                rm[g].copy_(rm0[g])
                rv[g].copy_(rv0[g])
        _lib.call("cn_bn_act_group_fwd_bf16", G, _tab([t.data_ptr() for t in xg]), _ld(xg[0]),
                  _tab([t.data_ptr() for t in gam]), _tab([t.data_ptr() for t in bet]),
                  _tab([t.data_ptr() for t in rm]), _tab([t.data_ptr() for t in rv]),
                  rg.data_ptr() if res else None, _ld(rg) if res else 0,
                  _tab([(ys[0] if summed else ys[g]).data_ptr() for g in range(G)]), _ld(ys[0]),
                  _tab([mean[g].data_ptr() for g in range(G)]), _tab([rstd[g].data_ptr() for g in range(G)]),
                  ws.data_ptr(), P, C, 1 if train else 0, 0.1, bns[0].eps, act, 1 if summed else 0, None, 0, _s())
        if summed:
            _close(ys[0], yr, 6e-3, "y")
        else:
            for g in range(G):
                _close(ys[g], outs[g], 6e-3, f"y{g}")
        if train:
            for g in range(G):
                _close(rm[g], bns[g].running_mean, 1e-4, "running_mean", abs_=1e-5)
                _close(rv[g], bns[g].running_var, 1e-4, "running_var", abs_=1e-5)
    assert int(ws[:768].view(torch.int32).abs().sum()) == 0  # tickets back to zero
    dyg = [_nhwc(dys[0])] * G if summed else [_nhwc(d) for d in dys]
    base = _r(_rand(*shape, seed=99))
    dxs = [_empty_nhwc(B, C, H, W) for _ in range(G)]
    dxs[-1] = _nhwc(base)  # the last group ACCUMULATES into an existing gradient
    accs = [0] * (G - 1) + [1]
    dg = [torch.zeros(C, device=dev) for _ in range(G)]
    db = [torch.zeros(C, device=dev) for _ in range(G)]
    _lib.call("cn_bn_act_group_bwd_bf16", G, _tab([t.data_ptr() for t in xg]), _ld(xg[0]),
              _tab([t.data_ptr() for t in dyg]), _ld(dyg[0]), _tab([mean[g].data_ptr() for g in range(G)]),
              _tab([rstd[g].data_ptr() for g in range(G)]), _tab([t.data_ptr() for t in gam]),
              _tab([t.data_ptr() for t in bet]), _tab([t.data_ptr() for t in dxs]), _ld(dxs[0]),
              (ctypes.c_int * G)(*accs), _tab([t.data_ptr() for t in dg]), _tab([t.data_ptr() for t in db]),
              ws.data_ptr(), P, C, 1 if train else 0, act, _s())
    for g in range(G):
        want = xr[g].grad + (base if accs[g] else 0.0)
        _close(dxs[g], want, 8e-3, f"dx{g}")
        _close(dg[g], bns[g].weight.grad, 2e-3, f"dgamma{g}", abs_=1e-3)
        _close(db[g], bns[g].bias.grad, 2e-3, f"dbeta{g}", abs_=1e-3)
    assert int(ws[:768].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("P,C,ldx", [(5000, 128, 136), (37, 8, 8), (320000, 128, 128), (700, 480, 480)])
def test_channel_sum_bf16_single_launch(P, C, ldx):
    """Bias gradients on the mixed-precision path: per-channel sums over pixels in ONE launch (last-block ticket),
    accumulate and overwrite forms, twice in a row on the same workspace (the counter returns to zero)."""
    from cultionet_amd import _lib

    dev = _dev()
    x = _r(_rand(P, C, seed=3) + 0.25)
    buf = torch.zeros((P, ldx), dtype=BF, device=dev)
    buf[:, :C] = x.to(BF).to(dev)
    ws = torch.zeros(_lib.query("cn_bn_workspace_floats_bf16", C), device=dev)
    want = x.double().sum(0)
    out = torch.full((C,), 3.0, device=dev)
    _lib.call("cn_channel_sum_bf16", buf.data_ptr(), ldx, P, C, out.data_ptr(), 0, ws.data_ptr(), _s())
    _close(out, want, 2e-3, "sum", abs_=2e-3 * float(x.abs().max()) * P ** 0.5)
    _lib.call("cn_channel_sum_bf16", buf.data_ptr(), ldx, P, C, out.data_ptr(), 1, ws.data_ptr(), _s())
    _close(out, 2 * want, 2e-3, "accumulated", abs_=4e-3 * float(x.abs().max()) * P ** 0.5)
    assert int(ws[:16].view(torch.int32).abs().sum()) == 0


def test_conv_group_bf16_statistics_feed_the_grouped_batchnorm():
    """cn_conv2d_fwd_grouped_bf16 with per-conv statistics rows (two convolutions of one shape in ONE launch, as the
    dilation branches of a ResidualAConv level) -> cn_bn_act_group_fwd_bf16(conv_sums=...): the BatchNorm batch
    statistics come from the conv epilogues' rows, no pass over the conv outputs."""
    from cultionet_amd import _lib

    dev = _dev()
    B, Cin, H, W, Cout, G = 2, 32, 50, 50, 64, 2
    P = B * H * W
    x = _r(_rand(B, Cin, H, W, seed=1))
    ws_ = [_r(_rand(Cout, Cin, 3, 3, seed=2 + g) * 0.1) for g in range(G)]
    pads, dils = [1, 2], [1, 2]
    ref = [F.conv2d(x, ws_[g], padding=pads[g], dilation=dils[g]) for g in range(G)]
    xg = _nhwc(x)
    wps = [_pack(w.to(dev), 9, Cin, Cout, 9, Cin * 9, 1) for w in ws_]
    ys = [_empty_nhwc(B, Cout, H, W) for _ in range(G)]
    rows = _lib.query("cn_conv2d_stats_rows_bf16", B, H, W, Cout, 3, 3, 1, max(pads), max(dils))
    stats = [torch.full((rows, 2, Cout), float("nan"), device=dev) for _ in range(G)]
    _lib.call("cn_conv2d_fwd_grouped_bf16", G, _tab([xg.data_ptr()] * G), _ld(xg), _tab([w.data_ptr() for w in wps]),
              None, _tab([y.data_ptr() for y in ys]), _ld(ys[0]), B, Cin, H, W, Cout, 3, 3, 1,
              (ctypes.c_int * G)(*pads), (ctypes.c_int * G)(*dils), 0, _tab([t.data_ptr() for t in stats]), _s())
    for g in range(G):
        _close(ys[g], ref[g], 6e-3, f"conv{g}")
        _close(stats[g][:, 0].sum(0), ref[g].sum(dim=(0, 2, 3)), 2e-3, "sum", abs_=2e-2 * float(ref[g].abs().max()))
        _close(stats[g][:, 1].sum(0), (ref[g] ** 2).sum(dim=(0, 2, 3)), 2e-3, "sumsq")
    bns = [torch.nn.BatchNorm2d(Cout).train() for _ in range(G)]
    yr = sum(F.silu(bns[g](ref[g])) for g in range(G))
    gam = [bn.weight.detach().to(dev) for bn in bns]
    bet = [bn.bias.detach().to(dev) for bn in bns]
    rm = [torch.zeros(Cout, device=dev) for _ in range(G)]
    rv = [torch.ones(Cout, device=dev) for _ in range(G)]
    mean, rstd = torch.empty((G, Cout), device=dev), torch.empty((G, Cout), device=dev)
    out = _empty_nhwc(B, Cout, H, W)
    ws = torch.zeros(_lib.query("cn_bn_group_workspace_floats_bf16", G, Cout), device=dev)
    _lib.call("cn_bn_act_group_fwd_bf16", G, _tab([y.data_ptr() for y in ys]), _ld(ys[0]),
              _tab([t.data_ptr() for t in gam]), _tab([t.data_ptr() for t in bet]), _tab([t.data_ptr() for t in rm]),
              _tab([t.data_ptr() for t in rv]), None, 0, _tab([out.data_ptr()] * G), _ld(out),
              _tab([mean[g].data_ptr() for g in range(G)]), _tab([rstd[g].data_ptr() for g in range(G)]),
              ws.data_ptr(), P, Cout, 1, 0.1, 1e-5, 1, 1, _tab([t.data_ptr() for t in stats]), rows, _s())
    # (the statistics are those of the fp32 conv results, the normalised tensor is their bf16 rounding: 1.2e-2)
    _close(out, yr, 1.2e-2, "sum of SiLU(BN(conv))")
    for g in range(G):
        _close(rm[g], bns[g].running_mean, 2e-3, "running_mean", abs_=1e-4)
        _close(mean[g], ref[g].mean(dim=(0, 2, 3)), 2e-3, "mean", abs_=1e-4)


@pytest.mark.parametrize("B,Cin,H,W,Cout,G,expect", [
    (2, 32, 50, 50, 64, 2, 1),     # 40 tiles per conv, one cout block
    (3, 16, 25, 25, 40, 1, 1),     # ragged couts (40 of a 64-cout block), one group
    (4, 64, 100, 100, 256, 2, 1),  # 313 tiles x 2 cout blocks x 2 groups: 20 ticket groups per domain
    (2, 8, 13, 13, 8, 4, 1),       # tiny planes, four groups
    (14, 16, 100, 100, 128, 1, 0),  # 1120 tiles of 128 pixels > 1008: rows only, the caller runs the finalize
])
def test_conv_bnstats_bf16_finishes_its_own_batchnorm_statistics(B, Cin, H, W, Cout, G, expect):
    """cn_conv2d_fwd_grouped_bnstats_bf16: the convolution launch finishes the BatchNorm batch statistics of its outputs
    (two-level last-block ticket per group and cout block) -- mean / rstd / running statistics against torch's
    BatchNorm2d on the fp32 convolution, twice in a row on ONE workspace (the tickets must be back at zero), then
    cn_bn_act_group_fwd_bf16(conv_rows=-1) applies them. Above 1008 tiles the launch reports `finalized = 0`."""
    from cultionet_amd import _lib

    dev = _dev()
    P = B * H * W
    x = _r(_rand(B, Cin, H, W, seed=1))
    ws_ = [_r(_rand(Cout, Cin, 3, 3, seed=2 + g) * 0.1) for g in range(G)]
    pads = dils = [1 + (g % 2) for g in range(G)]
    ref = [F.conv2d(x, ws_[g], padding=pads[g], dilation=dils[g]) for g in range(G)]
    xg = _nhwc(x)
    wps = [_pack(w.to(dev), 9, Cin, Cout, 9, Cin * 9, 1) for w in ws_]
    ys = [_empty_nhwc(B, Cout, H, W) for _ in range(G)]
    rows = _lib.query("cn_conv2d_stats_rows_bf16", B, H, W, Cout, 3, 3, 1, max(pads), max(dils))
    stats = [torch.full((rows, 2, Cout), float("nan"), device=dev) for _ in range(G)]
    bns = [torch.nn.BatchNorm2d(Cout).train() for _ in range(G)]
    rm = [torch.zeros(Cout, device=dev) for _ in range(G)]
    rv = [torch.ones(Cout, device=dev) for _ in range(G)]
    mean = torch.full((G, Cout), float("nan"), device=dev)
    rstd = torch.full((G, Cout), float("nan"), device=dev)
    nws = _lib.query("cn_bn_group_workspace_floats_bf16", G, Cout)
    ws = torch.zeros(nws, device=dev)
    fin = ctypes.c_int(-7)
    for rep in range(2):
        _lib.call("cn_conv2d_fwd_grouped_bnstats_bf16", G, _tab([xg.data_ptr()] * G), _ld(xg),
                  _tab([w.data_ptr() for w in wps]), _tab([y.data_ptr() for y in ys]), _ld(ys[0]), B, Cin, H, W, Cout, 3, 3,
                  1, (ctypes.c_int * G)(*pads), (ctypes.c_int * G)(*dils), _tab([t.data_ptr() for t in stats]),
                  _tab([mean[g].data_ptr() for g in range(G)]), _tab([rstd[g].data_ptr() for g in range(G)]),
                  _tab([t.data_ptr() for t in rm]), _tab([t.data_ptr() for t in rv]), 0.1, 1e-5, ws.data_ptr(), nws,
                  ctypes.byref(fin), _s())
        torch.cuda.synchronize()
        assert fin.value == expect
        [bn(r) for bn, r in zip(bns, ref)]  # torch's running statistics move once per repetition too
        for g in range(G):
            _close(ys[g], ref[g], 6e-3, f"conv{g}")
            _close(stats[g][:, 0].sum(0), ref[g].sum(dim=(0, 2, 3)), 2e-3, "sum", abs_=2e-2 * float(ref[g].abs().max()))
        if not expect:
            assert torch.isnan(mean).all() and float(rm[0].abs().max()) == 0.0  # untouched: the caller finalizes
            continue
        for g in range(G):
            _close(mean[g], ref[g].mean(dim=(0, 2, 3)), 2e-3, "mean", abs_=1e-4)
            _close(rstd[g], 1.0 / torch.sqrt(ref[g].var(dim=(0, 2, 3), unbiased=False) + 1e-5), 2e-3, "rstd")
            _close(rm[g], bns[g].running_mean, 2e-3, f"running_mean rep {rep}", abs_=1e-4)
            _close(rv[g], bns[g].running_var, 2e-3, f"running_var rep {rep}")
        head = ws[:_lib.query("cn_bn_group_workspace_floats_bf16", 1, 8)][:4864].view(torch.int32)
        assert int(head.abs().sum()) == 0  # every ticket counter is back at zero
    gam = [bn.weight.detach().to(dev) for bn in bns]
    bet = [bn.bias.detach().to(dev) for bn in bns]
    out = _empty_nhwc(B, Cout, H, W)
    rm2, rv2 = [t.clone() for t in rm], [t.clone() for t in rv]
    _lib.call("cn_bn_act_group_fwd_bf16", G, _tab([y.data_ptr() for y in ys]), _ld(ys[0]),
              _tab([t.data_ptr() for t in gam]), _tab([t.data_ptr() for t in bet]), _tab([t.data_ptr() for t in rm]),
              _tab([t.data_ptr() for t in rv]), None, 0, _tab([out.data_ptr()] * G), _ld(out),
              _tab([mean[g].data_ptr() for g in range(G)]), _tab([rstd[g].data_ptr() for g in range(G)]),
              ws.data_ptr(), P, Cout, 1, 0.1, 1e-5, 1, 1,
              None if expect else _tab([t.data_ptr() for t in stats]), -1 if expect else rows, _s())
    yr = sum(F.silu(F.batch_norm(ref[g], None, None, bns[g].weight, bns[g].bias, True, 0.0, 1e-5)) for g in range(G))
    _close(out, yr, 1.2e-2, "sum of SiLU(BN(conv))")
    if expect:  # apply-only: the running statistics are not updated a second time
        for g in range(G):
            assert torch.equal(rm[g], rm2[g]) and torch.equal(rv[g], rv2[g])


@pytest.mark.parametrize("shape,res", [((2, 32, 20, 20), False), ((2, 128, 25, 25), True), ((1, 8, 13, 13), True),
                                       ((2, 256, 9, 9), False)])
def test_layernorm_c_bf16(shape, res):
    from cultionet_amd import _lib

    B, C, H, W = shape
    P = B * H * W
    x = _r(_rand(*shape, seed=1) * 1.5 + 0.3)
    r = _r(_rand(*shape, seed=2)) if res else None
    ln = torch.nn.LayerNorm(C)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * _rand(C, seed=3))
        ln.bias.copy_(0.1 * _rand(C, seed=4))
    xr = x.clone().requires_grad_(True)
    yr = ln(xr.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
    if res:
        yr = yr + r
    dy = _r(_rand(*shape, seed=7))
    yr.backward(dy)
    dev = _dev()
    xg = _nhwc(x, ld=C + 8)
    rg = _nhwc(r) if res else None
    y = _empty_nhwc(B, C, H, W)
    w, b = ln.weight.detach().to(dev), ln.bias.detach().to(dev)
    _lib.call("cn_layernorm_c_fwd_bf16", xg.data_ptr(), _ld(xg), w.data_ptr(), b.data_ptr(),
              rg.data_ptr() if res else None, _ld(rg) if res else 0, y.data_ptr(), _ld(y), P, C, ln.eps, _s())
    _close(y, yr, 6e-3, "y")
    dyg = _nhwc(dy)
    dx = _empty_nhwc(B, C, H, W)
    dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    _lib.call("cn_layernorm_c_bwd_bf16", xg.data_ptr(), _ld(xg), dyg.data_ptr(), _ld(dyg), w.data_ptr(), dx.data_ptr(),
              _ld(dx), dw.data_ptr(), db.data_ptr(), P, C, ln.eps, 0, _s())
    _close(dx, xr.grad, 8e-3, "dx")
    _close(dw, ln.weight.grad, 2e-3, "dw", abs_=1e-3)
    _close(db, ln.bias.grad, 2e-3, "db", abs_=1e-3)


@pytest.mark.parametrize("shape,heads,dil", [((2, 32, 12, 12), 4, 1), ((1, 32, 14, 15), 4, 2), ((2, 32, 9, 9), 8, 1),
                                             ((1, 128, 25, 25), 8, 1), ((2, 128, 20, 20), 4, 2)])
def test_na2d_bf16(shape, heads, dil):
    from cultionet_amd import _lib
    from oracle import na2d_ref

    B, C, H, W = shape
    qkv = _r(_rand(B, 3 * C, H, W, seed=1))
    D = C // heads
    qr = qkv.clone().requires_grad_(True)
    # oracle layout: q, k, v as [B, heads, H, W, D]
    t = qr.view(B, 3, heads, D, H, W).permute(1, 0, 2, 4, 5, 3)
    q, k_, v = t[0], t[1], t[2]
    attn = na2d_ref.na2d_qk(q * (D ** -0.5), k_, 3, dil).softmax(dim=-1)
    o = na2d_ref.na2d_av(attn, v, 3, dil)  # [B, heads, H, W, D]
    outr = o.permute(0, 1, 4, 2, 3).reshape(B, C, H, W)
    dy = _r(_rand(B, C, H, W, seed=2))
    outr.backward(dy)
    dev = _dev()
    qg = _nhwc(qkv)
    out = _empty_nhwc(B, C, H, W)
    at = torch.empty((B, heads, 9, H, W), device=dev)
    _lib.call("cn_na2d_fwd_bf16", qg.data_ptr(), _ld(qg), out.data_ptr(), _ld(out), at.data_ptr(), B, C, heads, H, W, 3,
              dil, 0.0, 0, None, _s())
    _close(out, outr, 6e-3, "out")
    dyg = _nhwc(dy)
    dq = _empty_nhwc(B, 3 * C, H, W)
    dat = torch.empty_like(at)
    _lib.call("cn_na2d_bwd_bf16", qg.data_ptr(), _ld(qg), dyg.data_ptr(), _ld(dyg), at.data_ptr(), dat.data_ptr(),
              dq.data_ptr(), _ld(dq), B, C, heads, H, W, 3, dil, 0.0, 0, None, _s())
    _close(dq, qr.grad, 8e-3, "dqkv")


@pytest.mark.parametrize("shape,heads,dil", [((2, 32, 12, 12), 4, 1), ((1, 128, 20, 21), 4, 2)])
def test_na2d_bf16_attention_dropout(shape, heads, dil):
    """attn_drop > 0 on the mixed-precision path (the reference's default dropout=0.1 reaches natten's attn_drop in the
    decoder, unet_parts.py:452-525): the bf16 kernels draw the SAME counter-based masks as the fp32 kernels, so on the
    same bf16-rounded operands and seed both paths must agree to bf16 rounding, forward and backward."""
    from cultionet_amd import _lib

    B, C, H, W = shape
    dev = _dev()
    p, seed = 0.25, 0xABCDEF12345
    qkv = _r(_rand(B, 3 * C, H, W, seed=1))
    dy = _r(_rand(B, C, H, W, seed=2))
    # fp32 NCHW kernels
    q32 = qkv.to(dev)
    o32 = torch.empty((B, C, H, W), device=dev)
    a32 = torch.empty((B, heads, 9, H, W), device=dev)
    _lib.call("cn_na2d_fwd_f32", q32.data_ptr(), 3 * C * H * W, o32.data_ptr(), C * H * W, a32.data_ptr(), B, C, heads,
              H, W, 3, dil, p, seed, None, _s())
    dy32 = dy.to(dev)
    da32 = torch.empty_like(a32)
    dq32 = torch.empty_like(q32)
    _lib.call("cn_na2d_bwd_f32", q32.data_ptr(), 3 * C * H * W, dy32.data_ptr(), C * H * W, a32.data_ptr(),
              da32.data_ptr(), dq32.data_ptr(), 3 * C * H * W, B, C, heads, H, W, 3, dil, p, seed, None, _s())
    # bf16 NHWC kernels
    qg = _nhwc(qkv)
    out = _empty_nhwc(B, C, H, W)
    at = torch.empty((B, heads, 9, H, W), device=dev)
    _lib.call("cn_na2d_fwd_bf16", qg.data_ptr(), _ld(qg), out.data_ptr(), _ld(out), at.data_ptr(), B, C, heads, H, W, 3,
              dil, p, seed, None, _s())
    _close(at, a32, 1e-5, "saved probabilities (undropped)")
    _close(out, o32, 6e-3, "out")
    # the dropout is real: without it the output differs
    out0 = _empty_nhwc(B, C, H, W)
    _lib.call("cn_na2d_fwd_bf16", qg.data_ptr(), _ld(qg), out0.data_ptr(), _ld(out0), at.data_ptr(), B, C, heads, H, W, 3,
              dil, 0.0, 0, None, _s())
    assert (out0.float() - out.float()).abs().max() > 0.05
    dyg = _nhwc(dy)
    dq = _empty_nhwc(B, 3 * C, H, W)
    dat = torch.empty_like(at)
    _lib.call("cn_na2d_bwd_bf16", qg.data_ptr(), _ld(qg), dyg.data_ptr(), _ld(dyg), at.data_ptr(), dat.data_ptr(),
              dq.data_ptr(), _ld(dq), B, C, heads, H, W, 3, dil, p, seed, None, _s())
    _close(dq, dq32, 8e-3, "dqkv")


@pytest.mark.parametrize("channelwise", [True, False])
def test_dropout_bf16_draws_the_fp32_masks(channelwise):
    """cn_dropout_bf16 (NHWC) against cn_dropout_f32 (NCHW) on the same seed: identical keep pattern, kept values
    x / (1 - p) to bf16 rounding, backward (accumulate) with the same mask."""
    from cultionet_amd import _lib

    dev = _dev()
    B, C, H, W = 3, 24, 9, 11
    p, seed = 0.3, 0x1234567
    x = _r(_rand(B, C, H, W, seed=5).abs() + 0.5)
    x32 = x.to(dev)
    y32 = torch.empty_like(x32)
    cw = 1 if channelwise else 0
    _lib.call("cn_dropout_f32", x32.data_ptr(), C * H * W, y32.data_ptr(), C * H * W, B, C, H * W, p, seed, None, cw, 0, _s())
    xg = _nhwc(x, ld=C + 8)
    y = _empty_nhwc(B, C, H, W, ld=C + 16)
    _lib.call("cn_dropout_bf16", xg.data_ptr(), _ld(xg), y.data_ptr(), _ld(y), B, C, H * W, p, seed, None, cw, 0, _s())
    torch.cuda.synchronize()
    assert torch.equal(y.float().cpu() != 0, y32.cpu() != 0)
    _close(y, y32, 4e-3, "y")
    frac = (y32 != 0).float().mean().item()
    assert abs(frac - (1 - p)) < (0.15 if channelwise else 0.03), frac
    base = _r(_rand(B, C, H, W, seed=6))
    acc = _nhwc(base)
    _lib.call("cn_dropout_bf16", xg.data_ptr(), _ld(xg), acc.data_ptr(), _ld(acc), B, C, H * W, p, seed, None, cw, 1, _s())
    _close(acc, y32.cpu() + base, 8e-3, "accumulate")


@pytest.mark.parametrize("case", [(2, 16, 13, 13, 14, 14), (1, 32, 49, 49, 50, 50), (2, 8, 97, 97, 100, 100),
                                  (1, 8, 25, 25, 25, 25), (2, 16, 7, 9, 20, 23), (1, 8, 40, 40, 13, 17),
                                  (2, 32, 25, 25, 100, 100), (1, 16, 13, 13, 100, 100), (1, 8, 100, 100, 25, 25),
                                  (1, 8, 5, 300, 9, 310), (1, 8, 1, 6, 4, 1), (1, 8, 3, 3, 64, 64),
                                  (1, 8, 4, 260, 6, 500), (1, 16, 51, 51, 100, 100)])
def test_bilinear_bf16(case):
    from cultionet_amd import _lib

    B, C, Hi, Wi, Ho, Wo = case
    x = _r(_rand(B, C, Hi, Wi, seed=1))
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=True)
    dy = _r(_rand(B, C, Ho, Wo, seed=2))
    yr.backward(dy)
    xg = _nhwc(x, ld=C + 8)
    y = _empty_nhwc(B, C, Ho, Wo, ld=C + 24)
    _lib.call("cn_bilinear_fwd_bf16", xg.data_ptr(), _ld(xg), y.data_ptr(), _ld(y), B, C, Hi, Wi, Ho, Wo, _s())
    _close(y, yr, 6e-3, "y")
    dyg = _nhwc(dy)
    dx = _empty_nhwc(B, C, Hi, Wi)
    _lib.call("cn_bilinear_bwd_bf16", dyg.data_ptr(), _ld(dyg), dx.data_ptr(), _ld(dx), B, C, Hi, Wi, Ho, Wo, 0, _s())
    _close(dx, xr.grad, 6e-3, "dx")
    # accumulate mode: dx += adjoint
    base = _r(_rand(B, C, Hi, Wi, seed=3))
    dx2 = _nhwc(base)
    _lib.call("cn_bilinear_bwd_bf16", dyg.data_ptr(), _ld(dyg), dx2.data_ptr(), _ld(dx2), B, C, Hi, Wi, Ho, Wo, 1, _s())
    _close(dx2, xr.grad + base, 1.2e-2, "dx accumulate")


def test_converters_and_slices_bf16():
    from cultionet_amd import _lib

    dev = _dev()
    B, C, H, W = 2, 9, 13, 11
    x = _rand(B, C, H, W, seed=1)
    xd = x.to(dev)
    buf = torch.full((B, H, W, 24), float("nan"), dtype=BF, device=dev)
    _lib.call("cn_convert_f32nchw_to_bf16nhwc", xd.data_ptr(), C * H * W, buf.data_ptr(), 24, B, C, 16, H * W, _s())
    got = buf[..., :16].float().cpu()
    assert torch.equal(got[..., :C], _r(x).permute(0, 2, 3, 1))
    assert torch.equal(got[..., C:], torch.zeros(B, H, W, 16 - C))
    back = torch.full((B, C, H, W), 2.0, device=dev)
    _lib.call("cn_convert_bf16nhwc_to_f32nchw", buf.data_ptr(), 24, back.data_ptr(), C * H * W, B, C, H * W, 1, _s())
    assert torch.equal(back.cpu(), _r(x) + 2.0)
    # slice copy / add / zero
    a = _r(_rand(B, 16, H, W, seed=2))
    c = _r(_rand(B, 16, H, W, seed=3))
    ag, cg = _nhwc(a, ld=40), _nhwc(c)
    d = _empty_nhwc(B, 16, H, W, ld=32, fill=0.0)
    _lib.call("cn_copy_bf16", ag.data_ptr(), _ld(ag), d.data_ptr(), _ld(d), B * H * W, 16, 0, _s())
    assert torch.equal(d.float().cpu(), a)
    _lib.call("cn_copy_bf16", cg.data_ptr(), _ld(cg), d.data_ptr(), _ld(d), B * H * W, 16, 1, _s())
    assert torch.equal(d.float().cpu(), _r(a + c))
    _lib.call("cn_add_bf16", ag.data_ptr(), _ld(ag), cg.data_ptr(), _ld(cg), d.data_ptr(), _ld(d), B * H * W, 16, _s())
    assert torch.equal(d.float().cpu(), _r(a + c))
    _lib.call("cn_zero_bf16", d.data_ptr(), _ld(d), B * H * W, 16, _s())
    assert float(d.float().abs().max()) == 0.0


@pytest.mark.parametrize("case", [(2, 32, 25, 25, 64, 3, 1, 1, 1, 1, True), (1, 72, 13, 13, 128, 3, 1, 1, 1, 0, False),
                                  (2, 16, 28, 28, 16, 3, 2, 1, 1, 0, False), (2, 40, 14, 14, 128, 1, 1, 0, 1, 1, True),
                                  (2, 128, 50, 50, 128, 3, 1, 1, 1, 1, True), (1, 64, 20, 20, 64, 3, 1, 2, 2, 1, False),
                                  (3, 128, 110, 110, 128, 3, 1, 1, 1, 1, True)])
def test_conv_bn_act_fused_eval_bf16(case):
    """cn_conv2d_fwd_fused_bf16 + cn_bn_fold_f32 + cn_pack_weights_scaled_bf16: y = res + SiLU(BN_eval(conv(x))) in one
    launch against torch (conv2d -> batch_norm(training=False) -> silu -> + res) in fp32 on the bf16-rounded operands.
    The folded weights W * gamma / sigma are rounded to bf16 AFTER the scaling: one more half-ulp than the unfused form,
    hence the slightly wider tolerance (8e-3 of the output range)."""
    from cultionet_amd import _lib

    B, Cin, H, W, Cout, k, s, p, d, act, with_res = case
    T = k * k
    dev = _dev()
    w = _rand(Cout, Cin, k, k, seed=2, scale=(Cin * T) ** -0.5)
    gamma, beta = 1 + 0.2 * _rand(Cout, seed=3), 0.3 * _rand(Cout, seed=4)
    mean = 0.2 * _rand(Cout, seed=5)
    var = torch.rand(Cout, generator=torch.Generator().manual_seed(6)) + 0.4
    eps = 1e-5
    x = _r(_rand(B, Cin, H, W, seed=1))
    scale = gamma / torch.sqrt(var + eps)
    # reference: the folded weights as the kernel rounds them
    wf = _r(w * scale.view(-1, 1, 1, 1))
    yr = F.conv2d(x, wf, None, stride=s, padding=p, dilation=d) + (beta - mean * scale).view(1, -1, 1, 1)
    if act:
        yr = F.silu(yr)
    Ho, Wo = yr.shape[-2:]
    res = _r(_rand(B, Cout, Ho, Wo, seed=7)) if with_res else None
    if with_res:
        yr = yr + res
    # and it is the module semantics: conv -> BatchNorm(eval) -> SiLU (+ res), up to that extra rounding
    ym = F.batch_norm(F.conv2d(x, _r(w), None, stride=s, padding=p, dilation=d), mean, var, gamma, beta, False, 0.0, eps)
    ym = (F.silu(ym) if act else ym) + (res if with_res else 0)
    assert (ym - yr).abs().max() <= 2e-2 * max(1.0, float(ym.abs().max()))
    sc = torch.empty(Cout, device=dev)
    sh = torch.empty(Cout, device=dev)
    gd, bd, md, vd = (t.to(dev) for t in (gamma, beta, mean, var))
    _lib.call("cn_bn_fold_f32", gd.data_ptr(), bd.data_ptr(), md.data_ptr(), vd.data_ptr(), None, eps, Cout,
              sc.data_ptr(), sh.data_ptr(), _s())
    _close(sc, scale, 1e-6, "scale")
    wd = w.to(dev)
    wp = torch.empty(_lib.query("cn_bconv_packed_elems", T, Cin, Cout), dtype=BF, device=dev)
    _lib.call("cn_pack_weights_scaled_bf16", wd.data_ptr(), sc.data_ptr(), wp.data_ptr(), T, Cin, Cout, T, Cin * T, 1, _s())
    xg = _nhwc(x, ld=Cin + 8)
    y = _empty_nhwc(B, Cout, Ho, Wo, ld=Cout + 16)
    rg = _nhwc(res, ld=Cout + 8) if with_res else None
    _lib.call("cn_conv2d_fwd_fused_bf16", xg.data_ptr(), _ld(xg), wp.data_ptr(), sh.data_ptr(),
              rg.data_ptr() if with_res else None, _ld(rg) if with_res else 0, y.data_ptr(), _ld(y), B, Cin, H, W, Cout,
              k, k, s, p, d, act, _s())
    _close(y, yr, 8e-3, "y")
    if with_res:  # in place: res aliases y (the ResUNet-a running sum accumulated where it lives)
        y2 = _nhwc(res, ld=Cout + 8)
        _lib.call("cn_conv2d_fwd_fused_bf16", xg.data_ptr(), _ld(xg), wp.data_ptr(), sh.data_ptr(), y2.data_ptr(),
                  _ld(y2), y2.data_ptr(), _ld(y2), B, Cin, H, W, Cout, k, k, s, p, d, act, _s())
        _close(y2, yr, 8e-3, "y in place")


@pytest.mark.parametrize("shape", [(128, 128, 3), (130, 72, 3), (128, 480, 1), (8, 128, 3), (256, 40, 1), (512, 256, 3)])
def test_pack_weights_batched_bf16_matches_single(shape):
    """The LDS-tiled batched repack (fp32 masters -> bf16 MFMA fragments, all layers in one launch) against the plain
    gather pack, for the stride patterns of Conv2d fwd / bwd-data and ConvTranspose2d fwd / bwd-data: bit-identical."""
    import struct

    from cultionet_amd import _lib

    cout, cin, k = shape
    taps = k * k
    dev = _dev()
    w = _rand(cout, cin, k, k, seed=5).to(dev)
    s = _s()
    pats = [(cin, cout, taps, cin * taps), (cout, cin, cin * taps, taps),  # conv fwd, conv bwd-data
            (cout, cin, cin * taps, taps), (cin, cout, taps, cin * taps)]  # convT fwd (w as [Cin=cout][Cout=cin]), bwd
    singles, outs, buf = [], [], bytearray()
    for (K, N, sk, sn) in pats:
        n = _lib.query("cn_bconv_packed_elems", taps, K, N)
        a = torch.full((n,), float("nan"), dtype=BF, device=dev)
        b = torch.full((n,), float("nan"), dtype=BF, device=dev)
        _lib.call("cn_pack_weights_bf16", w.data_ptr(), a.data_ptr(), taps, K, N, sk, sn, 1, s)
        buf += struct.pack("<QQiiiiiiqqqQ", w.data_ptr(), b.data_ptr(), taps, K, N, (K + 15) // 16, (N + 31) // 32, 0, sk,
                           sn, 1, 0)
        singles.append(a)
        outs.append(b)
    table = torch.frombuffer(buf, dtype=torch.uint8).clone().to(dev)
    _lib.call("cn_pack_weights_batched_bf16", table.data_ptr(), len(pats), s)
    torch.cuda.synchronize()
    for a, b in zip(singles, outs):
        assert torch.equal(a.view(torch.int16).cpu(), b.view(torch.int16).cpu())
