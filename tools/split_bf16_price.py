"""Pricing (NOT shipping) an fp32 convolution as split-bf16 products on the bf16 matrix pipe.

fp32 MFMA runs at 1/16 of the dense bf16 rate on gfx950. Writing each fp32 operand as a sum of three bf16 values,
a = a0 + a1 + a2 (a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1): 24 mantissa bits), the product a * w is
recovered to ~2^-24 relative by the six bf16 products with i + j <= 2 accumulated in fp32:
    a w ~= a0 w0 + (a0 w1 + a1 w0) + (a0 w2 + a1 w1 + a2 w0).
This tool prices that on the layer that carries most of the fp32 step -- 128 -> 128, 3x3, 8 x 100^2 -- with the SHIPPED
kernels: six launches of cn_conv2d_fwd_bf16 (fp32 NCHW epilogue, accumulate) on pre-split operands, plus the split passes,
against cn_conv2d_fwd_f32; error against torch fp32 on the CPU at the fp32 kernels' tolerance (2e-5 * scale).

    python tools/split_bf16_price.py [B] [C] [H]
"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from cultionet_amd import _lib  # noqa: E402

B, C, H = (int(v) for v in (sys.argv[1:] + ["8", "128", "100"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(B, C, H, H, generator=g)
w = torch.randn(C, C, 3, 3, generator=g) * (1.0 / (C * 9) ** 0.5)
ref = F.conv2d(x.double(), w.double(), padding=1).float()
s = torch.cuda.current_stream().cuda_stream


def split3(t):
    t0 = t.to(torch.bfloat16)
    r = t - t0.float()
    t1 = r.to(torch.bfloat16)
    t2 = (r - t1.float()).to(torch.bfloat16)
    return t0, t1, t2


xd, wd = x.to(dev), w.to(dev)
# fp32 kernel
kp, np_ = _lib.query("cn_conv_kpad", C), _lib.query("cn_conv_npad", C)
wp32 = torch.empty(9 * kp * np_, device=dev)
_lib.call("cn_pack_weights_f32", wd.data_ptr(), wp32.data_ptr(), 9, C, C, 9, C * 9, 1, s)
y32 = torch.empty(B, C, H, H, device=dev)
wsf = torch.empty(16 << 20, device=dev)
_lib.call("cn_conv_set_workspace", s, wsf.data_ptr(), wsf.numel())


def run32():
    _lib.call("cn_conv2d_fwd_f32", xd.data_ptr(), C * H * H, wp32.data_ptr(), None, y32.data_ptr(), C * H * H, B, C, H, H, C,
              3, 3, 1, 1, 1, 0, s)


# split operands: activations NHWC bf16 x3, packed weights x3
xs = [t.permute(0, 2, 3, 1).contiguous() for t in split3(xd)]
wsplit = split3(wd)
wps = []
for t in wsplit:
    tf = t.float().contiguous()
    wp = torch.empty(_lib.query("cn_bconv_packed_elems", 9, C, C), dtype=torch.bfloat16, device=dev)
    _lib.call("cn_pack_weights_bf16", tf.data_ptr(), wp.data_ptr(), 9, C, C, 9, C * 9, 1, s)
    wps.append(wp)
ysp = torch.empty(B, C, H, H, device=dev)
pairs = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]


def run_split(npairs=6):
    for n, (i, j) in enumerate(pairs[:npairs]):
        _lib.call("cn_conv2d_fwd_bf16", xs[i].data_ptr(), C, wps[j].data_ptr(), None, ysp.data_ptr(), 0, C * H * H, B, C, H, H,
                  C, 3, 3, 1, 1, 1, 1 if n else 0, 1, None, s)


def split_pass():  # the activation split a real implementation would do at production time (3 bf16 tensors per fp32 one)
    return [t.permute(0, 2, 3, 1).contiguous() for t in split3(xd)]


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


run32()
run_split()
torch.cuda.synchronize()
scale = float(ref.abs().max())
e32 = float((y32.cpu() - ref).abs().max()) / scale
esp = float((ysp.cpu() - ref).abs().max()) / scale
run_split(3)
torch.cuda.synchronize()
esp3 = float((ysp.cpu() - ref).abs().max()) / scale
flop = 2.0 * B * H * H * C * C * 9
t32, t6, t3, tsp = timeit(run32), timeit(run_split), timeit(lambda: run_split(3)), timeit(split_pass, 10)
print(f"layer {C}->{C} 3x3 @ {B} x {H}^2: {flop / 1e9:.2f} GFLOP")
print(f"fp32 MFMA kernel        : {t32:8.1f} us  {flop / t32 / 1e6:7.1f} TFLOP/s   max err / scale {e32:.2e}")
print(f"split-bf16, 6 products  : {t6:8.1f} us  {flop / t6 / 1e6:7.1f} TFLOP/s (fp32-equivalent)   max err / scale {esp:.2e}")
print(f"split-bf16, 3 products  : {t3:8.1f} us  {flop / t3 / 1e6:7.1f} TFLOP/s (fp32-equivalent)   max err / scale {esp3:.2e}")
print(f"activation split (torch ops, 3 bf16 NHWC copies of the fp32 tensor): {tsp:8.1f} us")
