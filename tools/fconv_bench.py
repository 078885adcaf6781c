"""Micro-benchmark of one fp32 conv / wgrad shape through the C ABI (for rocprofv3 --pmc passes).

    python tools/fconv_bench.py conv 8 128 100 100 128 3 [iters]
    python tools/fconv_bench.py wgrad 8 128 100 100 128 3 [iters]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cultionet_amd import _lib

kind = sys.argv[1]
B, Cin, H, W, Cout, k = (int(v) for v in sys.argv[2:8])
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 20
dev = torch.device("cuda:0")
T = k * k
p = k // 2
s = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, Cin, H, W, device=dev)
dy = torch.randn(B, Cout, H, W, device=dev)
w = torch.randn(Cout, Cin, k, k, device=dev) * (Cin * T) ** -0.5
wp = torch.empty(T * _lib.query("cn_conv_kpad", Cin) * _lib.query("cn_conv_npad", Cout), device=dev)
_lib.call("cn_pack_weights_f32", w.data_ptr(), wp.data_ptr(), T, Cin, Cout, T, Cin * T, 1, s)
y = torch.empty(B, Cout, H, W, device=dev)
nws = B * (Cout * (H * (W + 1) + 3) + Cin * (H * W + 3)) + (1 << 22)
ws = torch.empty(nws, device=dev)
cws = torch.empty(1 << 24, device=dev)
_lib.call("cn_conv_set_workspace", s, cws.data_ptr(), cws.numel())
dw = torch.zeros(Cout, Cin, k, k, device=dev)


def run():
    if kind == "conv":
        _lib.call("cn_conv2d_fwd_f32", x.data_ptr(), Cin * H * W, wp.data_ptr(), None, y.data_ptr(), Cout * H * W, B, Cin,
                  H, W, Cout, k, k, 1, p, 1, 0, s)
    else:
        _lib.call("cn_conv2d_bwd_weight_f32", x.data_ptr(), Cin * H * W, dy.data_ptr(), Cout * H * W, dw.data_ptr(), B,
                  Cin, H, W, Cout, k, k, 1, p, 1, ws.data_ptr(), nws, s)


for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
fl = 2.0 * B * H * W * Cout * Cin * T
print(f"{kind} fp32 B{B} {Cin}->{Cout} {H}x{W} k{k}: {dt * 1e6:.1f} us  {fl / dt / 1e12:.1f} TFLOP/s")
