"""How much does a compute-stream kernel slow down NEXT TO a weight gradient of the side stream?  (round 5)

A low-priority stream (cn_stream_create, as the engine's side stream) runs `cn_conv2d_bwd_weight_f32` launches back to
back; the compute stream meanwhile times K launches of a small kernel with HIP events. Reports alone / co-running.
    python tools/corun.py [side_cin side_cout]
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cultionet_amd import _lib

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
B, H, W = 8, 100, 100
Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
Cout = int(sys.argv[2]) if len(sys.argv) > 2 else 128
x = torch.randn(B, Cin, H, W, device=dev)
dy = torch.randn(B, Cout, H, W, device=dev)
dw = torch.zeros(Cout, Cin, 3, 3, device=dev)
ws = torch.empty(16 << 20, device=dev)
rng = (ctypes.c_int * 2)()
_lib.call("cn_stream_priority_range", rng)
h = ctypes.c_void_p()
_lib.call("cn_stream_create", int(rng[0]), None, 0, ctypes.byref(h))
side = torch.cuda.ExternalStream(h.value, device=dev)
main = torch.cuda.current_stream()


def wgrad(n):
    for _ in range(n):
        _lib.call("cn_conv2d_bwd_weight_f32", x.data_ptr(), Cin * H * W, dy.data_ptr(), Cout * H * W, dw.data_ptr(), B, Cin, H,
                  W, Cout, 3, 3, 1, 1, 1, ws.data_ptr(), ws.numel(), side.cuda_stream)


C = 128
src = torch.randn(B, C, 99, 99, device=dev)
dst = torch.empty(B, C, 100, 100, device=dev)
big = torch.randn(B, C, H, W, device=dev)
big2 = torch.empty_like(big)
small = torch.randn(B, C, 25, 25, device=dev)
small2 = torch.empty_like(small)
ms = main.cuda_stream
victims = {
    "bilinear_bwd 100->99 (x128 ch)": lambda: _lib.call("cn_bilinear_bwd_f32", dst.data_ptr(), C * 10000, src.data_ptr(), C * 9801,
                                                        B, C, 99, 99, 100, 100, 0, 0, 0, ms),
    "bilinear_fwd 99->100": lambda: _lib.call("cn_bilinear_fwd_f32", src.data_ptr(), C * 9801, dst.data_ptr(), C * 10000, B, C,
                                               99, 99, 100, 100, 0, 0, ms),
    "copy 41 MB": lambda: _lib.call("cn_copy_f32", big.data_ptr(), C * H * W, big2.data_ptr(), C * H * W, B, C * H * W, 0, ms),
    "copy 2.6 MB": lambda: _lib.call("cn_copy_f32", small.data_ptr(), C * 625, small2.data_ptr(), C * 625, B, C * 625, 0, ms),
    "fill 41 MB": lambda: _lib.call("cn_fill_f32", big2.data_ptr(), big2.numel(), 0.0, ms),
}


def timed(fn, k, corun):
    torch.cuda.synchronize()
    if corun:
        wgrad(corun)
        time.sleep(0.002)  # the side stream is well inside its first launch
    evs = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main)
        fn()
        b.record(main)
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return t[len(t) // 2], t[0], t[-1]


wgrad(3)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(side); wgrad(10); b.record(side); torch.cuda.synchronize()
wg_us = a.elapsed_time(b) * 100
print(f"side kernel: wgrad {Cin}->{Cout} 3x3 at 8x100^2: {wg_us:.0f} us per launch alone")
for name, fn in victims.items():
    for _ in range(3):
        fn()
    alone = timed(fn, 20, 0)
    co = timed(fn, 20, 30)
    print(f"{name:32s} alone {alone[0]:7.1f} us (min {alone[1]:.1f})   next to wgrad {co[0]:7.1f} us (min {co[1]:.1f} max {co[2]:.1f})  x{co[0] / alone[0]:.1f}")
a.record(side); wgrad(10); b.record(side)
for _ in range(200):
    victims["copy 2.6 MB"]()
torch.cuda.synchronize()
print(f"wgrad with 200 small copies beside it: {a.elapsed_time(b) * 100:.0f} us per launch")
