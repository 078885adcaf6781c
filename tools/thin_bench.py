"""Times the direct 128 -> 3 x 3 head convolutions (cn_thin_conv3x3_{fwd,bwd_data,bwd_weight}_f32) on the BASELINE head
shape: python tools/thin_bench.py [B] [Cin] [H]"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from cultionet_amd import _lib  # noqa: E402

B, Cin, H = (int(v) for v in (sys.argv[1:] + ["8", "128", "100"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
HW = H * H
x = torch.randn(B, Cin, H, H, device=dev)
ws = [torch.randn(3, Cin, 3, 3, device=dev) * 0.05 for _ in range(3)]
dws = [torch.zeros_like(w) for w in ws]
y = torch.empty(B, 9, H, H, device=dev)
dy = torch.randn(B, 9, H, H, device=dev)
dx = torch.empty_like(x)
wpk = torch.empty(Cin * 84, device=dev)
wtab = (ctypes.c_void_p * 3)(*[w.data_ptr() for w in ws])
dwtab = (ctypes.c_void_p * 3)(*[w.data_ptr() for w in dws])
s = torch.cuda.current_stream().cuda_stream


def fwd():
    _lib.call("cn_thin_conv3x3_fwd_f32", x.data_ptr(), Cin * HW, wtab, None, y.data_ptr(), 9 * HW, B, Cin, H, H, 3, 3, 0, 1,
              wpk.data_ptr(), s)


def bwd_data():
    _lib.call("cn_thin_conv3x3_bwd_data_f32", dy.data_ptr(), 9 * HW, wtab, dx.data_ptr(), Cin * HW, B, Cin, H, H, 3, 3, 0, 1, 0,
              wpk.data_ptr(), s)


def bwd_weight():
    _lib.call("cn_thin_conv3x3_bwd_weight_f32", x.data_ptr(), Cin * HW, dy.data_ptr(), 9 * HW, dwtab, B, Cin, H, H, 3, 3, 0, 1, s)


ref = torch.cat([torch.nn.functional.conv2d(x, w, padding=1) for w in ws], dim=1)
fwd()
torch.cuda.synchronize()
print("fwd max |err|", float((y - ref).abs().max()))
for name, fn in (("fwd", fwd), ("bwd_data", bwd_data), ("bwd_weight", bwd_weight)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e6:.1f} us (incl. the pack launch)  B={B} Cin={Cin} H={H}")
