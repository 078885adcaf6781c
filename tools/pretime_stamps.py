"""s_memtime stamps of block 0 / wave 0 of the fused PreTimeReduction output pass (diagnostic build):
make -C cultionet_amd/csrc ptstamp && CN_LIB_PATH=cultionet_amd/csrc/libcultionet_hip_ptstamp.so python tools/pretime_stamps.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CN_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cultionet_amd",
                                                   "csrc", "libcultionet_hip_ptstamp.so"))
import torch  # noqa: E402

from cultionet_amd import _lib  # noqa: E402

B, C, T, H, Cout = 32, 3, 12, 100, 32
dev = torch.device("cuda:0")
HW = H * H
x = torch.randn(B, C * T, H, H, device=dev)
ps = []
for k in (3, 5):
    Tp = T - k + 1
    ps += [torch.randn(C, C, k, device=dev) * 0.3, torch.randn(Cout, C, Tp, device=dev) * 0.2, torch.ones(C, device=dev),
           torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(Cout, device=dev),
           torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)]
ps += [torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)]
params = (ctypes.c_void_p * 22)(*[t.data_ptr() for t in ps])
stats = (ctypes.c_void_p * 8)(*([None] * 8))
bn = (ctypes.c_float * 4)(1e-5, 0.1, 1e-5, 0.1)
need = _lib.query("cn_pretime_workspace_floats", B, C, T, HW, Cout, 0)
ws = torch.zeros(need, device=dev)
y = torch.empty(B, H, H, Cout, dtype=torch.bfloat16, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    _lib.call("cn_pretime_fwd_f32", x.data_ptr(), C * T * HW, params, stats, y.data_ptr(), Cout, 1, B, C, T, HW, Cout, 0, bn,
              1e-5, ws.data_ptr(), ws.numel(), s)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 32)()
lib = _lib.load()
lib.cn_pretime_read_stamps.argtypes = [ctypes.c_void_p]
assert lib.cn_pretime_read_stamps(out) == 0
v = list(out)
print("stamps (100 MHz ticks -> us = /100): start", v[0])
print("image copy      ", (v[1] - v[0]) / 100.0, "us")
i = 2
while i + 2 < 26 and v[i + 2] > 0:
    print(f"tile: stage x {(v[i + 1] - v[i]) / 100.0:.2f} us, convs {(v[i + 2] - v[i + 1]) / 100.0:.2f} us", end="")
    nxt = v[i + 3] if (i + 3 < 26 and v[i + 3] > 0) else v[30]
    print(f", epilogue {(nxt - v[i + 2]) / 100.0:.2f} us")
    i += 3
print("total", (v[30] - v[0]) / 100.0, "us")
