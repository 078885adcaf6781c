"""In-kernel timeline of the bf16 weight-gradient kernel (diagnostic build with -DCNBW_STAMP; see cn_bwgrad.hip).

    make -C cultionet_amd/csrc stamp
    CN_LIB_PATH=cultionet_amd/csrc/libcultionet_hip_stamp.so python tools/bwgrad_stamps.py 32 128 100 100 128 3
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cultionet_amd import _lib

B, Cin, H, W, Cout, k = (int(v) for v in sys.argv[1:7])
dev = torch.device("cuda:0")
BF = torch.bfloat16
p = k // 2
s = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, H, W, Cin, device=dev).to(BF)
dy = torch.randn(B, H, W, Cout, device=dev).to(BF)
nws = _lib.query("cn_bwgrad_workspace_floats", B, Cin, H, W, Cout, k, k, 1, p, 1, 0)
ws = torch.empty(nws, device=dev)
dw = torch.zeros(Cout, Cin, k, k, device=dev)
for _ in range(3):
    _lib.call("cn_conv2d_bwd_weight_bf16", x.data_ptr(), Cin, dy.data_ptr(), Cout, dw.data_ptr(), B, Cin, H, W, Cout, k, k,
              1, p, 1, ws.data_ptr(), nws, s)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 256)()
assert lib.cn_bwgrad_read_stamps(buf) == 0
st = [v for v in buf if v]
rel = [v - st[0] for v in st]
print("per tile: top, barrier 1, LDS stores, barrier 2, fetch issue, multiply")
i = 1
n = 0
while i + 7 <= len(rel) and n < 10:
    c = rel[i:i + 7]
    print(f"tile {n}: top {c[0]}  b1 +{c[1] - c[0]}  store +{c[2] - c[1]}  b2 +{c[3] - c[2]}  fetch +{c[4] - c[3]}  "
          f"mma +{c[5] - c[4]}  (next top +{c[6] - c[5]})  total {c[6] - c[0]}")
    i += 6
    n += 1
