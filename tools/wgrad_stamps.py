"""In-kernel timeline of the fp32 weight-gradient kernel (diagnostic build with -DCNW_STAMP; see cn_wgrad.hip).

    make -C cultionet_amd/csrc stamp
    CN_LIB_PATH=cultionet_amd/csrc/libcultionet_hip_stamp.so python tools/wgrad_stamps.py 8 128 100 100 128 3
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cultionet_amd import _lib

B, Cin, H, W, Cout, k = (int(v) for v in sys.argv[1:7])
dev = torch.device("cuda:0")
p = k // 2
s = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, Cin, H, W, device=dev)
dy = torch.randn(B, Cout, H, W, device=dev)
nws = B * (Cout * (H * (W + 1) + 3) + Cin * (H * W + 3)) + (1 << 22)
ws = torch.empty(nws, device=dev)
dw = torch.zeros(Cout, Cin, k, k, device=dev)
for _ in range(3):
    _lib.call("cn_conv2d_bwd_weight_f32", x.data_ptr(), Cin * H * W, dy.data_ptr(), Cout * H * W, dw.data_ptr(), B, Cin, H,
              W, Cout, k, k, 1, p, 1, ws.data_ptr(), nws, s)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 256)()
assert lib.cn_wgrad_read_stamps(buf) == 0
st = [v for v in buf if v]
t0 = st[0]
rel = [v - t0 for v in st]
print("stamps (cycles from the end of the prologue); per chunk: top, barrier, DMA issued, [per row: masked-lead, groups], end")
print(rel[:1])
i = 1
n = 0
while i + 6 <= len(rel) and n < 8:
    c = rel[i:i + 6]
    print(f"chunk {n}: top {c[0]}  barrier +{c[1] - c[0]}  DMA issue +{c[2] - c[1]}  lead-masked +{c[3] - c[2]}  "
          f"groups +{c[4] - c[3]}  tail +{c[5] - c[4]}  total {c[5] - c[0]}")
    i += 6
    n += 1
print("last stamp", rel[-1], "n", len(rel))
