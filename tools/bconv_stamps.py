"""In-kernel timeline of the bf16 conv (diagnostic build with -DCNB_STAMP; see cn_bconv.hip).

    make -C cultionet_amd/csrc stamp     # builds libcultionet_hip_stamp.so
    CN_LIB_PATH=cultionet_amd/csrc/libcultionet_hip_stamp.so python tools/bconv_stamps.py 32 128 100 100 128 3
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
T, p = k * k, k // 2
s = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, H, W, Cin, device=dev).to(BF)
w = torch.randn(Cout, Cin, k, k, device=dev) * (Cin * T) ** -0.5
wp = torch.empty(_lib.query("cn_bconv_packed_elems", T, Cin, Cout), dtype=BF, device=dev)
_lib.call("cn_pack_weights_bf16", w.data_ptr(), wp.data_ptr(), T, Cin, Cout, T, Cin * T, 1, s)
y = torch.empty(B, H, W, Cout, dtype=BF, device=dev)
for _ in range(5):
    _lib.call("cn_conv2d_fwd_bf16", x.data_ptr(), Cin, wp.data_ptr(), None, y.data_ptr(), Cout, 0, B, Cin, H, W, Cout, k, k,
              1, p, 1, 0, 0, None, s)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 256)()
assert lib.cn_bconv_read_stamps(buf) == 0
for slot in range(2):
    st = list(buf[slot * 128:(slot + 1) * 128])
    t0 = st[0]
    print(f"--- block slot {slot}: start->prologue end {st[1] - t0}, loop start {st[2] - t0}, epilogue start {st[3] - t0}, "
          f"stores done {st[4] - t0}, end {st[5] - t0}")
    for ch in range(4):
        a = st[100 + ch * 4:104 + ch * 4]
        if a[0]:
            print(f"    chunk {ch}: stage begin {a[0] - t0}  barrier1 +{a[1] - a[0]}  ds_write +{a[2] - a[1]}  barrier2 +{a[3] - a[2]}")
    print("    epilogue groups start at:", [v - t0 for v in st[96:100] if v])
    steps = [v for v in st[8:96] if v]
    d = [steps[i + 1] - steps[i] for i in range(len(steps) - 1)]
    print("    half-steps (cycles):", d[:40])
