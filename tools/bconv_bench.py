"""Micro-benchmark of one bf16 conv / wgrad shape through the C ABI (for rocprofv3 --pmc passes).

    python tools/bconv_bench.py conv 32 128 100 100 128 3 [iters]
    python tools/bconv_bench.py wgrad 32 128 100 100 128 3 [iters]
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
stats_on = os.environ.get("STATS", "1") == "1"
dev = torch.device("cuda:0")
BF = torch.bfloat16
T = k * k
p = k // 2
s = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, H, W, Cin, device=dev).to(BF)
dy = torch.randn(B, H, W, Cout, device=dev).to(BF)
w = torch.randn(Cout, Cin, k, k, device=dev) * (Cin * T) ** -0.5
wp = torch.empty(_lib.query("cn_bconv_packed_elems", T, Cin, Cout), dtype=BF, device=dev)
_lib.call("cn_pack_weights_bf16", w.data_ptr(), wp.data_ptr(), T, Cin, Cout, T, Cin * T, 1, s)
y = torch.empty(B, H, W, Cout, dtype=BF, device=dev)
rows = _lib.query("cn_conv2d_stats_rows_bf16", B, H, W, Cout, k, k, 1, p, 1)
stats = torch.empty(rows * 2 * Cout, device=dev)
nws = _lib.query("cn_bwgrad_workspace_floats", B, Cin, H, W, Cout, k, k, 1, p, 1, 0)
ws = torch.empty(nws, device=dev)
dw = torch.zeros(Cout, Cin, k, k, device=dev)


def run():
    if kind == "conv":
        _lib.call("cn_conv2d_fwd_bf16", x.data_ptr(), Cin, wp.data_ptr(), None, y.data_ptr(), Cout, 0, B, Cin, H, W, Cout,
                  k, k, 1, p, 1, 0, 0, stats.data_ptr() if stats_on else None, s)
    else:
        _lib.call("cn_conv2d_bwd_weight_bf16", x.data_ptr(), Cin, dy.data_ptr(), Cout, dw.data_ptr(), B, Cin, H, W, Cout,
                  k, k, 1, p, 1, ws.data_ptr(), nws, s)


for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
fl = 2.0 * B * H * W * Cout * Cin * T
print(f"{kind} B{B} {Cin}->{Cout} {H}x{W} k{k}: {dt * 1e6:.1f} us  {fl / dt / 1e12:.1f} TFLOP/s")
