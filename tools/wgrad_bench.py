"""Times the fp32 weight gradient (cn_conv2d_bwd_weight_f32, contraction + slice sum) of a few 3 x 3 layer shapes:
python tools/wgrad_bench.py            (B Cin H W Cout rows from the list below, or one shape on the command line)"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from cultionet_amd import _lib  # noqa: E402

shapes = [(8, 128, 100, 100, 128), (8, 32, 100, 100, 32), (8, 160, 100, 100, 128), (8, 64, 50, 50, 64), (8, 128, 26, 26, 128)]
if len(sys.argv) > 5:
    shapes = [tuple(int(v) for v in sys.argv[1:6])]
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
for B, Cin, H, W, Cout in shapes:
    x = torch.randn(B, Cin, H, W, device=dev)
    dy = torch.randn(B, Cout, H, W, device=dev)
    nws = B * (Cout * (H * (W + 1) + 3) + Cin * (H * W + 3)) + (1 << 24)
    ws = torch.empty(nws, device=dev)
    dw = torch.zeros(Cout, Cin, 3, 3, device=dev)

    def run():
        _lib.call("cn_conv2d_bwd_weight_f32", x.data_ptr(), Cin * H * W, dy.data_ptr(), Cout * H * W, dw.data_ptr(), B, Cin,
                  H, W, Cout, 3, 3, 1, 1, 1, ws.data_ptr(), nws, s)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fl = 2.0 * B * H * W * Cin * Cout * 9
    dw.zero_()
    run()
    ref = torch.nn.grad.conv2d_weight(x, dw.shape, dy, padding=1)
    err = float((dw - ref).abs().max() / ref.abs().max())
    print(f"B={B} {Cin}->{Cout} {H}x{W}: {dt * 1e6:7.1f} us  {fl / dt / 1e12:6.1f} TFLOP/s ({fl / dt / 157.3e12:.3f} of the f32 peak)  rel err {err:.1e}")
