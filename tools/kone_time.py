"""Time one conv forward (HIP events, median of 7): python tools/kone_time.py Cin Cout H"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from cultionet_amd import engine as E, _lib
dev = torch.device('cuda:0')
B, Cin, Cout, H, k = 8, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 3
conv = nn.Conv2d(Cin, Cout, k, padding=1, bias=False).to(dev)
store = E.ParamStore(conv)
x = torch.randn(B, Cin, H, H, device=dev); y = torch.empty(B, Cout, H, H, device=dev)
with E.using_store(store):
    pw = E.packed_conv(conv, True)
s = E._stream()
f = lambda: _lib.call("cn_conv2d_fwd_f32", x.data_ptr(), E.bstride(x), pw.fwd.data_ptr(), None, y.data_ptr(), E.bstride(y), B, Cin, H, H, Cout, k, k, 1, 1, 1, 0, s)
f(); torch.cuda.synchronize()
ts = []
for _ in range(7):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort()
fl = 2.0 * B * H * H * Cin * Cout * 9
print(f"{Cin}->{Cout} {H}^2: {ts[3]*1e3:7.1f} us  {fl/ts[3]/1e9:6.1f} TF")
