"""Run one conv shape repeatedly (for rocprofv3 --pmc)."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from cultionet_amd import engine as E, _lib
dev = torch.device('cuda:0')
B, Cin, Cout, H, k = 8, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 3
mode = sys.argv[4] if len(sys.argv) > 4 else "fwd"
conv = nn.Conv2d(Cin, Cout, k, padding=1, bias=False).to(dev)
store = E.ParamStore(conv)
x = torch.randn(B, Cin, H, H, device=dev); dy = torch.randn(B, Cout, H, H, device=dev); y = torch.empty_like(dy)
with E.using_store(store):
    pw = E.packed_conv(conv, True)
s = E._stream()
dw = store.grad_of(conv.weight)
for _ in range(6):
    if mode == "fwd":
        _lib.call("cn_conv2d_fwd_f32", x.data_ptr(), E.bstride(x), pw.fwd.data_ptr(), None, y.data_ptr(), E.bstride(y), B, Cin, H, H, Cout, k, k, 1, 1, 1, 0, s)
    else:
        _lib.call("cn_conv2d_bwd_weight_f32", x.data_ptr(), E.bstride(x), dy.data_ptr(), E.bstride(dy), dw.data_ptr(), B, Cin, H, H, Cout, k, k, 1, 1, 1, None, 0, s)
torch.cuda.synchronize()
