"""Micro-benchmarks of the contraction kernels on the TowerUNet shapes (HIP events, median of 5)."""
import sys, ctypes
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from cultionet_amd import engine as E, _lib

dev = torch.device('cuda:0')
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts)//2]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wsb = torch.empty(64 << 20, dtype=torch.float32, device=dev)  # scratch for aligned copies
shapes = [(128,128,100,3),(480,128,100,3),(128,128,50,3),(576,128,50,3),(128,128,25,3),(640,128,25,3),(32,32,100,3),(64,64,50,3),
          (480,128,100,1),(128,384,100,1),(128,9,100,3),(256,128,13,1)]
for Cin, Cout, H, k in shapes:
    conv = nn.Conv2d(Cin, Cout, k, padding=k//2, bias=False).to(dev)
    store = E.ParamStore(conv)
    x = torch.randn(B, Cin, H, H, device=dev); dy = torch.randn(B, Cout, H, H, device=dev)
    y = torch.empty_like(dy); dx = torch.empty_like(x)
    fl = 2.0*B*H*H*Cin*Cout*k*k
    with E.using_store(store):
        pw = E.packed_conv(conv, True)
    s = E._stream()
    f_fwd = lambda: _lib.call("cn_conv2d_fwd_f32", x.data_ptr(), E.bstride(x), pw.fwd.data_ptr(), None, y.data_ptr(), E.bstride(y), B, Cin, H, H, Cout, k, k, 1, k//2, 1, 0, s)
    f_dg = lambda: _lib.call("cn_conv2d_bwd_data_f32", dy.data_ptr(), E.bstride(dy), pw.bwd.data_ptr(), dx.data_ptr(), E.bstride(dx), B, Cin, H, H, Cout, k, k, 1, k//2, 1, 0, s)
    dw = store.grad_of(conv.weight)
    f_wg = lambda: _lib.call("cn_conv2d_bwd_weight_f32", x.data_ptr(), E.bstride(x), dy.data_ptr(), E.bstride(dy), dw.data_ptr(), B, Cin, H, H, Cout, k, k, 1, k//2, 1, wsb.data_ptr(), wsb.numel(), s)
    t1, t2, t3 = timeit(f_fwd), timeit(f_dg), timeit(f_wg)
    print(f"B{B} {Cin:4d}->{Cout:4d} {H:3d}^2 k{k}: fwd {t1*1e3:8.1f}us {fl/t1/1e9:6.1f}TF | dgrad {t2*1e3:8.1f}us {fl/t2/1e9:6.1f}TF | wgrad {t3*1e3:8.1f}us {fl/t3/1e9:6.1f}TF", flush=True)
# conv transpose
for C, H in [(128,50),(128,25),(256,13),(64,50)]:
    conv = nn.ConvTranspose2d(C, C, 3, stride=2, padding=1).to(dev)
    store = E.ParamStore(conv)
    Ho = 2*H-1
    x = torch.randn(B, C, H, H, device=dev); dy = torch.randn(B, C, Ho, Ho, device=dev); y = torch.empty_like(dy); dx = torch.empty_like(x)
    fl = 2.0*B*H*H*C*C*9
    with E.using_store(store):
        pw = E.packed_convT(conv, True)
    s = E._stream()
    f_fwd = lambda: _lib.call("cn_conv_transpose2d_fwd_f32", x.data_ptr(), E.bstride(x), pw.fwd.data_ptr(), conv.bias.data_ptr(), y.data_ptr(), E.bstride(y), B, C, H, H, C, 3, 3, 2, 1, 0, s)
    f_dg = lambda: _lib.call("cn_conv_transpose2d_bwd_data_f32", dy.data_ptr(), E.bstride(dy), pw.bwd.data_ptr(), dx.data_ptr(), E.bstride(dx), B, C, H, H, C, 3, 3, 2, 1, 0, s)
    dw = store.grad_of(conv.weight)
    f_wg = lambda: _lib.call("cn_conv_transpose2d_bwd_weight_f32", x.data_ptr(), E.bstride(x), dy.data_ptr(), E.bstride(dy), dw.data_ptr(), B, C, H, H, C, 3, 3, 2, 1, wsb.data_ptr(), wsb.numel(), s)
    t1, t2, t3 = timeit(f_fwd), timeit(f_dg), timeit(f_wg)
    print(f"B{B} convT {C:4d} {H:3d}->{Ho}: fwd {t1*1e3:8.1f}us {fl/t1/1e9:6.1f}TF | dgrad {t2*1e3:8.1f}us {fl/t2/1e9:6.1f}TF | wgrad {t3*1e3:8.1f}us {fl/t3/1e9:6.1f}TF", flush=True)
# streaming kernels: BN fwd/bwd GB/s
for C, H in [(128,100),(128,50),(32,100)]:
    bn = nn.BatchNorm2d(C).to(dev); store = E.ParamStore(bn)
    x = torch.randn(B, C, H, H, device=dev)
    with E.using_store(store), E.recording(True) as tape:
        xv = E.Var(x, True)
        f = lambda: E.bn_act(xv, bn, 1)
        t1 = timeit(f)
        yv = E.bn_act(xv, bn, 1); yv.grad = torch.randn_like(x)
        # backward timing (re-run closure)
        node = tape.nodes[-1]
        def fb():
            yv.grad = g0; xv.grad = None; node()
        g0 = torch.randn_like(x)
        t2 = timeit(fb)
    nbytes = x.numel()*4
    print(f"BN+SiLU {C}x{H}^2: fwd {t1*1e3:7.1f}us ({3*nbytes/t1/1e6:6.0f} GB/s alg 2R+1W) bwd {t2*1e3:7.1f}us ({5*nbytes/t2/1e6:6.0f} GB/s alg 4R+1W)")
