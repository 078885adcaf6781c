"""Times the fused PreTimeReduction entry points (cn_pretime_fwd_f32 / cn_pretime_bwd_f32) on synthetic data:
python tools/pretime_bench.py [B] [C] [T] [H] [Cout] [kind]   (kind 0 = fp32 NCHW out, 1 = bf16 NHWC out)"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from cultionet_amd import _lib  # noqa: E402

B, C, T, H, Cout, kind = (int(v) for v in (sys.argv[1:] + ["32", "3", "12", "100", "32", "1"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
HW = H * H
x = torch.randn(B, C * T, H, H, device=dev)
k3, k5 = 3, 5
ps = []
for k in (3, 5):
    Tp = T - k + 1
    ps += [torch.randn(C, C, k, device=dev) * 0.3, torch.randn(Cout, C, Tp, device=dev) * 0.2, torch.ones(C, device=dev),
           torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(Cout, device=dev),
           torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)]
ps += [torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)]
params = (ctypes.c_void_p * 22)(*[t.data_ptr() for t in ps])
st = torch.empty(2 * (2 * C + 2 * Cout), device=dev)
offs, o = [], 0
for _ in range(2):
    for n in (C, C, Cout, Cout):
        offs.append(o)
        o += n
stats = (ctypes.c_void_p * 8)(*[st[i:].data_ptr() for i in offs])
bn = (ctypes.c_float * 4)(1e-5, 0.1, 1e-5, 0.1)
need = _lib.query("cn_pretime_workspace_floats", B, C, T, HW, Cout, 1)
only_fwd = need < 0
if only_fwd:
    need = _lib.query("cn_pretime_workspace_floats", B, C, T, HW, Cout, 0)
ws = torch.zeros(need, device=dev)
if kind == 0:
    y = torch.empty(B, Cout, H, H, device=dev); ys = Cout * HW
else:
    y = torch.empty(B, H, H, Cout, dtype=torch.bfloat16, device=dev); ys = Cout
dy = torch.randn_like(y)
gl = []
for k in (3, 5):
    Tp = T - k + 1
    gl += [torch.zeros(C, C, k, device=dev), torch.zeros(Cout, C, Tp, device=dev), torch.zeros(C, device=dev),
           torch.zeros(C, device=dev), torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev)]
gl += [torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev)]
grads = (ctypes.c_void_p * 14)(*[t.data_ptr() for t in gl])
s = torch.cuda.current_stream().cuda_stream


def fwd(tr):
    _lib.call("cn_pretime_fwd_f32", x.data_ptr(), C * T * HW, params, stats, y.data_ptr(), ys, kind, B, C, T, HW, Cout, tr,
              bn, 1e-5, ws.data_ptr(), ws.numel(), s)


def bwd():
    _lib.call("cn_pretime_bwd_f32", x.data_ptr(), C * T * HW, params, stats, dy.data_ptr(), ys, kind, grads, B, C, T, HW,
              Cout, 1, bn, 1e-5, ws.data_ptr(), ws.numel(), s)


for name, fn in (("train fwd", lambda: fwd(1)), ("eval fwd", lambda: fwd(0))) + (() if only_fwd else (("bwd", bwd),)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e6:.1f} us  (B={B} C={C} T={T} H={H} Cout={Cout} kind={kind})")

# per-pass kernel times (HIP events inside the library, one window over 10 training forwards + 10 backwards)
if not only_fwd:
    prof = (ctypes.c_double * 24)()
    _lib.call("cn_profile_set_filter", None)
    _lib.call("cn_profile_begin")
    for _ in range(10):
        fwd(1)
        bwd()
    torch.cuda.synchronize()
    _lib.call("cn_profile_end", prof)
    rows = []
    for r in range(16):
        nb = ctypes.create_string_buffer(96)
        o3 = (ctypes.c_double * 3)()
        nk = _lib.query("cn_profile_top", r, nb, 96, o3)
        if r >= nk:
            break
        rows.append((nb.value.decode(), o3[0] / max(o3[2], 1) * 1e3))
    print("   passes: " + "  ".join(f"{n.replace('cn_pretime_kernel', 'k')} {t:.1f}" for n, t in sorted(rows)))
