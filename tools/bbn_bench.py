"""Times the grouped bf16 BatchNorm backward (cn_bn_act_group_bwd_bf16): the statistics launch alone (no dx) and the
whole call. python tools/bbn_bench.py [P] [C] [G] [act]   (default: 320000 32 2 1 -- batch 32 x 100^2, one level)"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from cultionet_amd import _lib  # noqa: E402

P, C, G, act = (int(v) for v in (sys.argv[1:] + ["320000", "32", "2", "1"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
xs = [torch.randn(P, C, device=dev).to(torch.bfloat16) for _ in range(G)]
dys = [torch.randn(P, C, device=dev).to(torch.bfloat16) for _ in range(G)]
dxs = [torch.empty(P, C, device=dev, dtype=torch.bfloat16) for _ in range(G)]
means = [x.float().mean(0) for x in xs]
rstds = [1.0 / torch.sqrt(x.float().var(0, unbiased=False) + 1e-5) for x in xs]
gam = [torch.rand(C, device=dev) + 0.5 for _ in range(G)]
bet = [torch.randn(C, device=dev) * 0.1 for _ in range(G)]
dg = [torch.zeros(C, device=dev) for _ in range(G)]
db = [torch.zeros(C, device=dev) for _ in range(G)]
ws = torch.zeros(_lib.query("cn_bn_group_workspace_floats_bf16", G, C), device=dev)


def tab(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts])


s = torch.cuda.current_stream().cuda_stream
acc = (ctypes.c_int * G)(*([0] * G))


def run(with_dx):
    _lib.call("cn_bn_act_group_bwd_bf16", G, tab(xs), C, tab(dys), C, tab(means), tab(rstds), tab(gam), tab(bet),
              tab(dxs if with_dx else [None] * G), C, acc, tab(dg), tab(db), ws.data_ptr(), P, C, 1, act, s)


for name, wd in (("statistics launch alone", False), ("statistics + apply", True)):
    for _ in range(3):
        run(wd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        run(wd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    mb = G * P * C * 2 * (2 if not wd else 5) / 1e6
    print(f"{name}: {dt * 1e6:.1f} us  ({mb / dt / 1e6:.2f} TB/s algorithmic)  P={P} C={C} G={G} act={act}")
# reference check of dgamma / dbeta of branch 0 (one call on zeroed accumulators)
for t in dg + db:
    t.zero_()
run(False)
torch.cuda.synchronize()
x = xs[0].float(); dy = dys[0].float()
xh = (x - means[0]) * rstds[0]
v = gam[0] * xh + bet[0]
sg = torch.sigmoid(v)
dz = dy * (sg * (1 + v * (1 - sg))) if act else dy
print("dgamma rel err", float(((dz * xh).sum(0) - dg[0]).abs().max() / (dz * xh).sum(0).abs().max()),
      "dbeta rel err", float((dz.sum(0) - db[0]).abs().max() / dz.sum(0).abs().max()))
