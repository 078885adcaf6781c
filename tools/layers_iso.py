"""Per-launch table of the contraction kernels of one training step, every kernel ALONE on the GPU (weight-gradient side
stream and branch streams off), aggregated by launch description -- the isolated counterpart of CN_PROF_DUMP +
tools/layerprof.py on a bench run.

    python tools/layers_iso.py [f32|bf16] [batch] [steps] [top]

Prints total ms/step, then per distinct launch: us/step, launches/step, us/launch, TFLOP/s, us/step lost against the MFMA
peak of the precision.
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cultionet_amd  # noqa: E402

cultionet_amd.configure_runtime()
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if dtype == "bf16" else 8)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
top = int(sys.argv[4]) if len(sys.argv) > 4 else 60
dump = tempfile.mktemp(suffix=".tsv")
os.environ["CN_PROF_DUMP"] = dump

import ctypes  # noqa: E402
from collections import defaultdict  # noqa: E402

import torch  # noqa: E402

from cultionet_amd import _lib, engine as E, synthetic as S  # noqa: E402
from cultionet_amd.data import Data  # noqa: E402
from cultionet_amd.lightning import CultionetLitModel, HipTrainer  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=32, dropout=0.0).to(dev).train()
tr = HipTrainer(lit, gradient_clip_val=1.0, precision="bf16-mixed" if dtype == "bf16" else "32-true")
x, y, bd = S.seeded_batch(B, height=100, width=100, seed=3, with_mask=True)
batch = Data(x=x.to(dev), y=y.to(dev), bdist=bd.to(dev))
E.overlap_wgrad(False)
with E.branch_streams(False):
    for _ in range(3):
        tr.training_step(batch)
    torch.cuda.synchronize()
    _lib.call("cn_profile_set_filter", None)
    _lib.call("cn_profile_begin")
    for _ in range(steps):
        tr.training_step(batch)
    torch.cuda.synchronize()
    _lib.call("cn_profile_end", (ctypes.c_double * 24)())
agg = defaultdict(lambda: [0.0, 0.0, 0])
for line in open(dump):
    kind, desc, us, fl = line.rstrip("\n").split("\t")
    a = agg[desc]
    a[0] += float(us); a[1] += float(fl); a[2] += 1
os.unlink(dump)
peak = 2500e6 if dtype == "bf16" else 157.3e6
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in agg.values())
flops = sum(v[1] for v in agg.values())
print(f"{dtype} batch {B}: contraction kernels alone {tot / steps / 1e3:.2f} ms/step, {flops / tot / 1e6:.1f} TFLOP/s "
      f"({flops / tot / peak:.3f} of peak) over {len(rows)} distinct launches")
for desc, (us, fl, n) in rows[:top]:
    tf = fl / us / 1e6 if us > 0 else 0.0
    print(f"{us / steps:9.1f} us/step  n={n / steps:5.1f}  {us / n:8.1f} us  {tf:7.1f} TF  lost {(us - fl / peak) / steps:8.1f} us/step  {desc}")
