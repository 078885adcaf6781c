"""Which Python line issues the hipMemcpyAsync calls of a training step? (torch.profiler with stacks.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from cultionet_amd import synthetic as S
from cultionet_amd.data import Data
from cultionet_amd.lightning import CultionetLitModel, HipTrainer

dev = torch.device("cuda:0")
lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=32, dropout=0.0)
m = lit.cultionet_model.mask_model
m.load_state_dict(S.seeded_state_dict(m.state_dict()))
lit = lit.to(dev).train()
x, y, bd = S.seeded_batch(4)
b = Data(x=x.to(dev), y=y.to(dev), bdist=bd.to(dev))
tr = HipTrainer(lit, precision=sys.argv[1] if len(sys.argv) > 1 else "bf16-mixed")
for _ in range(3):
    tr.training_step(b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.training_step(b)
    torch.cuda.synchronize()
from collections import Counter
c = Counter()
for e in prof.events():
    n = e.name.lower()
    if "memcpy" in n or "memset" in n or "copy_" in n or "aten::" in n:
        st = [s for s in (e.stack or []) if "cultionet_amd" in s or "bench" in s]
        c[(e.name, st[0] if st else "?")] += 1
for k, v in c.most_common(40):
    print(v, k)
