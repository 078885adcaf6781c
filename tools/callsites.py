"""Which host call sites issue the copy / fill / channel-sum launches of one native training step (the launches that are
pure data movement): name, count per step and the four innermost Python frames.

    python tools/callsites.py [bf16-mixed|32-true] [batch]
"""
import collections, sys, traceback, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cultionet_amd import _lib, synthetic as S
from cultionet_amd.data import Data
from cultionet_amd.lightning import CultionetLitModel, HipTrainer
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16-mixed"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=32, dropout=0.0)
m = lit.cultionet_model.mask_model
m.load_state_dict(S.seeded_state_dict(m.state_dict()))
lit = lit.to("cuda:0").train()
tr = HipTrainer(lit, precision=prec)
x, y, bd = S.seeded_batch(B, height=100, width=100, seed=1)
b = Data(x=x.cuda(), y=y.cuda(), bdist=bd.cuda())
for _ in range(2): tr.training_step(b)
torch.cuda.synchronize()
orig = _lib.call
sites = collections.Counter()
def rec(name, *a):
    if name in ("cn_copy_f32", "cn_fill_f32", "cn_channel_sum_f32", "cn_copy_bf16", "cn_copy_strided_f32") or "copy" in name or "fill" in name:
        st = traceback.extract_stack(limit=7)[:-1]
        sites[(name, " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(st[-4:])))] += 1
    return orig(name, *a)
_lib.call = rec
import cultionet_amd.engine as E
tr.training_step(b)
_lib.call = orig
for (n, s), c in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(c, n, s)
