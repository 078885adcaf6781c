import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cultionet_amd import synthetic as S
from cultionet_amd.data import Data
from cultionet_amd.lightning import CultionetLitModel, HipTrainer
B = 32
x, y, bd = S.seeded_batch(B, seed=7)
batch = Data(x=x.cuda(), y=y.cuda(), bdist=bd.cuda())
lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=32, dropout=0.0)
m = lit.cultionet_model.mask_model
m.load_state_dict(S.seeded_state_dict(m.state_dict()))
lit = lit.to("cuda:0").train()
tr = HipTrainer(lit, precision="bf16-mixed")
for _ in range(4):
    tr.training_step(batch)
torch.cuda.synchronize()
print("HEAD_STREAMS", os.environ.get("CN_HEAD_STREAMS", "1"), "peak allocated GiB", round(torch.cuda.max_memory_allocated() / 2**30, 2), "reserved GiB", round(torch.cuda.memory_reserved() / 2**30, 2))
