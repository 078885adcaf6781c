"""Host-side (Python + ctypes + HIP runtime) profile of one native training step: enqueue time with an idle queue and
the cProfile top list.

    python tools/hostprof.py [bf16|f32] [batch]
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cultionet_amd import synthetic as O
from cultionet_amd.data import Data
from cultionet_amd.lightning import CultionetLitModel, HipTrainer

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if prec == "bf16" else 8)
dev = torch.device("cuda:0")
lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=32, dropout=0.0)
m = lit.cultionet_model.mask_model
m.load_state_dict(O.seeded_state_dict(m.state_dict()))
lit = lit.to(dev).train()
x, y, bdist = O.seeded_batch(B, seed=7)
batch = Data(x=x.to(dev), y=y.to(dev), bdist=bdist.to(dev), lon=torch.zeros(B, device=dev), lat=torch.zeros(B, device=dev))
tr = HipTrainer(lit, gradient_clip_val=1.0, precision="bf16-mixed" if prec == "bf16" else "32-true")
for _ in range(5):
    tr.training_step(batch)
torch.cuda.synchronize()
idle = []
for _ in range(5):
    torch.cuda.synchronize()
    a = time.perf_counter()
    tr.training_step(batch)
    idle.append(time.perf_counter() - a)
    torch.cuda.synchronize()
print(f"{prec} batch {B}: enqueue with an idle queue {min(idle) * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    torch.cuda.synchronize()
    tr.training_step(batch)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
