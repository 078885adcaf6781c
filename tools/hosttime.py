"""Host (Python + ctypes + HIP runtime) enqueue time of one native training step, against its GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cultionet_amd import synthetic as O
from cultionet_amd.data import Data
from cultionet_amd.lightning import CultionetLitModel, HipTrainer

dev = torch.device("cuda:0")
B = 8
lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=32, dropout=0.0)
m = lit.cultionet_model.mask_model
m.load_state_dict(O.seeded_state_dict(m.state_dict()))
lit = lit.to(dev).train()
x, y, bdist = O.seeded_batch(B, seed=7)
batch = Data(x=x.to(dev), y=y.to(dev), bdist=bdist.to(dev), lon=torch.zeros(B, device=dev), lat=torch.zeros(B, device=dev))
tr = HipTrainer(lit, gradient_clip_val=1.0)
for _ in range(5):
    tr.training_step(batch)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter()
    tr.training_step(batch)
    host.append(time.perf_counter() - a)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 10
# enqueue-only time: the queue is never empty while the GPU is the bottleneck, so measure with an idle GPU per step
idle = []
for _ in range(5):
    torch.cuda.synchronize()
    a = time.perf_counter()
    tr.training_step(batch)
    idle.append(time.perf_counter() - a)
    torch.cuda.synchronize()
print(f"wall {wall * 1e3:.2f} ms/step; host call returns after {sum(host) / len(host) * 1e3:.2f} ms (back-pressured), "
      f"{min(idle) * 1e3:.2f} ms with an idle queue")
