"""Prototype: capture one native training step in a HIP graph and compare replay time with eager launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cultionet_amd import synthetic as O
from cultionet_amd.data import Data
from cultionet_amd.lightning import CultionetLitModel, HipTrainer

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = sys.argv[2] if len(sys.argv) > 2 else "32-true"
lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=32, dropout=0.0)
m = lit.cultionet_model.mask_model
m.load_state_dict(O.seeded_state_dict(m.state_dict()))
lit = lit.to(dev).train()
x, y, bdist = O.seeded_batch(B, seed=7)
batch = Data(x=x.to(dev), y=y.to(dev), bdist=bdist.to(dev), lon=torch.zeros(B, device=dev), lat=torch.zeros(B, device=dev))
tr = HipTrainer(lit, gradient_clip_val=1.0, precision=prec)
for _ in range(5):
    loss = tr.training_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    tr.training_step(batch)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 10
print(f"eager {eager * 1e3:.2f} ms/step  loss {float(loss):.6f}", flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    tr.training_step(batch)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        gl = tr.training_step(batch)
torch.cuda.synchronize()
print("captured", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
rep = (time.perf_counter() - t0) / 10
print(f"graph replay {rep * 1e3:.2f} ms/step  loss {float(gl):.6f}")
