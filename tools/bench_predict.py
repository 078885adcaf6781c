"""Eval-mode forward throughput on the large-tile shape of BASELINE configs[4] ([1,4,25,256,256], hidden 32):
pixels/s of CultionetLitModel.predict_step (no tape, BatchNorm running statistics)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cultionet_amd import synthetic as O
from cultionet_amd.data import Data
from cultionet_amd.lightning import CultionetLitModel

dev = torch.device("cuda:0")
B, C, T, H, W = 1, 4, 25, 256, 256
lit = CultionetLitModel(in_channels=C, in_time=T, hidden_channels=32, dropout=0.0)
m = lit.cultionet_model.mask_model
m.load_state_dict(O.seeded_state_dict(m.state_dict()))
lit = lit.to(dev).eval()
x, y, bdist = O.seeded_batch(B, channels=C, time=T, height=H, width=W, seed=11)
batch = Data(x=x.to(dev), y=y.to(dev), bdist=bdist.to(dev), lon=torch.zeros(B, device=dev), lat=torch.zeros(B, device=dev))
with torch.no_grad():
    for _ in range(3):
        lit.predict_step(batch)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        lit.predict_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
print(f"predict [1,4,25,256,256] hidden 32: {dt * 1e3:.2f} ms / tile, {B * H * W / dt / 1e6:.2f} Mpixel/s, "
      f"{425.8 / dt / 1e3:.1f} TFLOP/s")
