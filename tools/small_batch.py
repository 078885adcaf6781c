"""Native training step at the reference's default batch of 4 (scripts/args.yml:248-254): eager launches vs the recorded
launch plan (HipTrainer(replay=True)).   python tools/small_batch.py [bf16|f32] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cultionet_amd import synthetic as S
from cultionet_amd.data import Data
from cultionet_amd.lightning import CultionetLitModel, HipTrainer

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H = int(sys.argv[3]) if len(sys.argv) > 3 else 32
DROP = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
x, y, bd = S.seeded_batch(B, seed=7)
batch = Data(x=x.cuda(), y=y.cuda(), bdist=bd.cuda())
for replay in (False, True):
    lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=H, dropout=DROP)
    m = lit.cultionet_model.mask_model
    m.load_state_dict(S.seeded_state_dict(m.state_dict()))
    lit = lit.to("cuda:0").train()
    tr = HipTrainer(lit, precision="bf16-mixed" if prec == "bf16" else "32-true", replay=replay)
    for _ in range(6):
        tr.training_step(batch)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        loss = tr.training_step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{prec} batch {B} replay={replay}: {dt * 1e3:.2f} ms/step = {B / dt:.0f} chips/s, host enqueue "
          f"{(t1 - t0) / n * 1e3:.2f} ms/step, loss {float(loss.item()):.6f}"
          + (f", plan: {tr._plan.n_calls} C calls + {len(tr._plan.ops) - tr._plan.n_calls} stream ops, "
             f"{sum(t.numel() * t.element_size() for t in tr._plan.keep) / 2**30:.2f} GiB kept" if replay else ""))
    # one step at a time on an idle queue: host time to enqueue the step vs the time until the GPU has finished it
    hs, gs = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.training_step(batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hs.append(t1 - t0); gs.append(t2 - t0)
    hs.sort(); gs.sort()
    print(f"   single step, idle queue: host enqueue {hs[5] * 1e3:.2f} ms, until done {gs[5] * 1e3:.2f} ms")
