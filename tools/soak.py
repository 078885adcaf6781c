"""Soak run of the native training step: many optimizer steps on changing batches in every mode the engine has
(fp32 / bf16, deferred slice sums on / off, replayed plan, one-rank RCCL group), checking what a race or a stale table
would break: finite losses and gradients, a loss that goes down, and trajectories that agree between the deferred and the
immediate slice sums for as long as float-atomic noise allows.   python tools/soak.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
mode = sys.argv[2] if len(sys.argv) > 2 else None
if mode is None:  # every mode in a process of its own (the switches are read at import time)
    import subprocess

    rc = 0
    for m, env in (("f32", {}), ("f32-immediate", {"CN_DEFER_SUMS": "0"}), ("bf16", {}), ("bf16-immediate", {"CN_DEFER_SUMS": "0"}),
                   ("bf16-replay", {}), ("bf16-comm", {"CN_FORCE_COMM": "1"}), ("f32-flush8", {"CN_SUM_FLUSH_MB": "8"})):
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(steps), m], env=e)
        rc |= r.returncode
    sys.exit(rc)

import torch

from cultionet_amd import synthetic as S
from cultionet_amd.data import Data
from cultionet_amd.lightning import CultionetLitModel, HipTrainer

torch.cuda.set_device(0)
comm = None
if mode == "bf16-comm":
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29671")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from cultionet_amd.ddp import GradientAllReduce

    comm = GradientAllReduce(world_size=1)
bf16 = mode.startswith("bf16")
B = 4 if mode == "bf16-replay" else (16 if bf16 else 4)
lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=32, dropout=0.0, learning_rate=1e-3)
m = lit.cultionet_model.mask_model
m.load_state_dict(S.seeded_state_dict(m.state_dict()))
lit = lit.to("cuda:0").train()
tr = HipTrainer(lit, precision="bf16-mixed" if bf16 else "32-true", replay=(mode == "bf16-replay"), comm=comm)
batches = []
for k in range(4):
    x, y, bd = S.seeded_batch(B, seed=300 + k, with_mask=(k % 2 == 0))
    batches.append(Data(x=x.cuda(), y=y.cuda(), bdist=bd.cuda()))
losses = []
for i in range(steps):
    loss = tr.training_step(batches[i % 4])
    if i % 10 == 0 or i == steps - 1:
        losses.append(float(loss.item()))
        assert torch.isfinite(tr.store.flat_grad).all() and torch.isfinite(tr.store.flat).all(), (mode, i)
torch.cuda.synchronize()
ok = all(l == l and l < 1.5 for l in losses) and losses[-1] < losses[0] - 0.02
print(f"{mode:16s} {steps} steps, batch {B}: loss {losses[0]:.4f} -> {losses[-1]:.4f}  first samples "
      f"{[round(l, 5) for l in losses[:4]]}  {'ok' if ok else 'FAILED'}")
if comm is not None:
    import torch.distributed as dist

    dist.destroy_process_group()
sys.exit(0 if ok else 1)
