import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29618")
import torch
import torch.distributed as dist
from cultionet_amd import synthetic as S
from cultionet_amd.data import Data
from cultionet_amd.lightning import CultionetLitModel, HipTrainer

mode = sys.argv[1] if len(sys.argv) > 1 else "pg"
torch.cuda.set_device(0)
if mode in ("pg", "pg_used"):
    dist.init_process_group("nccl", rank=0, world_size=1)
    if mode == "pg_used":
        t = torch.ones(1024, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
x, y, bd = S.seeded_batch(32, seed=7)
batch = Data(x=x.cuda(), y=y.cuda(), bdist=bd.cuda())
lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=32, dropout=0.0)
m = lit.cultionet_model.mask_model
m.load_state_dict(S.seeded_state_dict(m.state_dict()))
lit = lit.to("cuda:0").train()
tr = HipTrainer(lit, precision="bf16-mixed")
for _ in range(5):
    tr.training_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    tr.training_step(batch)
torch.cuda.synchronize()
print(mode, "ms/step", (time.perf_counter() - t0) / 10 * 1e3)
if mode.startswith("pg"):
    dist.destroy_process_group()
