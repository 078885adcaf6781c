"""Where a data-parallel step (one-rank RCCL group) spends its time with / without the branch streams: HIP events around
forward+loss, backward (incl. the bucket launches) and the optimizer.  CN_KEEP_BRANCH_STREAMS=1 keeps spawn() active."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29617")
import torch
import torch.distributed as dist
from cultionet_amd import engine as E, synthetic as S
from cultionet_amd.data import Data
from cultionet_amd.ddp import GradientAllReduce
from cultionet_amd.lightning import CultionetLitModel, HipTrainer

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x, y, bd = S.seeded_batch(32, seed=7)
batch = Data(x=x.cuda(), y=y.cuda(), bdist=bd.cuda())
lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=32, dropout=0.0)
m = lit.cultionet_model.mask_model
m.load_state_dict(S.seeded_state_dict(m.state_dict()))
lit = lit.to("cuda:0").train()
comm = GradientAllReduce()
tr = HipTrainer(lit, precision="bf16-mixed", comm=comm)
orig = comm.backward
evs = []

def timed_backward(tape, store):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); t0 = time.perf_counter()
    orig(tape, store)
    t1 = time.perf_counter(); b.record()
    evs.append((a, b, t1 - t0))

comm.backward = timed_backward
for _ in range(5):
    tr.training_step(batch)
torch.cuda.synchronize(); evs.clear()
s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
s0.record(); t0 = time.perf_counter()
n = 10
for _ in range(n):
    tr.training_step(batch)
t1 = time.perf_counter(); s1.record(); torch.cuda.synchronize()
print("step ms (events)", s0.elapsed_time(s1) / n, "host ms", (t1 - t0) / n * 1e3)
print("backward ms (events on the compute stream)", sum(a.elapsed_time(b) for a, b, _ in evs) / len(evs),
      "host ms inside comm.backward", sum(h for _, _, h in evs) / len(evs) * 1e3)
if os.environ.get("CN_PROFILE_HOST") == "1":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        tr.training_step(batch)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
dist.destroy_process_group()
