"""Sliding-window scene prediction under a profiler: `rocprofv3 --kernel-trace --stats -d D -o p -- python3 tools/predict_prof.py [bf16|f32] [reps]`
then `python3 tools/prof_db.py D/p_results.db 40`. Prints wall time per scene and host enqueue time per scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cultionet_amd import synthetic as S
from cultionet_amd.lightning import CultionetLitModel
from cultionet_amd.predict import SlidingWindowPredictor

prec = "bf16-mixed" if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else "32-true"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rp = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
dev = torch.device("cuda:0")
lit = CultionetLitModel(in_channels=4, in_time=25, hidden_channels=32, dropout=0.0)
m = lit.cultionet_model.mask_model
m.load_state_dict(S.seeded_state_dict(m.state_dict()))
lit = lit.to(dev).eval()
HS = 600
scene = (torch.rand(4, 25, HS, HS, generator=torch.Generator().manual_seed(11)) * 10000.0).to(torch.int16).to(dev)
sp = SlidingWindowPredictor(lit, window_size=100, padding=5, batch_size=bs, precision=prec, replay=rp)
for _ in range(2):
    sp.predict_scene(scene)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    sp.predict_scene(scene)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"peak allocated {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB")
print(f"{prec} batch {bs} replay {rp}: {(t2 - t0) / reps * 1e3:.2f} ms per scene ({HS * HS / ((t2 - t0) / reps) / 1e6:.1f} Mpx/s), "
      f"host enqueue {(t1 - t0) / reps * 1e3:.2f} ms per scene")
if os.environ.get("CN_PROF_DUMP"):  # per-launch table of the contraction kernels: python tools/layerprof.py $CN_PROF_DUMP 3 bf16
    import ctypes
    from cultionet_amd import _lib
    _lib.call("cn_profile_set_filter", None)
    _lib.call("cn_profile_begin")
    for _ in range(3):
        sp.predict_scene(scene)
    torch.cuda.synchronize()
    _lib.call("cn_profile_end", (ctypes.c_double * 24)())
