"""Host-side profile of the eval forward (the sliding-window predictor's inner call): enqueue time per forward with an
idle queue and the cProfile top list.   python tools/hostprof_eval.py [bf16|f32] [batch]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cultionet_amd import synthetic as S
from cultionet_amd.lightning import CultionetLitModel

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
lit = CultionetLitModel(in_channels=4, in_time=25, hidden_channels=32, dropout=0.0)
m = lit.cultionet_model.mask_model
m.load_state_dict(S.seeded_state_dict(m.state_dict()))
lit = lit.to(dev).eval()
x = torch.rand(B, 4, 25, 110, 110, device=dev)
ctx = torch.autocast("cuda", dtype=torch.bfloat16, enabled=prec == "bf16")
with torch.no_grad(), ctx:
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    idle = []
    for _ in range(5):
        torch.cuda.synchronize()
        a = time.perf_counter()
        m(x)
        idle.append(time.perf_counter() - a)
        torch.cuda.synchronize()
        b = time.perf_counter()
    print(f"{prec} batch {B}: enqueue {min(idle) * 1e3:.2f} ms per forward")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        torch.cuda.synchronize()
        m(x)
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
