import sys, time, os
sys.path.insert(0,'.')
import torch
from oracle import towerunet_oracle as O
m = O.TowerUNet(3,12,hidden_channels=32); m.load_state_dict(O.seeded_state_dict(m.state_dict())); m.train()
opt = torch.optim.AdamW(m.parameters(), lr=0.01, weight_decay=1e-3, eps=1e-4, betas=(0.9,0.98))
x,y,b = O.seeded_batch(8, seed=7)
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
os.system("lscpu | egrep 'Model name|Socket|Core|Thread' ")
for nt in (4, 8, 16, 24):
    torch.set_num_threads(nt)
    ts=[]
    for i in range(2):
        t=time.perf_counter()
        opt.zero_grad(); l,_=O.calc_loss(m(x),y,b); l.backward(); torch.nn.utils.clip_grad_norm_(m.parameters(),1.0); opt.step()
        ts.append(time.perf_counter()-t)
    print(nt, "threads: step times", ts, "chips/s", 8/ts[-1], flush=True)
