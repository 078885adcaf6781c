import torch, numpy as np, sys
sys.path.insert(0,'.')
import torch.nn.functional as F
from cultionet_amd import engine as E
g = torch.Generator().manual_seed(23)
x = torch.randn(1,16,49,49,generator=g)
y = F.interpolate(x, size=(50,50), mode="bilinear", align_corners=True)
with E.recording(False):
    yy = E.resize_bilinear(E.Var(x.cuda()), (50,50)).t.cpu()
d = (yy-y).abs()
print("max", d.max().item())
idx = np.unravel_index(d.argmax().item(), d.shape); print(idx)
# per-row / per-col error profile
print("by oy", d.amax(dim=(0,1,3))[:12], d.amax(dim=(0,1,3))[-12:])
print("by ox", d.amax(dim=(0,1,2))[:12], d.amax(dim=(0,1,2))[-12:])
