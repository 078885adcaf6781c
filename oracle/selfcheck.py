"""GPU self-check used by __graft_entry__.smoke() and tests/: one tiny TowerUNet training step on the HIP path,
compared with the CPU oracle. TEST INFRASTRUCTURE (lives in oracle/, never imported by cultionet_amd)."""
from __future__ import annotations

import torch


def build_pair(hidden: int = 8, in_channels: int = 3, in_time: int = 12, device: str = "cuda:0", **kw):
    """(HIP LitModel on `device`, CPU oracle TowerUNet) holding identical key-seeded weights."""
    from . import towerunet_oracle as O

    from cultionet_amd.lightning import CultionetLitModel

    lit = CultionetLitModel(in_channels=in_channels, in_time=in_time, hidden_channels=hidden, dropout=0.0, **kw)
    okw = {k: v for k, v in kw.items() if k in ("attention_weights", "dilations", "pool_by_max", "res_block_type", "batchnorm_first")}
    ref = O.TowerUNet(in_channels, in_time, hidden_channels=hidden, **okw)
    sd = O.seeded_state_dict(ref.state_dict())
    ref.load_state_dict(sd)
    lit.cultionet_model.mask_model.load_state_dict(sd)
    lit = lit.to(device)
    return lit, ref


def smoke_check(device: str = "cuda:0", hidden: int = 8, B: int = 2, H: int = 28, W: int = 28, tol: float = 1e-4):
    from . import towerunet_oracle as O

    from cultionet_amd.data import Data
    from cultionet_amd.lightning import HipTrainer

    lit, ref = build_pair(hidden=hidden, device=device)
    lit.train()
    ref.train()
    x, y, bdist = O.seeded_batch(B, height=H, width=W, seed=7, with_mask=True)
    pred = ref(x)
    loss_ref, _ = O.calc_loss(pred, y, bdist)
    loss_ref.backward()

    batch = Data(x=x.to(device), y=y.to(device), bdist=bdist.to(device))
    trainer = HipTrainer(lit)
    loss = trainer.forward_backward(batch)
    torch.cuda.synchronize()
    dl = abs(float(loss.item()) - float(loss_ref.item()))
    assert dl <= tol, f"loss mismatch: hip {float(loss.item()):.7f} vs oracle {float(loss_ref.item()):.7f}"
    store = trainer.store
    model = lit.cultionet_model.mask_model
    # gradient error RELATIVE to each parameter's own gradient norm (gradients here are of order 1e-4..1e-2: an absolute
    # bound would pass a wrong one); the floor -- 1e-3 of the largest per-parameter norm -- only keeps parameters whose
    # true gradient is rounding noise (e.g. a bias in front of a BatchNorm) from dividing by ~0. Same 2e-3 as -m gpu.
    pairs = [(n, store.grad_of(p).cpu().double(), pr.grad.double())
             for (n, p), (_, pr) in zip(model.named_parameters(), ref.named_parameters())]
    floor = 1e-3 * max(float(gr.norm()) for _, _, gr in pairs)
    worst, worst_name = 0.0, ""
    for n, g, gr in pairs:
        rel = float((g - gr).norm()) / max(float(gr.norm()), floor)
        if rel > worst:
            worst, worst_name = rel, n
    assert worst <= 2e-3, f"gradient mismatch {worst:.3e} (relative to the parameter's gradient norm) at {worst_name}"
    trainer.optimizer_step()
    torch.cuda.synchronize()
    print(f"smoke ok: loss {float(loss.item()):.6f} (|d|={dl:.2e}), worst rel grad err {worst:.2e}")
    return dl, worst
