"""validation metrics (_shared_eval_step / validation_step / test_step) and the transfer model.

Metrics fixtures: the REAL reference's validation_step on seeded weights / inputs (oracle/make_golden.py
--metrics-only) with the torchmetrics scorers restated in oracle/metrics_ref.py (torchmetrics is not installed in
either image: "restatement-checked"). Tolerance: 1e-4 on every metric (fp32 path), as for the loss."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(g):
    from cultionet_amd.data import Data
    from oracle import towerunet_oracle as O
    from oracle.make_golden import calibrate_bn
    from oracle.selfcheck import build_pair

    hidden, B, H, W, with_mask, seed = (int(v) for v in g["meta"])
    lit, _ = build_pair(hidden=hidden, device="cuda:0")
    model = lit.cultionet_model.mask_model
    xc, _, _ = O.seeded_batch(B, height=H, width=W, seed=seed + 1000)
    calibrate_bn(model, lambda: model(xc.cuda()))
    x, y, bdist = O.seeded_batch(B, height=H, width=W, seed=seed, with_mask=bool(with_mask))
    batch = Data(x=x.cuda(), y=y.cuda(), bdist=bdist.cuda(), lon=torch.zeros(B).cuda(), lat=torch.zeros(B).cuda())
    return lit, batch


@pytest.mark.parametrize("name", ["val_h8_b2_28_masked", "val_h8_b2_28"])
def test_validation_step_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    lit, batch = _setup(g)
    lit.eval()
    met = lit.validation_step(batch)
    assert set(met) == {"vef1", "vcf1", "vmae", "val_score", "val_loss", "val_dloss", "val_eloss", "val_closs"}
    for k in met:
        assert abs(float(met[k]) - float(g[k])) <= 1e-4, (k, float(met[k]), float(g[k]))
    t = lit.test_step(batch)
    assert abs(float(t["test_score"]) - float(g["val_score"])) <= 1e-4
    assert abs(float(t["tef1"]) - float(g["vef1"])) <= 1e-4 and abs(float(t["tmae"]) - float(g["vmae"])) <= 1e-4


def test_eval_metrics_kernel_against_restatement():
    """The fused metrics kernel against oracle/metrics_ref.py on random maps, incl. masked pixels and the degenerate
    confusion matrices torchmetrics >= 1.0 special-cases: no positive prediction (eps-regularised ratio), every
    prediction right (1) and every prediction wrong (-1)."""
    from cultionet_amd import _lib
    from oracle import metrics_ref as M

    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    B, H, W = 3, 37, 41
    for case in range(5):
        dist, edge, crop = (torch.rand(B, 1, H, W, generator=gen) for _ in range(3))
        bdist = torch.rand(B, H, W, generator=gen)
        y = torch.randint(-1 if case != 1 else 0, 3, (B, H, W), generator=gen)
        if case == 2:
            edge = edge * 0.4  # no positive edge prediction at all
            y = torch.where(y == 2, torch.zeros_like(y), y)
        if case == 3:  # edge predictions all right, crop predictions all wrong
            edge = (y == 2).float().unsqueeze(1) * 0.8 + 0.1
            crop = 1.0 - ((y > 0) & (y < 2)).float().unsqueeze(1) * 0.8 - 0.1
        if case == 4:  # no true edge pixel and no predicted one: both marginals empty, all right -> 1
            y = torch.where(y == 2, torch.ones_like(y), y)
            edge = edge * 0.4
        loss = torch.tensor([0.625])
        valid = y != -1
        te, tc = (y == 2).long()[valid], ((y > 0) & (y < 2)).long()[valid]
        pe, pc = (edge[:, 0] > 0.5).long()[valid], (crop[:, 0] > 0.5).long()[valid]
        mae = M.MeanAbsoluteError()(dist[:, 0][valid], bdist[valid])
        mse = M.MeanSquaredError()(dist[:, 0][valid], bdist[valid])
        ef, cf = M.FBetaScore(beta=2.0)(pe, te), M.FBetaScore(beta=2.0)(pc, tc)
        em, cm = M.MatthewsCorrCoef()(pe, te), M.MatthewsCorrCoef()(pc, tc)
        score = loss[0] + (1 - ef) + (1 - cf) + mae + (1 - em.clamp_min(0)) + (1 - cm.clamp_min(0))
        want = torch.stack([mae, mse, ef, cf, em, cm, score]).float()
        counts = torch.empty(11, dtype=torch.float64, device=dev)
        out = torch.empty(7, device=dev)
        dd, ed, cd, bd, yd, ld = (t.to(dev).contiguous() for t in (dist, edge, crop, bdist, y, loss))
        _lib.call("cn_eval_metrics_f32", dd.data_ptr(), ed.data_ptr(), cd.data_ptr(), bd.data_ptr(), yd.data_ptr(), 2, 0.5,
                  y.numel(), ld.data_ptr(), counts.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert (out.cpu() - want).abs().max() <= 2e-6, (case, out.cpu(), want)


@pytest.mark.parametrize("finetune", [None, "fc", "all"])
def test_transfer_model(tmp_path, finetune):
    """CultionetLitTransferModel (lightning.py:686-818): loads the pretrained checkpoint, freezes / replaces the heads
    as the reference does, and trains only what is trainable through the drop-in path."""
    from cultionet_amd import synthetic as S
    from cultionet_amd.data import Data
    from cultionet_amd.lightning import CultionetLitModel, CultionetLitTransferModel

    kw = dict(in_channels=3, in_time=12, hidden_channels=8, dropout=0.0)
    base = CultionetLitModel(**kw)
    mm = base.cultionet_model.mask_model
    mm.load_state_dict(S.seeded_state_dict(mm.state_dict()))
    ckpt = tmp_path / "last.ckpt"
    torch.save({"state_dict": base.state_dict(), "hyper_parameters": dict(base.hparams)}, ckpt)
    lit = CultionetLitTransferModel(pretrained_ckpt_file=ckpt, finetune=finetune, **kw)
    assert lit.is_transfer_model and lit.model_attr == "cultionet_transfer_TowerUNet"
    # upstream registers the network under BOTH names (lightning.py:742-744 + :797-801): its transfer checkpoints carry
    # cultionet_model.* and cultionet_transfer_TowerUNet.* -- same key set here, and a strict round trip
    sd = lit.state_dict()
    a = {k[len("cultionet_model."):] for k in sd if k.startswith("cultionet_model.")}
    b = {k[len("cultionet_transfer_TowerUNet."):] for k in sd if k.startswith("cultionet_transfer_TowerUNet.")}
    assert a == b and len(a) == len(base.state_dict()) and len(sd) == 2 * len(a)
    assert len(list(lit.parameters())) == len(list(base.parameters()))  # the shared module is not counted twice
    res = lit.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = lit.cultionet_model.mask_model
    named = dict(lit.cultionet_model.named_parameters())
    heads = {n for n in named if n.startswith("mask_model.final_")}
    if finetune == "all":
        assert all(p.requires_grad for p in named.values())
    else:
        assert all(named[n].requires_grad for n in heads)
        assert not any(p.requires_grad for n, p in named.items() if n not in heads)
    base_sd = {k.replace("cultionet_TowerUNet.", ""): v for k, v in base.state_dict().items()}
    same_heads = all(torch.equal(named[n].detach().cpu(), base_sd[n]) for n in heads)
    assert same_heads == (finetune in ("fc", "all"))  # default mode re-initialises final_a/b/c + final_combine
    lit = lit.to("cuda:0").train()
    x, y, bdist = S.seeded_batch(2, height=28, width=28, seed=3, with_mask=True)
    batch = Data(x=x.cuda(), y=y.cuda(), bdist=bdist.cuda())
    params = [p for p in lit.cultionet_model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=0.01, weight_decay=1e-3, eps=1e-4, betas=(0.9, 0.98))
    before = {n: p.detach().clone() for n, p in lit.cultionet_model.named_parameters()}
    l0 = None
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss = lit.training_step(batch)
        loss.backward()
        opt.step()
        l0 = l0 if l0 is not None else float(loss)
    assert np.isfinite(float(loss)) and float(loss) != l0
    for n, p in lit.cultionet_model.named_parameters():
        changed = not torch.equal(p.detach(), before[n])
        assert changed == p.requires_grad or (p.requires_grad and p.numel() > 0), n
        if not p.requires_grad:
            assert not changed, n
