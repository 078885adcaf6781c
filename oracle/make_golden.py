"""Generate tests/golden/*.npz from the REAL reference (stub-imported) -- container only.

TEST INFRASTRUCTURE ONLY. Run from the repo root:

    TORCHDYNAMO_DISABLE=1 python -m oracle.make_golden          (--keys-only: TORCHDYNAMO_DISABLE=0, so that
                                                                 torch.compile renames pre_unet as upstream does)

Every vector is produced by /root/reference's own code (CultionetLitModel.forward,
calc_loss, autograd) on PyTorch-CPU fp32 with key-seeded weights
(oracle.towerunet_oracle.seeded_state_dict) and seeded inputs (seeded_batch), so
the GPU box can rebuild identical weights/inputs without a weight file. The
fixtures hold inputs' seeds and expected outputs only (data, no source).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import refimport
from . import towerunet_oracle as O

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _ref_model(ns, hidden, in_channels=3, in_time=12, **kw):
    m = ns.CultionetLitModel(in_channels=in_channels, in_time=in_time, hidden_channels=hidden, dropout=0.0, **kw)
    m.load_state_dict(O.seeded_state_dict(m.state_dict()))
    return m


def _train_case(ns, hidden, B, H, W, with_mask, seed=7, stages=False, loss_name="TanimotoComplementLoss",
                autocast=False, **kw):
    """autocast=True: the same step under torch.autocast("cpu", dtype=torch.bfloat16) -- what
    lightning.Trainer(precision="bf16-mixed") wraps around training_step (the reference's default is the fp16
    flavour "16-mixed", model.py:168-186; CPU autocast has no fp16, and bf16 is the MI355X-native choice)."""
    import contextlib

    m = _ref_model(ns, hidden, loss_name=loss_name, **kw)
    m.train()
    x, y, bdist = O.seeded_batch(B, height=H, width=W, seed=seed, with_mask=with_mask)
    batch = ns.Data(x=x, y=y, bdist=bdist, lon=torch.zeros(B), lat=torch.zeros(B))
    rec = {}
    hooks = []
    if stages:
        tu = m.cultionet_model.mask_model

        def grab(name):
            def fn(mod, inp, out):
                if isinstance(out, dict):
                    for k, v in out.items():
                        rec[f"stage.{k}"] = v.detach().numpy().copy()
                else:
                    rec[f"stage.{name}"] = out.detach().numpy().copy()
            return fn

        for name in ("pre_unet", "encoder", "decoder", "tower_fusion"):
            hooks.append(getattr(tu, name).register_forward_hook(grab(name)))
    ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast else contextlib.nullcontext()
    if autocast:
        # torch 2.10 CPU (oneDNN) corrupts the heap in the bf16 backward of nn.Conv3d(3, C, (10,1,1)) at 100x100
        # ("double free or corruption"; 50x50 and smaller are fine), so PreTimeReduction -- 0.1 % of the FLOPs -- is
        # evaluated in fp32 inside the autocast region. The HIP mixed-precision path makes the same choice.
        pre = m.cultionet_model.mask_model.pre_unet
        inner = pre.forward

        def fp32_forward(*a, **k):
            with torch.autocast("cpu", enabled=False):
                return inner(*a, **k)

        pre.forward = fp32_forward
    with ctx:
        pred = m(batch)
        loss, rep = m.calc_loss(batch, pred)
    loss.backward()
    for h in hooks:
        h.remove()
    out = dict(rec)
    for k in ("distance", "edge", "crop"):
        out[k] = pred[k].detach().float().numpy()
    out["loss"] = np.float64(loss.item())
    for k, v in rep.items():
        out[k] = np.float64(v.item())
    names, norms, firsts, projs, numels = [], [], [], [], []
    for n, p in m.named_parameters():
        names.append(n.replace("cultionet_TowerUNet.mask_model.", ""))
        norms.append(float(p.grad.double().norm()))
        first, proj = O.grad_probe(names[-1], p.grad)  # element-level probe (norms alone cannot see a swap)
        firsts.append(first.numpy())
        projs.append(proj)
        numels.append(p.numel())
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, dtype=np.float64)
    out["grad_probe_first"] = np.stack(firsts).astype(np.float32)
    out["grad_probe_proj"] = np.array(projs, dtype=np.float64)
    out["grad_numel"] = np.array(numels, dtype=np.int64)
    out["margin"] = np.float64(min(float((pred[k].float() - 0.5).abs().min()) for k in ("distance", "edge", "crop")))
    # running statistics after one train-mode forward (BN momentum path)
    sd = m.state_dict()
    k0 = next(k for k in sd if "tower_fusion.tower_a.res_conv" in k and k.endswith("running_mean"))[:-len("running_mean")]
    out["bn_key"] = np.array(k0.replace("cultionet_TowerUNet.mask_model.", ""))
    out["bn_running_mean"] = sd[k0 + "running_mean"].numpy().copy()
    out["bn_running_var"] = sd[k0 + "running_var"].numpy().copy()
    out["meta"] = np.array([hidden, B, H, W, int(with_mask), seed])
    return out


def calibrate_bn(model, forward):
    """Make the BatchNorm running statistics those of one train-mode pass (momentum 1.0), as a trained
    checkpoint's would be; with the raw key-seeded running stats eval-mode activations are not
    normalised and the forward is numerically chaotic (any two fp32 implementations diverge)."""
    bns = [m for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    old = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    model.train()
    with torch.no_grad():
        forward()
    for m, mo in zip(bns, old):
        m.momentum = mo
    model.eval()


def _eval_case(ns, hidden, B, C, T, H, W, seed=11, crop=64):
    m = _ref_model(ns, hidden, in_channels=C, in_time=T)
    xc, yc, bc = O.seeded_batch(B, channels=C, time=T, height=H, width=W, seed=seed + 1000)
    cal = ns.Data(x=xc, y=yc, bdist=bc, lon=torch.zeros(B), lat=torch.zeros(B))
    calibrate_bn(m, lambda: m(cal))
    x, y, bdist = O.seeded_batch(B, channels=C, time=T, height=H, width=W, seed=seed)
    batch = ns.Data(x=x, y=y, bdist=bdist, lon=torch.zeros(B), lat=torch.zeros(B))
    with torch.no_grad():
        pred = m(batch)
    out = {"meta": np.array([hidden, B, C, T, H, W, seed])}
    for k in ("distance", "edge", "crop"):
        p = pred[k]
        out[f"{k}_sum"] = np.float64(p.double().sum().item())
        out[f"{k}_crop"] = p[:, :, :crop, :crop].numpy().copy() if crop else p.numpy().copy()
        out[f"{k}_rowsum"] = p.double().sum(dim=(0, 1, 3)).numpy()
    return out


def main():
    assert refimport.available(), "/root/reference is required to generate fixtures"
    torch.set_float32_matmul_precision("highest")
    torch.set_num_threads(8)
    ns = refimport.import_reference()
    os.makedirs(OUT, exist_ok=True)

    def save(name, d):
        path = os.path.join(OUT, name)
        np.savez_compressed(path, **d)
        print(name, os.path.getsize(path) // 1024, "KiB", "loss" in d and d["loss"])

    if "--keys-only" in sys.argv:
        # checkpoint compatibility: the state-dict key set (with shapes) of the REAL reference's CultionetLitModel at
        # the default configuration, once as written by upstream (torch.compile wraps pre_unet, nunet.py:141 => keys
        # spelt pre_unet._orig_mod.*) -- tests/test_checkpoint_keys.py loads / writes both spellings against it
        m = ns.CultionetLitModel(in_channels=3, in_time=12, hidden_channels=32, dropout=0.0)
        sd = m.state_dict()
        save("state_dict_keys_h32.npz", {"keys": np.array(list(sd.keys())),
                                         "shapes": np.array([",".join(map(str, v.shape)) for v in sd.values()])})
        return
    if "--h64-only" in sys.argv:
        # the CLI default width (scripts/args.yml:220-226: hidden_channels 64)
        save("train_h64_b1_100.npz", _train_case(ns, 64, 1, 100, 100, True))
        return
    if "--metrics-only" in sys.argv:
        # validation metrics: the reference's own validation_step (eval mode, no_grad, as Lightning
        # calls it) with the torchmetrics scorers of oracle/metrics_ref.py (or a real torchmetrics when importable)
        for name, mask in (("val_h8_b2_28_masked", True), ("val_h8_b2_28", False)):
            m = _ref_model(ns, 8)
            xc, yc, bc = O.seeded_batch(2, height=28, width=28, seed=1011)
            cal = ns.Data(x=xc, y=yc, bdist=bc, lon=torch.zeros(2), lat=torch.zeros(2))
            calibrate_bn(m, lambda: m(cal))
            x, y, bdist = O.seeded_batch(2, height=28, width=28, seed=11, with_mask=mask)
            batch = ns.Data(x=x, y=y, bdist=bdist, lon=torch.zeros(2), lat=torch.zeros(2))
            with torch.no_grad():
                met = m.validation_step(batch)
            out = {k: np.float64(float(v)) for k, v in met.items()}
            out["meta"] = np.array([8, 2, 28, 28, int(mask), 11])
            save(name + ".npz", out)
        return
    if "--bf16-h64-only" in sys.argv:
        # the reference CLI's default operating point (model.py:52,56,86; scripts/args.yml:220-226,248-254): hidden 64,
        # batch 4, 16-mixed -- under CPU bf16 autocast, plus the fp32 twin of the same case
        args = (64, 4, 100, 100, True)
        a = _train_case(ns, *args, autocast=True)
        f = _train_case(ns, *args, autocast=False)
        for k in ("distance", "edge", "crop", "loss", "dloss", "eloss", "closs", "grad_norms"):
            a["fp32_" + k] = f[k]
        save("train_bf16_h64_b4_100.npz", a)
        return
    if "--bf16-only" in sys.argv:
        # mixed-precision fixtures (BASELINE configs[2]): the reference under CPU bf16 autocast, plus the fp32 run of
        # the SAME case so the tests can state the tolerance relative to the reference's own bf16 deviation
        for name, args, kw in (("h8_b2_28", (8, 2, 28, 28, True), {}),
                               ("h32_b1_100", (32, 1, 100, 100, False), {}),
                               ("h32_b4_100", (32, 4, 100, 100, True), {})):
            a = _train_case(ns, *args, autocast=True, **kw)
            f = _train_case(ns, *args, autocast=False, **kw)
            for k in ("distance", "edge", "crop", "loss", "dloss", "eloss", "closs", "grad_norms"):
                a["fp32_" + k] = f[k]
            save(f"train_bf16_{name}.npz", a)
        return
    if "--variants-only" in sys.argv:
        save("train_h8_b2_28_poolmax.npz", _train_case(ns, 8, 2, 28, 28, True, pool_by_max=True))
        save("train_h8_b2_28_res.npz", _train_case(ns, 8, 2, 28, 28, False, res_block_type="res", attention_weights=None))
        save("train_h8_b2_28_bnfirst.npz", _train_case(ns, 8, 2, 28, 28, False, batchnorm_first=True))
        return
    if "--sca-only" in sys.argv:
        save("train_h8_b2_28_sca.npz", _train_case(ns, 8, 2, 28, 28, True, attention_weights="spatial_channel"))
        return
    if "--eval-only" in sys.argv:
        save("eval_h32_b1_4x25x256.npz", _eval_case(ns, 32, 1, 4, 25, 256, 256))
        save("eval_h8_b2_28.npz", _eval_case(ns, 8, 2, 3, 12, 28, 28, crop=0))
        return
    # (i) small model with per-stage activations, train mode (odd sizes: 28->14->7->4)
    save("train_h8_b2_28.npz", _train_case(ns, 8, 2, 28, 28, False, stages=True))
    save("train_h8_b2_28_masked.npz", _train_case(ns, 8, 2, 28, 28, True, stages=False))
    # other selectable losses (args.yml:436-442)
    save("train_h8_b2_28_tanimoto.npz", _train_case(ns, 8, 2, 28, 28, True, loss_name="TanimotoDistLoss"))
    save("train_h8_b2_28_combined.npz", _train_case(ns, 8, 2, 28, 28, True, loss_name="TanimotoCombined"))
    # attention_weights=None variant (tests/test_train.py default)
    save("train_h8_b2_28_noattn.npz", _train_case(ns, 8, 2, 28, 28, False, attention_weights=None))
    # true dilated convs (dilations entry >= 3)
    save("train_h8_b2_28_dil3.npz", _train_case(ns, 8, 2, 28, 28, False, dilations=[1, 3]))
    # variant blocks: pool_by_max, res_block_type="res", batchnorm_first
    save("train_h8_b2_28_poolmax.npz", _train_case(ns, 8, 2, 28, 28, True, pool_by_max=True))
    save("train_h8_b2_28_res.npz", _train_case(ns, 8, 2, 28, 28, False, res_block_type="res", attention_weights=None))
    save("train_h8_b2_28_bnfirst.npz", _train_case(ns, 8, 2, 28, 28, False, batchnorm_first=True))
    # attention_weights="spatial_channel" (SpatialChannelAttention in the decoder's RESA blocks)
    save("train_h8_b2_28_sca.npz", _train_case(ns, 8, 2, 28, 28, True, attention_weights="spatial_channel"))
    # (ii) BASELINE configs[0] / configs[1] shapes at hidden 32
    save("train_h32_b1_100.npz", _train_case(ns, 32, 1, 100, 100, False))
    save("train_h32_b1_100_masked.npz", _train_case(ns, 32, 1, 100, 100, True))
    if "--no-big" not in sys.argv:
        save("train_h32_b8_100.npz", _train_case(ns, 32, 8, 100, 100, False))
        # (iv) configs[4]: large-tile eval forward
        save("eval_h32_b1_4x25x256.npz", _eval_case(ns, 32, 1, 4, 25, 256, 256))
    # eval-mode (running stats) small case, whole outputs
    save("eval_h8_b2_28.npz", _eval_case(ns, 8, 2, 3, 12, 28, 28, crop=0))


if __name__ == "__main__":
    main()
