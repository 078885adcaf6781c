"""The CPU oracle (oracle/towerunet_oracle.py) against the committed golden vectors.

The vectors in tests/golden were produced by the REAL reference (stub-imported
from /root/reference by oracle/make_golden.py). This pins the oracle without
needing /root/reference at run time. Tolerance: the oracle runs the same ATen
kernels in the same order as the reference, so outputs are expected to be equal
to ~1e-6 (exactly equal on the generating machine; thread-count dependent
summation order elsewhere).
"""
import os

import numpy as np
import pytest
import torch

from oracle import towerunet_oracle as O

TOL = 2e-6


def _run_train(g, **model_kw):
    hidden, B, H, W, with_mask, seed = (int(v) for v in g["meta"])
    torch.manual_seed(0)
    m = O.TowerUNet(3, 12, hidden_channels=hidden, **model_kw)
    m.load_state_dict(O.seeded_state_dict(m.state_dict()))
    m.train()
    x, y, bdist = O.seeded_batch(B, height=H, width=W, seed=seed, with_mask=bool(with_mask))
    return m, x, y, bdist


@pytest.mark.parametrize(
    "name,kw,loss_name",
    [
        ("train_h8_b2_28", {}, "TanimotoComplementLoss"),
        ("train_h8_b2_28_masked", {}, "TanimotoComplementLoss"),
        ("train_h8_b2_28_tanimoto", {}, "TanimotoDistLoss"),
        ("train_h8_b2_28_combined", {}, "TanimotoCombined"),
        ("train_h8_b2_28_noattn", {"attention_weights": None}, "TanimotoComplementLoss"),
        ("train_h8_b2_28_dil3", {"dilations": [1, 3]}, "TanimotoComplementLoss"),
        ("train_h32_b1_100_masked", {}, "TanimotoComplementLoss"),
        ("train_h8_b2_28_poolmax", {"pool_by_max": True}, "TanimotoComplementLoss"),
        ("train_h8_b2_28_res", {"res_block_type": "res", "attention_weights": None}, "TanimotoComplementLoss"),
        ("train_h8_b2_28_bnfirst", {"batchnorm_first": True}, "TanimotoComplementLoss"),
        ("train_h8_b2_28_sca", {"attention_weights": "spatial_channel"}, "TanimotoComplementLoss"),
    ],
)
def test_oracle_train_matches_reference_vectors(golden_dir, name, kw, loss_name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    m, x, y, bdist = _run_train(g, **kw)
    pred, stages = m(x, return_stages=True)
    loss, rep = O.calc_loss(pred, y, bdist, loss_name=loss_name)
    loss.backward()
    for k in ("distance", "edge", "crop"):
        assert np.abs(pred[k].detach().numpy() - g[k]).max() <= TOL, k
    assert abs(loss.item() - float(g["loss"])) <= TOL
    for k in ("dloss", "eloss", "closs"):
        assert abs(rep[k].item() - float(g[k])) <= TOL
    for key in g.files:
        if key.startswith("stage."):
            s = key[len("stage."):]
            s = {"pre_unet": "embeddings"}.get(s, s)
            assert np.abs(stages[s].detach().numpy() - g[key]).max() <= 1e-5, key
    names = list(g["grad_names"])
    norms = {n: float(p.grad.double().norm()) for n, p in m.named_parameters()}
    for n, ref in zip(names, g["grad_norms"]):
        assert abs(norms[n] - ref) <= 1e-5 * max(1.0, abs(ref)), n
    # element-level probes of the reference's gradients (first 64 elements + a fixed random projection per parameter)
    params = dict(m.named_parameters())
    for i, n in enumerate(names):
        first, proj = O.grad_probe(str(n), params[str(n)].grad)
        scale = max(1.0, float(np.abs(g["grad_probe_first"][i]).max()), float(g["grad_norms"][i]))
        assert np.abs(first.numpy() - g["grad_probe_first"][i]).max() <= 1e-5 * scale, n
        assert abs(proj - float(g["grad_probe_proj"][i])) <= 1e-5 * scale, n
    sd = m.state_dict()
    k0 = str(g["bn_key"]) if "bn_key" in g.files else "tower_fusion.tower_a.res_conv.res_modules.0.block.0.seq.1."
    assert np.abs(sd[k0 + "running_mean"].numpy() - g["bn_running_mean"]).max() <= TOL
    assert np.abs(sd[k0 + "running_var"].numpy() - g["bn_running_var"]).max() <= TOL


def test_oracle_eval_matches_reference_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "eval_h8_b2_28.npz"))
    hidden, B, C, T, H, W, seed = (int(v) for v in g["meta"])
    from oracle.make_golden import calibrate_bn

    m = O.TowerUNet(C, T, hidden_channels=hidden)
    m.load_state_dict(O.seeded_state_dict(m.state_dict()))
    xc, _, _ = O.seeded_batch(B, channels=C, time=T, height=H, width=W, seed=seed + 1000)
    calibrate_bn(m, lambda: m(xc))
    x, _, _ = O.seeded_batch(B, channels=C, time=T, height=H, width=W, seed=seed)
    with torch.no_grad():
        pred = m(x)
    for k in ("distance", "edge", "crop"):
        assert np.abs(pred[k].numpy() - g[f"{k}_crop"]).max() <= TOL


def test_state_dict_surface():
    m = O.TowerUNet(3, 12, hidden_channels=32)
    sd = m.state_dict()
    assert len(sd) == 442  # 442 tensors at the default config
    assert sum(p.numel() for p in m.parameters()) == 10581723
    assert "decoder.up_au.res_conv.attention_conv.2.qkv.weight" in sd
    assert "final_combine.final_edge.1.gamma" in sd


def test_loss_known_answers():
    """Known answers of /root/reference/tests/test_loss.py:109-145 (3 decimals)."""
    rng = np.random.default_rng(100)
    B, H, W = 2, 20, 20
    rng.uniform(low=-3, high=3, size=(B, 2, H, W))  # INPUTS_CROP_LOGIT (advances the stream)
    crop_prob = torch.from_numpy(rng.dirichlet((0.5, 0.5), size=(B * H * W))).float()
    crop_prob = crop_prob.reshape(B, H, W, 2).permute(0, 3, 1, 2)
    rng.random((B, 1, H, W))  # INPUTS_EDGE_PROB
    dist = torch.from_numpy(rng.random((B, 1, H, W))).float()
    targets = torch.from_numpy(rng.integers(low=0, high=2, size=(B, H, W))).long()
    rng.integers(low=0, high=1, size=(B, H, W))
    dist_t = torch.from_numpy(rng.random((B, H, W))).float()
    mask = torch.from_numpy(rng.integers(low=0, high=2, size=(B, 1, H, W))).long()

    r = lambda v: round(float(v), 3)
    assert r(O.tanimoto_dist_loss(crop_prob, targets)) == 0.611
    assert r(O.tanimoto_dist_loss(crop_prob, targets, mask)) == 0.431
    assert r(O.tanimoto_complement_loss(crop_prob, targets)) == 0.824
    assert r(O.tanimoto_complement_loss(crop_prob, targets, mask)) == 0.692
    assert r(O.tanimoto_combined_loss(crop_prob, targets)) == 0.717
    assert r(O.tanimoto_combined_loss(crop_prob, targets, mask)) == 0.561
    assert r(O.tanimoto_dist_loss(dist, dist_t, one_hot_targets=False)) == 0.417
    assert r(O.tanimoto_complement_loss(dist, dist_t, one_hot_targets=False)) == 0.704
