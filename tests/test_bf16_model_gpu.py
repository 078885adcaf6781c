"""End-to-end parity of the bf16 mixed-precision path (BASELINE configs[2]) against fixtures from the REAL reference.

tests/golden/train_bf16_*.npz (oracle/make_golden.py --bf16-only) hold, for each case, the reference's training step
under torch.autocast(bfloat16) -- what lightning's precision="bf16-mixed" wraps around training_step (the reference
default is the fp16 flavour "16-mixed", model.py:168-186) -- AND its plain fp32 step on the same seeded weights / inputs.
bf16 carries 8 significant bits, so two correct mixed-precision implementations differ from fp32 (and from each other)
at the 1e-2 level on individual pixels; the reference's own bf16 run deviates from its fp32 run by max 0.013-0.055 /
mean 0.002-0.0045 on the probability maps and <= 6e-5 on the loss. Tolerances (stated against the fp32 reference, the
ground truth both approximate):
    probability maps   mean |d| <= 6e-3,  max |d| <= 8e-2   (and no worse than 1.5x the reference's own bf16 deviation)
    loss               |d| <= 5e-4
    gradient norms     median relative deviation <= 1e-2, 90th percentile <= 6e-2
The HIP path keeps BatchNorm/LayerNorm statistics, the time reduction, the heads, the loss and every accumulation in
fp32, so it is expected to sit INSIDE the reference's own bf16 deviation.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KEYS = ("distance", "edge", "crop")


def _setup(g):
    from cultionet_amd.data import Data
    from oracle import towerunet_oracle as O
    from oracle.selfcheck import build_pair

    hidden, B, H, W, with_mask, seed = (int(v) for v in g["meta"])
    lit, ref = build_pair(hidden=hidden, device="cuda:0")
    x, y, bdist = O.seeded_batch(B, height=H, width=W, seed=seed, with_mask=bool(with_mask))
    batch = Data(x=x.cuda(), y=y.cuda(), bdist=bdist.cuda(), lon=torch.zeros(B).cuda(), lat=torch.zeros(B).cuda())
    return lit, batch


# train_bf16_h64_b4_100: the reference CLI's DEFAULT operating point -- hidden 64, batch 4, 16-mixed (model.py:52,56,86;
# scripts/args.yml:220-226,248-254) -- from the real reference (oracle/make_golden.py --bf16-h64-only)
@pytest.mark.parametrize("name", ["train_bf16_h8_b2_28", "train_bf16_h32_b1_100", "train_bf16_h32_b4_100",
                                  "train_bf16_h64_b4_100"])
def test_bf16_train_step_matches_reference(golden_dir, name):
    from cultionet_amd import engine as E
    from cultionet_amd.lightning import HipTrainer

    g = np.load(os.path.join(golden_dir, name + ".npz"))
    lit, batch = _setup(g)
    lit.train()
    trainer = HipTrainer(lit, precision="bf16-mixed")
    # forward alone first (outputs), then the full step
    model = lit.cultionet_model.mask_model
    store = model.param_store()
    with E.using_store(store), E.recording(False), E.mixed_precision(True):
        outs = model.forward_vars(model.input_var(batch.x))
    # (train-mode forward updated the running statistics once; the step below does so again -- irrelevant here)
    for k in KEYS:
        p = outs[k].t.float().cpu().numpy()
        d32 = np.abs(p - g["fp32_" + k])
        dref = np.abs(g[k] - g["fp32_" + k])
        assert d32.mean() <= 6e-3 and d32.max() <= 8e-2, (k, d32.mean(), d32.max())
        assert d32.mean() <= 1.5 * dref.mean() + 1e-4, (k, d32.mean(), dref.mean())
        assert np.abs(p - g[k]).max() <= 0.12, k  # against the reference's own bf16 run (two different roundings)
        # > 0.5 masks (only asserted on the fp32 path). bf16 cannot promise identity at the
        # threshold itself -- random-init maps hover around 0.5 -- so: identical wherever the fp32 probability is farther
        # from 0.5 than the bf16 tolerance, and overall agreement no worse than the REFERENCE's own bf16 run minus 1 %
        f32 = g["fp32_" + k]
        clear = np.abs(f32 - 0.5) > 8e-2
        assert np.array_equal((p > 0.5)[clear], (f32 > 0.5)[clear]), k
        agree = float(((p > 0.5) == (f32 > 0.5)).mean())
        agree_ref = float(((g[k] > 0.5) == (f32 > 0.5)).mean())
        assert agree >= agree_ref - 0.01, (k, agree, agree_ref)
    loss = trainer.forward_backward(batch)
    torch.cuda.synchronize()
    assert abs(float(loss.item()) - float(g["fp32_loss"])) <= 5e-4, (float(loss.item()), float(g["fp32_loss"]))
    norms = {n: float(trainer.store.grad_of(p).double().norm()) for n, p in model.named_parameters()}
    rel = np.array([abs(norms[str(n)] - r) / max(abs(r), 1e-4) for n, r in zip(g["grad_names"], g["fp32_grad_norms"])])
    relref = np.abs(g["grad_norms"] - g["fp32_grad_norms"]) / np.maximum(np.abs(g["fp32_grad_norms"]), 1e-4)
    assert np.median(rel) <= 1e-2 and np.percentile(rel, 90) <= 6e-2, (np.median(rel), np.percentile(rel, 90))
    assert np.median(rel) <= 1.5 * np.median(relref) + 1e-3, (np.median(rel), np.median(relref))
    trainer.optimizer_step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_bf16_matches_own_fp32_path_and_trains():
    """The bf16 and fp32 HIP paths on the same weights / batch: losses agree to 5e-4 for several optimizer steps, and
    the loss goes down (mixed precision trains, not just runs)."""
    from cultionet_amd.data import Data
    from cultionet_amd.lightning import HipTrainer
    from oracle import towerunet_oracle as O
    from oracle.selfcheck import build_pair

    x, y, bdist = O.seeded_batch(4, height=50, width=50, seed=3, with_mask=True)
    batch = Data(x=x.cuda(), y=y.cuda(), bdist=bdist.cuda())
    losses = {}
    for prec in ("32-true", "bf16-mixed"):
        lit, _ = build_pair(hidden=16, device="cuda:0")
        lit.train()
        tr = HipTrainer(lit, precision=prec)
        losses[prec] = [float(tr.training_step(batch).item()) for _ in range(6)]
    a, b = np.array(losses["32-true"]), np.array(losses["bf16-mixed"])
    assert np.abs(a[0] - b[0]) <= 5e-4, (a, b)
    assert np.abs(a - b).max() <= 2e-2, (a, b)
    assert b[-1] < b[0] - 1e-3, b


def test_bf16_bench_shape_under_stream_overlap_is_stable():
    """The bench configuration in miniature (hidden 32, 100x100, weight gradients on the side stream): several steps
    back to back without a sync in between, twice from the same state -- the losses must be finite and agree run to
    run to the rounding noise of the float-atomic reductions (statistics rows of the narrow layers, LayerNorm / bias
    gradients), amplified over the optimizer steps in between (AdamW normalises near-zero gradients, so
    1e-7 of summation noise moves such a parameter by a full learning-rate step): 5e-4 after four steps. (An in-flight prefetch landing in a recycled register of the conv kernel showed up exactly here: as a GPU
    memory fault, only when the second stream delayed the loads.)"""
    from cultionet_amd.data import Data
    from cultionet_amd.lightning import HipTrainer
    from oracle import towerunet_oracle as O
    from oracle.selfcheck import build_pair

    x, y, bdist = O.seeded_batch(16, height=100, width=100, seed=9)
    batch = Data(x=x.cuda(), y=y.cuda(), bdist=bdist.cuda())
    runs = []
    for _ in range(2):
        lit, _ = build_pair(hidden=32, device="cuda:0")
        lit.train()
        tr = HipTrainer(lit, precision="bf16-mixed")
        ls = [tr.training_step(batch).clone() for _ in range(4)]
        torch.cuda.synchronize()
        runs.append([float(l.item()) for l in ls])
    assert all(np.isfinite(v) for v in runs[0]), runs
    assert np.abs(np.array(runs[0]) - np.array(runs[1])).max() <= 5e-4, runs
    assert runs[0][-1] < runs[0][0], runs


def test_bf16_eval_forward_and_dropin_autocast(golden_dir):
    """Eval-mode forward on the bf16 path (running statistics), and the drop-in surface under torch.autocast: the
    LightningModule forward picks the bf16 path up from the autocast state Lightning's precision plugin sets."""
    from oracle.make_golden import calibrate_bn

    g = np.load(os.path.join(golden_dir, "train_bf16_h8_b2_28.npz"))
    lit, batch = _setup(g)
    model = lit.cultionet_model.mask_model
    calibrate_bn(model, lambda: model(batch.x))
    with torch.no_grad():
        p32 = lit(batch)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            p16 = lit(batch)
    for k in KEYS:
        d = (p32[k] - p16[k].float()).abs()
        assert float(d.max()) > 0.0  # the bf16 path really ran
        assert float(d.mean()) <= 6e-3 and float(d.max()) <= 8e-2, (k, float(d.mean()), float(d.max()))
    # inference fuses every ConvBlock2d into ONE launch (BatchNorm folded into the packed weights, SiLU and the
    # ResUNet-a sum in the conv epilogue): against the unfused bf16 form (conv, BatchNorm apply, ...) the maps agree to
    # bf16 rounding, and the fused form launches far fewer kernels
    from cultionet_amd import _lib
    from cultionet_amd import engine as E

    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        _lib.query("cn_launch_count", 1)
        lit(batch)
        n_fused = _lib.query("cn_launch_count", 1)
        prev = E.eval_fusion(False)
        try:
            lit(batch)
            n_plain = _lib.query("cn_launch_count", 1)
            p16u = lit(batch)
        finally:
            E.eval_fusion(prev)
    assert n_fused <= n_plain - 60, (n_fused, n_plain)
    for k in KEYS:  # two different bf16 roundings of the same fp32 function: each within 6e-3 / 8e-2 of fp32 (above)
        d = (p16u[k].float() - p16[k].float()).abs()
        assert float(d.mean()) <= 9e-3 and float(d.max()) <= 0.12, (k, float(d.mean()), float(d.max()))
    # a train-mode forward updates running statistics in-kernel: the folded copies must follow
    lit.train()
    with torch.no_grad():
        model(batch.x * 1.5 + 0.1)
    lit.eval()
    with torch.no_grad():
        q32 = lit(batch)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            q16 = lit(batch)
    assert float((q32["crop"] - p32["crop"]).abs().max()) > 1e-4  # the statistics did move
    for k in KEYS:
        d = (q32[k] - q16[k].float()).abs()
        assert float(d.mean()) <= 6e-3 and float(d.max()) <= 8e-2, (k, float(d.mean()), float(d.max()))
    lit.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = lit(batch)
        loss, _ = lit.calc_loss(batch, pred)
    loss.backward()
    assert abs(float(loss) - float(g["fp32_loss"])) <= 5e-4
    gsum = sum(float(p.grad.abs().sum()) for p in model.parameters())
    assert np.isfinite(gsum) and gsum > 0


def test_bf16_dropin_with_default_dropout_trains():
    """The reference's default operating point (model.py:59,86,168-186): precision "16-mixed" + dropout 0.1, driven the
    drop-in way -- forward(Data) under torch.autocast, calc_loss, loss.backward(), torch AdamW. Dropout2d after the
    encoder blocks and natten's attn / proj dropout run on the bf16 NHWC kernels with counter-based masks: the step is
    reproducible from the seed, finite, and the loss goes down over a few optimizer steps."""
    from cultionet_amd import engine as E
    from cultionet_amd import synthetic as S
    from cultionet_amd.data import Data
    from cultionet_amd.lightning import CultionetLitModel

    dev = torch.device("cuda:0")
    lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=8, dropout=0.1)
    m = lit.cultionet_model.mask_model
    m.load_state_dict(S.seeded_state_dict(m.state_dict()))
    lit = lit.to(dev).train()
    x, y, bdist = S.seeded_batch(2, height=28, width=28, with_mask=True)
    batch = Data(x=x.to(dev), y=y.to(dev), bdist=bdist.to(dev))
    opt = torch.optim.AdamW(m.parameters(), lr=0.005, weight_decay=1e-3, eps=1e-4, betas=(0.9, 0.98))

    def step(seed):
        E.manual_seed(seed)
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = lit(batch)
            loss, _ = lit.calc_loss(batch, pred)
        loss.backward()
        return float(loss), torch.cat([p.grad.flatten() for p in m.parameters()]).clone()

    l1, g1 = step(11)
    l2, g2 = step(11)
    # same seed -> same masks: equal up to the rounding noise of the float-atomic statistics rows of the narrow layers
    # (hidden 8: every conv is "narrow"); another seed draws other masks and moves the loss by far more
    assert abs(l1 - l2) <= 1e-5 and torch.isfinite(g1).all()
    assert (g1 - g2).norm() <= 2e-2 * g1.norm()
    l3, g3 = step(12)
    assert abs(l3 - l1) > 2e-5 and (g3 - g1).norm() > 0.1 * g1.norm()
    losses = []
    for i in range(8):
        l, _ = step(100 + i)
        opt.step()
        losses.append(l)
    assert all(np.isfinite(losses)) and min(losses[-3:]) < losses[0], losses


def test_native_bf16_step_with_dropout_runs():
    from cultionet_amd import engine as E
    from cultionet_amd import synthetic as S
    from cultionet_amd.data import Data
    from cultionet_amd.lightning import CultionetLitModel, HipTrainer

    dev = torch.device("cuda:0")
    lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=8, dropout=0.2)
    m = lit.cultionet_model.mask_model
    m.load_state_dict(S.seeded_state_dict(m.state_dict()))
    lit = lit.to(dev).train()
    x, y, bdist = S.seeded_batch(2, height=28, width=28, with_mask=True)
    batch = Data(x=x.to(dev), y=y.to(dev), bdist=bdist.to(dev))
    tr = HipTrainer(lit, precision="bf16-mixed")
    E.manual_seed(7)
    l1 = float(tr.forward_backward(batch).item())
    E.manual_seed(7)
    l2 = float(tr.forward_backward(batch).item())
    assert abs(l1 - l2) <= 1e-5 and torch.isfinite(tr.store.flat_grad).all() and 0.3 < l1 < 1.0


def test_precision_selection_of_the_dropin_forward():
    """autograd_bridge._autocast_bf16: ambient bf16 autocast selects the mixed path (announced once), fp16 autocast is
    served by it too (announced), and an explicit ``hip_precision`` wins over whatever autocast region is open."""
    import warnings

    from cultionet_amd import autograd_bridge as AB
    from cultionet_amd import synthetic as S
    from cultionet_amd.lightning import CultionetLitModel

    dev = torch.device("cuda:0")
    lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=8, dropout=0.0)
    m = lit.cultionet_model.mask_model
    m.load_state_dict(S.seeded_state_dict(m.state_dict()))
    lit = lit.to(dev).eval()
    x, _, _ = S.seeded_batch(2, height=28, width=28)
    xd = x.to(dev)
    with torch.no_grad():
        ref32 = m(xd)["crop"].clone()
        AB._warned.clear()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            with torch.autocast("cuda", dtype=torch.bfloat16):
                mixed = m(xd)["crop"].clone()
                m(xd)
            assert sum("bf16 mixed-precision" in str(i.message) for i in w) == 1  # once, not per call
        assert not torch.equal(mixed, ref32) and (mixed - ref32).abs().max() < 0.1
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            with torch.autocast("cuda", dtype=torch.float16):
                half = m(xd)["crop"].clone()
            assert any("fp16 autocast" in str(i.message) for i in w)
        assert torch.equal(half, mixed)  # the same bf16 path
        lit.hip_precision = "32-true"
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert torch.equal(m(xd)["crop"], ref32)
        lit.hip_precision = "bf16-mixed"
        assert torch.equal(m(xd)["crop"], mixed)  # no autocast region needed
        lit.hip_precision = None
        assert torch.equal(m(xd)["crop"], ref32)
    with pytest.raises(ValueError):
        lit.hip_precision = "64-true"
