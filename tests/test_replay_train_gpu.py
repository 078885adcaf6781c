"""HipTrainer(replay=True): forward + loss + backward from a recorded launch plan (cultionet_amd/replay.py) against the
eager step on identical weights and batches. The plan repeats the same kernels on private buffers in the same stream
order, so the trajectories agree to the noise the eager step has against ITSELF (float-atomic parameter-gradient sums)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(precision, hidden=8, B=2, H=28, W=28):
    from cultionet_amd import synthetic as S
    from cultionet_amd.data import Data
    from cultionet_amd.lightning import CultionetLitModel, HipTrainer

    trainers = []
    for replay in (False, True):
        lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=hidden, dropout=0.0)
        m = lit.cultionet_model.mask_model
        m.load_state_dict(S.seeded_state_dict(m.state_dict()))
        lit = lit.to("cuda:0").train()
        trainers.append(HipTrainer(lit, precision=precision, replay=replay))
    batches = []
    for k in range(3):
        x, y, bd = S.seeded_batch(B, height=H, width=W, seed=50 + k, with_mask=True)
        batches.append(Data(x=x.cuda(), y=y.cuda(), bdist=bd.cuda()))
    return trainers, batches


# (bf16: two EAGER runs of this hidden-8 net differ by ~3e-5 after one optimizer step -- float-atomic parameter-gradient
# sums, amplified by AdamW -- and by up to ~5e-4 after eight: the replayed trajectory is held to 1e-3)
@pytest.mark.parametrize("precision,tol", [("32-true", 2e-6), ("bf16-mixed", 1e-3)])
def test_replayed_steps_follow_the_eager_trajectory(precision, tol):
    (eager, plan), batches = _pair(precision)
    le, lp = [], []
    for i in range(8):
        b = batches[i % len(batches)]  # a different batch (different tensors) every step: copied into the plan's inputs
        le.append(float(eager.training_step(b).item()))
        lp.append(float(plan.training_step(b).item()))
    assert plan._plan is not None and plan._plan.n_calls > 100  # steps 3.. ran from the plan
    assert np.abs(np.array(le) - np.array(lp)).max() <= tol, (le, lp)
    assert le[-1] < le[0]
    pe = dict(eager.model.named_parameters())
    worst = max(float((p.detach() - pe[n].detach()).abs().max()) for n, p in plan.model.named_parameters())
    assert worst <= (50 * tol if precision == "32-true" else 0.05), worst
    # outputs of a replayed step are the plan's buffers
    for k in ("distance", "edge", "crop"):
        assert torch.isfinite(plan.last_outputs[k]).all() and plan.last_outputs[k].shape == eager.last_outputs[k].shape
    # running statistics moved identically (the recorded BatchNorm launches update them in place)
    se, sp = eager.model.state_dict(), plan.model.state_dict()
    k0 = next(k for k in se if k.endswith("running_var") and "tower_a" in k)
    # (mixed precision: eight AdamW steps amplify the eager step's own float-atomic noise -- eager vs eager differs by
    # as much, tests/test_bf16_model_gpu.py::test_bf16_bench_shape_under_stream_overlap_is_stable -- hence relative)
    assert float((se[k0] - sp[k0]).abs().max()) <= (50 * tol if precision == "32-true" else 0.03 * float(se[k0].abs().max()))
    nb = next(k for k in se if k.endswith("num_batches_tracked"))
    assert int(se[nb]) == int(sp[nb]) == 8


def test_replay_falls_back_for_new_shapes():
    from cultionet_amd import synthetic as S
    from cultionet_amd.data import Data
    from cultionet_amd.lightning import CultionetLitModel, HipTrainer

    (eager, plan), batches = _pair("32-true")
    for i in range(4):
        plan.training_step(batches[0])
    first = plan._plan
    assert first is not None
    x, y, bd = S.seeded_batch(3, height=28, width=28, seed=9)  # another batch size: a new plan, not a wrong replay
    other = Data(x=x.cuda(), y=y.cuda(), bdist=bd.cuda())
    l_plan = float(plan.training_step(other).item())
    assert plan._plan is None and np.isfinite(l_plan)  # eager again for the new shape ...
    for _ in range(3):
        plan.training_step(other)
    assert plan._plan is not None and plan._plan is not first  # ... until its own plan is recorded


@pytest.mark.parametrize("precision", ["32-true", "bf16-mixed"])
def test_replay_with_dropout_draws_the_eager_masks(precision):
    """The reference's default dropout (0.1: Dropout2d after the encoder blocks, natten attn_drop / proj_drop in the
    decoder) under replay: the per-step part of every mask seed is a device word bumped by the plan's first launch, so a
    REPLAYED step draws fresh masks every step -- and exactly the masks the eager step draws from the same seed. The
    two trainers are stepped in separate phases (they share the process-wide seed state)."""
    from cultionet_amd import engine as E
    from cultionet_amd import synthetic as S
    from cultionet_amd.data import Data
    from cultionet_amd.lightning import CultionetLitModel, HipTrainer

    batches = []
    for k in range(3):
        x, y, bd = S.seeded_batch(2, height=28, width=28, seed=50 + k, with_mask=True)
        batches.append(Data(x=x.cuda(), y=y.cuda(), bdist=bd.cuda()))
    runs = {}
    for replay in (False, True):
        lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=8, dropout=0.25)
        m = lit.cultionet_model.mask_model
        m.load_state_dict(S.seeded_state_dict(m.state_dict()))
        lit = lit.to("cuda:0").train()
        tr = HipTrainer(lit, precision=precision, replay=replay)
        E.manual_seed(1234)
        losses, outs = [], []
        for i in range(7):
            losses.append(float(tr.training_step(batches[i % 3]).item()))
            outs.append(tr.last_outputs["crop"].float().clone())
        if replay:
            assert tr._plan is not None and tr._plan.n_calls > 100  # steps 3.. ran from the plan, dropout and all
        runs[replay] = (losses, outs)
    le, lp = np.array(runs[False][0]), np.array(runs[True][0])
    # fp32: the replayed trajectory IS the eager one (2e-6: identical masks at every step -- a single different mask
    # moves the loss by ~1e-2). bf16: two EAGER runs already differ by ~3e-5 after one optimizer step (float-atomic
    # parameter-gradient sums amplified by AdamW on a dropout-thinned hidden-8 net), growing to ~5e-4 over seven steps.
    tol = 2e-6 if precision == "32-true" else 2e-3
    assert np.abs(le - lp).max() <= tol, (le, lp)
    for a, b in zip(runs[False][1], runs[True][1]):
        assert float((a - b).abs().max()) <= (1e-5 if precision == "32-true" else 6e-2)
    lit0 = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=8, dropout=0.0)
    m0 = lit0.cultionet_model.mask_model
    m0.load_state_dict(S.seeded_state_dict(m0.state_dict()))
    tr0 = HipTrainer(lit0.to("cuda:0").train(), precision=precision)
    l0 = float(tr0.training_step(batches[0]).item())
    assert abs(l0 - le[0]) > 1e-4, (l0, le[0])  # the masks are real: with dropout off the first loss differs


def test_replayed_masks_change_every_step():
    """Forward-only view of the device step word: the same batch through the same recorded plan gives different
    activations on consecutive replays (fresh masks), and manual_seed() makes a run repeatable."""
    from cultionet_amd import engine as E
    from cultionet_amd import synthetic as S
    from cultionet_amd.data import Data
    from cultionet_amd.lightning import CultionetLitModel, HipTrainer

    x, y, bd = S.seeded_batch(2, height=28, width=28, seed=3, with_mask=True)
    b = Data(x=x.cuda(), y=y.cuda(), bdist=bd.cuda())

    def run():
        lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=8, dropout=0.3)
        m = lit.cultionet_model.mask_model
        m.load_state_dict(S.seeded_state_dict(m.state_dict()))
        tr = HipTrainer(lit.to("cuda:0").train(), replay=True, lr_fn=lambda step: 0.0)  # weights frozen: lr 0
        E.manual_seed(77)
        outs = []
        for _ in range(6):
            tr.training_step(b)
            outs.append(tr.last_outputs["distance"].clone())
        assert tr._plan is not None
        return outs

    a, c = run(), run()
    for i in range(6):
        assert torch.equal(a[i], c[i]), i  # repeatable from the seed, replayed steps included
    # lr 0 keeps the weights fixed (weight decay scales with lr), BatchNorm uses batch statistics: only the masks differ
    assert float((a[3] - a[4]).abs().max()) > 1e-4 and float((a[4] - a[5]).abs().max()) > 1e-4


def test_replay_casts_noncanonical_labels_every_step():
    """Labels that are not int64 (int32 from a dataset) and a non-contiguous distance tensor are cast / re-strided by
    the copy into the plan's canonical input buffers on EVERY replayed step -- not once at record time."""
    (eager, plan), batches = _pair("32-true")
    from cultionet_amd.data import Data

    alt = [Data(x=b.x, y=b.y.to(torch.int32), bdist=b.bdist.transpose(1, 2).contiguous().transpose(1, 2))
           for b in batches]
    assert not alt[0].bdist.is_contiguous()
    le, lp = [], []
    for i in range(7):
        le.append(float(eager.training_step(batches[i % 3]).item()))
        lp.append(float(plan.training_step(alt[i % 3]).item()))
    assert plan._plan is not None
    assert np.abs(np.array(le) - np.array(lp)).max() <= 2e-6, (le, lp)


def test_plan_recorded_without_a_weight_update_still_repacks():
    """A plan recorded from forward_backward() calls with NO optimizer step in between (gradient-accumulation style) must
    still contain the batched weight re-pack: later replays follow optimizer updates like the eager step does."""
    (eager, plan), batches = _pair("32-true")
    for _ in range(3):  # two eager passes + the recorded one, the parameters untouched in between
        eager.forward_backward(batches[0])
        plan.forward_backward(batches[0])
    assert plan._plan is not None
    le, lp = [], []
    for i in range(5):
        le.append(float(eager.training_step(batches[i % 3]).item()))
        lp.append(float(plan.training_step(batches[i % 3]).item()))
    assert np.abs(np.array(le) - np.array(lp)).max() <= 2e-6, (le, lp)
    assert le[-1] < le[0]
