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


@pytest.mark.parametrize("precision,tol", [("32-true", 2e-6), ("bf16-mixed", 2e-4)])
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
    assert worst <= 50 * tol, worst
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


def test_replay_falls_back_for_new_shapes_and_dropout():
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
    # dropout > 0 needs a fresh seed per step: the trainer stays eager
    lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=8, dropout=0.1).to("cuda:0").train()
    tr = HipTrainer(lit, replay=True)
    for _ in range(4):
        tr.training_step(batches[0])
    assert tr._plan is None


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
