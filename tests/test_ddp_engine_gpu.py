"""Two data-parallel ranks of the REAL engine (ParamStore + tape + GradientAllReduce + fused clip/AdamW) against the
SURVEY 8(e) oracle: two independent CPU-oracle shards, gradients averaged, clip_grad_norm_(1.0), AdamW -- what torch
DDP under Lightning computes for the reference (model.py:101,168-186). BatchNorm statistics stay per rank."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(outdir, hidden, B, H, W, world=2, extra=()):
    """Start `world` ranks of tests/ddp_worker.py (gloo over loopback, every rank on cuda:0); returns their records."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["GLOO_SOCKET_IFNAME"] = "lo"  # the container hostname may not resolve: keep gloo's pairs on loopback
    os.makedirs(outdir, exist_ok=True)

    def launch():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_worker.py"), str(r), str(world),
                                   str(port), str(outdir), str(hidden), str(B), str(H), str(W)] + [str(e) for e in extra],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
        outs = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=180)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                return None, None
            outs.append(out.decode(errors="replace"))
        return procs, outs

    # the rendezvous (a just-released ephemeral port, gloo's full-mesh connect) failed once in a full-suite run on the
    # GPU box and the workers sat in connectFullMesh until the timeout: one retry on a fresh port
    procs, outs = launch()
    if procs is None:
        procs, outs = launch()
    assert procs is not None, "both launches of the two ranks timed out"
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(os.path.join(outdir, f"rank{r}.pt"), weights_only=False) for r in range(world)]


def test_two_ranks_bf16_engine_match_two_oracle_shards(tmp_path):
    """The MIXED-PRECISION engine under data parallelism (BASELINE configs[3] is bf16 at per-GPU batch 32): two ranks,
    bf16 NHWC activations / fp32 master weights, bucketed all-reduce of the fp32 flat gradient. Per-rank losses against
    the fp32 oracle shards at the bf16 tolerance (5e-4), replicas bitwise identical after the update, the update itself
    against the oracle's averaged-gradient AdamW step in the median (AdamW normalises every gradient to a +-lr move, so
    single elements whose tiny gradient changes sign under bf16 rounding differ by 2 lr: a max is meaningless)."""
    from oracle import towerunet_oracle as O

    hidden, B, H, W, world = 16, 2, 28, 28, 2
    got = _run_ranks(tmp_path, hidden, B, H, W, world, extra=("bf16-mixed",))
    assert all(g["buckets"] >= 2 for g in got)
    models, losses = [], []
    for r in range(world):
        m = O.TowerUNet(3, 12, hidden_channels=hidden)
        m.load_state_dict(O.seeded_state_dict(m.state_dict()))
        m.train()
        x, y, bdist = O.seeded_batch(B, height=H, width=W, seed=7 + r, with_mask=True)
        loss, _ = O.calc_loss(m(x), y, bdist)
        loss.backward()
        models.append(m)
        losses.append(float(loss))
    ref = models[0]
    before = {n: p.detach().clone() for n, p in ref.named_parameters()}
    with torch.no_grad():
        for ps in zip(*[m.parameters() for m in models]):
            ps[0].grad = sum(p.grad for p in ps) / world
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
    torch.optim.AdamW(ref.parameters(), lr=0.01, weight_decay=1e-3, eps=1e-4, betas=(0.9, 0.98)).step()
    for r in range(world):
        assert abs(got[r]["loss"] - losses[r]) <= 5e-4, (r, got[r]["loss"], losses[r])
    for n, pr in ref.named_parameters():
        assert torch.equal(got[0]["state"][n], got[1]["state"][n]), n  # replicas bitwise identical
    moved_ref = torch.cat([(p.detach() - before[n]).flatten() for n, p in ref.named_parameters()])
    moved_got = torch.cat([(got[0]["state"][n] - before[n]).flatten() for n, _ in ref.named_parameters()])
    agree = (torch.sign(moved_ref) == torch.sign(moved_got)).float().mean()
    assert float(agree) >= 0.97, float(agree)  # the update direction of (nearly) every element
    assert float((moved_ref - moved_got).abs().median()) <= 2e-4


def test_two_ranks_fed_from_pinned_host_batches(tmp_path):
    """DeviceFeeder + data parallelism: raw int16 batches in pinned host memory, copied on the copy stream and prepared
    on the device while the step, the weight-gradient side stream and the bucket stream are busy -- three optimizer steps,
    a fresh batch each -- against the same raw data prepared on the host with the reference's arithmetic and kept
    resident. Same losses (1e-5), same final weights (1e-5), replicas bitwise identical."""
    hidden, B, H, W = 8, 2, 28, 28
    fed = _run_ranks(os.path.join(tmp_path, "fed"), hidden, B, H, W, extra=("32-true", 1, 3))
    ctl = _run_ranks(os.path.join(tmp_path, "ctl"), hidden, B, H, W, extra=("32-true", 2, 3))
    for r in range(2):
        assert len(fed[r]["losses"]) == 3
        assert max(abs(a - b) for a, b in zip(fed[r]["losses"], ctl[r]["losses"])) <= 1e-5, (fed[r]["losses"], ctl[r]["losses"])
        for n, v in ctl[r]["state"].items():
            if v.is_floating_point():
                assert float((fed[r]["state"][n] - v).abs().max()) <= 1e-4, n  # (three AdamW steps amplify summation noise)
    for n, v in fed[0]["state"].items():
        if v.is_floating_point() and "running_" not in n:
            assert torch.equal(v, fed[1]["state"][n]), n


def test_two_ranks_real_engine_match_two_oracle_shards(tmp_path):
    from oracle import towerunet_oracle as O

    hidden, B, H, W, world = 8, 2, 28, 28, 2
    got = _run_ranks(tmp_path, hidden, B, H, W, world)
    assert all(g["buckets"] >= 2 for g in got), [g["buckets"] for g in got]

    # the oracle: identical key-seeded weights, one shard per rank, averaged gradients, clip, AdamW
    models, losses = [], []
    for r in range(world):
        m = O.TowerUNet(3, 12, hidden_channels=hidden)
        m.load_state_dict(O.seeded_state_dict(m.state_dict()))
        m.train()
        x, y, bdist = O.seeded_batch(B, height=H, width=W, seed=7 + r, with_mask=True)
        loss, _ = O.calc_loss(m(x), y, bdist)
        loss.backward()
        models.append(m)
        losses.append(float(loss))
    ref = models[0]
    with torch.no_grad():
        for ps in zip(*[m.parameters() for m in models]):
            ps[0].grad = sum(p.grad for p in ps) / world
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
    opt = torch.optim.AdamW(ref.parameters(), lr=0.01, weight_decay=1e-3, eps=1e-4, betas=(0.9, 0.98))
    opt.step()

    for r in range(world):
        assert abs(got[r]["loss"] - losses[r]) <= 1e-4, (r, got[r]["loss"], losses[r])  # per-rank loss (no sync_dist)
    refp = dict(ref.named_parameters())
    for r in range(world):
        for n, pr in refp.items():
            d = (got[r]["state"][n] - pr.detach()).abs().max()
            assert d <= 2e-4, (r, n, float(d))
    # replicas are bitwise identical after the step (same averaged gradient, same update)
    for n in refp:
        assert torch.equal(got[0]["state"][n], got[1]["state"][n]), n
    # BatchNorm running statistics stay per rank (different shards => different statistics)
    k = next(k for k in got[0]["state"] if k.endswith("running_mean") and "tower_a" in k)
    assert not torch.equal(got[0]["state"][k], got[1]["state"][k])
