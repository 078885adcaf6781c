"""Benchmark of the TowerUNet training hot path on MI355X (BASELINE.json: train chips/sec).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one synthetic batch already resident in HBM:
forward + Tanimoto loss + backward + global-norm clip + AdamW, all in the hand-written HIP kernels
(cultionet_amd.lightning.HipTrainer). Workload at N=1 = BASELINE configs[1]: TowerUNet fp32,
hidden 32, batch 8 of [3,12,100,100] chips. N > 1 shards chips over ranks (weak scaling, per-GPU batch
fixed) with one bucketed RCCL all-reduce of the flat gradient overlapped with the backward tape.

Rank 0 prints ONE JSON line; `roofline` is the dominant kernel's algorithmic FLOP/s from HIP events
recorded on the launch stream inside the timed region; `cpu_baseline` is the CPU oracle (a port of the
reference's PyTorch-CPU path) timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FWD_GFLOP_PER_CHIP = {32: 64.88, 64: 258.0}  # forward 2*MAC FLOPs at [1,3,12,100,100]
PEAK_F32_MFMA_TFLOPS = 157.3                 # MI355X_MICROARCH.md: f32-input MFMA == f32 vector peak
KIND_NAMES = ["cn_conv_igemm_kernel<NT=128>", "cn_conv_igemm_kernel<NT<=64>", "cn_wgrad_kernel<3x3>",
              "cn_wgrad_kernel<1x1>"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="chips per GPU")
    ap.add_argument("--hidden", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = best of the documented sweep (16)")
    return ap.parse_args()


def cpu_baseline(batch: int, hidden: int, steps: int, threads: int = 0):
    """The oracle (port of the reference CPU path) timed on the host cores: fwd + loss + bwd + AdamW."""
    import torch

    from oracle import towerunet_oracle as O

    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # Thread count: the best of a sweep on the GPU box's host (2 x EPYC 9575F, 256 logical CPUs), batch 8:
    # 4 thr 2.1, 8 thr 2.8, 16 thr 3.3-3.5, 24 thr 3.0, 32 thr 2.8, 64 thr 1.6, 128 thr 0.74, 256 thr 0.03 chips/s
    # (oneDNN scales poorly on these small 100x100 chips); `cores` reports the threads actually used.
    cores = min(threads, avail) if threads else min(16, avail)
    torch.set_num_threads(cores)
    m = O.TowerUNet(3, 12, hidden_channels=hidden)
    m.load_state_dict(O.seeded_state_dict(m.state_dict()))
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=0.01, weight_decay=1e-3, eps=1e-4, betas=(0.9, 0.98))
    x, y, bdist = O.seeded_batch(batch, seed=7)

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = O.calc_loss(m(x), y, bdist)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        return float(loss)

    step()  # warm-up (oneDNN primitive creation)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return {
        "value": batch * steps / dt,
        "unit": "chips/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{steps} train steps (1 warm-up discarded) of batch {batch} x [3,12,100,100], hidden {hidden}, "
                  f"fp32, torch {torch.__version__} CPU, {cores} threads",
    }


# rocprof kernel-name prefixes behind each profiled kind (template instantiations of one kernel family)
KIND_PATTERNS = {
    0: ("cn_conv_igemm_vec_kernel<4, 1,", "cn_conv_igemm_vec_kernel<2, 2,", "cn_conv_igemm_kernel<2, 2,",
        "cn_conv1x1_kernel<"),
    1: ("cn_conv_igemm_vec_kernel<1,", "cn_conv_igemm_kernel<1,"),
    2: ("cn_wgrad_vec_kernel<9,", "cn_wgrad_kernel<9>"),
    3: ("cn_wgrad_vec_kernel<1,", "cn_wgrad_kernel<1>"),
}


def pmc_traffic(prefixes):
    """HBM-side bytes per launch of a kernel family, from the committed PMC passes of THIS command
    (profiles/r01_pmc_traffic.json: FETCH_SIZE x2 (gfx950) + WRITE_SIZE, separate rocprofv3 --pmc runs,
    tools/pmc_traffic.py). PMC counters cannot be collected from inside the timed run, so the figure is read
    from that file; None if it is absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            kernels = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    tot, n = 0.0, 0
    for name, v in kernels.items():
        if name.startswith(prefixes):
            tot += v["hbm_bytes_per_launch"] * v["launches"]
            n += v["launches"]
    return tot / n if n else None


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = None
    use_dist = world > 1 or os.environ.get("CN_FORCE_COMM") == "1"  # CN_FORCE_COMM: exercise RCCL with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"  # RCCL's version banner goes to stdout; keep stdout to ONE JSON line
        # RCCL logs (e.g. its rsmi warnings) default to stdout too: send them to stderr
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        from cultionet_amd.ddp import GradientAllReduce

        comm = GradientAllReduce(world_size=world)

    from cultionet_amd import _lib
    from cultionet_amd.data import Data
    from cultionet_amd.lightning import CultionetLitModel, HipTrainer
    from cultionet_amd import synthetic as O

    _lib.load()
    B, hidden = args.batch, args.hidden
    lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=hidden, dropout=0.0)
    model = lit.cultionet_model.mask_model
    model.load_state_dict(O.seeded_state_dict(model.state_dict()))
    lit = lit.to(dev).train()
    x, y, bdist = O.seeded_batch(B, seed=7 + rank)
    batch = Data(x=x.to(dev), y=y.to(dev), bdist=bdist.to(dev), lon=torch.zeros(B, device=dev),
                 lat=torch.zeros(B, device=dev))
    trainer = HipTrainer(lit, gradient_clip_val=1.0, comm=comm)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.training_step(batch)
    sync()
    _lib.call("cn_profile_begin")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.training_step(batch)
    sync()
    dt = time.perf_counter() - t0
    prof = (ctypes.c_double * 24)()
    _lib.call("cn_profile_end", prof)
    loss_val = float(loss.item())

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        chips = world * B * args.steps
        value = chips / dt
        kinds = [(prof[3 * k], prof[3 * k + 1], prof[3 * k + 2]) for k in range(4)]
        dom = max(range(4), key=lambda k: kinds[k][0])
        ms, flops, launches = kinds[dom]
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        train_gflop = 3.0 * FWD_GFLOP_PER_CHIP.get(hidden, 0.0)
        out = {
            "metric": "train_chips_per_sec",
            "value": value,
            "unit": "chips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"TowerUNet train step (fwd + Tanimoto + bwd + clip + AdamW), hidden {hidden}, "
                            f"per-GPU batch {B} x [3,12,100,100] fp32 (BASELINE configs[1])",
                "global_batch": world * B,
                "parallelism": f"dp{world}" if world > 1 else "single",
                "loss": loss_val,
            },
            "roofline": {
                "bound": "mfma",
                "kernel": KIND_NAMES[dom],
                "achieved": achieved,
                "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                # the committed PMC passes are of the default workload (hidden 32, batch 8)
                "traffic": pmc_traffic(KIND_PATTERNS[dom]) if (hidden, B) == (32, 8) else None,
                "avg_launch_us": ms * 1e3 / launches if launches else None,
                "launches_per_step": launches / args.steps,
                "share_of_step_time": ms * 1e-3 / dt,
                "by_kernel": {KIND_NAMES[k]: {"ms_per_step": kinds[k][0] / args.steps,
                                             "tflops": (kinds[k][1] / (kinds[k][0] * 1e-3) / 1e12) if kinds[k][0] else 0.0,
                                             "launches_per_step": kinds[k][2] / args.steps} for k in range(4)},
                "end_to_end_tflops": value * train_gflop / 1e3,
                "end_to_end_frac": value * train_gflop / 1e3 / PEAK_F32_MFMA_TFLOPS,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, hidden, args.cpu_steps, args.cpu_threads)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)  # the ONE stdout line, after RCCL has been torn down


if __name__ == "__main__":
    main()
