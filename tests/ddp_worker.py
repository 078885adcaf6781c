"""Child process of tests/test_ddp_engine_gpu.py: one data-parallel rank of the REAL engine.

    python tests/ddp_worker.py RANK WORLD PORT OUTDIR HIDDEN B H W [PRECISION [FEED [STEPS]]]

PRECISION "32-true" (default) | "bf16-mixed". FEED 0: resident batches; 1: RAW int16 batches in pinned host memory
through cultionet_amd.feeder.DeviceFeeder (copy stream + cn_prepare_chips_f32); 2: the same raw data prepared on the
host with the reference's arithmetic (the control for FEED 1). STEPS optimizer steps, a different batch each.

Default: every rank drives cuda:0 (the GPU box has one GPU) over the gloo backend -- RCCL refuses two ranks on one
device, gloo all-reduces device tensors through the host -- with the same HipTrainer + GradientAllReduce objects bench.py
uses with RCCL. CN_DDP_BACKEND=nccl (tests/test_ddp_rccl_gpu.py): rank r drives cuda:r and the buckets are RCCL
all-reduces over xGMI, the configuration of BASELINE configs[3]. Rank r > 0 deliberately starts from DIFFERENT (randomly initialised) weights: the construction-time
broadcast must make the replicas identical, as torch DDP does for the reference (model.py:101,184).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    outdir = sys.argv[4]
    hidden, B, H, W = (int(v) for v in sys.argv[5:9])
    precision = sys.argv[9] if len(sys.argv) > 9 else "32-true"
    feed = int(sys.argv[10]) if len(sys.argv) > 10 else 0
    steps = int(sys.argv[11]) if len(sys.argv) > 11 else 1
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    backend = os.environ.get("CN_DDP_BACKEND", "gloo")
    dev = f"cuda:{rank}" if backend == "nccl" else "cuda:0"
    if backend == "nccl":
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from cultionet_amd import synthetic as S
    from cultionet_amd.data import Data
    from cultionet_amd.ddp import GradientAllReduce
    from cultionet_amd.lightning import CultionetLitModel, HipTrainer

    torch.manual_seed(1234 + rank)
    lit = CultionetLitModel(in_channels=3, in_time=12, hidden_channels=hidden, dropout=0.0)
    model = lit.cultionet_model.mask_model
    if rank == 0:
        model.load_state_dict(S.seeded_state_dict(model.state_dict()))
    lit = lit.to(dev).train()
    comm = GradientAllReduce(world_size=world, bucket_mb=float(os.environ.get("CN_DDP_BUCKET_MB", "0.25")))  # small buckets: several launches mid-backward
    trainer = HipTrainer(lit, gradient_clip_val=1.0, comm=comm, precision=precision)
    hosts = []
    for k in range(steps):
        x, y, bdist = S.seeded_batch(B, height=H, width=W, seed=7 + rank + 100 * k, with_mask=True)
        if feed == 0:
            hosts.append(Data(x=x.to(dev), y=y.to(dev), bdist=bdist.to(dev)))
        else:  # raw reflectances, as a dataset stores them (int16, scale 1e-4)
            xr = (x.abs() * 3000.0).clamp(0, 20000).to(torch.int16)
            br = (bdist * 10000.0).to(torch.int16)
            if feed == 1:
                hosts.append(Data(x=xr.pin_memory(), y=y.to(torch.int32).pin_memory(), bdist=br.pin_memory()))
            else:  # the reference's host arithmetic (data/datasets.py:443-446): x / 10000 -> clip(1e-9, 1)
                hosts.append(Data(x=(xr.float() / 10000.0).clip(1e-9, 1).to(dev), y=y.to(dev),
                                  bdist=(br.float() / 10000.0).clip(1e-9, 1).to(dev)))
    losses = []
    if feed == 1:
        from cultionet_amd.feeder import DeviceFeeder

        for b in DeviceFeeder(dev).iterate(hosts):
            losses.append(trainer.training_step(b).clone())
    else:
        for b in hosts:
            losses.append(trainer.training_step(b).clone())
    torch.cuda.synchronize()
    loss = losses[0]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save({"state": sd, "loss": float(loss.item()), "losses": [float(l.item()) for l in losses],
                "buckets": len(comm._plan), "world_size": dist.get_world_size(), "backend": dist.get_backend(),
                "device": torch.cuda.get_device_name(dev), "device_index": torch.cuda.current_device()},
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
