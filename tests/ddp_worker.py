"""Child process of tests/test_ddp_engine_gpu.py: one data-parallel rank of the REAL engine.

    python tests/ddp_worker.py RANK WORLD PORT OUTDIR HIDDEN B H W

Every rank drives cuda:0 (the GPU box has one GPU) over the gloo backend -- RCCL refuses two ranks on one device,
gloo all-reduces device tensors through the host -- with the same HipTrainer + GradientAllReduce objects bench.py
uses with RCCL. Rank r > 0 deliberately starts from DIFFERENT (randomly initialised) weights: the construction-time
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
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
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
    lit = lit.to("cuda:0").train()
    x, y, bdist = S.seeded_batch(B, height=H, width=W, seed=7 + rank, with_mask=True)
    batch = Data(x=x.cuda(), y=y.cuda(), bdist=bdist.cuda())
    comm = GradientAllReduce(world_size=world, bucket_mb=0.25)  # small buckets: several launches mid-backward
    trainer = HipTrainer(lit, gradient_clip_val=1.0, comm=comm)
    loss = trainer.training_step(batch)
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save({"state": sd, "loss": float(loss.item()), "buckets": len(comm._plan)},
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
