"""N > 1 path on CPU: world_size-2 gloo run of the bucketed gradient all-reduce driven by a tape
(the same GradientAllReduce object the GPU path uses with RCCL), plus the bucket planner."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cultionet_amd.ddp import GradientAllReduce, plan_buckets


def test_plan_buckets_cover_and_order():
    offsets = [0, 100, 300, 1000, 1500]
    sizes = [100, 200, 700, 500, 500]
    ready = [0, 3, 5, 9, 12]
    plan = plan_buckets(offsets, sizes, ready, 2000, 600)
    assert plan[0][1] == 2000 and plan[-1][0] == 0
    for (lo, hi, _), (lo2, hi2, _) in zip(plan, plan[1:]):
        assert lo == hi2  # contiguous, walking towards the start
    # a bucket is ready only when its earliest-forward parameter is done
    for lo, hi, r in plan:
        rs = [ready[i] for i, o in enumerate(offsets) if lo <= o < hi]
        assert r == min(rs)
    assert sum(hi - lo for lo, hi, _ in plan) == 2000


class _Store:
    def __init__(self, numel, offsets, sizes):
        self.numel = numel
        self.offsets = offsets
        self.params = [torch.empty(s) for s in sizes]
        self.flat_grad = torch.zeros(numel)


class _Tape:
    def __init__(self):
        self.nodes = []
        self.marks = {}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    offsets, sizes = [0, 64, 192, 448], [64, 128, 256, 64]
    store = _Store(512, offsets, sizes)
    tape = _Tape()
    # forward order: param i is used by node i; backward node i writes rank-dependent gradients
    for i, (o, s) in enumerate(zip(offsets, sizes)):
        def node(o=o, s=s, i=i):
            store.flat_grad[o:o + s] += (rank + 1) * (i + 1)
        tape.nodes.append(node)
        tape.marks[o] = i
    comm = GradientAllReduce(world_size=world, bucket_mb=128 * 4 / (1 << 20))
    comm.backward(tape, store)
    expect = torch.cat([torch.full((s,), float(sum(r + 1 for r in range(world)) * (i + 1))) for i, s in enumerate(sizes)])
    q.put((rank, bool(torch.equal(store.flat_grad, expect)), len(comm._plan)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert all(n >= 2 for _, _, n in res)  # more than one bucket was exercised
