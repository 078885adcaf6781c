"""cultionet_amd: MI355X-native TowerUNet hot path behind cultionet's module / LightningModule surface."""
import os as _os


def configure_runtime(hw_queues: int = 5) -> bool:
    """Cap the HIP runtime's hardware-queue pool at ``hw_queues`` per stream priority (``GPU_MAX_HW_QUEUES``) -- OPT-IN,
    never done at import (a drop-in library does not edit its host's environment behind its back).

    Why 5 (round 6, tools/one_aux_try.sh, profiles/r06_hw_queues.txt): this chip runs a process's hardware queues
    concurrently only while there are at most seven; the engine owns one lowest-priority queue (weight gradients) and one
    highest-priority queue (the auxiliary branch stream), which leaves five for everything of normal priority -- the
    compute stream, the bucket stream and the six streams a torch.distributed / RCCL process group brings. With the
    runtime's default of 4 the compute stream ends up sharing a queue with a collective (23.5 instead of 19.4 ms per bf16
    step under a process group, round 2); with 8 (rounds 2-5) a data-parallel rank had to give up its auxiliary stream
    (2162 chips/s against 2249 at 5, single process 2279-2285). The runtime reads the variable when it initialises, so
    this must run before the first HIP call of the process (``torch.cuda.is_available()`` included). Returns True when
    the variable is (now or already) set, False when the runtime was initialised before and the call came too late to
    matter. ``bench.py``, ``tests/conftest.py`` and ``__graft_entry__.smoke()`` call it; an application that trains
    through ``HipTrainer`` or Lightning should do the same at start-up."""
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return True
    import torch

    if torch.cuda.is_initialized():
        return False
    _os.environ["GPU_MAX_HW_QUEUES"] = str(int(hw_queues))
    return True
