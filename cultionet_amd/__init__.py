"""cultionet_amd: MI355X-native TowerUNet hot path behind cultionet's module / LightningModule surface."""
import os as _os


def configure_runtime(hw_queues: int = 8) -> bool:
    """Ask the HIP runtime for ``hw_queues`` hardware queues per device (``GPU_MAX_HW_QUEUES``) -- OPT-IN, never done at
    import (a drop-in library does not edit its host's environment behind its back).

    Why it matters: the training step runs on three HIP streams (compute, weight gradients, RCCL buckets) plus RCCL's
    own; with the runtime's default of 4 hardware queues two of them can land on ONE queue and serialise (measured:
    23.5 instead of 19.4 ms per bf16 step as soon as a process group exists; 16 queues were worse for a process that
    runs two configurations -- DESIGN section 7). The runtime reads the variable when it initialises, so this must run
    before the first HIP call of the process (``torch.cuda.is_available()`` included). Returns True when the variable is
    (now or already) set, False when the runtime was initialised before and the call came too late to matter.
    ``bench.py``, ``tests/conftest.py`` and ``__graft_entry__.smoke()`` call it; an application that trains through
    ``HipTrainer`` or Lightning should do the same at start-up."""
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return True
    import torch

    if torch.cuda.is_initialized():
        return False
    _os.environ["GPU_MAX_HW_QUEUES"] = str(int(hw_queues))
    return True
