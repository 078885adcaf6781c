"""cultionet_amd: MI355X-native TowerUNet hot path behind cultionet's module / LightningModule surface."""
import os as _os

# The step runs on three HIP streams (compute, weight gradients, RCCL buckets) plus RCCL's own; with the runtime's default
# of 4 hardware queues two of them can land on ONE queue and serialise (measured: 23.5 instead of 19.4 ms per bf16 step as
# soon as a process group exists). Must be set before the HIP runtime initialises, i.e. before the first torch.cuda call.
# Round 3: 16 -- with the host-feed copy stream (cultionet_amd/feeder.py) a fifth stream shared a queue with the compute
# stream at 8 (346 instead of 368 chips/s with a fresh batch per step).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
