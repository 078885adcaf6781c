"""cultionet_amd: MI355X-native TowerUNet hot path behind cultionet's module / LightningModule surface."""
import os as _os

# The step runs on three HIP streams (compute, weight gradients, RCCL buckets) plus RCCL's own; with the runtime's default
# of 4 hardware queues two of them can land on ONE queue and serialise (measured: 23.5 instead of 19.4 ms per bf16 step as
# soon as a process group exists). Must be set before the HIP runtime initialises, i.e. before the first torch.cuda call.
# (Round 3: 16 queues were tried -- the host-feed copy stream then no longer shares a queue under a one-rank RCCL group
# (346 -> 367 chips/s with a fresh batch per step), but the second configuration of a process, bench.py's bf16 block,
# dropped from 1937 to 1611 chips/s: which streams end up on one hardware queue depends on the creation history of the
# process, and 8 is the setting under which every block of the default bench line matches its stand-alone run.)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
