"""Process-level runtime defaults of the product, applied BEFORE the HIP runtime initialises (no torch import here).

`GPU_MAX_HW_QUEUES=2` for single-GPU processes: the camera pipeline keeps three HIP streams by kernel type busy
(front end / memory-bound / tile kernels); with the runtime's default of four hardware queues the 28 M headline
workload measured 153.6-155.6 img/s, with two 155.8-157.9 (three interleaved rounds).  Multi-GPU
ranks keep the runtime default (RCCL's own streams want their queues).  An exported value always wins.

Called by the product's entry points only: bench.py at the top of the process, and `clm_gs_amd/__init__.py` when the
process is `python -m clm_gs_amd.trainer`.  A library user's `import clm_gs_amd` does not touch the environment.
"""
import os


def single_gpu_runtime_defaults():
    """Single-GPU = no launcher variable says otherwise (torchrun exports WORLD_SIZE / LOCAL_WORLD_SIZE / RANK; a rank
    started by hand with only RANK set is treated as multi-GPU too: the runtime default is the safe side)."""
    multi = (int(os.environ.get("WORLD_SIZE", "1")) > 1 or int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > 1
             or int(os.environ.get("RANK", "0")) > 0)
    if not multi:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
    return os.environ.get("GPU_MAX_HW_QUEUES")
