"""Process-level runtime defaults of the product, applied BEFORE the HIP runtime initialises (no torch import here).

`GPU_MAX_HW_QUEUES=2` for single-GPU processes: the camera pipeline keeps three HIP streams by kernel type busy
(front end / memory-bound / tile kernels); with the runtime's default of four hardware queues the 28 M headline
workload measured 153.6-155.6 img/s, with two 155.8-157.9 (three interleaved rounds).  Multi-GPU
ranks keep the runtime default (RCCL's own streams want their queues).  An exported value always wins.

Called by bench.py and by `python -m clm_gs_amd.trainer` at the top of the process; `clm_gs_amd/__init__.py` calls it
too, which covers library users as long as the package is imported before the first HIP call.
"""
import os


def single_gpu_runtime_defaults():
    if int(os.environ.get("WORLD_SIZE", "1")) == 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
    return os.environ.get("GPU_MAX_HW_QUEUES")
