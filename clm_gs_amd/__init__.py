"""clm_gs_amd -- MI355X (gfx950) native drop-in for the CLM-GS training hot path.

Python stays the host language (as in the reference); every operator the
reference engines import from gsplat / clm_kernels / cpu_adam / fast_tsp is
served by hand-written HIP kernels behind the C ABI in ``include/clmgs.h``
(``libclmgs_hip.so``).  There is no CPU fallback: importing an operator module
without the built library raises.
"""

__version__ = "0.1.0"

from .runtime_env import single_gpu_runtime_defaults as _defaults

_defaults()  # hardware-queue default for single-GPU processes (runtime_env.py); a no-op once HIP has initialised
