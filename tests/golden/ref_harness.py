"""BUILD-CONTAINER ONLY: runs the reference's own Python (strategies/, densification, optimizer,
arguments under /root/reference) on the CPU with the absent native submodules replaced by this
repo's oracle, so that the ORCHESTRATION the reference spells out (which op is called with what,
how gradients accumulate over a batch, the optimizer hyper-parameter scaling, the densification
control flow) produces fixtures the HIP engines are compared with.

Nothing here is imported by the package or by the tests that run on the GPU box; the fixtures it
writes (tests/golden/*.npz) are data.  It does NOT stand in for the reference's kernels: the fake
`gsplat` / `clm_kernels` are oracle/gs_oracle.py, so a fixture pins "reference engine code driving
the oracle's arithmetic", not the arithmetic of the absent CUDA sources (that stays unpinned).

What the harness fakes, and why:
  * modules gsplat, clm_kernels, cpu_adam, fast_tsp, simple_knn._C, plyfile, numba(.cuda),
    torchvision, cv2, imageio, PIL-free stubs: absent third-party dependencies;
  * device strings: every "cuda" the reference hard-codes is rewritten to "cpu" by a
    TorchFunctionMode, Tensor.cuda()/pin_memory() are identities; torch.cuda.nvtx / synchronize /
    empty_cache / memory queries are no-ops;
  * torch.compile is disabled (no Triton here).
"""
import contextlib
import os
import sys
import types

import torch
from torch.overrides import TorchFunctionMode

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _is_cuda_dev(x):
    if isinstance(x, str):
        return x.startswith("cuda")
    if isinstance(x, torch.device):
        return x.type == "cuda"
    return False


class CudaToCpu(TorchFunctionMode):
    """device='cuda' -> 'cpu' for every torch call made while active.  claim_cuda: while True,
    `tensor.is_cuda` answers True (optimizer.py:113-119 asserts it while sorting the parameter groups)."""
    claim_cuda = False

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        name = getattr(func, "__name__", "")
        if self.claim_cuda and name == "__get__" and getattr(func, "__self__", None) is torch.Tensor.is_cuda:
            return True
        if name in ("cuda", "pin_memory") and args and isinstance(args[0], torch.Tensor):
            return args[0]
        if name == "is_pinned":
            return True
        if "device" in kwargs and _is_cuda_dev(kwargs["device"]):
            kwargs["device"] = "cpu"
        if "pin_memory" in kwargs:
            kwargs["pin_memory"] = False
        args = tuple("cpu" if _is_cuda_dev(a) else a for a in args)
        return func(*args, **kwargs)


def _noop(*a, **k):
    return None


class _FakeEvent:
    def __init__(self, *a, **k):
        pass

    record = wait = synchronize = _noop

    def elapsed_time(self, other):
        return 0.0

    def query(self):
        return True


class _FakeStream:
    def __init__(self, *a, **k):
        pass

    wait_stream = wait_event = synchronize = _noop

    def record_event(self, ev=None):
        return ev or _FakeEvent()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def install_stubs(oracle_gsplat=True):
    """Insert the fake third-party modules into sys.modules and neutralise torch.cuda side calls.
    Returns the oracle module that backs gsplat."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    if REF not in sys.path:
        sys.path.insert(1, REF)
    from oracle import gs_oracle as O

    gs = types.ModuleType("gsplat")
    for n in ("fully_fused_projection", "spherical_harmonics", "isect_tiles", "isect_offset_encode",
              "rasterize_to_pixels"):
        setattr(gs, n, getattr(O, n))
    sys.modules["gsplat"] = gs

    from oracle import clm_oracle as CO
    ck = types.ModuleType("clm_kernels")
    ck.fused_ssim = O.fused_ssim
    for n in ("send_shs2gpu_stream", "send_shs2gpu_stream_retention", "send_shs2cpu_grad_buffer_stream",
              "send_shs2cpu_grad_buffer_stream_retention", "spherical_harmonics_bwd_inplace", "scatter_to_bit",
              "extract_ffs", "compute_cnt_h", "set_signal"):
        setattr(ck, n, getattr(CO, n))

    def selective_adam_update(param, grad, m, v, visibility, lr, b1, b2, eps, N, M):
        O.selective_adam(param, grad, m, v, visibility, lr, b1, b2, eps)
    ck.selective_adam_update = selective_adam_update
    sys.modules["clm_kernels"] = ck

    ca = types.ModuleType("cpu_adam")
    ca.FusedCPUAdam = CO.FusedCPUAdam
    ca.CPUAdam = CO.CPUAdam
    sys.modules["cpu_adam"] = ca
    ft = types.ModuleType("fast_tsp")
    ft.find_tour = CO.find_tour
    sys.modules["fast_tsp"] = ft
    for name in ("plyfile", "numba", "numba.cuda", "torchvision", "torchvision.utils",
                 "cv2", "imageio", "simple_knn", "simple_knn._C", "psutil_stub"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = _noop
    sys.modules["simple_knn"]._C = sys.modules["simple_knn._C"]
    sys.modules["numba"].cuda = sys.modules["numba.cuda"]
    sys.modules["numba.cuda"].pinned_array = lambda shape, dtype=None: __import__("numpy").zeros(shape, dtype)
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]

    torch._dynamo.config.disable = True
    nv = torch.cuda.nvtx
    nv.range_push = nv.range_pop = nv.mark = _noop
    for n in ("synchronize", "empty_cache", "reset_peak_memory_stats"):
        setattr(torch.cuda, n, _noop)
    for n in ("memory_allocated", "max_memory_allocated", "memory_reserved", "max_memory_reserved"):
        setattr(torch.cuda, n, lambda *a, **k: 0)
    torch.cuda.Stream = _FakeStream
    torch.cuda.Event = _FakeEvent
    torch.cuda.current_stream = lambda *a, **k: _FakeStream()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    return O


def reference_default_args(**over):
    """The reference's own argparse defaults (arguments/__init__.py:22-59 ParamGroup machinery;
    train.py:849-870 builds the same six groups)."""
    from argparse import ArgumentParser

    import arguments as A
    parser = ArgumentParser()
    groups = [A.AuxiliaryParams(parser), A.ModelParams(parser), A.PipelineParams(parser),
              A.OptimizationParams(parser), A.BenchmarkParams(parser), A.DebugParams(parser)]
    args = parser.parse_args([])
    for k, v in over.items():
        assert hasattr(args, k), k
        setattr(args, k, v)
    return args, groups


class NullLog:
    def write(self, s):
        return len(s)

    def flush(self):
        pass


class RefCamera:
    """Exactly the attributes the engines read off scene.cameras.Camera (scene/cameras.py:39-126,
    train.py:278-312) for a given world->camera matrix; no image decoding."""

    def __init__(self, uid, w2c, fovx, fovy, width, height, image_u8):
        import math
        self.uid = uid
        self.FoVx, self.FoVy = float(fovx), float(fovy)
        self.image_width, self.image_height = int(width), int(height)
        self.image_name = f"cam_{uid:05d}"
        w2c = torch.as_tensor(w2c, dtype=torch.float32)
        self.world_view_transform = w2c.t().contiguous()
        self.original_image = image_u8
        self._math = math
        self.K = self.create_k_on_gpu()
        self.camtoworlds = torch.inverse(w2c)[None]

    def create_k_on_gpu(self):
        m = self._math
        fx = self.image_width / (2 * m.tan(self.FoVx * 0.5))
        fy = self.image_height / (2 * m.tan(self.FoVy * 0.5))
        return torch.tensor([[fx, 0, self.image_width / 2.0], [0, fy, self.image_height / 2.0], [0, 0, 1]])
