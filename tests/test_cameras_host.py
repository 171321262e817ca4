"""The host constants a Camera caches at construction are exactly what fused._cam_host reads
back from the device tensors (scene/cameras.py:39-126 convention)."""
import numpy as np
import torch


def test_cached_host_constants_equal_device_readback():
    from clm_gs_amd.cameras import Camera
    from clm_gs_amd.fused import _cam_host
    g = torch.Generator().manual_seed(0)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    w2c = torch.eye(4)
    w2c[:3, :3] = q
    w2c[:3, 3] = torch.randn(3, generator=g)
    cam = Camera(3, w2c, 1.1, 0.7, 640, 480, device="cpu")
    cached = cam._clmgs_host
    del cam._clmgs_host
    fresh = _cam_host(cam)
    for a, b in zip(cached, fresh):
        assert a.dtype == np.float32 and a.shape == b.shape
        np.testing.assert_array_equal(a, b)
