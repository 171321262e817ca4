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


def test_column_lr_table_is_cached_per_value_set():
    """FusedCPUAdam._col_lr: one tensor per set of learning rates (its device copy must not be
    rebuilt -- and re-uploaded, which blocks the host -- every batch), refreshed when a rate changes."""
    from clm_gs_amd.cpu_adam import FusedCPUAdam
    p = torch.nn.Parameter(torch.zeros(5, 48))
    opt = FusedCPUAdam([p], columns_sizes=[3, 45], columns_lr=[0.0025, 0.000125],
                       state_tensors=(torch.zeros(5, 48), torch.zeros(5, 48)))  # no pinned allocation on CPU
    dev = torch.device("cpu")
    a = opt._col_lr(dev)
    assert a.shape == (48,) and torch.allclose(a[:3], torch.tensor(0.0025)) and torch.allclose(a[3:], torch.tensor(0.000125))
    assert opt._col_lr(dev) is a
    opt.columns_lr[0] = 0.001
    b = opt._col_lr(dev)
    assert b is not a and torch.allclose(b[:3], torch.tensor(0.001)) and torch.allclose(b[3:], torch.tensor(0.000125))


def test_bucket_size_properties():
    """Capacity buckets of the data-dependent buffers: >= n, at most 12.5 % more (beyond the 4096
    floor), monotone, idempotent, and few distinct values over a +-5 % spread of sizes."""
    from clm_gs_amd.gsplat import bucket_size
    prev = 0
    for n in list(range(0, 20000, 37)) + [10 ** 6 + k * 9973 for k in range(200)] + [8631784, 12085881, 2 ** 31 - 1]:
        b = bucket_size(n)
        assert b >= n and bucket_size(b) == b
        if n > 4096:
            assert b <= n * 1.125 + 1
    for n in range(0, 3_000_000, 1013):
        b = bucket_size(n)
        assert b >= prev
        prev = b
    assert len({bucket_size(int(8.6e6 * (1 + 0.05 * (k - 10) / 10))) for k in range(21)}) <= 3


def test_row_permutation_helpers():
    """utils.gather_rows / select_rows (chunked replacements of t[order] / t[mask] for very long row
    tables) equal plain indexing whatever the chunk size; utils.morton_order is a permutation that puts
    points of the same ground-plane cell next to each other."""
    from clm_gs_amd import utils
    g = torch.Generator().manual_seed(5)
    for shape in ((1000,), (1000, 4), (1000, 16, 3)):
        t = torch.randn(*shape, generator=g)
        order = torch.randperm(1000, generator=g)
        mask = torch.rand(1000, generator=g) < 0.3
        for chunk in (1, 7, 128, 1000, 4096):
            assert torch.equal(utils.gather_rows(t, order, chunk=chunk), t[order])
            assert torch.equal(utils.select_rows(t, mask, chunk=chunk), t[mask])
    assert utils.select_rows(torch.zeros(0, 4), torch.zeros(0, dtype=torch.bool)).shape == (0, 4)
    xyz = torch.rand(5000, 3, generator=g) * torch.tensor([100.0, 100.0, 3.0])
    perm = utils.morton_order(xyz)
    assert torch.equal(torch.sort(perm).values, torch.arange(5000))
    p = xyz[perm][:, :2]
    step_sorted = (p[1:] - p[:-1]).norm(dim=1).median()
    step_random = (xyz[1:, :2] - xyz[:-1, :2]).norm(dim=1).median()
    assert step_sorted < 0.1 * step_random  # consecutive rows are spatial neighbours


def test_intersection_capacity_hysteresis():
    """fused._capacity_for: the capacity the per-camera intersection lists are built for (device-count forms).
    None until a count has been seen; at least margin x the largest recent count; unchanged while the count drifts by a
    few per cent (no reallocation, no hipMalloc); regrown before the count can reach it; shrunk once the scene has
    become much sparser; off with device_side_counts=False."""
    from types import SimpleNamespace
    from clm_gs_amd import fused
    key = ("test", 1)
    fused._CAPACITY.pop(key, None)
    fused._CAP_HELD.pop(key, None)
    a = SimpleNamespace(device_side_counts=True, isect_capacity_margin=1.25, isect_capacity_floor=4096)
    assert fused._capacity_for(key, a) is None
    fused._observe_count(key, 1_000_000)
    cap0 = fused._capacity_for(key, a)
    assert cap0 >= 1_250_000 + 4096 and cap0 <= 1.13 * (1_250_000 + 4096)
    changes, cap, n = 0, cap0, 1_000_000
    for _ in range(200):                       # +0.1 % per camera: 22 % in all
        n = int(n * 1.001)
        fused._observe_count(key, n)
        c = fused._capacity_for(key, a)
        assert c >= int(n * 1.10)              # always room for the next camera's growth
        changes += c != cap
        cap = c
    assert changes <= 2                        # not one reallocation per 1/8-octave bucket
    for _ in range(400):                       # the scene thins out (pruning): the decaying maximum follows, 2 % per camera
        fused._observe_count(key, 100_000)
    small = fused._capacity_for(key, a)
    assert small < cap / 4 and small >= 125_000
    assert fused._capacity_for(key, SimpleNamespace(device_side_counts=False)) is None
    fused._CAPACITY.pop(key, None)
    fused._CAP_HELD.pop(key, None)
