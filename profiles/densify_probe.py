"""Where a densification at full size spends its time (python profiles/densify_probe.py [n_gaussians]): the bench scene,
a few training batches for the statistics, then gsplat_densification + spatial_sort with a device synchronisation
around every model method (flush_lazy_rows, densify_and_clone, densify_and_split, prune_points, permute_rows,
reset_opacity) -- twice, so that the second round shows the cost with a warm allocator and loaded device code; the model is
set up as trainer.training sets it up (row tables with 5 % head room, the re-sort fused into the prune)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from clm_gs_amd import utils
from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload, clm_offload_train_one_batch
from clm_gs_amd.synthetic import nadir_cameras, perturbed_copy, synth_gaussians

N = int(sys.argv[1]) if len(sys.argv) > 1 else 28_000_000
W, H, bsz = 4608, 3456, 4
args = utils.default_args(bsz=bsz, sh_residency="hbm", densify_from_iter=0, densification_interval=16,
                          densify_until_iter=10_000, opacity_reset_interval=32, prealloc_capacity=int(N * 1.05) // 16 * 16)
args.clm_offload = True
utils.set_args(args)
utils.set_img_size(H, W)
sc = synth_gaussians(N, seed=0, device="cuda")
cams = nadir_cameras(16, N, W, H, 0.10, seed=0, device="cuda")
g = torch.Generator().manual_seed(1)
for c in cams:
    c.original_image = (torch.rand(3, H, W, generator=g) * 255).to(torch.uint8).cuda()
m = GaussianModelCLMOffload(3)
m.create_from_tensors(sc["xyz"], sc["shs48"], sc["scaling"], sc["rotation"], sc["opacity"], spatial_lr_scale=sc["lr_extent"])
extent = sc["extent"]
del sc
m.active_sh_degree = 3
m.training_setup(args)
m.spatial_sort()
m.fuse_sort_into_prune = True  # as trainer.training sets it: the prune's compaction also re-sorts


class _Scene:
    cameras_extent = extent


comm, gen = torch.cuda.Stream(), torch.Generator(device="cuda").manual_seed(1)
times = {}


def timed(name, fn):
    def wrap(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        times.setdefault(name, []).append(round(time.perf_counter() - t0, 4))
        return r
    return wrap


for name in ("flush_lazy_rows", "flush_small", "densify_and_clone", "densify_and_split", "prune_points", "permute_rows",
             "reset_opacity", "densify_and_prune", "spatial_sort", "_regather_row_tables", "_append_rows"):
    setattr(m, name, timed(name, getattr(m, name)))
from clm_gs_amd.densification import gsplat_densification
it = 1
out = []
for rnd in range(2):
    for b in range(4):
        utils.set_cur_iter(it)
        m.update_learning_rate(it)
        batch = cams[(b % 4) * bsz:(b % 4 + 1) * bsz]
        clm_offload_train_one_batch(m, _Scene, batch, m.parameters_grad_buffer, None, None, comm, gen)
        it += bsz
    torch.cuda.synchronize()
    times.clear()
    ms0 = torch.cuda.memory_stats()
    n0 = m.get_xyz.shape[0]
    t0 = time.perf_counter()
    utils.set_cur_iter(it - bsz)
    gsplat_densification(16 * (rnd + 1), _Scene, m, None)
    m.spatial_sort()
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats()
    out.append({"round": rnd, "total_s": round(tot, 4), "n_before_after": [n0, int(m.get_xyz.shape[0])],
                "device_mallocs": int(ms1["num_device_alloc"] - ms0["num_device_alloc"]),
                "peak_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "by_method_s": dict(times)})
print(json.dumps(out))
