"""Thin training loop around the engines (scope row f1; reference: train.py:68-666).

Keeps what the published numbers depend on: the image-strided iteration counter
(train.py:202-204), xyz LR schedule per image index, SH-degree ramp every 1000 images
(:253-254), the engine call (:316-432), gsplat_densification (:452), the no_offload optimizer
epilogue (:533-578), the end-to-end timer that pauses during evaluation (:438-447, utils/timer.py:
87-111), and the exact log strings release_scripts/log2csv.py:54-102 scrapes.  `training` takes cameras
as objects; `train_from_colmap` (and `python -m clm_gs_amd.trainer -s ... -m ...`) builds them from a COLMAP
directory (colmap_scene.py).  MatrixCity / Blender readers, checkpoint directories and the rest of the
reference's CLI are out of scope.
"""
import gc
import os
import random
import time

import torch

from . import dp, utils
from .densification import gsplat_densification


class End2endTimer:
    """utils/timer.py:87-111: wall clock that can be paused; throughput = iterations / total."""

    def __init__(self):
        self.total_time = 0.0
        self.last_time_point = None

    def start(self):
        torch.cuda.synchronize()
        self.last_time_point = time.time()

    def stop(self):
        torch.cuda.synchronize()
        self.total_time += time.time() - self.last_time_point
        self.last_time_point = None

    def print_time(self, log_file, n_iterations):
        log_file.write("end2end total_time: {:.3f} s, iterations: {}, throughput {:.2f} it/s\n".format(
            self.total_time, n_iterations, n_iterations / max(self.total_time, 1e-9)))


def clm_hbm_only(args):
    return bool(getattr(args, "clm_offload", False)) and getattr(args, "sh_residency", "hbm") == "hbm"


def psnr(img1, img2):
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def _pinned_gb(gaussians):
    tot = 0
    names = ("parameters_buffer", "parameters_grad_buffer", "_exp_avg_buffer", "_exp_avg_sq_buffer")
    if hasattr(gaussians, "_small"):  # naive_offload: its own two pinned tables (+ grads are transient)
        names = ("_small", "_parameters")
    for name in names:
        t = getattr(gaussians, name, None)
        if isinstance(t, torch.Tensor) and t.numel() and not t.is_cuda:
            tot += t.numel() * t.element_size()
    return tot / 2 ** 30


def memory_line(iteration, bsz, gaussians, what="densify_and_prune"):
    """utils/general_utils.py:216-240 format (the part log2csv.py reads)."""
    return ("iteration[{},{}) {}. Now num of 3dgs: {}. Now Memory usage: {} GB. Max Memory usage: {} GB. "
            "Now Pinned Memory: {} GB\n").format(
        iteration, iteration + bsz, what, gaussians.get_xyz.shape[0],
        torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.max_memory_allocated() / 2 ** 30,
        _pinned_gb(gaussians))


@torch.no_grad()
def evaluate(name, iteration, cameras, render_fn, log_file, max_images=10 ** 9):
    """train.py:669-846 in essence: mean L1 / PSNR over a camera set."""
    l1s, ps = [], []
    for cam in cameras[:max_images]:
        img = torch.clamp(render_fn(cam), 0.0, 1.0)
        gt = torch.clamp(cam.original_image.float() / 255.0, 0.0, 1.0)
        l1s.append((img - gt).abs().mean().item())
        ps.append(psnr(img[None], gt[None]).mean().item())
    l1, p = sum(l1s) / len(l1s), sum(ps) / len(ps)
    log_file.write("[ITER {}] Evaluating {}: L1 {} PSNR {}\n".format(iteration, name, l1, p))
    return l1, p


class _LossLog:
    """The `iteration[a,b) loss: ...` line of a batch needs the batch's loss VALUES on the host.  The reference reads
    them right after the batch (train.py: torch.stack(losses).cpu()), which drains the device before the next batch is
    enqueued; here the values travel on a side stream behind an event recorded after the batch, and the line is written
    when the NEXT batch has been enqueued (same text, same order): the host never waits for a device that has nothing
    queued behind it.  `defer_loss_log=False` in the args restores the immediate read."""

    def __init__(self, log_file, defer):
        self.log_file, self.defer, self.pending, self.stream = log_file, defer, None, None
        self.hosts, self.turn = {}, 0  # two pinned buffers per batch size, used alternately (no allocation per batch)

    def _write(self, head, host_vals, tail):
        self.log_file.write(head + " ".join("%.6f" % l for l in host_vals.tolist()) + tail)

    def push(self, head, losses, tail):
        if not self.defer:
            self._write(head, torch.stack(losses).cpu(), tail)
            return
        prev = self.pending
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        stacked = torch.stack(losses)  # enqueued on the current stream, right behind the batch
        ev2 = torch.cuda.Event()
        ev2.record(torch.cuda.current_stream())
        self.turn ^= 1
        key = (tuple(stacked.shape), self.turn)
        host = self.hosts.get(key)
        if host is None:
            host = self.hosts[key] = torch.empty(stacked.shape, dtype=stacked.dtype, pin_memory=True)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev2)
            host.copy_(stacked, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        stacked.record_stream(self.stream)
        self.pending = (head, host, tail, done)
        if prev is not None:
            self._flush(prev)

    def _flush(self, item):
        head, host, tail, done = item
        done.synchronize()
        self._write(head, host, tail)

    def flush(self):
        if self.pending is not None:
            self._flush(self.pending)
            self.pending = None


def _warm_structural_ops(device, n=70_000):
    """The torch operators of a densification (masks, selections, concatenations, the split's sampling, the Z-order
    keys) run ONCE on small tensors.  ROCm loads device code lazily, at the first launch out of each code object: the
    first densify_and_prune of a process paid 110-130 ms for that on top of its 30 ms of work (bench.py --trainer-trace),
    inside the end-to-end clock.  SETUP, like the allocator reservation below; private generator (the global random
    stream and the model's split generator are not touched), nothing of the model is read or written."""
    from .utils import build_rotation, inverse_sigmoid, morton_order, select_rows, take_rows
    with torch.no_grad():
        gen = torch.Generator(device=device)
        gen.manual_seed(0)
        acc = torch.rand((n, 1), device=device, generator=gen)
        den = (torch.rand((n, 1), device=device, generator=gen) > 0.3).float()
        mr = torch.zeros((n,), device=device)
        d4 = torch.rand((n, 4), device=device, generator=gen)
        mr = torch.maximum(mr, d4[:, 0]); acc = acc + d4[:, 1:2]; den = den + d4[:, 2:3]; d4.zero_()
        grads = acc / den
        grads[grads.isnan()] = 0.0
        xyz = torch.randn((n, 3), device=device, generator=gen)
        scal = torch.randn((n, 3), device=device, generator=gen) - 3.0
        rot = torch.randn((n, 4), device=device, generator=gen)
        opa = torch.randn((n, 1), device=device, generator=gen)
        shs = torch.rand((n, 48), device=device, generator=gen)
        sel = torch.norm(grads, dim=-1) >= 0.5
        sel &= torch.max(torch.exp(scal), dim=1).values <= 0.06
        k = int(sel.sum())
        new = [select_rows(t, sel) for t in (xyz, shs, opa, scal, rot)]
        cat = [torch.cat((t, e), dim=0) for t, e in zip((xyz, shs, opa, scal, rot), new)]
        mom = torch.cat((torch.zeros_like(xyz), torch.zeros_like(new[0])), dim=0)
        padded = torch.zeros((n + k,), device=device)
        padded[:n] = grads.squeeze()
        sel2 = padded >= 0.5
        sel2 &= torch.max(torch.exp(cat[3]), dim=1).values > 0.06
        stds = select_rows(torch.exp(cat[3]), sel2).repeat(2, 1)
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=gen)
        rots = build_rotation(select_rows(cat[4], sel2)).repeat(2, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + select_rows(cat[0], sel2).repeat(2, 1)
        new_scal = torch.log(select_rows(torch.exp(cat[3]), sel2).repeat(2, 1) / 1.6)
        prune = torch.cat((sel2, torch.zeros(2 * int(sel2.sum()), device=device, dtype=torch.bool)))
        xyz2 = torch.cat((cat[0], new_xyz), dim=0)
        opa2 = torch.cat((cat[2], select_rows(cat[2], sel2).repeat(2, 1)), dim=0)
        scal2 = torch.cat((cat[3], new_scal), dim=0)
        mask = (torch.sigmoid(opa2) < 0.005).squeeze()
        mask = torch.logical_or(mask, torch.exp(scal2).max(dim=1).values > 0.1)
        mask = torch.logical_or(mask, prune)
        keep = ~mask
        m = int(keep.sum())
        idx = torch.nonzero(keep).flatten()
        idx = take_rows(idx, morton_order(take_rows(xyz2, idx)))
        stamps = torch.zeros((xyz2.shape[0],), dtype=torch.int32, device=device)
        stamps[:m] = 7
        new_opa = inverse_sigmoid(torch.min(torch.sigmoid(opa2), torch.ones_like(opa2) * 0.01))
        _ = (torch.zeros_like(new_opa), mom, mr, idx, stamps)
    torch.cuda.synchronize()


def training(gaussians, scene, train_cameras, test_cameras, log_file, iterations=None,
             test_iterations=(), background=None, shuffle_seed=0, phase_times=None):
    """Runs `iterations` images of training; returns the End2endTimer.
    `phase_times` (optional dict): host wall time per phase inside the end-to-end clock is accumulated into it
    ("engine" = enqueueing the batches, "densify" = gsplat_densification incl. the device work it waits for,
    "resort" = the Z-order re-sort after a densification, "log" = waiting for loss values, "final_sync"); a callable under
    "iter_hook" is removed from the dict and called with the image counter after every iteration (diagnosis).

    Camera-DP: when a process group is up every rank runs this same loop over the
    same shuffled order, takes cameras rank::ranks of each GLOBAL batch (bsz x ranks images), and
    the image counter strides by the global batch; the engine does the one exchange per batch,
    densification reduces its statistics (densification.py) and draws identical split samples."""
    args = utils.get_args()
    bsz = args.bsz
    ws, rk = dp.world_size(), dp.rank()
    gbsz = bsz * ws
    if ws > 1:
        assert clm_hbm_only(args), "camera-DP trains clm_offload with sh_residency='hbm'"
        if getattr(gaussians, "split_generator", None) is None:
            dp.seed_split_generator(gaussians)
    iterations = iterations or args.iterations
    utils.set_log_file(log_file)
    clm = bool(getattr(args, "clm_offload", False))
    naive = bool(getattr(args, "naive_offload", False))
    if naive:
        from .strategies.naive_offload import naive_offload_eval_one_cam, naive_offload_train_one_batch
        render_fn = lambda cam: naive_offload_eval_one_cam(gaussians, scene, cam, background)
    elif clm:
        from .strategies.clm_offload import clm_offload_eval_one_cam, clm_offload_train_one_batch
        comm_stream = torch.cuda.Stream()
        perm_generator = torch.Generator(device="cuda")
        perm_generator.manual_seed(1)
        render_fn = lambda cam: clm_offload_eval_one_cam(cam, gaussians, background, scene)
    else:
        from .strategies.no_offload import baseline_accumGrads_impl, baseline_accumGrads_micro_step

        def render_fn(cam):
            img, _, _, _ = baseline_accumGrads_micro_step(
                gaussians.get_xyz, gaussians.get_opacity, gaussians.get_scaling, gaussians.get_rotation,
                gaussians.get_features, gaussians.active_sh_degree, cam, background, mode="test")
            return img
    spatial = bool(getattr(args, "spatial_row_order", True)) and hasattr(gaussians, "spatial_sort") and not naive
    if spatial:
        gaussians.spatial_sort()  # rows along a Z-order curve: a camera's rows become contiguous runs
        # the prune that ends a densification re-sorts in the same pass over the row tables (clm_offload model, HBM rows)
        gaussians.fuse_sort_into_prune = clm_hbm_only(args)
    # a full cyclic GC pass over torch's ~10^6 long-lived objects stalls the enqueueing thread for
    # ~100 ms: freeze what exists now, later collections only see what the loop allocates
    gc.collect()
    gc.freeze()
    rng = random.Random(shuffle_seed)
    order = []
    # camera-DP, locality exchange (dp.py): every rank draws its batches from the cameras that look mostly at
    # the rows it owns (index ranges of the Z-ordered tables = spatial regions); the deal is recomputed when
    # densification has grown the model by 10 % (rows are re-sorted after every densification)
    locality = ws > 1 and bool(getattr(args, "dp_locality", False))
    pool, dealt_at = None, 0

    def deal():
        nonlocal pool, dealt_at, order
        ranks_of, _ = dp.deal_cameras(train_cameras, gaussians, ws)
        dealt_at, order = gaussians.get_xyz.shape[0], []
        # every rank computes the same deal, so every rank sees the same smallest pool -- no collective needed to agree.
        # A rank with fewer than bsz cameras could not draw a batch while the others are already inside the batch's
        # collectives (hang): then ALL ranks fall back to the strided deal of the global shuffled order (the locality
        # exchange is correct for any camera assignment, it just moves more border rows).
        smallest = min(sum(1 for q in ranks_of if q == r_) for r_ in range(ws))
        if smallest < bsz:
            pool = None
            log_file.write("camera-DP locality deal: smallest pool {} < bsz {} ({} cameras / {} ranks): strided deal\n".format(
                smallest, bsz, len(train_cameras), ws))
        else:
            pool = [i for i, q in enumerate(ranks_of) if q == rk]
    if locality:
        assert spatial, "dp_locality needs the Z-ordered row tables (spatial_row_order)"
        deal()
    from . import _lib

    def _check_device():
        # the host has just synchronised: an asynchronous device fault / a timed-out pass of the profiling build's
        # look-back binning route surfaces HERE, not at the end of the run with the model unsaved
        _lib.check_device_errors()
    defer = bool(getattr(args, "defer_loss_log", True))
    loss_log = _LossLog(log_file, defer)
    pt = phase_times if phase_times is not None else {}
    iter_hook = pt.pop("iter_hook", None)  # diagnosis only (bench.py --trainer-trace): called after every iteration

    def _acc(key, t_start):
        pt[key] = pt.get(key, 0.0) + time.perf_counter() - t_start
    if clm_hbm_only(args) and getattr(args, "allocator_reservoir", True) and getattr(args, "fused_front_end", True):
        # allocator warm-up: one block per stream pool instead of dozens of hipMalloc calls spread over the first batches
        # and every densification (strategies/clm_offload/engine.py reserve_working_set).  SETUP, like the reference's
        # --prealloc_capacity buffers (train.py:107-115): before the end-to-end clock starts (utils/timer.py:87-111 starts
        # it at the first iteration); its own wall time is reported as phase "reserve".  (Fresh device memory costs
        # ~0.1 s per GB on some boxes and nothing on others -- measured 0.002 .. 1.3 s for the same 10 GB.)
        from .strategies.clm_offload.engine import reserve_working_set
        _t = time.perf_counter()
        pt["reserved_bytes"] = float(sum(reserve_working_set(gaussians).values()))
        torch.cuda.synchronize()
        _acc("reserve", _t)
        if not args.disable_auto_densification and getattr(args, "warm_structural_ops", True):
            _t = time.perf_counter()
            _warm_structural_ops(gaussians._xyz.device)
            _acc("warm_ops", _t)
    if torch.cuda.is_available():  # (diagnosis: how many hipMallocs happen INSIDE the end-to-end clock -- bench.py reports it)
        pt["device_mallocs_before_clock"] = float(torch.cuda.memory_stats().get("num_device_alloc", 0))
    timer = End2endTimer()
    timer.start()
    next_batch = None
    for iteration in range(1, iterations + 1, gbsz):
        utils.set_cur_iter(iteration)
        gaussians.update_learning_rate(iteration)
        if utils.check_update_at_this_iter(iteration, gbsz, 1000, 0):
            gaussians.oneupSHdegree()
        def draw():
            nonlocal order
            if locality and pool is not None:
                if len(order) < bsz:  # new epoch of THIS rank's pool
                    order = list(pool)
                    rng.shuffle(order)
                return [train_cameras[order.pop()] for _ in range(bsz)]
            if len(order) < gbsz:  # new epoch: shuffle, drop_last (train.py:156-167)
                order = list(range(len(train_cameras)))
                rng.shuffle(order)
            return [train_cameras[order.pop()] for _ in range(gbsz)][rk::ws]
        # the loader is one batch ahead: the host-resident engine stages the next batch's untouched rows early
        batch = next_batch if next_batch is not None else draw()
        next_batch = draw() if iteration + gbsz <= iterations else None
        if clm and getattr(args, "sh_residency", "hbm") == "host":
            from .strategies.clm_offload.engine import hint_next_batch
            hint_next_batch(gaussians, next_batch)
        _t = time.perf_counter()
        if naive:
            losses, visibility = naive_offload_train_one_batch(gaussians, scene, batch, background,
                                                               sparse_adam=args.sparse_adam)
            names, sparsity = [c.image_name for c in batch], None
        elif clm:
            losses, ordered_cams, sparsity = clm_offload_train_one_batch(
                gaussians, scene, batch, gaussians.parameters_grad_buffer, background, None, comm_stream,
                perm_generator)
            names = [batch[i].image_name for i in ordered_cams]
        else:
            losses, visibility = baseline_accumGrads_impl(gaussians, scene, batch, background,
                                                          sparse_adam=args.sparse_adam)
            names, sparsity = [c.image_name for c in batch], None
        _acc("engine", _t)
        _t = time.perf_counter()
        loss_log.push("iteration[{},{}) loss: ".format(iteration, iteration + gbsz), losses,
                      " image: {}".format(names) + ((" sparsity: " + " ".join("%.4f" % s for s in sparsity) + "\n")
                                                   if sparsity else "\n"))
        _acc("log", _t)
        if any(iteration <= t < iteration + gbsz for t in test_iterations):
            loss_log.flush()
            timer.stop()  # evaluation is excluded from the throughput figure
            _check_device()
            if hasattr(gaussians, "flush_lazy_rows"):  # deferred row steps (and, owner-computes DP, the exchange)
                gaussians.flush_lazy_rows()
            evaluate("train", iteration, train_cameras, render_fn, log_file, max_images=5)
            if test_cameras:
                evaluate("test", iteration, test_cameras, render_fn, log_file)
            timer.start()
        n_before = gaussians.get_xyz.shape[0]
        _t = time.perf_counter()
        gsplat_densification(iteration, scene, gaussians, None)
        _acc("densify", _t)
        if gaussians.get_xyz.shape[0] != n_before:
            _check_device()  # (densify_and_prune read its counts back: the device is drained)
        if spatial and gaussians.get_xyz.shape[0] != n_before:
            _t = time.perf_counter()
            gaussians.spatial_sort()  # clones / splits were appended at the end of the tables
            _acc("resort", _t)
            if locality and abs(gaussians.get_xyz.shape[0] - dealt_at) > 0.1 * dealt_at:
                deal()
                next_batch = None  # drawn from the old pool
        if gaussians.get_xyz.shape[0] != n_before or utils.check_update_at_this_iter(
                iteration, gbsz, args.densification_interval, 0):
            loss_log.flush()  # keep the reference's line order: the batch's loss line, then the memory line
            log_file.write(memory_line(iteration, gbsz, gaussians))
        if not clm and not naive:  # train.py:533-578
            if args.lr_scale_mode != "accumu":
                for p in gaussians.all_parameters():
                    if p.grad is not None:
                        p.grad /= bsz
            if args.sparse_adam:
                gaussians.optimizer.step(visibility=visibility)
            else:
                gaussians.optimizer.step()
            gaussians.optimizer.zero_grad(set_to_none=True)
        if not defer:
            torch.cuda.synchronize()
        if iter_hook is not None:
            iter_hook(iteration)
    _t = time.perf_counter()
    loss_log.flush()
    timer.stop()
    _acc("final_sync", _t)
    _check_device()
    n_done = ((iterations - 1) // gbsz + 1) * gbsz + 1
    timer.print_time(log_file, n_done)
    log_file.write(memory_line(iteration, gbsz, gaussians, what="final"))
    log_file.write("Max Memory usage: {} GB.\n".format(torch.cuda.max_memory_allocated() / 2 ** 30))
    return timer


def train_from_colmap(source_path, model_path, strategy="clm_offload", iterations=None, eval=False, resolution=1,
                      images="images", test_iterations=(), save=True, **arg_overrides):
    """A COLMAP directory -> trained model (row f1): `colmap_scene.load_colmap_scene` (cameras, held-out
    split, scene radius, sparse points) -> `create_from_pcd` with `spatial_lr_scale = cameras_extent` ->
    `training_setup` -> `training` (log lines of the reference in `<model_path>/python_ws=1_rk=0.log`) ->
    `point_cloud/iteration_N/point_cloud.ply` (scene/__init__.py:41-143, train.py:60-131 for the order of
    these steps).  `strategy`: clm_offload | no_offload | naive_offload; `arg_overrides`: any flag of
    `utils.default_args`.  Returns (gaussians, scene, timer)."""
    from .colmap_scene import load_colmap_scene
    from .strategies.clm_offload import GaussianModelCLMOffload
    from .strategies.naive_offload import GaussianModelNaiveOffload
    from .strategies.no_offload import GaussianModelNoOffload
    assert strategy in ("clm_offload", "no_offload", "naive_offload"), strategy
    args = utils.default_args(**arg_overrides)
    setattr(args, strategy, True)
    args.source_path, args.model_path, args.eval = source_path, model_path, bool(eval)
    if iterations is not None:
        args.iterations = int(iterations)
    utils.set_args(args)
    scene = load_colmap_scene(source_path, images=images, eval=eval, resolution=resolution, device="cuda")
    if scene.point_cloud is None:
        raise ValueError(f"{source_path}/sparse/0 holds no points3D.bin / points3D.txt to initialise from")
    sizes = {(c.image_height, c.image_width) for c in scene.train_cameras + scene.test_cameras}
    assert len(sizes) == 1, f"all images must share one size (utils.get_img_width/height are global): {sizes}"
    h, w = next(iter(sizes))
    utils.set_img_size(h, w)
    utils.set_cur_iter(1)
    gaussians = {"clm_offload": GaussianModelCLMOffload, "no_offload": GaussianModelNoOffload,
                 "naive_offload": GaussianModelNaiveOffload}[strategy](args.sh_degree)
    gaussians.create_from_pcd(scene.point_cloud, scene.cameras_extent)
    gaussians.training_setup(args)
    os.makedirs(model_path, exist_ok=True)
    prev_log = utils.get_log_file()
    try:
        with open(os.path.join(model_path, "python_ws=1_rk=0.log"), "w") as log_file:
            timer = training(gaussians, scene, scene.train_cameras, scene.test_cameras, log_file,
                             iterations=args.iterations, test_iterations=test_iterations)
    finally:
        utils.set_log_file(prev_log)  # never leave a closed file as the process-wide log
    if save:
        if hasattr(gaussians, "flush_lazy_rows"):
            gaussians.flush_lazy_rows()  # ALL ranks (a collective under owner-computes / locality camera-DP) ...
        if dp.rank() == 0:               # ... then rank-0-only I/O (no collective left inside save_ply: nothing is dirty)
            gaussians.save_ply(os.path.join(model_path, "point_cloud", f"iteration_{args.iterations}", "point_cloud.ply"))
    return gaussians, scene, timer


if __name__ == "__main__":  # python -m clm_gs_amd.trainer -s <colmap dir> -m <output dir> [--clm_offload] ...
    # (clm_gs_amd/__init__.py has applied runtime_env.single_gpu_runtime_defaults() before torch was imported: the
    # hardware-queue default bench.py runs with is the trainer's too)
    import argparse
    ap = argparse.ArgumentParser(description="train a 3DGS model from a COLMAP directory (flag names of train.py)")
    ap.add_argument("-s", "--source_path", required=True)
    ap.add_argument("-m", "--model_path", required=True)
    ap.add_argument("-i", "--images", default="images")
    ap.add_argument("-r", "--resolution", type=float, default=1)
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--iterations", type=int, default=30000)
    ap.add_argument("--bsz", type=int, default=4)
    ap.add_argument("--test_iterations", type=int, nargs="*", default=[7000, 30000])
    grp = ap.add_mutually_exclusive_group()
    for s_ in ("clm_offload", "no_offload", "naive_offload"):
        grp.add_argument("--" + s_, action="store_true")
    ap.add_argument("--sh_residency", choices=["hbm", "host"], default="hbm")
    ap.add_argument("--sh_hbm_budget_gb", type=float, default=0.0,
                    help="sh_residency=host: HBM that keeps the first rows of the Z-ordered SH table resident (768 B per row)")
    a = ap.parse_args()
    strat = "no_offload" if a.no_offload else ("naive_offload" if a.naive_offload else "clm_offload")
    _, _, t = train_from_colmap(a.source_path, a.model_path, strategy=strat, iterations=a.iterations, eval=a.eval,
                                resolution=a.resolution, images=a.images, test_iterations=tuple(a.test_iterations),
                                bsz=a.bsz, sh_residency=a.sh_residency, sh_hbm_budget_gb=a.sh_hbm_budget_gb)
