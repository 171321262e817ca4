#!/usr/bin/env python
"""Headline benchmark: training images/s (+ peak GPU bytes) of the clm_offload hot path on a
seeded synthetic Rubble-4K-shaped scene (BASELINE.json: 28 M Gaussians, 4608x3456, bsz 4).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of `bsz` cameras
(clm_offload_train_one_batch: visibility filters -> per camera projection / SH / tile binning
+ sort / rasterize / loss / backward -> Adam).  N > 1: one process per GPU (launched by
torch.distributed.run), camera data parallel, weak scaling (every rank renders its own bsz
cameras; one gradient exchange per batch over RCCL).  Rank 0 prints ONE JSON line.

`vs_baseline` = value / the reference's own published figure for the same configuration and
strategy (derived there from published training time / iterations, measured
on the reference's testbed, 1x RTX 4090 -- different hardware, stated in `baseline.source`); null
for multi-GPU runs and for configurations the reference publishes nothing for.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _requested_gpus(argv):
    """--gpus N from the command line, before argparse (and before torch is imported)."""
    for i, tok in enumerate(argv):
        if tok == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if tok.startswith("--gpus="):
            return int(tok.split("=", 1)[1])
    return 1


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no RANK / WORLD_SIZE in the environment (the driver's plain
    command form): this process becomes the launcher -- it re-executes itself as N ranks, one per GPU, under
    torch.distributed.run (127.0.0.1 rendezvous on a free port), exactly the command the contract gives for
    N > 1; rank 0 prints the ONE JSON line on the inherited stdout.  Fewer than N visible devices is an error
    (CLMGS_SHARE_GPU=1, the test hook for several ranks on one device, lifts it)."""
    n = _requested_gpus(sys.argv[1:])
    if n <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return
    import socket
    import torch  # device count only: no context is created, and this process is replaced below
    have = torch.cuda.device_count()
    if have < n and os.environ.get("CLMGS_SHARE_GPU") != "1":
        sys.stderr.write(f"bench: --gpus {n} but only {have} GPU(s) are visible to this process "
                         f"(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?)\n")
        sys.exit(2)
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what the host driver supports (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


_self_launch()

# Two hardware queues for the process's HIP streams (the runtime's default is 4): the camera pipeline's three streams
# by kernel type then share two queues.  Measured interleaved, three rounds in one call (28 M, GT resident): 1 queue
# 147.2 / 146.9 / 148.2, 2 queues 155.8 / 157.9 / 157.0, 3: 153.0 / 155.4 / 155.7, 4: 154.5 / 154.5 / 155.6, default
# 153.6 / 154.0 / 155.5 img/s.  Single-GPU runs only (RCCL's own streams want their queues); an exported value wins.
# The same default is applied by the product's own entry points (clm_gs_amd.utils.single_gpu_runtime_defaults, called
# by `python -m clm_gs_amd.trainer` before HIP initialises), so a trainer run and a bench run share the configuration.
from clm_gs_amd.runtime_env import single_gpu_runtime_defaults  # noqa: E402  (no torch import inside)
single_gpu_runtime_defaults()

import torch  # noqa: E402

CONFIGS = {
    # name: (N gaussians, W, H, bsz, visible fraction per camera, description)
    "rubble28m": (28_000_000, 4608, 3456, 4, 0.10, "Rubble-4K 28M Gaussians, clm_offload, 1xMI355X (paper headline row)"),
    "rubble10m": (10_000_000, 4608, 3456, 4, 0.15, "Rubble-4K 10M Gaussians, clm_offload"),
    "bicycle6m": (6_000_000, 1237, 822, 4, 0.25, "MipNeRF360 Bicycle ~6M Gaussians"),
    "bigcity102m": (102_231_360, 1920, 1080, 8, 0.02, "BigCity Aerial 102M Gaussians, clm_offload, sparse Adam, "
                    "no densification; bsz 64 = 8 GPUs x 8 cameras (bigcity.sh:54-85)"),
    "small": (200_000, 640, 480, 4, 0.3, "CI-size smoke configuration"),
}

# algorithmic bytes per launch of each kernel: n rows in, V visible, I
# intersections, P pixels, T tiles.  Used for roofline.achieved = bytes / measured duration.
# I is the REFERENCE's intersection count (gsplat.isect_tiles: 3-sigma tile boxes) -- the work the
# algorithm defines; the list this build actually sorts and blends after exact per-tile culling
# is reported beside it as I_emitted_avg.
ALGO_BYTES = {
    "clmgs_projection_fwd": lambda n, V, I, P, T: 68 * n,
    "clmgs_projection_bwd": lambda n, V, I, P, T: 132 * n,
    "clmgs_sh_fwd": lambda n, V, I, P, T: 216 * V,
    "clmgs_sh_bwd": lambda n, V, I, P, T: 420 * V,
    "clmgs_isect_count": lambda n, V, I, P, T: 28 * n,
    "clmgs_isect_emit_sort": lambda n, V, I, P, T: (12 + 8 + 24 * 6) * I,
    "clmgs_isect_offsets": lambda n, V, I, P, T: 8 * I + 4 * T,
    "clmgs_rasterize_fwd": lambda n, V, I, P, T: 40 * I + 20 * P,
    "clmgs_rasterize_bwd": lambda n, V, I, P, T: 76 * I + 24 * P,
    "clmgs_ssim_fwd": lambda n, V, I, P, T: (12 + 12 + 36) * P,
    "clmgs_ssim_bwd": lambda n, V, I, P, T: (36 + 24 + 12) * P,
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# VALU issue rate of the chip (profiles/ilp_probe.hip -> r02_ilp_probe.jsonl: one-wave workgroups of v_fma_f32
# chains): 1 140 G wave64 instructions/s with 8 waves/SIMD resident, but only 870-914 with the 5 waves/SIMD the
# backward tile kernel's 96 VGPRs allow (825 at 4, 675 at 2): occupancy, not instruction-level parallelism,
# sets the issue rate.  profiles/valu_calib.hip gives the RELATIVE costs (exp / rcp / permlane swap 2.9 slots,
# DPP add 1.5).  The compute-side roofline of the alpha-blend kernels.
VALU_PEAK_G = 1140.0
VALU_CEILING_5_WAVES_G = 900.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="rubble28m", choices=sorted(CONFIGS))
    ap.add_argument("--strategy", default="clm_offload", choices=["clm_offload", "no_offload", "naive_offload"])
    ap.add_argument("--residency", default="hbm", choices=["hbm", "host"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline budget")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--prime-seconds", type=float, default=15.0,
                    help="untimed evaluation renders (model untouched) before the warm-up steps: the first minute "
                         "of a fresh box runs the same build 3-5 %% slower (143-151 vs 149-158 img/s as first / "
                         "later process; 30 s of priming: 150.9 vs 152.7)")
    ap.add_argument("--allocator-reservoir-gb", type=float, default=2.0,
                    help="one device block of this size is allocated and freed (left in the caching allocator) after the "
                         "warm-up steps, so that slightly larger requests inside the timed region split it instead of "
                         "calling hipMalloc")
    ap.add_argument("--camera-order", default="shuffle", choices=["shuffle", "path"],
                    help="shuffle: batches are seeded random draws from the survey, as the reference's DataLoader "
                         "(shuffle=True, train.py:156-167); path: consecutive cameras of the lawn-mower path (neighbours "
                         "overlap ~80 %%: far fewer distinct rows per batch than a shuffled loader sees)")
    ap.add_argument("--row-order", default="morton", choices=["morton", "random"],
                    help="morton: the scene's rows are stored along a Z-order curve of (x, y) (utils.morton_order, applied "
                         "once when the model is built, as a loader would); random: the generator's order")
    ap.add_argument("--no-host-leg", action="store_true",
                    help="skip the second leg (the same workload with the SH rows + Adam state in pinned host memory)")
    ap.add_argument("--host-staging", default="window", choices=["window", "batch"],
                    help="host-resident leg: per-camera staging windows (strategies/clm_offload/host_window.py, the default) "
                         "or the union of the batch's rows (the round-2..5 form)")
    ap.add_argument("--no-host-staging-pair", action="store_true",
                    help="skip the second, shorter host-resident leg with the other staging form")
    ap.add_argument("--host-budget-gb", type=float, default=0.0,
                    help="host-resident leg: sh_hbm_budget_gb of that leg (768 B per resident row); the default run adds a "
                         "third, shorter leg at --host-budget-leg-gb beside the budget-0 leg")
    ap.add_argument("--host-budget-leg-gb", type=float, default=-1.0,
                    help="budget of the extra host-resident leg (-1: half of the rows; 0: no such leg)")
    ap.add_argument("--host-steps", type=int, default=20)
    ap.add_argument("--host-warmup", type=int, default=2)
    ap.add_argument("--no-host-hint", action="store_true",
                    help="host-resident leg: do not tell the engine the next batch's cameras (no speculative prefetch)")
    ap.add_argument("--gt", default="both", choices=["both", "resident", "host"],
                    help="where the ground-truth images are when the timed region starts.  resident: in HBM (the bench "
                         "contract: inputs resident before the timed region; `value`); host: in pinned host memory, every "
                         "batch's images uploaded on a side stream one batch ahead (the reference: train.py:310-312); both "
                         "(default): `value` with resident images, then the same K steps again with streamed images as "
                         "`value_gt_streamed`")
    ap.add_argument("--gt-layout", default="chw", choices=["chw", "hwc"],
                    help="memory layout of the synthetic ground-truth images: chw = planar contiguous [3,H,W] as the "
                         "reference's Camera holds them (default); hwc = the strided view rounds 1-3 used by accident "
                         "(kept for A/B: costs a transposing copy per camera)")
    ap.add_argument("--dp-mode", default="auto", choices=["auto", "allreduce", "owner", "locality"],
                    help="camera-DP exchange for --gpus > 1 (clm_gs_amd/dp.py); auto = locality (Z-ordered rows required)")
    ap.add_argument("--scene", default="slab", choices=["slab", "heavy"],
                    help="scale distribution of the synthetic Gaussians (clm_gs_amd/synthetic.py SCENE_KINDS): slab = SURVEY "
                         "8d's generator (I/V = 3.8 at 4K); heavy = heavy-tailed scales with a measured I/V of ~10 (long "
                         "per-tile lists), same N, same image size")
    ap.add_argument("--no-trainer-leg", action="store_true",
                    help="skip the third leg: clm_gs_amd.trainer.training on the bench scene (densify + opacity reset inside the "
                         "end-to-end clock) -> trainer_img_s / trainer_peak_gpu_bytes")
    ap.add_argument("--trainer-images", type=int, default=400)
    ap.add_argument("--trainer-trace", action="store_true",
                    help="diagnosis: drain the device after every trainer iteration / model method and report where time and hipMallocs go")
    ap.add_argument("--trainer-grad-threshold", type=float, default=0.0002)
    ap.add_argument("--no-preflight", action="store_true",
                    help="--gpus > 1: skip the camera-DP pre-flight (clm_gs_amd/dp_preflight.py: the exchange's collectives, the "
                         "exchange itself on a seeded table and two tiny locality batches, run by one child process per rank on "
                         "the real backend; a failure or a timeout makes the run fall back to --dp-mode allreduce)")
    ap.add_argument("--preflight-timeout", type=float, default=float(os.environ.get("CLMGS_PREFLIGHT_TIMEOUT", "200")))
    ap.add_argument("--no-allreduce-leg", action="store_true",
                    help="--gpus > 1 with the locality / owner exchange: skip the second, short leg that runs the plain "
                         "all-reduce camera-DP (north_star's design) on a fresh model -> dp.allreduce_leg")
    ap.add_argument("--allreduce-steps", type=int, default=6)
    ap.add_argument("--allreduce-timeout", type=float, default=float(os.environ.get("CLMGS_ALLREDUCE_LEG_TIMEOUT", "180")),
                    help="watchdog of the all-reduce leg: after this many seconds rank 0 prints the line without the leg")
    ap.add_argument("--no-heavy-leg", action="store_true",
                    help="skip the extra short single-GPU leg on the heavy-tailed scene (--scene heavy, I/V ~ 11) -> value_heavy")
    ap.add_argument("--heavy-steps", type=int, default=8)
    ap.add_argument("--bsz", type=int, default=0, choices=[0, 4, 8, 16, 32, 64],
                    help="cameras per batch and GPU instead of the configuration's (the reference's release scripts run "
                         "BigCity at bsz 64, release_scripts/bigcity.sh:73-92: `--config bigcity102m --bsz 64` is the "
                         "single-GPU form of that row); the optimizer hyper-parameters scale with it (lr_scale_mode sqrt)")
    ap.add_argument("--opt", action="append", default=[], help="engine option override, key=value")
    return ap.parse_args()


def make_gt_images(cams, scene, args_ns, width, height):
    """GT = render of a perturbed copy of the scene (synthetic.perturbed_copy: jittered positions + a
    systematic opacity / size / colour error) quantised to uint8."""
    from clm_gs_amd import utils
    from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload, clm_offload_eval_one_cam
    from clm_gs_amd.synthetic import perturbed_copy

    gt_model = GaussianModelCLMOffload(3, only_for_rendering=True)
    t = perturbed_copy(scene)
    gt_model.create_from_tensors(t["xyz"], t["shs48"], t["scaling"], t["rotation"], t["opacity"])
    del t
    gt_model.active_sh_degree = 3
    for c in cams:
        img = clm_offload_eval_one_cam(c, gt_model, None, None)
        # [3,H,W] planar CONTIGUOUS like the reference's Camera holds it (scene/cameras.py:74); the renderer's
        # image is an [H,W,3] buffer viewed as [3,H,W], and until round 3 the quantised copy inherited those
        # strides: every timed camera then paid a transposing u8 copy (47.8 MB, 0.04-1.6 ms) in the loss stage
        c.original_image = (img.clamp(0, 1) * 255.0).round().to(torch.uint8)
        if getattr(args_ns, "gt_layout", "chw") == "chw":
            c.original_image = c.original_image.contiguous()
    del gt_model
    torch.cuda.empty_cache()


def cpu_baseline(gaussians, cam, width, height, budget_s):
    """C oracle (OpenMP port) on a bounded sample: one micro-batch (camera 0) of the same scene,
    cropped to a centred window sized to ~budget_s of CPU work; forward + loss + backward.  The SAME
    window then goes through the product's fused HIP path and the two results are compared
    (oracle/camera_parity.py): `parity` = image PSNR, |loss|, radii / intersection-count mismatches and the
    rel-L2 error of every gradient tensor -- at the size the bench runs, on the model the timed steps
    have just trained."""
    from clm_gs_amd import _lib
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # all host cores this process may run on: the affinity mask AND the cgroup CPU quota -- the
    # GPU boxes show 256 hardware threads but grant 16 CPUs (cpu.max); 256 threads on a 16-CPU quota ran the
    # same sample 36x slower than 16 threads
    usable = max(1, min(affinity, _lib.lib().clmgs_host_usable_cpus()))
    os.environ.setdefault("OMP_NUM_THREADS", str(usable))
    from oracle import c_oracle as C
    from oracle import camera_parity as CP
    C.set_num_threads(usable)
    if hasattr(gaussians, "flush_lazy_rows"):
        gaussians.flush_lazy_rows()  # the SH rows both sides read are current

    def oracle_only(cw, ch):
        from clm_gs_amd import utils
        from clm_gs_amd.strategies.base_engine import calculate_filters
        wcam, w, h = CP.window_camera(cam, width, height, cw, ch)
        utils.set_img_size(h, w)
        try:
            with torch.no_grad():
                filters, _, _ = calculate_filters([wcam], gaussians.get_xyz, gaussians.get_opacity,
                                                  gaussians.get_scaling, gaussians.get_rotation)
            inp = CP.oracle_inputs(gaussians, wcam, filters[0])
        finally:
            utils.set_img_size(height, width)
        return CP.oracle_camera(inp, w, h, int(gaussians.active_sh_degree))[1]

    cw, ch = min(width, 512), min(height, 384)
    t_cal = oracle_only(cw, ch)  # calibration crop
    frac = (cw * ch) / float(width * height)
    scale = max(1.0, min(1.0 / frac, budget_s / max(t_cal, 1e-3)))
    s = math.sqrt(scale)
    cw2, ch2 = min(width, int(cw * s) // 16 * 16), min(height, int(ch * s) // 16 * 16)
    if cw2 * ch2 >= 0.9 * width * height:  # the budget reaches (nearly) the whole image: take all of it
        cw2, ch2 = width, height
    rep, t = CP.camera_parity(gaussians, cam, width, height, rows="visible", cw=cw2, ch=ch2)
    frac2 = (cw2 * ch2) / float(width * height)
    parity_ok = rep["violations"] == []
    # ---- the rest of a batch on the CPU (an engine-level baseline): the visibility cull of
    # calculate_filters (one projection of ALL N rows per camera, base_engine.py:18-76) and the dense Adam over the 59
    # floats of every row (optimizer.py:91-184), each timed on a contiguous sample of rows and scaled to N
    import ctypes

    import numpy as np
    from oracle import gs_oracle as O
    N = int(gaussians._xyz.shape[0])
    bsz = int(__import__("clm_gs_amd").utils.get_args().bsz)
    n_s = min(N, 4_000_000)
    with torch.no_grad():
        m_ = gaussians._xyz.detach()[:n_s].cpu().numpy()
        q_ = torch.nn.functional.normalize(gaussians._rotation.detach()[:n_s]).cpu().numpy()
        s_ = torch.exp(gaussians._scaling.detach()[:n_s]).cpu().numpy()
    vm_ = np.ascontiguousarray(cam.world_view_transform.t().contiguous().cpu().numpy(), dtype=np.float32)
    K_ = np.ascontiguousarray(cam.K.cpu().numpy(), dtype=np.float32)
    radii = np.zeros(n_s, np.int32)
    m2, dep, con = np.zeros((n_s, 2), np.float32), np.zeros(n_s, np.float32), np.zeros((n_s, 3), np.float32)
    P_ = lambda a_: a_.ctypes.data_as(ctypes.c_void_p)
    f_ = ctypes.c_float
    tc0 = time.perf_counter()
    C.lib().orc_project(n_s, P_(m_), P_(q_), P_(s_), P_(vm_), P_(K_), width, height, f_(0.3), f_(0.01), f_(1e10), f_(0.0),
                        P_(radii), P_(m2), P_(dep), P_(con))
    t_cull = (time.perf_counter() - tc0) * N / float(n_s)
    n_a = min(N, 1_000_000)
    torch_threads_before = torch.get_num_threads()
    torch.set_num_threads(usable)
    g_ = torch.Generator().manual_seed(0)
    pa, ga = torch.randn(n_a, 59, generator=g_), torch.randn(n_a, 59, generator=g_)
    ma, va = torch.zeros(n_a, 59), torch.zeros(n_a, 59)
    O.adam_rows(pa, ga, ma, va, None, torch.full((59,), 1e-3), 0.9, 0.999, 1e-15, 1, 0.25)  # warm the allocator
    ta0 = time.perf_counter()
    O.adam_rows(pa, ga, ma, va, None, torch.full((59,), 1e-3), 0.9, 0.999, 1e-15, 2, 0.25)
    t_adam = (time.perf_counter() - ta0) * N / float(n_a)
    torch.set_num_threads(torch_threads_before)  # the legs after this one keep their own CPU budget (the host pool)
    t_cam = t / frac2
    batch_s = bsz * (t_cam + t_cull) + t_adam
    return {
        "value": bsz / batch_s, "unit": "img/s", "cores": C.num_threads(), "kind": "port",
        "sample": (f"one whole batch of {bsz} cameras on the CPU, assembled from timed samples "
                   f"({C.num_threads()} threads = the CPUs this container may use: cgroup quota / affinity, of "
                   f"{os.cpu_count()} hardware threads): per camera the visibility cull of all {N} rows "
                   f"(oracle/clmgs_oracle.c orc_project on {n_s} rows, scaled: {t_cull:.2f}s) + render forward + loss + "
                   f"backward of its {rep['rows']} visible rows (oracle/clmgs_oracle.c, centred {cw2}x{ch2} window = "
                   f"{frac2:.4f} of the {width}x{height} image, {rep['n_isects_oracle']} intersections, {t:.2f}s -> "
                   f"{t_cam:.2f}s per image); per batch the dense Adam over 59 floats x {N} rows (oracle adam_rows on "
                   f"{n_a} rows, scaled: {t_adam:.2f}s); value = {bsz} / ({bsz} x ({t_cam:.2f} + {t_cull:.2f}) + {t_adam:.2f}) s"),
        "render_only_value": frac2 / t,
        "seconds": {"camera_fwd_loss_bwd": round(t_cam, 3), "cull_per_camera": round(t_cull, 3), "adam_per_batch": round(t_adam, 3)},
        "parity": {"ok": bool(parity_ok),
                   "what": "the same window through the fused HIP path (clm_gs_amd.fused.camera_forward/backward) vs the "
                           "oracle (oracle/camera_parity.py: image >= 60 dB, |loss| <= 1e-5, radii / intersection total "
                           "bit-exact up to counted fp32 ceil() ties, raw-parameter gradients rel-L2 <= 1e-3 per tensor on "
                           "the same loss cotangent; `natural` = each side's own cotangent, bounded by the counted "
                           "sign(image-gt) ties of the L1 term)",
                   **{k: (round(v, 9) if isinstance(v, float) else v) for k, v in rep.items()}},
    }


def gt_to_pinned_host(cams):
    """Host-resident mode: the cameras' ground-truth images live in pinned host memory, as the reference's
    OffloadSceneDataset keeps them (utils/camera_utils.py:75-126), and are uploaded per batch."""
    from clm_gs_amd.host import pinned_empty
    for c in cams:
        if getattr(c, "image_host", None) is None and c.original_image is not None:
            h = pinned_empty(tuple(c.original_image.shape), dtype=torch.uint8)
            h.copy_(c.original_image)
            c.image_host, c.original_image = h, None


class GtFeeder:
    """train.py:310-312 with one batch of look-ahead: while batch b renders, batch b+1's images travel from pinned
    host memory to HBM on the side stream (SDMA, no compute unit involved) into a ring of preallocated device
    buffers (3 batches deep: no allocation inside the timed region; a slot is overwritten only after the batch
    that read it has been enqueued completely -- release() records the event the next upload waits for)."""
    DEPTH = 3

    def __init__(self, stream):
        self.stream, self.pending, self.ring, self.freed = stream, {}, {}, {}

    def start(self, key, batch):
        if key in self.pending or not batch:
            return
        slot = key % self.DEPTH
        bufs = self.ring.get(slot)
        if bufs is None or len(bufs) != len(batch) or bufs[0].shape != batch[0].image_host.shape:
            # NEW buffers: the caching allocator hands out blocks that the current (default) stream has freed but
            # whose last readers may still be QUEUED there (stream-ordered reuse is only safe on the freeing
            # stream).  The upload below runs on the feeder stream, so it must first wait for everything the default
            # stream has enqueued so far -- otherwise it overwrites e.g. the previous batch's intersection lists under
            # the kernels still reading them (round 4: GPU memory fault in the second, streamed pass of
            # `--strategy no_offload` / `overlap_cameras=false`, present since round 2; the pipelined default never hit it).
            bufs = self.ring[slot] = [torch.empty(tuple(c.image_host.shape), dtype=torch.uint8, device="cuda") for c in batch]
            self.stream.wait_stream(torch.cuda.current_stream())
        ev_free = self.freed.pop(slot, None)
        with torch.cuda.stream(self.stream):
            if ev_free is not None:
                self.stream.wait_event(ev_free)
            for c, buf in zip(batch, bufs):
                buf.copy_(c.image_host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.pending[key] = (batch, bufs, ev)

    def take(self, key, batch):
        self.start(key, batch)
        b, imgs, ev = self.pending.pop(key)
        torch.cuda.current_stream().wait_event(ev)
        for c, im in zip(b, imgs):
            c.original_image = im

    def release(self, key):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.freed[key % self.DEPTH] = ev


def upload_gt(batch, stream):
    """train.py:310-312: the batch's GT images go to the GPU (here: asynchronously on the side stream)."""
    cur = torch.cuda.current_stream()
    with torch.cuda.stream(stream):
        for c in batch:
            c.original_image = c.image_host.to("cuda", non_blocking=True)
            # allocated on the side stream, read by kernels of the current one: the allocator must not hand the block to the
            # NEXT batch's upload (side stream again) while those kernels are still queued
            c.original_image.record_stream(cur)
    cur.wait_stream(stream)


def host_resident_leg(a, N, W, H, bsz, vis_frac, cams, lr_extent, extent):
    """Second leg: the SAME workload with sh_residency="host" (SH rows, their gradients and Adam state in
    pinned host memory; deferred host row optimizer; staging rows moved with hipMemcpyAsync on the side
    stream) -> img/s and peak GPU bytes of the offloading configuration of the metric."""
    import gc

    from clm_gs_amd import _lib, utils
    from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload, clm_offload_train_one_batch
    from clm_gs_amd.strategies.clm_offload.engine import hint_next_batch
    from clm_gs_amd.synthetic import synth_gaussians
    host_steps = max(1, min(a.host_steps, len(cams) // bsz - a.host_warmup))
    n_b = a.host_warmup + host_steps
    cams = cams[:n_b * bsz]
    gt_to_pinned_host(cams)
    for c in cams:
        c.original_image = None
    gc.collect()
    torch.cuda.empty_cache()
    args = utils.default_args(bsz=bsz, sh_residency="host", host_staging=a.host_staging,
                              sh_hbm_budget_gb=float(a.host_budget_gb))
    args.clm_offload = True
    for kv in a.opt:  # (engine options of the host-resident mode given with --opt apply to this leg too)
        k, v = kv.split("=", 1)
        if (k.startswith("host_") or k == "sh_hbm_budget_gb") and hasattr(args, k):
            cur = getattr(args, k)
            setattr(args, k, v.lower() in ("1", "true", "yes") if isinstance(cur, bool) else type(cur)(v))
    utils.set_args(args)
    utils.set_img_size(H, W)
    scene = synth_gaussians(N, seed=0, device="cuda", kind=a.scene)
    if a.row_order == "morton":
        order = utils.morton_order(scene["xyz"])
        for k in ("xyz", "scaling", "rotation", "opacity", "shs48"):
            scene[k] = utils.gather_rows(scene[k], order)
        del order
    g = GaussianModelCLMOffload(3)
    g.create_from_tensors(scene["xyz"], scene["shs48"], scene["scaling"], scene["rotation"], scene["opacity"],
                          spatial_lr_scale=lr_extent)
    del scene
    g.active_sh_degree = 3
    g.training_setup(args)
    gc.collect()
    torch.cuda.empty_cache()
    comm = torch.cuda.Stream()
    gen = torch.Generator(device="cuda").manual_seed(1)
    n_threads = _lib.lib().clmgs_host_pool_start(0)

    class _Scene:
        cameras_extent = extent
    it = [1]
    losses_all = []

    def step(b):
        batch = cams[b * bsz:(b + 1) * bsz]
        upload_gt(batch, comm)
        utils.set_cur_iter(it[0])
        g.update_learning_rate(it[0])
        hint_next_batch(g, cams[(b + 1) * bsz:(b + 2) * bsz] if not a.no_host_hint else None)  # what a data loader knows
        ls, _, _ = clm_offload_train_one_batch(g, _Scene, batch, g.parameters_grad_buffer, None, None, comm, gen)
        for c in batch:
            c.original_image = None
        it[0] += bsz
        return ls
    for b in range(a.host_warmup):
        losses_all += list(step(b))
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    _lib.STATS.setdefault("touched_rows", []).clear()
    _lib.HOST_REGIONS = {}
    _lib.REGION_TRACE = {} if os.environ.get("CLMGS_REGION_TRACE") == "1" else None
    _lib.STATS["host_prepare_s"] = 0.0
    from clm_gs_amd import telemetry as _tel
    tel = _tel.Sampler()
    tel.__enter__()
    t0 = time.perf_counter()
    for b in range(a.host_warmup, n_b):
        losses_all += list(step(b))
    t_batches = time.perf_counter() - t0
    g.flush_lazy_rows()  # the deferred row steps still waiting after the last batch: inside the timed region
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tel.__exit__(None, None, None)
    regions, _lib.HOST_REGIONS = _lib.HOST_REGIONS, None
    peak = torch.cuda.max_memory_allocated()
    tr = _lib.STATS.get("touched_rows", [])
    T_all = T = sum(tr) / max(1, len(tr))
    K_res = g.hbm_prefix_rows()
    if K_res:  # sh_hbm_budget_gb: rows [0, K) never cross the link and are stepped on the GPU
        th = _lib.STATS.get("host_touched_rows", [])[-host_steps:]
        T = sum(th) / max(1, len(th))
    link_bytes = 2 * 192.0 * T + 4.0 * T + bsz * 3.0 * H * W  # rows down + gradient rows up + row list + GT images
    vals = [float(x) for x in losses_all]
    k2 = min(2 * bsz, max(bsz, len(vals) // 2))
    out = {"value": round(host_steps * bsz / dt, 3), "unit": "img/s", "ms_per_step": round(dt / host_steps * 1e3, 2),
           "steps": host_steps, "warmup": a.host_warmup, "peak_gpu_bytes": int(peak),
           "pinned_host_bytes": int(4 * g.parameters_buffer.shape[0] * 192),
           "touched_rows_per_batch": round(T_all, 1), "host_touched_rows_per_batch": round(T, 1),
           "sh_hbm_budget_gb": float(args.sh_hbm_budget_gb), "hbm_resident_rows": int(K_res),
           "hbm_resident_bytes": int(K_res) * 768, "host_threads": n_threads,
           "late_rows_per_batch": (round(sum(_lib.STATS.get("host_late_rows", [])[-host_steps:]) / host_steps, 1)
                                   if _lib.STATS.get("host_late_rows") else None),
           "speculative_prefetch": not a.no_host_hint, "staging": a.host_staging,
           "final_flush_ms": round((dt - t_batches) * 1e3, 1),
           "value_steady": round(host_steps * bsz / t_batches, 3),
           "value_note": "value = timed batches + the flush of the deferred host row steps still waiting after the last batch "
                         "(one-off, final_flush_ms); value_steady = the batches alone",
           "host_rows_per_s_per_thread": round(T * host_steps / t_batches / max(1, n_threads), 1),
           "host_pool_busy_fraction": round(_lib.STATS.get("host_prepare_s", 0.0) / t_batches, 3),
           "host_pool_note": "fraction of the timed batches during which the host thread pool (deferred row optimizer + staging "
                             "copy, 1 536 B of host memory traffic per touched row) was working: near 1.0 the leg is bound by the "
                             "CPUs the container grants, not by the link or the GPU",
           "host_ms_per_step": {k: round(v / host_steps * 1e3, 2) for k, v in regions.items()},
           **({"host_ms_trace": _lib.REGION_TRACE} if _lib.REGION_TRACE else {}),
           "link": {"bytes_per_batch": round(link_bytes, 1), "achieved_GBps": round(link_bytes * host_steps / dt / 1e9, 2),
                    "peak_GBps": 57.0, "note": "peak = hipMemcpyAsync pinned<->HBM measured on this node type, EITHER direction or "
                    "both together (profiles/r02_probe_host_link.json); every touched SH row crosses once per direction per batch"},
           "loss_first": round(sum(vals[:k2]) / k2, 6), "loss_last": round(sum(vals[-k2:]) / k2, 6),
           "clocks": tel.summary(),
           "gt_images": "pinned host, uploaded per batch (train.py:310-312)"}
    del g
    gc.collect()
    torch.cuda.empty_cache()
    return out


def trainer_leg(a, N, W, H, bsz, cams, lr_extent, extent):
    """Row f1 at full size: the product's own training loop, `clm_gs_amd.trainer.training`, on the
    bench scene for `--trainer-images` images -- LR schedule per image, the engine call, ONE densify_and_prune (+ the
    Z-order re-sort of every table) and ONE opacity reset inside the reference's End2endTimer (utils/timer.py:87-111),
    the evaluation outside it (train.py:438-447) -- with NONE of bench.py's measurement aids (no priming renders, no
    allocator reservoir, a fresh model).  Reports the reference's own log figures: `end2end total_time ... it/s` and
    `Max Memory usage`, plus the host time per phase."""
    import gc
    import io
    import re

    from clm_gs_amd import _lib, trainer, utils
    from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload
    from clm_gs_amd.synthetic import synth_gaussians
    n_img = int(a.trainer_images)
    gc.collect()
    torch.cuda.empty_cache()
    # what the earlier legs of this process still hold (ground-truth images of the run's cameras, 4.8 GB at 100 x 4K, and
    # whatever their objects have not released): part of max_memory_allocated below, not of the training run's own need
    held_before = int(torch.cuda.memory_allocated())
    for c in cams:  # the loop takes cameras whose ground truth is on the GPU (trainer.training docstring)
        if c.original_image is None:
            c.original_image = c.image_host.to("cuda")
    # the reference's cadence (densification.py:5-20, arguments/__init__.py: densification_interval 100): a
    # densify_and_prune every 100 images -- 3 in the default 400-image leg -- and one opacity reset
    args = utils.default_args(bsz=bsz, sh_residency="hbm", iterations=n_img,
                              densify_from_iter=50, densification_interval=100,
                              densify_until_iter=n_img - 50, opacity_reset_interval=3 * (n_img // 4),
                              densify_grad_threshold=float(a.trainer_grad_threshold),
                              # row tables with 5 % of head room, as a training run sizes them (the reference's
                              # --prealloc_capacity, train.py:107-115): the densification appends in place instead of
                              # re-allocating all four tables at 1.5x
                              prealloc_capacity=int(N * 1.05) // 16 * 16)
    args.clm_offload = True
    for kv in a.opt:  # the engine options of the command line apply to this leg too (A/B runs)
        k, v = kv.split("=", 1)
        cur = getattr(args, k)
        if isinstance(cur, bool):
            setattr(args, k, v.lower() in ("1", "true", "yes"))
        else:
            setattr(args, k, type(cur)(v))
    utils.set_args(args)
    utils.set_img_size(H, W)
    scene = synth_gaussians(N, seed=0, device="cuda", kind=a.scene)
    g = GaussianModelCLMOffload(3)
    g.create_from_tensors(scene["xyz"], scene["shs48"], scene["scaling"], scene["rotation"], scene["opacity"],
                          spatial_lr_scale=lr_extent)
    del scene
    g.active_sh_degree = 3
    g.training_setup(args)  # (training() puts the rows in Z-order itself, as after loading a point cloud)
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()

    class _Scene:
        cameras_extent = extent
    log, phases = io.StringIO(), {}
    trace = None
    if a.trainer_trace:
        # diagnosis (NOT the reported figure: it drains the device after every iteration and around every model method):
        # true time per iteration, device mallocs and reserved bytes after it, and the structural methods one by one
        trace = {"iter": [], "methods": {}}
        t_last = [time.perf_counter()]
        seen = {(sg["address"], sg["total_size"]) for sg in torch.cuda.memory_snapshot()}
        trace["new_segments"] = []
        n_alloc = [torch.cuda.memory_stats()["num_device_alloc"]]

        def hook(it):
            torch.cuda.synchronize()
            now = time.perf_counter()
            st = torch.cuda.memory_stats()
            if st["num_device_alloc"] != n_alloc[0]:  # which pool / stream asked the device for memory, and how much
                n_alloc[0] = st["num_device_alloc"]
                for sg in torch.cuda.memory_snapshot():
                    key = (sg["address"], sg["total_size"])
                    if key not in seen:
                        seen.add(key)
                        trace["new_segments"].append([it, sg["total_size"], sg["segment_type"], sg["stream"],
                                                      max((b["size"] for b in sg["blocks"]), default=0)])
            trace["iter"].append([it, round(1e3 * (now - t_last[0]), 2), int(st["num_device_alloc"]),
                                  round(st["reserved_bytes.all.current"] / 1e9, 2), round(st["allocated_bytes.all.peak"] / 1e9, 2)])
            t_last[0] = time.perf_counter()
        phases["iter_hook"] = hook

        def timed(name, fn):
            def wrap(*aa, **kk):
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                r = fn(*aa, **kk)
                torch.cuda.synchronize()
                st = torch.cuda.memory_stats()
                trace["methods"].setdefault(name, []).append([round(1e3 * (time.perf_counter() - t_), 2), int(st["num_device_alloc"])])
                return r
            return wrap
        for name in ("flush_lazy_rows", "flush_small", "densify_and_clone", "densify_and_split", "prune_points", "permute_rows",
                     "reset_opacity", "densify_and_prune", "spatial_sort", "_append_rows", "_regather_row_tables"):
            if hasattr(g, name):
                setattr(g, name, timed(name, getattr(g, name)))
    ms0 = torch.cuda.memory_stats()
    from clm_gs_amd import telemetry as _tel
    tel = _tel.Sampler()
    tel.__enter__()
    t0 = time.perf_counter()
    timer = trainer.training(g, _Scene, cams, [], log, iterations=n_img, test_iterations=(n_img,), phase_times=phases)
    wall = time.perf_counter() - t0
    tel.__exit__(None, None, None)
    ms1 = torch.cuda.memory_stats()
    text = log.getvalue()
    m = re.search(r"end2end total_time: ([0-9.]+) s, iterations: (\d+), throughput ([0-9.]+) it/s", text)
    mem = re.findall(r"Max Memory usage: ([0-9.]+) GB", text)
    dens = re.findall(r"iteration\[(\d+),(\d+)\) densify_and_prune\. Now num of 3dgs: (\d+)", text)
    psnr = re.findall(r"Evaluating train: L1 ([0-9.eE+-]+) PSNR ([0-9.eE+-]+)", text)
    first = re.search(r"iteration\[1,\d+\) loss: ([0-9. ]+) image", text)
    last = re.findall(r"iteration\[\d+,\d+\) loss: ([0-9. ]+) image", text)
    n_after = int(g.get_xyz.shape[0])
    out = {"images": n_img, "end2end_total_time_s": float(m.group(1)) if m else None,
           "iterations_logged": int(m.group(2)) if m else None,
           "trainer_img_s": float(m.group(3)) if m else None,
           "trainer_peak_gpu_bytes": int(torch.cuda.max_memory_allocated()),
           "allocated_before_leg_bytes": held_before,
           "max_memory_usage_GB_logged": float(mem[-1]) if mem else None,
           "wall_s_incl_eval": round(wall, 3),
           "n_gaussians_before_after": [N, n_after],
           "densify_lines": [[int(x) for x in d] for d in dens],
           "eval_train_l1_psnr": [float(x) for x in psnr[-1]] if psnr else None,
           "loss_first_last_batch": [[float(x) for x in first.group(1).split()] if first else None,
                                     [float(x) for x in last[-1].split()] if last else None],
           "host_seconds_by_phase": {k: round(v, 4) for k, v in phases.items()
                                     if k not in ("reserved_bytes", "device_mallocs_before_clock")},
           "allocator_reserved_bytes": int(phases.get("reserved_bytes", 0)),
           "device_mallocs": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
           "device_mallocs_inside_the_clock": (int(ms1.get("num_device_alloc", 0) - phases["device_mallocs_before_clock"])
                                               if "device_mallocs_before_clock" in phases else None),
           "clocks": tel.summary(),
           "schedule": {"densify_at_images": list(range(100, n_img - 49, 100)), "opacity_reset_at_image": 3 * (n_img // 4),
                        "densify_grad_threshold": float(a.trainer_grad_threshold)},
           "what": "clm_gs_amd.trainer.training on a fresh model of the bench scene: shuffled epochs over the run's cameras, "
                   "LR schedule, engine call per batch, a densify_and_prune (+ Z-order re-sort in the same compaction) every 100 "
                   "images as the reference schedules it and one opacity reset INSIDE the End2endTimer, evaluation outside it; "
                   "none of bench.py's aids (no priming renders; the allocator reservation is the trainer's own SETUP step, done before "
                   "the end-to-end clock starts like the reference's --prealloc_capacity buffers, its wall time reported as "
                   "host_seconds_by_phase.reserve); loss lines written one batch late (defer_loss_log) so the host never "
                   "drains the device between batches"}
    if trace is not None:
        steady = sorted(x[1] for x in trace["iter"])
        out["trace"] = {"median_iteration_ms": steady[len(steady) // 2],
                        "slow_iterations": [x for x in trace["iter"] if x[1] > 1.5 * steady[len(steady) // 2]],
                        "first_iterations": trace["iter"][:4], "last_iteration": trace["iter"][-1],
                        "methods_ms_mallocs": trace["methods"], "new_segments": trace["new_segments"], "mallocs_at_start": int(ms0.get("num_device_alloc", 0))}
    del g
    gc.collect()
    torch.cuda.empty_cache()
    return out


def run_preflight(world, rank, device, backend, timeout_s):
    """Camera-DP pre-flight on the real backend (clm_gs_amd/dp_preflight.py): every rank runs ONE child process; the
    children form their own group and exercise the exchange.  -> (summary dict, ok on ALL ranks).  The verdict is agreed
    through the run's own group with object collectives (all_gather_object), the most basic thing it must be able to do."""
    import shutil
    import tempfile

    import torch.distributed as dist
    from clm_gs_amd import dp_preflight
    box = [tempfile.mkdtemp(prefix="clmgs_preflight_") if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    rep_ = dp_preflight.run(rank, world, backend, device, box[0], timeout_s)
    reps = [None] * world
    dist.all_gather_object(reps, rep_)
    if rank == 0:
        shutil.rmtree(box[0], ignore_errors=True)
    ok = all(bool(r and r.get("ok")) for r in reps)
    errors = {str(r["rank"]): r.get("error") for r in reps if r and not r.get("ok")}
    return {"ok": ok, "wall_s_max": max(r.get("wall_s", 0.0) for r in reps),
            "stages_s_rank0": reps[0].get("stages_s"), "errors": errors or None,
            "what": "one child process per rank, own process group on the run's backend: raw collectives (uneven / empty "
                    "all_to_all splits, reduce_scatter, all_gather), dp.border_plan + B + D in parts on a seeded table, two "
                    "tiny locality batches + flush with the replicas' checksums compared"}, ok


def heavy_leg(a):
    """`value_heavy`: the same workload on the heavy-tailed scene (--scene heavy: I/V ~ 11, long per-tile lists -- what
    opaque trained scenes add), as a short run of this script in a child process (fresh allocator, nothing resident
    from the other legs).  The headline `value` / `config.scene` stay the slab scene of SURVEY 8d."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--scene", "heavy", "--config", a.config, "--steps", str(a.heavy_steps),
           "--warmup", "2", "--no-cpu-baseline", "--no-host-leg", "--no-trainer-leg", "--no-heavy-leg", "--gt", "resident",
           "--prime-seconds", "3", "--camera-order", a.camera_order, "--row-order", a.row_order]
    for kv in a.opt:
        cmd += ["--opt", kv]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.perf_counter()
    pr = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    if pr.returncode != 0 or len(lines) != 1:
        return {"value": None, "error": f"rc {pr.returncode}: {pr.stderr[-300:]}"}
    j = json.loads(lines[0])
    solo = j.get("kernels_solo_ms") or {}
    return {"value": j["value"], "unit": "img/s", "ms_per_step": j["ms_per_step"], "steps": j["steps"], "warmup": j["warmup"],
            "scene": "heavy", "I_over_V": j["measured"]["I_over_V"], "I_avg": j["measured"]["I_avg"],
            "I_emitted_avg": j["measured"]["I_emitted_avg"], "peak_gpu_bytes": j["peak_gpu_bytes"],
            "loss_first_last": [j["measured"]["loss_first"], j["measured"]["loss_last"]],
            "roofline_frac": (j.get("roofline") or {}).get("frac"), "roofline_frac_solo": (j.get("roofline") or {}).get("frac_solo"),
            "tile_kernels_solo_ms": {k: solo.get(k) for k in ("clmgs_rasterize_fwd", "clmgs_rasterize_bwd")},
            "binning_solo_ms": {k: v for k, v in solo.items() if "isect" in k},
            "clocks": j["measured"].get("clocks"),
            "wall_s": round(time.perf_counter() - t0, 1)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if os.environ.get("CLMGS_SHARE_GPU") == "1":  # test hook: several ranks on one device
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    under_torchrun = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if world > 1 or under_torchrun:  # a 1-rank launch through torch.distributed.run still builds the RCCL group
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CLMGS_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if world != a.gpus:
        sys.stderr.write(f"bench: --gpus {a.gpus} but WORLD_SIZE is {world}: launch with --nproc-per-node {a.gpus} "
                         f"(or plain `python bench.py --gpus {a.gpus}`, which starts its own ranks)\n")
        sys.exit(2)
    if world > 1 and os.environ.get("CLMGS_SHARE_GPU") != "1" and torch.cuda.device_count() < world:
        sys.stderr.write(f"bench: {world} ranks but {torch.cuda.device_count()} visible GPU(s)\n")
        sys.exit(2)

    from clm_gs_amd import _lib, utils
    from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians

    N, W, H, bsz, vis_frac, desc = CONFIGS[a.config]
    if a.bsz:
        bsz = a.bsz
    args = utils.default_args(bsz=bsz, sh_residency=a.residency)
    if a.config == "bigcity102m":  # as scripted by the reference: sparse Adam, densification off
        args.sparse_adam = True
        args.disable_auto_densification = True
    setattr(args, a.strategy, True)
    for kv in a.opt:
        k, v = kv.split("=", 1)
        cur = getattr(args, k)
        if isinstance(cur, bool) and v.lower() in ("1", "true", "yes", "0", "false", "no"):
            setattr(args, k, v.lower() in ("1", "true", "yes"))
        else:
            setattr(args, k, v if isinstance(cur, bool) else type(cur)(v))
    dp_mode = a.dp_mode
    if dp_mode == "auto":
        dp_mode = "locality" if (world > 1 and a.strategy == "clm_offload" and a.residency == "hbm"
                                 and a.row_order == "morton") else "allreduce"
    preflight = dp_fallback = None
    if world > 1 and dp_mode in ("locality", "owner") and not a.no_preflight:
        # the first contact with the real backend must not be able to lose the run: see clm_gs_amd/dp_preflight.py
        preflight, pf_ok = run_preflight(world, rank, local_rank, os.environ.get("CLMGS_DIST_BACKEND", "nccl"),
                                         a.preflight_timeout)
        if not pf_ok:
            dp_fallback = {"from": dp_mode, "to": "allreduce", "reason": preflight["errors"]}
            if rank == 0:
                sys.stderr.write(f"bench: camera-DP pre-flight FAILED ({preflight['errors']}): falling back from "
                                 f"--dp-mode {dp_mode} to the plain all-reduce exchange\n")
            dp_mode = "allreduce"
    if world > 1 or under_torchrun:
        args.dp_locality = dp_mode == "locality"
        args.dp_owner_computes = dp_mode == "owner"
    utils.set_args(args)
    utils.set_img_size(H, W)
    torch.manual_seed(0)

    scene = synth_gaussians(N, seed=0, device="cuda", kind=a.scene)
    if a.row_order == "morton":
        order = utils.morton_order(scene["xyz"])
        for k in ("xyz", "scaling", "rotation", "opacity", "shs48"):
            scene[k] = utils.gather_rows(scene[k], order)
        del order
    n_batches = a.warmup + a.steps
    # weak scaling: every rank owns its own cameras (seeded by rank)
    all_cams = nadir_cameras(n_batches * bsz * world, N, W, H, vis_frac, seed=0, device="cuda")
    if a.camera_order == "shuffle":
        perm = torch.randperm(len(all_cams), generator=torch.Generator().manual_seed(7)).tolist()
        all_cams = [all_cams[i] for i in perm]
    deal_info = None
    if world > 1 and dp_mode == "locality":
        # locality deal (dp.deal_cameras): every camera goes to the rank owning most of its rows (index ranges of
        # the Z-ordered tables = spatial regions); every rank computes the same deal and renders its own pool
        from types import SimpleNamespace
        from clm_gs_amd import dp as _dp
        ranks_of, shares = _dp.deal_cameras(all_cams, SimpleNamespace(_xyz=scene["xyz"], _scaling=scene["scaling"],
                                                                      _rotation=scene["rotation"]), world)
        cams = [c for c, q in zip(all_cams, ranks_of) if q == rank]
        deal_info = {"local_share": round(sum(int(shares[c, q]) for c, q in enumerate(ranks_of)) / float(shares.sum()), 4),
                     "cameras_per_rank": len(cams)}
    else:
        cams = all_cams[rank::world]
    args.gt_layout = a.gt_layout
    make_gt_images(cams, scene, args, W, H)
    gt_mode = {"v": "host" if (a.gt == "host" or (a.strategy == "clm_offload" and a.residency == "host")) else "resident"}
    if gt_mode["v"] == "host":
        gt_to_pinned_host(cams)
        torch.cuda.empty_cache()

    if a.strategy == "clm_offload":
        from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload, clm_offload_train_one_batch
        gaussians = GaussianModelCLMOffload(3)
    elif a.strategy == "naive_offload":
        from clm_gs_amd.strategies.naive_offload import GaussianModelNaiveOffload, naive_offload_train_one_batch
        gaussians = GaussianModelNaiveOffload(3)
    else:
        from clm_gs_amd.strategies.no_offload import GaussianModelNoOffload, baseline_accumGrads_impl
        gaussians = GaussianModelNoOffload(3)
    gaussians.create_from_tensors(scene["xyz"], scene["shs48"], scene["scaling"], scene["rotation"],
                                  scene["opacity"], spatial_lr_scale=scene["lr_extent"])
    extent, lr_extent = scene["extent"], scene["lr_extent"]
    del scene
    gaussians.active_sh_degree = 3
    gaussians.training_setup(args)
    torch.cuda.empty_cache()
    comm_stream = torch.cuda.Stream()
    perm_generator = torch.Generator(device="cuda").manual_seed(1)

    from clm_gs_amd import telemetry as _tel
    # (before the warm-up steps, so that a kernel trace of the run ends with the timed region)
    # how the caching allocator placed the row tables the gather kernels walk (one hipMalloc each, or blocks inside
    # larger segments): clm_gs_amd/telemetry.py tensor_alloc_info
    alloc_info = None
    try:
        _tabs = {}
        if a.strategy == "clm_offload" and getattr(gaussians, "_parameters", None) is not None and gaussians._parameters.is_cuda:
            _st = gaussians.optimizer.cpu_adam.state[gaussians._parameters]
            _tabs = {"sh_rows": gaussians._parameters.data, "sh_exp_avg": _st["exp_avg"], "sh_exp_avg_sq": _st["exp_avg_sq"],
                     "sh_grad_rows": gaussians.parameters_grad_buffer, "xyz": gaussians._xyz.data}
        alloc_info = _tel.tensor_alloc_info(_tabs) if _tabs else None
        # ... and how fast THIS process streams each of them (a plain read of the whole table, best of 3, event-timed): the
        # same table has read at different rates in different processes on one box (physical placement is the driver's), and
        # the gather kernels' durations move with it
        for _nm, _t in _tabs.items():
            _flat, _best = _t.reshape(-1), None
            for _ in range(3):
                _e0, _e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                _e0.record()
                _flat.sum()
                _e1.record()
                _e1.synchronize()
                _ms = _e0.elapsed_time(_e1)
                _best = _ms if _best is None else min(_best, _ms)
            alloc_info[_nm]["read_GBps"] = round(_flat.numel() * 4 / (_best * 1e-3) / 1e9, 1)
        # ... and the rate of a row GATHER from each row table (what the deferred row pass and the projection kernels do to
        # them; clm_gs_amd/placement.py): this is the figure that moves with the physical placement
        from clm_gs_amd import placement as _pl
        _rows48 = {k: v for k, v in _tabs.items() if v.dim() == 2 and v.shape[1] == 48}
        if _rows48:
            _cap = min(int(v.shape[0]) for v in _rows48.values())
            _idx = _pl._probe_index(_cap, torch.device("cuda"))
            _scr = torch.empty((_idx.numel(), 48), device="cuda")
            for _nm, _t in _rows48.items():
                alloc_info[_nm]["gather_GBps"] = round(_pl.gather_rate(_t, _idx, _scr), 1)
                alloc_info[_nm]["gather_in_place_GBps"] = round(_pl.gather_rate(_t, _idx, _scr, in_place=True), 1)
            _idx = _scr = None
        if _pl.probe_log():
            alloc_info["placement_probe"] = _pl.probe_log()
        _rows48 = None
        _tabs = _flat = _t = _st = None  # (no reference to the tables may outlive this block: the later legs free the model)
    except Exception as e:  # reporting only
        alloc_info = {"error": f"{type(e).__name__}: {e}"}
        _tabs = _flat = _t = _st = _rows48 = _idx = _scr = None

    class _Scene:
        cameras_extent = extent

    state = {"iteration": 1}

    feeder = GtFeeder(torch.cuda.Stream())

    def step(batch_idx):
        batch = cams[batch_idx * bsz:(batch_idx + 1) * bsz]
        if gt_mode["v"] == "host":
            feeder.take(batch_idx, batch)
            feeder.start(batch_idx + 1, cams[(batch_idx + 1) * bsz:(batch_idx + 2) * bsz])
        utils.set_cur_iter(state["iteration"])
        gaussians.update_learning_rate(state["iteration"])
        if a.strategy == "clm_offload":
            if a.residency == "host" and not a.no_host_hint:
                from clm_gs_amd.strategies.clm_offload.engine import hint_next_batch
                hint_next_batch(gaussians, cams[(batch_idx + 1) * bsz:(batch_idx + 2) * bsz])
            losses, _, sparsity = clm_offload_train_one_batch(
                gaussians, _Scene, batch, gaussians.parameters_grad_buffer, None, None, comm_stream,
                perm_generator)
        elif a.strategy == "naive_offload":
            losses, _ = naive_offload_train_one_batch(gaussians, _Scene, batch, None)
            sparsity = None
        else:
            losses, _ = baseline_accumGrads_impl(gaussians, _Scene, batch, None)
            for p in gaussians.all_parameters():
                if p.grad is not None:
                    p.grad /= bsz
            gaussians.optimizer.step()
            gaussians.optimizer.zero_grad(set_to_none=True)
            sparsity = None
        state["iteration"] += bsz * world
        if gt_mode["v"] == "host":
            for c in batch:
                c.original_image = None
            feeder.release(batch_idx)
        return losses, sparsity

    grouped = world > 1 or under_torchrun

    def fence():
        torch.cuda.synchronize()
        if grouped:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    # torch leaves ~10^6 long-lived Python objects behind; a full (generation-2) cyclic collection over
    # them stalls the enqueueing thread for ~100 ms (seen once in a 20-step run).  Freeze what exists
    # now: later collections only look at what the steps allocate.
    import gc
    gc.collect()
    gc.freeze()
    if a.prime_seconds > 0 and a.strategy == "clm_offload" and a.residency == "hbm":
        # A fresh box's first GPU process otherwise measures the clock / power ramp: the same build
        # gave 128-135 img/s as the first process of a box and 140-145 as the second.  Untimed
        # evaluation renders of the bench's own model and cameras; the model is not modified.
        from clm_gs_amd.strategies.clm_offload import clm_offload_eval_one_cam
        tp, i = time.perf_counter(), 0
        while time.perf_counter() - tp < a.prime_seconds:
            clm_offload_eval_one_cam(cams[i % len(cams)], gaussians, None, None)
            i += 1
        torch.cuda.synchronize()
    all_losses = []  # 0-dim device tensors, read after the timed region
    for b in range(a.warmup):
        all_losses += list(step(b)[0])
    fence()
    # The caching allocator of a process that has trained for minutes holds free blocks of every size it will need;
    # after a handful of warm-up steps it does not, and a request a few percent larger than anything seen so far
    # (the scene's intersection count grows as the optimisation converges) goes to hipMalloc -- 15-20 ms each in the
    # first process of a fresh box (six of them were a 90 ms hiccup inside 20 timed steps).  One cached 2 GB block
    # to split from stands in for that history; it is freed (cached), not held: peak_gpu_bytes counts allocated bytes.
    if a.allocator_reservoir_gb > 0:
        _r = torch.empty((int(a.allocator_reservoir_gb * (1 << 30)),), dtype=torch.uint8, device="cuda")
        del _r
    torch.cuda.reset_peak_memory_stats()
    _lib.STATS["n_isects"].clear()
    _lib.STATS["n_emitted"].clear()
    _lib.STATS.setdefault("touched_rows", []).clear()
    sparsities = []
    _lib.STATS["host_wait_s"] = 0.0
    _lib.STATS["isect_capacity_redo"] = 0
    ms0 = torch.cuda.memory_stats()
    if os.environ.get("CLMGS_HOST_REGIONS") == "1":
        _lib.HOST_REGIONS = {}
    from clm_gs_amd import dp as _dpm
    _dpm.reset_wire()
    # device telemetry of the timed region (clm_gs_amd/telemetry.py): full snapshots before / after, sclk + socket power
    # sampled at 10 Hz by a host thread that only reads sysfs files -- nothing is enqueued on the device for it
    from clm_gs_amd import telemetry as _tel
    _tel_main = _tel.Sampler()
    _tel_main.__enter__()
    # ---- the timed region: exactly K steps, NO per-kernel event instrumentation inside it
    t0 = time.perf_counter()
    step_marks = []
    for b in range(a.warmup, a.warmup + a.steps):
        losses, sp = step(b)
        all_losses += list(losses)
        step_marks.append(time.perf_counter())
        if sp:
            sparsities += sp
    # deferred row optimizers (clm_offload): the SH-row Adam step of a batch is applied at the rows' next
    # touch; whatever is still waiting after the K-th batch is applied HERE, inside the timed region
    # (camera-DP owner modes: the optimizer work only -- every rank catches up the rows it owns; the all-gather that
    # completes the replicas for evaluation / saving is not training work and runs after the fence)
    owner_dp = (world > 1 or under_torchrun) and dp_mode in ("owner", "locality")
    if hasattr(gaussians, "flush_lazy_rows"):
        gaussians.flush_lazy_rows(exchange=False) if owner_dp else gaussians.flush_lazy_rows()
    t_enq = time.perf_counter() - t0  # host: enqueue work + size readbacks, before the final fence
    fence()
    dt = time.perf_counter() - t0
    _tel_main.__exit__(None, None, None)
    _lib.check_device_errors()  # a look-back scan / sort pass of the binning chain that gave up raises here
    host_wait = _lib.STATS["host_wait_s"]
    n_redo = int(_lib.STATS.get("isect_capacity_redo", 0))
    ms1 = torch.cuda.memory_stats()
    dev_allocs = int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0))
    dev_frees = int(ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0))
    host_regions, _lib.HOST_REGIONS = _lib.HOST_REGIONS, None
    if grouped:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    peak_timed = torch.cuda.max_memory_allocated()
    wire = _dpm.wire_bytes()
    completing_flush_ms = None
    if owner_dp:
        tf0 = time.perf_counter()
        gaussians.flush_lazy_rows()  # replicas complete again (collective), outside the timed region
        torch.cuda.synchronize()
        completing_flush_ms = (time.perf_counter() - tf0) * 1e3

    def replicas_equal():
        """Checksums (float64 sums of every parameter tensor) of the completed replicas, compared over the ranks."""
        if not grouped:
            return None
        with torch.no_grad():
            sums = torch.stack([t.detach().double().sum() for t in gaussians.all_parameters()])
        allsums = [torch.empty_like(sums) for _ in range(world)]
        torch.distributed.all_gather(allsums, sums)
        return bool(all(torch.equal(x, allsums[0]) for x in allsums))
    rep_equal = replicas_equal() if (world > 1 and a.strategy == "clm_offload") else None
    n_loss_timed = len(all_losses)
    # ---- the same K steps once more with the ground-truth images where the reference keeps them (pinned host
    # memory, train.py:310-312), every batch's images uploaded on a side stream one batch ahead: reported beside
    # `value` as value_gt_streamed; the passes below (instrumented, solo) keep that mode
    dt_res = peak_res = None
    if a.gt == "both" and gt_mode["v"] == "resident":
        gt_to_pinned_host(cams)  # (no empty_cache: the allocator keeps the blocks the steps have been using)
        gt_mode["v"] = "host"
        feeder.start(a.warmup, cams[a.warmup * bsz:(a.warmup + 1) * bsz])  # ring buffers + first batch: before the clock
        fence()
        torch.cuda.reset_peak_memory_stats()
        tr0 = time.perf_counter()
        for b in range(a.warmup, a.warmup + a.steps):
            all_losses += list(step(b)[0])
        if hasattr(gaussians, "flush_lazy_rows"):
            gaussians.flush_lazy_rows(exchange=False) if owner_dp else gaussians.flush_lazy_rows()
        fence()
        dt_res = time.perf_counter() - tr0
        if grouped:
            t = torch.tensor([dt_res], device="cuda", dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt_res = float(t.item())
        peak_res = torch.cuda.max_memory_allocated()
    loss_vals = [float(l) for l in all_losses]
    # ---- the same K steps again (same cameras, the model has moved on) with an event pair around every
    # C-ABI call on the stream it is launched on: the per-kernel table and the roofline entry.  Its
    # throughput is reported as value_instrumented; `value` above never carries the instrumentation.
    dt_instr = None
    dp_phase_ms = None
    if not a.no_kernel_timing:
        _lib.TIMING = {}
        if grouped:
            _dpm.PHASES = {}
        fence()
        ti = time.perf_counter()
        for b in range(a.warmup, a.warmup + a.steps):
            step(b)
        fence()
        dt_instr = time.perf_counter() - ti
        if grouped:
            dp_phase_ms, _dpm.PHASES = _dpm.phase_summary(a.steps), None
    peak = peak_timed
    def _merge_dev(tm):  # the device-count forms (clmgs_*_dev) are the same kernels as their exact forms
        out_ = {}
        for k_, (c_, ms_) in tm.items():
            b_ = k_[:-4] if k_.endswith("_dev") else k_
            c0, m0 = out_.get(b_, (0, 0.0))
            out_[b_] = (c0 + c_, m0 + ms_)
        return out_
    timing = _merge_dev(_lib.timing_summary())
    _lib.TIMING = None
    # One extra UNTIMED batch on a single stream: the same kernels without anything co-running, so
    # the roofline entry can show the solo launch duration beside the in-situ one (under the
    # kernel-type streams every launch shares the chip and its wall duration stretches).
    solo = {}
    if not a.no_kernel_timing and world == 1 and a.strategy == "clm_offload":
        keep_mode = args.overlap_cameras
        args.overlap_cameras = False
        n_stat = len(_lib.STATS["n_isects"])
        _lib.TIMING = {}
        step(a.warmup + a.steps - 1)
        torch.cuda.synchronize()
        solo = {k: ms / c for k, (c, ms) in _merge_dev(_lib.timing_summary()).items() if c}
        _lib.TIMING = None
        args.overlap_cameras = keep_mode
        del _lib.STATS["n_isects"][n_stat:]
        del _lib.STATS["n_emitted"][n_stat:]
    # ---- evidence, outside every timed region: the block-skipping visibility pass on the TRAINED state.  The engine culls a
    # batch from blocks of 256 rows that a conservative test (positions dilated by the waiting steps' drift bound) lets
    # through, after stepping only those blocks' small attributes (gaussian_model.small_catch_up); here the next batch's
    # filters from that route are compared, index for index, with the exact pass over ALL rows after every waiting step
    # has been applied (flush_small -> clmgs_visibility_select_count without block flags).
    block_skip = None
    if (a.strategy == "clm_offload" and a.residency == "hbm" and world == 1 and not grouped
            and getattr(gaussians, "small_deferred", False) and (gaussians._small_def or {}).get("hist")):
        from clm_gs_amd.strategies.base_engine import select_filters
        nxt = cams[a.warmup * bsz:(a.warmup + 1) * bsz]
        with torch.no_grad():
            waiting = len(gaussians._small_def["hist"])
            flags = gaussians.small_catch_up(nxt)
            f_skip, t_skip = select_filters(nxt, gaussians._xyz.detach(), gaussians._scaling.detach(),
                                            gaussians._rotation.detach(), block_flags=flags)
            gaussians.flush_small()
            f_all, t_all = select_filters(nxt, gaussians._xyz.detach(), gaussians._scaling.detach(),
                                          gaussians._rotation.detach())
            eq = bool(torch.equal(t_skip, t_all) and all(torch.equal(x, y) for x, y in zip(f_skip, f_all)))
            block_skip = {"filters_equal": eq, "recorded_steps_waiting": waiting,
                          "blocks_flagged_fraction": round(float(flags.float().mean()), 4) if flags is not None else None,
                          "rows_selected": [int(x.numel()) for x in f_all], "union_rows": int(t_all.numel())}
        assert eq, "block-skipping visibility pass selected different rows than the exact pass on the trained state"
    n_images = a.steps * bsz
    k2 = min(2 * bsz, max(bsz, len(loss_vals) // 2))
    loss_first = sum(loss_vals[:k2]) / k2
    loss_last = sum(loss_vals[-k2:]) / k2
    train_ok = loss_last <= 1.05 * loss_first
    tr = _lib.STATS.get("touched_rows", [])[:a.steps]
    touched_avg = sum(tr) / max(1, len(tr)) if tr else float(N)
    # deferred small-attribute Adam: the share of the blocks of 256 rows the LAST batch's candidate test flagged (stepped,
    # and read by the exact visibility pass); None in the modes that step every row every batch
    small_flagged = None
    _sd = getattr(gaussians, "_small_def", None)
    if _sd and _sd.get("blk_flag") is not None:
        small_flagged = round(float(_sd["blk_flag"].float().mean()), 4)
    isects = _lib.STATS["n_isects"][:n_images] if a.strategy == "clm_offload" else _lib.STATS["n_isects"]
    I_avg = sum(isects) / max(1, len(isects))
    emitted = _lib.STATS["n_emitted"][:n_images] if a.strategy == "clm_offload" else _lib.STATS["n_emitted"]
    I_emitted = sum(emitted) / max(1, len(emitted))
    V_avg = (sum(sparsities) / max(1, len(sparsities))) * N if sparsities else float(N)
    n_rows = V_avg if a.strategy == "clm_offload" else float(N)
    P, T = W * H, math.ceil(W / 16) * math.ceil(H / 16)

    dist_backend = torch.distributed.get_backend() if grouped else None
    def _finish(allreduce_leg, clean=True):
        """Everything after the timed work: the process group is left (clean=False: the watchdog of the all-reduce leg
        calls this from its own thread while the main thread may be stuck in a collective), rank 0 assembles and prints
        the ONE line."""
        nonlocal gaussians
        if grouped and clean:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        if rank != 0:
            return
        value = n_images * world / dt
        kernels = {}
        for name, (calls, ms) in timing.items():
            if name in ALGO_BYTES and calls:
                avg_ms = ms / calls
                b = ALGO_BYTES[name](n_rows, V_avg, I_avg, P, T)
                kernels[name] = {"calls": calls, "avg_ms": round(avg_ms, 4),
                                 "algo_GBps": round(b / (avg_ms * 1e-3) / 1e9, 1),
                                 "share_of_step": round(ms / ((dt_instr or dt) * 1e3), 4)}
            elif calls:
                kernels[name] = {"calls": calls, "avg_ms": round(ms / calls, 4),
                                 "share_of_step": round(ms / ((dt_instr or dt) * 1e3), 4)}
        roofline = None
        if kernels:
            dom = max((k for k in kernels if k in ALGO_BYTES), key=lambda k: kernels[k]["calls"] * kernels[k]["avg_ms"])
            ach = kernels[dom]["algo_GBps"]
            traffic, traffic_src, valu = None, None, None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    traffic = tj.get(a.config, {}).get(dom, {}).get("traffic")
                    valu = tj.get(a.config, {}).get(dom, {}).get("valu_insts")
                    traffic_src = ("NOT measured in this run: profiles/pmc_traffic.json (" + str(tj.get("_source", "rocprofv3 --pmc "
                                   "FETCH_SIZE / WRITE_SIZE passes, profiles/collect.sh")) + ")") if traffic is not None else None
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "traffic_source": traffic_src,
                        # the same figure on the list this build actually walks (after exact per-tile culling) --
                        # `frac` charges the reference's unculled intersection count, the work the algorithm defines
                        "algo_bytes_per_launch_processed": ALGO_BYTES[dom](n_rows, V_avg, I_emitted, P, T),
                        "frac_processed": round(ALGO_BYTES[dom](n_rows, V_avg, I_emitted, P, T)
                                                / (kernels[dom]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "algo_bytes_per_launch": ALGO_BYTES[dom](n_rows, V_avg, I_avg, P, T),
                        "avg_launch_ms": kernels[dom]["avg_ms"],
                        "avg_launch_ms_solo": round(solo[dom], 4) if dom in solo else None,
                        "frac_solo": round(ALGO_BYTES[dom](n_rows, V_avg, I_avg, P, T) / (solo[dom] * 1e-3) / 1e9
                                           / HBM_PEAK_GBS, 5) if dom in solo else None,
                        "compute": ({"unit": "G wave-instr/s (VALU)", "peak": VALU_PEAK_G,
                                     "peak_theoretical": 1229.0,  # 1024 SIMDs x 2.4 GHz / 2 cycles (MI355X_MICROARCH.md)
                                     "valu_insts_per_launch": valu,
                                     "achieved": round(valu / (kernels[dom]["avg_ms"] * 1e-3) / 1e9, 1),
                                     "frac": round(valu / (kernels[dom]["avg_ms"] * 1e-3) / 1e9 / VALU_PEAK_G, 4),
                                     "achieved_solo": round(valu / (solo[dom] * 1e-3) / 1e9, 1) if dom in solo else None,
                                     "frac_solo": round(valu / (solo[dom] * 1e-3) / 1e9 / VALU_PEAK_G, 4) if dom in solo else None,
                                     "frac_solo_of_theoretical": round(valu / (solo[dom] * 1e-3) / 1e9 / 1229.0, 4) if dom in solo else None,
                                     "ceiling_at_kernel_occupancy": VALU_CEILING_5_WAVES_G,
                                     "frac_solo_of_ceiling": round(valu / (solo[dom] * 1e-3) / 1e9 / VALU_CEILING_5_WAVES_G, 4)
                                     if dom in solo else None,
                                     "note": "instruction count from the SQ_INSTS_VALU pass in profiles/ (NOT this run), "
                                             "durations from this run; peak = plain-FMA issue rate at 8 waves/SIMD, ceiling = "
                                             "the same at the kernel's 5 waves/SIMD (profiles/r02_ilp_probe.jsonl) -- the "
                                             "kernel's compares / selects / v_min, DPP adds and exp / rcp cost 1.6-3 issue slots "
                                             "each (profiles/r06_valu_calib.jsonl); it is bound by the dependent-issue chain of "
                                             "its 5 one-wave workgroups per SIMD, not by the slot count (DESIGN.md section 3, "
                                             "round 6)"}
                                    if valu else None),
                        "pairs_per_s": round(256.0 * I_avg / (kernels[dom]["avg_ms"] * 1e-3), 1),
                        "note": "alpha-blend kernels are ALU/LDS-bound on 256*I pixel-Gaussian pairs; "
                                "HBM fraction is structurally low, pairs/s reported beside it. achieved/frac use "
                                "the launch duration inside the timed region (kernel-type streams: launches share "
                                "the chip); *_solo = the same launch with nothing co-running (extra untimed batch)"}
        # whole-image algorithmic bytes A(image) of SURVEY 8d, for the end-to-end HBM figure
        p_pass = 6
        A_img = 228 * n_rows + 636 * V_avg + (220 * V_avg if a.strategy == "clm_offload" else 0) + \
            (144 + 24 * p_pass) * I_avg + 143 * P + 4 * T
        # Adam per batch = 28 B x 59 floats per row, row-sparse over the rows the batch touches
        # (untouched rows have a zero gradient; this build defers their momentum decay, clm_offload hbm mode)
        adam_rows = touched_avg if (a.strategy == "clm_offload" and a.residency == "hbm") else float(N)
        adam_img = 1652.0 * adam_rows / bsz
        # published numbers of BASELINE.md section 1 for exactly this (config, strategy), single GPU
        published = {("rubble28m", "clm_offload"): 4.04, ("rubble28m", "naive_offload"): 2.45,
                     ("rubble10m", "clm_offload"): 8.08, ("rubble10m", "no_offload"): 8.55,
                     ("rubble10m", "naive_offload"): 4.49, ("bicycle6m", "no_offload"): 40.9,
                     ("bicycle6m", "clm_offload"): 22.3, ("bicycle6m", "naive_offload"): 12.1}
        ref_img_s = published.get((a.config, a.strategy)) if (world == 1 and a.residency == "hbm") else None
        vs_baseline = round(value / ref_img_s, 3) if ref_img_s else None
        out = {
            "metric": "training images/s (Rubble-4K 28M Gaussians clm_offload)" if a.config == "rubble28m"
            else f"training images/s ({a.config} {a.strategy})",
            "value": round(value, 4), "unit": "img/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "host_ms_per_step": {"enqueue": round((t_enq - host_wait) / a.steps * 1e3, 3),
                                 "blocked_in_size_readbacks": round(host_wait / a.steps * 1e3, 3),
                                 "step_returns_ms": [round((t - t0) * 1e3, 2) for t in step_marks],
                                 "device_mallocs_in_timed_region": dev_allocs, "device_frees_in_timed_region": dev_frees,
                                 "cameras_redone_over_capacity": n_redo,
                                 **({"regions": {k: round(v / a.steps * 1e3, 3) for k, v in host_regions.items()}}
                                    if host_regions else {})},
            "vs_baseline": vs_baseline, "dtype": "f32", "data": "synthetic", "dist_backend": dist_backend,
            "dp": ({"mode": dp_mode, "fallback": dp_fallback, "preflight": preflight, "deal": deal_info,
                    "replicas_equal": rep_equal,
                    "completing_flush_ms": round(completing_flush_ms, 2) if completing_flush_ms is not None else None,
                    "phase_ms": dp_phase_ms,
                    "phase_ms_note": "per step, from the instrumented pass (value_instrumented): device_ms = event pairs on the "
                                     "stream each phase is enqueued on (B0 / B1 / D0 on the exchange's side stream, i.e. under "
                                     "rendering), host_ms = wall time of the enqueueing block incl. its host reads; plan = "
                                     "dp.border_plan, S = small_prepare (candidates + fetch), catch_up_own = deferred row steps of "
                                     "the rows anybody renders, B = parameter rows out, D = gradient lines home, D_apply = "
                                     "owner-side accumulation, tail_exchange = everything the exchange adds after the last backward",
                    "allreduce_leg": allreduce_leg,
                    "small_attributes_at_owner": bool(getattr(gaussians, "small_owner", False)),
                    "row_moments_sharded": bool(getattr(gaussians, "moments_sharded", False)),
                    "dp_exchange_bytes_per_step": round(wire.get("total", 0) / a.steps, 1),
                    "by_collective_per_step": {k: round(v / a.steps, 1) for k, v in wire.items() if k != "total"},
                    "note": "bytes rank 0 SENDS per batch (clm_gs_amd.dp.wire_bytes: ring model for all-reduce, exact sizes for "
                            "all_to_all / all_gather); the timed region ends with every rank catching up the rows it owns, the "
                            "all-gather that completes the replicas for evaluation / saving runs after it and is not counted"}
                   if (world > 1 or under_torchrun) else None),
            "config": {"workload": desc, "name": a.config, "strategy": a.strategy, "n_gaussians": N,
                       "width": W, "height": H, "bsz_per_gpu": bsz, "global_batch": bsz * world,
                       "sh_degree": 3, "sh_residency": a.residency if a.strategy == "clm_offload" else "hbm",
                       "parallelism": f"camera-dp{world}", "visible_fraction_target": vis_frac,
                       "camera_order": a.camera_order, "row_order": a.row_order, "scene": a.scene,
                       "untimed_priming_s": a.prime_seconds, "allocator_reservoir_gb": a.allocator_reservoir_gb,
                       "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
            "value_instrumented": round(n_images * world / dt_instr, 4) if dt_instr else None,
            "gt_images": ("pinned host memory, every batch uploaded on a side stream one batch ahead (train.py:310-312)"
                          if (a.gt == "host" or a.residency == "host") else
                          "all resident in HBM before the timed region (bench contract); value_gt_streamed = the same K steps with "
                          "the images in pinned host memory, uploaded per batch as the reference does (train.py:310-312)"),
            "value_gt_streamed": round(n_images * world / dt_res, 4) if dt_res else None,
            "peak_gpu_bytes": int(peak),
            "peak_gpu_bytes_gt_streamed": int(peak_res) if peak_res else None,
            "peak_gpu_bytes_note": ("sh_residency=hbm keeps the whole model + optimizer state in HBM by design (SURVEY 7: 288 GB): "
                                    f"{944 * N / 1e9:.1f} GB of the peak are the {N} x 944 B of parameters, moments and gradient rows, "
                                    f"{60 * N / 1e9:.1f} GB the packed small-attribute mirror / gradient / stamp tables; the rest is one "
                                    "to two cameras' working set (stream-ordered frees) and the resident GT images of the run "
                                    f"({len(cams)} x {3 * H * W / 1e6:.1f} MB = {len(cams) * 3 * H * W / 1e9:.1f} GB; not in the gt_streamed figure).  The reference's 13.0 GB "
                                    "is its offloading configuration (SH rows + Adam state in host memory): compare host_resident.peak_gpu_bytes"),
            "measured": {"V_avg": round(V_avg, 1), "I_avg": round(I_avg, 1), "I_over_V": round(I_avg / max(V_avg, 1), 3),
                         "I_emitted_avg": round(I_emitted, 1), "I_emitted_first_last": [emitted[0], emitted[-1]] if emitted else None,
                         "touched_rows_per_batch": round(touched_avg, 1),
                         "small_blocks_flagged_fraction": small_flagged,
                         "clocks": _tel_main.summary(), "alloc": alloc_info,
                         "block_skip_filters_equal": block_skip["filters_equal"] if block_skip else None,
                         "block_skip": block_skip,
                         "pixels": P, "tiles": T, "loss_first": round(loss_first, 6), "loss_last": round(loss_last, 6),
                         "loss_per_batch": [round(sum(loss_vals[i:i + bsz]) / bsz, 5) for i in range(0, len(loss_vals), bsz)]},
            "training_check": {"ok": bool(train_ok), "rule": "mean loss of the last 2 batches <= 1.05 x mean loss of the first 2 "
                               "(warm-up included): the timed optimisation must not diverge"},
            "end_to_end_hbm": {"algo_bytes_per_image": round(A_img + adam_img, 1), "adam_rows_per_batch": round(adam_rows, 1),
                               "achieved_GBps": round((A_img + adam_img) * value / world / 1e9, 1),
                               "frac_of_8TBps": round((A_img + adam_img) * value / world / 8e12, 5)},
            "baseline": {"img_s": ref_img_s, "source": "BASELINE.md section 1 (reference's own testbed: 1x RTX 4090 + "
                         "16-core host; img/s derived there from published training time / iterations)"}
            if ref_img_s else None,
            "roofline": roofline, "kernels": kernels,
            # the same C-ABI calls with NOTHING co-running (one extra untimed batch on a single stream): what a call costs
            # by itself, as opposed to its stretched duration next to the other streams' kernels
            "kernels_solo_ms": {k: round(v, 4) for k, v in sorted(solo.items())} if solo else None,
        }
        if not a.no_cpu_baseline and world == 1:
            try:
                if cams[0].original_image is None:  # GT images live in pinned host memory (--gt host)
                    cams[0].original_image = cams[0].image_host.to("cuda")
                out["cpu_baseline"] = cpu_baseline(gaussians, cams[0], W, H, a.cpu_seconds)
            except Exception as e:  # the baseline is reporting, never the product path
                out["cpu_baseline"] = {"value": None, "unit": "img/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        if (not a.no_host_leg and world == 1 and not grouped and a.strategy == "clm_offload" and a.residency == "hbm"
                and a.config in ("rubble28m", "rubble10m", "small")):
            try:
                del gaussians
                out["host_resident"] = host_resident_leg(a, N, W, H, bsz, vis_frac, cams, lr_extent, extent)
            except Exception as e:  # reported, never fatal for the headline
                out["host_resident"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
            # ... and the other staging form beside it (a shorter leg): per-camera windows trade ~10 % of the rate for 5 GB
            if not a.no_host_staging_pair:
                try:
                    import copy
                    a2 = copy.copy(a)
                    a2.host_staging = "batch" if a.host_staging == "window" else "window"
                    a2.host_steps = min(a.host_steps, 10)
                    h2 = host_resident_leg(a2, N, W, H, bsz, vis_frac, cams, lr_extent, extent)
                    out["host_resident_other_staging"] = {k: h2.get(k) for k in (
                        "staging", "value", "value_steady", "ms_per_step", "steps", "peak_gpu_bytes", "late_rows_per_batch",
                        "host_pool_busy_fraction", "link")}
                except Exception as e:
                    out["host_resident_other_staging"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
            # ... and the window form with an HBM budget: half of the rows (by default) resident -- rendered from and stepped
            # in HBM -- so the link and the host pool carry the other half only
            if a.host_budget_leg_gb != 0.0 and a.host_staging == "window" and not a.host_budget_gb:
                try:
                    import copy
                    a3 = copy.copy(a)
                    a3.host_budget_gb = a.host_budget_leg_gb if a.host_budget_leg_gb > 0 else (N // 2) * 768 / 1e9
                    a3.host_steps = min(a.host_steps, 10)
                    h3 = host_resident_leg(a3, N, W, H, bsz, vis_frac, cams, lr_extent, extent)
                    out["host_resident_hbm_budget"] = {k: h3.get(k) for k in (
                        "staging", "sh_hbm_budget_gb", "hbm_resident_rows", "hbm_resident_bytes", "value", "value_steady",
                        "ms_per_step", "steps", "peak_gpu_bytes", "touched_rows_per_batch", "host_touched_rows_per_batch",
                        "late_rows_per_batch", "host_pool_busy_fraction", "final_flush_ms", "host_ms_per_step", "link",
                        "loss_first", "loss_last")}
                except Exception as e:
                    out["host_resident_hbm_budget"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        if (not a.no_trainer_leg and world == 1 and not grouped and a.strategy == "clm_offload" and a.residency == "hbm"
                and a.config in ("rubble28m", "rubble10m", "small")):
            try:
                gaussians = None
                out["trainer"] = trainer_leg(a, N, W, H, bsz, cams, lr_extent, extent)
                out["trainer_img_s"] = out["trainer"]["trainer_img_s"]
                out["trainer_peak_gpu_bytes"] = out["trainer"]["trainer_peak_gpu_bytes"]
                if out["trainer_img_s"]:
                    out["trainer_vs_value"] = round(out["trainer_img_s"] / value, 4)
            except Exception as e:  # reported, never fatal for the headline
                out["trainer"] = {"trainer_img_s": None, "error": f"{type(e).__name__}: {e}"}
        if (not a.no_heavy_leg and world == 1 and not grouped and a.strategy == "clm_offload" and a.residency == "hbm"
                and a.scene == "slab" and a.config in ("rubble28m", "rubble10m", "small")):
            try:
                gaussians = None
                gc.collect()
                torch.cuda.empty_cache()
                out["heavy"] = heavy_leg(a)
                out["value_heavy"] = out["heavy"].get("value")
            except Exception as e:  # reported, never fatal for the headline
                out["heavy"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out))
        sys.stdout.flush()
        if not train_ok:
            sys.stderr.write(f"bench: training check FAILED: loss {loss_first:.5f} -> {loss_last:.5f}\n")
            sys.exit(3)

    # ---- camera-DP: a second, short leg with the PLAIN all-reduce exchange (north_star's design: every rank steps every
    # row, one all-reduce of the touched rows' gradient lines per batch) on a fresh model, so that the scaling record
    # carries the simple design beside the locality one
    allreduce_leg = None
    if world > 1 and dp_mode != "allreduce" and a.strategy == "clm_offload" and not a.no_allreduce_leg:
        # The headline has been measured; this leg is extra.  A watchdog per rank: if the leg does not finish in
        # --allreduce-timeout seconds (a collective that never returns), rank 0 prints the line WITHOUT it and every rank
        # leaves the process -- the run cannot be lost to its appendix.
        import threading
        leg_done = threading.Event()

        def _bail():
            if leg_done.is_set():
                return
            try:
                _finish({"value": None, "error": f"timed out after {a.allreduce_timeout:.0f} s (watchdog)"}, clean=False)
                sys.stdout.flush()
            finally:
                os._exit(0)
        watchdog = threading.Timer(float(a.allreduce_timeout), _bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            gaussians = None
            gc.unfreeze()
            gc.collect()
            torch.cuda.empty_cache()
            args2 = utils.default_args(**{**vars(args), "dp_locality": False, "dp_owner_computes": False})
            utils.set_args(args2)
            sc2 = synth_gaussians(N, seed=0, device="cuda", kind=a.scene)
            if a.row_order == "morton":
                o2 = utils.morton_order(sc2["xyz"])
                for k in ("xyz", "scaling", "rotation", "opacity", "shs48"):
                    sc2[k] = utils.gather_rows(sc2[k], o2)
                del o2
            gaussians = GaussianModelCLMOffload(3)
            gaussians.create_from_tensors(sc2["xyz"], sc2["shs48"], sc2["scaling"], sc2["rotation"], sc2["opacity"],
                                          spatial_lr_scale=lr_extent)
            del sc2
            gaussians.active_sh_degree = 3
            gaussians.training_setup(args2)
            state["iteration"] = 1
            n_ar = max(1, min(a.allreduce_steps, a.steps))
            for b in range(min(2, a.warmup)):
                step(b)
            fence()
            _dpm.reset_wire()
            ta0 = time.perf_counter()
            for b in range(a.warmup, a.warmup + n_ar):
                step(b)
            gaussians.flush_lazy_rows()
            fence()
            dt_ar = time.perf_counter() - ta0
            t = torch.tensor([dt_ar], device="cuda", dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt_ar = float(t.item())
            w_ar = _dpm.wire_bytes()
            allreduce_leg = {"mode": "allreduce", "value": round(n_ar * bsz * world / dt_ar, 4), "unit": "img/s",
                             "ms_per_step": round(dt_ar / n_ar * 1e3, 3), "steps": n_ar, "warmup": min(2, a.warmup),
                             "dp_exchange_bytes_per_step": round(w_ar.get("total", 0) / n_ar, 1),
                             "replicas_equal": replicas_equal(),
                             "what": "fresh model, dp_locality / dp_owner_computes off: OR of the touched-row masks, ONE "
                                     "all-reduce of (packed small gradient | SH gradient row) of the globally touched rows per "
                                     "batch, every rank steps every row (dp.py all-reduce exchange)"}
            utils.set_args(args)
        except Exception as e:  # reported, never fatal for the headline
            allreduce_leg = {"value": None, "error": f"{type(e).__name__}: {e}"}
        leg_done.set()
        watchdog.cancel()
    _finish(allreduce_leg)


if __name__ == "__main__":
    main()
