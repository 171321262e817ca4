"""Densification schedule and per-micro-batch statistics (reference: densification.py:5-147)."""

from . import dp, utils
from .clm_kernels import densify_stats


def _in_window(args, iteration):
    return (not args.disable_auto_densification) and iteration <= args.densify_until_iter


def gsplat_densification(iteration, scene, gaussians, batched_screenspace_pkg=None):
    """Every densification_interval images inside (densify_from_iter, densify_until_iter]:
    densify_and_prune; every opacity_reset_interval images: reset_opacity (densification.py:5-56).
    Triggers use the bsz-stride test so no image index is skipped (camera-DP: the image counter
    strides by the GLOBAL batch, bsz x ranks)."""
    args = utils.get_args()
    gbsz = args.bsz * dp.world_size()
    timers = utils.get_timers()
    if not _in_window(args, iteration):
        return
    timers.start("densification")
    if iteration > args.densify_from_iter and utils.check_update_at_this_iter(
            iteration, gbsz, args.densification_interval, 0):
        assert not args.stop_update_param
        gaussians.optimizer.zero_grad(set_to_none=True)
        if getattr(gaussians, "small_owner", False):
            # camera-DP with the small attributes at their owners: scales / opacities / positions of foreign rows are
            # stale until the replicas are completed -- the clone / split / prune masks are computed from them
            gaussians.flush_lazy_rows()
        if getattr(gaussians, "small_deferred", False):
            gaussians.flush_small()  # single GPU, small attributes stepped per block: the same, for the waiting steps
        dp.allreduce_densify_stats(gaussians)  # camera-DP: sum / sum / max over ranks
        timers.start("densify_and_prune")
        size_threshold = 20 if iteration > args.opacity_reset_interval else None
        gaussians.densify_and_prune(args.densify_grad_threshold, args.min_opacity,
                                    scene.cameras_extent, size_threshold)
        timers.stop("densify_and_prune")
        utils.inc_densify_iter()
    if utils.check_update_at_this_iter(iteration, gbsz, args.opacity_reset_interval, 0):
        timers.start("reset_opacity")
        gaussians.reset_opacity()
        timers.stop("reset_opacity")
    timers.stop("densification")


def update_densification_stats_offload_accum_grads(scene, gaussians, image_height, image_width,
                                                   send2gpu_final_filter_indices, means2d_grad,
                                                   radii):
    """densification.py:59-102 -> GaussianModel.gsplat_add_densification_stats_exact_filter."""
    args = utils.get_args()
    assert radii.shape[0] == send2gpu_final_filter_indices.shape[0] == means2d_grad.shape[0]
    if _in_window(args, utils.get_cur_iter()):
        gaussians.gsplat_add_densification_stats_exact_filter(
            means2d_grad, radii, send2gpu_final_filter_indices, image_width, image_height)


def update_densification_stats_baseline_accum_grads(scene, gaussians, image_height, image_width,
                                                    means2d_grad, radii, visibility):
    """densification.py:105-147: max_radii2D, |grad| accumulation and counts over radii > 0."""
    args = utils.get_args()
    if _in_window(args, utils.get_cur_iter()):
        densify_stats(None, means2d_grad.reshape(-1, 2), radii.reshape(-1), image_width,
                      image_height, gaussians.max_radii2D, gaussians.xyz_gradient_accum,
                      gaussians.denom, only_visible=True)
