"""naive_offload engine (reference: strategies/naive_offload/engine.py:20-357).

Per batch: copy ALL parameters host -> HBM, visibility filters, every camera through the same
fused per-camera kernels as the other strategies (gradients accumulate in HBM), copy ALL
gradients HBM -> pinned host, host Adam over every row (or the visible rows when sparse_adam).
Same call signatures and return values as the reference."""
import torch

from ... import utils
from ...fused import train_one_camera
from ..base_engine import calculate_filters, pipeline_forward_one_step


class _DeviceReplica:
    """What fused.train_one_camera reads from a model: raw parameter tensors with .grad, the SH
    degree and the (device-resident) densification statistics."""

    def __init__(self, gaussians):
        small = gaussians._small.detach().to("cuda", non_blocking=True)
        self.shs = gaussians._parameters.detach().to("cuda", non_blocking=True)
        self._xyz = small[:, 0:3].contiguous()
        self._opacity = small[:, 3:4].contiguous()
        self._scaling = small[:, 4:7].contiguous()
        self._rotation = small[:, 7:11].contiguous()
        self.active_sh_degree = gaussians.active_sh_degree
        self.max_radii2D = gaussians.max_radii2D
        self.xyz_gradient_accum = gaussians.xyz_gradient_accum
        self.denom = gaussians.denom

    def zero_grads(self):
        for t in (self._xyz, self._opacity, self._scaling, self._rotation):
            t.grad = torch.zeros_like(t)
        self.shs_grad = torch.zeros_like(self.shs)


def naive_offload_train_one_batch(gaussians, scene, batched_cameras, background, sparse_adam=False):
    args = utils.get_args()
    bsz = len(batched_cameras)
    rep = _DeviceReplica(gaussians)
    with torch.no_grad():
        filters, _, _ = calculate_filters(batched_cameras, rep._xyz, None, rep._scaling, rep._rotation,
                                          raw=True)
    rep.zero_grads()
    n = rep._xyz.shape[0]
    visibility = torch.zeros((n,), dtype=torch.bool, device="cuda") if sparse_adam else None
    losses = []
    for cam, f in zip(batched_cameras, filters):
        losses.append(train_one_camera(rep, cam, f, rep.shs, 1, rep.shs_grad, background, cam.original_image))
        if sparse_adam:
            visibility[f] = True
    # all gradients back to the pinned host buffers: two contiguous blocks
    small_grad = torch.zeros((n, 12), device="cuda")
    small_grad[:, 0:3] = rep._xyz.grad
    small_grad[:, 3:4] = rep._opacity.grad
    small_grad[:, 4:7] = rep._scaling.grad
    small_grad[:, 7:11] = rep._rotation.grad
    gaussians._small.grad.copy_(small_grad, non_blocking=True)
    gaussians._parameters.grad.copy_(rep.shs_grad, non_blocking=True)
    torch.cuda.synchronize()
    if not args.stop_update_param:
        scale = 1.0 / args.bsz  # "param.grad /= args.bsz", folded into the host Adam
        if sparse_adam:
            idx = torch.nonzero(visibility).flatten().to(torch.int32).cpu()
            gaussians.optimizer.sparse_step(sparse_indices=idx, grad_scale=scale)
            # rows without gradients keep theirs at zero for the next batch
            gaussians._small.grad.zero_()
            gaussians._parameters.grad.zero_()
        else:
            gaussians.optimizer.step(grad_scale=scale)
    gaussians.optimizer.zero_grad(set_to_none=True)
    assert bsz == len(losses)
    return losses, visibility


def naive_offload_eval_one_cam(gaussians, scene, camera, background):
    """Whole model to the GPU, one render (engine.py:20-46) -> image[3,H,W]."""
    with torch.no_grad():
        rep = _DeviceReplica(gaussians)
        image, _, _ = pipeline_forward_one_step(
            gaussians.opacity_activation(rep._opacity), gaussians.scaling_activation(rep._scaling),
            gaussians.rotation_activation(rep._rotation), rep._xyz, rep.shs, camera, scene, gaussians,
            background, None, eval=True)
    return image


def render_single_image(gaussians, scene, camera, background=None):
    """Strategy-dispatching single-image render by MODEL TYPE, clamped to [0,1], -> [3,H,W] (round-1 helper, kept
    for callers that have no args object).  The reference-shaped entry -- render_bigcity_images.py:638-722's
    signature, [H,W,3] result, PNG output, camera paths -- is clm_gs_amd.render_trajectory."""
    name = type(gaussians).__name__
    if name == "GaussianModelNaiveOffload":
        img = naive_offload_eval_one_cam(gaussians, scene, camera, background)
    elif name == "GaussianModelCLMOffload":
        from ..clm_offload import clm_offload_eval_one_cam
        img = clm_offload_eval_one_cam(camera, gaussians, background, scene)
    else:
        from ..no_offload import baseline_accumGrads_micro_step
        with torch.no_grad():
            img, _, _, _ = baseline_accumGrads_micro_step(
                gaussians.get_xyz, gaussians.get_opacity, gaussians.get_scaling, gaussians.get_rotation,
                gaussians.get_features, gaussians.active_sh_degree, camera, background, mode="test")
    return torch.clamp(img, 0.0, 1.0)
