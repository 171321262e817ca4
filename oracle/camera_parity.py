"""Full-size camera parity: the product's fused HIP path against oracle/clmgs_oracle.c on the SAME rows.

TEST INFRASTRUCTURE ONLY (see the header of clmgs_oracle.c / gs_oracle.py: "parity unpinned" at the
native-kernel boundary).  Callers: tests/test_gpu_fullsize.py (the -m gpu parity tests at BASELINE.json's
full sizes) and bench.py's cpu_baseline leg (which times the oracle anyway and reports the comparison).
The product never imports this module.

One camera (reference: strategies/base_engine.py:106-207 forward, base_engine.py:79-103 loss, the backward
of both, densification.py:59-102 statistics):

  HIP     clm_gs_amd.fused.camera_forward / camera_backward over the camera's visible rows: raw parameters
          in, image + loss + gradients w.r.t. the RAW parameters (activation VJPs inside the kernels)
  oracle  render_forward / loss_and_backward on the activated rows (activations by torch on the same
          device tensors), gradients chained back through sigmoid / exp / normalise in float64

compare() returns every measured error; the callers hold the tolerances.
"""
import math
import time

import numpy as np
import torch

from . import c_oracle as C


class _Cam:
    """A camera restricted to a pixel window [x0, x0+cw) x [y0, y0+ch): same pose, principal point shifted."""

    def __init__(self, cam, x0, y0, cw, ch, gt_u8):
        self.world_view_transform = cam.world_view_transform
        self.camtoworlds = cam.camtoworlds
        self.FoVx, self.FoVy = cam.FoVx, cam.FoVy
        K = (cam.K if getattr(cam, "K", None) is not None else cam.create_k_on_gpu()).clone()
        K[0, 2] -= x0
        K[1, 2] -= y0
        self.K = K
        self.image_width, self.image_height = cw, ch
        self.original_image = gt_u8
        self.uid = getattr(cam, "uid", 0)


def window_camera(cam, width, height, cw=None, ch=None):
    """The camera itself (cw/ch None or full size) or a centred, tile-aligned cw x ch window of it."""
    if cw is None or (cw >= width and ch >= height):
        return cam, width, height
    x0, y0 = (width - cw) // 2 // 16 * 16, (height - ch) // 2 // 16 * 16
    gt = cam.original_image[:, y0:y0 + ch, x0:x0 + cw].contiguous()
    return _Cam(cam, x0, y0, cw, ch, gt), cw, ch


def _rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _sh_rows_of(gaussians, rows):
    p = getattr(gaussians, "_parameters", None)
    if p is None or p.numel() == 0:  # no_offload model: dc | rest as one [N,48] view
        p = gaussians.get_features.detach().reshape(-1, 48)
    p = p.detach()
    if rows is None:
        return p.contiguous() if p.is_cuda else p.to(gaussians._xyz.device)
    if p.is_cuda:
        from clm_gs_amd import utils
        return utils.take_rows(p, rows).contiguous()
    return p[rows.cpu()].to(gaussians._xyz.device)


def hip_camera(gaussians, cam, rows, width, height):
    """The product path for one camera over `rows` (int64 device tensor; None = all rows).
    -> dict of numpy arrays: image[H,W,3], loss, v_image[3,H,W] (the loss cotangent the backward consumed),
    radii[V], n_isects (reference's 3-sigma count),
    g_xyz / g_opacity / g_scaling / g_rotation [V,.] (RAW-parameter gradients), g_shs[V,48],
    stats max_radii2D / xyz_gradient_accum / denom [V]."""
    from clm_gs_amd import _lib, fused, utils
    args = utils.get_args()
    dev = gaussians._xyz.device
    N = gaussians._xyz.shape[0]
    keep_size = (int(utils.get_img_height()), int(utils.get_img_width()))
    keep = {k: getattr(gaussians, k, None) for k in ("max_radii2D", "xyz_gradient_accum", "denom")}
    keep_grads = [p.grad for p in (gaussians._xyz, gaussians._opacity, gaussians._scaling, gaussians._rotation)]
    keep_args = (args.disable_auto_densification, args.densify_until_iter)
    utils.set_img_size(height, width)
    try:
        args.disable_auto_densification, args.densify_until_iter = False, 1 << 60
        sh = _sh_rows_of(gaussians, rows)
        g_sh = torch.zeros_like(sh)
        for p in (gaussians._xyz, gaussians._opacity, gaussians._scaling, gaussians._rotation):
            p.grad = torch.zeros_like(p)
        gaussians.max_radii2D = torch.zeros((N,), device=dev)
        gaussians.xyz_gradient_accum = torch.zeros((N, 1), device=dev)
        gaussians.denom = torch.zeros((N, 1), device=dev)
        p = fused.camera_forward(gaussians, cam, rows, sh, 0, None, cam.original_image)
        fused.camera_verify(gaussians, p)  # device-side counts: the forward may have to be redone at exact size
        v_out = p.v_out.clone()            # d loss / d image [H,W,3] (camera_backward frees it)
        means2d = p.means2d.reshape(-1, 2).clone()
        fused.camera_backward(gaussians, p, g_sh, update_stats=True, stats_only_visible=rows is None)
        loss = fused.camera_loss(p)
        torch.cuda.synchronize()
        sel = (lambda t: t.detach()) if rows is None else (lambda t: utils.take_rows(t.detach(), rows))
        out = dict(
            image=p.out.cpu().numpy(), loss=float(loss), radii=p.radii.reshape(-1).cpu().numpy(),
            n_isects=int(_lib.STATS["n_isects"][-1]), n_emitted=int(_lib.STATS["n_emitted"][-1]),
            g_xyz=sel(gaussians._xyz.grad).cpu().numpy(), g_opacity=sel(gaussians._opacity.grad).cpu().numpy(),
            g_scaling=sel(gaussians._scaling.grad).cpu().numpy(), g_rotation=sel(gaussians._rotation.grad).cpu().numpy(),
            g_shs=g_sh.cpu().numpy(), means2d=means2d.cpu().numpy(), v_image=v_out.permute(2, 0, 1).contiguous().cpu().numpy(),
            max_radii2D=sel(gaussians.max_radii2D).cpu().numpy(),
            xyz_gradient_accum=sel(gaussians.xyz_gradient_accum).reshape(-1).cpu().numpy(),
            denom=sel(gaussians.denom).reshape(-1).cpu().numpy())
        del p, g_sh, sh
        return out
    finally:
        utils.set_img_size(*keep_size)
        args.disable_auto_densification, args.densify_until_iter = keep_args
        for k, v in keep.items():
            setattr(gaussians, k, v)
        for p, g in zip((gaussians._xyz, gaussians._opacity, gaussians._scaling, gaussians._rotation), keep_grads):
            p.grad = g


def oracle_inputs(gaussians, cam, rows):
    """Activated rows as numpy (what crosses the kernel boundary in the reference,
    strategies/no_offload/engine.py:114-119) + the raw quaternions for the chain rule."""
    from clm_gs_amd import utils
    sel = (lambda t: t.detach()) if rows is None else (lambda t: utils.take_rows(t.detach(), rows))
    with torch.no_grad():
        raw_q = sel(gaussians._rotation)
        d = dict(
            means=sel(gaussians._xyz).cpu().numpy(),
            quats=torch.nn.functional.normalize(raw_q).cpu().numpy(),
            scales=torch.exp(sel(gaussians._scaling)).cpu().numpy(),
            opac=torch.sigmoid(sel(gaussians._opacity)).cpu().numpy(),
            shs=_sh_rows_of(gaussians, rows).cpu().numpy(),
            raw_q=raw_q.cpu().numpy(),
            viewmat=cam.world_view_transform.t().contiguous().cpu().numpy(),
            K=cam.K.cpu().numpy().copy(), gt=cam.original_image.cpu().numpy())
    return d


def _to_raw(inp, fw, bw, width, height):
    """Oracle gradients (w.r.t. the activated inputs) chained to the RAW parameters in float64, plus the
    densification statistics of densification.py:59-102."""
    q = inp["raw_q"].astype(np.float64)
    nrm = np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
    qh = q / nrm
    vq = bw["v_quats"].astype(np.float64)
    g_rot = (vq - (vq * qh).sum(1, keepdims=True) * qh) / nrm           # VJP of q / |q|
    g_sca = bw["v_scales"].astype(np.float64) * inp["scales"]           # VJP of exp
    o = inp["opac"].astype(np.float64).reshape(-1, 1)
    g_opa = bw["v_opac"].astype(np.float64).reshape(-1, 1) * o * (1 - o)  # VJP of sigmoid
    vis = fw["radii"] > 0
    g2 = bw["v_means2d"].astype(np.float64) * np.array([0.5 * width, 0.5 * height])  # densification.py:84-86
    return dict(g_xyz=bw["v_means"], g_opacity=g_opa, g_scaling=g_sca, g_rotation=g_rot, g_shs=bw["v_shs48"],
                xyz_gradient_accum=np.where(vis, np.linalg.norm(g2, axis=1), 0.0))


def oracle_camera(inp, width, height, sh_degree=3, v_image_hip=None):
    """C oracle forward + loss + backward.  -> (dict like hip_camera's, seconds of the oracle's own
    forward + loss + backward).  With v_image_hip (the cotangent the HIP backward consumed) a SECOND
    oracle backward runs on that cotangent: `same_cotangent` holds its gradients -- the comparison of
    everything below the loss, free of the L1 term's sign(image - gt) ties (see compare())."""
    t0 = time.perf_counter()
    fw = C.render_forward(inp["means"], inp["quats"], inp["scales"], inp["opac"], inp["shs"], sh_degree,
                          inp["viewmat"], inp["K"], width, height)
    bw = C.loss_and_backward(fw, inp["gt"])
    dt = time.perf_counter() - t0
    vis = fw["radii"] > 0
    out = dict(image=fw["image_hwc"], loss=bw["loss"], v_image=bw["v_image"], radii=fw["radii"],
               n_isects=fw["n_isects"], means2d=fw["means2d"], tiles_per_gauss=fw["tiles_per_gauss"],
               tw=fw["tw"], th=fw["th"], conics=fw["conics"],
               max_radii2D=np.where(vis, fw["radii"], 0).astype(np.float32),
               denom=vis.astype(np.float32), n_visible=int(vis.sum()), gt=inp["gt"])
    out.update(_to_raw(inp, fw, bw, width, height))
    if v_image_hip is not None:
        bw2 = C.loss_and_backward(fw, inp["gt"], v_image=v_image_hip, loss=bw["loss"])
        out["same_cotangent"] = _to_raw(inp, fw, bw2, width, height)
    return out, dt


def _tiles(mx, my, r, tw, th):
    """Tile-box size of gsplat's isect_tiles in float32, as oracle/clmgs_oracle.c tile_box computes it."""
    f = np.float32
    tr, tx, ty = r.astype(f) / f(16), mx.astype(f) / f(16), my.astype(f) / f(16)
    x0 = np.clip(np.floor(tx - tr), 0, tw); x1 = np.clip(np.ceil(tx + tr), 0, tw)
    y0 = np.clip(np.floor(ty - tr), 0, th); y1 = np.clip(np.ceil(ty + tr), 0, th)
    return ((x1 - x0) * (y1 - y0)).astype(np.int64)


def _edge_gap(mx, my, r):
    """Distance of the nearest tile-box edge ((m -+ r) / 16, four of them) to an integer, in tiles."""
    e = np.stack([(mx - r) / 16.0, (mx + r) / 16.0, (my - r) / 16.0, (my + r) / 16.0], axis=1).astype(np.float64)
    return np.abs(e - np.round(e)).min(axis=1)


GRAD_KEYS = ("g_xyz", "g_opacity", "g_scaling", "g_rotation", "g_shs")


def compare(hip, orc):
    """Every error of one camera.

    * image PSNR, |loss|;
    * radii: count of differing rows.  r = ceil(3 sqrt(lambda_1)) is a step function of fp32 arithmetic
      done in two different operation orders, so among millions of rows a handful sit within one ulp of
      a step: `radii_off_by_one` of them differ by exactly 1, `radii_cull_ties` are culled on one side
      only (the off-screen / radius tests are ties too), `radii_unexplained` = the rest (must be 0);
    * intersection total: it must equal the sum of gsplat's tile boxes at the HIP path's own means2d / radii
      bit for bit (`n_isects_from_hip_boxes`), and the oracle's total corrected for the counted tie rows
      (`radii_mismatch` rows + `tile_box_ties` rows: same radius, a box edge (m -+ r) / 16 within 1e-3 of an
      integer, pixel centres differing in the last bits) must give the same number (`n_isects_explained`);
    * loss cotangent d loss / d image: the L1 term contributes 0.8 sign(image - gt) / numel, a step
      function again; `cotangent_sign_flips` pixels-channels lie on different sides of gt in the two
      images (|image - gt| below the fp32 difference of the two renders), every other element agrees to
      `cotangent_rel_l2_without_flips`;
    * gradients, natural: each side backpropagates its OWN cotangent -- the flipped elements inject
      2 x 0.8 / numel each, which bounds the agreement at ~2 sqrt(flips / numel) whatever the kernels do;
    * gradients, same cotangent: the oracle backward re-run on the cotangent the HIP backward consumed --
      the comparison of the alpha-blend / projection / SH backward proper."""
    mse = float(np.mean((hip["image"].astype(np.float64) - orc["image"].astype(np.float64)) ** 2))
    hr, orr = hip["radii"].astype(np.int64), orc["radii"].astype(np.int64)
    mis = np.nonzero(hr != orr)[0]
    off1 = int(np.count_nonzero((np.abs(hr[mis] - orr[mis]) == 1) & (hr[mis] > 0) & (orr[mis] > 0)))
    cull = int(np.count_nonzero((hr[mis] == 0) != (orr[mis] == 0)))
    # intersection total.  (i) the HIP total must equal the sum of gsplat's tile boxes evaluated at the HIP
    # path's OWN means2d / radii (bit for bit: the count kernel and the formula agree); (ii) rows whose box
    # differs from the oracle's although the radius is the same are ties of floor() / ceil() on pixel centres
    # that differ in the last bits: counted, and each must have a box edge within 1e-3 tile of an integer
    hv = hr > 0
    t_hip = np.zeros(hr.shape, np.int64)
    t_hip[hv] = _tiles(hip["means2d"][hv, 0], hip["means2d"][hv, 1], hr[hv], orc["tw"], orc["th"])
    t_orc = orc["tiles_per_gauss"].astype(np.int64)
    same_r = (hr == orr) & hv
    box_ties = np.nonzero(same_r & (t_hip != t_orc))[0]
    gap = _edge_gap(orc["means2d"][box_ties, 0].astype(np.float64), orc["means2d"][box_ties, 1].astype(np.float64),
                    orr[box_ties].astype(np.float64)) if box_ties.size else np.zeros(0)
    m2_diff = float(np.abs(hip["means2d"][same_r] - orc["means2d"][same_r]).max()) if same_r.any() else 0.0
    gt = orc["gt"].astype(np.float32) / 255.0
    ih, io = hip["image"].transpose(2, 0, 1), orc["image"].transpose(2, 0, 1)
    flips = np.sign(ih - gt) != np.sign(io - gt)
    vh, vo = hip["v_image"].astype(np.float64), orc["v_image"].astype(np.float64)
    keep = ~flips
    r = dict(
        psnr_db=10 * math.log10(1.0 / max(mse, 1e-30)),
        loss_abs=abs(hip["loss"] - orc["loss"]), loss=orc["loss"],
        rows=int(orr.shape[0]), n_visible=int(orc["n_visible"]),
        radii_mismatch=int(mis.size), radii_off_by_one=off1, radii_cull_ties=cull,
        radii_unexplained=int(mis.size) - off1 - cull,
        n_isects_hip=int(hip["n_isects"]), n_isects_oracle=int(orc["n_isects"]),
        n_isects_from_hip_boxes=int(t_hip.sum()), tile_box_ties=int(box_ties.size),
        tile_box_tie_max_edge_gap=float(gap.max()) if gap.size else 0.0, means2d_max_abs_diff=m2_diff,
        n_isects_explained=int(orc["n_isects"]) + int((t_hip - t_orc)[box_ties].sum()) + int((t_hip - t_orc)[mis].sum()),
        denom_mismatch=int(np.count_nonzero(hip["denom"] != orc["denom"])),
        max_radii2D_mismatch=int(np.count_nonzero(hip["max_radii2D"] != orc["max_radii2D"])),
        cotangent_sign_flips=int(flips.sum()), cotangent_numel=int(flips.size),
        # a sign flip means gt lies between the two renders, i.e. |image - gt| <= their difference: flips are ties by
        # construction and their NUMBER only says how close the model is to gt (a converged model has many); what
        # must stay rare is a pixel where the renders themselves differ visibly (an alpha >= 1/255 or T > 1e-4 tie
        # moves one Gaussian's contribution, up to ~4e-3, in one pixel)
        image_big_diff=int(np.count_nonzero(np.abs(ih - io) > 1e-4)), image_max_abs_diff=float(np.abs(ih - io).max()),
        cotangent_flip_max_gap=float(np.abs(io - gt)[flips].max()) if flips.any() else 0.0,
        cotangent_rel_l2=_rel_l2(vh, vo),
        cotangent_rel_l2_without_flips=_rel_l2(vh[keep], vo[keep]),
        natural_bound=2.0 * math.sqrt(float(flips.sum()) / flips.size),
        xyz_gradient_accum_rel_l2=_rel_l2(hip["xyz_gradient_accum"], orc["xyz_gradient_accum"]))
    for k in GRAD_KEYS:
        r[k + "_rel_l2"] = _rel_l2(hip[k], orc[k])
    sc = orc.get("same_cotangent")
    if sc is not None:
        for k in GRAD_KEYS + ("xyz_gradient_accum",):
            r["same_cotangent_" + k + "_rel_l2"] = _rel_l2(hip[k], sc[k])
    return r


# Tolerances (fp32 on both sides, different operation orders).  Step functions of fp32 values cannot be
# bit-compared across two implementations at millions of samples; every exception is counted and explained.
TOL = dict(psnr_db=60.0, loss_abs=1e-5, same_cotangent_rel_l2=1e-3, natural_rel_l2=5e-3,
           cotangent_rel_l2_without_flips=1e-3, tie_fraction=1e-5)


def within_tolerance(rep):
    """-> list of violated rules (empty = parity holds)."""
    bad = []
    if rep["psnr_db"] < TOL["psnr_db"]:
        bad.append("psnr")
    if rep["loss_abs"] > TOL["loss_abs"]:
        bad.append("loss")
    if rep["radii_unexplained"] or rep["radii_mismatch"] > max(8, TOL["tie_fraction"] * rep["rows"]):
        bad.append("radii")
    if not (rep["n_isects_hip"] == rep["n_isects_from_hip_boxes"] == rep["n_isects_explained"]):
        bad.append("n_isects")
    if rep["tile_box_ties"] > max(8, 1e-4 * rep["rows"]) or rep["tile_box_tie_max_edge_gap"] > 1e-3 \
            or rep["means2d_max_abs_diff"] > 1e-2:
        bad.append("tile_box_ties")
    if rep["denom_mismatch"] > rep["radii_cull_ties"] or rep["max_radii2D_mismatch"] > rep["radii_mismatch"]:
        bad.append("stats")
    if rep["cotangent_rel_l2_without_flips"] > TOL["cotangent_rel_l2_without_flips"]:
        bad.append("cotangent")
    # pixels the two renders disagree on by more than 1e-4 are ties of the alpha >= 1/255 test (one Gaussian's
    # contribution, <= c / 255 ~ 4e-3, enters one render and not the other; measured 2e-4 of the elements at 28 M /
    # 4K, thousands of ties among 1e9 evaluated pairs): rare, and bounded by a few such contributions
    if rep["image_big_diff"] > 1e-3 * rep["cotangent_numel"] or rep["image_max_abs_diff"] > 0.02:
        bad.append("image_ties")
    for k in GRAD_KEYS:
        if rep[k + "_rel_l2"] > max(TOL["natural_rel_l2"], 2 * rep["natural_bound"]):
            bad.append("natural:" + k)
        if rep.get("same_cotangent_" + k + "_rel_l2", 0.0) > TOL["same_cotangent_rel_l2"]:
            bad.append("same_cotangent:" + k)
    if rep.get("same_cotangent_xyz_gradient_accum_rel_l2", 0.0) > TOL["same_cotangent_rel_l2"]:
        bad.append("same_cotangent:xyz_gradient_accum")
    return bad


def camera_parity(gaussians, cam, width, height, rows="visible", cw=None, ch=None, threads=None):
    """HIP vs oracle for one camera (optionally a centred cw x ch window of it).
    rows: "visible" = the camera's filter (calculate_filters, the clm path), None = all rows (no_offload).
    -> (report dict from compare() + sizes + `violations`, seconds of the oracle's own fwd + loss + bwd)."""
    from clm_gs_amd import utils
    from clm_gs_amd.strategies.base_engine import calculate_filters
    if threads:
        C.set_num_threads(int(threads))
    wcam, w, h = window_camera(cam, width, height, cw, ch)
    keep_size = (int(utils.get_img_height()), int(utils.get_img_width()))
    utils.set_img_size(h, w)
    try:
        if isinstance(rows, str):
            with torch.no_grad():
                filters, _, _ = calculate_filters([wcam], gaussians.get_xyz, gaussians.get_opacity,
                                                  gaussians.get_scaling, gaussians.get_rotation)
            rows = filters[0]
        hip = hip_camera(gaussians, wcam, rows, w, h)
        inp = oracle_inputs(gaussians, wcam, rows)
    finally:
        utils.set_img_size(*keep_size)
    orc, dt = oracle_camera(inp, w, h, int(gaussians.active_sh_degree), v_image_hip=hip["v_image"])
    rep = compare(hip, orc)
    rep.update(width=w, height=h, oracle_seconds=round(dt, 3), oracle_threads=C.num_threads(),
               n_emitted_hip=hip["n_emitted"])
    rep["violations"] = within_tolerance(rep)
    return rep, dt
