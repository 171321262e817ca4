"""ctypes front end of oracle/libclmgs_oracle.so (the plain-C, OpenMP restatement).

TEST INFRASTRUCTURE ONLY -- see the header of clmgs_oracle.c / gs_oracle.py ("parity
unpinned").  Used for mid-size parity tests and as bench.py's cpu_baseline (kind "port").
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libclmgs_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(["make", "-C", _HERE])
        _lib = ctypes.CDLL(_PATH)
        _lib.orc_isect.restype = ctypes.c_int64
        _lib.orc_loss.restype = ctypes.c_float
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def render_forward(means, quats, scales, opac, shs48, sh_degree, viewmat, K, width, height,
                   background=None, radius_clip=0.0):
    """One camera, forward only.  All inputs numpy float32; quats/scales/opac ACTIVATED
    (normalised / exp / sigmoid), exactly what crosses the kernel boundary in the reference
    (strategies/no_offload/engine.py:114-119).  Returns a dict of every intermediate."""
    L = lib()
    means, quats, scales, opac, shs48 = map(_f32, (means, quats, scales, opac, shs48))
    viewmat, K = _f32(viewmat), _f32(K)
    n = means.shape[0]
    radii = np.zeros(n, np.int32)
    m2, depths, conics = np.zeros((n, 2), np.float32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    f = ctypes.c_float
    L.orc_project(n, _p(means), _p(quats), _p(scales), _p(viewmat), _p(K), width, height, f(0.3),
                  f(0.01), f(1e10), f(radius_clip), _p(radii), _p(m2), _p(depths), _p(conics))
    campos = np.linalg.inv(viewmat.astype(np.float64))[:3, 3].astype(np.float32)
    dirs = _f32(means - campos[None])
    masks = (radii > 0).astype(np.uint8)
    sh_col = np.zeros((n, 3), np.float32)
    L.orc_sh(n, sh_degree, _p(dirs), _p(shs48), _p(masks), _p(sh_col))
    colors = np.maximum(sh_col + 0.5, 0.0).astype(np.float32)
    tw, th = math.ceil(width / 16), math.ceil(height / 16)
    n_isects = L.orc_isect(n, _p(m2), _p(radii), _p(depths), tw, th, None, None, None, None)
    ids = np.zeros(max(n_isects, 1), np.int64)
    fids = np.zeros(max(n_isects, 1), np.int32)
    offsets = np.zeros(tw * th, np.int32)
    tpg = np.zeros(n, np.int32)
    L.orc_isect(n, _p(m2), _p(radii), _p(depths), tw, th, _p(ids), _p(fids), _p(offsets), _p(tpg))
    out = np.zeros((height, width, 3), np.float32)
    alpha = np.zeros((height, width), np.float32)
    last = np.zeros((height, width), np.int32)
    bg = _f32(background) if background is not None else None
    L.orc_rasterize(width, height, tw, th, ctypes.c_int64(n_isects), _p(m2), _p(conics), _p(colors),
                    _p(opac.reshape(-1)), _p(bg), _p(offsets), _p(fids), _p(out), _p(alpha), _p(last))
    return dict(n=n, radii=radii, means2d=m2, depths=depths, conics=conics, dirs=dirs, masks=masks,
                sh_col=sh_col, colors=colors, tw=tw, th=th, n_isects=int(n_isects),
                isect_ids=ids[:n_isects], flatten_ids=fids[:n_isects], offsets=offsets,
                tiles_per_gauss=tpg, image_hwc=out, alpha=alpha, last_ids=last, bg=bg,
                inputs=dict(means=means, quats=quats, scales=scales, opac=opac, shs48=shs48,
                            viewmat=viewmat, K=K, sh_degree=sh_degree, width=width, height=height))


def loss_and_cotangent(fw, gt_u8):
    """-> (loss, d loss / d image as [3,H,W])."""
    L = lib()
    i = fw["inputs"]
    img_chw = np.ascontiguousarray(fw["image_hwc"].transpose(2, 0, 1))
    v_img = np.zeros_like(img_chw)
    gt = np.ascontiguousarray(gt_u8, dtype=np.uint8)
    loss = L.orc_loss(i["height"], i["width"], _p(img_chw), _p(gt), _p(v_img))
    return float(loss), v_img


def loss_and_backward(fw, gt_u8, v_image=None, loss=None):
    """Loss (0.8 L1 + 0.2 (1-SSIM)) and gradients w.r.t. the ACTIVATED inputs of render_forward.
    v_image ([3,H,W]): use this loss cotangent instead of the oracle's own (the backward of everything
    below the loss on a given cotangent); `loss` then labels the result."""
    L = lib()
    i = fw["inputs"]
    n, width, height = fw["n"], i["width"], i["height"]
    if v_image is None:
        loss, v_img = loss_and_cotangent(fw, gt_u8)
    else:
        v_img = np.ascontiguousarray(v_image, dtype=np.float32)
        loss = float("nan") if loss is None else loss
    v_out = np.ascontiguousarray(v_img.transpose(1, 2, 0))
    v_m2, v_con = np.zeros((n, 2), np.float32), np.zeros((n, 3), np.float32)
    v_col, v_op = np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
    L.orc_rasterize_bwd(n, width, height, fw["tw"], fw["th"], ctypes.c_int64(fw["n_isects"]),
                        _p(fw["means2d"]), _p(fw["conics"]), _p(fw["colors"]),
                        _p(i["opac"].reshape(-1)), _p(fw["bg"]), _p(fw["offsets"]),
                        _p(fw["flatten_ids"] if fw["n_isects"] else np.zeros(1, np.int32)),
                        _p(fw["alpha"]), _p(fw["last_ids"]), _p(v_out), None, _p(v_m2), _p(v_con),
                        _p(v_col), _p(v_op))
    v_shcol = np.where(fw["sh_col"] + 0.5 > 0, v_col, 0).astype(np.float32)  # clamp_min backward
    v_shs = np.zeros((n, 48), np.float32)
    v_dirs = np.zeros((n, 3), np.float32)
    L.orc_sh_bwd(n, i["sh_degree"], _p(fw["dirs"]), _p(i["shs48"]), _p(fw["masks"]), _p(v_shcol),
                 _p(v_shs), 0, _p(v_dirs))
    v_means, v_quats, v_scales = np.zeros((n, 3), np.float32), np.zeros((n, 4), np.float32), np.zeros((n, 3), np.float32)
    f = ctypes.c_float
    L.orc_project_bwd(n, _p(i["means"]), _p(i["quats"]), _p(i["scales"]), _p(i["viewmat"]), _p(i["K"]),
                      width, height, f(0.3), _p(fw["radii"]), _p(v_m2), None, _p(v_con), _p(v_means),
                      _p(v_quats), _p(v_scales))
    v_means = v_means + v_dirs
    return dict(loss=float(loss), v_image=v_img, v_means2d=v_m2, v_conics=v_con, v_colors=v_col,
                v_opac=v_op, v_shs48=v_shs, v_means=v_means, v_quats=v_quats, v_scales=v_scales)
