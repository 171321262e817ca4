"""Trajectory / evaluation renderer (reference: render_bigcity_images.py:149-268, 638-722, 753-1024).

A trained model (any of the three strategies) is rendered along a camera path: frames are cameras at
equal arc-length steps of a closed polyline at a fixed height, all with one fixed rotation; every frame
goes through the strategy's own single-camera eval entry (`*_eval_one_cam`, the same HIP kernels as
training) and is written as an 8-bit PNG.

    python -m clm_gs_amd.render_trajectory -m <model dir | .ply> --clm_offload --n_frames 120 \
        --manual_height 30 --width 1920 --height 1080 [--output_dir DIR] [--hull x,y x,y ...]

`render_single_image` keeps the reference's signature and return value ([H,W,3] in [0,1] on the GPU).
No imageio / PIL in this image: PNGs are written with zlib (8-bit RGB, no interlace).
"""
import argparse
import math
import os
import struct
import sys
import zlib

import numpy as np
import torch

from . import utils
from .cameras import Camera

# the BigCity hull of the reference (render_bigcity_images.py:172-182), closed: (x, y) world coordinates
BIGCITY_HULL = ((0.0, 35.0), (10.0, 30.0), (15.0, 20.0), (12.0, 0.0), (-7.0, 0.0), (-20.0, 20.0), (0.0, 35.0))
# the fixed camera-to-world rotation main() ends up using (render_bigcity_images.py:934-936): looking down -z
R_LOOK_DOWN = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0]])


def polyline_trajectory(R_fixed, height_z, n_frames, FoVx, FoVy, width, height, hull=BIGCITY_HULL, device="cuda"):
    """Cameras at arc length i / n_frames x perimeter along `hull` (a closed (x, y) polyline) at z = height_z
    (generate_convex_hull_trajectory_v2).  R_fixed is the camera-to-world rotation (the 3DGS `R` convention:
    world->camera rotation = R^T, T = -R^T C)."""
    pts = np.array([[x, y, height_z] for x, y in hull], dtype=np.float64)
    seg = np.linalg.norm(pts[1:] - pts[:-1], axis=1)
    cum = np.concatenate(([0.0], np.cumsum(seg)))
    total = cum[-1]
    R = np.asarray(R_fixed, dtype=np.float64)
    cams = []
    for i in range(int(n_frames)):
        d = (i / float(n_frames)) * total
        k = int(np.searchsorted(cum, d, side="left"))
        k = min(max(k, 1), len(cum) - 1)
        if cum[k - 1] > d:  # d sits exactly on an earlier vertex
            k -= 1
        a = (d - cum[k - 1]) / seg[k - 1] if seg[k - 1] > 0 else 0.0
        C = (1 - a) * pts[k - 1] + a * pts[k]
        w2c = np.eye(4)
        w2c[:3, :3] = R.T
        w2c[:3, 3] = -R.T @ C
        cams.append(Camera(i, torch.from_numpy(w2c).float(), FoVx, FoVy, width, height,
                           image_name=f"convex_hull_frame_{i:05d}", device=device))
    return cams


def write_png(path, rgb_u8):
    """8-bit RGB PNG from a [H,W,3] uint8 array."""
    a = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
    h, w, c = a.shape
    assert c == 3
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(png)


def read_png(path):
    """Inverse of write_png (8-bit RGB, filter 0 rows) -- used by the tests."""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(b):
        n, tag = struct.unpack(">I", b[pos:pos + 4])[0], b[pos + 4:pos + 8]
        data = b[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h = struct.unpack(">II", data[:8])
        elif tag == b"IDAT":
            idat += data
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert not raw[:, 0].any()
    return raw[:, 1:].reshape(h, w, 3).copy()


def render_single_image(camera, gaussians, background, save_path, args, scene=None):
    """One frame through the strategy's eval entry, clamped to [0,1], saved to `save_path` unless
    args.save_video (render_bigcity_images.py:638-722).  -> colors [H,W,3] on the GPU."""
    if getattr(args, "naive_offload", False):
        from .strategies.naive_offload import naive_offload_eval_one_cam
        img = naive_offload_eval_one_cam(gaussians=gaussians, scene=scene, camera=camera, background=background)
    elif getattr(args, "clm_offload", False):
        from .strategies.clm_offload import clm_offload_eval_one_cam
        img = clm_offload_eval_one_cam(camera=camera, gaussians=gaussians, background=background, scene=scene)
    elif getattr(args, "no_offload", False):
        from .strategies.no_offload import baseline_accumGrads_micro_step
        with torch.no_grad():
            img, _, _, _ = baseline_accumGrads_micro_step(
                means3D=gaussians.get_xyz, opacities=gaussians.get_opacity, scales=gaussians.get_scaling,
                rotations=gaussians.get_rotation, shs=gaussians.get_features, sh_degree=gaussians.active_sh_degree,
                camera=camera, background=background, mode="eval")
    else:
        raise ValueError("Invalid offload configuration")
    colors = torch.clamp(img, 0.0, 1.0)
    if colors.shape[0] == 3:
        colors = colors.permute(1, 2, 0)
    if save_path and not getattr(args, "save_video", False):
        write_png(save_path, (colors * 255).to(torch.uint8).cpu().numpy())
    return colors


def render_trajectory(gaussians, cameras, args, output_dir, background=None, log=None):
    """All frames of a path -> output_dir/frame_%05d.png; returns the list of paths."""
    os.makedirs(output_dir, exist_ok=True)
    paths = []
    for i, cam in enumerate(cameras):
        p = os.path.join(output_dir, f"frame_{i:05d}.png")
        render_single_image(cam, gaussians, background, p, args, scene=None)
        paths.append(p)
    if log is not None:
        log.write(f"Rendering completed. Images saved to: {output_dir}\n")
    return paths


def _find_ply(model_path, iteration):
    if model_path.endswith(".ply"):
        return model_path
    pc = os.path.join(model_path, "point_cloud")
    if iteration < 0:  # searchForMaxIteration (utils/system_utils.py)
        its = [int(d.split("_")[-1]) for d in os.listdir(pc) if d.startswith("iteration_")]
        iteration = max(its)
    return os.path.join(pc, f"iteration_{iteration}", "point_cloud.ply")


def main(argv=None):
    ap = argparse.ArgumentParser(description="Trajectory rendering")
    ap.add_argument("-m", "--model_path", required=True, help="model directory (point_cloud/iteration_*/point_cloud.ply) or a .ply")
    ap.add_argument("--iteration", type=int, default=-1)
    ap.add_argument("--traj_path", default="original")
    ap.add_argument("--manual_height", type=float, default=30.0)
    ap.add_argument("--n_frames", type=int, default=120)
    ap.add_argument("--output_dir", default=None)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--fovx", type=float, default=None, help="radians; default: focal = 0.8 x width")
    ap.add_argument("--hull", nargs="*", default=None, help="closed polyline as x,y pairs (default: the BigCity hull)")
    ap.add_argument("--white_background", action="store_true")
    ap.add_argument("--save_video", action="store_true", help="skip the per-frame PNGs (no video writer in this image)")
    g = ap.add_mutually_exclusive_group()
    g.add_argument("--clm_offload", action="store_true")
    g.add_argument("--naive_offload", action="store_true")
    g.add_argument("--no_offload", action="store_true")
    a = ap.parse_args(argv)
    if not (a.clm_offload or a.naive_offload or a.no_offload):
        a.clm_offload = True
    args = utils.default_args(bsz=4)
    for k in ("clm_offload", "naive_offload", "no_offload", "save_video"):
        setattr(args, k, getattr(a, k))
    utils.set_args(args)
    utils.set_img_size(a.height, a.width)
    if a.clm_offload:
        from .strategies.clm_offload import GaussianModelCLMOffload as M
    elif a.naive_offload:
        from .strategies.naive_offload import GaussianModelNaiveOffload as M
    else:
        from .strategies.no_offload import GaussianModelNoOffload as M
    gaussians = M(3, only_for_rendering=True)
    gaussians.load_ply(_find_ply(a.model_path, a.iteration))
    gaussians.active_sh_degree = gaussians.max_sh_degree
    out = a.output_dir or os.path.join(a.model_path if os.path.isdir(a.model_path) else os.path.dirname(a.model_path),
                                       "render_images")
    os.makedirs(out, exist_ok=True)
    fovx = a.fovx if a.fovx is not None else 2 * math.atan(a.width / (2 * 0.8 * a.width))
    fovy = 2 * math.atan(math.tan(fovx / 2) * a.height / a.width)
    hull = BIGCITY_HULL if not a.hull else tuple(tuple(float(v) for v in p.split(",")) for p in a.hull)
    cams = polyline_trajectory(R_LOOK_DOWN, a.manual_height, a.n_frames, fovx, fovy, a.width, a.height, hull)
    bg = torch.tensor([1.0, 1.0, 1.0], device="cuda") if a.white_background else None
    with open(os.path.join(out, "render_images.log"), "w") as log:
        log.write(f"Model path: {a.model_path}\nTrajectory type: {a.traj_path}\nNumber of frames: {a.n_frames}\n")
        with torch.no_grad():
            paths = render_trajectory(gaussians, cams, args, out, bg, log)
    print(f"Images saved to: {out} ({len(paths)} frames)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
