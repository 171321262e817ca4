"""COLMAP-lite scene loader: a COLMAP model directory -> the cameras, initial
point cloud and scene radius the engines train from, so a real dataset runs when one is present.

Behaviour follows the reference's reader (scene/dataset_readers.py:59-252, scene/colmap_loader.py):
`<source>/sparse/0/{cameras,images,points3D}.bin` (text `.txt` as the fallback), cameras of the models
SIMPLE_PINHOLE / PINHOLE / OPENCV (distortion ignored, as there), images sorted by name, every
`llffhold`-th image held out when `eval`, scene radius = 1.1 x the largest distance of a training camera
centre from their mean (getNerfppNorm).  The file formats are COLMAP's published model formats
(https://colmap.github.io/format.html); the parsing below is written against that description with
numpy record reads.  Images are decoded with PIL to uint8 [3,H,W] (what the engines' fused loss reads).
"""
import math
import os
from types import SimpleNamespace

import numpy as np
import torch

from .cameras import Camera

# COLMAP camera models: id -> (name, number of parameters)
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5),
                 4: ("OPENCV", 8), 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5),
                 8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}
_MODEL_BY_NAME = {name: (mid, n) for mid, (name, n) in CAMERA_MODELS.items()}


class _Reader:
    """Little-endian cursor over a whole model file held in memory."""

    def __init__(self, path):
        with open(path, "rb") as f:
            self.buf = f.read()
        self.pos = 0

    def take(self, dtype, count=1):
        dt = np.dtype(dtype).newbyteorder("<")
        n = dt.itemsize * count
        if self.pos + n > len(self.buf):
            raise ValueError("truncated COLMAP model file")
        out = np.frombuffer(self.buf, dtype=dt, count=count, offset=self.pos)
        self.pos += n
        return out

    def cstring(self):
        end = self.buf.index(b"\x00", self.pos)
        s = self.buf[self.pos:end].decode("utf-8")
        self.pos = end + 1
        return s


def read_cameras_binary(path):
    """cameras.bin -> {camera_id: namespace(id, model, width, height, params)}."""
    r = _Reader(path)
    out = {}
    for _ in range(int(r.take("u8")[0])):
        cam_id, model_id = (int(v) for v in r.take("i4", 2))
        width, height = (int(v) for v in r.take("u8", 2))
        if model_id not in CAMERA_MODELS:
            raise ValueError(f"unknown COLMAP camera model id {model_id}")
        name, n_par = CAMERA_MODELS[model_id]
        out[cam_id] = SimpleNamespace(id=cam_id, model=name, width=width, height=height,
                                      params=r.take("f8", n_par).copy())
    return out


def read_images_binary(path):
    """images.bin -> {image_id: namespace(id, qvec[w,x,y,z], tvec, camera_id, name)} (the 2-D
    observations are skipped)."""
    r = _Reader(path)
    out = {}
    for _ in range(int(r.take("u8")[0])):
        image_id = int(r.take("i4")[0])
        pose = r.take("f8", 7).copy()
        camera_id = int(r.take("i4")[0])
        name = r.cstring()
        n_obs = int(r.take("u8")[0])
        r.pos += 24 * n_obs  # (x f8, y f8, point3D_id i8) per observation
        out[image_id] = SimpleNamespace(id=image_id, qvec=pose[:4], tvec=pose[4:], camera_id=camera_id, name=name)
    return out


def read_points3d_binary(path):
    """points3D.bin -> (xyz float64 [n,3], rgb uint8 [n,3], error float64 [n])."""
    r = _Reader(path)
    n = int(r.take("u8")[0])
    xyz, rgb, err = np.empty((n, 3)), np.empty((n, 3), dtype=np.uint8), np.empty((n,))
    for i in range(n):
        r.pos += 8  # point id
        xyz[i] = r.take("f8", 3)
        rgb[i] = r.take("u1", 3)
        err[i] = r.take("f8")[0]
        track_len = int(r.take("u8")[0])
        r.pos += 8 * track_len  # track: (image_id i4, point2D_idx i4) each
    return xyz, rgb, err


def _data_lines(path):
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if line and not line.startswith("#"):
                yield line


def read_cameras_text(path):
    out = {}
    for line in _data_lines(path):
        t = line.split()
        if t[1] not in _MODEL_BY_NAME:
            raise ValueError(f"unknown COLMAP camera model {t[1]}")
        out[int(t[0])] = SimpleNamespace(id=int(t[0]), model=t[1], width=int(t[2]), height=int(t[3]),
                                         params=np.array([float(v) for v in t[4:]]))
    return out


def read_images_text(path):
    """images.txt holds TWO lines per image (pose line, observations line; the second may be empty)."""
    out = {}
    with open(path, "r") as f:
        lines = [ln.rstrip("\n") for ln in f if not ln.lstrip().startswith("#")]
    i = 0
    while i < len(lines):
        if not lines[i].strip():
            i += 1
            continue
        t = lines[i].split()
        out[int(t[0])] = SimpleNamespace(id=int(t[0]), qvec=np.array([float(v) for v in t[1:5]]),
                                         tvec=np.array([float(v) for v in t[5:8]]), camera_id=int(t[8]),
                                         name=" ".join(t[9:]))
        i += 2
    return out


def read_points3d_text(path):
    rows = [ln.split() for ln in _data_lines(path)]
    xyz = np.array([[float(v) for v in t[1:4]] for t in rows]).reshape(-1, 3)
    rgb = np.array([[int(v) for v in t[4:7]] for t in rows], dtype=np.uint8).reshape(-1, 3)
    err = np.array([float(t[7]) for t in rows])
    return xyz, rgb, err


def quat_to_rotation(q):
    """COLMAP quaternion (w, x, y, z), world -> camera rotation matrix."""
    w, x, y, z = (float(v) for v in q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def focal_to_fov(focal, pixels):
    return 2.0 * math.atan(pixels / (2.0 * focal))


def _intrinsics_to_fov(cam):
    if cam.model == "SIMPLE_PINHOLE":
        fx = fy = cam.params[0]
    elif cam.model in ("PINHOLE", "OPENCV"):  # OPENCV: the four distortion terms are ignored, as upstream
        fx, fy = cam.params[0], cam.params[1]
    else:
        raise ValueError(f"COLMAP camera model {cam.model} is not handled: undistort the dataset "
                         "(PINHOLE / SIMPLE_PINHOLE / OPENCV cameras only)")
    return focal_to_fov(fx, cam.width), focal_to_fov(fy, cam.height)


def scene_radius(world_to_cams):
    """(translate, radius) of getNerfppNorm: radius = 1.1 x max distance of a camera centre from their mean."""
    centres = np.stack([np.linalg.inv(m)[:3, 3] for m in world_to_cams])
    mean = centres.mean(axis=0)
    return -mean, 1.1 * float(np.linalg.norm(centres - mean, axis=1).max())


def _decode_image(path, resolution):
    from PIL import Image
    with Image.open(path) as im:
        im = im.convert("RGB")
        if resolution not in (1, None):
            im = im.resize((round(im.width / resolution), round(im.height / resolution)))
        a = np.array(im, dtype=np.uint8)  # a writable copy
    return torch.from_numpy(a).permute(2, 0, 1).contiguous()


def read_model(sparse_dir):
    """(cameras, images, (xyz, rgb)) from a `sparse/0` directory: binary files first, text as the fallback."""
    def pick(stem, rb, rt):
        b, t = os.path.join(sparse_dir, stem + ".bin"), os.path.join(sparse_dir, stem + ".txt")
        if os.path.exists(b):
            return rb(b)
        if os.path.exists(t):
            return rt(t)
        raise FileNotFoundError(f"no {stem}.bin / {stem}.txt in {sparse_dir}")
    cams = pick("cameras", read_cameras_binary, read_cameras_text)
    imgs = pick("images", read_images_binary, read_images_text)
    try:
        xyz, rgb, _ = pick("points3D", read_points3d_binary, read_points3d_text)
    except FileNotFoundError:
        xyz, rgb = None, None
    return cams, imgs, (xyz, rgb)


def load_colmap_scene(source_path, images="images", eval=False, llffhold=10, resolution=1, device="cuda",
                      load_images=True):
    """-> namespace(train_cameras, test_cameras, point_cloud(points, colors in [0,1]) or None,
    cameras_extent, nerf_normalization).  `cameras_extent` is what the trainer passes as
    `spatial_lr_scale` and what densification compares scales with (train.py:118-131)."""
    cams, imgs, (xyz, rgb) = read_model(os.path.join(source_path, "sparse", "0"))
    folder = os.path.join(source_path, images or "images")
    recs = []
    for im in imgs.values():
        intr = cams[im.camera_id]
        fovx, fovy = _intrinsics_to_fov(intr)
        w2c = np.eye(4)
        w2c[:3, :3] = quat_to_rotation(im.qvec)
        w2c[:3, 3] = im.tvec
        base = os.path.basename(im.name)
        recs.append(SimpleNamespace(uid=intr.id, w2c=w2c, fovx=fovx, fovy=fovy, width=intr.width, height=intr.height,
                                    path=os.path.join(folder, base), name=base.split(".")[0]))
    recs.sort(key=lambda r: r.name)
    if eval:
        train = [r for i, r in enumerate(recs) if i % llffhold != 0]
        test = [r for i, r in enumerate(recs) if i % llffhold == 0]
    else:
        train, test = recs, []
    translate, radius = scene_radius([r.w2c for r in train])

    def build(r):
        img = None
        w, h = r.width, r.height
        if load_images:
            img = _decode_image(r.path, resolution)
            h, w = int(img.shape[1]), int(img.shape[2])  # the decoded size wins, as upstream (image.size)
        elif resolution not in (1, None):
            w, h = round(w / resolution), round(h / resolution)
        return Camera(r.uid, torch.from_numpy(r.w2c).float(), r.fovx, r.fovy, w, h, image_u8=img,
                      image_name=r.name, device=device)

    pcd = None
    if xyz is not None and len(xyz):
        pcd = SimpleNamespace(points=xyz.astype(np.float32), colors=rgb.astype(np.float32) / 255.0,
                              normals=np.zeros_like(xyz, dtype=np.float32))
    return SimpleNamespace(train_cameras=[build(r) for r in train], test_cameras=[build(r) for r in test],
                           point_cloud=pcd, cameras_extent=radius,
                           nerf_normalization={"translate": translate, "radius": radius})
