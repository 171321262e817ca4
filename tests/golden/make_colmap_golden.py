"""Row f1 (COLMAP-lite loader): writes a tiny synthetic COLMAP model (tests/golden/colmap_tiny: binary AND
text model files + 24x16 PNG images -- data this script generates, no reference content) and records what
the reference's OWN reader makes of it (build container only):

    python tests/golden/make_colmap_golden.py

`scene/dataset_readers.py::readColmapSceneInfo` (-> readColmapCameras, getNerfppNorm, the llffhold split)
and `scene/colmap_loader.py::read_points3D_binary` are imported from /root/reference and run on the model;
their outputs go to tests/golden/colmap_expected.json.  The reference imports `plyfile` (absent in this
image) at module level; an empty stand-in module satisfies the import -- nothing of it is called: the
.ply conversion is skipped because a points3D.ply placeholder exists in the temporary copy the reader sees.
"""
import importlib.util
import json
import os
import shutil
import struct
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
G = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(G, "colmap_tiny")


def synth_model(seed=7, n_img=12, n_pts=40):
    rng = np.random.default_rng(seed)
    cams = {1: ("PINHOLE", 24, 16, [30.0, 28.0, 12.0, 8.0]), 2: ("SIMPLE_PINHOLE", 24, 16, [26.0, 12.0, 8.0]),
            3: ("OPENCV", 24, 16, [31.0, 29.0, 12.0, 8.0, 0.01, -0.02, 0.001, 0.002])}
    imgs = {}
    for i in range(n_img):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        t = rng.normal(size=3) * 3.0
        imgs[i + 1] = (q, t, 1 + i % 3, f"view_{(n_img - i):03d}.png",  # names NOT in id order
                       [(float(rng.uniform(0, 24)), float(rng.uniform(0, 16)), int(rng.integers(-1, n_pts)))
                        for _ in range(int(rng.integers(0, 5)))])
    pts = {}
    for j in range(n_pts):
        pts[j + 10] = (rng.normal(size=3) * 2.0, rng.integers(0, 256, size=3), float(rng.uniform(0, 2)),
                       [(int(rng.integers(1, n_img + 1)), int(rng.integers(0, 4))) for _ in range(int(rng.integers(1, 4)))])
    return cams, imgs, pts


MODEL_IDS = {"SIMPLE_PINHOLE": 0, "PINHOLE": 1, "OPENCV": 4}


def write_binary(d, cams, imgs, pts):
    with open(os.path.join(d, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cams)))
        for cid, (model, w, h, par) in cams.items():
            f.write(struct.pack("<iiQQ", cid, MODEL_IDS[model], w, h))
            f.write(struct.pack(f"<{len(par)}d", *par))
    with open(os.path.join(d, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(imgs)))
        for iid, (q, t, cid, name, obs) in imgs.items():
            f.write(struct.pack("<i4d3di", iid, *q, *t, cid))
            f.write(name.encode() + b"\x00")
            f.write(struct.pack("<Q", len(obs)))
            for x, y, pid in obs:
                f.write(struct.pack("<ddq", x, y, pid))
    with open(os.path.join(d, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(pts)))
        for pid, (xyz, rgb, err, track) in pts.items():
            f.write(struct.pack("<Q3d3BdQ", pid, *xyz, *[int(c) for c in rgb], err, len(track)))
            for a, b in track:
                f.write(struct.pack("<ii", a, b))


def write_text(d, cams, imgs, pts):
    with open(os.path.join(d, "cameras.txt"), "w") as f:
        f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n")
        for cid, (model, w, h, par) in cams.items():
            f.write(f"{cid} {model} {w} {h} " + " ".join(repr(float(p)) for p in par) + "\n")
    with open(os.path.join(d, "images.txt"), "w") as f:
        f.write("# Image list with two lines of data per image:\n")
        for iid, (q, t, cid, name, obs) in imgs.items():
            f.write(f"{iid} " + " ".join(repr(float(v)) for v in list(q) + list(t)) + f" {cid} {name}\n")
            f.write(" ".join(f"{x!r} {y!r} {pid}" for x, y, pid in obs) + "\n")
    with open(os.path.join(d, "points3D.txt"), "w") as f:
        f.write("# 3D point list with one line of data per point:\n")
        for pid, (xyz, rgb, err, track) in pts.items():
            f.write(f"{pid} " + " ".join(repr(float(v)) for v in xyz) + " " + " ".join(str(int(c)) for c in rgb)
                    + f" {err!r} " + " ".join(f"{a} {b}" for a, b in track) + "\n")


def write_images(d, imgs, seed=3):
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(d, exist_ok=True)
    for _, (_, _, _, name, _) in imgs.items():
        Image.fromarray(rng.integers(0, 256, size=(16, 24, 3), dtype=np.uint8)).save(os.path.join(d, name))


def main():
    cams, imgs, pts = synth_model()
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(os.path.join(OUT, "sparse", "0"))
    os.makedirs(os.path.join(OUT, "sparse_txt", "0"))
    write_binary(os.path.join(OUT, "sparse", "0"), cams, imgs, pts)
    write_text(os.path.join(OUT, "sparse_txt", "0"), cams, imgs, pts)
    write_images(os.path.join(OUT, "images"), imgs)

    # ---- the reference's reader on a temporary copy (it wants to write points3D.ply next to the model)
    sys.path.insert(0, REF)
    sys.modules.setdefault("plyfile", types.SimpleNamespace(PlyData=None, PlyElement=None))
    pkg = types.ModuleType("scene")  # the package's __init__ pulls the whole training stack: load two files only
    pkg.__path__ = [os.path.join(REF, "scene")]
    sys.modules["scene"] = pkg
    for name in ("colmap_loader", "dataset_readers"):
        spec = importlib.util.spec_from_file_location(f"scene.{name}", os.path.join(REF, "scene", f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"scene.{name}"] = mod
        spec.loader.exec_module(mod)
    import utils.general_utils as gu
    dr, cl = sys.modules["scene.dataset_readers"], sys.modules["scene.colmap_loader"]
    gu.set_args(types.SimpleNamespace(dense_ply_file="", load_pt_path="skip"))
    expected = {}
    with tempfile.TemporaryDirectory() as tmp:
        work = os.path.join(tmp, "scene")
        shutil.copytree(OUT, work)
        open(os.path.join(work, "sparse", "0", "points3D.ply"), "w").close()  # placeholder: no conversion
        for tag, ev in (("eval", True), ("all", False)):
            info = dr.readColmapSceneInfo(work, None, ev)  # llffhold: the reference's default
            expected[tag] = {
                "train": [c.image_name for c in info.train_cameras], "test": [c.image_name for c in info.test_cameras],
                "radius": float(info.nerf_normalization["radius"]),
                "translate": [float(v) for v in info.nerf_normalization["translate"]],
                "cameras": {c.image_name: {"uid": int(c.uid), "R": np.asarray(c.R).tolist(), "T": np.asarray(c.T).tolist(),
                                           "FovX": float(c.FovX), "FovY": float(c.FovY), "width": int(c.width),
                                           "height": int(c.height)}
                            for c in list(info.train_cameras) + list(info.test_cameras)}}
        xyz, rgb, err = cl.read_points3D_binary(os.path.join(work, "sparse", "0", "points3D.bin"))
        expected["points"] = {"xyz": np.asarray(xyz).tolist(), "rgb": np.asarray(rgb).tolist(),
                              "error": np.asarray(err).reshape(-1).tolist()}
    with open(os.path.join(G, "colmap_expected.json"), "w") as f:
        json.dump(expected, f)
    print("wrote", OUT, "and colmap_expected.json:", len(expected["eval"]["train"]), "train /",
          len(expected["eval"]["test"]), "test cameras,", len(expected["points"]["xyz"]), "points")


if __name__ == "__main__":
    main()
