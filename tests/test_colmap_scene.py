"""Row f1: clm_gs_amd.colmap_scene against what the reference's OWN reader made of the same tiny COLMAP
model (tests/golden/make_colmap_golden.py ran scene/dataset_readers.py::readColmapSceneInfo in the build
container; colmap_expected.json holds its cameras, split, scene radius and points)."""
import json
import os
import shutil

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SRC = os.path.join(G, "colmap_tiny")
EXP = json.load(open(os.path.join(G, "colmap_expected.json")))


def _check_cameras(scene, exp):
    assert [c.image_name for c in scene.train_cameras] == exp["train"]
    assert [c.image_name for c in scene.test_cameras] == exp["test"]
    # getWorld2View2 builds float32 matrices: the reference's radius / centre carry float32 rounding
    assert abs(scene.cameras_extent - exp["radius"]) < 1e-6 * max(1.0, exp["radius"])
    assert np.allclose(scene.nerf_normalization["translate"], exp["translate"], atol=1e-6)
    for c in scene.train_cameras + scene.test_cameras:
        e = exp["cameras"][c.image_name]
        # reference: R = qvec2rotmat(q).T, T = tvec, world_view_transform = getWorld2View2(R, T).T -> the
        # world->camera matrix is [R.T | T]; this package's Camera stores its transpose the same way
        w2c = np.eye(4)
        w2c[:3, :3] = np.asarray(e["R"]).T
        w2c[:3, 3] = e["T"]
        assert np.allclose(c.world_view_transform.cpu().numpy().T, w2c, atol=1e-6)
        assert abs(c.FoVx - e["FovX"]) < 1e-12 and abs(c.FoVy - e["FovY"]) < 1e-12
        assert (c.image_width, c.image_height, c.uid) == (e["width"], e["height"], e["uid"])


@pytest.mark.parametrize("eval_split", [True, False])
def test_binary_model_matches_reference_reader(eval_split):
    from clm_gs_amd.colmap_scene import load_colmap_scene
    scene = load_colmap_scene(SRC, eval=eval_split, device="cpu")
    _check_cameras(scene, EXP["eval" if eval_split else "all"])
    assert np.allclose(scene.point_cloud.points, np.asarray(EXP["points"]["xyz"], dtype=np.float32))
    assert np.array_equal(np.round(scene.point_cloud.colors * 255).astype(np.int64), np.asarray(EXP["points"]["rgb"]))
    from PIL import Image
    cam = scene.train_cameras[0]
    assert cam.original_image.dtype == torch.uint8 and tuple(cam.original_image.shape) == (3, 16, 24)
    ref = np.asarray(Image.open(os.path.join(SRC, "images", cam.image_name + ".png")).convert("RGB"))
    assert np.array_equal(cam.original_image.permute(1, 2, 0).numpy(), ref)


def test_text_model_equals_binary_model(tmp_path):
    """sparse/0 with only the .txt files (the reference's fallback) gives the same scene."""
    from clm_gs_amd.colmap_scene import load_colmap_scene, read_model
    work = tmp_path / "scene"
    os.makedirs(work / "sparse")
    shutil.copytree(os.path.join(SRC, "sparse_txt", "0"), work / "sparse" / "0")
    shutil.copytree(os.path.join(SRC, "images"), work / "images")
    scene = load_colmap_scene(str(work), eval=True, device="cpu")
    _check_cameras(scene, EXP["eval"])
    cb, ib, (xb, rb) = read_model(os.path.join(SRC, "sparse", "0"))
    ct, it, (xt, rt) = read_model(str(work / "sparse" / "0"))
    assert cb.keys() == ct.keys() and ib.keys() == it.keys()
    for k in cb:
        assert (cb[k].model, cb[k].width, cb[k].height) == (ct[k].model, ct[k].width, ct[k].height)
        assert np.array_equal(cb[k].params, ct[k].params)
    for k in ib:
        assert ib[k].name == it[k].name and ib[k].camera_id == it[k].camera_id
        assert np.array_equal(ib[k].qvec, it[k].qvec) and np.array_equal(ib[k].tvec, it[k].tvec)
    assert np.array_equal(xb, xt) and np.array_equal(rb, rt)


def test_errors_and_options(tmp_path):
    from clm_gs_amd import colmap_scene as C
    with pytest.raises(FileNotFoundError):
        C.read_model(str(tmp_path))
    cam = C.SimpleNamespace(model="RADIAL", width=10, height=10, params=np.zeros(5))
    with pytest.raises(ValueError, match="not handled"):
        C._intrinsics_to_fov(cam)
    with open(tmp_path / "cameras.bin", "wb") as f:
        f.write(b"\x02\x00\x00\x00\x00\x00\x00\x00\x01\x00")
    with pytest.raises(ValueError, match="truncated"):
        C.read_cameras_binary(str(tmp_path / "cameras.bin"))
    half = C.load_colmap_scene(SRC, resolution=2, device="cpu")
    assert (half.train_cameras[0].image_width, half.train_cameras[0].image_height) == (12, 8)
    lazy = C.load_colmap_scene(SRC, load_images=False, device="cpu")
    assert lazy.train_cameras[0].original_image is None and lazy.train_cameras[0].image_width == 24


@pytest.mark.gpu
@pytest.mark.parametrize("strategy", ["clm_offload", "no_offload"])
def test_training_from_a_colmap_directory(tmp_path, strategy):
    """End to end through trainer.train_from_colmap: COLMAP directory (poses + sparse points of the
    fixture, images RENDERED by this engine from a hidden perturbed model and written as PNGs) -> cameras
    -> create_from_pcd (distCUDA2 scales) -> training loop with the reference's log lines -> .ply.  The
    loss against the dataset's images falls and the saved model reloads."""
    from PIL import Image
    from clm_gs_amd import trainer, utils
    from clm_gs_amd.colmap_scene import load_colmap_scene
    from clm_gs_amd.io_ply import load_ply
    from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload, clm_offload_eval_one_cam
    work = tmp_path / "scene"
    shutil.copytree(SRC, work)
    # poses of the fixture are random: put the "true" geometry where the cameras can see it -- a blob in
    # front of every camera -- by rewriting points3D and rendering the dataset images from it
    poses = load_colmap_scene(str(work), device="cuda", load_images=False)
    g = torch.Generator().manual_seed(11)
    pts = []
    for c in poses.train_cameras:
        c2w = c.camtoworlds[0].cpu()
        local = torch.cat([torch.randn(150, 2, generator=g) * 0.6, 4.0 + torch.rand(150, 1, generator=g)], 1)
        pts.append(local @ c2w[:3, :3].T + c2w[:3, 3])
    pts = torch.cat(pts).numpy().astype(np.float64)
    rgb = (torch.rand(len(pts), 3, generator=g) * 255).to(torch.uint8).numpy()
    with open(work / "sparse" / "0" / "points3D.txt", "w") as f:
        for i, (p, cc) in enumerate(zip(pts, rgb)):
            f.write(f"{i + 1} {float(p[0])!r} {float(p[1])!r} {float(p[2])!r} {int(cc[0])} {int(cc[1])} {int(cc[2])} 0.5 1 0\n")
    os.remove(work / "sparse" / "0" / "points3D.bin")  # the text file is the fallback the loader takes now
    args = utils.default_args(bsz=4, clm_offload=True)
    utils.set_args(args)
    utils.set_img_size(16, 24)
    utils.set_cur_iter(1)
    scene = load_colmap_scene(str(work), device="cuda", load_images=False)
    truth = GaussianModelCLMOffload(3, only_for_rendering=True)
    truth.args = args
    truth.create_from_pcd(scene.point_cloud, scene.cameras_extent)
    truth.active_sh_degree = 0
    with torch.no_grad():
        truth._scaling += 0.7
        truth._opacity += 2.0
    for c in scene.train_cameras:
        img = (clm_offload_eval_one_cam(c, truth, None, None).clamp(0, 1) * 255).round().to(torch.uint8)
        assert int(img.max()) > 0, "every camera sees the blob in front of it"
        Image.fromarray(img.permute(1, 2, 0).cpu().numpy()).save(work / "images" / (c.image_name + ".png"))
    out = tmp_path / "out"
    gaussians, sc, timer = trainer.train_from_colmap(
        str(work), str(out), strategy=strategy, iterations=160, eval=True, test_iterations=(1, 157),
        bsz=4, disable_auto_densification=True)
    log = open(out / "python_ws=1_rk=0.log").read()
    psnr = [float(ln.split("PSNR ")[1]) for ln in log.splitlines() if "Evaluating train:" in ln]
    assert len(psnr) == 2 and psnr[1] > psnr[0] + 0.5, psnr
    assert "Evaluating test:" in log and "end2end total_time:" in log
    assert len(sc.test_cameras) == 2 and len(sc.train_cameras) == 10
    saved = load_ply(str(out / "point_cloud" / "iteration_160" / "point_cloud.ply"))
    assert saved["xyz"].shape[0] == gaussians.get_xyz.shape[0] == len(pts)
