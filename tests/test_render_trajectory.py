"""Trajectory renderer (row f4; reference render_bigcity_images.py:149-268, 638-722).

CPU: the camera path == what the reference's own generate_convex_hull_trajectory_v2 produced
(tests/golden/trajectory_expected.json, written by tests/golden/make_trajectory_golden.py), PNG round trip.
GPU: frames rendered along a path through every strategy's eval entry == direct eval renders, files on disk."""
import json
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_polyline_trajectory_matches_reference_generator():
    from clm_gs_amd.render_trajectory import BIGCITY_HULL, R_LOOK_DOWN, polyline_trajectory
    ref = json.load(open(os.path.join(G, "trajectory_expected.json")))
    assert np.array_equal(np.array(ref["R_fixed"], dtype=np.float64), R_LOOK_DOWN)
    for case in ref["cases"]:
        cams = polyline_trajectory(R_LOOK_DOWN, case["height_z"], case["n_frames"], case["FoVx"], case["FoVy"],
                                   case["width"], case["height"], hull=BIGCITY_HULL, device="cpu")
        assert len(cams) == len(case["cameras"]) == case["n_frames"]
        for c, r in zip(cams, case["cameras"]):
            assert c.image_name == r["image_name"] and c.uid == r["uid"]
            wvt = np.array(r["world_view_transform"])
            assert np.allclose(c.world_view_transform.numpy(), wvt, atol=2e-5), (c.image_name,)
            centre = torch.inverse(c.world_view_transform.t().double())[:3, 3].numpy()
            assert np.allclose(centre, np.array(r["centre"]), atol=1e-4)
            assert c.image_width == case["width"] and abs(c.FoVx - case["FoVx"]) < 1e-12


def test_png_round_trip(tmp_path):
    from clm_gs_amd.render_trajectory import read_png, write_png
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    p = str(tmp_path / "a" / "f.png")
    write_png(p, img)
    assert np.array_equal(read_png(p), img)


@pytest.mark.gpu
@pytest.mark.parametrize("strategy", ["clm_offload", "no_offload", "naive_offload"])
def test_render_trajectory_frames_equal_direct_eval(dev, tmp_path, strategy):
    from clm_gs_amd import utils
    from clm_gs_amd.render_trajectory import (R_LOOK_DOWN, polyline_trajectory, read_png, render_single_image,
                                              render_trajectory)
    from clm_gs_amd.synthetic import synth_gaussians
    W, H, N = 96, 64, 3000
    args = utils.default_args(bsz=4)
    setattr(args, strategy, True)
    args.save_video = False
    utils.set_args(args)
    utils.set_img_size(H, W)
    utils.set_cur_iter(1)
    sc = synth_gaussians(N, seed=0, device="cuda")
    if strategy == "clm_offload":
        from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload as M
    elif strategy == "no_offload":
        from clm_gs_amd.strategies.no_offload import GaussianModelNoOffload as M
    else:
        from clm_gs_amd.strategies.naive_offload import GaussianModelNaiveOffload as M
    m = M(3)
    m.create_from_tensors(sc["xyz"], sc["shs48"], sc["scaling"], sc["rotation"], sc["opacity"], spatial_lr_scale=1.0)
    m.active_sh_degree = 3
    L = sc["extent"]
    hull = ((-0.3 * L, -0.3 * L), (0.3 * L, -0.3 * L), (0.3 * L, 0.3 * L), (-0.3 * L, 0.3 * L), (-0.3 * L, -0.3 * L))
    cams = polyline_trajectory(R_LOOK_DOWN, 0.1 * L + 25.0, 5, 1.1, 0.8, W, H, hull=hull)
    # the nadir rotation of the synthetic scenes looks down -z with y flipped; R_LOOK_DOWN looks down -z as well
    with torch.no_grad():
        paths = render_trajectory(m, cams, args, str(tmp_path / "frames"))
        assert [os.path.basename(p) for p in paths] == [f"frame_{i:05d}.png" for i in range(5)]
        seen = 0
        for cam, p in zip(cams, paths):
            col = render_single_image(cam, m, None, None, args)
            assert col.shape == (H, W, 3) and float(col.min()) >= 0.0 and float(col.max()) <= 1.0
            assert np.array_equal(read_png(p), (col * 255).to(torch.uint8).cpu().numpy())
            seen += int(col.sum() > 0)
        assert seen >= 3, "the path looks at the scene"
    # the three strategies' eval entries render the same frame
    if strategy != "clm_offload":
        from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload, clm_offload_eval_one_cam
        ref = GaussianModelCLMOffload(3, only_for_rendering=True)
        ref.create_from_tensors(sc["xyz"], sc["shs48"], sc["scaling"], sc["rotation"], sc["opacity"])
        ref.active_sh_degree = 3
        a = render_single_image(cams[1], m, None, None, args)
        b = clm_offload_eval_one_cam(cams[1], ref, None, None).clamp(0, 1).permute(1, 2, 0)
        assert float((a - b).abs().max()) < 1e-4
