"""CPU-only: pin the oracle (and the product's host-side helpers) against golden vectors
produced by the REFERENCE's own importable utilities (tests/golden/make_golden.py)."""
import math
import os

import numpy as np
import torch

from oracle import gs_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def test_sh_matches_reference_eval_sh():
    d = _load("sh.npz")
    dirs = torch.from_numpy(d["dirs"])
    coeffs = torch.from_numpy(d["sh_CK"]).permute(0, 2, 1).contiguous()  # [n,C,K] -> [n,K,3]
    for deg in range(4):
        out = O.spherical_harmonics(deg, dirs, coeffs)
        assert np.abs(out.numpy() - d[f"deg{deg}"]).max() < 2e-6
    from clm_gs_amd import utils
    rgb = torch.from_numpy(d["rgb"])
    assert np.allclose(utils.RGB2SH(rgb).numpy(), d["rgb2sh"], atol=1e-6)
    assert np.allclose(utils.SH2RGB(rgb).numpy(), d["sh2rgb"], atol=1e-6)


def test_ssim_l1_psnr_match_reference():
    d = _load("loss.npz")
    a = torch.from_numpy(d["img1"]).requires_grad_()
    b = torch.from_numpy(d["img2"])
    s = O.fused_ssim(a, b)
    assert abs(s.item() - float(d["ssim"])) < 1e-6
    s.backward()
    assert np.abs(a.grad.numpy() - d["ssim_grad"]).max() < 1e-7
    assert abs((a.detach() - b).abs().mean().item() - float(d["l1"])) < 1e-7
    assert np.allclose(O.psnr(a.detach(), b).numpy(), d["psnr"], atol=1e-4)
    w = O._ssim_window(torch.float32).numpy()
    assert np.abs(w - d["window"][0, 0]).max() < 1e-8
    # the C oracle's loss uses the same definition
    from oracle import c_oracle as C
    import ctypes
    img = np.ascontiguousarray(d["img1"][0])
    gt_u8 = np.clip(np.round(d["img2"][0] * 255), 0, 255).astype(np.uint8)
    gt = torch.from_numpy(gt_u8)
    want = O.training_loss(torch.from_numpy(img), gt).item()
    got = C.lib().orc_loss(img.shape[1], img.shape[2], img.ctypes.data_as(ctypes.c_void_p),
                           gt_u8.ctypes.data_as(ctypes.c_void_p), None)
    assert abs(got - want) < 1e-6


def test_rotation_and_covariance_match_reference():
    d = _load("rotation.npz")
    q, s = torch.from_numpy(d["q"]), torch.from_numpy(d["s"])
    R = O.quat_to_rotmat(q)
    assert np.abs(R.numpy() - d["R"]).max() < 1e-6
    cov = O.quat_scale_to_covar(q, s)
    c6 = torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], -1)
    assert np.abs(c6.numpy() - d["cov6"]).max() < 1e-5
    from clm_gs_amd import utils
    assert np.abs(utils.build_rotation(q).numpy() - d["R"]).max() < 1e-6


def test_camera_conventions_match_reference():
    """world_view_transform = getWorld2View2(R,T).T (scene/cameras.py:87-91); the engines pass its
    transpose as viewmat, i.e. the plain world->camera matrix."""
    d = _load("camera.npz")
    from clm_gs_amd.cameras import Camera
    cam = Camera(0, d["w2v"], 0.9, 0.7, 1237, 822, device="cpu")
    assert np.abs(cam.world_view_transform.numpy() - d["w2v"].T).max() < 1e-6
    fx = 1237 / (2 * math.tan(0.45))
    assert abs(float(cam.K[0, 0]) - fx) < 1e-3
    assert abs(float(d["fov2focal"]) - fx) < 1e-3  # fov2focal(0.9, 1237)
    assert abs(2 * math.atan(822 / (2 * 1100.0)) - float(d["focal2fov"])) < 1e-9
    c2w = np.linalg.inv(d["w2v"])
    assert np.abs(cam.camtoworlds[0].numpy() - c2w).max() < 1e-4


def test_schedules_match_reference():
    d = _load("schedule.npz")
    from clm_gs_amd import utils
    f = utils.get_expon_lr_func(0.00016 * 2, 0.0000016 * 2, lr_delay_mult=0.01, max_steps=30000)
    assert np.allclose([f(int(s)) for s in d["steps"]], d["lr"], rtol=1e-12)
    for (bsz, iv), row in zip(d["trig_cfg"], d["trig"]):
        got = [utils.check_update_at_this_iter(it, int(bsz), int(iv), 0) for it in range(1, 260)]
        assert got == row.tolist()
    x = torch.from_numpy(d["x"])
    assert np.allclose(utils.inverse_sigmoid(x).numpy(), d["inv_sigmoid"], atol=1e-6)
