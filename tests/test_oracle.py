"""CPU-only: the two oracles agree with each other, and the product's per-element formulas
(clm_gs_amd/csrc/gs_math.h, compiled here with plain g++ as a test shim) agree with autograd
of the torch oracle.  No GPU, no HIP runtime."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import c_oracle as C
from oracle import gs_oracle as O
from tests.scenes import psnr, rel_l2, small_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _activated(s):
    quats = torch.nn.functional.normalize(s["quats"])
    return s["means"], quats, s["scales"], s["opac"], s["shs"]


def test_c_oracle_matches_torch_oracle_forward_and_backward():
    s = small_scene(n=700, width=80, height=56, seed=21)
    w, h = s["width"], s["height"]
    means, quats, scales, opac, shs = _activated(s)
    p0 = [t.clone().double().requires_grad_() for t in (means, opac, scales, quats, shs)]
    img0, m2, radii, aux = O.render_one_camera(*p0, 3, s["viewmat"].double(), s["K"].double(), w, h)
    l0 = O.training_loss(img0, s["gt"])
    l0.backward()
    fw = C.render_forward(means.numpy(), quats.numpy(), scales.numpy(), opac.numpy(),
                          shs.reshape(-1, 48).numpy(), 3, s["viewmat"].numpy(), s["K"].numpy(), w, h)
    assert np.array_equal(fw["radii"], radii[0].numpy())
    # depth bits come from fp32 here and from rounded fp64 there: tile part exact, order ~exact
    assert np.array_equal(fw["isect_ids"] >> 32, (aux["isect_ids"] >> 32).numpy())
    assert (fw["flatten_ids"] == aux["flatten_ids"].numpy()).mean() > 0.999
    assert np.array_equal(fw["offsets"], aux["offsets"].flatten().numpy())
    img_c = torch.from_numpy(fw["image_hwc"]).permute(2, 0, 1)
    assert psnr(img_c, img0) > 60
    bw = C.loss_and_backward(fw, s["gt"].numpy())
    assert abs(bw["loss"] - l0.item()) < 1e-5
    # gradients wrt ACTIVATED inputs: compare through the same activations
    assert rel_l2(torch.from_numpy(bw["v_means"]), p0[0].grad) < 3e-3  # v_dirs part is finite-difference
    assert rel_l2(torch.from_numpy(bw["v_opac"]), p0[1].grad.flatten()) < 2e-4
    assert rel_l2(torch.from_numpy(bw["v_scales"]), p0[2].grad) < 2e-4
    assert rel_l2(torch.from_numpy(bw["v_quats"]), p0[3].grad) < 2e-4
    assert rel_l2(torch.from_numpy(bw["v_shs48"]), p0[4].grad.reshape(-1, 48)) < 2e-4


def test_c_oracle_background_and_ragged_size():
    s = small_scene(n=300, width=37, height=21, seed=22)
    w, h = s["width"], s["height"]
    means, quats, scales, opac, shs = _activated(s)
    bg = torch.tensor([0.1, 0.6, 0.3])
    img0, _, _, _ = O.render_one_camera(means, opac, scales, quats, shs, 2, s["viewmat"], s["K"], w, h, background=bg)
    fw = C.render_forward(means.numpy(), quats.numpy(), scales.numpy(), opac.numpy(),
                          shs.reshape(-1, 48).numpy(), 2, s["viewmat"].numpy(), s["K"].numpy(), w, h,
                          background=bg.numpy())
    assert psnr(torch.from_numpy(fw["image_hwc"]).permute(2, 0, 1), img0) > 60


def test_empty_scene_renders_background():
    n = 5
    means = torch.zeros(n, 3); means[:, 2] = -5.0  # behind the camera: all culled
    quats = torch.zeros(n, 4); quats[:, 0] = 1
    scales = torch.ones(n, 3) * 0.1
    opac = torch.ones(n, 1) * 0.5
    shs = torch.zeros(n, 16, 3)
    vm, K = torch.eye(4), torch.tensor([[30.0, 0, 16], [0, 30.0, 8], [0, 0, 1]])
    bg = torch.tensor([0.2, 0.4, 0.6])
    img, _, radii, aux = O.render_one_camera(means, opac, scales, quats, shs, 0, vm, K, 32, 16, background=bg)
    assert radii.sum() == 0 and aux["flatten_ids"].numel() == 0
    assert torch.allclose(img, bg[:, None, None].expand(3, 16, 32))
    fw = C.render_forward(means.numpy(), quats.numpy(), scales.numpy(), opac.numpy(),
                          shs.reshape(-1, 48).numpy(), 0, vm.numpy(), K.numpy(), 32, 16, background=bg.numpy())
    assert fw["n_isects"] == 0
    assert np.allclose(fw["image_hwc"], bg.numpy()[None, None, :])


@pytest.fixture(scope="module")
def shim():
    out = os.path.join("/tmp", f"gs_math_shim_{os.getpid()}.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "clm_gs_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_shim", "gs_math_shim.cpp"), "-o", out])
    return ctypes.CDLL(out)


def _P(t):
    return ctypes.c_void_p(t.data_ptr())


def test_product_projection_formulas_vs_autograd(shim):
    s = small_scene(n=3000, width=200, height=120, seed=23, spread=3.0, depth=5.0)
    N, W, H = 3000, 200.0, 120.0
    means, quats, scales = s["means"].contiguous(), s["quats"].contiguous(), s["scales"].contiguous()
    vm, K = s["viewmat"].contiguous(), s["K"].clone()
    K[0, 2] += 3.0; K[1, 2] -= 2.0  # off-centre principal point
    f = ctypes.c_float
    radii = torch.zeros(N, dtype=torch.int32); m2 = torch.zeros(N, 2); d = torch.zeros(N); cn = torch.zeros(N, 3)
    shim.shim_project_fwd(N, _P(means), _P(quats), _P(scales), _P(vm), _P(K), f(W), f(H), f(0.3), f(0.01),
                          f(1e10), f(0.0), _P(radii), _P(m2), _P(d), _P(cn))
    md, qd, sd = [t.double().requires_grad_() for t in (means, quats, scales)]
    r0, m0, d0, c0, _ = O.fully_fused_projection(md, None, qd, sd, vm.double()[None], K.double()[None], 200, 120)
    assert torch.equal(r0[0], radii)
    ok = radii > 0
    assert ok.sum() > 500
    assert rel_l2(cn[ok], c0[0][ok]) < 1e-5 and rel_l2(m2[ok], m0[0][ok]) < 1e-6
    g = torch.Generator().manual_seed(1)
    vm2, vd, vc = torch.randn(N, 2, generator=g), torch.randn(N, generator=g), torch.randn(N, 3, generator=g)
    ((m0[0] * vm2.double()).sum() + (d0[0] * vd.double()).sum() + (c0[0] * vc.double()).sum()).backward()
    vmn, vq, vs = torch.zeros(N, 3), torch.zeros(N, 4), torch.zeros(N, 3)
    shim.shim_project_bwd(N, _P(means), _P(quats), _P(scales), _P(vm), _P(K), f(W), f(H), f(0.3), _P(radii),
                          _P(vm2), _P(vd), _P(vc), _P(vmn), _P(vq), _P(vs))
    assert rel_l2(vmn[ok], md.grad[ok]) < 1e-5
    assert rel_l2(vq[ok], qd.grad[ok]) < 1e-5
    assert rel_l2(vs[ok], sd.grad[ok]) < 1e-5
    assert vmn[~ok].abs().max() == 0


@pytest.mark.parametrize("deg", [1, 2, 3])
def test_product_sh_basis_vs_autograd(shim, deg):
    N = 1000
    dirs = torch.randn(N, 3, generator=torch.Generator().manual_seed(deg))
    B, Bx, By, Bz = (torch.zeros(N, 16) for _ in range(4))
    shim.shim_sh_basis(N, deg, _P(dirs), _P(B), _P(Bx), _P(By), _P(Bz))
    dn = (dirs / dirs.norm(dim=-1, keepdim=True)).double().requires_grad_()
    Bo = O.sh_basis(deg, dn)
    nb = (deg + 1) ** 2
    assert (Bo.float() - B[:, :nb]).abs().max() < 1e-6
    w = torch.randn(N, nb, generator=torch.Generator().manual_seed(7)).double()
    (Bo * w).sum().backward()
    g = torch.stack([(Bx[:, :nb].double() * w).sum(1), (By[:, :nb].double() * w).sum(1), (Bz[:, :nb].double() * w).sum(1)], -1)
    g = g - (g * dn.detach()).sum(-1, keepdim=True) * dn.detach()
    assert (g - dn.grad).abs().max() < 1e-5


def test_ssim_window_constants_match_product():
    """c_win in clm_gs_amd/csrc/ssim.hip == normalised 11-tap Gaussian of utils/loss_utils.py:26-33."""
    src = open(os.path.join(ROOT, "clm_gs_amd", "csrc", "ssim.hip")).read()
    body = src.split("c_win[11] = {")[1].split("}")[0]
    vals = [float(v.strip().rstrip("f")) for v in body.split(",") if v.strip()]
    g = [math.exp(-((x - 5) ** 2) / (2 * 1.5 ** 2)) for x in range(11)]
    ref = [v / sum(g) for v in g]
    assert len(vals) == 11 and max(abs(a - b) for a, b in zip(vals, ref)) < 1e-9
