"""Generates tests/golden/*.npz by IMPORTING the reference's own pure-Python utilities from
/root/reference (build container only; the reference never travels to the GPU box).

    python tests/golden/make_golden.py

Fixtures are data only: seeded inputs and the outputs the reference code produced for them.
Sources: utils/sh_utils.py (eval_sh, RGB2SH, SH2RGB), utils/loss_utils.py (ssim, l1_loss),
utils/image_utils.py (psnr), utils/general_utils.py (build_rotation, build_scaling_rotation,
strip_symmetric, get_expon_lr_func, check_update_at_this_iter, inverse_sigmoid),
utils/graphics_utils.py (getWorld2View2, getProjectionMatrix, fov2focal, focal2fov).
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from utils import general_utils as gu  # noqa: E402
from utils import graphics_utils as gr  # noqa: E402
from utils import image_utils as iu  # noqa: E402
from utils import loss_utils as lu  # noqa: E402
from utils import sh_utils as su  # noqa: E402


def main():
    g = torch.Generator().manual_seed(20250926)
    # --- spherical harmonics: reference layout sh[..., C, K], unit dirs
    n = 257
    dirs = torch.randn(n, 3, generator=g)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    sh = torch.randn(n, 3, 16, generator=g)
    sh_out = {f"deg{d}": su.eval_sh(d, sh, dirs).numpy() for d in range(4)}
    rgb = torch.rand(n, 3, generator=g)
    np.savez(os.path.join(OUT, "sh.npz"), dirs=dirs.numpy(), sh_CK=sh.numpy(), rgb=rgb.numpy(),
             rgb2sh=su.RGB2SH(rgb).numpy(), sh2rgb=su.SH2RGB(rgb).numpy(), **sh_out)
    # --- SSIM / L1 / PSNR
    a = torch.rand(1, 3, 37, 53, generator=g)
    b = (a + 0.1 * torch.randn(1, 3, 37, 53, generator=g)).clamp(0, 1)
    a_req = a.clone().requires_grad_()
    s = lu.ssim(a_req, b)
    s.backward()
    np.savez(os.path.join(OUT, "loss.npz"), img1=a.numpy(), img2=b.numpy(), ssim=s.item(),
             ssim_grad=a_req.grad.numpy(), l1=lu.l1_loss(a, b).item(), psnr=iu.psnr(a, b).numpy(),
             window=lu.create_window(11, 3).numpy())
    # --- rotations / covariance
    q = torch.randn(64, 4, generator=g)
    sc = torch.exp(torch.randn(64, 3, generator=g) * 0.3)
    R = gu.build_rotation(q)
    L = gu.build_scaling_rotation(sc, q)
    cov6 = gu.strip_symmetric(L @ L.transpose(1, 2))
    np.savez(os.path.join(OUT, "rotation.npz"), q=q.numpy(), s=sc.numpy(), R=R.numpy(), L=L.numpy(), cov6=cov6.numpy())
    # --- cameras
    Rm = gu.build_rotation(torch.randn(1, 4, generator=g))[0].numpy().astype(np.float64)
    T = np.array([0.3, -1.2, 4.0])
    w2v = gr.getWorld2View2(Rm, T, np.array([0.1, 0.2, -0.3]), 1.5)
    proj = gr.getProjectionMatrix(0.01, 100.0, 0.9, 0.7).numpy()
    np.savez(os.path.join(OUT, "camera.npz"), R=Rm, T=T, w2v=w2v, proj=proj,
             fov2focal=gr.fov2focal(0.9, 1237), focal2fov=gr.focal2fov(1100.0, 822))
    # --- schedules
    f = gu.get_expon_lr_func(0.00016 * 2, 0.0000016 * 2, lr_delay_mult=0.01, max_steps=30000)
    steps = np.array([0, 1, 4, 100, 999, 15000, 30000, 50000])
    lr = np.array([f(int(s_)) for s_ in steps])
    trig = np.array([[gu.check_update_at_this_iter(it, bsz, iv, 0) for it in range(1, 260)]
                     for bsz, iv in ((1, 100), (4, 100), (4, 7), (64, 100))])
    x = torch.linspace(0.01, 0.99, 33)
    np.savez(os.path.join(OUT, "schedule.npz"), steps=steps, lr=lr, trig=trig, trig_cfg=np.array([[1, 100], [4, 100], [4, 7], [64, 100]]),
             x=x.numpy(), inv_sigmoid=gu.inverse_sigmoid(x).numpy())
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
