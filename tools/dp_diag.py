import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import dp_worker as D
from clm_gs_amd import utils
from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians
torch.cuda.set_device(0)
def run(bsz, overlap=True, steps=D.STEPS, lazy=True):
    args = utils.default_args(bsz=bsz, sh_residency="hbm", overlap_cameras=overlap, lazy_dense_adam=lazy)
    args.clm_offload = True
    utils.set_args(args); utils.set_img_size(D.H, D.W)
    sc = synth_gaussians(D.N, seed=0, device="cuda")
    cams = nadir_cameras(D.STEPS * 8, D.N, D.W, D.H, 0.35, seed=0, device="cuda")
    g = torch.Generator().manual_seed(5)
    for c in cams:
        c.original_image = (torch.rand(3, D.H, D.W, generator=g) * 255).to(torch.uint8).cuda()
    batches = [cams[s * 8:(s + 1) * 8] for s in range(steps)]
    if bsz == 4:
        batches = [b[i:i+4] for b in batches for i in (0, 4)]
    return D._train(D._model(sc, args), batches, args)
def rel(a, b): return [float((x - y).norm() / y.norm()) for x, y in zip(a, b)]
a = run(8); b = run(8)
print("solo8 vs solo8", rel(a, b))
c = run(8, overlap=False)
print("solo8 vs solo8 no-overlap", rel(a, c))
d = run(8, lazy=False)
print("solo8 vs solo8 eager adam", rel(a, d))
e = run(4)
print("solo8 vs 6 steps of bsz4 (expected different)", rel(a, e))
