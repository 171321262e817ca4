"""2 ranks on one GPU (gloo): which rows' first-step update differs from the solo bsz-8 run?"""
import os, sys, json, torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dp_worker as D
from clm_gs_amd import utils, dp
from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
def setup(bsz):
    args = utils.default_args(bsz=bsz, sh_residency="hbm"); args.clm_offload = True
    utils.set_args(args); utils.set_img_size(D.H, D.W)
    sc = synth_gaussians(D.N, seed=0, device="cuda")
    cams = nadir_cameras(8, D.N, D.W, D.H, 0.35, seed=0, device="cuda")
    g = torch.Generator().manual_seed(5)
    for c in cams:
        c.original_image = (torch.rand(3, D.H, D.W, generator=g) * 255).to(torch.uint8).cuda()
    return args, sc, cams
rec = {}
orig = dp.allreduce_small_grads
def spy(grads):
    rec["pre"] = [g.clone() for g in grads]
    orig(grads)
    rec["post"] = [g.clone() for g in grads]
dp.allreduce_small_grads = spy
import clm_gs_amd.strategies.clm_offload.engine as E
args, sc, cams = setup(4)
m = D._model(sc, args)
p0 = m._opacity.detach().clone()
mine = D._train(m, [cams[rank::world]], args)
pre_other = [g.clone() for g in rec["pre"]]
for g in pre_other:
    dist.broadcast(g, src=1)
dist.barrier(); dist.destroy_process_group()
if rank == 0:
    # solo
    grads = {}
    og = E._gpu_adam_step
    def spy2(gaussians, a, vm):
        grads["g"] = [gaussians._xyz.grad.clone(), gaussians._opacity.grad.clone(), gaussians._scaling.grad.clone(), gaussians._rotation.grad.clone()]
        og(gaussians, a, vm)
    E._gpu_adam_step = spy2
    args1, sc1, cams1 = setup(8)
    m1 = D._model(sc1, args1)
    solo = D._train(m1, [cams1], args1)
    names = ["xyz", "opacity", "scaling", "rotation"]
    for i, n in enumerate(names):
        s = grads["g"][i]            # sum over 8 cams
        d_post = rec["post"][i] * 2  # mean over ranks -> sum
        d_sum = rec["pre"][i] + pre_other[i]
        print(n, "solo sum norm", float(s.norm()), "dp(post*2) rel", float((d_post - s).norm() / s.norm()),
              "manual r0+r1 rel", float((d_sum - s).norm() / s.norm()),
              "r0 norm", float(rec["pre"][i].norm()), "r1 norm", float(pre_other[i].norm()))
    print("dp opacity moved", float((mine[1]-p0).norm()), "solo moved", float((solo[1]-p0).norm()))
    print("param rel", [float((a - b).norm() / b.norm()) for a, b in zip(mine, solo)])
