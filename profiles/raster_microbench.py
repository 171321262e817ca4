"""Solo timings of the two tile kernels on one camera of the bench scene, for A/B runs of library builds:
    CLMGS_LIB_PATH=clm_gs_amd/libclmgs_hip_X.so python profiles/raster_microbench.py [slab|heavy] [reps]
One forward of camera 0 (fused.camera_forward, exact sizes), then `reps` back-to-back launches of
clmgs_rasterize_fwd and of clmgs_rasterize_bwd (slot mode, the engine's form) on the camera's own lists, event-timed
on the launch stream; the partial-line table of the last backward is checksummed so that two builds can be compared."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clm_gs_amd import _lib, fused, utils  # noqa: E402
from clm_gs_amd._lib import check, dptr  # noqa: E402
from clm_gs_amd.strategies.base_engine import select_filters  # noqa: E402
from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload  # noqa: E402
from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "slab"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N, W, H = 28_000_000, 4608, 3456
args = utils.default_args(bsz=4, sh_residency="hbm")
args.clm_offload = True
utils.set_args(args)
utils.set_img_size(H, W)
sc = synth_gaussians(N, seed=0, device="cuda", kind=kind)
order = utils.morton_order(sc["xyz"])
for k in ("xyz", "scaling", "rotation", "opacity", "shs48"):
    sc[k] = utils.gather_rows(sc[k], order)
m = GaussianModelCLMOffload(3, only_for_rendering=True)
m.create_from_tensors(sc["xyz"], sc["shs48"], sc["scaling"], sc["rotation"], sc["opacity"])
m.active_sh_degree = 3
cam = nadir_cameras(4, N, W, H, 0.10, seed=0, device="cuda")[1]
with torch.no_grad():
    filters, _ = select_filters([cam], m._xyz.detach(), m._scaling.detach(), m._rotation.detach())
f = filters[0]
g = torch.Generator().manual_seed(1)
cam.original_image = (torch.rand(3, H, W, generator=g) * 255).to(torch.uint8).cuda()
p = fused.camera_forward(m, cam, f, m._parameters.data, 1, None, cam.original_image)
torch.cuda.synchronize()
L = _lib.lib()
V, I = p.V, p.fids.numel()
tw, th = (W + 15) // 16, (H + 15) // 16
st = _lib.stream()
part = torch.empty((max(I, 1), 16), device="cuda")


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out, al, last = torch.empty_like(p.out), torch.empty_like(p.alphas), torch.empty_like(p.last_ids)
fwd = timed(lambda: check(L.clmgs_rasterize_fwd(st, 1, V, I, None, None, None, None, None, W, H, 16, tw, th, dptr(p.offsets),
                                               dptr(p.fids), dptr(p.packed), dptr(out), dptr(al), dptr(last))))
assert torch.equal(out, p.out) and torch.equal(last, p.last_ids)
bwd = timed(lambda: check(L.clmgs_rasterize_bwd(st, 1, V, I, dptr(p.packed), None, W, H, 16, tw, th, dptr(p.offsets), dptr(p.fids),
                                               dptr(p.alphas), dptr(p.last_ids), dptr(p.v_out), None, None, None, None, None,
                                               None, dptr(p.emit_slot), dptr(p.row_cum), dptr(part))))
chk = part[:I].double()
print(json.dumps({"lib": os.path.basename(_lib.LIB_PATH), "scene": kind, "V": V, "I_emitted": I, "reps": reps,
                  "rasterize_fwd_ms": round(fwd, 4), "rasterize_bwd_ms": round(bwd, 4),
                  "partials_sum": float(chk.sum()), "partials_abs_sum": float(chk.abs().sum()),
                  "image_sum": float(out.double().sum())}))
