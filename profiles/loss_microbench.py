"""Solo timing of the two loss kernels at a given image size, planar [3,H,W] and interleaved [H,W,3] layouts
(python profiles/loss_microbench.py [W H]); CLMGS_LIB_PATH selects a library build."""
import sys
import torch
sys.path.insert(0, ".")
from clm_gs_amd import _lib
from clm_gs_amd._lib import check, dptr, stream
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4608, 3456)
L = _lib.lib()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
base = torch.rand((3, H, W), device=dev, generator=g)
gt = (torch.rand((3, H, W), device=dev, generator=g) * 255).to(torch.uint8)
maps = torch.empty((3, 3, H, W), device=dev)
one = torch.ones(1, device=dev)
res = {}
for name in ("planar", "interleaved"):
    if name == "planar":
        img = base.clone(); sc, sy, sx = H * W, W, 1
    else:
        img = base.permute(1, 2, 0).contiguous(); sc, sy, sx = 1, 3 * W, 3
    v_img = torch.empty_like(img)
    part = torch.zeros((L.clmgs_loss_slots(), 2), device=dev)
    def fwd():
        check(L.clmgs_l1_ssim_loss_fwd(stream(), H, W, dptr(img), sc, sy, sx, dptr(gt, torch.uint8), dptr(part),
                                       dptr(maps[0]), dptr(maps[1]), dptr(maps[2])))
    def bwd():
        check(L.clmgs_l1_ssim_loss_bwd(stream(), H, W, dptr(img), sc, sy, sx, dptr(gt, torch.uint8), dptr(one), 0.2,
                                       dptr(maps[0]), dptr(maps[1]), dptr(maps[2]), dptr(v_img)))
    for f, nm in ((fwd, "fwd"), (bwd, "bwd")):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            f()
        b.record()
        torch.cuda.synchronize()
        res[f"{name}_{nm}_ms"] = round(a.elapsed_time(b) / 20, 4)
    res[f"{name}_loss"] = (part.sum(0) / (23.0 * 3 * H * W)).tolist()
    res[f"{name}_vsum"] = float(v_img.double().abs().sum())
print(res)
