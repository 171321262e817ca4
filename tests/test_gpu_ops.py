"""-m gpu: every HIP operator (through the C ABI) against the CPU oracle.

Tolerances (fp32 path): forward images PSNR >= 60 dB and max-abs <= 2e-4;
gradients relative-L2 <= 2e-4 against autograd of the oracle run in float64
(float atomics make the summation order nondeterministic); integer outputs
(radii, tile counts, isect ids, offsets, last ids, filters) bit-exact.
"""
import math

import pytest
import torch

from oracle import gs_oracle as O
from tests.scenes import psnr, rel_l2, small_scene

pytestmark = pytest.mark.gpu

GRAD_TOL = 2e-4


def _to(dev, *ts):
    return [t.to(dev) for t in ts]


def test_library_is_native(dev):
    from clm_gs_amd import _lib
    assert _lib.lib().clmgs_version() >= 100


@pytest.mark.parametrize("C", [1, 3])
def test_projection_fwd_bwd(dev, C):
    from clm_gs_amd import gsplat as G
    s = small_scene(n=6000, width=200, height=120, seed=1, spread=3.0, depth=5.0)
    vms = torch.stack([s["viewmat"]] * C)
    for c in range(C):
        vms[c, 0, 3] += 0.4 * c
    Ks = torch.stack([s["K"]] * C)
    m, q, sc = [t.clone().double().requires_grad_() for t in (s["means"], s["quats"], s["scales"])]
    r0, m0, d0, c0, _ = O.fully_fused_projection(m, None, q, sc, vms.double(), Ks.double(), 200, 120)
    md, qd, sd = [t.clone().to(dev).requires_grad_() for t in (s["means"], s["quats"], s["scales"])]
    r1, m1, d1, c1, _ = G.fully_fused_projection(md, None, qd, sd, vms.to(dev), Ks.to(dev), 200, 120)
    assert torch.equal(r1.cpu(), r0), "radii must be bit-exact"
    assert (r0 > 0).sum() > 1000
    ok = r0 > 0
    assert ((m1.cpu() - m0.float()).abs() / (1 + m0.float().abs()))[ok].max() < 1e-5
    assert rel_l2(d1.cpu()[ok], d0[ok]) < 1e-6
    assert rel_l2(c1.cpu()[ok], c0[ok]) < 1e-5
    assert m1.cpu()[~ok].abs().max() == 0 and c1.cpu()[~ok].abs().max() == 0
    g = torch.Generator().manual_seed(5)
    vm, vd, vc = torch.randn(m0.shape, generator=g), torch.randn(d0.shape, generator=g), torch.randn(c0.shape, generator=g)
    ((m0 * vm.double()).sum() + (d0 * vd.double()).sum() + (c0 * vc.double()).sum()).backward()
    ((m1 * vm.to(dev)).sum() + (d1 * vd.to(dev)).sum() + (c1 * vc.to(dev)).sum()).backward()
    assert rel_l2(md.grad.cpu(), m.grad) < GRAD_TOL
    assert rel_l2(qd.grad.cpu(), q.grad) < GRAD_TOL
    assert rel_l2(sd.grad.cpu(), sc.grad) < GRAD_TOL


def test_projection_packed_matches_oracle(dev):
    from clm_gs_amd import gsplat as G
    s = small_scene(n=3000, width=96, height=64, seed=2, spread=4.0)
    vms = torch.stack([s["viewmat"], s["viewmat"]])
    vms[1, 0, 3] -= 1.0
    Ks = torch.stack([s["K"]] * 2)
    ref = O.fully_fused_projection(s["means"], None, s["quats"], s["scales"], vms, Ks, 96, 64, packed=True)
    out = G.fully_fused_projection(*_to(dev, s["means"]), None, *_to(dev, s["quats"], s["scales"], vms, Ks), 96, 64, packed=True)
    assert torch.equal(out[0].cpu(), ref[0]) and torch.equal(out[1].cpu(), ref[1])
    assert torch.equal(out[2].cpu(), ref[2])
    rad = G.visibility_radii(*_to(dev, s["means"], s["quats"], s["scales"], vms, Ks), 96, 64)
    cam, gid = torch.nonzero(rad > 0, as_tuple=True)
    assert torch.equal(cam.cpu(), ref[0]) and torch.equal(gid.cpu(), ref[1])


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
@pytest.mark.parametrize("masked", [False, True])
def test_sh_fwd_bwd(dev, deg, masked):
    from clm_gs_amd import gsplat as G
    g = torch.Generator().manual_seed(10 + deg)
    n = 1337
    dirs = torch.randn(1, n, 3, generator=g) * 3
    coeffs = torch.randn(1, n, 16, 3, generator=g)
    masks = (torch.rand(1, n, generator=g) > 0.3) if masked else None
    d0, c0 = dirs.double().requires_grad_(), coeffs.double().requires_grad_()
    col0 = O.spherical_harmonics(deg, d0, c0, masks)
    d1, c1 = dirs.to(dev).requires_grad_(), coeffs.to(dev).requires_grad_()
    col1 = G.spherical_harmonics(deg, d1, c1, masks.to(dev) if masked else None)
    assert (col1.cpu() - col0.float()).abs().max() < 1e-5
    v = torch.randn(col0.shape, generator=g)
    (col0 * v.double()).sum().backward()
    (col1 * v.to(dev)).sum().backward()
    assert rel_l2(c1.grad.cpu(), c0.grad) < 1e-5
    if deg > 0:
        assert rel_l2(d1.grad.cpu(), d0.grad) < 1e-4
    else:
        assert d1.grad.abs().max() == 0


def test_sh_bwd_inplace_accumulates(dev):
    from clm_gs_amd import clm_kernels as K
    g = torch.Generator().manual_seed(3)
    n, deg = 700, 2
    dirs = torch.randn(1, n, 3, generator=g)
    coeffs = torch.randn(1, n, 16, 3, generator=g)
    vcol = torch.randn(1, n, 3, generator=g)
    prior = torch.randn(n, 48, generator=g)
    d0, c0 = dirs.double().requires_grad_(), coeffs.double().requires_grad_()
    (O.spherical_harmonics(deg, d0, c0) * vcol.double()).sum().backward()
    buf = prior.clone().to(dev)
    v_dirs = K.spherical_harmonics_bwd_inplace(deg, dirs.to(dev), coeffs.to(dev), buf, vcol.to(dev))
    want = prior.double() + c0.grad.reshape(n, 48)
    assert rel_l2(buf.cpu(), want) < 1e-5
    assert rel_l2(v_dirs.cpu(), d0.grad) < 1e-4


def _project_cpu(s):
    return O.fully_fused_projection(s["means"], None, s["quats"], s["scales"], s["viewmat"][None], s["K"][None], s["width"], s["height"])


@pytest.mark.parametrize("wh", [(64, 48), (70, 37), (16, 16)])
def test_isect_bit_exact(dev, wh):
    from clm_gs_amd import gsplat as G
    w, h = wh
    s = small_scene(n=500, width=w, height=h, seed=4)
    radii, m2, d, cn, _ = _project_cpu(s)
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    t0, i0, f0 = O.isect_tiles(m2, radii, d, 16, tw, th)
    t1, i1, f1 = G.isect_tiles(m2.to(dev), radii.to(dev), d.to(dev), 16, tw, th)
    assert torch.equal(t1.cpu(), t0)
    assert torch.equal(i1.cpu(), i0)
    assert torch.equal(f1.cpu(), f0)
    o0 = O.isect_offset_encode(i0, 1, tw, th)
    o1 = G.isect_offset_encode(i1, 1, tw, th)
    assert torch.equal(o1.cpu(), o0)


def test_isect_empty(dev):
    from clm_gs_amd import gsplat as G
    n = 10
    radii = torch.zeros(1, n, dtype=torch.int32, device=dev)
    m2 = torch.zeros(1, n, 2, device=dev)
    d = torch.zeros(1, n, device=dev)
    t, i, f = G.isect_tiles(m2, radii, d, 16, 4, 3)
    assert i.numel() == 0 and f.numel() == 0 and t.sum() == 0
    off = G.isect_offset_encode(i, 1, 4, 3)
    assert off.shape == (1, 3, 4) and off.abs().sum() == 0


@pytest.mark.parametrize("bg", [None, (0.2, 0.5, 0.9)])
@pytest.mark.parametrize("wh", [(64, 48), (70, 37)])
def test_rasterize_fwd_bwd(dev, bg, wh):
    from clm_gs_amd import gsplat as G
    w, h = wh
    s = small_scene(n=600, width=w, height=h, seed=6, log_scale=-1.2)
    radii, m2, d, cn, _ = _project_cpu(s)
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    _, ids, fids = O.isect_tiles(m2, radii, d, 16, tw, th)
    off = O.isect_offset_encode(ids, 1, tw, th)
    g = torch.Generator().manual_seed(9)
    colors = torch.rand(1, 600, 3, generator=g)
    opac = s["opac"].reshape(1, -1)
    bgt = torch.tensor([bg]) if bg is not None else None
    a = [t.clone().double().requires_grad_() for t in (m2, cn, colors, opac)]
    img0, al0, last0 = O.rasterize_to_pixels(*a, w, h, 16, off, fids, backgrounds=bgt.double() if bg else None, return_last_ids=True)
    b = [t.clone().to(dev).requires_grad_() for t in (m2, cn, colors, opac)]
    img1, al1 = G.rasterize_to_pixels(*b, w, h, 16, off.to(dev), fids.to(dev), backgrounds=bgt.to(dev) if bg else None)
    assert img1.shape == (1, h, w, 3) and al1.shape == (1, h, w, 1)
    assert psnr(img1.cpu(), img0) > 60
    assert (img1.cpu() - img0.float()).abs().max() < 2e-4
    assert (al1.cpu() - al0.float()).abs().max() < 2e-4
    vi, va = torch.randn(img0.shape, generator=g), torch.randn(al0.shape, generator=g)
    ((img0 * vi.double()).sum() + (al0 * va.double()).sum()).backward()
    ((img1 * vi.to(dev)).sum() + (al1 * va.to(dev)).sum()).backward()
    for name, x, y in zip(("means2d", "conics", "colors", "opacities"), b, a):
        assert rel_l2(x.grad.cpu(), y.grad) < GRAD_TOL, name


def test_rasterize_last_ids_and_saturation(dev):
    """Opaque stack: exercises the T <= 1e-4 early stop and last_ids."""
    from clm_gs_amd import _lib, gsplat as G
    w, h = 48, 32
    s = small_scene(n=900, width=w, height=h, seed=7, spread=0.6, log_scale=-0.6)
    s["opac"] = torch.full_like(s["opac"], 0.97)
    radii, m2, d, cn, _ = _project_cpu(s)
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    _, ids, fids = O.isect_tiles(m2, radii, d, 16, tw, th)
    off = O.isect_offset_encode(ids, 1, tw, th)
    colors = torch.rand(1, 900, 3, generator=torch.Generator().manual_seed(1))
    opac = s["opac"].reshape(1, -1)
    img0, al0, last0 = O.rasterize_to_pixels(m2, cn, colors, opac, w, h, 16, off, fids, return_last_ids=True)
    assert (al0 > 0.9998).float().mean() > 0.3, "scene must saturate many pixels"
    # call the C ABI directly to also see last_ids
    L = _lib.lib()
    from clm_gs_amd._lib import dptr, stream, check
    t = [x.to(dev).contiguous() for x in (m2, cn, colors, opac, off, fids)]
    out = torch.empty(1, h, w, 3, device=dev); al = torch.empty(1, h, w, device=dev)
    last = torch.empty(1, h, w, dtype=torch.int32, device=dev)
    packed = torch.empty(900, 16, device=dev)
    assert L.clmgs_rasterize_pack_bytes(1, 900) == packed.numel() * 4
    check(L.clmgs_rasterize_fwd(stream(), 1, 900, fids.numel(), dptr(t[0]), dptr(t[1]), dptr(t[2]), dptr(t[3]), None,
                                w, h, 16, tw, th, dptr(t[4]), dptr(t[5]), dptr(packed), dptr(out), dptr(al), dptr(last)))
    assert psnr(out.cpu(), img0) > 60
    mism = (last.cpu() != last0).float().mean().item()
    assert mism < 0.01, f"last_ids mismatch fraction {mism} (threshold ties only)"


@pytest.mark.parametrize("hw", [(48, 64), (37, 70), (11, 9)])
def test_fused_ssim(dev, hw):
    from clm_gs_amd import clm_kernels as K
    h, w = hw
    g = torch.Generator().manual_seed(11)
    a, b = torch.rand(1, 3, h, w, generator=g), torch.rand(1, 3, h, w, generator=g)
    a0 = a.double().requires_grad_()
    s0 = O.fused_ssim(a0, b.double())
    a1 = a.to(dev).requires_grad_()
    s1 = K.fused_ssim(a1, b.to(dev))
    assert abs(s1.item() - s0.item()) < 1e-5
    (s0 * -0.2).backward()
    (s1 * -0.2).backward()
    assert rel_l2(a1.grad.cpu(), a0.grad) < 1e-4


def test_end_to_end_one_camera(dev):
    """strategies/no_offload/engine.py:15-101 composition: loss and all input grads."""
    from clm_gs_amd import clm_kernels as K, gsplat as G
    s = small_scene(n=800, width=80, height=56, seed=12)
    w, h = s["width"], s["height"]
    p0 = [t.clone().double().requires_grad_() for t in (s["means"], s["opac"], s["scales"], s["quats"], s["shs"])]
    img0, _, _, _ = O.render_one_camera(*p0, 3, s["viewmat"].double(), s["K"].double(), w, h)
    l0 = O.training_loss(img0, s["gt"])
    l0.backward()
    p1 = [t.clone().to(dev).requires_grad_() for t in (s["means"], s["opac"], s["scales"], s["quats"], s["shs"])]
    means, opac, scales, quats, shs = p1
    vm, Kd = s["viewmat"].to(dev), s["K"].to(dev)
    radii, m2, d, cn, _ = G.fully_fused_projection(means, None, quats, scales, vm[None], Kd[None], w, h)
    dirs = means[None] - torch.inverse(vm[None])[:, None, :3, 3]
    col = torch.clamp_min(G.spherical_harmonics(3, dirs, shs[None], masks=radii > 0) + 0.5, 0.0)
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    _, ids, fids = G.isect_tiles(m2, radii, d, 16, tw, th)
    off = G.isect_offset_encode(ids, 1, tw, th)
    img, _ = G.rasterize_to_pixels(m2, cn, col, opac.squeeze(1)[None], w, h, 16, off, fids)
    img1 = img[0].permute(2, 0, 1).contiguous()
    assert psnr(img1.cpu(), img0) > 60
    gt = torch.clamp(s["gt"].to(dev).float() / 255.0, 0, 1)
    l1 = 0.8 * (img1 - gt).abs().mean() + 0.2 * (1 - K.fused_ssim(img1[None], gt[None]))
    l1.backward()
    assert abs(l1.item() - l0.item()) < 1e-5
    for name, x, y in zip(("means", "opac", "scales", "quats", "shs"), p1, p0):
        assert rel_l2(x.grad.cpu(), y.grad) < 5e-4, name


def test_row_movers_and_bitmaps(dev):
    from clm_gs_amd import clm_kernels as K
    g = torch.Generator().manual_seed(13)
    N, n = 5000, 1200
    params = torch.randn(N, 48, generator=g)
    filt = torch.randperm(N, generator=g)[:n].sort().values
    shs = torch.empty(n, 48, device=dev)
    K.send_shs2gpu_stream(shs, params.to(dev), filt.to(dev))
    assert torch.equal(shs.cpu(), params[filt])
    # retention: half from "host", half from previous buffer
    nxt = torch.full((n, 48), float("nan"), device=dev)
    pos = torch.randperm(n, generator=g)
    hpos, dpos = pos[: n // 2], pos[n // 2:]
    hidx = torch.randint(0, N, (n // 2,), generator=g).to(torch.int32)
    didx = torch.randint(0, n, (n - n // 2,), generator=g).to(torch.int32)
    K.send_shs2gpu_stream_retention(nxt, params.to(dev), shs, hidx.to(dev), didx.to(dev),
                                    hpos.to(torch.int32).to(dev), dpos.to(torch.int32).to(dev))
    want = torch.empty(n, 48)
    want[hpos] = params[hidx.long()]
    want[dpos] = params[filt][didx.long()]
    assert torch.equal(nxt.cpu(), want)
    # gradient scatter-add
    gb = torch.randn(N, 48, generator=g)
    grad = torch.randn(n, 48, generator=g)
    gbd = gb.clone().to(dev)
    K.send_shs2cpu_grad_buffer_stream(grad.to(dev), gbd, filt.to(dev), True)
    want = gb.clone(); want[filt] += grad
    assert torch.allclose(gbd.cpu(), want, atol=1e-6)
    # bitmap / ffs / overlap
    bsz = 4
    filters = [torch.randperm(N, generator=g)[: 800 + 100 * i].sort().values for i in range(bsz)]
    bm = torch.zeros(N, dtype=torch.int8, device=dev)
    for i, f in enumerate(filters):
        K.scatter_to_bit(bm, f.to(dev), bsz - 1 - i)
    ref = torch.zeros(N, dtype=torch.int64)
    for i, f in enumerate(filters):
        ref[f] |= 1 << (bsz - 1 - i)
    assert torch.equal(bm.cpu().to(torch.int64) & 0xFF, ref)
    ffs = torch.empty(N, dtype=torch.uint8, device=dev)
    K.extract_ffs(bm, ffs)
    want_ffs = torch.tensor([(int(v) & -int(v)).bit_length() for v in ref.tolist()], dtype=torch.uint8)
    assert torch.equal(ffs.cpu(), want_ffs)
    cnt = K.pair_overlap_count(bm, bsz)
    sets = [set(f.tolist()) for f in filters]
    assert cnt.cpu().tolist() == [len(sets[i] & sets[i + 1]) for i in range(bsz - 1)]


def test_pinned_zero_copy_and_signal(dev):
    import ctypes, numpy as np
    from clm_gs_amd import clm_kernels as K, host as Hm
    N, n = 3000, 500
    buf = Hm.pinned_empty((N, 48))
    g = torch.Generator().manual_seed(14)
    buf.copy_(torch.randn(N, 48, generator=g))
    filt = torch.randperm(N, generator=g)[:n].sort().values
    shs = torch.empty(n, 48, device=dev)
    K.send_shs2gpu_stream(shs, buf, filt.to(dev), 32, 256)
    torch.cuda.synchronize()
    assert torch.equal(shs.cpu(), buf[filt])
    gb = Hm.pinned_empty((N, 48)); gb.zero_()
    grad = torch.randn(n, 48, generator=g)
    K.send_shs2cpu_grad_buffer_stream(grad.to(dev), gb, filt.to(dev), True, 32, 256)
    sig = Hm.pinned_empty((4,), dtype=torch.int32); sig.zero_()
    K.set_signal(sig, 2, 1)
    torch.cuda.synchronize()
    assert sig.tolist() == [0, 0, 1, 0]
    want = torch.zeros(N, 48); want[filt] = grad
    assert torch.equal(gb.clone(), want)


def test_adam_rows_and_selective(dev):
    from clm_gs_amd import clm_kernels as K
    g = torch.Generator().manual_seed(15)
    N, cols = 4000, 48
    p, gr = torch.randn(N, cols, generator=g), torch.randn(N, cols, generator=g)
    m, v = torch.rand(N, cols, generator=g) * 0.1, torch.rand(N, cols, generator=g) * 0.01
    rows = torch.randperm(N, generator=g)[:900].to(torch.int32)
    col_lr = torch.cat([torch.full((3,), 2.5e-3), torch.full((45,), 1.25e-4)])
    ref = [t.clone().double() for t in (p, gr, m, v)]
    O.adam_rows(*ref, rows, col_lr.double(), 0.9 ** 4, 0.999 ** 4, 1e-15 / 2, step=7, scale=0.25, zero_grad=True)
    d = [t.clone().to(dev) for t in (p, gr, m, v)]
    K.adam_rows(*d, rows.to(dev), col_lr.to(dev), 0.9 ** 4, 0.999 ** 4, 1e-15 / 2, 7, True, 0.25, True)
    for a, b in zip(d, ref):
        assert rel_l2(a.cpu(), b) < 1e-6
    # selective (masked, no bias correction) on a [N,3] tensor
    p3, g3 = torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g)
    m3, v3 = torch.zeros(N, 3), torch.zeros(N, 3)
    vis = torch.rand(N, generator=g) > 0.5
    ref = [t.clone().double() for t in (p3, g3, m3, v3)]
    O.selective_adam(*ref, vis, 1e-3, 0.9, 0.999, 1e-15)
    d = [t.clone().to(dev) for t in (p3, g3, m3, v3)]
    K.selective_adam_update(*d, vis.to(dev), 1e-3, 0.9, 0.999, 1e-15, N, 3)
    for a, b in zip(d, ref):
        assert rel_l2(a.cpu(), b) < 1e-6


def test_densify_stats(dev):
    from clm_gs_amd import clm_kernels as K
    g = torch.Generator().manual_seed(16)
    N, n = 2000, 600
    filt = torch.randperm(N, generator=g)[:n].sort().values
    vm = torch.randn(n, 2, generator=g)
    radii = torch.randint(0, 30, (n,), generator=g).to(torch.int32)
    mr, acc, den = torch.rand(N, generator=g) * 10, torch.rand(N, 1, generator=g), torch.rand(N, 1, generator=g)
    d = [t.clone().to(dev) for t in (mr, acc, den)]
    K.densify_stats(filt.to(dev), vm.to(dev), radii.to(dev), 640, 480, *d, only_visible=True)
    vis = radii > 0
    f = filt[vis]
    mr2, acc2, den2 = mr.clone(), acc.clone(), den.clone()
    mr2[f] = torch.maximum(mr2[f], radii[vis].float())
    gg = vm[vis] * torch.tensor([320.0, 240.0])
    acc2[f] += gg.norm(dim=-1, keepdim=True)
    den2[f] += 1
    assert torch.allclose(d[0].cpu(), mr2) and torch.allclose(d[1].cpu(), acc2, atol=1e-5) and torch.allclose(d[2].cpu(), den2)


@pytest.mark.parametrize("hw", [(48, 64), (37, 70)])
@pytest.mark.parametrize("layout", ["chw", "hwc_view"])
def test_fused_l1_ssim_loss(dev, hw, layout):
    """strategies/base_engine.py:79-103 in one kernel each way, reading the image through strides."""
    from clm_gs_amd import clm_kernels as K
    h, w = hw
    g = torch.Generator().manual_seed(17)
    img = torch.rand(3, h, w, generator=g)
    gt = (torch.rand(3, h, w, generator=g) * 255).to(torch.uint8)
    a0 = img.double().requires_grad_()
    l0 = O.training_loss(a0, gt)
    l0.backward()
    if layout == "chw":
        leaf = img.to(dev).requires_grad_()
        view = leaf
    else:
        leaf = img.permute(1, 2, 0).contiguous().to(dev).requires_grad_()  # [H,W,3] memory
        view = leaf.permute(2, 0, 1)
    l1 = K.fused_l1_ssim_loss(view, gt.to(dev), 0.2)
    l1.backward()
    assert abs(l1.item() - l0.item()) < 1e-5
    got = leaf.grad.cpu() if layout == "chw" else leaf.grad.permute(2, 0, 1).cpu()
    assert rel_l2(got, a0.grad) < 1e-4


def test_rasterize_large_gaussians_and_culling_exactness(dev):
    """Gaussians spanning many tiles plus tiny ones: the quadrant culling may only drop pairs the
    reference drops (alpha < 1/255), so image and gradients must still match the oracle."""
    from clm_gs_amd import gsplat as G
    w, h = 96, 80
    s = small_scene(n=300, width=w, height=h, seed=31, log_scale=-0.3)
    s["scales"][:100] *= 0.05  # sub-pixel splats
    s["opac"][::7] = 0.004     # below 1/255: never visible
    radii, m2, d, cn, _ = _project_cpu(s)
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    _, ids, fids = O.isect_tiles(m2, radii, d, 16, tw, th)
    off = O.isect_offset_encode(ids, 1, tw, th)
    g = torch.Generator().manual_seed(2)
    colors = torch.rand(1, 300, 3, generator=g)
    opac = s["opac"].reshape(1, -1)
    a = [t.clone().double().requires_grad_() for t in (m2, cn, colors, opac)]
    img0, al0 = O.rasterize_to_pixels(*a, w, h, 16, off, fids)
    b = [t.clone().to(dev).requires_grad_() for t in (m2, cn, colors, opac)]
    img1, al1 = G.rasterize_to_pixels(*b, w, h, 16, off.to(dev), fids.to(dev))
    assert psnr(img1.cpu(), img0) > 60 and (img1.cpu() - img0.float()).abs().max() < 2e-4
    vi = torch.randn(img0.shape, generator=g)
    (img0 * vi.double()).sum().backward()
    (img1 * vi.to(dev)).sum().backward()
    for name, x, y in zip(("means2d", "conics", "colors", "opacities"), b, a):
        assert rel_l2(x.grad.cpu(), y.grad) < GRAD_TOL, name


@pytest.mark.parametrize("wh", [(64, 48), (70, 37), (160, 128)])
def test_two_level_binning_equals_64bit_sort(dev, wh):
    """Depth sort of rows + one stable tile-id sort == the single stable (tile|depth) sort, bit for bit
    (including ties: duplicated depths keep row order)."""
    from clm_gs_amd import gsplat as G
    w, h = wh
    s = small_scene(n=900, width=w, height=h, seed=41, log_scale=-1.0)
    radii, m2, d, cn, _ = _project_cpu(s)
    d = d.clone()
    d[0, 100:140] = d[0, 100]  # exact depth ties
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    _, i0, f0 = G.isect_tiles(m2.to(dev), radii.to(dev), d.to(dev), 16, tw, th)
    o0 = G.isect_offset_encode(i0, 1, tw, th)
    f1, o1, i1 = G.isect_tiles_two_level(m2.to(dev), radii.to(dev), d.to(dev), 16, tw, th, want_isect_ids=True)
    assert torch.equal(f1, f0) and torch.equal(o1, o0) and torch.equal(i1, i0)
    _, ic, fc = O.isect_tiles(m2, radii, d, 16, tw, th)
    assert torch.equal(f1.cpu(), fc) and torch.equal(i1.cpu(), ic)


def test_distCUDA2_matches_brute_force(dev):
    """simple_knn.distCUDA2: mean squared distance to the 3 nearest neighbours (row f3)."""
    from clm_gs_amd.simple_knn import distCUDA2
    g = torch.Generator().manual_seed(51)
    pts = torch.cat([torch.randn(1500, 3, generator=g), torch.rand(700, 3, generator=g) * 10 - 5,
                     torch.randn(300, 3, generator=g) * 0.01 + 3.0])  # clusters + sparse halo
    d = torch.cdist(pts.double(), pts.double()) ** 2
    d.fill_diagonal_(float("inf"))
    want = d.topk(3, dim=1, largest=False).values.mean(dim=1)
    got = distCUDA2(pts.to(dev)).cpu().double()
    assert ((got - want).abs() / (want + 1e-12)).max() < 1e-4


@pytest.mark.parametrize("opaque", [False, True])
def test_rasterize_bwd_atomic_free_path(dev, opaque):
    """clmgs_rasterize_bwd with emit slots (plain stores + per-row sum) == the float-atomic path
    == autograd of the oracle; the slot path is bitwise reproducible; slots are a consistent
    permutation.  `opaque` saturates pixels so that entries behind the deepest contributor and
    culled entries (zero lines) are exercised."""
    from clm_gs_amd import _lib, gsplat as G
    from clm_gs_amd._lib import check, dptr, stream
    L = _lib.lib()
    w, h, n = 70, 53, 1200
    s = small_scene(n=n, width=w, height=h, seed=61, spread=0.7 if opaque else 1.0, log_scale=-0.7 if opaque else -1.1)
    if opaque:
        s["opac"] = torch.full_like(s["opac"], 0.95)
    radii, m2, d, cn, _ = _project_cpu(s)
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    g = torch.Generator().manual_seed(3)
    colors = torch.rand(1, n, 3, generator=g)
    opac = s["opac"].reshape(1, -1)
    f0, o0, _ = G.isect_tiles_two_level(m2.to(dev), radii.to(dev), d.to(dev), 16, tw, th)
    fids, off, _, (slot, row_cum) = G.isect_tiles_two_level(m2.to(dev), radii.to(dev), d.to(dev), 16, tw, th,
                                                            want_slots=True)
    assert torch.equal(fids, f0) and torch.equal(off, o0)
    I = fids.numel()
    assert torch.equal(torch.sort(slot.long()).values.cpu(), torch.arange(I))  # the slots are a permutation
    assert int(row_cum[-1]) == I and row_cum.numel() == n
    owner = torch.empty(I, dtype=torch.int64)
    owner[slot.long().cpu()] = fids.long().cpu()  # row id stored at each slot
    cu = row_cum.cpu()
    starts = torch.cat((torch.zeros(1, dtype=torch.int64), cu[:-1]))
    cnt = cu - starts
    for i in torch.nonzero(cnt).flatten()[:300].tolist():  # row i owns the contiguous slots [starts[i], cu[i])
        assert bool((owner[starts[i]:cu[i]] == i).all())
    assert bool((cnt[radii[0] <= 0] == 0).all())
    assert torch.equal(cnt, torch.bincount(fids.long().cpu(), minlength=n))

    t = [x.to(dev).contiguous() for x in (m2, cn, colors, opac)]
    out = torch.empty(1, h, w, 3, device=dev); al = torch.empty(1, h, w, device=dev)
    last = torch.empty(1, h, w, dtype=torch.int32, device=dev)
    packed = torch.empty(n, 16, device=dev)
    check(L.clmgs_rasterize_fwd(stream(), 1, n, I, dptr(t[0]), dptr(t[1]), dptr(t[2]), dptr(t[3]), None, w, h, 16,
                                tw, th, dptr(off), dptr(fids), dptr(packed), dptr(out), dptr(al), dptr(last)))
    vi = torch.randn(1, h, w, 3, generator=g).to(dev)
    va = torch.randn(1, h, w, generator=g).to(dev)

    def bwd(slots):
        pg = torch.full((n, 16), float("nan"), device=dev)
        LF = L.clmgs_rasterize_partials_bytes(1) // 4  # floats per partial line (16: one 64 B line)
        parts = torch.full((max(I, 1), LF), float("nan"), device=dev)  # every line must be overwritten
        outs = [torch.empty(n, 2, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev),
                torch.empty(n, device=dev)]
        assert L.clmgs_rasterize_partials_bytes(I) == parts.numel() * 4
        check(L.clmgs_rasterize_bwd(stream(), 1, n, I, dptr(packed), None, w, h, 16, tw, th, dptr(off), dptr(fids),
                                    dptr(al), dptr(last), dptr(vi), dptr(va), dptr(pg), *[dptr(x) for x in outs],
                                    *((dptr(slot), dptr(row_cum), dptr(parts)) if slots else (None,) * 3)))
        torch.cuda.synchronize()
        if slots:  # the sum the engine path folds into clmgs_preprocess_bwd: row ranges of the partial lines
            ps, st = parts.cpu(), starts.tolist()
            for i in torch.nonzero(cnt).flatten()[:100].tolist():
                want = torch.zeros(LF)
                for l in range(st[i], int(cu[i])):
                    want = want + ps[l]
                assert torch.equal(want[:9], pg[i].cpu()[:9]), i
        return [x.cpu() for x in outs]

    ga, gs1, gs2 = bwd(False), bwd(True), bwd(True)
    for x, y in zip(gs1, gs2):
        assert torch.equal(x, y), "slot path must be bitwise reproducible"
    a = [x.clone().double().requires_grad_() for x in (m2, cn, colors, opac)]
    img0, al0 = O.rasterize_to_pixels(*a, w, h, 16, off.cpu(), fids.cpu())
    ((img0 * vi.cpu().double()).sum() + (al0[..., 0] * va.cpu().double()).sum()).backward()
    for name, x, y, ref in zip(("means2d", "conics", "colors", "opacities"), gs1, ga, a):
        assert torch.isfinite(x).all(), name
        assert rel_l2(x, y) < 1e-5, name
        assert rel_l2(x.reshape(ref.grad.shape), ref.grad) < GRAD_TOL, name


def test_adam_catch_up_equals_replayed_zero_gradient_steps(dev):
    """clmgs_adam_catch_up == the eager zero-gradient updates it defers (clmgs_adam_rows with
    g = NULL, step by step): rows at different staleness, rows whose moments are all zero (left
    untouched, bit for bit), and the max_replay cut (first max_replay steps exact, then only the
    moments decay)."""
    from clm_gs_amd import clm_kernels as K
    g = torch.Generator(device="cuda").manual_seed(2)
    n, cols, b1, b2, eps = 3000, 48, 0.9, 0.999, 1e-15
    col_lr = torch.cat([torch.full((3,), 2.5e-3), torch.full((45,), 1.25e-4)]).cuda()
    p0 = torch.randn(n, cols, generator=g, device="cuda")
    m0 = torch.randn(n, cols, generator=g, device="cuda") * 1e-3
    v0 = torch.rand(n, cols, generator=g, device="cuda") * 1e-6
    zero_rows = torch.arange(0, n, 7, device="cuda")
    m0[zero_rows] = 0.0
    v0[zero_rows] = 0.0
    last = torch.randint(3, 12, (n,), generator=g, device="cuda", dtype=torch.int32)
    to_step = 14
    for max_replay in (256, 4):
        p, m, v = p0.clone(), m0.clone(), v0.clone()
        K.adam_catch_up(p, m, v, last.clone(), None, col_lr, b1, b2, eps, to_step, True, max_replay=max_replay)
        pe, me, ve = p0.clone(), m0.clone(), v0.clone()
        pcut = None
        for step in range(4, to_step + 1):           # eager: every step, rows that have missed it
            rows = torch.nonzero(last < step).flatten().to(torch.int32)
            if max_replay == 4:                      # p frozen after a row's first 4 replayed steps
                frozen = rows[(step - last[rows.long()]) > 4]
                keep = pe[frozen.long()].clone()
            K.adam_rows(pe, None, me, ve, rows, col_lr, b1, b2, eps, step, True, 1.0, False)
            if max_replay == 4:
                pe[frozen.long()] = keep
        assert torch.equal(p[zero_rows], p0[zero_rows]) and float(m[zero_rows].abs().max()) == 0.0
        assert rel_l2(p, pe) < 1e-7 and rel_l2(m, me) < 1e-6 and rel_l2(v, ve) < 1e-6
        assert (p - pe).abs().max() < 5e-6   # bias corrections: running product vs pow()


def test_visibility_select_equals_radii_nonzero(dev):
    """GPU-side filter selection == nonzero(radii > 0) per camera, and its extra row == the union."""
    from clm_gs_amd import gsplat as G
    from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians
    for n in (1, 63, 64, 65, 5000, 70001):
        sc = synth_gaussians(n, seed=3, device="cuda")
        cams = nadir_cameras(3, max(n, 2000), 160, 96, 0.4, seed=1, device="cuda")
        Ks = torch.stack([c.K for c in cams])
        vms = torch.stack([c.world_view_transform.t() for c in cams])
        radii = G.visibility_radii(sc["xyz"], sc["rotation"], sc["scaling"], vms, Ks, 160, 96, raw=True)
        filters, union = G.visibility_select(sc["xyz"], sc["rotation"], sc["scaling"], vms, Ks, 160, 96)
        assert len(filters) == 3
        for c in range(3):
            assert torch.equal(filters[c], torch.nonzero(radii[c] > 0).flatten())
        assert torch.equal(union, torch.nonzero((radii > 0).any(dim=0)).flatten())


def test_visibility_select_conservative_bound_stress(dev):
    """The two-phase selection (cheap conservative screen test, then the exact projection of the
    survivors) must equal the exact test everywhere: tilted / off-centre cameras, heavy-tailed
    scales (huge splats whose 3-sigma box reaches the image from far outside), rows behind the
    camera and near the clipping plane, and more cameras than one group of four."""
    from clm_gs_amd import gsplat as G
    from clm_gs_amd.cameras import Camera
    g = torch.Generator().manual_seed(11)
    n, w, h = 40000, 200, 120
    xyz = (torch.rand(n, 3, generator=g) - 0.5) * torch.tensor([40.0, 40.0, 40.0])
    rot = torch.randn(n, 4, generator=g)
    log_s = torch.randn(n, 3, generator=g) * 1.5 - 2.0      # exp: 0.003 .. 10+
    log_s[::97] += 4.0                                       # a few enormous ones
    rot[5::1013] = 0.0                                       # degenerate quaternions (exact path: NaN -> culled)
    log_s[7::1511, 1] = float("nan")
    log_s[11::1999, 2] = float("inf")
    xyz[13::2003, 0] = float("nan")
    cams = []
    for i in range(7):
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
        if torch.det(q) < 0:
            q[:, 0] = -q[:, 0]
        w2c = torch.eye(4)
        w2c[:3, :3] = q
        w2c[:3, 3] = torch.randn(3, generator=g) * 8.0
        cams.append(Camera(i, w2c, 0.9 + 0.1 * i, 0.6 + 0.05 * i, w, h, device="cuda"))
    Ks = torch.stack([c.K for c in cams])
    Ks[1, 0, 2] += 37.0   # off-centre principal points
    Ks[2, 1, 2] -= 21.0
    vms = torch.stack([c.world_view_transform.t() for c in cams])
    xyz, rot, log_s = xyz.cuda(), rot.cuda(), log_s.cuda()
    radii = G.visibility_radii(xyz, rot, log_s, vms, Ks, w, h, raw=True)
    filters, union = G.visibility_select(xyz, rot, log_s, vms, Ks, w, h)
    assert len(filters) == 7
    fracs = []
    for c in range(7):
        assert torch.equal(filters[c], torch.nonzero(radii[c] > 0).flatten())
        fracs.append(filters[c].numel() / n)
    assert torch.equal(union, torch.nonzero((radii > 0).any(dim=0)).flatten())
    assert 0.01 < min(fracs) and max(fracs) < 0.9, fracs   # the test exercises both outcomes


def test_rasterize_with_no_intersections(dev):
    """Edge case: nothing lands on the image.  Forward = background / alpha 0, backward = exact zeros,
    through both accumulation modes of the C ABI."""
    from clm_gs_amd import _lib, gsplat as G
    from clm_gs_amd._lib import check, dptr, stream
    L = _lib.lib()
    w, h, n = 40, 23, 50
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    m2 = torch.full((1, n, 2), -500.0, device=dev)           # far off screen
    radii = torch.zeros((1, n), dtype=torch.int32, device=dev)  # culled
    depths = torch.ones((1, n), device=dev)
    fids, off, _, (slot, row_cum) = G.isect_tiles_two_level(m2, radii, depths, 16, tw, th, want_slots=True)
    assert fids.numel() == 0 and int(off.abs().sum()) == 0 and int(row_cum[-1]) == 0
    packed = torch.zeros(n, 16, device=dev)
    bg = torch.tensor([[0.25, 0.5, 0.75]], device=dev)
    out = torch.empty(1, h, w, 3, device=dev); al = torch.empty(1, h, w, device=dev)
    last = torch.empty(1, h, w, dtype=torch.int32, device=dev)
    check(L.clmgs_rasterize_fwd(stream(), 1, n, 0, None, None, None, None, dptr(bg), w, h, 16, tw, th, dptr(off),
                                None, dptr(packed), dptr(out), dptr(al), dptr(last)))
    assert torch.equal(out[0], bg.expand(h, w, 3)) and float(al.abs().max()) == 0.0
    vi = torch.randn(1, h, w, 3, device=dev)
    for slots in (False, True):
        pg = torch.full((n, 16), float("nan"), device=dev)
        parts = torch.empty((1, 16), device=dev)
        outs = [torch.full((n, 2), 7.0, device=dev), torch.full((n, 3), 7.0, device=dev),
                torch.full((n, 3), 7.0, device=dev), torch.full((n,), 7.0, device=dev)]
        check(L.clmgs_rasterize_bwd(stream(), 1, n, 0, dptr(packed), dptr(bg), w, h, 16, tw, th, dptr(off), None,
                                    dptr(al), dptr(last), dptr(vi), None, dptr(pg), *[dptr(x) for x in outs],
                                    *((dptr(slot, torch.int32, True), dptr(row_cum), dptr(parts)) if slots
                                      else (None,) * 3)))
        for x in outs:
            assert float(x.abs().max()) == 0.0


@pytest.mark.parametrize("slack", [1.0, 1.4, 0.6])
def test_device_count_forms_equal_exact_forms(dev, slack):
    """clmgs_isect2_emit_sort_dev / clmgs_rasterize_fwd_dev / clmgs_rasterize_bwd_dev: lists built for a CAPACITY
    with the true count read on the device == the exact forms, element for element, when count <= capacity
    (slack 1.0: capacity == count, 1.4: the usual over-allocation).  slack 0.6: capacity exceeded -- nothing is
    written out of bounds (guard words intact), the caller sees count > capacity from the totals."""
    from clm_gs_amd import _lib, gsplat as G
    from clm_gs_amd._lib import check, dptr, stream
    L = _lib.lib()
    w, h, n = 150, 101, 2500
    s = small_scene(n=n, width=w, height=h, seed=71, log_scale=-1.0)
    radii, m2, d, cn, _ = _project_cpu(s)
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    g = torch.Generator().manual_seed(5)
    colors = torch.rand(1, n, 3, generator=g).to(dev)
    opac = s["opac"].reshape(1, -1).to(dev)
    m2, radii, d, cn = m2.to(dev), radii.to(dev), d.to(dev), cn.to(dev)
    # exact forms
    fids, off, _, (slot, row_cum) = G.isect_tiles_two_level(m2, radii, d, 16, tw, th, want_slots=True)
    I = fids.numel()
    packed = torch.empty((n, 16), device=dev)
    out, al, last = torch.empty((h, w, 3), device=dev), torch.empty((h, w), device=dev), torch.empty((h, w), dtype=torch.int32, device=dev)
    check(L.clmgs_rasterize_fwd(stream(), 1, n, I, dptr(m2), dptr(cn), dptr(colors), dptr(opac), None, w, h, 16, tw, th,
                                dptr(off), dptr(fids), dptr(packed), dptr(out), dptr(al), dptr(last)))
    v_out = torch.randn((h, w, 3), generator=torch.Generator().manual_seed(6)).to(dev)
    part = torch.full((max(I, 1), 16), 7.0, device=dev)
    check(L.clmgs_rasterize_bwd(stream(), 1, n, I, dptr(packed), None, w, h, 16, tw, th, dptr(off), dptr(fids), dptr(al),
                                dptr(last), dptr(v_out), None, None, None, None, None, None, dptr(slot), dptr(row_cum),
                                dptr(part)))
    # device-count forms
    cap = max(1, int(I * slack))
    c = G.isect2_begin(m2, radii, d, 16, tw, th, want_slots=True)
    GUARD = 64
    fids2 = torch.full((cap + GUARD,), -12345, dtype=torch.int32, device=dev)
    slot2 = torch.full((cap + GUARD,), -12345, dtype=torch.int32, device=dev)
    sb = L.clmgs_isect2_sort_temp_bytes(cap)
    temp = torch.empty((sb,), dtype=torch.uint8, device=dev)
    off2 = torch.full((1, th, tw), -1, dtype=torch.int32, device=dev)
    check(L.clmgs_isect2_emit_sort_dev(stream(), n, cap, dptr(c.totals), dptr(c.depths), dptr(c.order), dptr(c.cum),
                                       dptr(c.boxes), tw, th, dptr(fids2), dptr(off2), None, dptr(slot2), dptr(temp), sb,
                                       dptr(c.row_cum)))
    torch.cuda.synchronize()
    assert int(c.totals[0]) == I
    assert bool((fids2[cap:] == -12345).all()) and bool((slot2[cap:] == -12345).all()), "nothing beyond the capacity"
    if slack < 1.0:
        return  # capacity exceeded: the caller redoes the camera (fused.camera_verify); only in-bounds writes are promised
    assert torch.equal(fids2[:I], fids) and torch.equal(slot2[:I], slot) and torch.equal(off2, off)
    out2, al2, last2 = torch.empty_like(out), torch.empty_like(al), torch.empty_like(last)
    check(L.clmgs_rasterize_fwd_dev(stream(), 1, n, cap, dptr(c.totals), None, w, h, 16, tw, th, dptr(off2), dptr(fids2[:cap]),
                                    dptr(packed), dptr(out2), dptr(al2), dptr(last2)))
    assert torch.equal(out2, out) and torch.equal(al2, al) and torch.equal(last2, last)
    part2 = torch.full((cap + 4, 16), 7.0, device=dev)
    check(L.clmgs_rasterize_bwd_dev(stream(), 1, n, cap, dptr(c.totals), dptr(packed), None, w, h, 16, tw, th, dptr(off2),
                                    dptr(fids2[:cap]), dptr(al2), dptr(last2), dptr(v_out), None, dptr(slot2[:cap]),
                                    dptr(c.row_cum), dptr(part2)))
    torch.cuda.synchronize()
    assert torch.equal(part2[:I], part[:I]) and bool((part2[I:] == 7.0).all())


class _use_library:
    """with _use_library(path): clm_gs_amd._lib serves the entry points of ANOTHER build of the library (the profiling
    build, which can still walk the older routes of the binning chain: csrc/isect.hip binning_route)."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        from clm_gs_amd import _lib
        self.keep = (_lib._lib, _lib.LIB_PATH)
        _lib._lib, _lib.LIB_PATH = None, self.path
        return _lib.lib()

    def __exit__(self, *exc):
        from clm_gs_amd import _lib
        _lib._lib, _lib.LIB_PATH = self.keep
        return False


def _profile_library():
    import os
    from clm_gs_amd import _lib
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libclmgs_hip_prof.so")
    if not os.path.exists(path):
        pytest.skip("profiling build absent (make -C clm_gs_amd/csrc PROFILE=1 BUILD=build_prof OUT=../libclmgs_hip_prof.so; "
                    "__graft_entry__.build() makes it)")
    return path


def _binning_lists(G, m2, radii, d, tw, th, n, pk, cap_slack, route=None):
    """Every list the two binning calls produce, through whatever library clm_gs_amd._lib currently serves."""
    import os
    if route:
        os.environ["CLMGS_BINNING"] = route
    try:
        c = G.isect2_begin(m2, radii, d, 16, tw, th, want_isect_ids=True, want_slots=True, packed=pk)
        torch.cuda.synchronize()
        tot = c.totals.clone()
        res = [c.order[:n].clone(), c.cum[:n].clone(), c.boxes[:n].clone(), tot, c.row_cum[:n].clone()]
        cap = None if cap_slack is None else max(1, int(int(tot[0]) * cap_slack))
        fids, off, ids, (slot, _) = G.isect2_finish(c, capacity=cap)
        torch.cuda.synchronize()
        I = int(tot[0]) if cap is None else min(int(tot[0]), cap)
        return res + [fids[:I].clone(), off.clone(), ids[:I].clone(), slot[:I].clone()]
    finally:
        os.environ.pop("CLMGS_BINNING", None)


_LIST_NAMES = ("order", "cum", "boxes", "totals", "row_cum", "flatten_ids", "offsets", "isect_ids", "emit_slot")


@pytest.mark.parametrize("n,wh", [(3000, (150, 101)), (180_000, (640, 480))])
def test_single_launch_binning_equals_legacy_chain(dev, n, wh):
    """The PRODUCT library's binning chain (round 5: scans folded into their producers, multi-chunk histograms, segment
    row scans, the chunk-driven emit that also counts the first tile-sort digit, the last tile-sort pass writing
    flatten_ids / emit_slot itself) == the round-3 chain (three launches per digit and per scan), element for element:
    depth order, both cumulative counts, boxes / masks, totals, flatten_ids, emit_slot, offsets, isect_ids -- exact form,
    device-count form (also with a capacity BELOW the count), with and without exact tile culling.  The older routes
    live in the profiling build only (CLMGS_BINNING=legacy | lookback | r4, csrc/isect.hip): they are compared too --
    r4 = round 4's thread-per-rank emit + separate histogram, lookback = one launch per radix digit (measured slower).
    Also the visibility selection (look-back scan vs three-launch scan)."""
    from clm_gs_amd import _lib, gsplat as G
    from clm_gs_amd._lib import check, dptr, stream
    from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians
    prof = _profile_library()
    L = _lib.lib()
    w, h = wh
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    sc = synth_gaussians(n, seed=11, device="cuda")
    cam = nadir_cameras(1, n, w, h, 0.9, seed=2, device="cuda")[0]
    vm = cam.world_view_transform.t().contiguous()[None]
    K = cam.K[None]
    quats = torch.nn.functional.normalize(sc["rotation"])
    radii, m2, d, cn, _ = G.fully_fused_projection(sc["xyz"], None, quats, torch.exp(sc["scaling"]), vm, K, w, h)
    d = d.clone()
    d[0, : n // 50] = d[0, 0]  # exact depth ties: stability of every pass matters
    packed = torch.empty((n, 16), device=dev)
    colors = torch.rand(1, n, 3, device=dev)
    opac = torch.sigmoid(sc["opacity"]).reshape(1, -1).contiguous()
    fids0, off0, _ = G.isect_tiles_two_level(m2, radii, d, 16, tw, th)
    out, al, last = torch.empty((h, w, 3), device=dev), torch.empty((h, w), device=dev), torch.empty((h, w), dtype=torch.int32, device=dev)
    check(L.clmgs_rasterize_fwd(stream(), 1, n, fids0.numel(), dptr(m2), dptr(cn), dptr(colors), dptr(opac), None, w, h, 16, tw,
                                th, dptr(off0), dptr(fids0), dptr(packed), dptr(out), dptr(al), dptr(last)))  # fills `packed`
    for pk in (None, packed):
        with _use_library(prof):
            ref = {sl: _binning_lists(G, m2, radii, d, tw, th, n, pk, sl, "legacy") for sl in (None, 0.6)}
        assert int(ref[None][3][0]) > (100_000 if n > 100_000 else 100)
        for slack in (None, 1.0, 1.3, 0.6):
            want = ref[0.6] if slack == 0.6 else ref[None]
            got = _binning_lists(G, m2, radii, d, tw, th, n, pk, slack)  # the product library
            for nm, a, b in zip(_LIST_NAMES, got, want):
                assert torch.equal(a, b), ("product", nm, pk is not None, slack)
            with _use_library(prof):
                for route in ("fused", "r4", "lookback"):
                    got = _binning_lists(G, m2, radii, d, tw, th, n, pk, slack, route)
                    for nm, a, b in zip(_LIST_NAMES, got, want):
                        assert torch.equal(a, b), (route, nm, pk is not None, slack)
    # visibility selection through the look-back scan == through the three-launch scan
    cams = nadir_cameras(3, n, w, h, 0.4, seed=5, device="cuda")
    Ks = torch.stack([c.K for c in cams])
    vms = torch.stack([c.world_view_transform.t() for c in cams])
    import os
    with _use_library(prof):
        os.environ["CLMGS_BINNING"] = "lookback"
        try:
            f_ref, u_ref = G.visibility_select(sc["xyz"], sc["rotation"], sc["scaling"], vms, Ks, w, h)
        finally:
            os.environ.pop("CLMGS_BINNING", None)
        _lib.check_device_errors()
    f_new, u_new = G.visibility_select(sc["xyz"], sc["rotation"], sc["scaling"], vms, Ks, w, h)
    assert torch.equal(u_new, u_ref) and all(torch.equal(a, b) for a, b in zip(f_new, f_ref))
    _lib.check_device_errors()


def test_chunked_emit_with_boxes_spanning_many_chunks(dev):
    """The chunk-driven emit (isect2_emit_hist_kernel: one block per 1024 entries of the list) on what a thread-per-rank
    emit never had to think about: rows whose box covers the WHOLE image (1 200 tiles = more than a chunk, unmasked),
    long runs of culled rows and of rows outside the image (ranks that emit nothing, in the middle of the depth
    order), exact depth ties, a list that ends in the middle of a chunk, and capacities that cut a rank in two -- against
    the round-3 chain of the profiling build, element for element."""
    from clm_gs_amd import gsplat as G
    prof = _profile_library()
    n, w, h = 60_000, 640, 480
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    g = torch.Generator().manual_seed(3)
    m2 = (torch.rand(1, n, 2, generator=g) * torch.tensor([w * 1.4, h * 1.4]) - torch.tensor([w * 0.2, h * 0.2])).to(dev)
    radii = torch.randint(1, 40, (1, n), generator=g, dtype=torch.int32)
    radii[0, torch.randperm(n, generator=g)[:40]] = 3000       # the whole image
    radii[0, torch.randperm(n, generator=g)[: n // 5]] = 0      # culled rows (sorted last)
    radii[0, 1000:9000] = torch.where(torch.rand(8000, generator=g) < 0.9, torch.zeros(8000, dtype=torch.int32), radii[0, 1000:9000])
    radii = radii.to(dev)
    d = (torch.rand(1, n, generator=g) * 50 + 1).to(dev)
    d[0, :2000] = 7.0
    m2[0, 20000:26000] = torch.tensor([-500.0, -500.0], device=dev)  # boxes clipped to nothing: ranks with zero entries
    for slack in (None, 1.0, 0.37, 0.9991):
        with _use_library(prof):
            want = _binning_lists(G, m2, radii, d, tw, th, n, None, slack, "legacy")
            r4 = _binning_lists(G, m2, radii, d, tw, th, n, None, slack, "r4")
        got = _binning_lists(G, m2, radii, d, tw, th, n, None, slack)
        assert int(want[3][0]) > 40 * tw * th
        for nm, a, b, c in zip(_LIST_NAMES, got, want, r4):
            assert torch.equal(a, b) and torch.equal(c, b), (nm, slack)


@pytest.mark.parametrize("bsz", [4, 8, 32, 64])
def test_host_groups_equals_index_arithmetic(dev, bsz):
    """clmgs_host_groups (host-resident batch: rows grouped by the camera that uses them first / last, slot table, group
    sizes) == plain index arithmetic on the same visibility bitmap, for every bitmap word width, with and without a
    set of already-staged rows."""
    from clm_gs_amd import _lib
    from clm_gs_amd._lib import check, dptr, stream
    from clm_gs_amd.strategies.clm_offload.engine import _encode_bitmap
    L = _lib.lib()
    N = 50_000
    g = torch.Generator().manual_seed(bsz)
    filters = []
    for i in range(bsz):
        k = int(torch.randint(N // 50, N // 6, (1,), generator=g))
        filters.append(torch.sort(torch.randperm(N, generator=g)[:k]).values.to(dev))
    seen = torch.zeros(N, dtype=torch.bool, device=dev)
    for f in filters:
        seen[f] = True
    touched = torch.nonzero(seen).flatten()
    T = touched.numel()
    bitmap = _encode_bitmap(filters, N, bsz)
    first = torch.full((N,), bsz, dtype=torch.int64, device=dev)
    last = torch.full((N,), -1, dtype=torch.int64, device=dev)
    for i, f in enumerate(filters):
        first[f] = torch.minimum(first[f], torch.full_like(f, i))
        last[f] = i
    for staged_frac in (0.0, 0.4):
        staged = torch.zeros(N, dtype=torch.bool, device=dev)
        if staged_frac:
            staged[touched[torch.rand(T, generator=g).to(dev) < staged_frac]] = True
        late = touched[~staged[touched]]
        want_late = late[torch.sort(first[late], stable=True).indices]
        want_last = touched[torch.sort(last[touched], stable=True).indices]
        slot0 = 7
        late_sorted = torch.full((T,), -1, dtype=torch.int32, device=dev)
        rows_by_last = torch.empty((T,), dtype=torch.int32, device=dev)
        slot_of = torch.full((N,), -5, dtype=torch.int32, device=dev)
        counts = torch.empty((2 * bsz + 1,), dtype=torch.int64, device=dev)
        tb = L.clmgs_host_groups_temp_bytes(T)
        tmp = torch.empty((tb,), dtype=torch.uint8, device=dev)
        check(L.clmgs_host_groups(stream(), T, dptr(touched, torch.int64), dptr(bitmap), bitmap.element_size(), bsz,
                                  dptr(staged.view(torch.uint8)) if staged_frac else None, slot0, dptr(late_sorted),
                                  dptr(rows_by_last), dptr(slot_of), dptr(counts), dptr(tmp), tb))
        torch.cuda.synchronize()
        cl = counts.tolist()
        n_late = cl[2 * bsz]
        assert n_late == late.numel()
        assert torch.equal(late_sorted[:n_late].long(), want_late) and torch.equal(rows_by_last.long(), want_last)
        assert cl[:bsz] == torch.bincount(first[late], minlength=bsz).tolist()
        assert cl[bsz:2 * bsz] == torch.bincount(last[touched], minlength=bsz).tolist()
        assert torch.equal(slot_of[want_late].long(), torch.arange(slot0, slot0 + n_late, device=dev))
        untouched = torch.ones(N, dtype=torch.bool, device=dev)
        untouched[want_late] = False
        assert bool((slot_of[untouched] == -5).all())


@pytest.mark.gpu
def test_visibility_candidates_cover_every_state_inside_the_drift_bounds(dev):
    """Camera-DP, small attributes at their owners (clmgs_visibility_candidates): whatever the owner did to a row
    inside the bounds (mean moved by <= pos_margin, scales multiplied by <= scale_gain or shrunk, ANY rotation), the
    rows the exact cull keeps for the TRUE state are candidates of the STALE state.  Stress scene of the conservative
    bound test (tilted / off-centre cameras, heavy-tailed scales, rows behind the camera / at the near plane, NaNs),
    several margins; the candidate set is not the trivial one; own rows are never listed."""
    from clm_gs_amd import gsplat as G
    from clm_gs_amd.cameras import Camera
    g = torch.Generator().manual_seed(21)
    n, w, h = 60000, 200, 120
    xyz = (torch.rand(n, 3, generator=g) - 0.5) * torch.tensor([60.0, 60.0, 60.0])
    rot = torch.randn(n, 4, generator=g)
    log_s = torch.randn(n, 3, generator=g) * 1.2 - 2.0
    log_s[::97] += 3.0
    log_s[7::1511, 1] = float("nan")
    xyz[13::2003, 0] = float("nan")
    cams = []
    for i in range(6):
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
        if torch.det(q) < 0:
            q[:, 0] = -q[:, 0]
        w2c = torch.eye(4)
        w2c[:3, :3] = q
        w2c[:3, 3] = torch.randn(3, generator=g) * 8.0
        cams.append(Camera(i, w2c, 0.9 + 0.1 * i, 0.6 + 0.05 * i, w, h, device="cuda"))
    Ks = torch.stack([c.K for c in cams])
    Ks[1, 0, 2] += 37.0
    vms = torch.stack([c.world_view_transform.t() for c in cams])
    xyz, rot, log_s = xyz.cuda(), rot.cuda(), log_s.cuda()
    gc = torch.Generator(device="cuda").manual_seed(5)
    sizes = []
    for margin, gain in ((0.0, 1.0), (0.05, 1.1), (0.5, 1.5), (3.0, 4.0)):
        # the TRUE state: the stale one moved to the edge of the bounds in random directions
        d = torch.randn(n, 3, device="cuda", generator=gc)
        d = d / d.norm(dim=1, keepdim=True).clamp_min(1e-9) * margin * torch.rand(n, 1, device="cuda", generator=gc) ** 0.2
        true_xyz = xyz + d
        true_ls = log_s + (torch.rand(n, 3, device="cuda", generator=gc) * 2.0 - 1.0) * math.log(gain)
        true_rot = torch.randn(n, 4, device="cuda", generator=gc)
        _, union_true = G.visibility_select(true_xyz, true_rot, true_ls, vms, Ks, w, h)
        _, union_stale = G.visibility_select(xyz, rot, log_s, vms, Ks, w, h)
        cand = G.visibility_candidates(xyz, log_s, vms, Ks, w, h, pos_margin=margin, scale_gain=gain,
                                       own_lo=0, own_hi=0)
        assert cand.dtype == torch.int64 and bool((cand[1:] > cand[:-1]).all())
        is_c = torch.zeros(n, dtype=torch.bool, device="cuda")
        is_c[cand] = True
        missed = union_true[~is_c[union_true]]
        assert missed.numel() == 0, (margin, gain, missed[:10].tolist())
        assert bool(is_c[union_stale].all())              # ... and of course what the stale state itself shows
        sizes.append((cand.numel(), union_true.numel()))
        # own rows are never candidates, the others are unaffected by the range
        lo, hi = n // 3, 2 * n // 3
        c2 = G.visibility_candidates(xyz, log_s, vms, Ks, w, h, pos_margin=margin, scale_gain=gain, own_lo=lo, own_hi=hi)
        assert torch.equal(c2, cand[(cand < lo) | (cand >= hi)])
    assert sizes[0][0] < 1.6 * sizes[0][1] and sizes[1][0] < 1.8 * sizes[1][1], sizes   # tight at small margins
    assert sizes[-1][0] < n, sizes


@pytest.mark.gpu
def test_small_adam_on_a_row_range_and_small_rows_scatter(dev):
    """clmgs_adam_small_packed_range steps rows [lo, hi) exactly as the full call does and leaves every other row of
    every table bit for bit alone (ranges that are not multiples of the 256-row block, of 4, or of anything);
    clmgs_small_rows_scatter writes packed lines to the four tensors + the mirror at the listed rows only."""
    import ctypes
    from clm_gs_amd import _lib, dp
    n = 10007
    g = torch.Generator(device="cuda").manual_seed(3)
    widths = (3, 1, 3, 4)

    def state():
        gg = torch.Generator(device="cuda").manual_seed(9)
        ps = [torch.randn(n, w, device="cuda", generator=gg) for w in widths]
        ms = [torch.randn(n, w, device="cuda", generator=gg) * 0.1 for w in widths]
        vs = [torch.rand(n, w, device="cuda", generator=gg) * 0.01 for w in widths]
        pk = torch.empty(n, 12, device="cuda")
        _lib.check(_lib.lib().clmgs_pack_small(_lib.stream(), n, *[_lib.dptr(x) for x in ps], _lib.dptr(pk)))
        return ps, ms, vs, pk
    grads = torch.randn(n, 12, device="cuda", generator=g)
    stamp = torch.where(torch.rand(n, device="cuda", generator=g) < 0.5, 7, 3).to(torch.int32)
    arr = lambda xs: (ctypes.c_void_p * 4)(*[x.data_ptr() for x in xs])
    lrs = (ctypes.c_double * 4)(1e-2, 5e-2, 5e-3, 1e-3)

    def run(st, lo, hi):
        ps, ms, vs, pk = st
        _lib.check(_lib.lib().clmgs_adam_small_packed_range(
            _lib.stream(), n, lo, hi, arr(ps), arr(ms), arr(vs), lrs, _lib.dptr(pk), _lib.dptr(grads.clone()),
            0.9, 0.999, 1e-15, 5, 1, 0.25, _lib.dptr(stamp), 7))
    full = state()
    run(full, 0, -1)
    for lo, hi in ((0, n), (1, 2), (255, 257), (1001, 7777), (3334, 6669), (n - 1, n), (500, 500)):
        part, ref = state(), state()
        run(part, lo, hi)
        for kind in range(3):
            for t_part, t_full, t_ref in zip(part[kind], full[kind], ref[kind]):
                assert torch.equal(t_part[lo:hi], t_full[lo:hi])
                assert torch.equal(t_part[:lo], t_ref[:lo]) and torch.equal(t_part[hi:], t_ref[hi:])
        assert torch.equal(part[3][lo:hi], full[3][lo:hi])
        assert torch.equal(part[3][:lo], ref[3][:lo]) and torch.equal(part[3][hi:], ref[3][hi:])
    # scatter
    ps, _, _, pk = state()
    before = [p.clone() for p in ps] + [pk.clone()]
    rows = torch.randperm(n, device="cuda", generator=g)[:1234].sort().values
    lines = torch.randn(rows.numel(), 12, device="cuda", generator=g)
    dp.small_scatter(rows, lines, pk, ps)
    want = lines.clone()
    want[:, 11] = 0.0
    assert torch.equal(pk[rows], want)
    assert torch.equal(torch.cat([p[rows] for p in ps], dim=1), lines[:, :11])
    rest = torch.ones(n, dtype=torch.bool, device="cuda")
    rest[rows] = False
    for a, b in zip(ps + [pk], before):
        assert torch.equal(a[rest], b[rest])


def test_visibility_candidates_kernel_brackets_its_numpy_restatement(dev):
    """The kernel's candidate set pinned from both sides by the float64 restatement of tests/test_dp_drift_bounds.py:
    it contains the restatement evaluated WITHOUT the kernel's rounding slack (1 px + 0.1 % on the radius), and is
    contained in the restatement with TWICE that slack -- the kernel computes the documented test, nothing looser."""
    import numpy as np
    from clm_gs_amd import gsplat as G
    from clm_gs_amd.cameras import Camera
    from tests import test_dp_drift_bounds as T
    g = torch.Generator().manual_seed(31)
    n, w, h = 50000, 320, 200
    xyz = (torch.rand(n, 3, generator=g) - 0.5) * 60.0
    log_s = torch.randn(n, 3, generator=g) * 1.2 - 2.0
    for i, (d, gain) in enumerate(((0.0, 1.0), (0.3, 1.5), (2.0, 3.0))):
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
        if torch.det(q) < 0:
            q[:, 0] = -q[:, 0]
        w2c = torch.eye(4)
        w2c[:3, :3] = q
        w2c[:3, 3] = torch.randn(3, generator=g) * 6.0
        cam = Camera(i, w2c, 0.9, 0.7, w, h, device="cuda")
        K = cam.K.cpu().double().numpy()
        cand = G.visibility_candidates(xyz.cuda(), log_s.cuda(), cam.world_view_transform.t()[None].contiguous(),
                                       cam.K[None].contiguous(), w, h, pos_margin=d, scale_gain=gain, own_lo=0, own_hi=0)
        got = np.zeros(n, dtype=bool)
        got[cand.cpu().numpy()] = True
        cam_np = (q.double().numpy(), w2c[:3, 3].double().numpy(), K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        smax = np.exp(log_s.double().numpy().max(axis=1))

        def restated(slack_px, slack_rel):
            R, t, fx, fy, cx, cy = cam_np
            x, y, z = (xyz.double().numpy() @ R.T + t).T
            zl, zh = z - d, z + d
            alive = ~((zh < 0.01) | (zl > 1e10))
            zc = np.maximum(np.maximum(zl, 0.01), 1e-12)
            zf = np.maximum(np.minimum(zh, 1e10), zc)
            xh, xl, yh, yl = x + d, x - d, y + d, y - d
            tx_max, tx_min = np.where(xh > 0, xh / zc, xh / zf), np.where(xl < 0, xl / zc, xl / zf)
            ty_max, ty_min = np.where(yh > 0, yh / zc, yh / zf), np.where(yl < 0, yl / zc, yl / zf)
            B = (smax * gain) ** 2 / (zc * zc) * T._kc(fx, fy, cx, cy, w, h) + 0.3
            Rb = (1.0 + slack_rel) * (3.03 * np.sqrt(2 * B + 0.1) + 2.0) + slack_px
            out = ((fx * tx_max + cx + Rb <= 0) | (fx * tx_min + cx - Rb >= w)
                   | (fy * ty_max + cy + Rb <= 0) | (fy * ty_min + cy - Rb >= h))
            return alive & ~out
        inner, outer = restated(0.0, 0.0), restated(2.0, 0.0022)
        assert not np.any(inner & ~got), (d, gain, int(np.sum(inner & ~got)))
        assert not np.any(got & ~outer), (d, gain, int(np.sum(got & ~outer)))
        assert 0 < inner.sum() < n


def _tile_major_lists(G, m2, radii, d, tw, th, n, pk, cap_slack):
    c = G.isect3_begin(m2, radii, d, 16, tw, th, want_isect_ids=True, want_slots=True, packed=pk)
    torch.cuda.synchronize()
    tot = c.totals.clone()
    cap = None if cap_slack is None else max(1, int(int(tot[0]) * cap_slack))
    fids, off, ids, (slot, row_cum) = G.isect3_finish(c, capacity=cap)
    torch.cuda.synchronize()
    I = int(tot[0])
    return [tot, row_cum[:n].clone(), fids[:I].clone(), off.clone(), ids[:I].clone(), slot[:I].clone()]


@pytest.mark.parametrize("n,wh", [(3000, (150, 101)), (180_000, (640, 480))])
def test_tile_major_binning_equals_two_level_sort(dev, n, wh):
    """Round 5: the tile-major route (csrc/isect3.hip: per-tile counters, scatter, one LDS sort per tile by (depth bits,
    row index) -- the engine's default) produces the lists of the two-level route (csrc/isect.hip: depth sort of the rows
    + stable sort on the tile id) element for element: totals, row_cum, flatten_ids, offsets, isect_ids, emit_slot; exact
    form and device-count form with room to spare; with and without exact tile culling; with exact depth ties."""
    from clm_gs_amd import _lib, gsplat as G
    from clm_gs_amd._lib import check, dptr, stream
    from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians
    L = _lib.lib()
    w, h = wh
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    sc = synth_gaussians(n, seed=11, device="cuda")
    cam = nadir_cameras(1, n, w, h, 0.9, seed=2, device="cuda")[0]
    vm = cam.world_view_transform.t().contiguous()[None]
    K = cam.K[None]
    quats = torch.nn.functional.normalize(sc["rotation"])
    radii, m2, d, cn, _ = G.fully_fused_projection(sc["xyz"], None, quats, torch.exp(sc["scaling"]), vm, K, w, h)
    d = d.clone()
    d[0, : n // 50] = d[0, 0]  # exact depth ties: broken by row index in both routes
    packed = torch.empty((n, 16), device=dev)
    colors = torch.rand(1, n, 3, device=dev)
    opac = torch.sigmoid(sc["opacity"]).reshape(1, -1).contiguous()
    fids0, off0, _ = G.isect_tiles_two_level(m2, radii, d, 16, tw, th)
    out, al, last = torch.empty((h, w, 3), device=dev), torch.empty((h, w), device=dev), torch.empty((h, w), dtype=torch.int32, device=dev)
    check(L.clmgs_rasterize_fwd(stream(), 1, n, fids0.numel(), dptr(m2), dptr(cn), dptr(colors), dptr(opac), None, w, h, 16, tw,
                                th, dptr(off0), dptr(fids0), dptr(packed), dptr(out), dptr(al), dptr(last)))  # fills `packed`
    names = ("totals", "row_cum", "flatten_ids", "offsets", "isect_ids", "emit_slot")
    for pk in (None, packed):
        ref = _binning_lists(G, m2, radii, d, tw, th, n, pk, None)
        want = [ref[3], ref[4], ref[5], ref[6], ref[7], ref[8]]
        assert int(want[0][0]) > (100_000 if n > 100_000 else 100)
        for slack in (None, 1.0, 1.3):
            got = _tile_major_lists(G, m2, radii, d, tw, th, n, pk, slack)
            for nm, a, b in zip(names, got, want):
                assert torch.equal(a, b), (nm, pk is not None, slack)
        # a capacity below the count: memory-safe, every stored id a valid row, offsets monotone and within the capacity
        c = G.isect3_begin(m2, radii, d, 16, tw, th, want_slots=True, packed=pk)
        cap = int(int(want[0][0]) * 0.6)
        fids, off, _, (slot, _) = G.isect3_finish(c, capacity=cap)
        torch.cuda.synchronize()
        assert int(fids.min()) >= 0 and int(fids.max()) < n and int(off.max()) <= cap
        o = off.reshape(-1)
        assert bool((o[1:] >= o[:-1]).all())


def test_tile_major_binning_long_lists_and_whole_image_boxes(dev):
    """The tile-major route where its special paths run: tiles with more than 1 024 entries (the 256-thread LDS sort),
    one tile with more than 8 192 (the sort in global memory), rows whose box covers the whole image (expanded by a
    whole workgroup), empty tiles, rows that emit nothing, exact depth ties -- against the two-level route."""
    from clm_gs_amd import gsplat as G
    n, w, h = 70_000, 640, 480
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    g = torch.Generator().manual_seed(4)
    m2 = (torch.rand(1, n, 2, generator=g) * torch.tensor([w * 0.5, h * 0.5])).to(dev)   # everything in one quadrant
    radii = torch.randint(1, 30, (1, n), generator=g, dtype=torch.int32)
    m2[0, :12_000] = torch.tensor([100.0, 100.0], device=dev) + torch.rand(12_000, 2, generator=g).to(dev) * 4.0  # one tile, > 8192
    radii[0, :12_000] = 1
    m2[0, 12_000:16_000] = torch.tensor([300.0, 40.0], device=dev) + torch.rand(4_000, 2, generator=g).to(dev) * 4.0  # > 1024
    radii[0, 12_000:16_000] = 1
    radii[0, torch.randperm(n, generator=g)[:30]] = 3000       # the whole image
    radii[0, torch.randperm(n, generator=g)[: n // 6]] = 0      # culled
    radii = radii.to(dev)
    d = (torch.rand(1, n, generator=g) * 50 + 1).to(dev)
    d[0, :3000] = 7.0
    ref = _binning_lists(G, m2, radii, d, tw, th, n, None, None)
    want = [ref[3], ref[4], ref[5], ref[6], ref[7], ref[8]]
    cnt = torch.bincount((ref[7] >> 32).to(torch.int64), minlength=tw * th)
    assert int(cnt.max()) > 8192 and int((cnt > 1024).sum()) >= 2 and int((cnt <= 256).sum()) > 0
    for slack in (None, 1.0):
        got = _tile_major_lists(G, m2, radii, d, tw, th, n, None, slack)
        for nm, a, b in zip(("totals", "row_cum", "flatten_ids", "offsets", "isect_ids", "emit_slot"), got, want):
            assert torch.equal(a, b), (nm, slack)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 257, 100_003, 3_000_001])
def test_library_morton_order_equals_the_torch_form(dev, n):
    """clmgs_morton_order (one pass of IEEE double arithmetic + the stable radix sort) is the SAME permutation as
    utils.morton_order_torch -- on clustered points with many equal codes (ties keep the row order), exact duplicates,
    coordinates on the quantisation half-way points and a degenerate axis."""
    from clm_gs_amd import utils
    g = torch.Generator().manual_seed(n)
    xyz = torch.randn(n, 3, generator=g) * torch.tensor([40.0, 3.0, 1.0])
    if n > 1000:
        xyz[: n // 4, :2] = (xyz[: n // 4, :2] * 0.01).round() / 0.01           # many equal codes
        xyz[n // 4: n // 2] = xyz[: n // 2 - n // 4].clone()                             # exact duplicates
        lo, hi = xyz[:, 0].min().double(), xyz[:, 0].max().double()
        k = torch.arange(0, 2000, dtype=torch.float64)
        xyz[-2000:, 0] = (lo + (hi - lo) * (k + 0.5) / 65535.0).float()           # (near) half-way between two cells
    xyz = xyz.to(dev)
    for t in (xyz, torch.cat((xyz[:, :1], torch.full_like(xyz[:, :1], 2.5), xyz[:, 2:]), dim=1)):  # (y constant)
        a = utils.morton_order(t)
        b = utils.morton_order_torch(t)
        assert a.dtype == torch.int64 and a.shape == b.shape
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_row_mover_picks_per_row_tensors_like_indexing(dev):
    """The pick of a structural change (gaussian_model._gather_f32: clmgs_rows_gather on [N], [N,1], [N,3], [N,4]
    float tensors, int64 ids) equals plain indexing."""
    from clm_gs_amd.strategies.clm_offload.gaussian_model import _gather_f32
    g = torch.Generator().manual_seed(3)
    n = 200_001
    idx = torch.randperm(n, generator=g)[: n - 777].to(dev)
    for shape in ((n,), (n, 1), (n, 3), (n, 4), (n, 12)):
        t = torch.randn(shape, generator=g).to(dev)
        out = _gather_f32(t, idx)
        assert out.shape == (idx.numel(),) + tuple(shape[1:]) and out.is_contiguous()
        assert torch.equal(out, t[idx])
    assert _gather_f32(torch.randn(n, 3).to(dev), idx[:0]).shape == (0, 3)


@pytest.mark.gpu
def test_tile_major_binning_with_nothing_on_the_image(dev):
    """Edge cases of the tile-major route: every row culled (radius 0), every row off screen, and no rows at all --
    empty lists, all-zero offsets, row_cum of zeros; in the exact form and in the device-count form (capacity 1)."""
    from clm_gs_amd import gsplat as G
    w, h, n = 100, 70, 300
    tw, th = math.ceil(w / 16), math.ceil(h / 16)
    depths = torch.ones((1, n), device=dev)
    cases = [(torch.full((1, n, 2), 30.0, device=dev), torch.zeros((1, n), dtype=torch.int32, device=dev)),      # culled
             (torch.full((1, n, 2), -900.0, device=dev), torch.full((1, n), 5, dtype=torch.int32, device=dev))]  # off screen
    for m2, radii in cases:
        for cap in (None, 1):
            c = G.isect3_begin(m2, radii, depths, 16, tw, th, want_isect_ids=True, want_slots=True)
            fids, off, ids, (slot, row_cum) = G.isect3_finish(c, capacity=cap)
            torch.cuda.synchronize()
            assert int(c.totals[0]) == 0
            assert int(off.abs().sum()) == 0 and off.shape == (1, th, tw)
            assert int(row_cum[:n].abs().sum()) == 0
            if cap is None:
                assert fids.numel() == 0 and slot.numel() == 0
    e = torch.empty((1, 0), device=dev)
    c = G.isect3_begin(torch.empty((1, 0, 2), device=dev), torch.empty((1, 0), dtype=torch.int32, device=dev), e, 16, tw, th,
                       want_isect_ids=True, want_slots=True)
    fids, off, ids, (slot, row_cum) = G.isect3_finish(c)
    assert fids.numel() == 0 and int(off.abs().sum()) == 0 and ids.numel() == 0 and slot.numel() == 0


@pytest.mark.gpu
def test_morton_order_of_no_rows_and_of_one_point_many_times(dev):
    from clm_gs_amd import utils
    assert utils.morton_order(torch.empty((0, 3), device=dev)).numel() == 0
    same = torch.full((1000, 3), 3.25, device=dev)          # hi == lo on both axes: every code 0, the identity (stable)
    assert torch.equal(utils.morton_order(same), torch.arange(1000, device=dev))
