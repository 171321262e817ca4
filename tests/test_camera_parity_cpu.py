"""CPU: the full-size parity checker itself (oracle/camera_parity.py) -- its accounting of fp32 ties.

The checker compares two fp32 implementations at millions of samples, where step functions (ceil of the
3-sigma radius, the sign of image - gt in the L1 term) cannot agree bit for bit.  Here the "other
implementation" is the C oracle run on inputs perturbed in the last bits, which must (a) be accepted,
with every exception counted and explained, and (b) be rejected as soon as a real error is injected."""
import math

import numpy as np
import torch

from oracle import camera_parity as CP
from tests.scenes import small_scene


def _inp(n=3000, w=160, h=112, seed=3):
    s = small_scene(n=n, width=w, height=h, seed=seed)
    raw_q = s["quats"].numpy().astype(np.float32)
    g = torch.Generator().manual_seed(seed)
    gt = (torch.rand(3, h, w, generator=g) * 255).to(torch.uint8).numpy()
    return dict(means=s["means"].numpy(), quats=torch.nn.functional.normalize(s["quats"]).numpy(),
                scales=s["scales"].numpy(), opac=s["opac"].numpy(), shs=s["shs"].reshape(n, 48).numpy(),
                raw_q=raw_q, viewmat=s["viewmat"].numpy(), K=s["K"].numpy(), gt=gt), w, h


def _as_hip(orc, v_image=None):
    """Dress an oracle result as the dict hip_camera returns."""
    d = {k: orc[k] for k in ("image", "loss", "radii", "n_isects", "max_radii2D", "xyz_gradient_accum", "denom", "means2d") + CP.GRAD_KEYS}
    d["v_image"] = orc["v_image"] if v_image is None else v_image
    d["n_emitted"] = orc["n_isects"]
    return d


def test_identical_runs_have_no_ties_and_zero_error():
    inp, w, h = _inp()
    a, _ = CP.oracle_camera(inp, w, h)
    b, _ = CP.oracle_camera(inp, w, h, v_image_hip=a["v_image"])
    rep = CP.compare(_as_hip(a), b)
    assert rep["radii_mismatch"] == 0 and rep["cotangent_sign_flips"] == 0
    assert rep["n_isects_hip"] == rep["n_isects_from_hip_boxes"] == rep["n_isects_explained"] == rep["n_isects_oracle"]
    assert rep["psnr_db"] > 150 and rep["loss_abs"] == 0
    # SH / opacity gradients accumulate with float atomics-free loops in a fixed order: bitwise equal
    assert all(rep[k + "_rel_l2"] == 0.0 for k in CP.GRAD_KEYS)
    assert all(rep["same_cotangent_" + k + "_rel_l2"] == 0.0 for k in CP.GRAD_KEYS)
    assert CP.within_tolerance(rep) == []


def test_last_bit_perturbation_is_accepted_and_ties_are_counted():
    inp, w, h = _inp()
    ref, _ = CP.oracle_camera(inp, w, h)
    p = dict(inp)
    p["means"] = (inp["means"].astype(np.float64) * (1 + 3e-7)).astype(np.float32)   # another operation order
    p["scales"] = (inp["scales"].astype(np.float64) * (1 - 2e-7)).astype(np.float32)
    other, _ = CP.oracle_camera(p, w, h)
    ref2, _ = CP.oracle_camera(inp, w, h, v_image_hip=other["v_image"])
    rep = CP.compare(_as_hip(other), ref2)
    assert rep["radii_unexplained"] == 0
    assert rep["n_isects_hip"] == rep["n_isects_from_hip_boxes"] == rep["n_isects_explained"]
    assert rep["psnr_db"] > 90
    assert rep["cotangent_rel_l2_without_flips"] < 1e-4
    for k in CP.GRAD_KEYS:
        assert rep["same_cotangent_" + k + "_rel_l2"] < 1e-4, (k, rep)
    assert CP.within_tolerance(rep) == [], rep


def test_real_errors_are_rejected():
    inp, w, h = _inp()
    ref, _ = CP.oracle_camera(inp, w, h)
    # (1) a 1 % error in one gradient tensor
    bad = _as_hip(ref)
    bad["g_scaling"] = np.asarray(ref["g_scaling"]) * 1.01
    ref2, _ = CP.oracle_camera(inp, w, h, v_image_hip=ref["v_image"])
    v = CP.within_tolerance(CP.compare(bad, ref2))
    assert "same_cotangent:g_scaling" in v
    # (2) radii wrong by 2 on a few rows: not a tie
    bad = _as_hip(ref)
    r = ref["radii"].copy()
    vis = np.nonzero(r > 0)[0][:3]
    r[vis] += 2
    bad["radii"] = r
    assert "radii" in CP.within_tolerance(CP.compare(bad, ref2))
    # (3) an intersection total that the radii do not explain
    bad = _as_hip(ref)
    bad["n_isects"] = ref["n_isects"] + 7
    assert "n_isects" in CP.within_tolerance(CP.compare(bad, ref2))
    # (4) a shifted image
    bad = _as_hip(ref)
    bad["image"] = ref["image"] + 2e-3
    assert "psnr" in CP.within_tolerance(CP.compare(bad, ref2))
