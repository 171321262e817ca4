"""CPU: the two inequalities the small-attribute owner-computes exchange of camera-DP rests on
(clm_gs_amd/strategies/clm_offload/gaussian_model.py small_after_step / small_prepare, csrc/isect.hip vis_candidate).

1. Adam's step bound: no element moves by more than C(b1, b2) x lr in one step, whatever the gradient history --
   C = (1 - b1) / sqrt(1 - b2) / sqrt(1 - b1^2 / b2) = 7.27 for (0.9, 0.999).  Checked on random, sparse, sign-flipping
   and ADVERSARIAL histories (the geometric sequence that makes Cauchy-Schwarz tight), with moments restarted in the
   middle of a run (opacity reset / new rows) under the global step's bias correction.
2. The drift-dilated cull: a numpy restatement of the kernel's conservative screen test (vis_classify's cull branch) and
   of vis_candidate; for random cameras, rows and states inside the bounds, "not culled for the TRUE state" implies
   "candidate for the STALE state".
"""
import math

import numpy as np


def _adam_bound(b1, b2):
    return (1.0 - b1) / math.sqrt(1.0 - b2) / math.sqrt(1.0 - b1 * b1 / b2)


def _adam_max_ratio(grads, b1=0.9, b2=0.999, eps=1e-15, t0=1, m=0.0, v=0.0):
    """max over the steps of |p_t - p_{t-1}| / lr for torch.optim.Adam's update, starting at global step t0."""
    worst = 0.0
    for k, g in enumerate(grads):
        t = t0 + k
        m = b1 * m + (1.0 - b1) * g
        v = b2 * v + (1.0 - b2) * g * g
        step = (m / (1.0 - b1 ** t)) / (math.sqrt(v) / math.sqrt(1.0 - b2 ** t) + eps)
        worst = max(worst, abs(step))
    return worst


def test_adam_step_bound_holds_for_any_gradient_history():
    b1, b2 = 0.9, 0.999
    C = _adam_bound(b1, b2)
    assert 7.26 < C < 7.28
    rng = np.random.default_rng(0)
    worst = 0.0
    for trial in range(200):
        n = int(rng.integers(1, 400))
        kind = trial % 5
        if kind == 0:
            g = rng.standard_normal(n)
        elif kind == 1:                                   # sparse: long runs of zero gradient, then a spike
            g = np.where(rng.random(n) < 0.05, rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6), 0.0)
        elif kind == 2:                                   # constant sign, growing
            g = np.abs(rng.standard_normal(n)) * np.linspace(1e-3, 1e3, n)
        elif kind == 3:                                   # the tight case of Cauchy-Schwarz: g_{t-k} ~ (b1 / b2)^k
            g = (b1 / b2) ** np.arange(n)[::-1] * (1.0 if trial % 2 else -1.0)
        else:                                             # tiny then huge (v dominated by the last one)
            g = np.concatenate((np.full(n, 1e-8), [1e4]))
        t0 = 1 if trial % 3 else int(rng.integers(1, 5000))   # moments restarted at a late global step
        worst = max(worst, _adam_max_ratio(list(g), b1, b2, t0=t0))
    assert worst <= C * 1.0001, worst
    assert worst > 0.9 * C, worst                         # the adversarial history comes close: the bound is not slack


def _cam(rng, W, H):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    t = rng.standard_normal(3) * 6.0
    fx, fy = W / (2 * math.tan(rng.uniform(0.5, 1.2) / 2)), H / (2 * math.tan(rng.uniform(0.4, 1.0) / 2))
    cx, cy = W / 2 + rng.uniform(-30, 30), H / 2 + rng.uniform(-20, 20)
    return q, t, fx, fy, cx, cy


def _kc(fx, fy, cx, cy, W, H):
    tfx, tfy = 0.5 * W / fx, 0.5 * H / fy
    limx = max(abs((W - cx) / fx), abs(cx / fx)) + 0.3 * tfx
    limy = max(abs((H - cy) / fy), abs(cy / fy)) + 0.3 * tfy
    return 0.505 * (fx * fx * (1 + limx * limx) + fy * fy * (1 + limy * limy))


def _not_culled(cam, W, H, mean, smax, eps2d=0.3, near=0.01, far=1e10):
    """vis_classify(...) != 0 (csrc/isect.hip): the conservative screen test every visible row passes."""
    R, t, fx, fy, cx, cy = cam
    x, y, z = (mean @ R.T + t).T
    ok = (z >= near) & (z <= far)
    rz = 1.0 / np.where(ok, z, 1.0)
    mx, my = fx * x * rz + cx, fy * y * rz + cy
    B = smax * smax * rz * rz * _kc(fx, fy, cx, cy, W, H) + eps2d
    Rb = 3.03 * np.sqrt(2 * B + 0.1) + 2.0
    return ok & ~((mx + Rb <= 0) | (mx - Rb >= W) | (my + Rb <= 0) | (my - Rb >= H))


def _candidate(cam, W, H, mean, smax, d, gain, eps2d=0.3, near=0.01, far=1e10):
    """vis_candidate (csrc/isect.hip) in float64."""
    R, t, fx, fy, cx, cy = cam
    x, y, z = (mean @ R.T + t).T
    zl, zh = z - d, z + d
    alive = ~((zh < near) | (zl > far))
    zc = np.maximum(np.maximum(zl, near), 1e-12)
    zf = np.maximum(np.minimum(zh, far), zc)
    xh, xl, yh, yl = x + d, x - d, y + d, y - d
    tx_max = np.where(xh > 0, xh / zc, xh / zf)
    tx_min = np.where(xl < 0, xl / zc, xl / zf)
    ty_max = np.where(yh > 0, yh / zc, yh / zf)
    ty_min = np.where(yl < 0, yl / zc, yl / zf)
    sg = smax * gain
    B = sg * sg / (zc * zc) * _kc(fx, fy, cx, cy, W, H) + eps2d
    Rb = 1.001 * (3.03 * np.sqrt(2 * B + 0.1) + 2.0) + 1.0
    out = ((fx * tx_max + cx + Rb <= 0) | (fx * tx_min + cx - Rb >= W)
           | (fy * ty_max + cy + Rb <= 0) | (fy * ty_min + cy - Rb >= H))
    return alive & ~out


def test_drift_dilated_cull_contains_the_cull_of_every_state_inside_the_bounds():
    rng = np.random.default_rng(1)
    W, H = 320, 200
    n = 20000
    total_true = total_cand = 0
    for trial in range(24):
        cam = _cam(rng, W, H)
        stale = (rng.random((n, 3)) - 0.5) * 60.0
        smax = np.exp(rng.standard_normal(n) * 1.2 - 2.0)
        d, gain = [(0.0, 1.0), (0.01, 1.05), (0.3, 1.5), (2.0, 3.0)][trial % 4]
        # true states: moved up to d in a random direction (half of them to the very edge), scale x up to `gain`
        u = rng.standard_normal((n, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        r = np.where(rng.random(n) < 0.5, 1.0, rng.random(n))
        true_mean = stale + u * (d * r)[:, None]
        true_smax = smax * np.exp(rng.uniform(-1.0, 1.0, n) * math.log(gain))
        seen = _not_culled(cam, W, H, true_mean, true_smax)
        cand = _candidate(cam, W, H, stale, smax, d, gain)
        assert not np.any(seen & ~cand), (trial, d, gain, int(np.sum(seen & ~cand)))
        assert np.all(cand[_not_culled(cam, W, H, stale, smax)])          # margins >= 0: contains the plain test
        total_true += int(seen.sum())
        total_cand += int(cand.sum())
    assert 0 < total_true < total_cand < 24 * n                            # neither side is trivial
