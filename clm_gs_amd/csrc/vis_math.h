// Conservative visibility tests shared by the visibility kernels (isect.hip) and the deferred small-attribute Adam
// (clm_ops.hip): per-camera constants as they sit in LDS, and the drift-dilated candidate test.
#pragma once
#include "common.h"
#include "gs_math.h"

namespace clmgs {

constexpr int VB_CAM_F = 20;    // floats per camera in LDS: R[9] t[3] fx fy cx cy Kc (+pad)
constexpr int VB_MAX_CAMS = 64;

// Kc = 0.5 (fx^2 (1 + limx^2) + fy^2 (1 + limy^2)) * 1.01: bound of 0.5 |J|_F^2 z^2
__device__ __forceinline__ float vis_cam_kc(const Cam& c, float W, float H) {
  const float tan_fovx = 0.5f * W / c.fx, tan_fovy = 0.5f * H / c.fy;
  const float limx = fmaxf(fabsf((W - c.cx) / c.fx), fabsf(c.cx / c.fx)) + 0.3f * tan_fovx;
  const float limy = fmaxf(fabsf((H - c.cy) / c.fy), fabsf(c.cy / c.fy)) + 0.3f * tan_fovy;
  return 0.505f * (c.fx * c.fx * (1.f + limx * limx) + c.fy * c.fy * (1.f + limy * limy));
}

__device__ __forceinline__ Cam lds_cam(const float* f) {
  Cam c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.R[i] = f[i];
  c.t[0] = f[9]; c.t[1] = f[10]; c.t[2] = f[11];
  c.fx = f[12]; c.fy = f[13]; c.cx = f[14]; c.cy = f[15];
  return c;
}


// Camera-DP with the small attributes computed by their owners (clm_gs_amd/dp.py, step S): between two refreshes a
// rank's copy of a row it does not own is STALE -- the owner has moved the mean by at most `pos_margin` (Euclidean:
// every camera-space coordinate moves by no more than that, the view rotation is orthonormal) and multiplied the
// largest scale by at most `scale_gain` (bounds from Adam's step bound, gaussian_model.py small_after_step).  A row is
// a CANDIDATE if vis_classify could return non-zero for ANY state inside those bounds: depth interval against the
// planes, the extreme values of x'/z' and y'/z' over the box, the radius bound B at the closest admissible depth.
// Superset of every row the exact cull keeps, so after the candidates' current values have been fetched from their
// owners the exact pass over the whole table selects exactly what it selects on one rank: rows that are not
// candidates fail it with their stale values too (margins >= 0) and fail it with their true ones (superset).
// NaNs fail every cull comparison -> candidate.  One extra pixel and 0.1 % on the radius cover the rounding of this
// evaluation against vis_classify's.
__device__ __forceinline__ bool vis_candidate(const Cam& c, float kc, const float m[3], float smax2g, float d,
                                              float W, float H, float eps2d, float near_m, float far_m) {
  const float x = c.R[0] * m[0] + c.R[1] * m[1] + c.R[2] * m[2] + c.t[0];
  const float y = c.R[3] * m[0] + c.R[4] * m[1] + c.R[5] * m[2] + c.t[1];
  const float z = c.R[6] * m[0] + c.R[7] * m[1] + c.R[8] * m[2] + c.t[2];
  const float zl = z - d, zh = z + d;
  if (zh < near_m || zl > far_m) return false;
  const float zc = fmaxf(fmaxf(zl, near_m), 1e-12f), zf = fmaxf(fminf(zh, far_m), zc);
  const float rzc = 1.f / zc, rzf = 1.f / zf;
  const float xh = x + d, xl = x - d, yh = y + d, yl = y - d;
  const float tx_max = xh > 0.f ? xh * rzc : xh * rzf, tx_min = xl < 0.f ? xl * rzc : xl * rzf;
  const float ty_max = yh > 0.f ? yh * rzc : yh * rzf, ty_min = yl < 0.f ? yl * rzc : yl * rzf;
  const float B = smax2g * rzc * rzc * kc + eps2d;
  const float Rb = 1.001f * (3.03f * sqrtf(2.f * B + 0.1f) + 2.f) + 1.f;
  const float mx_max = c.fx * tx_max + c.cx, mx_min = c.fx * tx_min + c.cx;
  const float my_max = c.fy * ty_max + c.cy, my_min = c.fy * ty_min + c.cy;
  return !(mx_max + Rb <= 0.f || mx_min - Rb >= W || my_max + Rb <= 0.f || my_min - Rb >= H);
}


// the per-camera LDS record of the kernels above, from the host-side view matrix and intrinsics
__device__ __forceinline__ void vis_store_cam(float* f, const float* viewmat, const float* K, float W, float H) {
  const Cam cam = load_cam(viewmat, K);
  for (int i = 0; i < 9; ++i) f[i] = cam.R[i];
  f[9] = cam.t[0]; f[10] = cam.t[1]; f[11] = cam.t[2];
  f[12] = cam.fx; f[13] = cam.fy; f[14] = cam.cx; f[15] = cam.cy;
  f[16] = vis_cam_kc(cam, W, H);
}

}  // namespace clmgs
