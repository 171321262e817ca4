// Per-element arithmetic of the splatting hot path, written once and inlined
// into the gfx950 kernels.  The same header compiles with plain g++ (CLMGS_HD
// empty) so tests/ can check the formulas against autograd on a CPU-only box;
// the product never runs that build.
//
// Restates the published gsplat algorithms the reference calls
// (strategies/base_engine.py:36-47,161-203).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CLMGS_HD __host__ __device__ __forceinline__
#else
#define CLMGS_HD inline
#endif

namespace clmgs {

struct Cam {
  float R[9];  // world->camera rotation, row major
  float t[3];
  float fx, fy, cx, cy;
};

CLMGS_HD Cam load_cam(const float* viewmat /*4x4 row major*/, const float* K /*3x3*/) {
  Cam c;
  c.R[0] = viewmat[0]; c.R[1] = viewmat[1]; c.R[2] = viewmat[2];  c.t[0] = viewmat[3];
  c.R[3] = viewmat[4]; c.R[4] = viewmat[5]; c.R[5] = viewmat[6];  c.t[1] = viewmat[7];
  c.R[6] = viewmat[8]; c.R[7] = viewmat[9]; c.R[8] = viewmat[10]; c.t[2] = viewmat[11];
  c.fx = K[0]; c.cx = K[2]; c.fy = K[4]; c.cy = K[5];
  return c;
}

// ---------------------------------------------------------------- quaternion
// q = (w,x,y,z), normalised here (utils/general_utils.py:311-334 layout).
CLMGS_HD void quat_to_rotmat(const float q[4], float R[9]) {
  float inv = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
  R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z);       R[2] = 2.f * (x * z + w * y);
  R[3] = 2.f * (x * y + w * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
  R[6] = 2.f * (x * z - w * y);       R[7] = 2.f * (y * z + w * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// VJP of quat_to_rotmat (including the normalisation).
CLMGS_HD void quat_to_rotmat_vjp(const float q[4], const float vR[9], float vq[4]) {
  float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  float inv = 1.0f / sqrtf(n2);
  float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
  float vn[4];
  vn[0] = 2.f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
  vn[1] = 2.f * (-2.f * x * (vR[4] + vR[8]) + y * (vR[1] + vR[3]) + z * (vR[2] + vR[6]) + w * (vR[7] - vR[5]));
  vn[2] = 2.f * (x * (vR[1] + vR[3]) - 2.f * y * (vR[0] + vR[8]) + z * (vR[5] + vR[7]) + w * (vR[2] - vR[6]));
  vn[3] = 2.f * (x * (vR[2] + vR[6]) + y * (vR[5] + vR[7]) - 2.f * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
  float dot = vn[0] * w + vn[1] * x + vn[2] * y + vn[3] * z;
  vq[0] = (vn[0] - dot * w) * inv;
  vq[1] = (vn[1] - dot * x) * inv;
  vq[2] = (vn[2] - dot * y) * inv;
  vq[3] = (vn[3] - dot * z) * inv;
}

// ---------------------------------------------------------------- projection
struct Proj {
  int radius;       // 0 = culled
  float mx, my;     // pixel-space mean
  float depth;      // camera-space z
  float ca, cb, cc; // conic = inverse of blurred 2D covariance
};

// Intermediate values shared by forward and backward.
struct ProjMid {
  float p[3];        // camera-space mean
  float M[9];        // R(q) * diag(s)
  float Sc[6];       // camera-space covariance, upper triangle xx xy xz yy yz zz
  float J00, J02, J11, J12;
  float tx, ty;
  bool clamp_x, clamp_y;
  float c00, c01, c11, det;  // blurred 2D covariance
};

CLMGS_HD bool project_mid(const Cam& cam, const float m[3], const float q[4], const float s[3],
                          float W, float H, float eps2d, float near_plane, float far_plane,
                          ProjMid& o) {
  const float* Rv = cam.R;
  o.p[0] = Rv[0] * m[0] + Rv[1] * m[1] + Rv[2] * m[2] + cam.t[0];
  o.p[1] = Rv[3] * m[0] + Rv[4] * m[1] + Rv[5] * m[2] + cam.t[1];
  o.p[2] = Rv[6] * m[0] + Rv[7] * m[1] + Rv[8] * m[2] + cam.t[2];
  if (o.p[2] < near_plane || o.p[2] > far_plane) return false;
  float R[9];
  quat_to_rotmat(q, R);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o.M[i * 3 + j] = R[i * 3 + j] * s[j];
  // A = Rv * M  (3x3);  Sc = A A^T
  float A[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      A[i * 3 + j] = Rv[i * 3 + 0] * o.M[0 * 3 + j] + Rv[i * 3 + 1] * o.M[1 * 3 + j] + Rv[i * 3 + 2] * o.M[2 * 3 + j];
  o.Sc[0] = A[0] * A[0] + A[1] * A[1] + A[2] * A[2];
  o.Sc[1] = A[0] * A[3] + A[1] * A[4] + A[2] * A[5];
  o.Sc[2] = A[0] * A[6] + A[1] * A[7] + A[2] * A[8];
  o.Sc[3] = A[3] * A[3] + A[4] * A[4] + A[5] * A[5];
  o.Sc[4] = A[3] * A[6] + A[4] * A[7] + A[5] * A[8];
  o.Sc[5] = A[6] * A[6] + A[7] * A[7] + A[8] * A[8];

  float x = o.p[0], y = o.p[1], z = o.p[2];
  float tan_fovx = 0.5f * W / cam.fx, tan_fovy = 0.5f * H / cam.fy;
  float lim_x_pos = (W - cam.cx) / cam.fx + 0.3f * tan_fovx;
  float lim_x_neg = cam.cx / cam.fx + 0.3f * tan_fovx;
  float lim_y_pos = (H - cam.cy) / cam.fy + 0.3f * tan_fovy;
  float lim_y_neg = cam.cy / cam.fy + 0.3f * tan_fovy;
  float rz = 1.f / z, rz2 = rz * rz;
  float xr = x * rz, yr = y * rz;
  o.clamp_x = (xr < -lim_x_neg) || (xr > lim_x_pos);
  o.clamp_y = (yr < -lim_y_neg) || (yr > lim_y_pos);
  o.tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, xr));
  o.ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, yr));
  o.J00 = cam.fx * rz;
  o.J02 = -cam.fx * o.tx * rz2;
  o.J11 = cam.fy * rz;
  o.J12 = -cam.fy * o.ty * rz2;
  // cov2d = J Sc J^T
  float sxx = o.Sc[0], sxy = o.Sc[1], sxz = o.Sc[2], syy = o.Sc[3], syz = o.Sc[4], szz = o.Sc[5];
  float c00 = o.J00 * (o.J00 * sxx + o.J02 * sxz) + o.J02 * (o.J00 * sxz + o.J02 * szz);
  float c01 = o.J00 * (o.J11 * sxy + o.J12 * sxz) + o.J02 * (o.J11 * syz + o.J12 * szz);
  float c11 = o.J11 * (o.J11 * syy + o.J12 * syz) + o.J12 * (o.J11 * syz + o.J12 * szz);
  o.c00 = c00 + eps2d;
  o.c11 = c11 + eps2d;
  o.c01 = c01;
  o.det = o.c00 * o.c11 - o.c01 * o.c01;
  return o.det > 0.f;
}

CLMGS_HD Proj project_fwd(const Cam& cam, const float m[3], const float q[4], const float s[3],
                          float W, float H, float eps2d, float near_plane, float far_plane,
                          float radius_clip) {
  Proj r;
  r.radius = 0; r.mx = r.my = r.depth = r.ca = r.cb = r.cc = 0.f;
  ProjMid o;
  if (!project_mid(cam, m, q, s, W, H, eps2d, near_plane, far_plane, o)) return r;
  float rz = 1.f / o.p[2];
  float mx = cam.fx * o.p[0] * rz + cam.cx;
  float my = cam.fy * o.p[1] * rz + cam.cy;
  float b = 0.5f * (o.c00 + o.c11);
  float v1 = b + sqrtf(fmaxf(0.01f, b * b - o.det));
  float radius = ceilf(3.f * sqrtf(v1));
  if (radius <= radius_clip) return r;
  if (mx + radius <= 0.f || mx - radius >= W || my + radius <= 0.f || my - radius >= H) return r;
  float idet = 1.f / o.det;
  r.radius = (int)radius;
  r.mx = mx; r.my = my; r.depth = o.p[2];
  r.ca = o.c11 * idet; r.cb = -o.c01 * idet; r.cc = o.c00 * idet;
  return r;
}

// VJP of project_fwd for one (camera, Gaussian); caller guarantees radius > 0.
// v_m / v_q / v_s are OVERWRITTEN with this camera's contribution.
CLMGS_HD void project_bwd(const Cam& cam, const float m[3], const float q[4], const float s[3],
                          float W, float H, float eps2d,
                          const float v_mean2d[2], float v_depth, const float v_conic[3],
                          float v_m[3], float v_q[4], float v_s[3]) {
  ProjMid o;
  project_mid(cam, m, q, s, W, H, eps2d, -1e30f, 1e30f, o);
  float idet = 1.f / o.det;
  float a = o.c11 * idet, b = -o.c01 * idet, c = o.c00 * idet;  // conic
  // v_cov2d = -inv * V * inv, V = [[va, vb/2],[vb/2, vc]]
  float va = v_conic[0], vb = 0.5f * v_conic[1], vc = v_conic[2];
  // T = V * inv
  float t00 = va * a + vb * b, t01 = va * b + vb * c;
  float t10 = vb * a + vc * b, t11 = vb * b + vc * c;
  float g00 = -(a * t00 + b * t10);
  float g01 = -(a * t01 + b * t11);
  float g10 = -(b * t00 + c * t10);
  float g11 = -(b * t01 + c * t11);
  // symmetrise: cov2d's off-diagonal appears twice
  float G00 = g00, G11 = g11, G01 = 0.5f * (g01 + g10);

  float sxx = o.Sc[0], sxy = o.Sc[1], sxz = o.Sc[2], syy = o.Sc[3], syz = o.Sc[4], szz = o.Sc[5];
  float J00 = o.J00, J02 = o.J02, J11 = o.J11, J12 = o.J12;
  // v_Sc = J^T G J  (symmetric 3x3), J = [[J00,0,J02],[0,J11,J12]]
  float vS[9];
  vS[0] = J00 * G00 * J00;
  vS[1] = J00 * G01 * J11;
  vS[2] = J00 * (G00 * J02 + G01 * J12);
  vS[4] = J11 * G11 * J11;
  vS[5] = J11 * (G01 * J02 + G11 * J12);
  vS[8] = J02 * (G00 * J02 + G01 * J12) + J12 * (G01 * J02 + G11 * J12);
  vS[3] = vS[1]; vS[6] = vS[2]; vS[7] = vS[5];
  // v_J = 2 G J Sc   (G, Sc symmetric)
  // (J Sc) rows:
  float js00 = J00 * sxx + J02 * sxz, js01 = J00 * sxy + J02 * syz, js02 = J00 * sxz + J02 * szz;
  float js10 = J11 * sxy + J12 * sxz, js11 = J11 * syy + J12 * syz, js12 = J11 * syz + J12 * szz;
  float vJ00 = 2.f * (G00 * js00 + G01 * js10);
  float vJ02 = 2.f * (G00 * js02 + G01 * js12);
  float vJ11 = 2.f * (G01 * js01 + G11 * js11);
  float vJ12 = 2.f * (G01 * js02 + G11 * js12);

  float x = o.p[0], y = o.p[1], z = o.p[2];
  float rz = 1.f / z, rz2 = rz * rz, rz3 = rz2 * rz;
  float vp[3];
  vp[0] = cam.fx * rz * v_mean2d[0];
  vp[1] = cam.fy * rz * v_mean2d[1];
  vp[2] = -(cam.fx * x * v_mean2d[0] + cam.fy * y * v_mean2d[1]) * rz2 + v_depth;
  // J00 = fx/z, J11 = fy/z
  vp[2] += -cam.fx * rz2 * vJ00 - cam.fy * rz2 * vJ11;
  // J02 = -fx*tx/z^2 ; tx = x (free) or z*lim (clamped)
  if (!o.clamp_x) {
    vp[0] += -cam.fx * rz2 * vJ02;
    vp[2] += 2.f * cam.fx * o.tx * rz3 * vJ02;
  } else {
    vp[2] += cam.fx * o.tx * rz3 * vJ02;
  }
  if (!o.clamp_y) {
    vp[1] += -cam.fy * rz2 * vJ12;
    vp[2] += 2.f * cam.fy * o.ty * rz3 * vJ12;
  } else {
    vp[2] += cam.fy * o.ty * rz3 * vJ12;
  }
  const float* Rv = cam.R;
  // v_m = Rv^T vp
  v_m[0] = Rv[0] * vp[0] + Rv[3] * vp[1] + Rv[6] * vp[2];
  v_m[1] = Rv[1] * vp[0] + Rv[4] * vp[1] + Rv[7] * vp[2];
  v_m[2] = Rv[2] * vp[0] + Rv[5] * vp[1] + Rv[8] * vp[2];
  // v_Sigma(world) = Rv^T vS Rv
  float tmp[9], vW[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      tmp[i * 3 + j] = vS[i * 3 + 0] * Rv[0 * 3 + j] + vS[i * 3 + 1] * Rv[1 * 3 + j] + vS[i * 3 + 2] * Rv[2 * 3 + j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      vW[i * 3 + j] = Rv[0 * 3 + i] * tmp[0 * 3 + j] + Rv[1 * 3 + i] * tmp[1 * 3 + j] + Rv[2 * 3 + i] * tmp[2 * 3 + j];
  // Sigma = M M^T -> v_M = (vW + vW^T) M = 2 vW M (vW symmetric)
  float vM[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      vM[i * 3 + j] = 2.f * (vW[i * 3 + 0] * o.M[0 * 3 + j] + vW[i * 3 + 1] * o.M[1 * 3 + j] + vW[i * 3 + 2] * o.M[2 * 3 + j]);
  // M = R diag(s)
  float R[9];
  quat_to_rotmat(q, R);
  float vR[9];
  for (int j = 0; j < 3; ++j) {
    v_s[j] = R[0 * 3 + j] * vM[0 * 3 + j] + R[1 * 3 + j] * vM[1 * 3 + j] + R[2 * 3 + j] * vM[2 * 3 + j];
    for (int i = 0; i < 3; ++i) vR[i * 3 + j] = vM[i * 3 + j] * s[j];
  }
  quat_to_rotmat_vjp(q, vR, v_q);
}

// ----------------------------------------------------- spherical harmonics
// Real SH basis up to degree 3 on a normalised direction; same polynomial set
// as utils/sh_utils.py:73-103 with signs folded in.  nb = (deg+1)^2.
CLMGS_HD void sh_basis(int deg, float x, float y, float z, float B[16]) {
  B[0] = 0.28209479177387814f;
  if (deg < 1) return;
  B[1] = -0.4886025119029199f * y;
  B[2] = 0.4886025119029199f * z;
  B[3] = -0.4886025119029199f * x;
  if (deg < 2) return;
  float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  B[4] = 1.0925484305920792f * xy;
  B[5] = -1.0925484305920792f * yz;
  B[6] = 0.31539156525252005f * (2.f * zz - xx - yy);
  B[7] = -1.0925484305920792f * xz;
  B[8] = 0.5462742152960396f * (xx - yy);
  if (deg < 3) return;
  B[9] = -0.5900435899266435f * y * (3.f * xx - yy);
  B[10] = 2.890611442640554f * xy * z;
  B[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
  B[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
  B[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy);
  B[14] = 1.445305721320277f * z * (xx - yy);
  B[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
}

// dB/dx, dB/dy, dB/dz on the unit direction (before the normalisation VJP).
CLMGS_HD void sh_basis_grad(int deg, float x, float y, float z, float Bx[16], float By[16], float Bz[16]) {
  Bx[0] = By[0] = Bz[0] = 0.f;
  if (deg < 1) return;
  const float c1 = 0.4886025119029199f;
  Bx[1] = 0.f; By[1] = -c1; Bz[1] = 0.f;
  Bx[2] = 0.f; By[2] = 0.f; Bz[2] = c1;
  Bx[3] = -c1; By[3] = 0.f; Bz[3] = 0.f;
  if (deg < 2) return;
  const float c20 = 1.0925484305920792f, c22 = 0.31539156525252005f, c24 = 0.5462742152960396f;
  Bx[4] = c20 * y;        By[4] = c20 * x;        Bz[4] = 0.f;
  Bx[5] = 0.f;            By[5] = -c20 * z;       Bz[5] = -c20 * y;
  Bx[6] = -2.f * c22 * x; By[6] = -2.f * c22 * y; Bz[6] = 4.f * c22 * z;
  Bx[7] = -c20 * z;       By[7] = 0.f;            Bz[7] = -c20 * x;
  Bx[8] = 2.f * c24 * x;  By[8] = -2.f * c24 * y; Bz[8] = 0.f;
  if (deg < 3) return;
  float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  const float c30 = -0.5900435899266435f, c31 = 2.890611442640554f, c32 = -0.4570457994644658f,
              c33 = 0.3731763325901154f, c35 = 1.445305721320277f;
  // B9 = c30 * y (3xx - yy)
  Bx[9] = c30 * 6.f * xy;            By[9] = c30 * (3.f * xx - 3.f * yy); Bz[9] = 0.f;
  // B10 = c31 xyz
  Bx[10] = c31 * yz;                 By[10] = c31 * xz;                    Bz[10] = c31 * xy;
  // B11 = c32 y (4zz - xx - yy)
  Bx[11] = c32 * (-2.f * xy);        By[11] = c32 * (4.f * zz - xx - 3.f * yy); Bz[11] = c32 * 8.f * yz;
  // B12 = c33 z (2zz - 3xx - 3yy)
  Bx[12] = c33 * (-6.f * xz);        By[12] = c33 * (-6.f * yz);           Bz[12] = c33 * (6.f * zz - 3.f * xx - 3.f * yy);
  // B13 = c32 x (4zz - xx - yy)
  Bx[13] = c32 * (4.f * zz - 3.f * xx - yy); By[13] = c32 * (-2.f * xy);   Bz[13] = c32 * 8.f * xz;
  // B14 = c35 z (xx - yy)
  Bx[14] = c35 * 2.f * xz;           By[14] = -c35 * 2.f * yz;             Bz[14] = c35 * (xx - yy);
  // B15 = c30 x (xx - 3yy)
  Bx[15] = c30 * (3.f * xx - 3.f * yy); By[15] = c30 * (-6.f * xy);        Bz[15] = 0.f;
}

// ------------------------------------------------ exact footprint tests (culling)
// min over t in [lo, hi] of  q = 0.5 * (Af * f^2 + 2 * B * f * t + Ct * t^2)   (f fixed, Ct > 0)
CLMGS_HD float edge_min_sigma(float Af, float B, float Ct, float rcpCt, float f,
                                                float lo, float hi) {
  const float t = fminf(fmaxf(-B * f * rcpCt, lo), hi);
  return 0.5f * (Af * f * f + (2.f * B * f + Ct * t) * t);
}

// Exact minimum of sigma(d) = 0.5 (a dx^2 + 2 b dx dy + c dy^2) over the rectangle
// [ux0, ux1] x [vy0, vy1] of offsets from the Gaussian centre (convex: 0 if the centre is inside,
// otherwise attained on one of the four edges).
CLMGS_HD float rect_min_sigma(float ca, float cb, float cc, float rca, float rcc,
                                                float ux0, float ux1, float vy0, float vy1) {
  if (ux0 <= 0.f && ux1 >= 0.f && vy0 <= 0.f && vy1 >= 0.f) return 0.f;
  const float e0 = edge_min_sigma(ca, cb, cc, rcc, ux0, vy0, vy1);
  const float e1 = edge_min_sigma(ca, cb, cc, rcc, ux1, vy0, vy1);
  const float e2 = edge_min_sigma(cc, cb, ca, rca, vy0, ux0, ux1);
  const float e3 = edge_min_sigma(cc, cb, ca, rca, vy1, ux0, ux1);
  return fminf(fminf(e0, e1), fminf(e2, e3));
}

// ------------------------------------------------------------- alpha blend
// One (pixel, Gaussian) evaluation.  Returns false when the pair is skipped
// (sigma < 0 or alpha < 1/255).  dx,dy = mean - pixel centre.
CLMGS_HD bool blend_alpha(float dx, float dy, float ca, float cb, float cc, float opac,
                          float& alpha, float& gexp) {
  float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
  gexp = expf(-sigma);
  alpha = fminf(0.999f, opac * gexp);
  return !(sigma < 0.f || alpha < (1.f / 255.f));
}

}  // namespace clmgs
