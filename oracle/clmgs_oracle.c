/* clmgs_oracle.c -- plain-C (OpenMP) CPU restatement of the CLM-GS rasterization path.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ (mid-size parity), __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg (kind "port").  Nothing under clm_gs_amd/ links or loads it.
 *
 * PARITY UNPINNED at the kernel boundary -- see oracle/gs_oracle.py's header: the reference's
 * own arithmetic for this path lives in absent third-party CUDA submodules (gsplat @ b60e917c...,
 * clm_kernels unpinned), and the reference has no tests.  This file restates the published
 * algorithm and is itself checked against gs_oracle.py (torch
 * autograd) in tests/test_oracle.py.
 *
 * One entry point per stage, following the order of strategies/no_offload/engine.py:15-101:
 *   orc_project      A1  (engine.py:49-60)     orc_project_bwd
 *   orc_sh           A2  (engine.py:67-69)     orc_sh_bwd
 *   orc_isect        A3+A4 (engine.py:75-84)   (count, emit, stable sort, offsets)
 *   orc_rasterize    A5  (engine.py:86-97)     orc_rasterize_bwd  A6
 *   orc_loss         A7 + L1 (strategies/base_engine.py:79-103; utils/loss_utils.py:18-85)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* cpu_baseline leg: use every host core the process may run on (the OpenMP runtime may have been
 * initialised earlier, by another library, with fewer). */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------ A1 projection */
static void quat_rot(const float* q, float* R) {
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  float w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

static void mat3_mul(const float* A, const float* B, float* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0;
      for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}
static void mat3_mul_bt(const float* A, const float* B, float* C) { /* A * B^T */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0;
      for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[j * 3 + k];
      C[i * 3 + j] = s;
    }
}
static void mat3_mul_at(const float* A, const float* B, float* C) { /* A^T * B */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0;
      for (int k = 0; k < 3; ++k) s += A[k * 3 + i] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}

typedef struct {
  float p[3], Sigma_c[9], J[6], tx, ty, c00, c01, c11, det;
  int cl_x, cl_y;
} pm_t;

static int proj_mid(const float* vm, const float* K, const float* m, const float* q, const float* s,
                    float W, float H, float eps2d, float znear, float zfar, pm_t* o, float* Mout,
                    float* Rout) {
  float Rv[9] = {vm[0], vm[1], vm[2], vm[4], vm[5], vm[6], vm[8], vm[9], vm[10]};
  float t[3] = {vm[3], vm[7], vm[11]};
  for (int i = 0; i < 3; ++i) o->p[i] = Rv[i * 3] * m[0] + Rv[i * 3 + 1] * m[1] + Rv[i * 3 + 2] * m[2] + t[i];
  if (o->p[2] < znear || o->p[2] > zfar) return 0;
  float R[9], M[9], Sigma[9], tmp[9];
  quat_rot(q, R);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i * 3 + j] = R[i * 3 + j] * s[j];
  mat3_mul_bt(M, M, Sigma);
  mat3_mul(Rv, Sigma, tmp);
  mat3_mul_bt(tmp, Rv, o->Sigma_c);
  if (Mout) memcpy(Mout, M, sizeof(M));
  if (Rout) memcpy(Rout, R, sizeof(R));
  float fx = K[0], cx = K[2], fy = K[4], cy = K[5];
  float x = o->p[0], y = o->p[1], z = o->p[2];
  float tanx = 0.5f * W / fx, tany = 0.5f * H / fy;
  float lxp = (W - cx) / fx + 0.3f * tanx, lxn = cx / fx + 0.3f * tanx;
  float lyp = (H - cy) / fy + 0.3f * tany, lyn = cy / fy + 0.3f * tany;
  float xr = x / z, yr = y / z;
  o->cl_x = (xr < -lxn) || (xr > lxp);
  o->cl_y = (yr < -lyn) || (yr > lyp);
  o->tx = z * fminf(lxp, fmaxf(-lxn, xr));
  o->ty = z * fminf(lyp, fmaxf(-lyn, yr));
  float J[6] = {fx / z, 0, -fx * o->tx / (z * z), 0, fy / z, -fy * o->ty / (z * z)};
  memcpy(o->J, J, sizeof(J));
  /* cov2d = J Sigma_c J^T */
  float JS[6];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j) {
      float a = 0;
      for (int k = 0; k < 3; ++k) a += J[i * 3 + k] * o->Sigma_c[k * 3 + j];
      JS[i * 3 + j] = a;
    }
  float c[4];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) {
      float a = 0;
      for (int k = 0; k < 3; ++k) a += JS[i * 3 + k] * J[j * 3 + k];
      c[i * 2 + j] = a;
    }
  o->c00 = c[0] + eps2d; o->c01 = c[1]; o->c11 = c[3] + eps2d;
  o->det = o->c00 * o->c11 - o->c01 * o->c01;
  return o->det > 0.f;
}

void orc_project(int N, const float* means, const float* quats, const float* scales,
                 const float* viewmat, const float* K, int width, int height, float eps2d,
                 float znear, float zfar, float radius_clip, int32_t* radii, float* means2d,
                 float* depths, float* conics) {
  const float W = (float)width, H = (float)height;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; ++i) {
    pm_t o;
    radii[i] = 0; means2d[2 * i] = means2d[2 * i + 1] = 0; depths[i] = 0;
    conics[3 * i] = conics[3 * i + 1] = conics[3 * i + 2] = 0;
    if (!proj_mid(viewmat, K, means + 3 * i, quats + 4 * i, scales + 3 * i, W, H, eps2d, znear, zfar, &o, 0, 0))
      continue;
    float mx = K[0] * o.p[0] / o.p[2] + K[2], my = K[4] * o.p[1] / o.p[2] + K[5];
    float b = 0.5f * (o.c00 + o.c11);
    float v1 = b + sqrtf(fmaxf(0.01f, b * b - o.det));
    float r = ceilf(3.f * sqrtf(v1));
    if (r <= radius_clip) continue;
    if (mx + r <= 0 || mx - r >= W || my + r <= 0 || my - r >= H) continue;
    radii[i] = (int32_t)r; means2d[2 * i] = mx; means2d[2 * i + 1] = my; depths[i] = o.p[2];
    conics[3 * i] = o.c11 / o.det; conics[3 * i + 1] = -o.c01 / o.det; conics[3 * i + 2] = o.c00 / o.det;
  }
}

void orc_project_bwd(int N, const float* means, const float* quats, const float* scales,
                     const float* viewmat, const float* K, int width, int height, float eps2d,
                     const int32_t* radii, const float* v_means2d, const float* v_depths,
                     const float* v_conics, float* v_means, float* v_quats, float* v_scales) {
  const float W = (float)width, H = (float)height;
  const float* vm = viewmat;
  float Rv[9] = {vm[0], vm[1], vm[2], vm[4], vm[5], vm[6], vm[8], vm[9], vm[10]};
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; ++i) {
    for (int k = 0; k < 3; ++k) v_means[3 * i + k] = v_scales[3 * i + k] = 0;
    for (int k = 0; k < 4; ++k) v_quats[4 * i + k] = 0;
    if (radii[i] <= 0) continue;
    pm_t o; float M[9], R[9];
    proj_mid(vm, K, means + 3 * i, quats + 4 * i, scales + 3 * i, W, H, eps2d, -1e30f, 1e30f, &o, M, R);
    /* conic = inverse(cov2d): v_cov = -inv V inv with V symmetric from (va, vb/2, vc) */
    float a = o.c11 / o.det, b = -o.c01 / o.det, c = o.c00 / o.det;
    float V[4] = {v_conics[3 * i], 0.5f * v_conics[3 * i + 1], 0.5f * v_conics[3 * i + 1], v_conics[3 * i + 2]};
    float I2[4] = {a, b, b, c}, T[4], G[4];
    for (int r = 0; r < 2; ++r) for (int s2 = 0; s2 < 2; ++s2) T[r * 2 + s2] = V[r * 2] * I2[s2] + V[r * 2 + 1] * I2[2 + s2];
    for (int r = 0; r < 2; ++r) for (int s2 = 0; s2 < 2; ++s2) G[r * 2 + s2] = -(I2[r * 2] * T[s2] + I2[r * 2 + 1] * T[2 + s2]);
    float g01 = 0.5f * (G[1] + G[2]); G[1] = G[2] = g01;
    const float* J = o.J;
    /* v_Sigma_c = J^T G J ; v_J = 2 G J Sigma_c */
    float GJ[6], vSc[9], vJ[6];
    for (int r = 0; r < 2; ++r) for (int k = 0; k < 3; ++k) GJ[r * 3 + k] = G[r * 2] * J[k] + G[r * 2 + 1] * J[3 + k];
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) vSc[r * 3 + k] = J[r] * GJ[k] + J[3 + r] * GJ[3 + k];
    for (int r = 0; r < 2; ++r) for (int k = 0; k < 3; ++k) {
      float acc = 0; for (int l = 0; l < 3; ++l) acc += GJ[r * 3 + l] * o.Sigma_c[l * 3 + k];
      vJ[r * 3 + k] = 2.f * acc;
    }
    float fx = K[0], fy = K[4];
    float x = o.p[0], y = o.p[1], z = o.p[2], rz = 1.f / z, rz2 = rz * rz, rz3 = rz2 * rz;
    float vp[3];
    vp[0] = fx * rz * v_means2d[2 * i];
    vp[1] = fy * rz * v_means2d[2 * i + 1];
    vp[2] = -(fx * x * v_means2d[2 * i] + fy * y * v_means2d[2 * i + 1]) * rz2 + (v_depths ? v_depths[i] : 0.f);
    vp[2] += -fx * rz2 * vJ[0] - fy * rz2 * vJ[4];
    if (!o.cl_x) { vp[0] += -fx * rz2 * vJ[2]; vp[2] += 2.f * fx * o.tx * rz3 * vJ[2]; }
    else vp[2] += fx * o.tx * rz3 * vJ[2];
    if (!o.cl_y) { vp[1] += -fy * rz2 * vJ[5]; vp[2] += 2.f * fy * o.ty * rz3 * vJ[5]; }
    else vp[2] += fy * o.ty * rz3 * vJ[5];
    for (int k = 0; k < 3; ++k) v_means[3 * i + k] = Rv[k] * vp[0] + Rv[3 + k] * vp[1] + Rv[6 + k] * vp[2];
    float tmp[9], vSw[9], vM[9];
    mat3_mul_at(Rv, vSc, tmp);      /* Rv^T vSc */
    mat3_mul(tmp, Rv, vSw);          /* (Rv^T vSc) Rv */
    mat3_mul(vSw, M, vM);
    for (int k = 0; k < 9; ++k) vM[k] *= 2.f;
    const float* s = scales + 3 * i;
    float vR[9];
    for (int jx = 0; jx < 3; ++jx) {
      v_scales[3 * i + jx] = R[jx] * vM[jx] + R[3 + jx] * vM[3 + jx] + R[6 + jx] * vM[6 + jx];
      for (int r = 0; r < 3; ++r) vR[r * 3 + jx] = vM[r * 3 + jx] * s[jx];
    }
    const float* q = quats + 4 * i;
    float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    float w = q[0] / n, qx = q[1] / n, qy = q[2] / n, qz = q[3] / n, vn[4];
    vn[0] = 2.f * (qx * (vR[7] - vR[5]) + qy * (vR[2] - vR[6]) + qz * (vR[3] - vR[1]));
    vn[1] = 2.f * (-2.f * qx * (vR[4] + vR[8]) + qy * (vR[1] + vR[3]) + qz * (vR[2] + vR[6]) + w * (vR[7] - vR[5]));
    vn[2] = 2.f * (qx * (vR[1] + vR[3]) - 2.f * qy * (vR[0] + vR[8]) + qz * (vR[5] + vR[7]) + w * (vR[2] - vR[6]));
    vn[3] = 2.f * (qx * (vR[2] + vR[6]) + qy * (vR[5] + vR[7]) - 2.f * qz * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
    float qn[4] = {w, qx, qy, qz};
    float dot = vn[0] * w + vn[1] * qx + vn[2] * qy + vn[3] * qz;
    for (int k = 0; k < 4; ++k) v_quats[4 * i + k] = (vn[k] - dot * qn[k]) / n;
  }
}

/* ---------------------------------------------------------- A2 spherical harmonics */
static void sh_basis_c(int deg, float x, float y, float z, float* B) {
  B[0] = 0.28209479177387814f;
  if (deg < 1) return;
  const float C1 = 0.4886025119029199f;
  B[1] = -C1 * y; B[2] = C1 * z; B[3] = -C1 * x;
  if (deg < 2) return;
  float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  B[4] = 1.0925484305920792f * xy; B[5] = -1.0925484305920792f * yz;
  B[6] = 0.31539156525252005f * (2 * zz - xx - yy); B[7] = -1.0925484305920792f * xz;
  B[8] = 0.5462742152960396f * (xx - yy);
  if (deg < 3) return;
  B[9] = -0.5900435899266435f * y * (3 * xx - yy); B[10] = 2.890611442640554f * xy * z;
  B[11] = -0.4570457994644658f * y * (4 * zz - xx - yy);
  B[12] = 0.3731763325901154f * z * (2 * zz - 3 * xx - 3 * yy);
  B[13] = -0.4570457994644658f * x * (4 * zz - xx - yy); B[14] = 1.445305721320277f * z * (xx - yy);
  B[15] = -0.5900435899266435f * x * (xx - 3 * yy);
}

void orc_sh(int n, int deg, const float* dirs, const float* coeffs, const uint8_t* masks, float* colors) {
  const int nb = (deg + 1) * (deg + 1);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    colors[3 * i] = colors[3 * i + 1] = colors[3 * i + 2] = 0;
    if (masks && !masks[i]) continue;
    float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
    float inv = 1.f / sqrtf(x * x + y * y + z * z), B[16];
    sh_basis_c(deg, x * inv, y * inv, z * inv, B);
    for (int k = 0; k < nb; ++k)
      for (int c = 0; c < 3; ++c) colors[3 * i + c] += B[k] * coeffs[(size_t)i * 48 + 3 * k + c];
  }
}

/* v_coeffs overwritten (accumulate=0) or added to; v_dirs by central differences of the basis
 * in double (independent of the product's analytic gradient). */
void orc_sh_bwd(int n, int deg, const float* dirs, const float* coeffs, const uint8_t* masks,
                const float* v_colors, float* v_coeffs, int accumulate, float* v_dirs) {
  const int nb = (deg + 1) * (deg + 1);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    if (!accumulate) for (int k = 0; k < 48; ++k) v_coeffs[(size_t)i * 48 + k] = 0;
    if (v_dirs) v_dirs[3 * i] = v_dirs[3 * i + 1] = v_dirs[3 * i + 2] = 0;
    if (masks && !masks[i]) continue;
    float d[3] = {dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
    float inv = 1.f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), B[16];
    sh_basis_c(deg, d[0] * inv, d[1] * inv, d[2] * inv, B);
    for (int k = 0; k < nb; ++k)
      for (int c = 0; c < 3; ++c) v_coeffs[(size_t)i * 48 + 3 * k + c] += B[k] * v_colors[3 * i + c];
    if (v_dirs && deg > 0) {
      for (int ax = 0; ax < 3; ++ax) {
        double h = 1e-3 * sqrt((double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2]);
        double f[2];
        for (int sgn = 0; sgn < 2; ++sgn) {
          double e[3] = {d[0], d[1], d[2]};
          e[ax] += sgn ? h : -h;
          double in2 = 1.0 / sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
          float Bb[16];
          sh_basis_c(deg, (float)(e[0] * in2), (float)(e[1] * in2), (float)(e[2] * in2), Bb);
          double acc = 0;
          for (int k = 0; k < nb; ++k)
            for (int c = 0; c < 3; ++c) acc += (double)Bb[k] * coeffs[(size_t)i * 48 + 3 * k + c] * v_colors[3 * i + c];
          f[sgn] = acc;
        }
        v_dirs[3 * i + ax] = (float)((f[1] - f[0]) / (2 * h));
      }
    }
  }
}

/* -------------------------------------------------------------- A3/A4 tile binning */
typedef struct { uint64_t key; int32_t val; } kv_t;

static void tile_box(float mx, float my, float r, int tw, int th, int* b) {
  float tr = r / 16.f, tx = mx / 16.f, ty = my / 16.f;
  b[0] = (int)fminf(fmaxf(floorf(tx - tr), 0.f), (float)tw);
  b[1] = (int)fminf(fmaxf(floorf(ty - tr), 0.f), (float)th);
  b[2] = (int)fminf(fmaxf(ceilf(tx + tr), 0.f), (float)tw);
  b[3] = (int)fminf(fmaxf(ceilf(ty + tr), 0.f), (float)th);
}

static void radix_sort_kv(kv_t* a, kv_t* tmp, int64_t n, int bits) {
  /* stable LSD radix sort, 8-bit digits */
  for (int sh = 0; sh < bits; sh += 8) {
    int64_t cnt[257];
    memset(cnt, 0, sizeof(cnt));
    for (int64_t i = 0; i < n; ++i) cnt[((a[i].key >> sh) & 0xFF) + 1]++;
    for (int k = 0; k < 256; ++k) cnt[k + 1] += cnt[k];
    for (int64_t i = 0; i < n; ++i) tmp[cnt[(a[i].key >> sh) & 0xFF]++] = a[i];
    kv_t* t = a; a = tmp; tmp = t;
  }
  if (((bits + 7) / 8) & 1) memcpy(tmp, a, sizeof(kv_t) * (size_t)n); /* result back into caller's a */
}

/* Returns the number of intersections; if isect_ids == NULL only counts. */
int64_t orc_isect(int N, const float* means2d, const int32_t* radii, const float* depths, int tw,
                  int th, int64_t* isect_ids, int32_t* flatten_ids, int32_t* offsets,
                  int32_t* tiles_per_gauss) {
  int64_t total = 0;
  for (int i = 0; i < N; ++i) {
    int c = 0;
    if (radii[i] > 0) { int b[4]; tile_box(means2d[2 * i], means2d[2 * i + 1], (float)radii[i], tw, th, b); c = (b[2] - b[0]) * (b[3] - b[1]); }
    if (tiles_per_gauss) tiles_per_gauss[i] = c;
    total += c;
  }
  if (!isect_ids) return total;
  int nt = tw * th, tile_bits = 0;
  { unsigned v = (unsigned)nt; while (v >>= 1) ++tile_bits; ++tile_bits; }
  kv_t* a = (kv_t*)malloc(sizeof(kv_t) * (size_t)(total + 1));
  kv_t* t = (kv_t*)malloc(sizeof(kv_t) * (size_t)(total + 1));
  int64_t cur = 0;
  for (int i = 0; i < N; ++i) {
    if (radii[i] <= 0) continue;
    int b[4]; tile_box(means2d[2 * i], means2d[2 * i + 1], (float)radii[i], tw, th, b);
    uint32_t db; memcpy(&db, depths + i, 4);
    for (int y = b[1]; y < b[3]; ++y)
      for (int x = b[0]; x < b[2]; ++x) { a[cur].key = ((uint64_t)(y * tw + x) << 32) | db; a[cur].val = i; ++cur; }
  }
  radix_sort_kv(a, t, total, 32 + tile_bits + 1);
  for (int64_t i = 0; i < total; ++i) { isect_ids[i] = (int64_t)a[i].key; flatten_ids[i] = a[i].val; }
  /* offsets[t] = first index with tile >= t */
  int64_t j = 0;
  for (int tIdx = 0; tIdx < nt; ++tIdx) {
    while (j < total && (int)(a[j].key >> 32) < tIdx) ++j;
    offsets[tIdx] = (int32_t)j;
  }
  free(a); free(t);
  return total;
}

/* ------------------------------------------------------------------ A5 rasterize */
void orc_rasterize(int width, int height, int tw, int th, int64_t n_isects, const float* means2d,
                   const float* conics, const float* colors, const float* opac, const float* bg,
                   const int32_t* offsets, const int32_t* fids, float* out, float* alpha, int32_t* last) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < tw * th; ++tile) {
    int ty = tile / tw, tx = tile % tw;
    int s = offsets[tile], e = (tile == tw * th - 1) ? (int)n_isects : offsets[tile + 1];
    for (int py = ty * 16; py < ty * 16 + 16 && py < height; ++py)
      for (int px = tx * 16; px < tx * 16 + 16 && px < width; ++px) {
        float fx = px + 0.5f, fy = py + 0.5f, T = 1.f, c[3] = {0, 0, 0};
        int li = 0;
        for (int k = s; k < e; ++k) {
          int g = fids[k];
          float dx = means2d[2 * g] - fx, dy = means2d[2 * g + 1] - fy;
          float sg = 0.5f * (conics[3 * g] * dx * dx + conics[3 * g + 2] * dy * dy) + conics[3 * g + 1] * dx * dy;
          float al = fminf(0.999f, opac[g] * expf(-sg));
          if (sg < 0.f || al < 1.f / 255.f) continue;
          float nT = T * (1.f - al);
          if (nT <= 1e-4f) break;
          float vis = al * T;
          c[0] += colors[3 * g] * vis; c[1] += colors[3 * g + 1] * vis; c[2] += colors[3 * g + 2] * vis;
          li = k; T = nT;
        }
        size_t p = (size_t)py * width + px;
        for (int k = 0; k < 3; ++k) out[3 * p + k] = c[k] + (bg ? T * bg[k] : 0.f);
        alpha[p] = 1.f - T; last[p] = li;
      }
  }
}

/* A6: per-pixel back-to-front.  Each tile sums its pixels' contributions per list entry in a
 * private double buffer, then adds them to a shared double accumulator with omp atomics (order
 * effects are below fp32 resolution).  v_* outputs are overwritten. */
void orc_rasterize_bwd(int N, int width, int height, int tw, int th, int64_t n_isects,
                       const float* means2d, const float* conics, const float* colors,
                       const float* opac, const float* bg, const int32_t* offsets,
                       const int32_t* fids, const float* alpha, const int32_t* last,
                       const float* v_out, const float* v_alpha, float* v_means2d, float* v_conics,
                       float* v_colors, float* v_opac) {
  double* acc = (double*)calloc((size_t)N * 9 + 9, sizeof(double));
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < tw * th; ++tile) {
    int ty = tile / tw, tx = tile % tw;
    int s = offsets[tile], e = (tile == tw * th - 1) ? (int)n_isects : offsets[tile + 1];
    if (e <= s) continue;
    double* loc = (double*)calloc((size_t)(e - s) * 9, sizeof(double));
    for (int py = ty * 16; py < ty * 16 + 16 && py < height; ++py)
      for (int px = tx * 16; px < tx * 16 + 16 && px < width; ++px) {
        size_t p = (size_t)py * width + px;
        float fx = px + 0.5f, fy = py + 0.5f;
        float Tf = 1.f - alpha[p], T = Tf, buf[3] = {0, 0, 0};
        const float* vo = v_out + 3 * p;
        float va = v_alpha ? v_alpha[p] : 0.f;
        if (bg) va -= bg[0] * vo[0] + bg[1] * vo[1] + bg[2] * vo[2];
        for (int k = last[p]; k >= s; --k) {
          int g = fids[k];
          float dx = means2d[2 * g] - fx, dy = means2d[2 * g + 1] - fy;
          float ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
          float sg = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
          float ex = expf(-sg), al = fminf(0.999f, opac[g] * ex);
          if (sg < 0.f || al < 1.f / 255.f) continue;
          float ra = 1.f / (1.f - al);
          T *= ra;
          float fac = al * T;
          double* a = loc + (size_t)(k - s) * 9;
          float v_al = 0;
          for (int c = 0; c < 3; ++c) {
            a[5 + c] += fac * vo[c];
            v_al += (colors[3 * g + c] * T - buf[c] * ra) * vo[c];
          }
          v_al += Tf * ra * va;
          if (opac[g] * ex <= 0.999f) {
            float vs = -opac[g] * ex * v_al;
            a[2] += 0.5f * vs * dx * dx; a[3] += vs * dx * dy; a[4] += 0.5f * vs * dy * dy;
            a[0] += vs * (ca * dx + cb * dy); a[1] += vs * (cb * dx + cc * dy);
            a[8] += ex * v_al;
          }
          for (int c = 0; c < 3; ++c) buf[c] += colors[3 * g + c] * fac;
        }
      }
    for (int k = s; k < e; ++k) {
      const double* a = loc + (size_t)(k - s) * 9;
      double* d = acc + (size_t)fids[k] * 9;
      for (int c = 0; c < 9; ++c)
        if (a[c] != 0.0) {
#pragma omp atomic
          d[c] += a[c];
        }
    }
    free(loc);
  }
#pragma omp parallel for schedule(static)
  for (int g = 0; g < N; ++g) {
    const double* a = acc + (size_t)g * 9;
    v_means2d[2 * g] = (float)a[0]; v_means2d[2 * g + 1] = (float)a[1];
    v_conics[3 * g] = (float)a[2]; v_conics[3 * g + 1] = (float)a[3]; v_conics[3 * g + 2] = (float)a[4];
    v_colors[3 * g] = (float)a[5]; v_colors[3 * g + 1] = (float)a[6]; v_colors[3 * g + 2] = (float)a[7];
    v_opac[g] = (float)a[8];
  }
  free(acc);
}

/* --------------------------------------------------------------- A7 loss: L1 + SSIM */
static void ssim_window(float* w) {
  double g[11], s = 0;
  for (int i = 0; i < 11; ++i) { g[i] = exp(-((i - 5) * (i - 5)) / (2 * 1.5 * 1.5)); s += g[i]; }
  for (int i = 0; i < 11; ++i) w[i] = (float)(g[i] / s);
}

static void conv11(const float* in, float* out, float* tmp, int H, int W, const float* w) {
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float a = 0;
      for (int k = -5; k <= 5; ++k) { int xx = x + k; if (xx >= 0 && xx < W) a += w[k + 5] * in[(size_t)y * W + xx]; }
      tmp[(size_t)y * W + x] = a;
    }
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float a = 0;
      for (int k = -5; k <= 5; ++k) { int yy = y + k; if (yy >= 0 && yy < H) a += w[k + 5] * tmp[(size_t)yy * W + x]; }
      out[(size_t)y * W + x] = a;
    }
}

/* img [3,H,W] float, gt [3,H,W] u8.  Returns loss = 0.8 L1 + 0.2 (1 - SSIM); if v_img != NULL
 * also writes d loss / d img. */
float orc_loss(int H, int W, const float* img, const uint8_t* gt_u8, float* v_img) {
  const size_t P = (size_t)H * W;
  float w[11];
  ssim_window(w);
  float* gt = (float*)malloc(sizeof(float) * 3 * P);
  float* buf = (float*)malloc(sizeof(float) * 12 * P);
  double l1 = 0, ssim = 0;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  const double numel = 3.0 * (double)P;
  for (int c = 0; c < 3; ++c) {
    const float* x = img + c * P;
    float* y = gt + c * P;
    for (size_t i = 0; i < P; ++i) { float v = gt_u8[c * P + i] / 255.0f; y[i] = v < 0 ? 0 : (v > 1 ? 1 : v); }
    float *mu1 = buf, *mu2 = buf + P, *e11 = buf + 2 * P, *e22 = buf + 3 * P, *e12 = buf + 4 * P,
          *t0 = buf + 5 * P, *t1 = buf + 6 * P, *M1 = buf + 7 * P, *M2 = buf + 8 * P, *M3 = buf + 9 * P,
          *c1 = buf + 10 * P, *c2 = buf + 11 * P;
    conv11(x, mu1, t1, H, W, w); conv11(y, mu2, t1, H, W, w);
    for (size_t i = 0; i < P; ++i) t0[i] = x[i] * x[i];
    conv11(t0, e11, t1, H, W, w);
    for (size_t i = 0; i < P; ++i) t0[i] = y[i] * y[i];
    conv11(t0, e22, t1, H, W, w);
    for (size_t i = 0; i < P; ++i) t0[i] = x[i] * y[i];
    conv11(t0, e12, t1, H, W, w);
    for (size_t i = 0; i < P; ++i) {
      float m1 = mu1[i], m2 = mu2[i];
      float s1 = e11[i] - m1 * m1, s2 = e22[i] - m2 * m2, s12 = e12[i] - m1 * m2;
      float A = 2 * m1 * m2 + C1, B = 2 * s12 + C2, D = m1 * m1 + m2 * m2 + C1, E = s1 + s2 + C2;
      float val = A * B / (D * E);
      ssim += val;
      l1 += fabsf(x[i] - y[i]);
      float d_mu1 = 2 * m2 * B / (D * E) - val * 2 * m1 / D, d_s1 = -val / E, d_s12 = 2 * A / (D * E);
      M1[i] = d_mu1 - 2 * m1 * d_s1 - m2 * d_s12; M2[i] = d_s1; M3[i] = d_s12;
    }
    if (v_img) {
      conv11(M1, t0, t1, H, W, w); conv11(M2, c1, t1, H, W, w); conv11(M3, c2, t1, H, W, w);
      for (size_t i = 0; i < P; ++i) {
        float ds = t0[i] + 2 * x[i] * c1[i] + y[i] * c2[i];
        float sgn = (x[i] > y[i]) ? 1.f : ((x[i] < y[i]) ? -1.f : 0.f);
        v_img[c * P + i] = (float)(0.8 * sgn / numel - 0.2 * ds / numel);
      }
    }
  }
  free(gt); free(buf);
  return (float)(0.8 * l1 / numel + 0.2 * (1.0 - ssim / numel));
}
