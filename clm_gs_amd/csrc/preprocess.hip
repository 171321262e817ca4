// Fused per-camera front end and its VJP (gfx950): for the rows of one visibility filter,
//   gather raw attributes -> activations (exp / sigmoid; the projection normalises the quaternion)
//   -> EWA projection -> SH colour -> +0.5 clamp -> packed 64 B raster record
// and, backward, from the rasterizer's packed gradient lines,
//   clamp / SH / projection / activation VJPs -> accumulate into the FULL-size gradient tensors
//   and the SH gradient rows -> densification statistics.
//
// It computes exactly what the op-by-op path computes between the row gather and the rasterizer
// (strategies/clm_offload/engine.py:650-691 forward, :703-742 + densification.py:59-102 backward)
// but in 2 kernels instead of ~35 small ones (4 gathers, 3 activations, projection, dirs, SH,
// clamp, pack | unpack, clamp', SH', dirs', projection', activation', 4 index_add, stats).
// HBM-bound: 236 B of parameters in + ~100 B of per-row outputs per visible Gaussian.
#include "common.h"
#include "gs_math.h"

namespace clmgs {

constexpr int PP_ROWS = 256;
constexpr int PP_PITCH = 52;  // floats per LDS row (see sh.hip)

struct PreArgs {
  const int64_t* filter;        // [V] row ids, or NULL (identity)
  const float* xyz;             // [N,3]
  const float* opacity_raw;     // [N]
  const float* scaling_raw;     // [N,3]
  const float* rotation_raw;    // [N,4]
  const float* sh_rows;         // [N,48] indexed by row id, or [V,48] indexed by i (sh_by_filter=0)
  int sh_by_filter;
  float viewmat[16];
  float K[9];
  float campos[3];
  int width, height, degree;
  float eps2d, near_plane, far_plane, radius_clip;
};

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ void __launch_bounds__(PP_ROWS)
preprocess_fwd_kernel(int V, PreArgs a, int32_t* __restrict__ radii, float* __restrict__ means2d,
                      float* __restrict__ depths, float* __restrict__ conics,
                      float* __restrict__ colors, float* __restrict__ opacities,
                      float4* __restrict__ packed) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int64_t rowid[PP_ROWS];
  const Cam cam = load_cam(a.viewmat, a.K);
  const int nb = (a.degree + 1) * (a.degree + 1);
  const int nf4 = (nb * 3 + 3) / 4;
  const int n_chunks = (V + PP_ROWS - 1) / PP_ROWS;
  for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const int base = chunk * PP_ROWS;
    const int rows = min(PP_ROWS, V - base);
    __syncthreads();
    if (threadIdx.x < rows) rowid[threadIdx.x] = a.filter ? a.filter[base + threadIdx.x] : (int64_t)(base + threadIdx.x);
    __syncthreads();
    // stage the SH rows of this chunk (row-granular gather, 16 B per lane, coalesced within rows)
    for (int i = threadIdx.x; i < rows * nf4; i += PP_ROWS) {
      const int r = i / nf4, k = i - r * nf4;
      const int64_t src = a.sh_by_filter ? rowid[r] : (int64_t)(base + r);
      *reinterpret_cast<float4*>(lds + r * PP_PITCH + 4 * k) =
          *reinterpret_cast<const float4*>(a.sh_rows + src * 48 + 4 * k);
    }
    __syncthreads();
    if (threadIdx.x < rows) {
      const int i = base + threadIdx.x;
      const int64_t g = rowid[threadIdx.x];
      const float m[3] = {a.xyz[3 * g], a.xyz[3 * g + 1], a.xyz[3 * g + 2]};
      const float4 q4 = *reinterpret_cast<const float4*>(a.rotation_raw + 4 * g);
      const float q[4] = {q4.x, q4.y, q4.z, q4.w};
      const float s[3] = {__expf(a.scaling_raw[3 * g]), __expf(a.scaling_raw[3 * g + 1]), __expf(a.scaling_raw[3 * g + 2])};
      const float o = sigmoidf(a.opacity_raw[g]);
      const Proj p = project_fwd(cam, m, q, s, (float)a.width, (float)a.height, a.eps2d, a.near_plane,
                                 a.far_plane, a.radius_clip);
      float cr = 0.f, cg = 0.f, cb = 0.f;
      if (p.radius > 0) {
        float x = m[0] - a.campos[0], y = m[1] - a.campos[1], z = m[2] - a.campos[2];
        const float inv = 1.0f / sqrtf(x * x + y * y + z * z);
        x *= inv; y *= inv; z *= inv;
        float B[16];
        sh_basis(a.degree, x, y, z, B);
        const float* row = lds + threadIdx.x * PP_PITCH;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (k < nb) { cr += B[k] * row[3 * k]; cg += B[k] * row[3 * k + 1]; cb += B[k] * row[3 * k + 2]; }
        }
        cr = fmaxf(cr + 0.5f, 0.f); cg = fmaxf(cg + 0.5f, 0.f); cb = fmaxf(cb + 0.5f, 0.f);
      }
      radii[i] = p.radius;
      *reinterpret_cast<float2*>(means2d + 2 * (size_t)i) = make_float2(p.mx, p.my);
      depths[i] = p.depth;
      conics[3 * (size_t)i] = p.ca; conics[3 * (size_t)i + 1] = p.cb; conics[3 * (size_t)i + 2] = p.cc;
      colors[3 * (size_t)i] = cr; colors[3 * (size_t)i + 1] = cg; colors[3 * (size_t)i + 2] = cb;
      opacities[i] = o;
      float4* rec = packed + 4 * (size_t)i;
      rec[0] = make_float4(p.mx, p.my, o, p.ca);
      rec[1] = make_float4(p.cb, p.cc, cr, cg);
      rec[2] = make_float4(cb, 0.f, 0.f, 0.f);
    }
  }
}

struct PreGrads {
  float* g_xyz;          // [N,3]   accumulated at row ids
  float* g_opacity;      // [N]
  float* g_scaling;      // [N,3]
  float* g_rotation;     // [N,4]
  float* g_sh_rows;      // [N,48] (sh_by_filter) or [V,48]: accumulated
  float* max_radii2D;    // [N] or NULL (no statistics)
  float* grad_accum;     // [N]
  float* denom;          // [N]
  float* v_means2d_out;  // [V,2] or NULL: copy of the screen-space gradient (API parity)
  int stats_only_visible;  // 1: statistics only for rows with radius > 0 (no_offload semantics)
};

__global__ void __launch_bounds__(PP_ROWS)
preprocess_bwd_kernel(int V, PreArgs a, const int32_t* __restrict__ radii,
                      const float4* __restrict__ packed_grad, PreGrads o) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int64_t rowid[PP_ROWS];
  __shared__ uint8_t live[PP_ROWS];
  const Cam cam = load_cam(a.viewmat, a.K);
  const int nb = (a.degree + 1) * (a.degree + 1);
  const int nf4 = (nb * 3 + 3) / 4;
  const int n_chunks = (V + PP_ROWS - 1) / PP_ROWS;
  for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const int base = chunk * PP_ROWS;
    const int rows = min(PP_ROWS, V - base);
    __syncthreads();
    if (threadIdx.x < rows) {
      rowid[threadIdx.x] = a.filter ? a.filter[base + threadIdx.x] : (int64_t)(base + threadIdx.x);
      live[threadIdx.x] = radii[base + threadIdx.x] > 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * nf4; i += PP_ROWS) {
      const int r = i / nf4, k = i - r * nf4;
      if (!live[r]) continue;
      const int64_t src = a.sh_by_filter ? rowid[r] : (int64_t)(base + r);
      *reinterpret_cast<float4*>(lds + r * PP_PITCH + 4 * k) =
          *reinterpret_cast<const float4*>(a.sh_rows + src * 48 + 4 * k);
    }
    __syncthreads();
    if (threadIdx.x < rows) {
      const int i = base + threadIdx.x;
      const int64_t g = rowid[threadIdx.x];
      const int radius = radii[i];
      const float4 ga = packed_grad[4 * (size_t)i], gb = packed_grad[4 * (size_t)i + 1];
      const float go = packed_grad[4 * (size_t)i + 2].x;
      // line: x y ca cb | cc r g b | o
      const float v_m2[2] = {ga.x, ga.y};
      if (o.v_means2d_out) *reinterpret_cast<float2*>(o.v_means2d_out + 2 * (size_t)i) = make_float2(ga.x, ga.y);
      if (o.max_radii2D && (radius > 0 || !o.stats_only_visible)) {  // default: every filter row,
        // as gsplat_add_densification_stats_exact_filter; only_visible = the no_offload mask form
        const float gx = v_m2[0] * (0.5f * a.width), gy = v_m2[1] * (0.5f * a.height);
        o.max_radii2D[g] = fmaxf(o.max_radii2D[g], (float)radius);
        o.grad_accum[g] += sqrtf(gx * gx + gy * gy);
        o.denom[g] += 1.f;
      }
      float* row = lds + threadIdx.x * PP_PITCH;
      if (radius > 0) {
        const float m[3] = {a.xyz[3 * g], a.xyz[3 * g + 1], a.xyz[3 * g + 2]};
        const float4 q4 = *reinterpret_cast<const float4*>(a.rotation_raw + 4 * g);
        const float q[4] = {q4.x, q4.y, q4.z, q4.w};
        const float s[3] = {__expf(a.scaling_raw[3 * g]), __expf(a.scaling_raw[3 * g + 1]), __expf(a.scaling_raw[3 * g + 2])};
        const float op = sigmoidf(a.opacity_raw[g]);
        // ---- SH: recompute the pre-clamp colour for the clamp mask, then the VJP
        float dx = m[0] - a.campos[0], dy = m[1] - a.campos[1], dz = m[2] - a.campos[2];
        const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        const float x = dx * inv, y = dy * inv, z = dz * inv;
        float B[16];
        sh_basis(a.degree, x, y, z, B);
        float pr = 0.f, pg = 0.f, pb = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (k < nb) { pr += B[k] * row[3 * k]; pg += B[k] * row[3 * k + 1]; pb += B[k] * row[3 * k + 2]; }
        }
        const float vc[3] = {(pr + 0.5f > 0.f) ? gb.y : 0.f, (pg + 0.5f > 0.f) ? gb.z : 0.f,
                             (pb + 0.5f > 0.f) ? gb.w : 0.f};
        float vdx = 0.f, vdy = 0.f, vdz = 0.f;
        if (a.degree > 0) {
          float Bx[16], By[16], Bz[16];
          sh_basis_grad(a.degree, x, y, z, Bx, By, Bz);
          float ux = 0.f, uy = 0.f, uz = 0.f;
#pragma unroll
          for (int k = 1; k < 16; ++k) {
            if (k < nb) {
              const float vB = row[3 * k] * vc[0] + row[3 * k + 1] * vc[1] + row[3 * k + 2] * vc[2];
              ux += vB * Bx[k]; uy += vB * By[k]; uz += vB * Bz[k];
            }
          }
          const float dot = ux * x + uy * y + uz * z;
          vdx = (ux - dot * x) * inv; vdy = (uy - dot * y) * inv; vdz = (uz - dot * z) * inv;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {  // this lane's LDS row becomes its SH gradient row
          const float bk = (k < nb) ? B[k] : 0.f;
          row[3 * k] = bk * vc[0]; row[3 * k + 1] = bk * vc[1]; row[3 * k + 2] = bk * vc[2];
        }
        // ---- projection VJP (quaternion normalisation is inside; scale through exp)
        const float v_con[3] = {ga.z, ga.w, gb.x};
        float vm[3], vq[4], vs[3];
        project_bwd(cam, m, q, s, (float)a.width, (float)a.height, a.eps2d, v_m2, 0.f, v_con, vm, vq, vs);
        o.g_xyz[3 * g] += vm[0] + vdx; o.g_xyz[3 * g + 1] += vm[1] + vdy; o.g_xyz[3 * g + 2] += vm[2] + vdz;
        o.g_scaling[3 * g] += vs[0] * s[0]; o.g_scaling[3 * g + 1] += vs[1] * s[1]; o.g_scaling[3 * g + 2] += vs[2] * s[2];
        float4* gq = reinterpret_cast<float4*>(o.g_rotation + 4 * g);
        float4 cur = *gq;
        cur.x += vq[0]; cur.y += vq[1]; cur.z += vq[2]; cur.w += vq[3];
        *gq = cur;
        o.g_opacity[g] += go * op * (1.f - op);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * nf4; i += PP_ROWS) {
      const int r = i / nf4, k = i - r * nf4;
      if (!live[r]) continue;
      const int64_t dst_row = a.sh_by_filter ? rowid[r] : (int64_t)(base + r);
      float4* dst = reinterpret_cast<float4*>(o.g_sh_rows + dst_row * 48 + 4 * k);
      const float4 v = *reinterpret_cast<const float4*>(lds + r * PP_PITCH + 4 * k);
      float4 c = *dst;
      c.x += v.x; c.y += v.y; c.z += v.z; c.w += v.w;
      *dst = c;
    }
  }
}

}  // namespace clmgs

using namespace clmgs;

static void fill_args(PreArgs& a, const int64_t* filter, const float* xyz, const float* opacity_raw,
                      const float* scaling_raw, const float* rotation_raw, const float* sh_rows,
                      int sh_by_filter, const float* viewmat, const float* K, const float* campos,
                      int width, int height, int degree, float eps2d, float near_plane,
                      float far_plane, float radius_clip) {
  a.filter = filter; a.xyz = xyz; a.opacity_raw = opacity_raw; a.scaling_raw = scaling_raw;
  a.rotation_raw = rotation_raw; a.sh_rows = sh_rows; a.sh_by_filter = sh_by_filter;
  for (int i = 0; i < 16; ++i) a.viewmat[i] = viewmat[i];
  for (int i = 0; i < 9; ++i) a.K[i] = K[i];
  for (int i = 0; i < 3; ++i) a.campos[i] = campos[i];
  a.width = width; a.height = height; a.degree = degree;
  a.eps2d = eps2d; a.near_plane = near_plane; a.far_plane = far_plane; a.radius_clip = radius_clip;
}

extern "C" int clmgs_preprocess_fwd(void* stream, int V, const int64_t* filter, const float* xyz,
                                    const float* opacity_raw, const float* scaling_raw,
                                    const float* rotation_raw, const float* sh_rows,
                                    int sh_by_filter, const float* viewmat_host,
                                    const float* K_host, const float* campos_host, int width,
                                    int height, int degree, float eps2d, float near_plane,
                                    float far_plane, float radius_clip, int32_t* radii,
                                    float* means2d, float* depths, float* conics, float* colors,
                                    float* opacities, void* packed) {
  CLMGS_CHECK_ARG(V >= 0 && degree >= 0 && degree <= 3 && width > 0 && height > 0);
  if (V == 0) return 0;
  CLMGS_CHECK_ARG(xyz && opacity_raw && scaling_raw && rotation_raw && sh_rows && viewmat_host &&
                  K_host && campos_host && radii && means2d && depths && conics && colors &&
                  opacities && packed);
  PreArgs a;
  fill_args(a, filter, xyz, opacity_raw, scaling_raw, rotation_raw, sh_rows, sh_by_filter,
            viewmat_host, K_host, campos_host, width, height, degree, eps2d, near_plane, far_plane,
            radius_clip);
  const size_t lds = PP_ROWS * PP_PITCH * sizeof(float);
  const int grid = min(ceil_div(V, PP_ROWS), 256 * 3);
  hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(grid), dim3(PP_ROWS), lds, (hipStream_t)stream, V, a,
                     radii, means2d, depths, conics, colors, opacities, (float4*)packed);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_preprocess_bwd(void* stream, int V, const int64_t* filter, const float* xyz,
                                    const float* opacity_raw, const float* scaling_raw,
                                    const float* rotation_raw, const float* sh_rows,
                                    int sh_by_filter, const float* viewmat_host,
                                    const float* K_host, const float* campos_host, int width,
                                    int height, int degree, float eps2d, const int32_t* radii,
                                    const void* packed_grad, float* g_xyz, float* g_opacity,
                                    float* g_scaling, float* g_rotation, float* g_sh_rows,
                                    float* max_radii2D, float* grad_accum, float* denom,
                                    float* v_means2d_out, int stats_only_visible) {
  CLMGS_CHECK_ARG(V >= 0 && degree >= 0 && degree <= 3 && width > 0 && height > 0);
  if (V == 0) return 0;
  CLMGS_CHECK_ARG(xyz && opacity_raw && scaling_raw && rotation_raw && sh_rows && viewmat_host &&
                  K_host && campos_host && radii && packed_grad && g_xyz && g_opacity &&
                  g_scaling && g_rotation && g_sh_rows);
  CLMGS_CHECK_ARG(!max_radii2D || (grad_accum && denom));
  PreArgs a;
  fill_args(a, filter, xyz, opacity_raw, scaling_raw, rotation_raw, sh_rows, sh_by_filter,
            viewmat_host, K_host, campos_host, width, height, degree, eps2d, 0.f, 0.f, 0.f);
  PreGrads o{g_xyz, g_opacity, g_scaling, g_rotation, g_sh_rows, max_radii2D, grad_accum, denom,
             v_means2d_out, stats_only_visible};
  const size_t lds = PP_ROWS * PP_PITCH * sizeof(float);
  const int grid = min(ceil_div(V, PP_ROWS), 256 * 3);
  hipLaunchKernelGGL(preprocess_bwd_kernel, dim3(grid), dim3(PP_ROWS), lds, (hipStream_t)stream, V, a,
                     radii, (const float4*)packed_grad, o);
  CLMGS_LAUNCH_CHECK();
  return 0;
}
