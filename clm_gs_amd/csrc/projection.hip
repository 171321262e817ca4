// EWA projection of 3D Gaussians, forward and VJP (gfx950).
// Replaces gsplat.fully_fused_projection as called at
// strategies/base_engine.py:36-47,139-151, strategies/no_offload/engine.py:49-60,
// strategies/clm_offload/engine.py:51-63.
//
// HBM-bound streaming kernel: 40 B read + 28 B written per (camera, Gaussian)
// forward; one thread per pair, camera constants in SGPRs (uniform loads).
#include "common.h"
#include "gs_math.h"

namespace clmgs {

__global__ void __launch_bounds__(256)
projection_fwd_kernel(int C, int N, const float* __restrict__ means, const float* __restrict__ quats,
                      const float* __restrict__ scales, const float* __restrict__ viewmats,
                      const float* __restrict__ Ks, float W, float H, float eps2d, float near_plane,
                      float far_plane, float radius_clip, int32_t* __restrict__ radii,
                      float* __restrict__ means2d, float* __restrict__ depths,
                      float* __restrict__ conics) {
  const int c = blockIdx.y;
  const Cam cam = load_cam(viewmats + 16 * c, Ks + 9 * c);
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    float m[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
    const float4 q4 = *reinterpret_cast<const float4*>(quats + 4 * n);
    float q[4] = {q4.x, q4.y, q4.z, q4.w};
    float s[3] = {scales[3 * n], scales[3 * n + 1], scales[3 * n + 2]};
    Proj p = project_fwd(cam, m, q, s, W, H, eps2d, near_plane, far_plane, radius_clip);
    const size_t o = (size_t)c * N + n;
    radii[o] = p.radius;
    if (means2d) *reinterpret_cast<float2*>(means2d + 2 * o) = make_float2(p.mx, p.my);
    if (depths) depths[o] = p.depth;
    if (conics) {
      conics[3 * o] = p.ca; conics[3 * o + 1] = p.cb; conics[3 * o + 2] = p.cc;
    }
  }
}

// Visibility-only cull straight from the RAW parameters (log-scales; the quaternion is
// normalised inside project_fwd anyway): what calculate_filters needs, without first
// materialising exp / normalize / sigmoid over all N Gaussians every batch.
__global__ void __launch_bounds__(256)
visibility_raw_kernel(int C, int N, const float* __restrict__ means, const float* __restrict__ quats_raw,
                      const float* __restrict__ log_scales, const float* __restrict__ viewmats,
                      const float* __restrict__ Ks, float W, float H, float eps2d, float near_plane,
                      float far_plane, float radius_clip, int32_t* __restrict__ radii) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const float m[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
    const float4 q4 = *reinterpret_cast<const float4*>(quats_raw + 4 * n);
    const float q[4] = {q4.x, q4.y, q4.z, q4.w};
    const float s[3] = {__expf(log_scales[3 * n]), __expf(log_scales[3 * n + 1]), __expf(log_scales[3 * n + 2])};
    for (int c = 0; c < C; ++c) {  // parameters are read once for all cameras of the batch
      const Cam cam = load_cam(viewmats + 16 * c, Ks + 9 * c);
      const Proj p = project_fwd(cam, m, q, s, W, H, eps2d, near_plane, far_plane, radius_clip);
      radii[(size_t)c * N + n] = p.radius;
    }
  }
}

__global__ void __launch_bounds__(256)
projection_bwd_kernel(int C, int N, const float* __restrict__ means, const float* __restrict__ quats,
                      const float* __restrict__ scales, const float* __restrict__ viewmats,
                      const float* __restrict__ Ks, float W, float H, float eps2d,
                      const int32_t* __restrict__ radii, const float* __restrict__ v_means2d,
                      const float* __restrict__ v_depths, const float* __restrict__ v_conics,
                      float* __restrict__ v_means, float* __restrict__ v_quats,
                      float* __restrict__ v_scales) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    float am[3] = {0.f, 0.f, 0.f}, aq[4] = {0.f, 0.f, 0.f, 0.f}, as[3] = {0.f, 0.f, 0.f};
    bool loaded = false;
    float m[3], q[4], s[3];
    for (int c = 0; c < C; ++c) {
      const size_t o = (size_t)c * N + n;
      if (radii[o] <= 0) continue;
      if (!loaded) {
        m[0] = means[3 * n]; m[1] = means[3 * n + 1]; m[2] = means[3 * n + 2];
        const float4 q4 = *reinterpret_cast<const float4*>(quats + 4 * n);
        q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
        s[0] = scales[3 * n]; s[1] = scales[3 * n + 1]; s[2] = scales[3 * n + 2];
        loaded = true;
      }
      const Cam cam = load_cam(viewmats + 16 * c, Ks + 9 * c);
      float vm2[2] = {v_means2d[2 * o], v_means2d[2 * o + 1]};
      float vc[3] = {v_conics[3 * o], v_conics[3 * o + 1], v_conics[3 * o + 2]};
      float vd = v_depths ? v_depths[o] : 0.f;
      float vm[3], vq[4], vs[3];
      project_bwd(cam, m, q, s, W, H, eps2d, vm2, vd, vc, vm, vq, vs);
      for (int k = 0; k < 3; ++k) { am[k] += vm[k]; as[k] += vs[k]; }
      for (int k = 0; k < 4; ++k) aq[k] += vq[k];
    }
    v_means[3 * n] = am[0]; v_means[3 * n + 1] = am[1]; v_means[3 * n + 2] = am[2];
    *reinterpret_cast<float4*>(v_quats + 4 * n) = make_float4(aq[0], aq[1], aq[2], aq[3]);
    v_scales[3 * n] = as[0]; v_scales[3 * n + 1] = as[1]; v_scales[3 * n + 2] = as[2];
  }
}

}  // namespace clmgs

using namespace clmgs;

extern "C" int clmgs_projection_fwd(void* stream, int C, int N, const float* means,
                                    const float* quats, const float* scales,
                                    const float* viewmats, const float* Ks, int width, int height,
                                    float eps2d, float near_plane, float far_plane,
                                    float radius_clip, int32_t* radii, float* means2d,
                                    float* depths, float* conics) {
  CLMGS_CHECK_ARG(C >= 1 && N >= 0 && width > 0 && height > 0);
  if (N == 0) return 0;
  CLMGS_CHECK_ARG(means && quats && scales && viewmats && Ks && radii);
  dim3 grid(min(ceil_div(N, 256), 256 * 16), C);
  hipLaunchKernelGGL(projection_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, C, N, means,
                     quats, scales, viewmats, Ks, (float)width, (float)height, eps2d, near_plane,
                     far_plane, radius_clip, radii, means2d, depths, conics);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_visibility_raw(void* stream, int C, int N, const float* means,
                                    const float* quats_raw, const float* log_scales,
                                    const float* viewmats, const float* Ks, int width, int height,
                                    float eps2d, float near_plane, float far_plane,
                                    float radius_clip, int32_t* radii) {
  CLMGS_CHECK_ARG(C >= 1 && N >= 0 && width > 0 && height > 0);
  if (N == 0) return 0;
  CLMGS_CHECK_ARG(means && quats_raw && log_scales && viewmats && Ks && radii);
  hipLaunchKernelGGL(visibility_raw_kernel, dim3(min(ceil_div(N, 256), 256 * 16)), dim3(256), 0,
                     (hipStream_t)stream, C, N, means, quats_raw, log_scales, viewmats, Ks,
                     (float)width, (float)height, eps2d, near_plane, far_plane, radius_clip, radii);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_projection_bwd(void* stream, int C, int N, const float* means,
                                    const float* quats, const float* scales,
                                    const float* viewmats, const float* Ks, int width, int height,
                                    float eps2d, const int32_t* radii, const float* v_means2d,
                                    const float* v_depths, const float* v_conics, float* v_means,
                                    float* v_quats, float* v_scales) {
  CLMGS_CHECK_ARG(C >= 1 && N >= 0 && width > 0 && height > 0);
  if (N == 0) return 0;
  CLMGS_CHECK_ARG(means && quats && scales && viewmats && Ks && radii && v_means2d && v_conics &&
                  v_means && v_quats && v_scales);
  dim3 grid(min(ceil_div(N, 256), 256 * 16));
  hipLaunchKernelGGL(projection_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, C, N, means,
                     quats, scales, viewmats, Ks, (float)width, (float)height, eps2d, radii,
                     v_means2d, v_depths, v_conics, v_means, v_quats, v_scales);
  CLMGS_LAUNCH_CHECK();
  return 0;
}
