// View-dependent colour from spherical harmonics, forward and VJP (gfx950).
// Replaces gsplat.spherical_harmonics (strategies/base_engine.py:161-163,
// strategies/no_offload/engine.py:67-69, strategies/clm_offload/engine.py:73-76)
// and clm_kernels.spherical_harmonics_bwd_inplace (clm_offload/engine.py:709-716).
//
// HBM-bound: a Gaussian's coefficients are one 192 B row.  A block moves 256
// rows with fully coalesced 16 B/lane accesses into LDS (row pitch 52 floats:
// 16 B aligned and conflict-free for per-lane ds_read_b128), then every lane
// evaluates its own row out of LDS.  Masked-out rows are never fetched.
#include "common.h"
#include "gs_math.h"

namespace clmgs {

constexpr int SH_ROWS = 256;   // rows per block
constexpr int SH_PITCH = 52;   // floats per LDS row (48 + pad)

__device__ __forceinline__ int sh_row_f4(int degree) {
  const int nb = (degree + 1) * (degree + 1);
  return (nb * 3 + 3) / 4;  // float4s that cover the active coefficients
}

__device__ __forceinline__ void stage_rows_in(float* lds, const float* __restrict__ coeffs,
                                              const uint8_t* __restrict__ masks, int base,
                                              int rows, int nf4) {
  const int total = rows * nf4;
  for (int i = threadIdx.x; i < total; i += SH_ROWS) {
    const int r = i / nf4, k = i - r * nf4;
    if (masks && !masks[base + r]) continue;
    const float4 v = *reinterpret_cast<const float4*>(coeffs + (size_t)(base + r) * 48 + 4 * k);
    *reinterpret_cast<float4*>(lds + r * SH_PITCH + 4 * k) = v;
  }
}

__global__ void __launch_bounds__(SH_ROWS)
sh_fwd_kernel(int n, int degree, const float* __restrict__ dirs, const float* __restrict__ coeffs,
              const uint8_t* __restrict__ masks, float* __restrict__ colors) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nb = (degree + 1) * (degree + 1);
  const int nf4 = sh_row_f4(degree);
  const int n_chunks = (n + SH_ROWS - 1) / SH_ROWS;
  for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const int base = chunk * SH_ROWS;
    const int rows = min(SH_ROWS, n - base);
    __syncthreads();
    stage_rows_in(lds, coeffs, masks, base, rows, nf4);
    __syncthreads();
    const int g = base + threadIdx.x;
    if (threadIdx.x < rows) {
      float r = 0.f, gg = 0.f, b = 0.f;
      if (!masks || masks[g]) {
        float x = dirs[3 * g], y = dirs[3 * g + 1], z = dirs[3 * g + 2];
        const float inv = 1.0f / sqrtf(x * x + y * y + z * z);
        x *= inv; y *= inv; z *= inv;
        float B[16];
        sh_basis(degree, x, y, z, B);
        const float* row = lds + threadIdx.x * SH_PITCH;
#pragma unroll
        for (int k4 = 0; k4 < 12; ++k4) {
          if (k4 < nf4) {
            const float4 v = *reinterpret_cast<const float4*>(row + 4 * k4);
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int f = 4 * k4 + j;      // flat index = k*3 + c
              const int k = f / 3, c = f - 3 * k;
              if (k < nb) {
                const float t = B[k] * e[j];
                if (c == 0) r += t; else if (c == 1) gg += t; else b += t;
              }
            }
          }
        }
      }
      colors[3 * g] = r; colors[3 * g + 1] = gg; colors[3 * g + 2] = b;
    }
  }
}

template <bool ACCUM>
__global__ void __launch_bounds__(SH_ROWS)
sh_bwd_kernel(int n, int degree, const float* __restrict__ dirs, const float* __restrict__ coeffs,
              const uint8_t* __restrict__ masks, const float* __restrict__ v_colors,
              float* __restrict__ v_coeffs, float* __restrict__ v_dirs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nb = (degree + 1) * (degree + 1);
  const int nf4 = sh_row_f4(degree);
  const bool need_dirs = (v_dirs != nullptr) && degree > 0;
  const int n_chunks = (n + SH_ROWS - 1) / SH_ROWS;
  for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const int base = chunk * SH_ROWS;
    const int rows = min(SH_ROWS, n - base);
    __syncthreads();
    if (need_dirs) stage_rows_in(lds, coeffs, masks, base, rows, nf4);
    __syncthreads();
    const int g = base + threadIdx.x;
    if (threadIdx.x < rows) {
      const bool on = !masks || masks[g];
      float* row = lds + threadIdx.x * SH_PITCH;
      float vdx = 0.f, vdy = 0.f, vdz = 0.f;
      if (on) {
        float x = dirs[3 * g], y = dirs[3 * g + 1], z = dirs[3 * g + 2];
        const float inv = 1.0f / sqrtf(x * x + y * y + z * z);
        x *= inv; y *= inv; z *= inv;
        const float vc[3] = {v_colors[3 * g], v_colors[3 * g + 1], v_colors[3 * g + 2]};
        float B[16];
        sh_basis(degree, x, y, z, B);
        if (need_dirs) {
          float Bx[16], By[16], Bz[16];
          sh_basis_grad(degree, x, y, z, Bx, By, Bz);
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
        // this lane's row now becomes its v_coeffs row (only the lane itself read it)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const float bk = (k < nb) ? B[k] : 0.f;
          row[3 * k] = bk * vc[0]; row[3 * k + 1] = bk * vc[1]; row[3 * k + 2] = bk * vc[2];
        }
      } else if (!ACCUM) {
#pragma unroll
        for (int k = 0; k < 48; ++k) row[k] = 0.f;
      }
      if (v_dirs) { v_dirs[3 * g] = vdx; v_dirs[3 * g + 1] = vdy; v_dirs[3 * g + 2] = vdz; }
    }
    __syncthreads();
    // coalesced write-back of the 256 x 192 B gradient rows
    const int out_f4 = ACCUM ? nf4 : 12;
    const int total = rows * out_f4;
    for (int i = threadIdx.x; i < total; i += SH_ROWS) {
      const int r = i / out_f4, k = i - r * out_f4;
      if (ACCUM && masks && !masks[base + r]) continue;
      float4 v = *reinterpret_cast<const float4*>(lds + r * SH_PITCH + 4 * k);
      float4* dst = reinterpret_cast<float4*>(v_coeffs + (size_t)(base + r) * 48 + 4 * k);
      if (ACCUM) {
        const float4 o = *dst;
        // entries past the active degree inside the last float4 are exact zeros in lds
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      *dst = v;
    }
  }
}

}  // namespace clmgs

using namespace clmgs;

static int sh_grid(int n) { return min(ceil_div(n, SH_ROWS), 256 * 3); }

extern "C" int clmgs_sh_fwd(void* stream, int n, int degree, const float* dirs,
                            const float* coeffs, const uint8_t* masks, float* colors) {
  CLMGS_CHECK_ARG(n >= 0 && degree >= 0 && degree <= 3);
  if (n == 0) return 0;
  CLMGS_CHECK_ARG(dirs && coeffs && colors);
  const size_t lds = SH_ROWS * SH_PITCH * sizeof(float);
  hipLaunchKernelGGL(sh_fwd_kernel, dim3(sh_grid(n)), dim3(SH_ROWS), lds, (hipStream_t)stream, n,
                     degree, dirs, coeffs, masks, colors);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_sh_bwd(void* stream, int n, int degree, const float* dirs,
                            const float* coeffs, const uint8_t* masks, const float* v_colors,
                            float* v_coeffs, int accumulate, float* v_dirs) {
  CLMGS_CHECK_ARG(n >= 0 && degree >= 0 && degree <= 3);
  if (n == 0) return 0;
  CLMGS_CHECK_ARG(dirs && coeffs && v_colors && v_coeffs);
  const size_t lds = SH_ROWS * SH_PITCH * sizeof(float);
  if (accumulate)
    hipLaunchKernelGGL(sh_bwd_kernel<true>, dim3(sh_grid(n)), dim3(SH_ROWS), lds,
                       (hipStream_t)stream, n, degree, dirs, coeffs, masks, v_colors, v_coeffs,
                       v_dirs);
  else
    hipLaunchKernelGGL(sh_bwd_kernel<false>, dim3(sh_grid(n)), dim3(SH_ROWS), lds,
                       (hipStream_t)stream, n, degree, dirs, coeffs, masks, v_colors, v_coeffs,
                       v_dirs);
  CLMGS_LAUNCH_CHECK();
  return 0;
}
