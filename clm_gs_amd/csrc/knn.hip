// simple_knn._C.distCUDA2 replacement (scope row f3; call site: create_from_pcd,
// strategies/clm_offload/gaussian_model.py:60-63): mean squared distance of every point to its 3
// nearest neighbours.  Exact search on a uniform grid: points are pre-sorted by cell (host side),
// a thread walks growing Chebyshev shells of cells until the 3rd-best distance is <= the distance
// to the nearest unvisited shell.
#include "common.h"

namespace clmgs {

struct KnnGrid { float ox, oy, oz, inv_h, h; int gx, gy, gz; };

__device__ __forceinline__ void knn_insert(float d2, float best[3]) {
  if (d2 < best[2]) {
    if (d2 < best[1]) {
      best[2] = best[1];
      if (d2 < best[0]) { best[1] = best[0]; best[0] = d2; } else best[1] = d2;
    } else best[2] = d2;
  }
}

__global__ void __launch_bounds__(256)
knn3_kernel(int n, const float* __restrict__ pts_sorted, const int32_t* __restrict__ cell_start,
            KnnGrid g, int max_ring, float* __restrict__ mean_d2_sorted) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float px = pts_sorted[3 * i], py = pts_sorted[3 * i + 1], pz = pts_sorted[3 * i + 2];
    const int cx = min(max((int)((px - g.ox) * g.inv_h), 0), g.gx - 1);
    const int cy = min(max((int)((py - g.oy) * g.inv_h), 0), g.gy - 1);
    const int cz = min(max((int)((pz - g.oz) * g.inv_h), 0), g.gz - 1);
    float best[3] = {3.0e38f, 3.0e38f, 3.0e38f};
    for (int R = 0; R <= max_ring; ++R) {
      for (int z = cz - R; z <= cz + R; ++z) {
        if (z < 0 || z >= g.gz) continue;
        for (int y = cy - R; y <= cy + R; ++y) {
          if (y < 0 || y >= g.gy) continue;
          const bool face = (abs(z - cz) == R) || (abs(y - cy) == R);
          const int step = face ? 1 : 2 * R;  // interior rows: only the two x-ends belong to the shell
          for (int x = cx - R; x <= cx + R; x += (step > 0 ? step : 1)) {
            if (x < 0 || x >= g.gx) continue;
            const int64_t cell = ((int64_t)z * g.gy + y) * g.gx + x;
            for (int j = cell_start[cell]; j < cell_start[cell + 1]; ++j) {
              if (j == i) continue;
              const float dx = pts_sorted[3 * j] - px, dy = pts_sorted[3 * j + 1] - py, dz = pts_sorted[3 * j + 2] - pz;
              knn_insert(dx * dx + dy * dy + dz * dz, best);
            }
          }
        }
      }
      const float reach = (float)R * g.h;  // everything unvisited is farther than this
      if (best[2] <= reach * reach) break;
    }
    mean_d2_sorted[i] = (best[0] + best[1] + best[2]) * (1.f / 3.f);
  }
}

}  // namespace clmgs

using namespace clmgs;

extern "C" int clmgs_knn3_mean_dist2(void* stream, int n, const float* pts_sorted,
                                     const int32_t* cell_start, float ox, float oy, float oz,
                                     float h, int gx, int gy, int gz, int max_ring,
                                     float* mean_d2_sorted) {
  CLMGS_CHECK_ARG(n >= 0 && h > 0.f && gx > 0 && gy > 0 && gz > 0 && max_ring >= 1);
  if (n == 0) return 0;
  CLMGS_CHECK_ARG(pts_sorted && cell_start && mean_d2_sorted);
  KnnGrid g{ox, oy, oz, 1.f / h, h, gx, gy, gz};
  hipLaunchKernelGGL(knn3_kernel, dim3(min(ceil_div(n, 256), 256 * 8)), dim3(256), 0,
                     (hipStream_t)stream, n, pts_sorted, cell_start, g, max_ring, mean_d2_sorted);
  CLMGS_LAUNCH_CHECK();
  return 0;
}
