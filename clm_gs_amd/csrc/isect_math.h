// Tile boxes and exact tile masks of a projected Gaussian: shared by the two binning routes (isect.hip: depth sort +
// tile sort; isect3.hip: tile-major counting sort + per-tile depth sort).
#pragma once
#include "common.h"
#include "gs_math.h"

namespace clmgs {

struct TileBox { int x0, y0, x1, y1; };

__device__ __forceinline__ TileBox tile_box(float mx, float my, float radius, float tile_size,
                                            int tile_w, int tile_h) {
  const float tr = radius / tile_size, tx = mx / tile_size, ty = my / tile_size;
  TileBox b;
  b.x0 = (int)fminf(fmaxf(floorf(tx - tr), 0.f), (float)tile_w);
  b.y0 = (int)fminf(fmaxf(floorf(ty - tr), 0.f), (float)tile_h);
  b.x1 = (int)fminf(fmaxf(ceilf(tx + tr), 0.f), (float)tile_w);
  b.y1 = (int)fminf(fmaxf(ceilf(ty + tr), 0.f), (float)tile_h);
  return b;
}


// Tile box of a row packed in one word (x0 | y0 << 16 | x1 << 32 | y1 << 48; tile counts < 65536).
__device__ __forceinline__ unsigned long long pack_box(const TileBox& b) {
  return (unsigned long long)b.x0 | ((unsigned long long)b.y0 << 16) |
         ((unsigned long long)b.x1 << 32) | ((unsigned long long)b.y1 << 48);
}

// Tile mask of a row: bit t of the (row-major) tiles of its box is set when alpha >= 1/255 is
// reachable somewhere on the tile -- the exact minimum of sigma over the tile's rectangle of pixel
// centres against ln(255 o), the same test (and margin) the tile kernels apply per 8x8 quadrant.
// Boxes of more than 64 tiles and degenerate conics are not culled (mask = all ones).
__device__ __forceinline__ unsigned long long exact_tile_mask(const float4* __restrict__ rec, int x0,
                                                              int y0, int x1, int y1) {
  const int bw = x1 - x0, nt = bw * (y1 - y0);
  if (nt > 64) return ~0ull;
  const float4 A = rec[0], B = rec[1];  // x y opacity ca | cb cc . .
  const float mx = A.x, my = A.y, opac = A.z, ca = A.w, cb = B.x, cc = B.y;
  if (!(opac >= 1.f / 255.f)) return 0ull;
  const float det = ca * cc - cb * cb;
  const float Lm = __logf(255.f * opac) * 1.0001f + 1e-4f;
  if (!(det > 0.f) || !(ca > 0.f) || !(cc > 0.f) || !(Lm == Lm)) return ~0ull;
  const float rca = __builtin_amdgcn_rcpf(ca), rcc = __builtin_amdgcn_rcpf(cc);
  unsigned long long m = 0ull;
  int t = 0;
  for (int ty = y0; ty < y1; ++ty) {
    const float v0 = (float)(ty * 16) + 0.5f - my, v1 = v0 + 15.f;
    for (int tx = x0; tx < x1; ++tx, ++t) {
      const float u0 = (float)(tx * 16) + 0.5f - mx, u1 = u0 + 15.f;
      if (rect_min_sigma(ca, cb, cc, rca, rcc, u0, u1, v0, v1) <= Lm) m |= 1ull << t;
    }
  }
  return m;
}


}  // namespace clmgs
