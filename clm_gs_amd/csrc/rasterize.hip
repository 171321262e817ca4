// Per-tile front-to-back alpha blending and its VJP (gfx950).
// Replaces gsplat.rasterize_to_pixels forward + autograd backward
// (strategies/base_engine.py:192-203, strategies/no_offload/engine.py:86-97,
// strategies/clm_offload/engine.py:106-117; arithmetic: SURVEY.md A5/A6).
//
// CDNA4 mapping -- NOT the 16x16-threads-per-tile CUDA shape:
//   * one 64-lane wavefront owns one 16x16 tile; every lane carries 4 pixels, one in each 8x8
//     quadrant (lane = 8x8 position).  A tile's Gaussian record is read from LDS once per wave
//     and reused for up to 4 pixels per lane: LDS traffic per pixel-Gaussian pair is 1/4 of a
//     thread-per-pixel kernel, there is no multi-wave barrier, and the independent pixels give
//     the exp/fma chain ILP;
//   * exact sub-tile culling: the staging lane computes, from the conic and opacity, the exact
//     minimum of sigma over each 8x8 quadrant and a 4-bit mask of the quadrants where
//     alpha >= 1/255 is reachable.  Records whose box
//     misses the tile are compacted away with a ballot/popcount prefix (depth order kept),
//     quadrants it misses are skipped with wave-uniform branches.  Only pairs the reference
//     itself skips (alpha < 1/255) are dropped, so results are unchanged;
//   * backward: each lane first sums its pixels' contributions; the nine wave-wide sums of an entry then go
//     through LDS (eight of them, transposed: four ds_write2st64 + two ds_read_b128 + 7 adds + 3 DPP adds per
//     lane) and one DPP chain (the ninth), and ONE 64 B partial line per (Gaussian, tile) leaves the wave
//     (engine path: plain stores at the intersection's slot; gsplat-compatible op: float atomics), 64
//     Gaussians at a time by 64 lanes;
//   * blockIdx -> tile mapping is XCD-aware: each XCD's L2 sees a contiguous stripe of tiles,
//     so neighbouring tiles' shared Gaussians hit in L2.
#include <stdlib.h>

#include "common.h"
#include "gs_math.h"

namespace clmgs {

// Wave-uniform "some lane" / "no lane" tests straight from the compare's scalar mask.  (HIP's __any() goes through an
// int: the i1 is materialised with v_cndmask 0/1 and compared again -- two half-rate VALU per use in the blend loops.)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ bool wave_none(bool p) { return __builtin_amdgcn_ballot_w64(p) == 0ull; }

constexpr int TILE = 16;
constexpr int PPL = 4;        // pixels per lane (one per 8x8 quadrant)
constexpr float ALPHA_MIN = 1.f / 255.f;
constexpr float T_EPS = 1e-4f;
// The staging lane stores the conic pre-multiplied by log2(e) (and the 1/2 of the quadratic form), so
// the blend loops evaluate  s = log2(e) * sigma  with three FMA-class operations and feed v_exp_f32
// (a base-2 exponential) directly: two VALU fewer per (entry, quadrant) in both tile kernels.
constexpr float LOG2E = 1.4426950408889634f;
constexpr float CONIC_DIAG = 0.5f * LOG2E, CONIC_DIAG_INV = 2.f / LOG2E, CONIC_OFF_INV = 1.f / LOG2E;

// s = log2(e) * (0.5 (a dx^2 + c dy^2) + b dx dy) from the pre-scaled conic; identical code in the
// forward and the backward kernel, so both see the same alpha for every (entry, pixel)
__device__ __forceinline__ float scaled_sigma(float as, float bs, float cs, float dx, float dy) {
  return fmaf(cs * dy, dy, fmaf(as * dx, dx, (bs * dx) * dy));
}

// Per-Gaussian raster record, one 64 B line: {x, y, opacity, conic.a | conic.b, conic.c, r, g |
// b, -, -, - | -}.  The tile kernels gather ONE line per (Gaussian, tile) instead of touching
// four arrays (measured: 4.6 GB fetched per backward launch vs 1.3 GB algorithmic before).
constexpr int REC_F4 = 4;

__global__ void __launch_bounds__(256)
raster_pack_kernel(int64_t n, const float* __restrict__ means2d, const float* __restrict__ conics,
                   const float* __restrict__ colors, const float* __restrict__ opacities,
                   float4* __restrict__ packed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float2 m = *reinterpret_cast<const float2*>(means2d + 2 * i);
    const float* cn = conics + 3 * i;
    const float* cl = colors + 3 * i;
    float4* rec = packed + REC_F4 * i;
    rec[0] = make_float4(m.x, m.y, opacities[i], cn[0]);
    rec[1] = make_float4(cn[1], cn[2], cl[0], cl[1]);
    rec[2] = make_float4(cl[2], 0.f, 0.f, 0.f);
  }
}

// packed_grad line: x y ca cb | cc r g b | o - - - | -
__global__ void __launch_bounds__(256)
raster_unpack_grad_kernel(int64_t n, const float4* __restrict__ packed_grad,
                          float* __restrict__ v_means2d, float* __restrict__ v_conics,
                          float* __restrict__ v_colors, float* __restrict__ v_opacities) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = packed_grad[REC_F4 * i], b = packed_grad[REC_F4 * i + 1];
    const float o = packed_grad[REC_F4 * i + 2].x;
    *reinterpret_cast<float2*>(v_means2d + 2 * i) = make_float2(a.x, a.y);
    v_conics[3 * i] = a.z; v_conics[3 * i + 1] = a.w; v_conics[3 * i + 2] = b.x;
    v_colors[3 * i] = b.y; v_colors[3 * i + 1] = b.z; v_colors[3 * i + 2] = b.w;
    v_opacities[i] = o;
  }
}

struct TileLds {
  float4 a[64];   // x, y, opacity, conic.a
  float4 b[64];   // conic.b, conic.c, r, g
  float c[64];    // b
  int meta[64];   // (offset in staging round << 4) | quadrant mask
};

__device__ __forceinline__ void tile_range(const int32_t* __restrict__ offsets, int tile,
                                           int n_tiles_total, int64_t n_isects, int& s, int& e) {
  s = offsets[tile];
  e = (tile == n_tiles_total - 1) ? (int)n_isects : offsets[tile + 1];
}

// Quadrant mask of the pixels (centres) where alpha can reach 1/255, i.e. sigma <= L = ln(255 o).
// Exact: the minimum of sigma over each quadrant's rectangle of pixel centres is compared with L
// (a small margin absorbs rounding), so a quadrant is skipped only if EVERY pixel in it fails the
// reference's own alpha >= 1/255 test -- results are unchanged.  The test costs ~150 VALU but
// runs once per entry on the staging lane (64 entries per instruction): ~2.5 VALU per entry to
// save whole 64-lane blend passes.
__device__ __forceinline__ int quadrant_mask(float mx, float my, float opac, float ca, float cb,
                                             float cc, float tile_x0, float tile_y0) {
  if (!(opac >= ALPHA_MIN)) return 0;
  const float L = __logf(255.f * opac);
  const float det = ca * cc - cb * cb;
  if (!(det > 0.f) || !(ca > 0.f) || !(cc > 0.f)) return 15;  // degenerate conic: never cull
  const float Lm = L * 1.0001f + 1e-4f;
  if (!(Lm == Lm)) return 15;
  const float rca = __builtin_amdgcn_rcpf(ca), rcc = __builtin_amdgcn_rcpf(cc);
  // quadrant q covers pixel centres [q0 + 0.5, q0 + 7.5]; offsets are pixel - centre
  const float xa0 = tile_x0 + 0.5f - mx, xa1 = tile_x0 + 7.5f - mx;
  const float xb0 = tile_x0 + 8.5f - mx, xb1 = tile_x0 + 15.5f - mx;
  const float ya0 = tile_y0 + 0.5f - my, ya1 = tile_y0 + 7.5f - my;
  const float yb0 = tile_y0 + 8.5f - my, yb1 = tile_y0 + 15.5f - my;
  int m = 0;
  if (rect_min_sigma(ca, cb, cc, rca, rcc, xa0, xa1, ya0, ya1) <= Lm) m |= 1;
  if (rect_min_sigma(ca, cb, cc, rca, rcc, xb0, xb1, ya0, ya1) <= Lm) m |= 2;
  if (rect_min_sigma(ca, cb, cc, rca, rcc, xa0, xa1, yb0, yb1) <= Lm) m |= 4;
  if (rect_min_sigma(ca, cb, cc, rca, rcc, xb0, xb1, yb0, yb1) <= Lm) m |= 8;
  return m;
}

// SPECIAL entries (a wave-uniform flag in the entry's meta word, computed once per entry by the staging lane): opacity
// above 0.998, or a conic that is not positive definite with a condition number far from fp32's reach.  For every
// OTHER entry
//   * sigma cannot round below zero (sigma >= 0.5 lambda_min |d|^2, the rounding error of the three products is
//     <= 3 ulp x max(a, c) |d|^2, and det > 1e-5 (a + c)^2 puts lambda_min / max(a, c) fourteen times above that),
//     so gsplat's `sigma < 0` skip never fires, and
//   * o x G <= o <= 0.998 < 0.999, so the alpha clamp never fires and the saturated-alpha select is the identity,
// and the blend loops jump over those instructions with scalar branches on the flag: two compares, a v_min and a
// v_cndmask -- all half-rate VALU on this chip -- fewer per (entry, quadrant) in the backward, a compare and a v_min in
// the forward; same results bit for bit, one copy of the loop body.
#ifndef CLMGS_SPECIAL_ENTRIES
#define CLMGS_SPECIAL_ENTRIES 1
#endif
constexpr int META_SPECIAL = 16, META_SHIFT = 5;
__device__ __forceinline__ int special_entry(float opac, float ca, float cb, float cc) {
  const float tr = ca + cc;
  const bool plain = opac <= 0.998f && ca > 0.f && cc > 0.f && (ca * cc - cb * cb) > 1e-5f * tr * tr;
  return (CLMGS_SPECIAL_ENTRIES && plain) ? 0 : META_SPECIAL;
}

// waves/SIMD the forward is compiled for: 6 (80 VGPRs, six spilled dwords) measured 0.940 vs 0.970 ms at 5
// (87 VGPRs); 7 / 8 (more spills) 0.934 / 0.925 -- the issue rate of one-wave workgroups grows with the
// resident waves (profiles/ilp_probe.hip), the spills eat most of it
#ifndef CLMGS_FWD_WAVES
#define CLMGS_FWD_WAVES 6
#endif
#ifndef CLMGS_FWD_ASM
#define CLMGS_FWD_ASM 1
#endif
__global__ void __launch_bounds__(64, CLMGS_FWD_WAVES)
rasterize_fwd_kernel(int C, int N, int64_t n_isects, const float4* __restrict__ packed,
                     const float* __restrict__ backgrounds,
                     int W, int H, int tile_w, int tile_h, const int32_t* __restrict__ offsets,
                     const int32_t* __restrict__ flatten_ids, float* __restrict__ render_colors,
                     float* __restrict__ render_alphas, int32_t* __restrict__ last_ids,
                     const int64_t* __restrict__ n_dev) {
  __shared__ TileLds sm;
  if (n_dev) n_isects = min(n_isects, *n_dev);  // device-side count (the launch was prepared for a capacity)
  const int n_tiles = tile_w * tile_h;
  const int n_tiles_total = C * n_tiles;
  const int tile = (int)xcd_remap(blockIdx.x, (unsigned)n_tiles_total);
  const int cam = tile / n_tiles;
  const int t_in = tile - cam * n_tiles;
  const int ty = t_in / tile_w, tx = t_in - ty * tile_w;
  const int lane = threadIdx.x;
  const int qx = lane & 7, qy = lane >> 3;
  const float tile_x0 = (float)(tx * TILE), tile_y0 = (float)(ty * TILE);

  // Per-pixel state kept lean (the VGPR budget decides the waves per SIMD, and the issue rate of one-wave
  // workgroups grows with them): pixel centres are recomputed from the lane, and "this pixel still
  // accumulates" is the SIGN of T (T > 1e-4 while alive; a terminated or out-of-image pixel holds -T).
  float T[PPL], cr[PPL], cg[PPL], cb[PPL];
  int last[PPL];
  const float px0 = tile_x0 + (float)qx + 0.5f, py0 = tile_y0 + (float)qy + 0.5f;
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const int j = tx * TILE + 8 * (k & 1) + qx;
    const int i = ty * TILE + 8 * (k >> 1) + qy;
    cr[k] = cg[k] = cb[k] = 0.f; last[k] = 0;
    T[k] = ((i < H) && (j < W)) ? 1.f : -1.f;
  }

  int rs, re;
  tile_range(offsets, tile, n_tiles_total, n_isects, rs, re);

  // Software-pipelined staging: the id -> record gather of round r+1 is issued before round r's
  // blend loop and consumed after it, so its two dependent HBM/L2 latencies hide under compute.
  // Two-deep: ids run two rounds ahead of the blend loop, records one round ahead, so neither of
  // the two dependent gathers (id -> record) is ever waited for right after it is issued.
  float4 nA = make_float4(0.f, 0.f, 0.f, 0.f), nB = nA;
  float nblue = 0.f;
  int cur_g = (rs + lane < re) ? flatten_ids[rs + lane] : -1;        // ids of round 0
  int nxt_g = (rs + 64 + lane < re) ? flatten_ids[rs + 64 + lane] : -1;  // ids of round 1
  if (cur_g >= 0) {
    const float4* rec = packed + REC_F4 * (size_t)cur_g;
    nA = rec[0]; nB = rec[1]; nblue = rec[2].x;
  }
  // Per-quadrant termination (gsplat's per-pixel `done`, at the granularity this kernel branches on): bit k of
  // `alive` = some pixel of quadrant k still accumulates.  A quadrant whose 64 pixels have all terminated (or lie
  // outside the image) takes no further pass -- a pass over it changes nothing (`valid` needs T > 0) -- and an entry
  // that only reaches dead quadrants is not even staged.  Wave-uniform (SGPR), refreshed by the passes themselves.
  int alive = 0;
#pragma unroll
  for (int k = 0; k < PPL; ++k) alive |= __any(T[k] > 0.f) ? (1 << k) : 0;
  for (int bs = rs; bs < re; bs += 64) {
    if (!alive) break;
    const float4 A = nA, B = nB;
    const float blue = nblue;
    const int mask = (cur_g >= 0) ? (quadrant_mask(A.x, A.y, A.z, A.w, B.x, B.y, tile_x0, tile_y0) & alive) : 0;
    cur_g = nxt_g;
    if (cur_g >= 0) {  // records of the next round (their ids arrived a round ago)
      const float4* rec = packed + REC_F4 * (size_t)cur_g;
      nA = rec[0]; nB = rec[1]; nblue = rec[2].x;
    }
    {
      const int nidx = bs + 128 + lane;  // ids of the round after next
      nxt_g = (nidx < re) ? flatten_ids[nidx] : -1;
    }
    const unsigned long long bal = __ballot(mask != 0);
    const int pos = __popcll(bal & ((1ull << lane) - 1ull));
    const int bn = __popcll(bal);
    __syncthreads();
    if (mask) {
      sm.a[pos] = make_float4(A.x, A.y, A.z, A.w * CONIC_DIAG);
      sm.b[pos] = make_float4(B.x * LOG2E, B.y * CONIC_DIAG, B.z, B.w);
      sm.c[pos] = blue; sm.meta[pos] = (lane << 4) | mask;
    }
    __syncthreads();
    for (int t = 0; t < bn; ++t) {
      const float4 RA = sm.a[t];
      const float4 RB = sm.b[t];
      const int meta = __builtin_amdgcn_readfirstlane(sm.meta[t]);
      const float rblue = sm.c[t];
      const int gi = bs + (meta >> 4);
      int gi_v;  // the wave-uniform index in a VGPR, once per entry (v_cndmask cannot take it as a scalar
      asm("v_mov_b32 %0, %1" : "=v"(gi_v) : "s"(gi));  // next to its mask; the compiler re-moved it per pass)
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        if (meta & alive & (1 << k)) {  // wave-uniform: this quadrant can be touched and still accumulates
          const float dx = RA.x - (px0 + (float)(8 * (k & 1)));
          const float dy = RA.y - (py0 + (float)(8 * (k >> 1)));
          const float sigma = scaled_sigma(RA.w, RB.x, RB.y, dx, dy);  // log2(e) * sigma
          const float alpha = fminf(0.999f, RA.z * __builtin_amdgcn_exp2f(-sigma));
#if CLMGS_FWD_ASM
          // Round 6: three compares and three selects per pass instead of five and four, and no new scalar work (compares,
          // selects and v_min issue at half the FMA rate: profiles/r06_valu_calib.jsonl).  A finished / out-of-image pixel
          // holds a NEGATIVE T, so its next_T < 0 fails `next_T > eps` by itself (no `T > 0` compare); the two outcomes of a
          // hit are mask arithmetic on the compares' scalar results (s_and / s_andn2; the compiler issued v_cmp_nlt AND
          // v_cmp_lt for `goes_on` / `!goes_on`); the selects are written as v_cndmask on those scalar masks.  Results are
          // bit-identical to the select form below (CLMGS_FWD_ASM=0, the round-5 code).
          const float next_T = T[k] * (1.f - alpha);
          const unsigned long long hit_m = __builtin_amdgcn_ballot_w64(sigma >= 0.f) & __builtin_amdgcn_ballot_w64(alpha >= ALPHA_MIN);
          const unsigned long long go_m = __builtin_amdgcn_ballot_w64(next_T > T_EPS);
          const unsigned long long acc_m = hit_m & go_m, stop_m = hit_m & ~go_m;
          float vis = alpha * T[k];
          asm("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(vis) : "s"(acc_m));
          cr[k] += RB.z * vis; cg[k] += RB.w * vis; cb[k] += rblue * vis;
          asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(last[k]) : "v"(gi_v), "s"(acc_m));
          asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(T[k]) : "v"(next_T), "s"(acc_m));
          if (stop_m != 0ull) {
            // rare (a pixel stops once; finished pixels of a part-finished quadrant come by again): the sign of T is set
            // under a narrowed exec mask -- one full-rate v_or here instead of a fourth select in every pass.  (exec is all
            // ones around this point: 64 threads per workgroup, no divergent region open.)
            asm volatile("s_mov_b64 exec, %1\n\tv_or_b32_e32 %0, 0x80000000, %0\n\ts_mov_b64 exec, -1" : "+v"(T[k]) : "s"(stop_m));
            if (__builtin_amdgcn_ballot_w64(T[k] > 0.f) == 0ull) alive &= ~(1 << k);
          }
        }
      }
#else
          const bool valid = (T[k] > 0.f) && (sigma >= 0.f) && (alpha >= ALPHA_MIN);
          const float next_T = T[k] * (1.f - alpha);
          const bool goes_on = next_T > T_EPS;  // valid => alpha, T finite: one compare serves both cases
          const bool acc = valid && goes_on;
          const bool term = valid && !goes_on;
          const float vis = acc ? alpha * T[k] : 0.f;
          cr[k] += RB.z * vis; cg[k] += RB.w * vis; cb[k] += rblue * vis;
          last[k] = acc ? gi_v : last[k];
          T[k] = acc ? next_T : (term ? -T[k] : T[k]);
          // only a pass in which some pixel terminated can empty its quadrant: one ballot of `term` per pass
          // (it replaces the per-entry whole-tile test: three v_max + a compare over all four T)
          if (__any(term) && !__any(T[k] > 0.f)) alive &= ~(1 << k);
        }
      }
#endif
      if (!alive) break;
    }
  }

  float bgr = 0.f, bgg = 0.f, bgb = 0.f;
  if (backgrounds) { bgr = backgrounds[3 * cam]; bgg = backgrounds[3 * cam + 1]; bgb = backgrounds[3 * cam + 2]; }
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const int j = tx * TILE + 8 * (k & 1) + qx;
    const int i = ty * TILE + 8 * (k >> 1) + qy;
    if (i < H && j < W) {
      const size_t pix = ((size_t)cam * H + i) * W + j;
      const float Tf = fabsf(T[k]);
      render_colors[3 * pix] = cr[k] + Tf * bgr;
      render_colors[3 * pix + 1] = cg[k] + Tf * bgg;
      render_colors[3 * pix + 2] = cb[k] + Tf * bgb;
      render_alphas[pix] = 1.f - Tf;
      last_ids[pix] = last[k];
    }
  }
}

// Wave-wide sums of the backward through LDS (1) or through the permlane-swap / DPP butterflies (0).  See the loop.
#ifndef CLMGS_BWD_LDS_REDUCE
#define CLMGS_BWD_LDS_REDUCE 1
#endif
struct TileLdsBwd {
  float4 a[64];
  float4 b[64];
  float c[64];
  int meta[64];
  int id[64];        // Gaussian id (cam*N + g) of the compacted slot, or its emit slot (PART)
  __attribute__((aligned(16))) float acc[64][12];  // reduced per-Gaussian sums of this tile (9 used)
#if CLMGS_BWD_LDS_REDUCE
  __attribute__((aligned(16))) float red[8][64];   // transposed scratch of the wave-wide sums: [value][lane]
#endif
};

// DBG: profiling-only variants (1 = skip the atomics flush, 2 = skip the reduction too, 3 = phase
// timers + work counters accumulated into g_dbg, read with clmgs_debug_counters); the product
// launches DBG = 0.
__device__ unsigned long long g_dbg[16];
#define DBG_CLK() (DBG == 3 ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
#ifndef CLMGS_BWD_WAVES
#define CLMGS_BWD_WAVES 4  // 5 spills (96 VGPRs): scratch reloads force vmcnt(0) and kill the prefetch
#endif
// PART: atomic-free accumulation.  Every sorted intersection owns the line partials[slot] (PART_F4 float4 = 64 B)
// (slot = its emit index, see isect2_emit_kernel); the tile's wave STORES the reduced sums there
// (zeros for culled / unreached entries, so every line is written exactly once per launch) and
// raster_partials_sum_kernel adds each row's contiguous range.
template <int DBG, bool PART>
__global__ void __launch_bounds__(64, CLMGS_BWD_WAVES)
rasterize_bwd_kernel(int C, int N, int64_t n_isects, const float4* __restrict__ packed,
                     const float* __restrict__ backgrounds,
                     int W, int H, int tile_w, int tile_h, const int32_t* __restrict__ offsets,
                     const int32_t* __restrict__ flatten_ids,
                     const float* __restrict__ render_alphas, const int32_t* __restrict__ last_ids,
                     const float* __restrict__ v_render_colors,
                     const float* __restrict__ v_render_alphas, float* __restrict__ packed_grad,
                     const int32_t* __restrict__ emit_slot, float4* __restrict__ partials,
                     const int64_t* __restrict__ n_dev) {
  __shared__ TileLdsBwd sm;
  if (n_dev) n_isects = min(n_isects, *n_dev);
  const int n_tiles = tile_w * tile_h;
  const int n_tiles_total = C * n_tiles;
  const int tile = (int)xcd_remap(blockIdx.x, (unsigned)n_tiles_total);
  const int cam = tile / n_tiles;
  const int t_in = tile - cam * n_tiles;
  const int ty = t_in / tile_w, tx = t_in - ty * tile_w;
  const int lane = threadIdx.x;
  const int qx = lane & 7, qy = lane >> 3;
  const float tile_x0 = (float)(tx * TILE), tile_y0 = (float)(ty * TILE);

  int rs, re;
  tile_range(offsets, tile, n_tiles_total, n_isects, rs, re);
  if (re <= rs) return;

  // per-pixel state kept lean (VGPR budget decides waves/SIMD): pixel centres are recomputed from
  // the lane, "inside" is bin < 0, and T_final * v_alpha' is pre-multiplied.
  // Bk = (colour accumulated behind the current Gaussian) . v_rgb  -  T_final * v_alpha': the
  // behind-colour only ever appears dotted with the pixel's colour cotangent, so ONE running
  // scalar replaces three buffers and the alpha-cotangent term (fewer VGPRs, 7 fewer VALU/pair).
  float T[PPL], Bk[PPL], vr[PPL], vg[PPL], vb[PPL];
  int bin[PPL];
  int max_bin = -1;
  int qmax[PPL];  // per quadrant: the deepest contributor of its 64 pixels (wave-uniform)
  float bgr = 0.f, bgg = 0.f, bgb = 0.f;
  if (backgrounds) { bgr = backgrounds[3 * cam]; bgg = backgrounds[3 * cam + 1]; bgb = backgrounds[3 * cam + 2]; }
  const float px0 = tile_x0 + (float)qx + 0.5f, py0 = tile_y0 + (float)qy + 0.5f;
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const int j = tx * TILE + 8 * (k & 1) + qx;
    const int i = ty * TILE + 8 * (k >> 1) + qy;
    if ((i < H) && (j < W)) {
      const size_t pix = ((size_t)cam * H + i) * W + j;
      const float Tf = 1.f - render_alphas[pix];
      bin[k] = last_ids[pix];
      vr[k] = v_render_colors[3 * pix]; vg[k] = v_render_colors[3 * pix + 1]; vb[k] = v_render_colors[3 * pix + 2];
      float va = v_render_alphas ? v_render_alphas[pix] : 0.f;
      // d(out)/d(T_final) through the background term folds into the alpha cotangent
      va -= (bgr * vr[k] + bgg * vg[k] + bgb * vb[k]);
      Bk[k] = -Tf * va;
      T[k] = Tf;
      max_bin = max(max_bin, bin[k]);
    } else {
      T[k] = 1.f; bin[k] = -1; vr[k] = vg[k] = vb[k] = Bk[k] = 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < PPL; ++k) qmax[k] = __builtin_amdgcn_readfirstlane(wave_max_i32(bin[k]));
  max_bin = max(max(qmax[0], qmax[1]), max(qmax[2], qmax[3]));
  const int hi = min(re - 1, max_bin);  // nothing behind the deepest contributor matters
  unsigned long long c_stage = 0, c_loop = 0, c_flush = 0, n_ent = 0, n_valid = 0, n_quad = 0, n_round = 0;
  const unsigned long long t_begin = DBG_CLK();

  // two-deep software-pipelined staging (see the forward kernel): ids two rounds ahead,
  // records one round ahead; nothing is waited for right after issue, and the waits never
  // include the previous round's atomics
  float4 nA = make_float4(0.f, 0.f, 0.f, 0.f), nB = nA;
  float nblue = 0.f;
  int cur_g = (hi - lane >= rs) ? flatten_ids[hi - lane] : -1;
  int nxt_g = (hi - 64 - lane >= rs) ? flatten_ids[hi - 64 - lane] : -1;
  int cur_p = 0, nxt_p = 0;
  if (PART) {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int idx = max(hi + 1, rs) + lane; idx < re; idx += 64) {  // behind the deepest contributor: zeros
      float4* dst = partials + PART_F4 * (size_t)emit_slot[idx];
#pragma unroll
      for (int q = 0; q < PART_F4; ++q) dst[q] = z4;
    }
    cur_p = (hi - lane >= rs) ? emit_slot[hi - lane] : 0;
    nxt_p = (hi - 64 - lane >= rs) ? emit_slot[hi - 64 - lane] : 0;
  }
  if (cur_g >= 0) {
    const float4* rec = packed + REC_F4 * (size_t)cur_g;
    nA = rec[0]; nB = rec[1]; nblue = rec[2].x;
  }
  for (int bh = hi; bh >= rs; bh -= 64) {
    const unsigned long long tA = DBG_CLK();
    const float4 A = nA, B = nB;
    const float blue = nblue;
    const int gid = cur_g;
    const int pid = cur_p;
    // Per-quadrant termination: entry gi matters to quadrant k only while gi <= qmax[k] (`valid` needs
    // gi <= bin of the pixel); the staging lane drops the other quadrants from its mask, and an entry left
    // without a quadrant is not staged at all (its partial line is the zero line below).  Results unchanged.
    const int gidx = bh - lane;
    const int reach = (gidx <= qmax[0] ? 1 : 0) | (gidx <= qmax[1] ? 2 : 0) | (gidx <= qmax[2] ? 4 : 0) |
                      (gidx <= qmax[3] ? 8 : 0);
    const int mask = (gid >= 0) ? (quadrant_mask(A.x, A.y, A.z, A.w, B.x, B.y, tile_x0, tile_y0) & reach) : 0;
    cur_g = nxt_g;
    cur_p = nxt_p;
    if (cur_g >= 0) {
      const float4* rec = packed + REC_F4 * (size_t)cur_g;
      nA = rec[0]; nB = rec[1]; nblue = rec[2].x;
    }
    {
      const int nidx = bh - 128 - lane;
      nxt_g = (nidx >= rs) ? flatten_ids[nidx] : -1;
      if (PART) nxt_p = (nidx >= rs) ? emit_slot[nidx] : 0;
    }
    if (PART && gid >= 0 && mask == 0) {  // culled for this tile: its line is all zeros
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      float4* dst = partials + PART_F4 * (size_t)pid;
#pragma unroll
      for (int q = 0; q < PART_F4; ++q) dst[q] = z4;
    }
    const unsigned long long bal = __ballot(mask != 0);
    const int pos = __popcll(bal & ((1ull << lane) - 1ull));
    const int bn = __popcll(bal);
    __syncthreads();
    if (mask) {
      sm.a[pos] = make_float4(A.x, A.y, A.z, A.w * CONIC_DIAG);
      sm.b[pos] = make_float4(B.x * LOG2E, B.y * CONIC_DIAG, B.z, B.w);
      sm.c[pos] = blue; sm.meta[pos] = (lane << META_SHIFT) | special_entry(A.z, A.w, B.x, B.y) | mask;
      sm.id[pos] = PART ? pid : gid;
    }
    __syncthreads();
    const unsigned long long tB = DBG_CLK();
    unsigned long long touched = 0ull;
    for (int t = 0; t < bn; ++t) {
      const float4 RA = sm.a[t];
      const float4 RB = sm.b[t];
      const int meta = __builtin_amdgcn_readfirstlane(sm.meta[t]);
      const float rblue = sm.c[t];
      const int gi = bh - (meta >> META_SHIFT);
      const bool special = (meta & META_SPECIAL) != 0;  // wave-uniform
      if (DBG == 3) { n_ent++; n_quad += __popc(meta & 15); }
      // moments of w = v_sigma over the tile: the five screen-space gradients are linear in them
      // (g_x = a Sx + b Sy, g_y = b Sx + c Sy, g_conic = Sxx/2, Sxy, Syy/2), applied at the flush
      float g_r = 0.f, g_g = 0.f, g_b = 0.f, Sx = 0.f, Sy = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f,
            g_o = 0.f;
      // OR of the passes' valid masks, kept as scalar mask arithmetic on the compares' results (a per-lane flag costs a
      // v_mov per pass and a v_cmp at the end; measured -0.8 % slab / -1.2 % heavy, profiles/r06_raster_ab8.txt)
      unsigned long long any_valid = 0ull;
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
          if (meta & (1 << k)) {  // wave-uniform
            const float dx = RA.x - (px0 + (float)(8 * (k & 1)));
            const float dy = RA.y - (py0 + (float)(8 * (k >> 1)));
            const float sigma = scaled_sigma(RA.w, RB.x, RB.y, dx, dy);  // log2(e) * sigma
            const float gex = __builtin_amdgcn_exp2f(-sigma);
            const float oa = RA.z * gex;
            // (scalar branches on `special`; the empty asm keeps them branches -- if-converted they are selects again)
            float alpha = oa;
            bool ok = true;
            unsigned long long ok_m = ~0ull;
            if (__builtin_expect(special, 0)) {
              asm volatile("");
              alpha = fminf(0.999f, oa); ok = sigma >= 0.f; ok_m = __builtin_amdgcn_ballot_w64(sigma >= 0.f);
            }
            const bool valid = ok & (gi <= bin[k]) & (alpha >= ALPHA_MIN);
            any_valid |= ok_m & __builtin_amdgcn_ballot_w64(gi <= bin[k]) & __builtin_amdgcn_ballot_w64(alpha >= ALPHA_MIN);
            if (valid) {
              const float ra = __builtin_amdgcn_rcpf(1.f - alpha);  // v_rcp_f32 (1 ulp), not the 10-op IEEE divide
              T[k] *= ra;
              const float fac = alpha * T[k];
              g_r += fac * vr[k]; g_g += fac * vg[k]; g_b += fac * vb[k];
              const float cv = RB.z * vr[k] + RB.w * vg[k] + rblue * vb[k];
              // a saturated alpha (o * G > 0.999, clamped; special entries only) passes no gradient to sigma / opacity
              float v_alpha = T[k] * cv - ra * Bk[k];
              if (__builtin_expect(special, 0)) { asm volatile(""); v_alpha = (oa <= 0.999f) ? v_alpha : 0.f; }
              const float w = -oa * v_alpha;  // v_sigma
              const float wdx = w * dx, wdy = w * dy;
              Sx += wdx; Sy += wdy;
              Sxx += wdx * dx; Sxy += wdx * dy; Syy += wdy * dy;
              g_o += gex * v_alpha;
              Bk[k] += fac * cv;
            }
          }
        }
      if (any_valid == 0ull) continue;
      if (DBG == 3) n_valid++;
      if (DBG != 2) {
#if CLMGS_BWD_LDS_REDUCE
        // Eight of the nine wave-wide sums through LDS, transposed: lane L stores value q at red[q][L] (four
        // ds_write2st64_b32: the two values of a pair lie 64 dwords apart); lane j = 8 q + c then loads eight
        // floats of red[q][.] (two ds_read_b128, conflict-free: see below), adds them (7 VALU) and three
        // row_shr DPP adds finish value q in lane 8 q + 7.  ~12 issue slots for eight sums against ~37 for the two
        // permlane-swap butterflies; the LDS instructions issue beside other waves' VALU.  The ninth sum (opacity)
        // keeps its DPP chain.  One wave per workgroup: LDS operations of a wave execute in order, the fences only
        // keep the compiler from moving the loads above the stores.
        sm.red[0][lane] = Sx; sm.red[1][lane] = Sy; sm.red[2][lane] = Sxx; sm.red[3][lane] = Sxy;
        sm.red[4][lane] = Syy; sm.red[5][lane] = g_r; sm.red[6][lane] = g_g; sm.red[7][lane] = g_b;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        // (lane 8q + c takes floats [4c, 4c+4) and [32 + 4c, 32 + 4c + 4) of value q: consecutive lanes read consecutive
        //  16 B pieces -- the [8c, 8c+8) assignment had a stride of 32 B between lanes and a 2-way bank conflict on both loads)
        const float4* rp = reinterpret_cast<const float4*>(&sm.red[lane >> 3][4 * (lane & 7)]);
        const float4 r0 = rp[0], r1 = rp[8];
        g_o = wave_sum_to_lane63(g_o);
        float u = ((r0.x + r0.y) + (r0.z + r0.w)) + ((r1.x + r1.y) + (r1.z + r1.w));
        u = dpp_add<0x111>(u);
        u = dpp_add<0x112>(u);
        u = dpp_add<0x114>(u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();  // (the next entry's stores stay below this entry's loads)
        if ((lane & 7) == 7) sm.acc[t][lane >> 3] = u;   // Sx Sy Sxx Sxy | Syy r g b | o
        if (lane == 63) sm.acc[t][8] = g_o;
#else
        // 9 wave-wide sums: two 4-packs on the permlane-swap butterfly + one plain DPP chain
        const float u1 = wave_sum4_rows(Sx, Sy, Sxx, Sxy);       // lanes 15/31/47/63: Sx, Sxx, Sy, Sxy
        const float u2 = wave_sum4_rows(Syy, g_r, g_g, g_b);     //                    Syy, g, r, b
        g_o = wave_sum_to_lane63(g_o);
        if ((lane & 15) == 15) {
          const int r = lane >> 4;
          const int m = ((r & 1) << 1) | (r >> 1);               // row -> slot {0,2,1,3}
          float* a = sm.acc[t];                                   // Sx Sy Sxx Sxy | Syy r g b | o
          a[m] = u1;
          a[4 + m] = u2;
          if (lane == 63) a[8] = g_o;
        }
#endif
      }
      touched |= (1ull << t);
    }
    __syncthreads();
    const unsigned long long tC = DBG_CLK();
    if (DBG == 1 || DBG == 2) {
      if (((touched >> lane) & 1ull) && sm.acc[lane][0] == 1.2345e30f) packed_grad[0] = 1.f;
    } else if (PART) {
      if (lane < bn) {  // one full line (PART_F4 float4) per entry of the round, zeros if no pixel was valid
        const bool hit = (touched >> lane) & 1ull;
        const float4* a = reinterpret_cast<const float4*>(sm.acc[lane]);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 r0 = z4, r1 = z4, r2 = z4;
        if (hit) {  // moments -> gradients here (the conic is at hand): x y ca cb | cc r g b | o
          const float4 m0 = a[0], m1 = a[1];
          const float ca = sm.a[lane].w * CONIC_DIAG_INV, cb = sm.b[lane].x * CONIC_OFF_INV,
                      cc = sm.b[lane].y * CONIC_DIAG_INV;  // back from the pre-scaled form
          r0 = make_float4(ca * m0.x + cb * m0.y, cb * m0.x + cc * m0.y, 0.5f * m0.z, m0.w);
          r1 = make_float4(0.5f * m1.x, m1.y, m1.z, m1.w);
          r2 = make_float4(a[2].x, 0.f, 0.f, 0.f);
        }
        float4* dst = partials + PART_F4 * (size_t)sm.id[lane];
        dst[0] = r0; dst[1] = r1; dst[2] = r2;
        if (PART_F4 > 3) dst[3] = z4;
      }
    } else if ((touched >> lane) & 1ull) {
      // all nine atomics of a Gaussian land in its one 64 B gradient line
      float* dst = packed_grad + 4 * REC_F4 * (size_t)sm.id[lane];
      const float* a = sm.acc[lane];
      const float ca = sm.a[lane].w * CONIC_DIAG_INV, cb = sm.b[lane].x * CONIC_OFF_INV,
                      cc = sm.b[lane].y * CONIC_DIAG_INV;  // back from the pre-scaled form
      atomicAdd(dst + 0, ca * a[0] + cb * a[1]);  // x
      atomicAdd(dst + 1, cb * a[0] + cc * a[1]);  // y
      atomicAdd(dst + 2, 0.5f * a[2]);            // conic a
      atomicAdd(dst + 3, a[3]);                   // conic b
      atomicAdd(dst + 4, 0.5f * a[4]);            // conic c
#pragma unroll
      for (int c = 5; c < 9; ++c) atomicAdd(dst + c, a[c]);
    }
    if (DBG == 3) {
      const unsigned long long tD = DBG_CLK();
      c_stage += tB - tA; c_loop += tC - tB; c_flush += tD - tC; n_round++;
    }
  }
  if (DBG == 3 && lane == 0) {
    atomicAdd(&g_dbg[0], c_stage); atomicAdd(&g_dbg[1], c_loop); atomicAdd(&g_dbg[2], c_flush);
    atomicAdd(&g_dbg[3], DBG_CLK() - t_begin); atomicAdd(&g_dbg[4], n_ent); atomicAdd(&g_dbg[5], n_valid);
    atomicAdd(&g_dbg[6], n_quad); atomicAdd(&g_dbg[7], n_round); atomicAdd(&g_dbg[8], 1ull);
    atomicAdd(&g_dbg[9], (unsigned long long)(re - rs));
  }
}

// Per-row sum of the tile partials (API surface: the engine path folds this sum into
// clmgs_preprocess_bwd).  Row i owns the contiguous slot range [row_cum[i-1], row_cum[i]); ranges and
// gradient lines are both walked sequentially; fixed order (ascending slot).
// partials line = gradient line: x y ca cb | cc r g b | o - - - | -
__global__ void __launch_bounds__(256)
raster_partials_sum_kernel(int64_t n_rows, const int64_t* __restrict__ row_cum,
                           const float4* __restrict__ partials, float4* __restrict__ packed_grad) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s0 = r ? row_cum[r - 1] : 0;
    const int cnt = (int)(row_cum[r] - s0);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    float o = 0.f;
    const float4* src = partials + PART_F4 * (size_t)s0;
    for (int t = 0; t < cnt; t += 4) {  // four lines in flight per step (clamped; extra ones masked)
      float4 pa[4], pb[4];
      float po[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int tt = min(t + u, cnt - 1);
        pa[u] = src[PART_F4 * tt]; pb[u] = src[PART_F4 * tt + 1]; po[u] = src[PART_F4 * tt + 2].x;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (t + u < cnt) {
          a.x += pa[u].x; a.y += pa[u].y; a.z += pa[u].z; a.w += pa[u].w;
          b.x += pb[u].x; b.y += pb[u].y; b.z += pb[u].z; b.w += pb[u].w;
          o += po[u];
        }
      }
    }
    float4* dst = packed_grad + REC_F4 * r;
    dst[0] = a; dst[1] = b; dst[2] = make_float4(o, 0.f, 0.f, 0.f);
  }
}

}  // namespace clmgs

using namespace clmgs;

extern "C" size_t clmgs_rasterize_partials_bytes(int64_t n_isects) {
  return (size_t)(n_isects > 0 ? n_isects : 1) * (PART_F4 * 16);
}

// Profiling aid (CLMGS_BWD_DEBUG=3): stage / loop / flush / total cycles, entries, entries with a
// valid pixel, quadrant passes, staging rounds, tiles, list length.  out[16]; reset != 0 clears.
extern "C" int clmgs_debug_counters(unsigned long long* out, int reset) {
  CLMGS_HIP(hipDeviceSynchronize());
  CLMGS_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 16));
  if (reset) {
    unsigned long long z[16] = {0};
    CLMGS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof(z)));
  }
  return 0;
}

extern "C" size_t clmgs_rasterize_pack_bytes(int C, int N) {
  return (size_t)C * (size_t)N * REC_F4 * sizeof(float4);
}

static int rasterize_fwd_impl(void* stream, int C, int N, int64_t n_isects,
                              const float* means2d, const float* conics, const float* colors,
                              const float* opacities, const float* backgrounds, int width,
                              int height, int tile_size, int tile_width, int tile_height,
                              const int32_t* offsets, const int32_t* flatten_ids, void* packed,
                              float* render_colors, float* render_alphas, int32_t* last_ids,
                              const int64_t* n_dev) {
  CLMGS_CHECK_ARG(tile_size == TILE);
  CLMGS_CHECK_ARG(C >= 1 && width > 0 && height > 0 && tile_width * TILE >= width &&
                  tile_height * TILE >= height);
  CLMGS_CHECK_ARG(offsets && render_colors && render_alphas && last_ids);
  CLMGS_CHECK_ARG(n_isects == 0 || (flatten_ids && packed));
  CLMGS_CHECK_ARG(!means2d || (conics && colors && opacities));
  CLMGS_CHECK_ARG(((uintptr_t)packed & 63) == 0);
  hipStream_t s = (hipStream_t)stream;
  const int64_t CN = (int64_t)C * N;
  if (means2d && n_isects > 0 && CN > 0) {  // means2d == NULL: `packed` was filled by the caller
    hipLaunchKernelGGL(raster_pack_kernel, dim3(min(ceil_div(CN, 256), 256 * 8)), dim3(256), 0, s, CN,
                       means2d, conics, colors, opacities, (float4*)packed);
    CLMGS_LAUNCH_CHECK();
  }
  const int n_blocks = C * tile_width * tile_height;
#ifdef CLMGS_PROFILE_BUILD  // occupancy experiments: extra dynamic LDS per workgroup
  static const int fwd_pad = getenv("CLMGS_FWD_LDS_PAD") ? atoi(getenv("CLMGS_FWD_LDS_PAD")) : 0;
#else
  const int fwd_pad = 0;
#endif
  hipLaunchKernelGGL(rasterize_fwd_kernel, dim3(n_blocks), dim3(64), fwd_pad, s, C, N, n_isects,
                     (const float4*)packed, backgrounds, width, height, tile_width, tile_height,
                     offsets, flatten_ids, render_colors, render_alphas, last_ids, n_dev);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_rasterize_fwd(void* stream, int C, int N, int64_t n_isects,
                                   const float* means2d, const float* conics, const float* colors,
                                   const float* opacities, const float* backgrounds, int width,
                                   int height, int tile_size, int tile_width, int tile_height,
                                   const int32_t* offsets, const int32_t* flatten_ids, void* packed,
                                   float* render_colors, float* render_alphas, int32_t* last_ids) {
  return rasterize_fwd_impl(stream, C, N, n_isects, means2d, conics, colors, opacities, backgrounds, width, height,
                            tile_size, tile_width, tile_height, offsets, flatten_ids, packed, render_colors,
                            render_alphas, last_ids, nullptr);
}

// Device-count form (see clmgs_isect2_emit_sort_dev): `capacity` bounds the list, the true count is read on the
// device; the records are the caller's packed [N,16] lines.
extern "C" int clmgs_rasterize_fwd_dev(void* stream, int C, int N, int64_t capacity, const int64_t* n_isects_dev,
                                       const float* backgrounds, int width, int height, int tile_size,
                                       int tile_width, int tile_height, const int32_t* offsets,
                                       const int32_t* flatten_ids, void* packed, float* render_colors,
                                       float* render_alphas, int32_t* last_ids) {
  CLMGS_CHECK_ARG(n_isects_dev && capacity > 0);
  return rasterize_fwd_impl(stream, C, N, capacity, nullptr, nullptr, nullptr, nullptr, backgrounds, width, height,
                            tile_size, tile_width, tile_height, offsets, flatten_ids, packed, render_colors,
                            render_alphas, last_ids, n_isects_dev);
}

static int rasterize_bwd_impl(void* stream, int C, int N, int64_t n_isects, const void* packed,
                              const float* backgrounds, int width, int height, int tile_size,
                              int tile_width, int tile_height, const int32_t* offsets,
                              const int32_t* flatten_ids, const float* render_alphas,
                              const int32_t* last_ids, const float* v_render_colors,
                              const float* v_render_alphas, void* packed_grad,
                              float* v_means2d, float* v_conics, float* v_colors,
                              float* v_opacities, const int32_t* emit_slot,
                              const int64_t* row_cum, void* partials, const int64_t* n_dev) {
  CLMGS_CHECK_ARG(tile_size == TILE);
  CLMGS_CHECK_ARG(C >= 1 && width > 0 && height > 0 && tile_width * TILE >= width &&
                  tile_height * TILE >= height);
  CLMGS_CHECK_ARG(!v_means2d || (v_conics && v_colors && v_opacities));
  hipStream_t s = (hipStream_t)stream;
  const int64_t CN = (int64_t)C * N;
  if (CN == 0) return 0;  // nothing to differentiate: no argument below is required to exist
  // slot mode is selected by the partial-line table: a camera without a single intersection hands over
  // an EMPTY emit_slot (NULL data pointer) and must still be accepted (its gradients are all zero)
  const bool part = partials != nullptr;
  CLMGS_CHECK_ARG(!part || (C == 1 && (emit_slot || n_isects == 0) && (((uintptr_t)partials & 63) == 0)));
  // packed_grad == NULL (slot mode only): the caller sums the partial lines itself (clmgs_preprocess_bwd)
  CLMGS_CHECK_ARG(packed_grad ? ((((uintptr_t)packed_grad & 63) == 0) && (!part || row_cum)) : (part && !v_means2d));
  if (packed_grad && (!part || n_isects == 0))
    CLMGS_HIP(hipMemsetAsync(packed_grad, 0, clmgs_rasterize_pack_bytes(C, N), s));
  if (n_isects > 0) {
    CLMGS_CHECK_ARG(packed && offsets && flatten_ids && render_alphas && last_ids && v_render_colors);
    const int n_blocks = C * tile_width * tile_height;
    // The product build launches <0, PART> only.  The profiling variants (skip the flush / skip the
    // reduction / phase timers), which produce WRONG gradients by design, and the environment lookups
    // that select them exist only in a -DCLMGS_PROFILE_BUILD library (make PROFILE=1).
#ifdef CLMGS_PROFILE_BUILD
    static const int dbg = getenv("CLMGS_BWD_DEBUG") ? atoi(getenv("CLMGS_BWD_DEBUG")) : 0;
    static const int bwd_pad = getenv("CLMGS_BWD_LDS_PAD") ? atoi(getenv("CLMGS_BWD_LDS_PAD")) : 0;
#else
    const int bwd_pad = 0;
#endif
#define CLMGS_LAUNCH_BWD(D, P)                                                                     \
  hipLaunchKernelGGL((rasterize_bwd_kernel<D, P>), dim3(n_blocks), dim3(64), bwd_pad, s, C, N,     \
                     n_isects, (const float4*)packed, backgrounds, width, height, tile_width,      \
                     tile_height, offsets, flatten_ids, render_alphas, last_ids, v_render_colors,  \
                     v_render_alphas, (float*)packed_grad, emit_slot, (float4*)partials, n_dev)
#ifdef CLMGS_PROFILE_BUILD
    if (part) {
      if (dbg == 1) CLMGS_LAUNCH_BWD(1, true); else if (dbg == 3) CLMGS_LAUNCH_BWD(3, true);
      else CLMGS_LAUNCH_BWD(0, true);
    } else {
      if (dbg == 1) CLMGS_LAUNCH_BWD(1, false); else if (dbg == 2) CLMGS_LAUNCH_BWD(2, false);
      else if (dbg == 3) CLMGS_LAUNCH_BWD(3, false); else CLMGS_LAUNCH_BWD(0, false);
    }
#else
    if (part) CLMGS_LAUNCH_BWD(0, true); else CLMGS_LAUNCH_BWD(0, false);
#endif
#undef CLMGS_LAUNCH_BWD
    CLMGS_LAUNCH_CHECK();
    if (part && packed_grad) {
      hipLaunchKernelGGL(raster_partials_sum_kernel, dim3(min(ceil_div(CN, 256), 256 * 16)), dim3(256),
                         0, s, CN, row_cum, (const float4*)partials, (float4*)packed_grad);
      CLMGS_LAUNCH_CHECK();
    }
  }
  if (v_means2d) {  // NULL: the caller consumes the packed gradient lines directly
    hipLaunchKernelGGL(raster_unpack_grad_kernel, dim3(min(ceil_div(CN, 256), 256 * 8)), dim3(256), 0,
                       s, CN, (const float4*)packed_grad, v_means2d, v_conics, v_colors, v_opacities);
    CLMGS_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int clmgs_rasterize_bwd(void* stream, int C, int N, int64_t n_isects, const void* packed,
                                   const float* backgrounds, int width, int height, int tile_size,
                                   int tile_width, int tile_height, const int32_t* offsets,
                                   const int32_t* flatten_ids, const float* render_alphas,
                                   const int32_t* last_ids, const float* v_render_colors,
                                   const float* v_render_alphas, void* packed_grad,
                                   float* v_means2d, float* v_conics, float* v_colors,
                                   float* v_opacities, const int32_t* emit_slot,
                                   const int64_t* row_cum, void* partials) {
  return rasterize_bwd_impl(stream, C, N, n_isects, packed, backgrounds, width, height, tile_size, tile_width,
                            tile_height, offsets, flatten_ids, render_alphas, last_ids, v_render_colors,
                            v_render_alphas, packed_grad, v_means2d, v_conics, v_colors, v_opacities, emit_slot,
                            row_cum, partials, nullptr);
}

// Device-count form of the slot mode (one 64 B partial line per intersection, summed by the caller):
// `capacity` sizes `partials`, the true count is read on the device (see clmgs_isect2_emit_sort_dev).
extern "C" int clmgs_rasterize_bwd_dev(void* stream, int C, int N, int64_t capacity, const int64_t* n_isects_dev,
                                       const void* packed, const float* backgrounds, int width, int height,
                                       int tile_size, int tile_width, int tile_height, const int32_t* offsets,
                                       const int32_t* flatten_ids, const float* render_alphas,
                                       const int32_t* last_ids, const float* v_render_colors,
                                       const float* v_render_alphas, const int32_t* emit_slot,
                                       const int64_t* row_cum, void* partials) {
  CLMGS_CHECK_ARG(n_isects_dev && capacity > 0 && emit_slot && partials);
  return rasterize_bwd_impl(stream, C, N, capacity, packed, backgrounds, width, height, tile_size, tile_width,
                            tile_height, offsets, flatten_ids, render_alphas, last_ids, v_render_colors,
                            v_render_alphas, nullptr, nullptr, nullptr, nullptr, nullptr, emit_slot, row_cum,
                            partials, n_isects_dev);
}
