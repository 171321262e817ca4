// Tile-major binning (gfx950, round 5): the intersection lists of gsplat.isect_tiles + isect_offset_encode
// (strategies/base_engine.py:175-186) WITHOUT a global sort.
//
// The two-level route (isect.hip) sorts the V visible rows by depth (4 radix passes), emits the (tile, row) pairs in
// that order and sorts them again on the tile-id bits (2 passes): 23 dependent launches of 5-130 us, 0.85 ms per 4K
// camera although the lists are only 0.6 GB of traffic -- a latency chain, not a bandwidth problem.  The order it
// produces is (tile, depth bits, row index): stable LSD passes from an identity payload break depth ties by row index.
// That composite key is a STRICT total order (a row meets a tile once), so the same lists can be built tile by tile:
//
//   front   rows kernel      per row: tile box, exact tile mask, emitted count; one atomicAdd per emitted (row, tile) into
//                            the tile's counter; row_cum = scan of the emitted counts in ROW order (slot ranges of the
//                            atomic-free backward) -- scan finished by one small launch, which also leaves the totals
//   back    tile scan        exclusive scan of the tile counters = `offsets`, copied into the tiles' cursors
//           scatter          per emitted (row, tile): position = atomicAdd(cursor[tile], 1); one 16 B record
//                            {depth bits, row, slot} stored there (any order inside the tile)
//           tile sort        one wavefront per tile sorts its records by (depth bits, row) in LDS (a bitonic network in
//                            its all-ascending form, so no padding to a power of two is needed) and writes flatten_ids /
//                            emit_slot; tiles with more than 1024 entries go to a second, 256-thread kernel (LDS up to
//                            8192 entries, in global memory beyond)
//
// 7 launches instead of 23, ~0.2 GB of extra traffic for the records, and every list element for element what the
// two-level route produces (tests/test_gpu_ops.py).  Device-count form as in isect.hip: `capacity` sizes the buffers, the
// true count is read on the device; a count above the capacity leaves lists that are memory-safe but incomplete, and
// the caller redoes the camera (fused.camera_verify).
#include <stdlib.h>

#include "common.h"
#include "gs_math.h"
#include "isect_math.h"
#include "radix.h"

namespace clmgs {

constexpr int I3_ROUNDS = 4;
constexpr int I3_CHUNK = 256 * I3_ROUNDS;  // rows per block of the rows / scatter kernels
constexpr int I3_BIG = 64;                 // rows whose box holds more tiles than this are expanded by the whole block
constexpr int I3_SCAN_BLOCKS = 64;         // slices of the tile-counter scan

__host__ __device__ static inline int i3_chunks(int V) { return (V + I3_CHUNK - 1) / I3_CHUNK; }

struct RowBox { int x0, y0, bw, nt; unsigned long long mask; };

__device__ __forceinline__ RowBox unpack_row(unsigned long long b, unsigned long long m) {
  RowBox r;
  r.x0 = (int)(b & 0xFFFF); r.y0 = (int)((b >> 16) & 0xFFFF);
  const int x1 = (int)((b >> 32) & 0xFFFF), y1 = (int)(b >> 48);
  r.bw = x1 - r.x0; r.nt = r.bw * (y1 - r.y0);
  r.mask = r.nt > 64 ? ~0ull : (m & (r.nt == 64 ? ~0ull : ((1ull << r.nt) - 1ull)));
  return r;
}

// The rows of a block are 1024 consecutive rows of a Z-ordered table: a patch of ground whose tile boxes fall into a small
// WINDOW of the tile grid.  Both kernels that touch the per-tile counters aggregate inside the block first -- LDS
// atomics on a window of at most I3_WIN tiles -- and go to the global counters once per (block, tile): measured with one
// global atomic per (row, tile) the neighbouring rows of a wave serialise on the same few counters (316 us for 9.3 M
// atomics).  A block whose window is larger (rows in no spatial order) falls back to the global atomics.
constexpr int I3_WIN = 4096;

struct I3Window { int x0, y0, w, h; bool lds; };

// window of the boxes with <= I3_BIG tiles of this block's rows (block-wide min / max; two barriers)
__device__ __forceinline__ I3Window block_window(int x0, int y0, int x1, int y1, int* red /*[4][4] LDS*/) {
  // lanes without a box contribute the neutral element
  int v0 = x0, v1 = y0, v2 = -x1, v3 = -y1;  // min of all four
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    v0 = min(v0, __shfl_xor(v0, o, 64)); v1 = min(v1, __shfl_xor(v1, o, 64));
    v2 = min(v2, __shfl_xor(v2, o, 64)); v3 = min(v3, __shfl_xor(v3, o, 64));
  }
  const int wid = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[wid * 4 + 0] = v0; red[wid * 4 + 1] = v1; red[wid * 4 + 2] = v2; red[wid * 4 + 3] = v3; }
  __syncthreads();
  I3Window W;
  W.x0 = min(min(red[0], red[4]), min(red[8], red[12]));
  W.y0 = min(min(red[1], red[5]), min(red[9], red[13]));
  const int X1 = -min(min(red[2], red[6]), min(red[10], red[14]));
  const int Y1 = -min(min(red[3], red[7]), min(red[11], red[15]));
  W.w = max(X1 - W.x0, 0); W.h = max(Y1 - W.y0, 0);
  W.lds = (long long)W.w * W.h <= I3_WIN;
  return W;
}

// rows kernel: box / mask / emitted count of every row, tile counters, block-relative row_cum + block totals
__global__ void __launch_bounds__(256)
isect3_rows_kernel(int V, const int32_t* __restrict__ radii, const float* __restrict__ means2d,
                   const float4* __restrict__ packed, float tile_size, int tile_w, int tile_h,
                   unsigned long long* __restrict__ box_by_row, uint32_t* __restrict__ tile_cnt,
                   int64_t* __restrict__ row_cum, int64_t* __restrict__ blk_tot, int64_t* __restrict__ blk_ref) {
  __shared__ long long wsum[4];
  __shared__ int big_rows[I3_CHUNK];
  __shared__ int n_big;
  __shared__ unsigned long long rsum[4];
  __shared__ int red[16];
  __shared__ uint32_t win[I3_WIN];
  const int tid = threadIdx.x, chunk = blockIdx.x;
  if (tid == 0) n_big = 0;
  for (int q = tid; q < I3_WIN; q += 256) win[q] = 0u;
  __syncthreads();
  long long c[I3_ROUNDS];
  unsigned long long bx[I3_ROUNDS], mk[I3_ROUNDS];
  unsigned long long ref = 0ull;
  int wx0 = 1 << 30, wy0 = 1 << 30, wx1 = -(1 << 30), wy1 = -(1 << 30);
#pragma unroll
  for (int r = 0; r < I3_ROUNDS; ++r) {
    const int i = chunk * I3_CHUNK + r * 256 + tid;
    c[r] = 0; bx[r] = 0ull; mk[r] = 0ull;
    if (i < V) {
      const int rad = radii[i];
      unsigned long long b = 0ull, m = ~0ull;
      int cnt = 0;
      if (rad > 0) {
        const float2 mm = *reinterpret_cast<const float2*>(means2d + 2 * (size_t)i);
        const TileBox tb = tile_box(mm.x, mm.y, (float)rad, tile_size, tile_w, tile_h);
        b = pack_box(tb);
        if (packed) m = exact_tile_mask(packed + 4 * (size_t)i, tb.x0, tb.y0, tb.x1, tb.y1);
        const RowBox rb = unpack_row(b, m);
        ref += (unsigned long long)rb.nt;
        if (rb.nt > I3_BIG) {
          cnt = rb.nt;
          big_rows[atomicAdd(&n_big, 1)] = i;
        } else {
          cnt = __popcll(rb.mask);
          if (cnt) {
            bx[r] = b; mk[r] = rb.mask;
            wx0 = min(wx0, tb.x0); wy0 = min(wy0, tb.y0); wx1 = max(wx1, tb.x1); wy1 = max(wy1, tb.y1);
          }
        }
      }
      box_by_row[2 * (size_t)i] = cnt > 0 ? b : 0ull;
      box_by_row[2 * (size_t)i + 1] = m;
      c[r] = cnt;
    }
  }
  const I3Window W = block_window(wx0, wy0, wx1, wy1, red);
#pragma unroll
  for (int r = 0; r < I3_ROUNDS; ++r) {
    unsigned long long mm2 = mk[r];
    if (!mm2) continue;
    const RowBox rb = unpack_row(bx[r], mm2);
    while (mm2) {
      const int t = __ffsll((long long)mm2) - 1;
      mm2 &= mm2 - 1ull;
      const int ty = rb.y0 + t / rb.bw, tx = rb.x0 + t % rb.bw;
      if (W.lds) atomicAdd(&win[(ty - W.y0) * W.w + (tx - W.x0)], 1u);
      else atomicAdd(&tile_cnt[ty * tile_w + tx], 1u);
    }
  }
  {  // inclusive scan of the emitted counts in ROW order (block-relative; isect3_rows_finish_kernel adds the offsets)
    long long inc[I3_ROUNDS];
    long long carry = 0;
#pragma unroll
    for (int r = 0; r < I3_ROUNDS; ++r) {
      const long long incl = block_incl_scan_i64(c[r], wsum);
      inc[r] = carry + incl;
      carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    if (tid == 0) blk_tot[chunk] = carry;
#pragma unroll
    for (int r = 0; r < I3_ROUNDS; ++r) {
      const int i = chunk * I3_CHUNK + r * 256 + tid;
      if (i < V) row_cum[i] = inc[r];
    }
  }
  ref = (unsigned long long)wave_sum_i64((long long)ref);
  __syncthreads();  // (also: every LDS atomic of the window has been issued)
  if ((tid & 63) == 0) rsum[tid >> 6] = ref;
  if (W.lds) {
    const int nwin = W.w * W.h;
    for (int q = tid; q < nwin; q += 256) {
      const uint32_t v = win[q];
      if (v) atomicAdd(&tile_cnt[(W.y0 + q / W.w) * tile_w + W.x0 + q % W.w], v);
    }
  }
  __syncthreads();
  if (tid == 0) blk_ref[chunk] = (int64_t)(rsum[0] + rsum[1] + rsum[2] + rsum[3]);
  // rows with large boxes (every tile of the box is emitted): the block counts them together
  const int nb = n_big;
  for (int q = 0; q < nb; ++q) {
    const int i = big_rows[q];
    const RowBox rb = unpack_row(box_by_row[2 * (size_t)i], ~0ull);
    for (int t = tid; t < rb.nt; t += 256) {
      const int ty = t / rb.bw, tx = t - ty * rb.bw;
      atomicAdd(&tile_cnt[(rb.y0 + ty) * tile_w + rb.x0 + tx], 1u);
    }
  }
}

// row_cum += the totals of the blocks before; the last block leaves totals = {emitted, un-culled}
__global__ void __launch_bounds__(256)
isect3_rows_finish_kernel(int V, int64_t* __restrict__ row_cum, const int64_t* __restrict__ blk_tot,
                          const int64_t* __restrict__ blk_ref, int64_t* __restrict__ totals, int n_tiles,
                          const uint32_t* __restrict__ tile_cnt, int64_t* __restrict__ slice_tot) {
  __shared__ long long wsum[4];
  const int chunk = blockIdx.x, tid = threadIdx.x;
  // (the tile counters are complete: the first blocks also leave the totals of the I3_SCAN_BLOCKS slices the tile scan of
  //  the second half works on, so that no block of that scan has to walk more than its own slice)
  for (int sl = chunk; sl < I3_SCAN_BLOCKS; sl += gridDim.x) {
    const int per = (n_tiles + I3_SCAN_BLOCKS - 1) / I3_SCAN_BLOCKS;
    const int t_lo = min(n_tiles, sl * per), t_hi = min(n_tiles, t_lo + per);
    long long sum = 0;
    for (int t = t_lo + tid; t < t_hi; t += 256) sum += tile_cnt[t];
    sum = wave_sum_i64(sum);
    __syncthreads();
    if ((tid & 63) == 0) wsum[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) slice_tot[sl] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  long long part = 0;
  for (int j = tid; j < chunk; j += 256) part += blk_tot[j];
  part = wave_sum_i64(part);
  if ((tid & 63) == 0) wsum[tid >> 6] = part;
  __syncthreads();
  const long long off = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  if (chunk > 0) {
#pragma unroll
    for (int r = 0; r < I3_ROUNDS; ++r) {
      const int i = chunk * I3_CHUNK + r * 256 + tid;
      if (i < V) row_cum[i] += off;
    }
  }
  if (chunk == (int)gridDim.x - 1) {
    long long ref = 0;
    for (int j = tid; j < (int)gridDim.x; j += 256) ref += blk_ref[j];
    ref = wave_sum_i64(ref);
    __syncthreads();
    if ((tid & 63) == 0) wsum[tid >> 6] = ref;
    __syncthreads();
    if (tid == 0) { totals[0] = off + blk_tot[chunk]; totals[1] = wsum[0] + wsum[1] + wsum[2] + wsum[3]; }
  }
}

// exclusive scan of the tile counters = `offsets` (clamped to the number of entries the buffers hold) and the tiles'
// cursors.  I3_SCAN_BLOCKS blocks, each owning a contiguous slice of the tiles: a block first sums every counter before
// its slice from the slice totals the front half left (isect3_rows_finish_kernel), then scans its own slice -- one
// launch, no dependent chain of sweeps (a single 256-thread block walking all 62 208 counters took 91 us).
__global__ void __launch_bounds__(256)
isect3_tile_scan_kernel(int n_tiles, const uint32_t* __restrict__ tile_cnt, int64_t n, const int64_t* __restrict__ n_dev,
                        int32_t* __restrict__ offsets, uint32_t* __restrict__ cursor, uint32_t* __restrict__ big_count,
                        const int64_t* __restrict__ slice_tot) {
  __shared__ long long wsum[4];
  if (n_dev) n = min(n, *n_dev);
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 0) *big_count = 0u;  // work list of isect3_sort_big_kernel, filled by the small sort
  const int per = (n_tiles + I3_SCAN_BLOCKS - 1) / I3_SCAN_BLOCKS;
  const int t_lo = min(n_tiles, (int)blockIdx.x * per), t_hi = min(n_tiles, t_lo + per);
  long long carry = 0;
  for (int sl = 0; sl < (int)blockIdx.x; ++sl) carry += slice_tot[sl];  // (block-uniform: scalar loads)
  for (int base = t_lo; base < t_hi; base += 256) {
    const int t = base + tid;
    const uint32_t v = t < t_hi ? tile_cnt[t] : 0u;
    const long long incl = block_incl_scan_i64((long long)v, wsum);
    const long long ex = carry + incl - v;
    if (t < t_hi) {
      offsets[t] = (int32_t)min(ex, (long long)n);
      cursor[t] = (uint32_t)ex;  // un-clamped: the scatter drops whatever lands at or beyond n
    }
    carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}

struct I3Rec { uint32_t depth; int32_t row; int32_t slot; int32_t pad; };

// scatter: every emitted (row, tile) pair takes the next free place of its tile.  Per block: count per window tile in
// LDS, ONE global atomic per (block, tile) reserves that many places, a second pass hands out the places with LDS atomics.
__global__ void __launch_bounds__(256)
isect3_scatter_kernel(int V, const unsigned long long* __restrict__ box_by_row, const float* __restrict__ depths,
                      const int64_t* __restrict__ row_cum, int tile_w, uint32_t* __restrict__ cursor, int64_t n,
                      const int64_t* __restrict__ n_dev, int4* __restrict__ recs) {
  __shared__ int big_rows[I3_CHUNK];
  __shared__ int n_big;
  __shared__ int red[16];
  __shared__ uint32_t win[I3_WIN];   // pass 1: entries per window tile; pass 2: the next free place of the tile
  if (n_dev) n = min(n, *n_dev);
  const int tid = threadIdx.x, chunk = blockIdx.x;
  if (tid == 0) n_big = 0;
  for (int q = tid; q < I3_WIN; q += 256) win[q] = 0u;
  __syncthreads();
  unsigned long long bx[I3_ROUNDS], mk[I3_ROUNDS];
  int wx0 = 1 << 30, wy0 = 1 << 30, wx1 = -(1 << 30), wy1 = -(1 << 30);
#pragma unroll
  for (int r = 0; r < I3_ROUNDS; ++r) {
    const int i = chunk * I3_CHUNK + r * 256 + tid;
    bx[r] = 0ull; mk[r] = 0ull;
    if (i >= V) continue;
    const unsigned long long b = box_by_row[2 * (size_t)i];
    if (b == 0ull) continue;
    const RowBox rb = unpack_row(b, box_by_row[2 * (size_t)i + 1]);
    if (rb.nt > I3_BIG) { big_rows[atomicAdd(&n_big, 1)] = i; continue; }
    bx[r] = b; mk[r] = rb.mask;
    wx0 = min(wx0, rb.x0); wy0 = min(wy0, rb.y0);
    wx1 = max(wx1, rb.x0 + rb.bw); wy1 = max(wy1, rb.y0 + rb.nt / rb.bw);
  }
  const I3Window W = block_window(wx0, wy0, wx1, wy1, red);
  if (W.lds) {
#pragma unroll
    for (int r = 0; r < I3_ROUNDS; ++r) {
      unsigned long long mm = mk[r];
      if (!mm) continue;
      const RowBox rb = unpack_row(bx[r], mm);
      while (mm) {
        const int t = __ffsll((long long)mm) - 1;
        mm &= mm - 1ull;
        atomicAdd(&win[(rb.y0 + t / rb.bw - W.y0) * W.w + (rb.x0 + t % rb.bw - W.x0)], 1u);
      }
    }
    __syncthreads();
    const int nwin = W.w * W.h;
    for (int q = tid; q < nwin; q += 256) {
      const uint32_t v = win[q];
      if (v) win[q] = atomicAdd(&cursor[(W.y0 + q / W.w) * tile_w + W.x0 + q % W.w], v);  // the block's places in that tile
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < I3_ROUNDS; ++r) {
    unsigned long long mm = mk[r];
    if (!mm) continue;
    const int i = chunk * I3_CHUNK + r * 256 + tid;
    const RowBox rb = unpack_row(bx[r], mm);
    const uint32_t dk = (uint32_t)__float_as_int(depths[i]);
    int slot = i > 0 ? (int)row_cum[i - 1] : 0;
    while (mm) {
      const int t = __ffsll((long long)mm) - 1;
      mm &= mm - 1ull;
      const int ty = rb.y0 + t / rb.bw, tx = rb.x0 + t % rb.bw;
      const uint32_t pos = W.lds ? atomicAdd(&win[(ty - W.y0) * W.w + (tx - W.x0)], 1u)
                                 : atomicAdd(&cursor[ty * tile_w + tx], 1u);
      if ((int64_t)pos < n) recs[pos] = make_int4((int)dk, i, slot, 0);
      ++slot;
    }
  }
  __syncthreads();
  const int nb = n_big;
  for (int q = 0; q < nb; ++q) {
    const int i = big_rows[q];
    const RowBox rb = unpack_row(box_by_row[2 * (size_t)i], ~0ull);
    const uint32_t dk = (uint32_t)__float_as_int(depths[i]);
    const int slot0 = i > 0 ? (int)row_cum[i - 1] : 0;
    for (int t = tid; t < rb.nt; t += 256) {
      const int ty = t / rb.bw, tx = t - ty * rb.bw;
      const uint32_t pos = atomicAdd(&cursor[(rb.y0 + ty) * tile_w + rb.x0 + tx], 1u);
      if ((int64_t)pos < n) recs[pos] = make_int4((int)dk, i, slot0 + t, 0);
    }
  }
}

// Bitonic sorting network in its all-ascending form (the first step of every merge compares i with the mirror image
// of i inside the merged block, the others with i ^ j): every compare-exchange puts the smaller key at the lower index,
// so positions >= len behave as +infinity without being stored -- any length sorts without padding.
// keys/vals may live in LDS or in global memory; `sync` orders the steps.
template <int THREADS, typename SyncFn>
__device__ __forceinline__ void bitonic_sort_kv(unsigned long long* keys, int32_t* vals, int len, int tid, SyncFn sync) {
  int P = 1;
  while (P < len) P <<= 1;
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      sync();
      for (int c = tid; c < (P >> 1); c += THREADS) {
        // c-th compare-exchange of this step: lower index lo, partner hi > lo
        const int lo = ((c & ~(j - 1)) << 1) | (c & (j - 1));
        const int hi = (j == (k >> 1)) ? (lo ^ (k - 1)) : (lo | j);
        // (first step of the merge: lo runs over the lower half of every k-block -- the formula above enumerates exactly
        //  those for j = k/2 -- and its partner is the mirror image inside the block)
        if (hi < len) {
          const unsigned long long a = keys[lo], b = keys[hi];
          if (b < a) {
            keys[lo] = b; keys[hi] = a;
            const int32_t va = vals[lo], vb = vals[hi];
            vals[lo] = vb; vals[hi] = va;
          }
        }
      }
    }
  }
  sync();
}

constexpr int I3_SMALL = 1024;   // entries a one-wave workgroup sorts in LDS
constexpr int I3_LDS_BIG = 8192;  // entries the 256-thread kernel sorts in LDS; longer lists are sorted in global memory

__device__ __forceinline__ void i3_write_out(int tile, int s, int len, const unsigned long long* keys, const int32_t* vals,
                                             int tid, int threads, int32_t* __restrict__ flatten_ids,
                                             int32_t* __restrict__ emit_slot, int64_t* __restrict__ isect_ids) {
  for (int q = tid; q < len; q += threads) {
    const unsigned long long k = keys[q];
    flatten_ids[s + q] = (int32_t)(k & 0xFFFFFFFFull);
    if (emit_slot) emit_slot[s + q] = vals[q];
    if (isect_ids) isect_ids[s + q] = ((int64_t)tile << 32) | (int64_t)(k >> 32);
  }
}

// One wavefront per tile, lists of LO < len <= CAP entries sorted in CAP x 12 B of LDS by the bitonic network.  Two
// instances: (256, 512] and (512, 1024] (lists of up to 256 entries: the rank sort below).  Lists longer than 1 024 are put
// on the work list of the 256-thread kernel (by the last instance).
template <int LO, int CAP>
__global__ void __launch_bounds__(64)
isect3_sort_small_kernel(int n_tiles, const int32_t* __restrict__ offsets, int64_t n, const int64_t* __restrict__ n_dev,
                         const int4* __restrict__ recs, int32_t* __restrict__ flatten_ids,
                         int32_t* __restrict__ emit_slot, int64_t* __restrict__ isect_ids,
                         uint32_t* __restrict__ big_count, int32_t* __restrict__ big_list) {
  __shared__ unsigned long long keys[CAP];
  __shared__ int32_t vals[CAP];
  if (n_dev) n = min(n, *n_dev);
  const int tile = (int)xcd_remap(blockIdx.x, (unsigned)n_tiles);
  const int tid = threadIdx.x;
  const int s = offsets[tile];
  const int e = (tile == n_tiles - 1) ? (int)n : offsets[tile + 1];
  const int len = e - s;
  if (CAP == I3_SMALL && len > I3_SMALL && tid == 0) big_list[atomicAdd(big_count, 1u)] = tile;
  if (len <= LO || len > CAP) return;
  for (int q = tid; q < len; q += 64) {
    const int4 r = recs[s + q];
    keys[q] = ((unsigned long long)(uint32_t)r.x << 32) | (unsigned long long)(uint32_t)r.y;
    vals[q] = r.z;
  }
  bitonic_sort_kv<64>(keys, vals, len, tid, [] { __syncthreads(); });
  i3_write_out(tile, s, len, keys, vals, tid, 64, flatten_ids, emit_slot, isect_ids);
}

// Lists of at most 256 entries (almost every tile of a 4K camera of the slab scene: 149 entries on average): RANK sort.
// Every lane keeps up to four records in registers; the keys go to LDS once, and every lane counts, for each of its
// records, the keys of the tile that are smaller -- one broadcast LDS read (all lanes, one address) and two VALU
// operations per (key, record).  The count IS the record's place (the keys are distinct), so the outputs are written
// straight to it: no network of dependent LDS round trips (the bitonic instance of this class took 185 us per camera,
// LDS-bandwidth bound: 36 steps x 128 compare-exchanges x ~32 B), 2 KB of LDS, len^2 / 64 x ~2.3 instructions.
template <int E>
__device__ __forceinline__ void rank_sort_tile(unsigned long long* keys, int tile, int s, int len, int tid,
                                               const int4* __restrict__ recs, int32_t* __restrict__ flatten_ids,
                                               int32_t* __restrict__ emit_slot, int64_t* __restrict__ isect_ids) {
  unsigned long long key[E];
  int32_t slot[E];
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int q = tid + 64 * r;
    key[r] = ~0ull; slot[r] = 0;
    if (q < len) {
      const int4 rc = recs[s + q];
      key[r] = ((unsigned long long)(uint32_t)rc.x << 32) | (unsigned long long)(uint32_t)rc.y;
      slot[r] = rc.z;
      keys[q] = key[r];
    }
  }
  __syncthreads();
  int cnt[E];
#pragma unroll
  for (int r = 0; r < E; ++r) cnt[r] = 0;
  int j = 0;
  for (; j + 4 <= len; j += 4) {
    const unsigned long long k0 = keys[j], k1 = keys[j + 1], k2 = keys[j + 2], k3 = keys[j + 3];
#pragma unroll
    for (int r = 0; r < E; ++r)
      cnt[r] += (int)(k0 < key[r]) + (int)(k1 < key[r]) + (int)(k2 < key[r]) + (int)(k3 < key[r]);
  }
  for (; j < len; ++j) {
    const unsigned long long k0 = keys[j];
#pragma unroll
    for (int r = 0; r < E; ++r) cnt[r] += (int)(k0 < key[r]);
  }
#pragma unroll
  for (int r = 0; r < E; ++r) {
    if (tid + 64 * r < len) {
      const int o = s + cnt[r];
      flatten_ids[o] = (int32_t)(key[r] & 0xFFFFFFFFull);
      if (emit_slot) emit_slot[o] = slot[r];
      if (isect_ids) isect_ids[o] = ((int64_t)tile << 32) | (int64_t)(key[r] >> 32);
    }
  }
}

__global__ void __launch_bounds__(64)
isect3_sort_rank_kernel(int n_tiles, const int32_t* __restrict__ offsets, int64_t n, const int64_t* __restrict__ n_dev,
                        const int4* __restrict__ recs, int32_t* __restrict__ flatten_ids,
                        int32_t* __restrict__ emit_slot, int64_t* __restrict__ isect_ids) {
  constexpr int CAP = 256;
  __shared__ unsigned long long keys[CAP];
  if (n_dev) n = min(n, *n_dev);
  const int tile = (int)xcd_remap(blockIdx.x, (unsigned)n_tiles);
  const int tid = threadIdx.x;
  const int s = offsets[tile];
  const int e = (tile == n_tiles - 1) ? (int)n : offsets[tile + 1];
  const int len = e - s;
  if (len <= 0 || len > CAP) return;
  // records per lane by list length (wave-uniform): a list of 149 entries pays for three records per lane, not four
  if (len <= 64) rank_sort_tile<1>(keys, tile, s, len, tid, recs, flatten_ids, emit_slot, isect_ids);
  else if (len <= 128) rank_sort_tile<2>(keys, tile, s, len, tid, recs, flatten_ids, emit_slot, isect_ids);
  else if (len <= 192) rank_sort_tile<3>(keys, tile, s, len, tid, recs, flatten_ids, emit_slot, isect_ids);
  else rank_sort_tile<4>(keys, tile, s, len, tid, recs, flatten_ids, emit_slot, isect_ids);
}

// tiles with more than I3_SMALL entries (deep lists: heavy-tailed scenes, degenerate inputs), from the work list: one
// 256-thread workgroup per tile; in LDS up to I3_LDS_BIG entries, in global memory (key / value scratch of the caller) beyond
__global__ void __launch_bounds__(256)
isect3_sort_big_kernel(int n_tiles, const int32_t* __restrict__ offsets, int64_t n, const int64_t* __restrict__ n_dev,
                       const int4* __restrict__ recs, unsigned long long* __restrict__ gkeys, int32_t* __restrict__ gvals,
                       int32_t* __restrict__ flatten_ids, int32_t* __restrict__ emit_slot,
                       int64_t* __restrict__ isect_ids, const uint32_t* __restrict__ big_count,
                       const int32_t* __restrict__ big_list) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  if (n_dev) n = min(n, *n_dev);
  const int tid = threadIdx.x;
  const int n_big = (int)*big_count;
  for (int w = blockIdx.x; w < n_big; w += gridDim.x) {
    const int tile = big_list[w];
    const int s = offsets[tile];
    const int e = (tile == n_tiles - 1) ? (int)n : offsets[tile + 1];
    const int len = e - s;
    __syncthreads();
    unsigned long long* keys;
    int32_t* vals;
    if (len <= I3_LDS_BIG) {
      keys = reinterpret_cast<unsigned long long*>(dyn);
      vals = reinterpret_cast<int32_t*>(dyn + sizeof(unsigned long long) * I3_LDS_BIG);
    } else {
      keys = gkeys + s;
      vals = gvals + s;
    }
    for (int q = tid; q < len; q += 256) {
      const int4 r = recs[s + q];
      keys[q] = ((unsigned long long)(uint32_t)r.x << 32) | (unsigned long long)(uint32_t)r.y;
      vals[q] = r.z;
    }
    bitonic_sort_kv<256>(keys, vals, len, tid, [] { __threadfence_block(); __syncthreads(); });
    i3_write_out(tile, s, len, keys, vals, tid, 256, flatten_ids, emit_slot, isect_ids);
  }
}

}  // namespace clmgs

using namespace clmgs;

static inline size_t i3_front_layout(int V, int n_tiles, size_t* o_cnt, size_t* o_tot, size_t* o_ref) {
  size_t o = align_up((size_t)V * 16, 256);
  *o_cnt = o; o += align_up((size_t)n_tiles * 4, 256) + align_up((size_t)I3_SCAN_BLOCKS * 8, 256);  // counters | slice totals
  *o_tot = o; o += align_up((size_t)i3_chunks(V) * 8, 256);
  *o_ref = o; o += align_up((size_t)i3_chunks(V) * 8, 256);
  return o + 256;
}

extern "C" size_t clmgs_isect3_front_temp_bytes(int V, int n_tiles) {
  if (V <= 0) return 256;
  size_t a, b, c;
  return i3_front_layout(V, n_tiles, &a, &b, &c);
}

extern "C" int clmgs_isect3_front(void* stream, int V, const float* means2d, const int32_t* radii, int tile_size,
                                  int tile_width, int tile_height, const void* packed, int64_t* totals,
                                  int64_t* row_cum, void* temp, size_t temp_bytes) {
  CLMGS_CHECK_ARG(V >= 0 && tile_size > 0 && tile_width > 0 && tile_height > 0);
  CLMGS_CHECK_ARG(tile_width < 65536 && tile_height < 65536 && (int64_t)tile_width * tile_height < ((int64_t)1 << 31));
  if (V == 0) return 0;
  const int n_tiles = tile_width * tile_height;
  CLMGS_CHECK_ARG(means2d && radii && totals && row_cum && temp);
  CLMGS_CHECK_ARG(!packed || tile_size == 16);
  CLMGS_CHECK_ARG(temp_bytes >= clmgs_isect3_front_temp_bytes(V, n_tiles));
  hipStream_t s = (hipStream_t)stream;
  size_t o_cnt, o_tot, o_ref;
  i3_front_layout(V, n_tiles, &o_cnt, &o_tot, &o_ref);
  char* base = (char*)temp;
  unsigned long long* box_by_row = (unsigned long long*)base;
  uint32_t* tile_cnt = (uint32_t*)(base + o_cnt);
  int64_t* blk_tot = (int64_t*)(base + o_tot);
  int64_t* blk_ref = (int64_t*)(base + o_ref);
  CLMGS_HIP(hipMemsetAsync(tile_cnt, 0, (size_t)n_tiles * 4, s));
  const int nck = i3_chunks(V);
  hipLaunchKernelGGL(isect3_rows_kernel, dim3(nck), dim3(256), 0, s, V, radii, means2d, (const float4*)packed,
                     (float)tile_size, tile_width, tile_height, box_by_row, tile_cnt, row_cum, blk_tot, blk_ref);
  int64_t* slice_tot = (int64_t*)(base + o_cnt + align_up((size_t)n_tiles * 4, 256));
  hipLaunchKernelGGL(isect3_rows_finish_kernel, dim3(nck), dim3(256), 0, s, V, row_cum, blk_tot, blk_ref, totals, n_tiles,
                     (const uint32_t*)tile_cnt, slice_tot);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

static inline size_t i3_bin_layout(int64_t n, int n_tiles, size_t* o_cur, size_t* o_keys, size_t* o_vals) {
  size_t o = align_up((size_t)n * 16, 256);
  *o_cur = o; o += 2 * align_up((size_t)n_tiles * 4, 256) + 256;  // cursors | work list of the long tiles | its counter
  *o_keys = o; o += align_up((size_t)n * 8, 256);
  *o_vals = o; o += align_up((size_t)n * 4, 256);
  return o + 256;
}

extern "C" size_t clmgs_isect3_bin_temp_bytes(int64_t n_isects, int n_tiles) {
  if (n_isects <= 0) return 256;
  size_t a, b, c;
  return i3_bin_layout(n_isects, n_tiles, &a, &b, &c);
}

static int isect3_bin_impl(void* stream, int V, int64_t n_isects, const int64_t* n_dev, const float* depths,
                           int tile_width, int tile_height, const int64_t* row_cum, const void* front_temp,
                           int32_t* flatten_ids, int32_t* offsets, int64_t* isect_ids, int32_t* emit_slot, void* temp,
                           size_t temp_bytes) {
  CLMGS_CHECK_ARG(V >= 0 && n_isects >= 0 && offsets && tile_width > 0 && tile_height > 0);
  hipStream_t s = (hipStream_t)stream;
  const int n_tiles = tile_width * tile_height;
  if (n_isects == 0 || V == 0) {
    CLMGS_HIP(hipMemsetAsync(offsets, 0, sizeof(int32_t) * (size_t)n_tiles, s));
    return 0;
  }
  CLMGS_CHECK_ARG(n_isects < ((int64_t)1 << 31));
  CLMGS_CHECK_ARG(depths && row_cum && front_temp && flatten_ids && temp);
  CLMGS_CHECK_ARG(temp_bytes >= clmgs_isect3_bin_temp_bytes(n_isects, n_tiles));
  size_t f_cnt, f_tot, f_ref;
  i3_front_layout(V, n_tiles, &f_cnt, &f_tot, &f_ref);
  const unsigned long long* box_by_row = (const unsigned long long*)front_temp;
  const uint32_t* tile_cnt = (const uint32_t*)((const char*)front_temp + f_cnt);
  size_t o_cur, o_keys, o_vals;
  i3_bin_layout(n_isects, n_tiles, &o_cur, &o_keys, &o_vals);
  char* base = (char*)temp;
  int4* recs = (int4*)base;
  uint32_t* cursor = (uint32_t*)(base + o_cur);
  unsigned long long* gkeys = (unsigned long long*)(base + o_keys);
  int32_t* gvals = (int32_t*)(base + o_vals);
  int32_t* big_list = (int32_t*)(base + o_cur + align_up((size_t)n_tiles * 4, 256));
  uint32_t* big_count = (uint32_t*)(base + o_cur + 2 * align_up((size_t)n_tiles * 4, 256));
  const int64_t* slice_tot = (const int64_t*)((const char*)front_temp + f_cnt + align_up((size_t)n_tiles * 4, 256));
  hipLaunchKernelGGL(isect3_tile_scan_kernel, dim3(I3_SCAN_BLOCKS), dim3(256), 0, s, n_tiles, tile_cnt, n_isects, n_dev,
                     offsets, cursor, big_count, slice_tot);
  hipLaunchKernelGGL(isect3_scatter_kernel, dim3(i3_chunks(V)), dim3(256), 0, s, V, box_by_row, depths, row_cum,
                     tile_width, cursor, n_isects, n_dev, recs);
  hipLaunchKernelGGL(isect3_sort_rank_kernel, dim3(n_tiles), dim3(64), 0, s, n_tiles, (const int32_t*)offsets, n_isects,
                     n_dev, (const int4*)recs, flatten_ids, emit_slot, isect_ids);
  hipLaunchKernelGGL((isect3_sort_small_kernel<256, 512>), dim3(n_tiles), dim3(64), 0, s, n_tiles, (const int32_t*)offsets,
                     n_isects, n_dev, (const int4*)recs, flatten_ids, emit_slot, isect_ids, big_count, big_list);
  hipLaunchKernelGGL((isect3_sort_small_kernel<512, I3_SMALL>), dim3(n_tiles), dim3(64), 0, s, n_tiles,
                     (const int32_t*)offsets, n_isects, n_dev, (const int4*)recs, flatten_ids, emit_slot, isect_ids,
                     big_count, big_list);
  static bool lds_set = false;
  const size_t big_lds = (size_t)I3_LDS_BIG * 12;
  if (!lds_set) {
    CLMGS_HIP(hipFuncSetAttribute((const void*)isect3_sort_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)big_lds));
    lds_set = true;
  }
  hipLaunchKernelGGL(isect3_sort_big_kernel, dim3(min(n_tiles, 256)), dim3(256), big_lds, s, n_tiles,
                     (const int32_t*)offsets, n_isects, n_dev, (const int4*)recs, gkeys, gvals, flatten_ids, emit_slot,
                     isect_ids, (const uint32_t*)big_count, (const int32_t*)big_list);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_isect3_bin(void* stream, int V, int64_t n_isects, const float* depths, int tile_width,
                                int tile_height, const int64_t* row_cum, const void* front_temp, int32_t* flatten_ids,
                                int32_t* offsets, int64_t* isect_ids, int32_t* emit_slot, void* temp, size_t temp_bytes) {
  return isect3_bin_impl(stream, V, n_isects, nullptr, depths, tile_width, tile_height, row_cum, front_temp, flatten_ids,
                         offsets, isect_ids, emit_slot, temp, temp_bytes);
}

extern "C" int clmgs_isect3_bin_dev(void* stream, int V, int64_t capacity, const int64_t* n_isects_dev,
                                    const float* depths, int tile_width, int tile_height, const int64_t* row_cum,
                                    const void* front_temp, int32_t* flatten_ids, int32_t* offsets, int64_t* isect_ids,
                                    int32_t* emit_slot, void* temp, size_t temp_bytes) {
  CLMGS_CHECK_ARG(n_isects_dev && capacity > 0);
  return isect3_bin_impl(stream, V, capacity, n_isects_dev, depths, tile_width, tile_height, row_cum, front_temp,
                         flatten_ids, offsets, isect_ids, emit_slot, temp, temp_bytes);
}
