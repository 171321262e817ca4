// Tile binning: per-Gaussian tile counts, (tile | depth) keys, stable radix sort,
// per-tile start offsets (gfx950).  Replaces gsplat.isect_tiles and
// gsplat.isect_offset_encode (strategies/base_engine.py:175-186,
// strategies/no_offload/engine.py:75-84, strategies/clm_offload/engine.py:89-100).
// Sorting and scanning are the hand-written kernels of radix.h (no rocPRIM / hipCUB).
#include <stdlib.h>

#include "common.h"
#include "gs_math.h"
#include "radix.h"
#include "vis_math.h"
#include "isect_math.h"
#ifdef CLMGS_PROFILE_BUILD
#include "onesweep.h"  // round 4's one-launch-per-digit passes by decoupled look-back: measured slower, profiling builds only
#endif

namespace clmgs {
// Device error word of the look-back primitives (onesweep.h): bit 0 = scan look-back timed out, bit 1 = sort
// look-back timed out.  Read (and cleared) by clmgs_device_errors().
__device__ uint32_t g_dev_err;
uint32_t* device_error_word() {
  static thread_local uint32_t* p[16] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (!p[dev]) {
    void* a = nullptr;
    if (hipGetSymbolAddress(&a, HIP_SYMBOL(g_dev_err)) == hipSuccess) p[dev] = (uint32_t*)a;
  }
  return p[dev];
}

// CLMGS_LEGACY_BINNING=1: the round-3 chain (three launches per radix digit, three per scan) -- kept for A/B
// measurements and as the second, independent route of the equality tests.
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && e[0]) ? atoi(e) : dflt;
}
static int os_rounds(const char* name, int dflt) {
  const int r = env_int(name, dflt);
  return (r == 1 || r == 2 || r == 4 || r == 8) ? r : dflt;
}
// Routes through the binning chain.  The PRODUCT library has ONE (BIN_FUSED):
//   fused    three kernels per radix digit (multi-chunk histogram, segment-per-thread row scan, scatter); the two scans
//            folded into their producers + ONE finishing launch each; the tile ids are produced chunk by chunk with
//            coalesced stores by the kernel that also counts the first digit of the tile sort (isect2_emit_hist_kernel,
//            round 5); the last tile-sort pass writes flatten_ids / emit_slot directly.
// A profiling build (make PROFILE=1) can select the older ones per call with CLMGS_BINNING (the equality test of
// tests/test_gpu_ops.py runs in that build); all produce the same lists element for element:
//   r4       round 4's default: a thread-per-rank emit kernel (uncoalesced 4 + 8 B stores) + a separate histogram
//   lookback round 4's single-launch passes by decoupled look-back (onesweep.h) -- measured SLOWER on MI355X (a sort
//            pass of 3.3 M keys 60-79 us against 30 + 11 + 11 us, of 9.3 M keys 155-205 against 106 + 34 + 34;
//            DESIGN.md section 3)
//   legacy   the round-3 chain
enum { BIN_FUSED = 0, BIN_LOOKBACK = 1, BIN_LEGACY = 2, BIN_R4 = 3 };
static int binning_route() {
#ifdef CLMGS_PROFILE_BUILD
  const char* e = getenv("CLMGS_BINNING");
  if (e && e[0] == 'l' && e[1] == 'o') return BIN_LOOKBACK;
  if (e && e[0] == 'l' && e[1] == 'e') return BIN_LEGACY;
  if (e && e[0] == 'r' && e[1] == '4') return BIN_R4;
#endif
  return BIN_FUSED;
}
static bool legacy_binning() { return binning_route() == BIN_LEGACY; }

__global__ void __launch_bounds__(256)
isect_count_kernel(int64_t CN, const float* __restrict__ means2d, const int32_t* __restrict__ radii,
                   float tile_size, int tile_w, int tile_h, int32_t* __restrict__ tiles_per_gauss,
                   int64_t* __restrict__ cum) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < CN;
       i += (int64_t)gridDim.x * blockDim.x) {
    int cnt = 0;
    const int r = radii[i];
    if (r > 0) {
      const float2 m = *reinterpret_cast<const float2*>(means2d + 2 * i);
      const TileBox b = tile_box(m.x, m.y, (float)r, tile_size, tile_w, tile_h);
      cnt = (b.x1 - b.x0) * (b.y1 - b.y0);
    }
    tiles_per_gauss[i] = cnt;
    cum[i] = cnt;
  }
}

__global__ void __launch_bounds__(256)
isect_emit_kernel(int64_t CN, int N, const float* __restrict__ means2d,
                  const int32_t* __restrict__ radii, const float* __restrict__ depths,
                  const int64_t* __restrict__ cum, float tile_size, int tile_w, int tile_h,
                  int tile_bits, int64_t* __restrict__ keys, int32_t* __restrict__ vals) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < CN;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int r = radii[i];
    if (r <= 0) continue;
    const float2 m = *reinterpret_cast<const float2*>(means2d + 2 * i);
    const TileBox b = tile_box(m.x, m.y, (float)r, tile_size, tile_w, tile_h);
    int64_t cur = (i == 0) ? 0 : cum[i - 1];
    const int64_t cam = i / N;
    const int64_t depth_bits = (int64_t)(uint32_t)__float_as_int(depths[i]);
    const int64_t cam_part = cam << tile_bits;
    for (int ty = b.y0; ty < b.y1; ++ty) {
      for (int tx = b.x0; tx < b.x1; ++tx) {
        const int64_t tile = (int64_t)ty * tile_w + tx;
        keys[cur] = ((cam_part | tile) << 32) | depth_bits;
        vals[cur] = (int32_t)i;
        ++cur;
      }
    }
  }
}

__global__ void __launch_bounds__(256)
isect_offsets_kernel(int64_t n_isects, const int64_t* __restrict__ isect_ids, int n_tiles_total,
                     int n_tiles, int tile_bits, int32_t* __restrict__ offsets) {
  const int64_t mask = ((int64_t)1 << tile_bits) - 1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_isects;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = isect_ids[i] >> 32;
    const int cur = (int)((k >> tile_bits) * n_tiles + (k & mask));
    if (i == 0) {
      for (int t = 0; t <= cur; ++t) offsets[t] = 0;
    } else {
      const int64_t kp = isect_ids[i - 1] >> 32;
      const int prev = (int)((kp >> tile_bits) * n_tiles + (kp & mask));
      for (int t = prev + 1; t <= cur; ++t) offsets[t] = (int32_t)i;
    }
    if (i == n_isects - 1) {
      for (int t = cur + 1; t < n_tiles_total; ++t) offsets[t] = (int32_t)n_isects;
    }
  }
}

static inline int ilog2_floor(unsigned v) { int r = 0; while (v >>= 1) ++r; return r; }


}  // namespace clmgs

using namespace clmgs;

extern "C" size_t clmgs_isect_count_temp_bytes(int CN) {
  return scan_scratch_bytes(CN) + 256;
}

extern "C" int clmgs_isect_count(void* stream, int C, int N, const float* means2d,
                                 const int32_t* radii, int tile_size, int tile_width,
                                 int tile_height, int32_t* tiles_per_gauss, int64_t* cum,
                                 void* temp, size_t temp_bytes) {
  CLMGS_CHECK_ARG(C >= 1 && N >= 0 && tile_size > 0 && tile_width > 0 && tile_height > 0);
  const int64_t CN = (int64_t)C * N;
  if (CN == 0) return 0;
  CLMGS_CHECK_ARG(CN < ((int64_t)1 << 31));
  CLMGS_CHECK_ARG(means2d && radii && tiles_per_gauss && cum && temp);
  CLMGS_CHECK_ARG(temp_bytes >= clmgs_isect_count_temp_bytes((int)CN));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(isect_count_kernel, dim3(min(ceil_div(CN, 256), 256 * 16)), dim3(256), 0, s,
                     CN, means2d, radii, (float)tile_size, tile_width, tile_height,
                     tiles_per_gauss, cum);
  CLMGS_LAUNCH_CHECK();
  return inclusive_scan_i64(s, CN, cum, (int64_t*)temp);
}

extern "C" size_t clmgs_isect_sort_temp_bytes(int64_t n_isects) {
  if (n_isects <= 0) return 256;
  return align_up((size_t)n_isects * 8, 256) + 2 * align_up((size_t)n_isects * 4, 256) +
         radix_table_bytes(n_isects) + 256;
}

extern "C" int clmgs_isect_emit_sort(void* stream, int C, int N, int64_t n_isects,
                                     const float* means2d, const int32_t* radii,
                                     const float* depths, const int64_t* cum, int tile_size,
                                     int tile_width, int tile_height, int64_t* isect_ids,
                                     int32_t* flatten_ids, void* temp, size_t temp_bytes) {
  CLMGS_CHECK_ARG(C >= 1 && N >= 0 && n_isects >= 0);
  if (n_isects == 0) return 0;
  CLMGS_CHECK_ARG(n_isects < ((int64_t)1 << 31));
  CLMGS_CHECK_ARG(means2d && radii && depths && cum && isect_ids && flatten_ids && temp);
  CLMGS_CHECK_ARG(temp_bytes >= clmgs_isect_sort_temp_bytes(n_isects));
  hipStream_t s = (hipStream_t)stream;
  const int64_t CN = (int64_t)C * N;
  const int n_tiles = tile_width * tile_height;
  const int tile_bits = ilog2_floor((unsigned)n_tiles) + 1;
  const int cam_bits = ilog2_floor((unsigned)C) + 1;
  char* base = (char*)temp;
  uint64_t* keys_a = (uint64_t*)base; base += align_up((size_t)n_isects * 8, 256);
  int32_t* vals_a = (int32_t*)base; base += align_up((size_t)n_isects * 4, 256);
  int32_t* vals_b = (int32_t*)base; base += align_up((size_t)n_isects * 4, 256);
  uint32_t* table = (uint32_t*)base;
  hipLaunchKernelGGL(isect_emit_kernel, dim3(min(ceil_div(CN, 256), 256 * 16)), dim3(256), 0, s,
                     CN, N, means2d, radii, depths, cum, (float)tile_size, tile_width,
                     tile_height, tile_bits, (int64_t*)keys_a, vals_a);
  CLMGS_LAUNCH_CHECK();
  uint64_t* sorted = nullptr;  // the caller's isect_ids doubles as the second key buffer
  int rc = radix_sort_pairs<uint64_t>(s, n_isects, keys_a, (uint64_t*)isect_ids, vals_a, vals_b,
                                      flatten_ids, 0, 32 + tile_bits + cam_bits, table, &sorted, nullptr,
                                      !legacy_binning());
  if (rc) return rc;
  if (sorted != (uint64_t*)isect_ids)
    CLMGS_HIP(hipMemcpyAsync(isect_ids, sorted, (size_t)n_isects * 8, hipMemcpyDeviceToDevice, s));
  return 0;
}

extern "C" int clmgs_isect_offsets(void* stream, int64_t n_isects, const int64_t* isect_ids,
                                   int C, int tile_width, int tile_height, int32_t* offsets) {
  CLMGS_CHECK_ARG(C >= 1 && tile_width > 0 && tile_height > 0 && offsets && n_isects >= 0);
  hipStream_t s = (hipStream_t)stream;
  const int n_tiles = tile_width * tile_height;
  if (n_isects == 0) {
    CLMGS_HIP(hipMemsetAsync(offsets, 0, sizeof(int32_t) * (size_t)C * n_tiles, s));
    return 0;
  }
  CLMGS_CHECK_ARG(isect_ids);
  const int tile_bits = ilog2_floor((unsigned)n_tiles) + 1;
  hipLaunchKernelGGL(isect_offsets_kernel, dim3(min(ceil_div(n_isects, 256), 256 * 16)),
                     dim3(256), 0, s, n_isects, isect_ids, C * n_tiles, n_tiles, tile_bits,
                     offsets);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

// ======================================================================================
// Two-level binning (engine fast path).  Bit-identical order to the 64-bit (tile | depth)
// sort above at ~1/4 of its traffic:
//   A. stable sort of the V rows by depth bits (32-bit keys, culled rows last), then the tile
//      count of every row in that order + inclusive scan (-> I);
//   B. emit (tile id, row) in depth order, ONE stable sort on the tile-id bits only
//      (16 bits at 4K: 2 radix passes instead of 6 over 64-bit keys), offsets from the sorted ids.
// Ties: equal depths keep row order in A (stable), B is stable -> within a tile (depth, row)
// order, exactly what the single stable 64-bit sort of (row-major emit) produces.
// ======================================================================================
namespace clmgs {

// Row order (coalesced): depth key, identity payload, the row's tile box and tile mask -- the only
// place the means / radii / raster records are read; count and emit then work from 16 B per row
// (one gather in depth order, afterwards sequential) instead of re-gathering the arrays.
// packed == NULL: no exact culling (mask all ones, the gsplat.isect_tiles list).
__global__ void __launch_bounds__(256)
isect2_keys_kernel(int V, const int32_t* __restrict__ radii, const float* __restrict__ depths,
                   const float* __restrict__ means2d, const float4* __restrict__ packed,
                   float tile_size, int tile_w, int tile_h, uint32_t* __restrict__ keys,
                   int32_t* __restrict__ vals, unsigned long long* __restrict__ box_by_row,
                   int64_t* __restrict__ row_cnt) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) {
    const int r = radii[i];
    keys[i] = r > 0 ? (uint32_t)__float_as_int(depths[i]) : 0xFFFFFFFFu;
    vals[i] = i;
    unsigned long long b = 0ull, m = ~0ull;
    int cnt = 0;
    if (r > 0) {
      const float2 mm = *reinterpret_cast<const float2*>(means2d + 2 * (size_t)i);
      const TileBox tb = tile_box(mm.x, mm.y, (float)r, tile_size, tile_w, tile_h);
      b = pack_box(tb);
      if (packed) m = exact_tile_mask(packed + 4 * (size_t)i, tb.x0, tb.y0, tb.x1, tb.y1);
      const int nt = (tb.x1 - tb.x0) * (tb.y1 - tb.y0);
      cnt = nt <= 64 ? __popcll(m & (nt == 64 ? ~0ull : ((1ull << nt) - 1ull))) : nt;
    }
    box_by_row[2 * (size_t)i] = b;
    box_by_row[2 * (size_t)i + 1] = m;
    if (row_cnt) row_cnt[i] = cnt;  // emitted intersections of row i (same rule as isect2_count_kernel)
  }
}

// boxes[2*j] = packed box of rank j (0 = nothing to emit), boxes[2*j+1] = its tile mask;
// ref_total accumulates the UN-culled intersection count.
__global__ void __launch_bounds__(256)
isect2_count_kernel(int V, const int32_t* __restrict__ order,
                    const unsigned long long* __restrict__ box_by_row,
                    unsigned long long* __restrict__ boxes, int64_t* __restrict__ cum,
                    unsigned long long* __restrict__ ref_total) {
  unsigned long long ref = 0ull;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < V; j += gridDim.x * blockDim.x) {
    const int i = order[j];
    const unsigned long long b = box_by_row[2 * (size_t)i], m = box_by_row[2 * (size_t)i + 1];
    const int x0 = (int)(b & 0xFFFF), y0 = (int)((b >> 16) & 0xFFFF);
    const int x1 = (int)((b >> 32) & 0xFFFF), y1 = (int)(b >> 48);
    const int nt = (x1 - x0) * (y1 - y0);
    const int cnt = nt <= 64 ? __popcll(m & (nt == 64 ? ~0ull : ((1ull << nt) - 1ull))) : nt;
    boxes[2 * (size_t)j] = cnt > 0 ? b : 0ull;
    boxes[2 * (size_t)j + 1] = m;
    cum[j] = cnt;
    ref += (unsigned long long)nt;
  }
  if (ref_total) {  // one atomic per BLOCK: same-address atomics serialise at the memory side
    __shared__ unsigned long long wsum[4];
    ref = (unsigned long long)wave_sum_i64((long long)ref);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ref;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
      if (tot) atomicAdd(ref_total, tot);
    }
  }
}

// vals2 != NULL ("slot mode"): the sort payload is the pair (row id, SLOT).  Slots are numbered in ROW
// order: row i owns the contiguous range [row_cum[i-1], row_cum[i]) (its tiles in row-major tile
// order).  The backward tile kernel stores its per-(row, tile) partial gradients at the slot with
// plain stores; whoever consumes the row's gradient (clmgs_preprocess_bwd, or the row-sum kernel of
// clmgs_rasterize_bwd) adds the row's range in ascending order: no float atomics, deterministic, and
// both the partial lines and the consumer's rows are walked sequentially.
__global__ void __launch_bounds__(256)
isect2_emit_kernel(int V, const int32_t* __restrict__ order,
                   const unsigned long long* __restrict__ boxes,
                   const int64_t* __restrict__ cum, int tile_w, uint32_t* __restrict__ tkeys,
                   int32_t* __restrict__ vals, int2* __restrict__ vals2,
                   const int64_t* __restrict__ row_cum, int64_t cap) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < V; j += gridDim.x * blockDim.x) {
    const unsigned long long b = boxes[2 * (size_t)j];
    if (b == 0ull) continue;
    const unsigned long long m = boxes[2 * (size_t)j + 1];
    const int x0 = (int)(b & 0xFFFF), y0 = (int)((b >> 16) & 0xFFFF);
    const int x1 = (int)((b >> 32) & 0xFFFF), y1 = (int)(b >> 48);
    const bool masked = (x1 - x0) * (y1 - y0) <= 64;
    const int i = order[j];
    int64_t cur = (j == 0) ? 0 : cum[j - 1];
    int slot = (vals2 && i > 0) ? (int)row_cum[i - 1] : 0;
    int t = 0;
    for (int ty = y0; ty < y1; ++ty)
      for (int tx = x0; tx < x1; ++tx, ++t) {
        if (masked && !((m >> t) & 1ull)) continue;
        if (cur >= cap) return;  // device-count mode, capacity exceeded: dropped (the caller compares the totals)
        tkeys[cur] = (uint32_t)(ty * tile_w + tx);
        if (vals2) vals2[cur] = make_int2(i, slot++);
        else vals[cur] = i;
        ++cur;
      }
  }
}

// Round 5: the same list, produced CHUNK BY CHUNK.  Block b owns the entries [1024 b, 1024 (b+1)) of the unsorted
// list (the chunking of the first tile-sort pass).  It finds the first rank reaching into its chunk with a 256-ary
// search over `cum` (three dependent steps at 3 M ranks instead of 22), lets one thread per rank expand that rank's
// tiles into LDS, and then (i) stores tile ids and payload with coalesced 4 / 8 B-per-lane stores -- the
// thread-per-rank kernel above writes each rank's 2.8 entries at a different address, 112 MB at ~1 TB/s -- and (ii)
// counts the chunk's first digit into the radix table, which is the whole first histogram launch of the tile sort.
// Entry for entry the list of isect2_emit_kernel (same index = cum of the ranks before + the tile's place inside
// the rank's box, same payload).  Capacity form: entries at or beyond min(capacity, *n_dev) do not exist.
template <bool SLOTS>
__global__ void __launch_bounds__(256, 8)
isect2_emit_hist_kernel(int V, int64_t n, const int64_t* __restrict__ n_dev, const int32_t* __restrict__ order,
                        const unsigned long long* __restrict__ boxes, const int64_t* __restrict__ cum, int tile_w,
                        const int64_t* __restrict__ row_cum, uint32_t* __restrict__ tkeys,
                        int32_t* __restrict__ vals, int2* __restrict__ vals2, int n_blocks,
                        uint32_t* __restrict__ table /*[256][n_blocks]*/) {
  constexpr int CH = RS_MIN_CHUNK;
  static_assert(CH == 1024, "four entries per thread below");
  __shared__ uint32_t gkey[CH];
  __shared__ int2 gval2[SLOTS ? CH : 1];
  __shared__ int32_t gval[SLOTS ? 1 : CH];
  __shared__ uint32_t h[256];
  if (n_dev) n = min(n, *n_dev);
  const int tid = threadIdx.x;
  h[tid] = 0;
  const int64_t E0 = (int64_t)blockIdx.x * CH;
  if (E0 >= n) {  // (block-uniform) beyond the true count: an empty chunk, whose column of the table must still be zeros
    table[(size_t)tid * n_blocks + blockIdx.x] = 0;
    return;
  }
  const int64_t E1 = min(E0 + CH, n);
  // first rank whose inclusive count exceeds E0 (it exists: E0 < n <= cum[V-1]), narrowed to a stretch of <= 64 ranks
  // by a 256-ary search: two dependent steps at 3 M ranks (the first step's probes are the same for every block: L2
  // hits); the expansion below simply starts at the head of that stretch -- ranks that end at or before E0 emit nothing
  int lo = 0, hi = V;
  for (;;) {
    const int seg = (hi - lo + 255) / 256;
    const int q = min(hi - 1, lo + (tid + 1) * seg - 1);  // last rank of this thread's segment
    const int f = __syncthreads_count(cum[q] <= E0);      // segments entirely at or before E0
    if (f == 256) { lo = hi; break; }
    lo += f * seg;
    hi = min(hi, lo + seg);
    if (seg <= 64) break;
  }
  // expansion: RPT ranks per thread and round (2 measured slower: 154 vs 127 us), everything a rank needs requested at once (cum, box, mask, row id:
  // one level of latency; the row's first slot is the only dependent gather)
  constexpr int RPT = 1;
  for (int jb = lo; jb < V; jb += 256 * RPT) {
    int64_t incl[RPT], excl[RPT];
    unsigned long long bx[RPT], mk[RPT];
    int row[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      const int j = min(jb + u * 256 + tid, V - 1);  // clamped: the loads are unconditional
      incl[u] = cum[j];
      excl[u] = j ? cum[j - 1] : 0;
      bx[u] = boxes[2 * (size_t)j];
      mk[u] = boxes[2 * (size_t)j + 1];
      row[u] = order[j];
    }
    int slot0[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) slot0[u] = (SLOTS && row[u] > 0) ? (int)row_cum[row[u] - 1] : 0;
    bool more = false;
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      const int j = jb + u * 256 + tid;
      if (j < V && incl[u] > E0 && excl[u] < E1 && incl[u] > excl[u]) {
        const unsigned long long b = bx[u], m = mk[u];
        const int x0 = (int)(b & 0xFFFF), y0 = (int)((b >> 16) & 0xFFFF);
        const int x1 = (int)((b >> 32) & 0xFFFF), y1 = (int)(b >> 48);
        const int bw = x1 - x0, nt = bw * (y1 - y0);
        const int i = row[u];
        const int k_lo = (int)max((int64_t)0, E0 - excl[u]), k_hi = (int)min(incl[u] - excl[u], E1 - excl[u]);
        const int li0 = (int)(excl[u] - E0);  // LDS index of the rank's entry 0 (negative when the rank starts before the chunk)
        if (nt > 64) {  // unmasked box: entry k is tile k
          for (int k = k_lo; k < k_hi; ++k) {
            const int ty = k / bw, tx = k - ty * bw;
            gkey[li0 + k] = (uint32_t)((y0 + ty) * tile_w + x0 + tx);
            if (SLOTS) gval2[li0 + k] = make_int2(i, slot0[u] + k); else gval[li0 + k] = i;
          }
        } else {        // entry k is the k-th set bit of the tile mask
          unsigned long long mm = m & (nt == 64 ? ~0ull : ((1ull << nt) - 1ull));
          for (int k = 0; mm && k < k_hi; ++k) {
            const int t = __ffsll((long long)mm) - 1;
            mm &= mm - 1ull;
            if (k >= k_lo) {
              const int ty = t / bw, tx = t - ty * bw;
              gkey[li0 + k] = (uint32_t)((y0 + ty) * tile_w + x0 + tx);
              if (SLOTS) gval2[li0 + k] = make_int2(i, slot0[u] + k); else gval[li0 + k] = i;
            }
          }
        }
      }
      // the chunk is complete once the LAST rank of this round reaches E1 (only the lane holding it votes)
      if (j == min(jb + 256 * RPT - 1, V - 1)) more = incl[u] < E1;
    }
    if (!__syncthreads_or(more)) break;
  }
  __syncthreads();
  const int n_here = (int)(E1 - E0);
#pragma unroll
  for (int it = 0; it < CH / 256; ++it) {
    const int li = it * 256 + tid;
    if (li < n_here) {
      const uint32_t key = gkey[li];
      atomicAdd(&h[key & 0xFFu], 1u);
      tkeys[E0 + li] = key;
      if (SLOTS) vals2[E0 + li] = gval2[li]; else vals[E0 + li] = gval[li];
    }
  }
  __syncthreads();
  table[(size_t)tid * n_blocks + blockIdx.x] = h[tid];
}

// slot mode: the sorted (row id, emit index) pairs are split into flatten_ids / emit_slot here.
__global__ void __launch_bounds__(256)
isect2_offsets_kernel(int64_t n_isects, const uint32_t* __restrict__ tkeys, int n_tiles,
                      int32_t* __restrict__ offsets, int32_t* __restrict__ flatten_ids,
                      int32_t* __restrict__ emit_slot, const int2* __restrict__ sorted2,
                      const float* __restrict__ depths, int64_t* __restrict__ isect_ids,
                      const int64_t* __restrict__ n_dev) {
  if (n_dev) n_isects = min(n_isects, *n_dev);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_isects;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int cur = (int)tkeys[i];
    if (i == 0) {
      for (int t = 0; t <= cur; ++t) offsets[t] = 0;
    } else {
      const int prev = (int)tkeys[i - 1];
      for (int t = prev + 1; t <= cur; ++t) offsets[t] = (int32_t)i;
    }
    if (i == n_isects - 1)
      for (int t = cur + 1; t < n_tiles; ++t) offsets[t] = (int32_t)n_isects;
    int gid;
    if (sorted2) { const int2 v = sorted2[i]; gid = v.x; flatten_ids[i] = gid; emit_slot[i] = v.y; }
    else gid = flatten_ids[i];
    if (isect_ids)
      isect_ids[i] = ((int64_t)cur << 32) | (int64_t)(uint32_t)__float_as_int(depths[gid]);
  }
}


// ======================================================================================
// Round 4: the same chain in 11 launches instead of ~30 (onesweep.h).
//   memset(control block) -> keys_lb (keys, boxes, masks, per-row counts + their scan = row_cum, digit counts of all
//   four depth-sort passes) -> 4 x onesweep pass -> count_lb (boxes in depth order, counts + their scan = cum, totals)
//   | memset(control block) -> emit_lb (tile ids + payload, digit counts of the tile sort) -> 2 x onesweep pass (the last
//   one writes flatten_ids / emit_slot directly) -> offsets.
// Every list is element for element the legacy chain's (stable LSD passes in the same digit order).
// ======================================================================================
constexpr int LBK_ROUNDS = 4;
constexpr int LBK_CHUNK = 256 * LBK_ROUNDS;  // rows (ranks) per ticket of the keys / count kernels

__host__ __device__ static inline int lbk_chunks(int V) { return (V + LBK_CHUNK - 1) / LBK_CHUNK; }

// LOOKBACK = false (default route): chunk = blockIdx.x, the scan stays block-relative (row_cum gets the inclusive
// scan inside the chunk, block_tot[chunk] its total; scan_i64_finish_kernel adds the offsets), no digit counts.
template <bool LOOKBACK>
__global__ void __launch_bounds__(256)
isect2_keys_lb_kernel(int V, const int32_t* __restrict__ radii, const float* __restrict__ depths,
                      const float* __restrict__ means2d, const float4* __restrict__ packed, float tile_size,
                      int tile_w, int tile_h, uint32_t* __restrict__ keys, int32_t* __restrict__ vals,
                      unsigned long long* __restrict__ box_by_row, int64_t* __restrict__ row_cum,
                      uint32_t* __restrict__ ghist /*[4][256]*/, unsigned long long* __restrict__ status,
                      uint32_t* __restrict__ ticket_ctr, int64_t* __restrict__ totals, uint32_t* err,
                      int64_t* __restrict__ block_tot) {
  __shared__ uint32_t hist[LOOKBACK ? 4 : 1][256];
  __shared__ long long wsum[4];
  __shared__ long long sh[2];
  __shared__ int ticket_s;
  const int tid = threadIdx.x;
  if constexpr (LOOKBACK) {
#pragma unroll
    for (int p = 0; p < 4; ++p) hist[p][tid] = 0;
  }
  const int n_chunks = lbk_chunks(V);
  for (int iter = 0;; ++iter) {
    int chunk;
    if constexpr (LOOKBACK) {
      __syncthreads();
      if (tid == 0) ticket_s = (int)atomicAdd(ticket_ctr, 1u);
      __syncthreads();
      chunk = ticket_s;
    } else {
      chunk = blockIdx.x + iter * gridDim.x;
    }
    if (chunk >= n_chunks) break;
    if (chunk == 0 && tid == 0) { totals[0] = 0; totals[1] = 0; }  // the count kernel accumulates into them (later launch)
    long long c[LBK_ROUNDS];
#pragma unroll
    for (int r = 0; r < LBK_ROUNDS; ++r) {
      const int i = chunk * LBK_CHUNK + r * 256 + tid;
      c[r] = 0;
      if (i < V) {
        const int rad = radii[i];
        const uint32_t key = rad > 0 ? (uint32_t)__float_as_int(depths[i]) : 0xFFFFFFFFu;
        keys[i] = key;
        vals[i] = i;
        unsigned long long b = 0ull, m = ~0ull;
        int cnt = 0;
        if (rad > 0) {
          const float2 mm = *reinterpret_cast<const float2*>(means2d + 2 * (size_t)i);
          const TileBox tb = tile_box(mm.x, mm.y, (float)rad, tile_size, tile_w, tile_h);
          b = pack_box(tb);
          if (packed) m = exact_tile_mask(packed + 4 * (size_t)i, tb.x0, tb.y0, tb.x1, tb.y1);
          const int nt = (tb.x1 - tb.x0) * (tb.y1 - tb.y0);
          cnt = nt <= 64 ? __popcll(m & (nt == 64 ? ~0ull : ((1ull << nt) - 1ull))) : nt;
        }
        box_by_row[2 * (size_t)i] = b;
        box_by_row[2 * (size_t)i + 1] = m;
        c[r] = cnt;
        if constexpr (LOOKBACK) {
          atomicAdd(&hist[0][key & 0xFFu], 1u);
          atomicAdd(&hist[1][(key >> 8) & 0xFFu], 1u);
          atomicAdd(&hist[2][(key >> 16) & 0xFFu], 1u);
          atomicAdd(&hist[3][key >> 24], 1u);
        }
      }
    }
    if (row_cum) {  // inclusive scan of the emitted counts in ROW order (the slot ranges of the backward)
      long long inc[LBK_ROUNDS];
      long long carry = 0;
#pragma unroll
      for (int r = 0; r < LBK_ROUNDS; ++r) {
        const long long incl = block_incl_scan_i64(c[r], wsum);
        inc[r] = carry + incl;
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
      }
      long long excl = 0;
#ifdef CLMGS_PROFILE_BUILD
      if constexpr (LOOKBACK) excl = lb_chunk_prefix(status, chunk, carry, sh, err);
      else
#endif
      if (tid == 0) block_tot[chunk] = carry;
#pragma unroll
      for (int r = 0; r < LBK_ROUNDS; ++r) {
        const int i = chunk * LBK_CHUNK + r * 256 + tid;
        if (i < V) row_cum[i] = excl + inc[r];
      }
    }
  }
  if constexpr (LOOKBACK) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const uint32_t v = hist[p][tid];
      if (v) atomicAdd(&ghist[p * 256 + tid], v);
    }
  }
}

// boxes / masks in depth order, cum = inclusive scan of the emitted counts in that order, totals[0] = the grand
// total, totals[1] += the un-culled count (both zeroed by keys_lb).
template <bool LOOKBACK>
__global__ void __launch_bounds__(256)
isect2_count_lb_kernel(int V, const int32_t* __restrict__ order,
                       const unsigned long long* __restrict__ box_by_row,
                       unsigned long long* __restrict__ boxes, int64_t* __restrict__ cum,
                       int64_t* __restrict__ totals, unsigned long long* __restrict__ status,
                       uint32_t* __restrict__ ticket_ctr, uint32_t* err, int64_t* __restrict__ block_tot) {
  __shared__ long long wsum[4];
  __shared__ long long sh[2];
  __shared__ unsigned long long rsum[4];
  __shared__ int ticket_s;
  const int tid = threadIdx.x;
  const int n_chunks = lbk_chunks(V);
  unsigned long long ref = 0ull;
  for (int iter = 0;; ++iter) {
    int chunk;
    if constexpr (LOOKBACK) {
      __syncthreads();
      if (tid == 0) ticket_s = (int)atomicAdd(ticket_ctr, 1u);
      __syncthreads();
      chunk = ticket_s;
    } else {
      chunk = blockIdx.x + iter * gridDim.x;
    }
    if (chunk >= n_chunks) break;
    long long inc[LBK_ROUNDS];
    long long carry = 0;
#pragma unroll
    for (int r = 0; r < LBK_ROUNDS; ++r) {
      const int j = chunk * LBK_CHUNK + r * 256 + tid;
      long long cnt = 0;
      if (j < V) {
        const int i = order[j];
        const unsigned long long b = box_by_row[2 * (size_t)i], m = box_by_row[2 * (size_t)i + 1];
        const int x0 = (int)(b & 0xFFFF), y0 = (int)((b >> 16) & 0xFFFF);
        const int x1 = (int)((b >> 32) & 0xFFFF), y1 = (int)(b >> 48);
        const int nt = (x1 - x0) * (y1 - y0);
        cnt = nt <= 64 ? __popcll(m & (nt == 64 ? ~0ull : ((1ull << nt) - 1ull))) : nt;
        boxes[2 * (size_t)j] = cnt > 0 ? b : 0ull;
        boxes[2 * (size_t)j + 1] = m;
        ref += (unsigned long long)nt;
      }
      const long long incl = block_incl_scan_i64(cnt, wsum);
      inc[r] = carry + incl;
      carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    long long excl = 0;
#ifdef CLMGS_PROFILE_BUILD
    if constexpr (LOOKBACK) excl = lb_chunk_prefix(status, chunk, carry, sh, err);
    else
#endif
    if (tid == 0) block_tot[chunk] = carry;
#pragma unroll
    for (int r = 0; r < LBK_ROUNDS; ++r) {
      const int j = chunk * LBK_CHUNK + r * 256 + tid;
      if (j < V) cum[j] = excl + inc[r];
    }
    if (LOOKBACK && chunk == n_chunks - 1 && tid == 0) totals[0] = excl + carry;
  }
  ref = (unsigned long long)wave_sum_i64((long long)ref);
  __syncthreads();
  if ((tid & 63) == 0) rsum[tid >> 6] = ref;
  __syncthreads();
  if (tid == 0) {
    const unsigned long long tot = rsum[0] + rsum[1] + rsum[2] + rsum[3];
    if (tot) atomicAdd((unsigned long long*)(totals + 1), tot);
  }
}

// emit + the digit counts of every pass of the tile sort (entries beyond the capacity are neither written nor
// counted, so the counts always describe the min(capacity, total) keys the passes sort)
__global__ void __launch_bounds__(256)
isect2_emit_lb_kernel(int V, const int32_t* __restrict__ order,
                      const unsigned long long* __restrict__ boxes,
                      const int64_t* __restrict__ cum, int tile_w, uint32_t* __restrict__ tkeys,
                      int32_t* __restrict__ vals, int2* __restrict__ vals2,
                      const int64_t* __restrict__ row_cum, int64_t cap, int n_pass,
                      uint32_t* __restrict__ ghist /*[4][256]*/) {
  __shared__ uint32_t hist[4][256];
#pragma unroll
  for (int p = 0; p < 4; ++p) hist[p][threadIdx.x] = 0;
  __syncthreads();
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < V; j += gridDim.x * blockDim.x) {
    const unsigned long long b = boxes[2 * (size_t)j];
    if (b == 0ull) continue;
    const unsigned long long m = boxes[2 * (size_t)j + 1];
    const int x0 = (int)(b & 0xFFFF), y0 = (int)((b >> 16) & 0xFFFF);
    const int x1 = (int)((b >> 32) & 0xFFFF), y1 = (int)(b >> 48);
    const bool masked = (x1 - x0) * (y1 - y0) <= 64;
    const int i = order[j];
    int64_t cur = (j == 0) ? 0 : cum[j - 1];
    int slot = (vals2 && i > 0) ? (int)row_cum[i - 1] : 0;
    int t = 0;
    for (int ty = y0; ty < y1 && cur < cap; ++ty)
      for (int tx = x0; tx < x1; ++tx, ++t) {
        if (masked && !((m >> t) & 1ull)) continue;
        if (cur >= cap) break;  // device-count mode, capacity exceeded: dropped (the caller compares the totals)
        const uint32_t key = (uint32_t)(ty * tile_w + tx);
        tkeys[cur] = key;
        if (vals2) vals2[cur] = make_int2(i, slot++);
        else vals[cur] = i;
        ++cur;
        atomicAdd(&hist[0][key & 0xFFu], 1u);
        if (n_pass > 1) atomicAdd(&hist[1][(key >> 8) & 0xFFu], 1u);
        if (n_pass > 2) atomicAdd(&hist[2][(key >> 16) & 0xFFu], 1u);
        if (n_pass > 3) atomicAdd(&hist[3][key >> 24], 1u);
      }
  }
  __syncthreads();
  for (int p = 0; p < n_pass; ++p) {
    const uint32_t v = hist[p][threadIdx.x];
    if (v) atomicAdd(&ghist[p * 256 + threadIdx.x], v);
  }
}

// offsets from the sorted tile ids (+ optional isect_ids); a true count of 0 leaves no entry to derive them from:
// the grid fills them with zeros itself (no memset launch).
__global__ void __launch_bounds__(256)
isect2_offsets_lb_kernel(int64_t n_isects, const uint32_t* __restrict__ tkeys, int n_tiles,
                         int32_t* __restrict__ offsets, const int32_t* __restrict__ flatten_ids,
                         const float* __restrict__ depths, int64_t* __restrict__ isect_ids,
                         const int64_t* __restrict__ n_dev) {
  if (n_dev) n_isects = min(n_isects, *n_dev);
  if (n_isects <= 0) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n_tiles; t += gridDim.x * blockDim.x) offsets[t] = 0;
    return;
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_isects;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int cur = (int)tkeys[i];
    if (i == 0) {
      for (int t = 0; t <= cur; ++t) offsets[t] = 0;
    } else {
      const int prev = (int)tkeys[i - 1];
      for (int t = prev + 1; t <= cur; ++t) offsets[t] = (int32_t)i;
    }
    if (i == n_isects - 1)
      for (int t = cur + 1; t < n_tiles; ++t) offsets[t] = (int32_t)n_isects;
    if (isect_ids)
      isect_ids[i] = ((int64_t)cur << 32) | (int64_t)(uint32_t)__float_as_int(depths[flatten_ids[i]]);
  }
}

// Second half of a scan whose producer left block-relative inclusive values (chunks of LBK_CHUNK) and the chunk totals:
// every block sums the totals of the chunks before its own (a block-wide reduction over <= a few thousand L2-resident
// words) and adds that offset -- one launch instead of scan-the-totals + add.  last_out: the grand total.
__global__ void __launch_bounds__(256)
scan_i64_finish_kernel(int n, int64_t* __restrict__ data, const int64_t* __restrict__ block_tot,
                       int64_t* __restrict__ last_out) {
  __shared__ long long wsum[4];
  const int chunk = blockIdx.x, tid = threadIdx.x;
  long long part = 0;
  for (int j = tid; j < chunk; j += 256) part += block_tot[j];
  part = wave_sum_i64(part);
  if ((tid & 63) == 0) wsum[tid >> 6] = part;
  __syncthreads();
  const long long off = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  if (chunk > 0) {
#pragma unroll
    for (int r = 0; r < LBK_ROUNDS; ++r) {
      const int i = chunk * LBK_CHUNK + r * 256 + tid;
      if (i < n) data[i] += off;
    }
  }
  if (last_out && chunk == (int)gridDim.x - 1 && tid == 0) *last_out = off + block_tot[chunk];
}

// control block of clmgs_isect2_order_count: [4][256] digit counts | 16 ticket words | scan status of keys_lb | scan
// status of count_lb | 4 x sort status
#ifdef CLMGS_PROFILE_BUILD
static inline size_t order_ctrl_bytes(int V) {
  return 4096 + 256 + 2 * align_up((size_t)lbk_chunks(V) * 8, 256) + 4 * os_status_bytes(V);
}
static inline size_t sort_ctrl_bytes(int64_t n) { return 4096 + 256 + 4 * os_status_bytes(n); }
#else
static inline size_t order_ctrl_bytes(int) { return 0; }
static inline size_t sort_ctrl_bytes(int64_t) { return 0; }
#endif

}  // namespace clmgs

extern "C" size_t clmgs_isect2_order_temp_bytes(int V) {
  if (V <= 0) return 256;
  const size_t legacy = radix_table_bytes(V) + max(scan_scratch_bytes(V), (size_t)2 * lbk_chunks(V) * sizeof(int64_t) + 256);
  return 4 * align_up((size_t)V * 4, 256) + align_up((size_t)V * 16, 256) + max(legacy, order_ctrl_bytes(V)) + 256;
}

// order[V] i32 (rows by depth, culled last), cum[V] i64 (inclusive tile counts in that order),
// boxes[2V] u64 (packed tile box + tile mask of every rank, box 0 = nothing to emit; input of
// clmgs_isect2_emit_sort), totals[2] i64 device = {intersections to emit, un-culled count}.
// packed != NULL (the [V,16] raster records): exact per-tile culling.
// row_cum[V] i64 optional: inclusive emitted-intersection counts in ROW order (the slot ranges).
extern "C" int clmgs_isect2_order_count(void* stream, int V, const float* means2d,
                                        const int32_t* radii, const float* depths, int tile_size,
                                        int tile_width, int tile_height, const void* packed,
                                        int32_t* order, int64_t* cum, uint64_t* boxes,
                                        int64_t* totals, void* temp, size_t temp_bytes,
                                        int64_t* row_cum) {
  CLMGS_CHECK_ARG(V >= 0 && tile_size > 0 && tile_width > 0 && tile_height > 0);
  CLMGS_CHECK_ARG(tile_width < 65536 && tile_height < 65536);
  if (V == 0) return 0;
  CLMGS_CHECK_ARG(means2d && radii && depths && order && cum && boxes && totals && temp);
  CLMGS_CHECK_ARG(!packed || tile_size == 16);
  CLMGS_CHECK_ARG(temp_bytes >= clmgs_isect2_order_temp_bytes(V));
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)temp;
  uint32_t* k_a = (uint32_t*)base; base += align_up((size_t)V * 4, 256);
  uint32_t* k_b = (uint32_t*)base; base += align_up((size_t)V * 4, 256);
  int32_t* v_a = (int32_t*)base; base += align_up((size_t)V * 4, 256);
  int32_t* v_b = (int32_t*)base; base += align_up((size_t)V * 4, 256);
  unsigned long long* box_by_row = (unsigned long long*)base; base += align_up((size_t)V * 16, 256);
  uint32_t* table = (uint32_t*)base; base += radix_table_bytes(V);
  int64_t* scan_tmp = (int64_t*)base;
  if (binning_route() == BIN_FUSED || binning_route() == BIN_R4) {
    int64_t* tot_a = scan_tmp;                      // chunk totals of the two scans
    int64_t* tot_b = scan_tmp + lbk_chunks(V);
    const int nck = lbk_chunks(V);
    hipLaunchKernelGGL((isect2_keys_lb_kernel<false>), dim3(nck), dim3(256), 0, s, V, radii, depths, means2d,
                       (const float4*)packed, (float)tile_size, tile_width, tile_height, k_a, v_a, box_by_row,
                       row_cum, (uint32_t*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr, totals,
                       (uint32_t*)nullptr, tot_a);
    if (row_cum)
      hipLaunchKernelGGL(scan_i64_finish_kernel, dim3(nck), dim3(256), 0, s, V, row_cum, tot_a, (int64_t*)nullptr);
    CLMGS_LAUNCH_CHECK();
    uint32_t* sorted = nullptr;
    int rc = radix_sort_pairs<uint32_t>(s, V, k_a, k_b, v_a, v_b, order, 0, 32, table, &sorted);
    if (rc) return rc;
    hipLaunchKernelGGL((isect2_count_lb_kernel<false>), dim3(nck), dim3(256), 0, s, V, order, box_by_row,
                       (unsigned long long*)boxes, cum, totals, (unsigned long long*)nullptr, (uint32_t*)nullptr,
                       (uint32_t*)nullptr, tot_b);
    hipLaunchKernelGGL(scan_i64_finish_kernel, dim3(nck), dim3(256), 0, s, V, cum, tot_b, totals);
    CLMGS_LAUNCH_CHECK();
    return 0;
  }
#ifdef CLMGS_PROFILE_BUILD
  if (binning_route() == BIN_LOOKBACK) {
    // control block (zeroed by ONE memset): digit counts | tickets | scan status x 2 | sort status x 4
    char* ctrl = (char*)table;  // (the legacy routes' radix table + scan scratch: the temp size is the max of both)
    uint32_t* ghist = (uint32_t*)ctrl;
    uint32_t* tickets = (uint32_t*)(ctrl + 4096);
    const size_t sb = align_up((size_t)lbk_chunks(V) * 8, 256);
    unsigned long long* st_keys = (unsigned long long*)(ctrl + 4096 + 256);
    unsigned long long* st_count = (unsigned long long*)(ctrl + 4096 + 256 + sb);
    char* st_sort = ctrl + 4096 + 256 + 2 * sb;
    const size_t ssb = os_status_bytes(V);
    uint32_t* err = device_error_word();
    CLMGS_CHECK_ARG(err != nullptr);
    CLMGS_HIP(hipMemsetAsync(ctrl, 0, order_ctrl_bytes(V), s));
    const int nck = lbk_chunks(V);
    hipLaunchKernelGGL((isect2_keys_lb_kernel<true>), dim3(min(nck, 1024)), dim3(256), 0, s, V, radii, depths, means2d,
                       (const float4*)packed, (float)tile_size, tile_width, tile_height, k_a, v_a, box_by_row,
                       row_cum, ghist, st_keys, tickets + 0, totals, err, (int64_t*)nullptr);
    uint32_t* ksrc = k_a;
    uint32_t* kdst = k_b;
    int32_t* vsrc = v_a;
    for (int p = 0; p < 4; ++p) {
      int32_t* vdst = (p == 3) ? order : (vsrc == v_a ? v_b : v_a);
      launch_onesweep_pass<int32_t, false>(s, os_rounds("CLMGS_OS_ROUNDS_V", 8), (int64_t)V, (const int64_t*)nullptr, ksrc,
                                           vsrc, kdst, vdst, (int32_t*)nullptr, (int32_t*)nullptr, 8 * p,
                                           ghist + 256 * p, (uint32_t*)(st_sort + ssb * p), tickets + 2 + p, err);
      uint32_t* t_ = ksrc; ksrc = kdst; kdst = t_;
      vsrc = vdst;
    }
    hipLaunchKernelGGL((isect2_count_lb_kernel<true>), dim3(min(nck, 1024)), dim3(256), 0, s, V, order, box_by_row,
                       (unsigned long long*)boxes, cum, totals, st_count, tickets + 1, err, (int64_t*)nullptr);
    CLMGS_LAUNCH_CHECK();
    return 0;
  }
#endif
  const int grid = min(ceil_div(V, 256), 256 * 16);
  hipLaunchKernelGGL(isect2_keys_kernel, dim3(grid), dim3(256), 0, s, V, radii, depths, means2d,
                     (const float4*)packed, (float)tile_size, tile_width, tile_height, k_a, v_a,
                     box_by_row, row_cum);
  CLMGS_LAUNCH_CHECK();
  uint32_t* sorted = nullptr;
  int rc = radix_sort_pairs<uint32_t>(s, V, k_a, k_b, v_a, v_b, order, 0, 32, table, &sorted, nullptr, false);
  if (rc) return rc;
  if (row_cum) {
    rc = inclusive_scan_i64(s, V, row_cum, scan_tmp);
    if (rc) return rc;
  }
  CLMGS_HIP(hipMemsetAsync(totals, 0, 2 * sizeof(int64_t), s));
  hipLaunchKernelGGL(isect2_count_kernel, dim3(min(grid, 1024)), dim3(256), 0, s, V, order, box_by_row,
                     (unsigned long long*)boxes, cum, (unsigned long long*)(totals + 1));
  CLMGS_LAUNCH_CHECK();
  return inclusive_scan_i64(s, V, cum, scan_tmp, totals);  // totals[0] = cum[V-1], no extra copy
}

extern "C" size_t clmgs_isect2_sort_temp_bytes(int64_t n_isects) {
  if (n_isects <= 0) return 256;
  return 8 * align_up((size_t)n_isects * 4, 256) + max(radix_table_bytes(n_isects), sort_ctrl_bytes(n_isects)) + 256;
}

// flatten_ids[I] i32 (row ids, sorted by tile then depth), offsets[tile_w*tile_h] i32,
// isect_ids[I] i64 optional (NULL to skip).  emit_slot[I] optional: the emit index of every sorted
// intersection, consumed (with order / cum) by clmgs_rasterize_bwd's atomic-free accumulation.
static int isect2_emit_sort_impl(void* stream, int V, int64_t n_isects, const float* depths,
                                 const int32_t* order, const int64_t* cum,
                                 const uint64_t* boxes, int tile_width, int tile_height,
                                 int32_t* flatten_ids, int32_t* offsets, int64_t* isect_ids,
                                 int32_t* emit_slot, void* temp, size_t temp_bytes,
                                 const int64_t* row_cum, const int64_t* n_dev) {
  CLMGS_CHECK_ARG(V >= 0 && n_isects >= 0 && offsets);
  hipStream_t s = (hipStream_t)stream;
  const int n_tiles = tile_width * tile_height;
  const int route = binning_route();
  const bool lb = route == BIN_LOOKBACK && n_isects < ((int64_t)1 << 30);  // 30-bit counts in the look-back words
  const bool fused = route == BIN_FUSED || route == BIN_R4;
  if (n_isects == 0 || (n_dev && !lb && !fused)) {  // device-count mode: a true count of 0 leaves no thread to write the offsets
    CLMGS_HIP(hipMemsetAsync(offsets, 0, sizeof(int32_t) * (size_t)n_tiles, s));
    if (n_isects == 0) return 0;
  }
  CLMGS_CHECK_ARG(n_isects < ((int64_t)1 << 31));
  CLMGS_CHECK_ARG(depths && order && cum && boxes && flatten_ids && temp);
  CLMGS_CHECK_ARG(temp_bytes >= clmgs_isect2_sort_temp_bytes(n_isects));
  char* base = (char*)temp;
  const size_t a4 = align_up((size_t)n_isects * 4, 256);
  uint32_t* k_a = (uint32_t*)base; base += a4;
  uint32_t* k_b = (uint32_t*)base; base += a4;
  char* v_a = base; base += 2 * a4;   // int32 payload (plain) or int2 payload (slot mode)
  char* v_b = base; base += 2 * a4;
  char* v_f = base; base += 2 * a4;
  uint32_t* table = (uint32_t*)base;
  const bool slots = emit_slot != nullptr;
  CLMGS_CHECK_ARG(!slots || row_cum);
  const int tile_bits = ilog2_floor((unsigned)n_tiles) + 1;
#ifdef CLMGS_PROFILE_BUILD
  if (lb) {
    const int n_pass = (tile_bits + 7) / 8;
    char* ctrl = (char*)table;
    uint32_t* ghist = (uint32_t*)ctrl;
    uint32_t* tickets = (uint32_t*)(ctrl + 4096);
    char* st_sort = ctrl + 4096 + 256;
    const size_t ssb = os_status_bytes(n_isects);
    uint32_t* err = device_error_word();
    CLMGS_CHECK_ARG(err != nullptr);
    CLMGS_HIP(hipMemsetAsync(ctrl, 0, 4096 + 256 + (size_t)n_pass * ssb, s));
    hipLaunchKernelGGL(isect2_emit_lb_kernel, dim3(min(ceil_div(V, 256), 1024)), dim3(256), 0, s, V, order,
                       (const unsigned long long*)boxes, cum, tile_width, k_a, (int32_t*)v_a,
                       slots ? (int2*)v_a : nullptr, row_cum, n_isects, n_pass, ghist);
    uint32_t* ksrc = k_a;
    uint32_t* kdst = k_b;
    char* vsrc = v_a;
    const int rounds = os_rounds("CLMGS_OS_ROUNDS_I", 8);
    for (int p = 0; p < n_pass; ++p) {
      const bool last = p == n_pass - 1;
      char* vdst = (vsrc == v_a) ? v_b : v_a;
      uint32_t* st = (uint32_t*)(st_sort + ssb * p);
      if (slots) {
        if (last)
          launch_onesweep_pass<int2, true>(s, rounds, n_isects, n_dev, ksrc, (const int2*)vsrc, kdst, (int2*)nullptr,
                                           flatten_ids, emit_slot, 8 * p, ghist + 256 * p, st, tickets + p, err);
        else
          launch_onesweep_pass<int2, false>(s, rounds, n_isects, n_dev, ksrc, (const int2*)vsrc, kdst, (int2*)vdst,
                                            (int32_t*)nullptr, (int32_t*)nullptr, 8 * p, ghist + 256 * p, st,
                                            tickets + p, err);
      } else {
        launch_onesweep_pass<int32_t, false>(s, rounds, n_isects, n_dev, ksrc, (const int32_t*)vsrc, kdst,
                                             last ? flatten_ids : (int32_t*)vdst, (int32_t*)nullptr,
                                             (int32_t*)nullptr, 8 * p, ghist + 256 * p, st, tickets + p, err);
      }
      uint32_t* t_ = ksrc; ksrc = kdst; kdst = t_;
      vsrc = vdst;
    }
    hipLaunchKernelGGL(isect2_offsets_lb_kernel, dim3(min(ceil_div(n_isects, 256), 256 * 16)), dim3(256), 0, s,
                       n_isects, ksrc, n_tiles, offsets, flatten_ids, depths, isect_ids, n_dev);
    CLMGS_LAUNCH_CHECK();
    return 0;
  }
#endif
  uint32_t* sorted = nullptr;
  int rc;
  const bool emit_hist = route == BIN_FUSED && tile_bits > 0;
  if (emit_hist) {
    // round 5: one block per 1024-entry chunk of the (unsorted) list expands the ranks that fall into it through LDS,
    // stores tile ids + payload coalesced and leaves the chunk's first-digit counts in the radix table
    const int n_blocks = (int)((n_isects + RS_MIN_CHUNK - 1) / RS_MIN_CHUNK);
    if (slots)
      hipLaunchKernelGGL((isect2_emit_hist_kernel<true>), dim3(n_blocks), dim3(256), 0, s, V, n_isects, n_dev, order,
                         (const unsigned long long*)boxes, cum, tile_width, row_cum, k_a, (int32_t*)nullptr, (int2*)v_a,
                         n_blocks, table);
    else
      hipLaunchKernelGGL((isect2_emit_hist_kernel<false>), dim3(n_blocks), dim3(256), 0, s, V, n_isects, n_dev, order,
                         (const unsigned long long*)boxes, cum, tile_width, row_cum, k_a, (int32_t*)v_a, (int2*)nullptr,
                         n_blocks, table);
  } else {
    hipLaunchKernelGGL(isect2_emit_kernel, dim3(min(ceil_div(V, 256), 256 * 16)), dim3(256), 0, s, V,
                       order, (const unsigned long long*)boxes, cum, tile_width, k_a, (int32_t*)v_a,
                       slots ? (int2*)v_a : nullptr, row_cum, n_isects);
  }
  CLMGS_LAUNCH_CHECK();
  if (fused) {
    // the last pass writes flatten_ids / emit_slot itself (no int2 list to split afterwards); the offsets kernel then
    // reads the sorted tile ids only and zero-fills the offsets itself when the true count is 0
    if (slots)
      rc = radix_sort_pairs_impl<uint32_t, int2, RS_DEFAULT_ITEMS>(s, n_isects, k_a, k_b, (int2*)v_a, (int2*)v_b, (int2*)nullptr,
                                                                    0, tile_bits, table, &sorted, n_dev, flatten_ids, emit_slot,
                                                                    true, emit_hist);
    else
      rc = radix_sort_pairs_impl<uint32_t, int32_t, RS_DEFAULT_ITEMS>(s, n_isects, k_a, k_b, (int32_t*)v_a, (int32_t*)v_b,
                                                                       flatten_ids, 0, tile_bits, table, &sorted, n_dev,
                                                                       nullptr, nullptr, true, emit_hist);
    if (rc) return rc;
    hipLaunchKernelGGL(isect2_offsets_lb_kernel, dim3(min(ceil_div(n_isects, 256), 256 * 16)), dim3(256), 0, s,
                       n_isects, sorted, n_tiles, offsets, flatten_ids, depths, isect_ids, n_dev);
    CLMGS_LAUNCH_CHECK();
    return 0;
  }
  if (slots)
    rc = radix_sort_pairs<uint32_t, int2>(s, n_isects, k_a, k_b, (int2*)v_a, (int2*)v_b, (int2*)v_f, 0,
                                          tile_bits, table, &sorted, n_dev, false);
  else
    rc = radix_sort_pairs<uint32_t, int32_t>(s, n_isects, k_a, k_b, (int32_t*)v_a, (int32_t*)v_b,
                                             flatten_ids, 0, tile_bits, table, &sorted, n_dev, false);
  if (rc) return rc;
  hipLaunchKernelGGL(isect2_offsets_kernel, dim3(min(ceil_div(n_isects, 256), 256 * 16)), dim3(256), 0,
                     s, n_isects, sorted, n_tiles, offsets, flatten_ids, emit_slot,
                     slots ? (const int2*)v_f : nullptr, depths, isect_ids, n_dev);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_isect2_emit_sort(void* stream, int V, int64_t n_isects, const float* depths,
                                      const int32_t* order, const int64_t* cum,
                                      const uint64_t* boxes, int tile_width, int tile_height,
                                      int32_t* flatten_ids, int32_t* offsets, int64_t* isect_ids,
                                      int32_t* emit_slot, void* temp, size_t temp_bytes,
                                      const int64_t* row_cum) {
  return isect2_emit_sort_impl(stream, V, n_isects, depths, order, cum, boxes, tile_width, tile_height,
                               flatten_ids, offsets, isect_ids, emit_slot, temp, temp_bytes, row_cum, nullptr);
}

// Device-count form: `capacity` sizes every buffer and launch, the TRUE intersection count is read on the
// device from n_isects_dev (totals[0] of clmgs_isect2_order_count), so the host need not wait for it before
// enqueueing.  Intersections beyond the capacity are dropped: the caller MUST compare the count with the
// capacity (after the fact, from its asynchronous readback) and redo the camera with the exact form if it
// was exceeded.
extern "C" int clmgs_isect2_emit_sort_dev(void* stream, int V, int64_t capacity, const int64_t* n_isects_dev,
                                          const float* depths, const int32_t* order, const int64_t* cum,
                                          const uint64_t* boxes, int tile_width, int tile_height,
                                          int32_t* flatten_ids, int32_t* offsets, int64_t* isect_ids,
                                          int32_t* emit_slot, void* temp, size_t temp_bytes,
                                          const int64_t* row_cum) {
  CLMGS_CHECK_ARG(n_isects_dev && capacity > 0);
  return isect2_emit_sort_impl(stream, V, capacity, depths, order, cum, boxes, tile_width, tile_height,
                               flatten_ids, offsets, isect_ids, emit_slot, temp, temp_bytes, row_cum, n_isects_dev);
}

namespace clmgs {

// ---- selection on the GPU: the batch's visibility filters without materialising radii[C,N]
// A: one ballot word per (camera, 64 Gaussians) + its popcount; row C = union over the cameras
//    (the rows the batch touches).  14 MB of bit words at 28 M x 4 cameras instead of 448 MB of radii.
//
// Two phases per block of 256 rows and group of 4 cameras.  Phase 1: every (row, camera) pair is
// classified from its camera-space mean and an upper bound of the 3-sigma radius taken from the
// largest scale (~45 instructions instead of the ~200 of the exact projection): certainly culled,
// certainly visible, or undecided (means near the image border, huge splats, depths at the planes:
// 1-2 % of the pairs), and the undecided ones are compacted into a list.  Phase 2: the EXACT
// project_fwd runs over the list with all lanes busy.  The result is the exact test's, bit for
// bit: both shortcuts are conservative (see vis_classify).
constexpr int VB_CH = 4;        // cameras per group
// (VB_CAM_F, VB_MAX_CAMS, vis_cam_kc, lds_cam, vis_candidate: vis_math.h -- shared with the deferred small-attribute Adam)

// false only if project_fwd is certain to return radius 0.  With Sc's eigenvalues <= smax^2 and
// |tx / z| <= lim:  b = (c00 + c11) / 2 <= smax^2 Kc / z^2 + eps2d =: B;  det > 0 => b^2 - det < b^2,
// so v1 = b + sqrt(max(0.01, b^2 - det)) <= 2 B + 0.1 and radius = ceil(3 sqrt(v1)) <= 3 sqrt(2B + 0.1) + 1.
// Margins (1 % on the radius, 2 px, 1e-5 on the depth planes) cover the fp32 rounding differences
// between this short evaluation and the exact one; NaNs fall through to the exact test.
// 0 = certainly culled, 1 = certainly visible, 2 = run the exact projection.
// "Certainly visible": depth strictly inside the planes, mean inside the image by a pixel (the
// radius is >= ceil(3 sqrt(eps2d)) = 2, so none of the four off-screen tests can fire), and
// B < 400, which keeps the rounding error of det = c00 c11 - c01^2 (<= 2^-23 * 2 (2B)^2 ~ 0.08) below
// its analytic floor 0.3 (c00' + c11') + 0.09, so the exact path's det > 0 test cannot fail.
__device__ __forceinline__ int vis_classify(const Cam& c, float kc, const float m[3], float smax2,
                                            float W, float H, float eps2d, float near_m, float far_m,
                                            float near_p, float far_p, bool accept_ok) {
  const float x = c.R[0] * m[0] + c.R[1] * m[1] + c.R[2] * m[2] + c.t[0];
  const float y = c.R[3] * m[0] + c.R[4] * m[1] + c.R[5] * m[2] + c.t[1];
  const float z = c.R[6] * m[0] + c.R[7] * m[1] + c.R[8] * m[2] + c.t[2];
  if (z < near_m || z > far_m) return 0;
  const float rz = __builtin_amdgcn_rcpf(z);
  const float mx = c.fx * x * rz + c.cx, my = c.fy * y * rz + c.cy;
  const float B = smax2 * rz * rz * kc + eps2d;
  if (accept_ok && z > near_p && z < far_p && B < 400.f && mx >= 1.f && mx <= W - 1.f && my >= 1.f &&
      my <= H - 1.f)
    return 1;
  const float Rb = 3.03f * __builtin_amdgcn_sqrtf(2.f * B + 0.1f) + 2.f;
  return (mx + Rb <= 0.f || mx - Rb >= W || my + Rb <= 0.f || my - Rb >= H) ? 0 : 2;
}

__global__ void __launch_bounds__(256)
visibility_bits_kernel(int C, int N, int W64, const float* __restrict__ means,
                       const float* __restrict__ quats_raw, const float* __restrict__ log_scales,
                       const float* __restrict__ viewmats, const float* __restrict__ Ks, float W, float H,
                       float eps2d, float near_plane, float far_plane, float radius_clip,
                       unsigned long long* __restrict__ bits, int64_t* __restrict__ counts,
                       const uint8_t* __restrict__ blk_flag) {
  // blk_flag != NULL: one byte per block of 256 rows; 0 = the caller KNOWS that no row of the block passes the cull in any
  // camera (clmgs_adam_small_deferred's candidate test, a superset of this one): the block's words are zeros, its rows
  // are not read
  __shared__ float cam_s[VB_MAX_CAMS][VB_CAM_F];
  __shared__ float row_s[256][10];  // mean, raw quaternion, scales of the block's rows
  __shared__ unsigned short cand[256 * VB_CH];  // row | camera-in-group << 8
  __shared__ int n_cand;
  __shared__ unsigned long long wb[VB_CH][4];
  __shared__ unsigned long long anyw[4];
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const float near_m = near_plane - (fabsf(near_plane) * 1e-5f + 1e-6f);
  const float far_m = far_plane + fabsf(far_plane) * 1e-5f;
  const float near_p = near_plane + (fabsf(near_plane) * 1e-5f + 1e-6f);
  const float far_p = far_plane - fabsf(far_plane) * 1e-5f;
  const bool accept_ok = radius_clip < 1.5f && eps2d >= 0.3f;  // radius >= 2 and the det floor above
  const int wid = tid >> 6;
  for (int c = tid; c < C; c += 256) {
    const Cam cam = load_cam(viewmats + 16 * c, Ks + 9 * c);
    float* f = cam_s[c];
    for (int i = 0; i < 9; ++i) f[i] = cam.R[i];
    f[9] = cam.t[0]; f[10] = cam.t[1]; f[11] = cam.t[2];
    f[12] = cam.fx; f[13] = cam.fy; f[14] = cam.cx; f[15] = cam.cy;
    f[16] = vis_cam_kc(cam, W, H);
  }
  const int n_rb = (W64 + 3) / 4;
  // the next block of rows is fetched while the current one is classified (the kernel is a chain of
  // load -> classify -> barrier per block; without the prefetch its time was memory + compute)
  float nm0 = 0.f, nm1 = 0.f, nm2 = 0.f, nl0 = 0.f, nl1 = 0.f, nl2 = 0.f;
  float4 nq = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int rbn) {
    const int n = rbn * 256 + tid;
    const bool live_n = rbn < n_rb && (!blk_flag || blk_flag[rbn]);
    const int nn = (live_n && n < N) ? n : 0;
    nm0 = means[3 * nn]; nm1 = means[3 * nn + 1]; nm2 = means[3 * nn + 2];
    nq = *reinterpret_cast<const float4*>(quats_raw + 4 * nn);
    nl0 = log_scales[3 * nn]; nl1 = log_scales[3 * nn + 1]; nl2 = log_scales[3 * nn + 2];
  };
  fetch(blockIdx.x);
  for (int rb = blockIdx.x; rb < n_rb; rb += gridDim.x) {
    const int n = rb * 256 + tid;
    const bool in = n < N;
    const float m[3] = {nm0, nm1, nm2};
    const float4 q4 = nq;
    const float s0 = __expf(nl0), s1 = __expf(nl1), s2 = __expf(nl2);
    fetch(rb + gridDim.x);
    if (blk_flag && !blk_flag[rb]) {  // (block-uniform)
      if (tid < (C + 1) * 4) {
        const int c = tid >> 2, w = rb * 4 + (tid & 3);
        if (w < W64) { bits[(size_t)c * W64 + w] = 0ull; counts[(size_t)c * W64 + w] = 0; }
      }
      for (int q = tid + 256; q < (C + 1) * 4; q += 256) {  // (more than 63 cameras)
        const int c = q >> 2, w = rb * 4 + (q & 3);
        if (w < W64) { bits[(size_t)c * W64 + w] = 0ull; counts[(size_t)c * W64 + w] = 0; }
      }
      continue;
    }
    const float smax = fmaxf(s0, fmaxf(s1, s2));
    const float smax2 = smax * smax;
    // degenerate rows (zero / non-finite quaternion, NaN scale) make the exact projection return
    // NaNs -> "culled"; they are never fast-accepted, the exact path decides
    const float qn2 = q4.x * q4.x + q4.y * q4.y + q4.z * q4.z + q4.w * q4.w;
    const bool row_ok = accept_ok && qn2 >= 1e-20f && qn2 <= 1e20f && (s0 + s1 + s2) < 1e30f;
    lds_barrier();  // cam_s ready / the previous block of rows is done with the shared arrays
    {
      float* r = row_s[tid];
      r[0] = m[0]; r[1] = m[1]; r[2] = m[2]; r[3] = q4.x; r[4] = q4.y; r[5] = q4.z; r[6] = q4.w;
      r[7] = s0; r[8] = s1; r[9] = s2;
    }
    if (tid < 4) anyw[tid] = 0ull;
    for (int c0 = 0; c0 < C; c0 += VB_CH) {
      const int cn = min(VB_CH, C - c0);
      if (tid < VB_CH * 4) wb[tid >> 2][tid & 3] = 0ull;
      if (tid == 0) n_cand = 0;
      lds_barrier();
      for (int cc = 0; cc < cn; ++cc) {  // phase 1: conservative test, compaction
        // wave-uniform camera: its constants arrive by scalar loads (SGPR operands, no VALU / LDS)
        const Cam cam = load_cam(viewmats + 16 * (c0 + cc), Ks + 9 * (c0 + cc));
        const int cls = in ? vis_classify(cam, cam_s[c0 + cc][16], m, smax2, W, H, eps2d, near_m, far_m,
                                          near_p, far_p, row_ok) : 0;
        const unsigned long long acc = __ballot(cls == 1);
        if (lane == 0 && acc) atomicOr(&wb[cc][wid], acc);
        const bool pass = cls == 2;
        const unsigned long long b = __ballot(pass);
        int base = 0;
        if (lane == 0 && b) base = atomicAdd(&n_cand, __popcll(b));
        base = __builtin_amdgcn_readfirstlane(base);
        if (pass) cand[base + __popcll(b & lt)] = (unsigned short)(tid | (cc << 8));
      }
      lds_barrier();
      const int nc = n_cand;
      for (int k = tid; k < nc; k += 256) {  // phase 2: exact projection of the survivors
        const int e = cand[k], r = e & 255, cc = e >> 8;
        const Cam cam = lds_cam(cam_s[c0 + cc]);
        const float* rr = row_s[r];
        const float mm[3] = {rr[0], rr[1], rr[2]};
        const float qq[4] = {rr[3], rr[4], rr[5], rr[6]};
        const float ss[3] = {rr[7], rr[8], rr[9]};
        const Proj p = project_fwd(cam, mm, qq, ss, W, H, eps2d, near_plane, far_plane, radius_clip);
        if (p.radius > 0) atomicOr(&wb[cc][r >> 6], 1ull << (r & 63));
      }
      lds_barrier();
      if (tid < cn * 4) {
        const int cc = tid >> 2, j = tid & 3, w = rb * 4 + j;
        if (w < W64) {
          const unsigned long long b = wb[cc][j];
          bits[(size_t)(c0 + cc) * W64 + w] = b;
          counts[(size_t)(c0 + cc) * W64 + w] = __popcll(b);
          if (b) atomicOr(&anyw[j], b);
        }
      }
    }
    lds_barrier();
    if (tid < 4 && rb * 4 + tid < W64) {
      bits[(size_t)C * W64 + rb * 4 + tid] = anyw[tid];
      counts[(size_t)C * W64 + rb * 4 + tid] = __popcll(anyw[tid]);
    }
  }
}

__global__ void __launch_bounds__(256)
visibility_candidates_kernel(int C, int N, int own_lo, int own_hi, const float* __restrict__ means,
                             const float* __restrict__ log_scales, const float* __restrict__ viewmats,
                             const float* __restrict__ Ks, float W, float H, float eps2d, float near_plane,
                             float far_plane, float pos_margin, float scale_gain, uint8_t* __restrict__ mask) {
  __shared__ float cam_s[VB_MAX_CAMS][VB_CAM_F];
  const int tid = threadIdx.x;
  for (int c = tid; c < C; c += 256) {
    const Cam cam = load_cam(viewmats + 16 * c, Ks + 9 * c);
    float* f = cam_s[c];
    for (int i = 0; i < 9; ++i) f[i] = cam.R[i];
    f[9] = cam.t[0]; f[10] = cam.t[1]; f[11] = cam.t[2];
    f[12] = cam.fx; f[13] = cam.fy; f[14] = cam.cx; f[15] = cam.cy;
    f[16] = vis_cam_kc(cam, W, H);
  }
  __syncthreads();
  const float near_m = near_plane - (fabsf(near_plane) * 1e-5f + 1e-6f);
  const float far_m = far_plane + fabsf(far_plane) * 1e-5f;
  for (int n = blockIdx.x * 256 + tid; n < N; n += gridDim.x * 256) {
    if (n >= own_lo && n < own_hi) {  // own rows are current: nothing to fetch
      mask[n] = 0;
      continue;
    }
    const float m[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
    const float lmax = fmaxf(log_scales[3 * n], fmaxf(log_scales[3 * n + 1], log_scales[3 * n + 2]));
    const float sg = __expf(lmax) * scale_gain;
    const float smax2g = sg * sg;
    bool any = !(smax2g < 1e30f);  // NaN / overflowing scales: let the exact pass decide
    for (int c = 0; c < C && !any; ++c)
      any = vis_candidate(lds_cam(cam_s[c]), cam_s[c][16], m, smax2g, pos_margin, W, H, eps2d, near_m, far_m);
    mask[n] = any ? 1 : 0;
  }
}

// B: after the inclusive scan of the (C+1) x W64 counts: every set bit writes its index at its rank.
__global__ void __launch_bounds__(256)
visibility_emit_kernel(int C, int N, int W64, const unsigned long long* __restrict__ bits,
                       const int64_t* __restrict__ scan, int64_t* __restrict__ out) {
  const int64_t total = (int64_t)(C + 1) * W64 * 64;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t word = t >> 6;
    const int bit = (int)(t & 63);
    const unsigned long long b = bits[word];
    if ((b >> bit) & 1ull) {
      const int64_t end = scan[word];  // inclusive over all (row, word) pairs before and including this
      const int64_t pos = end - __popcll(b) + __popcll(b & ((1ull << bit) - 1ull));
      const int w = (int)(word % W64);
      out[pos] = (int64_t)w * 64 + bit;
    }
  }
}

}  // namespace clmgs

extern "C" size_t clmgs_visibility_select_temp_bytes(int C, int N) {
  const size_t words = (size_t)(C + 1) * (size_t)((N + 63) / 64);
#ifdef CLMGS_PROFILE_BUILD
  const size_t ctrl = lb_scan_ctrl_bytes((int64_t)words);
#else
  const size_t ctrl = 0;
#endif
  return 2 * align_up(words * 8, 256) + max(scan_scratch_bytes((int64_t)words), ctrl) + 256;
}

extern "C" int clmgs_visibility_select_count_blocks(void* stream, int C, int N, const float* means,
                                                    const float* quats_raw, const float* log_scales,
                                                    const float* viewmats, const float* Ks, int width, int height,
                                                    float eps2d, float near_plane, float far_plane,
                                                    float radius_clip, void* temp, size_t temp_bytes,
                                                    int64_t* cum_totals, const uint8_t* block_flags);

extern "C" int clmgs_visibility_select_count(void* stream, int C, int N, const float* means,
                                             const float* quats_raw, const float* log_scales,
                                             const float* viewmats, const float* Ks, int width,
                                             int height, float eps2d, float near_plane,
                                             float far_plane, float radius_clip, void* temp,
                                             size_t temp_bytes, int64_t* cum_totals) {
  return clmgs_visibility_select_count_blocks(stream, C, N, means, quats_raw, log_scales, viewmats, Ks, width, height,
                                              eps2d, near_plane, far_plane, radius_clip, temp, temp_bytes, cum_totals,
                                              nullptr);
}

// block_flags != NULL: one byte per 256 rows, 0 = no row of the block can pass the cull (see visibility_bits_kernel)
extern "C" int clmgs_visibility_select_count_blocks(void* stream, int C, int N, const float* means,
                                                    const float* quats_raw, const float* log_scales,
                                                    const float* viewmats, const float* Ks, int width, int height,
                                                    float eps2d, float near_plane, float far_plane,
                                                    float radius_clip, void* temp, size_t temp_bytes,
                                                    int64_t* cum_totals, const uint8_t* block_flags) {
  CLMGS_CHECK_ARG(C >= 1 && C <= 64 && N >= 1 && width > 0 && height > 0);  // bsz <= 64 (engine.py)
  CLMGS_CHECK_ARG(means && quats_raw && log_scales && viewmats && Ks && temp && cum_totals);
  CLMGS_CHECK_ARG(temp_bytes >= clmgs_visibility_select_temp_bytes(C, N));
  hipStream_t s = (hipStream_t)stream;
  const int W64 = (N + 63) / 64;
  const size_t words = (size_t)(C + 1) * W64;
  char* base = (char*)temp;
  unsigned long long* bits = (unsigned long long*)base; base += align_up(words * 8, 256);
  int64_t* counts = (int64_t*)base; base += align_up(words * 8, 256);
  int64_t* scratch = (int64_t*)base;
  hipLaunchKernelGGL(visibility_bits_kernel, dim3(min(ceil_div((int64_t)W64 * 64, 256), 256 * 8)),
                     dim3(256), 0, s, C, N, W64, means, quats_raw, log_scales, viewmats, Ks,
                     (float)width, (float)height, eps2d, near_plane, far_plane, radius_clip, bits, counts, block_flags);
  CLMGS_LAUNCH_CHECK();
  int rc;
#ifdef CLMGS_PROFILE_BUILD
  if (binning_route() == BIN_LOOKBACK) {  // one launch (decoupled look-back) + the memset of its control words
    CLMGS_HIP(hipMemsetAsync(scratch, 0, lb_scan_ctrl_bytes((int64_t)words), s));
    rc = lb_inclusive_scan_i64(s, (int64_t)words, counts, scratch);
  } else
#endif
  rc = inclusive_scan_i64(s, (int64_t)words, counts, scratch);
  if (rc) return rc;
  // cum_totals[r] = number of set bits in rows 0..r (device array of C+1, read back by the caller)
  for (int r = 0; r <= C; ++r)
    CLMGS_HIP(hipMemcpyAsync(cum_totals + r, counts + (size_t)(r + 1) * W64 - 1, sizeof(int64_t),
                             hipMemcpyDeviceToDevice, s));
  return 0;
}

extern "C" int clmgs_visibility_select_emit(void* stream, int C, int N, const void* temp,
                                            int64_t* out) {
  CLMGS_CHECK_ARG(C >= 1 && N >= 1 && temp && out);
  const int W64 = (N + 63) / 64;
  const size_t words = (size_t)(C + 1) * W64;
  const char* base = (const char*)temp;
  const unsigned long long* bits = (const unsigned long long*)base; base += align_up(words * 8, 256);
  const int64_t* scan = (const int64_t*)base;
  hipLaunchKernelGGL(visibility_emit_kernel, dim3(min(ceil_div((int64_t)words * 64, 256), 256 * 32)),
                     dim3(256), 0, (hipStream_t)stream, C, N, W64, bits, scan, out);
  CLMGS_LAUNCH_CHECK();
  return 0;
}


// Rows outside [own_lo, own_hi) that may be visible in any of the C cameras when their stored mean is off by up to
// pos_margin and their largest scale by a factor up to scale_gain: mask[N] (u8).  See vis_candidate.
extern "C" int clmgs_visibility_candidates(void* stream, int C, int N, int own_lo, int own_hi, const float* means,
                                           const float* log_scales, const float* viewmats, const float* Ks,
                                           int width, int height, float eps2d, float near_plane, float far_plane,
                                           float pos_margin, float scale_gain, uint8_t* mask) {
  CLMGS_CHECK_ARG(C >= 1 && C <= 64 && N >= 1 && width > 0 && height > 0);
  CLMGS_CHECK_ARG(means && log_scales && viewmats && Ks && mask);
  CLMGS_CHECK_ARG(pos_margin >= 0.f && scale_gain >= 1.f && own_lo >= 0 && own_hi >= own_lo);
  hipLaunchKernelGGL(visibility_candidates_kernel, dim3(min(ceil_div((int64_t)N, 256), 256 * 16)), dim3(256), 0,
                     (hipStream_t)stream, C, N, own_lo, own_hi, means, log_scales, viewmats, Ks, (float)width,
                     (float)height, eps2d, near_plane, far_plane, pos_margin, scale_gain, mask);
  CLMGS_LAUNCH_CHECK();
  return 0;
}


// Device error word of the look-back primitives: *out = bits (1 = a scan look-back, 2 = a sort look-back gave up
// after its spin bound: a workgroup of the launch never published -- results of that call are invalid); cleared when
// `reset`.  Synchronises the device.
extern "C" int clmgs_device_errors(uint32_t* out, int reset) {
  CLMGS_CHECK_ARG(out);
  CLMGS_HIP(hipDeviceSynchronize());
  CLMGS_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(clmgs::g_dev_err), sizeof(uint32_t)));
  if (reset && *out) {
    const uint32_t z = 0;
    CLMGS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(clmgs::g_dev_err), &z, sizeof(z)));
  }
  return 0;
}
