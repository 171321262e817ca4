// Hand-written stable LSD radix sort of (key, int32 value) pairs and an inclusive int64 scan for
// gfx950 -- the sorting / scanning primitives of the tile-binning stage (isect.hip).
//
// One pass per 8-bit digit, three kernels per pass:
//   histogram : every block counts the digits of its 4096-key chunk (LDS atomics) and writes a
//               digit-major [256][n_blocks] table;
//   scan      : exclusive scan of every digit's row (one block per digit) + the 256 row totals
//               (their exclusive scan, the global base of every digit, is redone by every
//               scatter block: cheaper than a fourth dependent launch per pass);
//   scatter   : every block re-reads its chunk (16 keys per thread kept in registers); the stable
//               rank of a key inside its wave comes from 8 ballots (lanes with the same digit) +
//               a popcount of the lanes below; the (round, wave) groups are ordered by ONE
//               per-digit prefix over a [16][4][256] LDS count table (2 barriers per 4096 keys).
// Stability (equal digits keep input order) is what makes multi-pass LSD sorting correct and
// what the two-level binning relies on for depth ties.
#pragma once
#include "common.h"

namespace clmgs {

constexpr int RS_THREADS = 256;
// Keys per thread: 4 (1024 keys per block, 12-20 KB of LDS).  8 is ~10 % faster solo, but the
// sorts run concurrently with the alpha-blend kernels, whose small blocks keep most of each CU's
// LDS and registers occupied: 35-42 KB blocks were starved (1-2 ms per pass instead of 0.03-0.12).
#ifndef CLMGS_RS_ITEMS
#define CLMGS_RS_ITEMS 4
#endif
constexpr int RS_DEFAULT_ITEMS = CLMGS_RS_ITEMS;
constexpr int RS_MIN_CHUNK = RS_THREADS * RS_DEFAULT_ITEMS;

template <typename KeyT, int RS_ITEMS>
__global__ void __launch_bounds__(RS_THREADS)
radix_hist_kernel(int64_t n, const KeyT* __restrict__ keys, int shift, int n_blocks,
                  uint32_t* __restrict__ table /*[256][n_blocks]*/, const int64_t* __restrict__ n_dev) {
  constexpr int RS_CHUNK = RS_THREADS * RS_ITEMS;
  if (n_dev) n = min(n, *n_dev);  // device-side count: the launch was sized for the capacity `n`
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
#pragma unroll 4
  for (int it = 0; it < RS_ITEMS; ++it) {
    const int64_t i = base + it * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[(unsigned)((keys[i] >> shift) & 0xFF)], 1u);
  }
  __syncthreads();
  table[(size_t)threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of every digit's row of the [256][n_blocks] table (block d = digit d) + row totals.
// 256-thread blocks on purpose (like every kernel of the binning chain): next to the alpha-blend
// kernels, whose one-wave blocks refill every freed wave slot, a 1024-thread block (16 slots on ONE
// CU at once) was not dispatched until the tile kernel had drained -- 1.5 ms instead of 5 us.
constexpr int RS_SCAN_THREADS = 256;
static __global__ void __launch_bounds__(RS_SCAN_THREADS)
radix_scan_rows_kernel(int n_blocks, uint32_t* __restrict__ table, uint32_t* __restrict__ row_tot) {
  __shared__ uint32_t wsum[RS_SCAN_THREADS / 64];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  uint32_t* row = table + (size_t)blockIdx.x * n_blocks;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int base = 0; base < n_blocks; base += RS_SCAN_THREADS) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < n_blocks ? row[i] : 0u;
    uint32_t x = v;  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += wsum[w];
    const uint32_t carry = carry_s;
    if (i < n_blocks) row[i] = carry + woff + x - v;
    __syncthreads();
    if (threadIdx.x == RS_SCAN_THREADS - 1) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) row_tot[blockIdx.x] = carry_s;
}

// Round 4 variants (the default route; the round-3 kernels above stay for CLMGS_BINNING=legacy):
// histogram -- one block counts HB consecutive chunks: HB x RS_ITEMS independent loads in flight per thread instead
// of RS_ITEMS, a quarter of the blocks (the kernel is a load -> LDS atomic -> store latency chain, not bandwidth).
constexpr int RS_HIST_HB = 4;
template <typename KeyT, int RS_ITEMS>
__global__ void __launch_bounds__(RS_THREADS)
radix_hist_multi_kernel(int64_t n, const KeyT* __restrict__ keys, int shift, int n_blocks,
                        uint32_t* __restrict__ table /*[256][n_blocks]*/, const int64_t* __restrict__ n_dev) {
  constexpr int RS_CHUNK = RS_THREADS * RS_ITEMS;
  if (n_dev) n = min(n, *n_dev);
  __shared__ uint32_t h[RS_HIST_HB][256];
#pragma unroll
  for (int c = 0; c < RS_HIST_HB; ++c) h[c][threadIdx.x] = 0;
  __syncthreads();
  const int c0 = blockIdx.x * RS_HIST_HB;
  KeyT k[RS_HIST_HB][RS_ITEMS];
  bool have[RS_HIST_HB][RS_ITEMS];
#pragma unroll
  for (int c = 0; c < RS_HIST_HB; ++c)
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
      const int64_t i = (int64_t)(c0 + c) * RS_CHUNK + it * RS_THREADS + threadIdx.x;
      have[c][it] = i < n;
      k[c][it] = have[c][it] ? keys[i] : KeyT{};
    }
#pragma unroll
  for (int c = 0; c < RS_HIST_HB; ++c)
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it)
      if (have[c][it]) atomicAdd(&h[c][(unsigned)((k[c][it] >> shift) & 0xFF)], 1u);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < RS_HIST_HB; ++c)
    if (c0 + c < n_blocks) table[(size_t)threadIdx.x * n_blocks + c0 + c] = h[c][threadIdx.x];
}

// row scan -- thread t of block d owns the contiguous segment [t K, (t+1) K) of digit d's row (K = ceil(n_blocks /
// 256)): independent loads, ONE block-wide scan of the 256 segment sums, a second sweep writing the exclusive values;
// the round-3 kernel walked the row in 256-wide steps with three barriers each (36 dependent steps at 9.3 M keys).
static __global__ void __launch_bounds__(RS_SCAN_THREADS)
radix_scan_rows_seg_kernel(int n_blocks, uint32_t* __restrict__ table, uint32_t* __restrict__ row_tot) {
  __shared__ uint32_t wsum[RS_SCAN_THREADS / 64];
  uint32_t* row = table + (size_t)blockIdx.x * n_blocks;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int K = (n_blocks + RS_SCAN_THREADS - 1) / RS_SCAN_THREADS;
  const int a = min(n_blocks, (int)threadIdx.x * K), b = min(n_blocks, a + K);
  uint32_t sum = 0;
#pragma unroll 8
  for (int i = a; i < b; ++i) sum += row[i];
  uint32_t x = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wid; ++w) woff += wsum[w];
  uint32_t run = woff + x - sum;
#pragma unroll 8
  for (int i = a; i < b; ++i) {
    const uint32_t v = row[i];
    row[i] = run;
    run += v;
  }
  if (threadIdx.x == RS_SCAN_THREADS - 1) row_tot[blockIdx.x] = run;
}

template <typename KeyT, typename ValT, int RS_ITEMS, bool SPLIT = false>
__global__ void __launch_bounds__(RS_THREADS)
radix_scatter_kernel(int64_t n, const KeyT* __restrict__ keys_in, const ValT* __restrict__ vals_in,
                     KeyT* __restrict__ keys_out, ValT* __restrict__ vals_out, int shift,
                     int n_blocks, const uint32_t* __restrict__ table,
                     const uint32_t* __restrict__ row_tot /*[256] keys per digit*/,
                     const int64_t* __restrict__ n_dev, int32_t* __restrict__ out_a = nullptr,
                     int32_t* __restrict__ out_b = nullptr) {
  constexpr int RS_CHUNK = RS_THREADS * RS_ITEMS;
  if (n_dev) n = min(n, *n_dev);
  // cnt[round][wave][digit]: first the number of keys of that digit in that (round, wave), then
  // (after the per-digit prefix) the offset of that group inside the block's digit bucket.
  __shared__ uint16_t cnt[RS_ITEMS][4][256];
  __shared__ uint32_t cursor[256];
  __shared__ uint32_t wtot[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  {
    // global base of digit d = exclusive scan of the 256 row totals, redone by every block (a
    // wave scan + 3 adds) instead of a separate one-block launch between the scan and this kernel
    const uint32_t tot = row_tot[threadIdx.x];
    uint32_t x = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wtot[wid] = x;
    uint32_t* z = reinterpret_cast<uint32_t*>(&cnt[0][0][0]);
    for (int i = threadIdx.x; i < RS_ITEMS * 4 * 256 / 2; i += RS_THREADS) z[i] = 0u;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += wtot[w];
    cursor[threadIdx.x] = table[(size_t)threadIdx.x * n_blocks + blockIdx.x] + woff + x - tot;
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
  KeyT key[RS_ITEMS];
  ValT val[RS_ITEMS];
  int meta[RS_ITEMS];  // digit | rank << 8 | have << 16
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    const int64_t i = base + it * RS_THREADS + threadIdx.x;
    const bool have = i < n;
    key[it] = 0; val[it] = ValT{};
    if (have) { key[it] = keys_in[i]; val[it] = vals_in[i]; }
    const unsigned digit = (unsigned)((key[it] >> shift) & 0xFF);
    unsigned long long peers = __ballot(have);  // lanes of this wave holding the same digit
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((digit >> b) & 1u);
      peers &= ((digit >> b) & 1u) ? m : ~m;
    }
    const int rank = __popcll(peers & lt);
    if (have && rank == 0) cnt[it][wid][digit] = (uint16_t)__popcll(peers);
    meta[it] = (int)digit | (rank << 8) | ((int)have << 16);
  }
  __syncthreads();
  __shared__ uint32_t dstart[256];  // first local (block-sorted) index of every digit
  {  // thread d owns digit d: exclusive prefix over the 64 (round, wave) groups, in input order
    unsigned running = 0;
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned c = cnt[it][w][threadIdx.x];
        cnt[it][w][threadIdx.x] = (uint16_t)running;
        running += c;
      }
    }
    // exclusive scan of the 256 digit totals of this block
    unsigned x = running;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wtot[wid] = x;
    __syncthreads();
    unsigned woff = 0;
    for (int w = 0; w < wid; ++w) woff += wtot[w];
    dstart[threadIdx.x] = woff + x - running;
  }
  __syncthreads();
  // Reorder inside the block through LDS so that the global stores below go out in runs of
  // consecutive addresses (lanes next to each other hold neighbours of the same digit bucket)
  // instead of 64 unrelated 4-byte writes per instruction.
  __shared__ KeyT skey[RS_CHUNK];
  __shared__ ValT sval[RS_CHUNK];
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    if (meta[it] >> 16) {
      const int digit = meta[it] & 0xFF, rank = (meta[it] >> 8) & 0xFF;
      const uint32_t li = dstart[digit] + cnt[it][wid][digit] + (uint32_t)rank;
      skey[li] = key[it];
      sval[li] = val[it];
    }
  }
  __syncthreads();
  const int n_here = (int)min((int64_t)RS_CHUNK, n - base);
#pragma unroll 4
  for (int it = 0; it < RS_ITEMS; ++it) {
    const int li = it * RS_THREADS + threadIdx.x;
    if (li < n_here) {
      const KeyT k = skey[li];
      const unsigned digit = (unsigned)((k >> shift) & 0xFF);
      const uint32_t pos = cursor[digit] + (uint32_t)li - dstart[digit];
      keys_out[pos] = k;
      if constexpr (SPLIT) {  // final pass of an (a, b)-pair payload: the two halves go to their own arrays
        const ValT v = sval[li];
        out_a[pos] = v.x;
        out_b[pos] = v.y;
      } else {
        vals_out[pos] = sval[li];
      }
    }
  }
}

static inline size_t radix_table_bytes(int64_t n) {
  const int64_t nb = (n + RS_MIN_CHUNK - 1) / RS_MIN_CHUNK;  // upper bound over both block sizes
  return ((size_t)(nb + 1) * 256 * sizeof(uint32_t) + 255) / 256 * 256;  // table + 256 digit bases
}

// Sorts n pairs on key bits [begin_bit, end_bit).  Input in (keysA, valsA); keysB / valsB are
// scratch of the same size; `table` is radix_table_bytes(n) of scratch.  The sorted VALUES are
// written to vals_final (distinct from valsA / valsB); *keys_sorted tells which key buffer holds
// the sorted keys.
template <typename KeyT, typename ValT, int ITEMS>
static int radix_sort_pairs_impl(hipStream_t s, int64_t n, KeyT* keysA, KeyT* keysB, ValT* valsA,
                                 ValT* valsB, ValT* vals_final, int begin_bit, int end_bit,
                                 uint32_t* table, KeyT** keys_sorted, const int64_t* n_dev = nullptr,
                                 int32_t* split_a = nullptr, int32_t* split_b = nullptr, bool round4 = true,
                                 bool hist0_done = false) {
  constexpr int RS_CHUNK = RS_THREADS * ITEMS;
  const int passes = (end_bit - begin_bit + 7) / 8;
  const int n_blocks = (int)((n + RS_CHUNK - 1) / RS_CHUNK);
  uint32_t* row_tot = table + (size_t)n_blocks * 256;  // keys per digit
  KeyT* ksrc = keysA;
  ValT* vsrc = valsA;
  if (passes == 0) {
    CLMGS_HIP(hipMemcpyAsync(vals_final, valsA, sizeof(ValT) * (size_t)n, hipMemcpyDeviceToDevice, s));
    *keys_sorted = keysA;
    return 0;
  }
  for (int p = 0; p < passes; ++p) {
    const int shift = begin_bit + 8 * p;
    KeyT* kdst = (ksrc == keysA) ? keysB : keysA;
    ValT* vdst = (p == passes - 1) ? vals_final : ((vsrc == valsA) ? valsB : valsA);
    if (round4) {
      // hist0_done: the producer of the keys (isect2_emit_hist_kernel) has already left the first digit's counts in `table`
      if (!(hist0_done && p == 0))
        hipLaunchKernelGGL((radix_hist_multi_kernel<KeyT, ITEMS>), dim3((n_blocks + RS_HIST_HB - 1) / RS_HIST_HB),
                           dim3(RS_THREADS), 0, s, n, ksrc, shift, n_blocks, table, n_dev);
      hipLaunchKernelGGL(radix_scan_rows_seg_kernel, dim3(256), dim3(RS_SCAN_THREADS), 0, s, n_blocks, table, row_tot);
    } else {
      hipLaunchKernelGGL((radix_hist_kernel<KeyT, ITEMS>), dim3(n_blocks), dim3(RS_THREADS), 0, s, n, ksrc, shift,
                         n_blocks, table, n_dev);
      hipLaunchKernelGGL(radix_scan_rows_kernel, dim3(256), dim3(RS_SCAN_THREADS), 0, s, n_blocks, table, row_tot);
    }
    if constexpr (sizeof(ValT) == 8) {
      if (split_a && p == passes - 1) {
        hipLaunchKernelGGL((radix_scatter_kernel<KeyT, ValT, ITEMS, true>), dim3(n_blocks), dim3(RS_THREADS), 0, s, n, ksrc,
                           vsrc, kdst, (ValT*)nullptr, shift, n_blocks, table, row_tot, n_dev, split_a, split_b);
        CLMGS_LAUNCH_CHECK();
        ksrc = kdst;
        continue;
      }
    }
    hipLaunchKernelGGL((radix_scatter_kernel<KeyT, ValT, ITEMS>), dim3(n_blocks), dim3(RS_THREADS), 0, s, n, ksrc, vsrc,
                       kdst, vdst, shift, n_blocks, table, row_tot, n_dev, (int32_t*)nullptr, (int32_t*)nullptr);
    CLMGS_LAUNCH_CHECK();
    ksrc = kdst;
    vsrc = vdst;
  }
  *keys_sorted = ksrc;
  return 0;
}

template <typename KeyT, typename ValT = int32_t>
static int radix_sort_pairs(hipStream_t s, int64_t n, KeyT* keysA, KeyT* keysB, ValT* valsA,
                            ValT* valsB, ValT* vals_final, int begin_bit, int end_bit,
                            uint32_t* table, KeyT** keys_sorted, const int64_t* n_dev = nullptr, bool round4 = true) {
  return radix_sort_pairs_impl<KeyT, ValT, RS_DEFAULT_ITEMS>(s, n, keysA, keysB, valsA, valsB, vals_final,
                                                      begin_bit, end_bit, table, keys_sorted, n_dev, nullptr, nullptr,
                                                      round4);
}

// Inclusive scan of 256 values (one per thread, thread order) -> inclusive result; wsum[0..3] (LDS) get the four wave
// totals.  Two barriers.
__device__ __forceinline__ long long block_incl_scan_i64(long long v, long long* wsum) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  long long x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const long long y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  __syncthreads();  // wsum free (previous use)
  if (lane == 63) wsum[wid] = x;
  __syncthreads();
  long long off = 0;
  for (int w = 0; w < wid; ++w) off += wsum[w];
  return x + off;
}

// ------------------------------------------------------------- inclusive scan of int64
constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS = 8;
constexpr int SC_CHUNK = SC_THREADS * SC_ITEMS;

__device__ __forceinline__ int64_t wave_incl_scan_i64(int64_t x, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int64_t y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  return x;
}

// pass 1: per-block inclusive scan in place + block totals
static __global__ void __launch_bounds__(SC_THREADS)
scan_i64_blocks_kernel(int64_t n, int64_t* __restrict__ data, int64_t* __restrict__ block_tot) {
  __shared__ int64_t wsum[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * SC_CHUNK + (int64_t)threadIdx.x * SC_ITEMS;
  int64_t v[SC_ITEMS];
  int64_t run = 0;
#pragma unroll
  for (int k = 0; k < SC_ITEMS; ++k) {
    v[k] = (base + k < n) ? data[base + k] : 0;
    run += v[k];
    v[k] = run;  // inclusive inside the thread
  }
  const int64_t incl = wave_incl_scan_i64(run, lane);
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  int64_t off = incl - run;
  for (int w = 0; w < wid; ++w) off += wsum[w];
#pragma unroll
  for (int k = 0; k < SC_ITEMS; ++k)
    if (base + k < n) data[base + k] = v[k] + off;
  if (threadIdx.x == SC_THREADS - 1) block_tot[blockIdx.x] = off + run;
}

// pass 2: exclusive scan of the block totals, one block (256 threads, see radix_scan_rows_kernel)
static __global__ void __launch_bounds__(SC_THREADS)
scan_i64_totals_kernel(int nb, int64_t* __restrict__ tot) {
  __shared__ int64_t wsum[SC_THREADS / 64];
  __shared__ int64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int base = 0; base < nb; base += SC_THREADS) {
    const int i = base + threadIdx.x;
    const int64_t v = i < nb ? tot[i] : 0;
    const int64_t x = wave_incl_scan_i64(v, lane);
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    int64_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += wsum[w];
    const int64_t carry = carry_s;
    if (i < nb) tot[i] = carry + woff + x - v;
    __syncthreads();
    if (threadIdx.x == SC_THREADS - 1) carry_s = carry + woff + x;
    __syncthreads();
  }
}

// pass 3: add the block offsets; last_out (optional) receives the grand total data[n-1]
static __global__ void __launch_bounds__(SC_THREADS)
scan_i64_add_kernel(int64_t n, int64_t* __restrict__ data, const int64_t* __restrict__ tot,
                    int64_t* __restrict__ last_out) {
  const int64_t off = tot[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * SC_CHUNK;
  for (int k = threadIdx.x; k < SC_CHUNK; k += SC_THREADS)
    if (base + k < n) {
      const int64_t v = data[base + k] + off;
      data[base + k] = v;
      if (last_out && base + k == n - 1) *last_out = v;
    }
}

static inline size_t scan_scratch_bytes(int64_t n) {
  return (((n + SC_CHUNK - 1) / SC_CHUNK) * sizeof(int64_t) + 255) / 256 * 256 + 256;
}

static int inclusive_scan_i64(hipStream_t s, int64_t n, int64_t* data, int64_t* scratch,
                              int64_t* last_out = nullptr) {
  if (n <= 0) return 0;
  const int nb = (int)((n + SC_CHUNK - 1) / SC_CHUNK);
  hipLaunchKernelGGL(scan_i64_blocks_kernel, dim3(nb), dim3(SC_THREADS), 0, s, n, data, scratch);
  hipLaunchKernelGGL(scan_i64_totals_kernel, dim3(1), dim3(SC_THREADS), 0, s, nb, scratch);
  hipLaunchKernelGGL(scan_i64_add_kernel, dim3(nb), dim3(SC_THREADS), 0, s, n, data, scratch, last_out);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

}  // namespace clmgs
