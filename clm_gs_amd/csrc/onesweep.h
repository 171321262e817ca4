// Single-launch primitives of the tile-binning stage for gfx950 (isect.hip): an inclusive int64 scan and one
// radix-sort pass per launch, both by decoupled look-back between workgroups of ONE launch.
//
// Why it was built (profiles/r03_kernel_stats.csv): the binning chain of a camera was ~30 launches of 6-60 us each --
// three kernels per 8-bit digit (block histograms, a scan of the [256][blocks] table, the scatter) and three per scan.
// Here a sort pass is ONE launch (digit counts of ALL passes are taken up front, by the kernel that produces the
// keys) and a scan is folded into the kernel that produces its input.
// What was MEASURED (round 4): slower than the three-launch form on MI355X -- 60-79 us per pass
// over 3.3 M keys against 30 + 11 + 11, 155-205 us over 9.3 M keys against 106 + 34 + 34, for every block shape
// (1 / 2 / 4 / 8 rounds of 1024 keys per workgroup) and every look-back window (2-16 words per thread and step); the
// single-word scan 40 us for 2.2 M elements against 37 us.  The route is selected with CLMGS_BINNING=lookback (the
// default is `fused`, isect.hip) and kept as a tested alternative: element-for-element equal lists.
//
// Inter-workgroup protocol (MI355X guide, Guideline 16, form R2 "the data IS the flag"): a workgroup takes a TICKET
// from a device counter -- its chunk of the input is the ticket, so every lower chunk belongs to a workgroup that is
// already running (forward progress without any assumption on dispatch order) -- and publishes per-chunk
// aggregates as single naturally aligned words {flag, value} with relaxed AGENT-scope atomic stores (write-through
// past the non-coherent per-XCD L2s); successors poll those words with relaxed agent-scope atomic loads.  No payload
// travels separately from its flag, hence no fences.  Every polled word is zeroed by ONE hipMemsetAsync of the
// call's control block before the first kernel of the call; every spin is bounded (a timeout raises the library's
// device error word, read by clmgs_device_errors(), and the kernel still terminates).
//
// A look-back step of a sort pass reads a 1 KB row of 256 per-digit words; with everything dispatched at once the
// PREFIX frontier grows quadratically in the number of steps, so a workgroup of the first wave reads ~sqrt(2 k W)
// rows (k = its ticket, W = words in flight per thread).
#pragma once
#include "common.h"

namespace clmgs {

uint32_t* device_error_word();  // isect.hip: the address of the library's zero-initialised device error word

constexpr uint32_t LB_AGG = 1u, LB_PREFIX = 2u;
constexpr unsigned LB_SPIN_LIMIT = 1u << 22;  // polls (~1 us each): seconds -- only a lost workgroup gets there

__device__ __forceinline__ uint32_t lb_ld32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lb_st32(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long lb_ld64(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lb_st64(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------
// Chunk-level exclusive prefix of an int64 quantity by decoupled look-back, for kernels that process ticket-ordered
// chunks and want "sum of all lower chunks".  status[chunk]: {flag:2 | value:62}.  Called by ALL 256 threads of the
// block with the block's chunk total (same value in every thread); returns the exclusive prefix to every thread.
// `sh` is 2 words of LDS scratch.  wave 0 looks 64 predecessors back per step.
__device__ __forceinline__ long long lb_chunk_prefix(unsigned long long* __restrict__ status, int chunk,
                                                     long long total, long long* sh, uint32_t* err) {
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0)
    lb_st64(status + chunk, ((unsigned long long)(chunk == 0 ? LB_PREFIX : LB_AGG) << 62) | (unsigned long long)total);
  if (tid < 64) {
    long long excl = 0;
    int pos = chunk - 1;
    unsigned spins = 0;
    while (pos >= 0) {
      const int idx = pos - lane;
      unsigned long long w = ((unsigned long long)LB_PREFIX << 62);  // before chunk 0: prefix 0
      if (idx >= 0) w = lb_ld64(status + idx);
      const unsigned flag = (unsigned)(w >> 62);
      const unsigned long long pre = __ballot(flag == LB_PREFIX);
      const unsigned long long inv = __ballot(flag == 0u);
      // lanes nearer than the first PREFIX must all be ready
      const int first_pre = pre ? (int)__builtin_ctzll(pre) : 64;
      const unsigned long long need = first_pre >= 63 ? ~0ull : ((1ull << (first_pre + 1)) - 1ull);
      if (inv & need) {
        if (++spins > LB_SPIN_LIMIT) { if (lane == 0) atomicOr(err, 1u); break; }
        __builtin_amdgcn_s_sleep(2);
        continue;
      }
      long long v = (lane <= first_pre) ? (long long)(w & ((1ull << 62) - 1ull)) : 0;
      v = wave_sum_i64(v);
      excl += v;
      if (pre) break;
      pos -= 64;
    }
    if (lane == 0) {
      sh[0] = excl;
      if (chunk > 0)
        lb_st64(status + chunk, ((unsigned long long)LB_PREFIX << 62) | (unsigned long long)(excl + total));
    }
  }
  __syncthreads();
  return sh[0];
}

// (block_incl_scan_i64 lives in radix.h: the default route uses it too)

// ---------------------------------------------------------------------------------------------------------------
// Stand-alone single-launch inclusive scan of int64 data in place (visibility select).  Chunk = 2048 elements.
constexpr int LBS_ITEMS = 8;
constexpr int LBS_CHUNK = 256 * LBS_ITEMS;

__global__ void __launch_bounds__(256)
lb_scan_i64_kernel(int64_t n, int64_t* __restrict__ data, unsigned long long* __restrict__ status,
                   uint32_t* __restrict__ ticket_ctr, int64_t* __restrict__ last_out, uint32_t* err) {
  __shared__ long long wsum[4];
  __shared__ long long sh[2];
  __shared__ int ticket_s;
  const int n_chunks = (int)((n + LBS_CHUNK - 1) / LBS_CHUNK);
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) ticket_s = (int)atomicAdd(ticket_ctr, 1u);
    __syncthreads();
    const int chunk = ticket_s;
    if (chunk >= n_chunks) return;
    const int64_t base = (int64_t)chunk * LBS_CHUNK + (int64_t)threadIdx.x * LBS_ITEMS;
    long long v[LBS_ITEMS];
    long long run = 0;
#pragma unroll
    for (int k = 0; k < LBS_ITEMS; ++k) {
      v[k] = (base + k < n) ? data[base + k] : 0;
      run += v[k];
      v[k] = run;
    }
    const long long incl = block_incl_scan_i64(run, wsum);
    const long long total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const long long excl = lb_chunk_prefix(status, chunk, total, sh, err);
    const long long off = excl + incl - run;
#pragma unroll
    for (int k = 0; k < LBS_ITEMS; ++k)
      if (base + k < n) {
        const long long o = v[k] + off;
        data[base + k] = o;
        if (last_out && base + k == n - 1) *last_out = o;
      }
  }
}

static inline size_t lb_scan_ctrl_bytes(int64_t n) {  // status words + ticket (zeroed by the caller's memset)
  return align_up((size_t)((n + LBS_CHUNK - 1) / LBS_CHUNK) * 8 + 64, 256);
}

// ctrl: lb_scan_ctrl_bytes(n) bytes, ZEROED on `s` before this call (the caller's one memset).
static int lb_inclusive_scan_i64(hipStream_t s, int64_t n, int64_t* data, void* ctrl, int64_t* last_out = nullptr) {
  if (n <= 0) return 0;
  const int n_chunks = (int)((n + LBS_CHUNK - 1) / LBS_CHUNK);
  unsigned long long* status = (unsigned long long*)ctrl;
  uint32_t* ticket = (uint32_t*)(status + n_chunks);
  hipLaunchKernelGGL(lb_scan_i64_kernel, dim3(min(n_chunks, 2048)), dim3(256), 0, s, n, data, status, ticket, last_out,
                     device_error_word());
  CLMGS_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// One LSD radix-sort pass (8-bit digit, stable) per launch.  32-bit keys; ValT = int32_t or int2.
//   ghist[256]  : digit counts of THIS pass over all n keys (taken up front by the producer of the keys)
//   status      : [n_blocks][256] words {flag:2 | count:30}, zeroed by the call's memset
//   ticket_ctr  : zeroed likewise
// SPLIT: ValT == int2 and this is the final pass: .x goes to out_a, .y to out_b (flatten_ids / emit_slot).
constexpr int OS_ROUND_ITEMS = 4;                      // keys per thread and ranking round (1024 per round)
constexpr int OS_ROUND = 256 * OS_ROUND_ITEMS;
#ifndef CLMGS_OS_WINDOW
#define CLMGS_OS_WINDOW 4
#endif
constexpr int OS_WINDOW = CLMGS_OS_WINDOW;                           // look-back words in flight per thread

static inline int os_blocks(int64_t n, int rounds) { return (int)((n + (int64_t)OS_ROUND * rounds - 1) / ((int64_t)OS_ROUND * rounds)); }
// (sized for the thinnest block shape, 1 round: the shape is a run-time choice)
static inline size_t os_status_bytes(int64_t n) { return align_up((size_t)os_blocks(n, 1) * 256 * 4, 256); }

template <typename ValT, bool SPLIT, int OS_ROUNDS>
__global__ void __launch_bounds__(256)
onesweep_pass_kernel(int64_t n, const int64_t* __restrict__ n_dev, const uint32_t* __restrict__ keys_in,
                     const ValT* __restrict__ vals_in, uint32_t* __restrict__ keys_out, ValT* __restrict__ vals_out,
                     int32_t* __restrict__ out_a, int32_t* __restrict__ out_b, int shift,
                     const uint32_t* __restrict__ ghist, uint32_t* __restrict__ status,
                     uint32_t* __restrict__ ticket_ctr, uint32_t* __restrict__ err) {
  __shared__ uint16_t cnt[OS_ROUND_ITEMS][4][256];
  __shared__ uint32_t cursor[256];
  __shared__ uint32_t dstart[256];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t wtot[4];
  __shared__ uint32_t skey[OS_ROUND];
  __shared__ ValT sval[OS_ROUND];
  __shared__ int ticket_s;
  constexpr int OS_CHUNK = OS_ROUND * OS_ROUNDS;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t cap = n;
  if (n_dev) n = min(n, *n_dev);
  if (tid == 0) ticket_s = (int)atomicAdd(ticket_ctr, 1u);
  hist[tid] = 0;
  __syncthreads();
  const int ticket = ticket_s;
  const int64_t base = (int64_t)ticket * OS_CHUNK;
  if (base >= n) return;  // (whole block; every later ticket is empty as well, nobody waits for this one)
  // ---- global base of every digit: exclusive scan of the pass's digit counts
  uint32_t gbase;
  {
    const uint32_t tot = ghist[tid];
    uint32_t x = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wtot[wid] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += wtot[w];
    gbase = woff + x - tot;
  }
  // ---- digit counts of this block's chunk
  const int n_here = (int)min((int64_t)OS_CHUNK, n - base);
#pragma unroll
  for (int k = 0; k < OS_ROUNDS * OS_ROUND_ITEMS; ++k) {
    const int li = k * 256 + tid;
    if (li < n_here) atomicAdd(&hist[(keys_in[base + li] >> shift) & 0xFFu], 1u);
  }
  __syncthreads();
  const uint32_t my_cnt = hist[tid];
  uint32_t* row = status + (size_t)ticket * 256;
  lb_st32(row + tid, ((ticket == 0 ? LB_PREFIX : LB_AGG) << 30) | my_cnt);
  // ---- decoupled look-back, thread d = digit d, OS_WINDOW predecessors per step
  uint32_t excl = 0;
  if (ticket > 0) {
    int t = ticket - 1;
    unsigned spins = 0;
    bool done = false;
    while (!done) {
      uint32_t w[OS_WINDOW];
#pragma unroll
      for (int j = 0; j < OS_WINDOW; ++j)
        w[j] = (t - j >= 0) ? lb_ld32(status + (size_t)(t - j) * 256 + tid) : (LB_PREFIX << 30);
      bool stalled = false;
#pragma unroll
      for (int j = 0; j < OS_WINDOW; ++j) {
        if (done || stalled) continue;
        const uint32_t f = w[j] >> 30;
        if (f == 0u) { stalled = true; continue; }
        excl += w[j] & 0x3FFFFFFFu;
        --t;
        if (f == LB_PREFIX) done = true;
      }
      if (stalled) {
        if (++spins > LB_SPIN_LIMIT) { atomicOr(err, 2u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    lb_st32(row + tid, (LB_PREFIX << 30) | (excl + my_cnt));
  }
  cursor[tid] = gbase + excl;
  const unsigned long long lt = (1ull << lane) - 1ull;
  // ---- rank + scatter, 1024 keys per round (stable inside the round, rounds in input order)
  for (int r = 0; r < OS_ROUNDS; ++r) {
    const int rbase = r * OS_ROUND;
    if (rbase >= n_here) break;  // uniform
    {
      uint32_t* z = reinterpret_cast<uint32_t*>(&cnt[0][0][0]);
      for (int i = tid; i < OS_ROUND_ITEMS * 4 * 256 / 2; i += 256) z[i] = 0u;
    }
    __syncthreads();
    uint32_t key[OS_ROUND_ITEMS];
    ValT val[OS_ROUND_ITEMS];
    int meta[OS_ROUND_ITEMS];
#pragma unroll
    for (int it = 0; it < OS_ROUND_ITEMS; ++it) {
      const int li = rbase + it * 256 + tid;
      const bool have = li < n_here;
      key[it] = 0; val[it] = ValT{};
      if (have) { key[it] = keys_in[base + li]; val[it] = vals_in[base + li]; }
      const unsigned digit = (key[it] >> shift) & 0xFFu;
      unsigned long long peers = __ballot(have);
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
    unsigned running = 0;
    {
#pragma unroll
      for (int it = 0; it < OS_ROUND_ITEMS; ++it) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const unsigned c = cnt[it][w][tid];
          cnt[it][w][tid] = (uint16_t)running;
          running += c;
        }
      }
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
      dstart[tid] = woff + x - running;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < OS_ROUND_ITEMS; ++it) {
      if (meta[it] >> 16) {
        const int digit = meta[it] & 0xFF, rank = (meta[it] >> 8) & 0xFF;
        const uint32_t li = dstart[digit] + cnt[it][wid][digit] + (uint32_t)rank;
        skey[li] = key[it];
        sval[li] = val[it];
      }
    }
    __syncthreads();
    const int n_round = min(OS_ROUND, n_here - rbase);
#pragma unroll
    for (int it = 0; it < OS_ROUND_ITEMS; ++it) {
      const int li = it * 256 + tid;
      if (li < n_round) {
        const uint32_t k = skey[li];
        const unsigned digit = (k >> shift) & 0xFFu;
        const uint32_t pos = cursor[digit] + (uint32_t)li - dstart[digit];
        if ((int64_t)pos < cap) {  // (always, unless a look-back timed out: never write out of bounds)
          keys_out[pos] = k;
          if constexpr (SPLIT) {
            const ValT v = sval[li];
            out_a[pos] = v.x;
            out_b[pos] = v.y;
          } else {
            vals_out[pos] = sval[li];
          }
        }
      }
    }
    __syncthreads();
    cursor[tid] += running;  // digit `tid`'s keys of this round are placed
  }
}

}  // namespace clmgs

namespace clmgs {
// One pass, block shape chosen at run time (rounds of 1024 keys per workgroup: 1, 2, 4 or 8).
template <typename ValT, bool SPLIT>
static void launch_onesweep_pass(hipStream_t s, int rounds, int64_t n, const int64_t* n_dev, const uint32_t* keys_in,
                                 const ValT* vals_in, uint32_t* keys_out, ValT* vals_out, int32_t* out_a, int32_t* out_b,
                                 int shift, const uint32_t* ghist, uint32_t* status, uint32_t* ticket, uint32_t* err) {
  const dim3 grid(os_blocks(n, rounds));
#define CLMGS_OS_LAUNCH(R)                                                                                          \
  hipLaunchKernelGGL((onesweep_pass_kernel<ValT, SPLIT, R>), grid, dim3(256), 0, s, n, n_dev, keys_in, vals_in,      \
                     keys_out, vals_out, out_a, out_b, shift, ghist, status, ticket, err)
  switch (rounds) {
    case 1: CLMGS_OS_LAUNCH(1); break;
    case 2: CLMGS_OS_LAUNCH(2); break;
    case 4: CLMGS_OS_LAUNCH(4); break;
    default: CLMGS_OS_LAUNCH(8); break;
  }
#undef CLMGS_OS_LAUNCH
}
}  // namespace clmgs
