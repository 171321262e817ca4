// Shared helpers for the gfx950 kernels of libclmgs_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/clmgs.h"

namespace clmgs {

void set_error(const char* fmt, ...);

#define CLMGS_CHECK_ARG(cond)                                                       \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      clmgs::set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond);   \
      return CLMGS_EINVAL;                                                          \
    }                                                                               \
  } while (0)

#define CLMGS_HIP(expr)                                                             \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      clmgs::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                \
                       hipGetErrorString(_e));                                      \
      return (int)_e;                                                               \
    }                                                                               \
  } while (0)

#define CLMGS_LAUNCH_CHECK() CLMGS_HIP(hipGetLastError())

// Workgroup barrier that orders LDS only: unlike __syncthreads() (whose workgroup fence also
// drains vmcnt) it lets global loads issued before it stay in flight, so a prefetch of the next
// block of input survives the barriers of the current one.  Use only where every cross-thread
// hand-over between the barriers goes through LDS.
#if defined(__HIPCC__)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#endif

// Partial-gradient line of the atomic-free rasterize backward: 9 floats (x y ca cb | cc r g b | o) in PART_F4 float4.
// 4 = one 64 B line per (row, tile) intersection, a fourth float4 of zeros.  Round 4 measured the 48 B form (PART_F4 =
// 3, 16 B x I_emitted less written and read back): rasterize_bwd 1.76 -> 1.82 ms, preprocess_bwd 0.59 -> 0.61 ms solo --
// SLOWER: a 48 B line straddles two 64 B memory lines, and because slots are numbered in ROW order the two halves
// of a memory line are written by different tiles at different times (two partial-line writes instead of one full
// one).  Kept at 4; the constant is the single place to change it (clmgs_rasterize_partials_bytes reports it).
constexpr int PART_F4 = 4;

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed,
// speed only).  Remap so each XCD's L2 sees one contiguous range of work items.
// Bijective for any n (MI355X guide, "XCD swizzle must be bijective").
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned n) {
  const unsigned X = 8;
  unsigned xcd = b % X, slot = b / X;
  unsigned q = n / X, r = n % X;
  unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// ------------------------------------------------------------- wave64 sum
// DPP butterfly inside rows of 16, then row broadcasts (gfx9 DPP row_bcast).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  int iv = __float_as_int(v);
  int r = __builtin_amdgcn_update_dpp(0, iv, CTRL, 0xf, 0xf, true);
  return v + __int_as_float(r);
}

// Every lane ends up with ... only lane 63 is guaranteed to hold the full sum.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0x111>(v);  // row_shr:1
  v = dpp_add<0x112>(v);  // row_shr:2
  v = dpp_add<0x114>(v);  // row_shr:4
  v = dpp_add<0x118>(v);  // row_shr:8   -> lane 15 of each row holds the row sum
  {
    int iv = __float_as_int(v);
    int r = __builtin_amdgcn_update_dpp(0, iv, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1,3
    v += __int_as_float(r);
  }
  {
    int iv = __float_as_int(v);
    int r = __builtin_amdgcn_update_dpp(0, iv, 0x143, 0xc, 0xf, true);  // row_bcast:31 -> rows 2,3
    v += __int_as_float(r);
  }
  return v;
}

// Four wave-wide sums for the price of ~1.6: gfx950's v_permlane32_swap / v_permlane16_swap fold
// the 32- and 16-lane levels of TWO values per instruction ("reduce-scatter" butterfly), after
// which one register carries a,c,b,d in rows 0..3 and four row_shr DPP adds finish all four.
// Result: lane 15 holds sum(a), lane 31 sum(c), lane 47 sum(b), lane 63 sum(d).
__device__ __forceinline__ float wave_sum4_rows(float a, float b, float c, float d) {
  auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  const float s1 = __uint_as_float(r1[0]) + __uint_as_float(r1[1]);  // lanes 0-31: a, 32-63: b
  auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(d), false, false);
  const float s2 = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);  // lanes 0-31: c, 32-63: d
  auto r3 = __builtin_amdgcn_permlane16_swap(__float_as_uint(s1), __float_as_uint(s2), false, false);
  float u = __uint_as_float(r3[0]) + __uint_as_float(r3[1]);          // rows: a, c, b, d
  u = dpp_add<0x111>(u);
  u = dpp_add<0x112>(u);
  u = dpp_add<0x114>(u);
  u = dpp_add<0x118>(u);
  return u;
}

__device__ __forceinline__ float wave_sum(float v) {
  v = wave_sum_to_lane63(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace clmgs
