// CLM offload data movers, visibility bitmaps, row-sparse Adam and densification
// statistics (gfx950).  Replaces the clm_kernels entry points used at
// strategies/clm_offload/engine.py:152-153,200-204,227-232,499-505,622-636,789-825
// and optimizer.py:76-88; stats: strategies/clm_offload/gaussian_model.py:833-851.
//
// All of these are HBM- (or host-link-) bound row/byte movers: a 48-float row is
// 12 lanes x 16 B, so 64 lanes move 5 1/3 rows per instruction fully coalesced.
#include <stdlib.h>

#include "common.h"
#include "vis_math.h"
#include "radix.h"

namespace clmgs {
uint32_t* device_error_word();  // isect.hip: the library's zero-initialised device error word


template <typename IdxT>
__device__ __forceinline__ int64_t row_of(const void* idx, int64_t i) {
  return idx ? (int64_t)reinterpret_cast<const IdxT*>(idx)[i] : i;
}

// cols % 4 == 0 fast path: one float4 per lane.
template <typename IdxT, bool ADD>
__global__ void __launch_bounds__(256)
rows_move_f4_kernel(float* __restrict__ dst, const float* __restrict__ src,
                    const void* __restrict__ dst_idx, const void* __restrict__ src_idx,
                    int64_t n_rows, int f4_per_row) {
  const int64_t total = n_rows * f4_per_row;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / f4_per_row;
    const int k = (int)(i - r * f4_per_row);
    const int64_t sr = row_of<IdxT>(src_idx, r), dr = row_of<IdxT>(dst_idx, r);
    const float4 v = reinterpret_cast<const float4*>(src)[sr * f4_per_row + k];
    float4* d = reinterpret_cast<float4*>(dst) + dr * f4_per_row + k;
    if (ADD) {
      float4 o = *d;
      o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
      *d = o;
    } else {
      *d = v;
    }
  }
}

template <typename IdxT, bool ADD>
__global__ void __launch_bounds__(256)
rows_move_f1_kernel(float* __restrict__ dst, const float* __restrict__ src,
                    const void* __restrict__ dst_idx, const void* __restrict__ src_idx,
                    int64_t n_rows, int cols) {
  const int64_t total = n_rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int k = (int)(i - r * cols);
    const int64_t sr = row_of<IdxT>(src_idx, r), dr = row_of<IdxT>(dst_idx, r);
    const float v = src[sr * cols + k];
    if (ADD) dst[dr * cols + k] += v; else dst[dr * cols + k] = v;
  }
}

template <bool ADD>
static int rows_move(void* stream, float* dst, const float* src, const void* dst_idx,
                     const void* src_idx, int idx_is_64, int64_t n_rows, int cols,
                     int grid_blocks) {
  CLMGS_CHECK_ARG(n_rows >= 0 && cols > 0);
  if (n_rows == 0) return 0;
  CLMGS_CHECK_ARG(dst && src);
  hipStream_t s = (hipStream_t)stream;
  const bool f4 = (cols % 4 == 0) && (((uintptr_t)dst | (uintptr_t)src) % 16 == 0);
  const int64_t total = f4 ? n_rows * (cols / 4) : n_rows * cols;
  int grid = grid_blocks > 0 ? grid_blocks : min(ceil_div(total, 256), 256 * 8);
  if (f4) {
    if (idx_is_64)
      hipLaunchKernelGGL((rows_move_f4_kernel<int64_t, ADD>), dim3(grid), dim3(256), 0, s, dst, src,
                         dst_idx, src_idx, n_rows, cols / 4);
    else
      hipLaunchKernelGGL((rows_move_f4_kernel<int32_t, ADD>), dim3(grid), dim3(256), 0, s, dst, src,
                         dst_idx, src_idx, n_rows, cols / 4);
  } else {
    if (idx_is_64)
      hipLaunchKernelGGL((rows_move_f1_kernel<int64_t, ADD>), dim3(grid), dim3(256), 0, s, dst, src,
                         dst_idx, src_idx, n_rows, cols);
    else
      hipLaunchKernelGGL((rows_move_f1_kernel<int32_t, ADD>), dim3(grid), dim3(256), 0, s, dst, src,
                         dst_idx, src_idx, n_rows, cols);
  }
  CLMGS_LAUNCH_CHECK();
  return 0;
}

// Message of the camera-DP locality exchange, step F (clm_gs_amd/dp.py publish_rows): [chunk, cols] rows then chunk row
// ids.  rows[i] = stamp[own[i]] == step ? table[own[i]] : 0 (a row whose line is an earlier step's travels as zeros),
// ids[i] = bits of int32(own[i] - lo).  One pass instead of gather + stamp gather + mask + id conversion + copy.
__global__ void __launch_bounds__(256)
publish_pack_kernel(float* __restrict__ msg, const float* __restrict__ table, const int64_t* __restrict__ own,
                    const int32_t* __restrict__ stamp, int step, int64_t lo, int64_t n, int64_t chunk, int f4_per_row) {
  const int64_t total = n * f4_per_row;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / f4_per_row;
    const int k = (int)(i - r * f4_per_row);
    const int64_t row = own[r];
    const bool live = !stamp || stamp[row] == step;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) v = reinterpret_cast<const float4*>(table)[row * f4_per_row + k];
    reinterpret_cast<float4*>(msg)[i] = v;
    if (k == 0) reinterpret_cast<int32_t*>(msg)[chunk * f4_per_row * 4 + r] = (int32_t)(row - lo);
  }
}

// ------------------------------------------------------------------ row groups of the host-resident batch
// For every row a batch touches: the FIRST and the LAST camera of the batch that uses it, from the visibility bitmap
// (bit bsz-1-i = camera i, as the reference encodes it, clm_offload/engine.py:137-153).  Keys for two stable one-digit
// radix sorts: kf = first camera (255 = the row is already staged, not "late"), kl = last camera.
template <typename T>
__global__ void __launch_bounds__(256)
host_group_keys_kernel(int64_t n, const int64_t* __restrict__ touched, const T* __restrict__ bitmap, int bsz,
                       const uint8_t* __restrict__ staged, uint32_t* __restrict__ kf, uint32_t* __restrict__ kl,
                       int32_t* __restrict__ rows32) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = touched[j];
    unsigned long long bm = (unsigned long long)(typename std::make_unsigned<T>::type)bitmap[row];
    if (bsz < 64) bm &= (1ull << bsz) - 1ull;
    int first = 254, last = 254;  // a touched row is in at least one filter; 254 only for an inconsistent input
    if (bm) {
      first = (bsz - 1) - (63 - __clzll((long long)bm));
      last = (bsz - 1) - (__ffsll((long long)bm) - 1);
    }
    kf[j] = (staged && staged[row]) ? 255u : (uint32_t)first;
    kl[j] = (uint32_t)last;
    rows32[j] = (int32_t)row;
  }
}

// counts[0..bsz) = late rows per first camera, counts[bsz..2 bsz) = rows per last camera, counts[2 bsz] = late rows;
// slot_of[late_sorted[k]] = slot0 + k
__global__ void __launch_bounds__(256)
host_group_finish_kernel(int64_t n, int bsz, const uint32_t* __restrict__ tot_first, const uint32_t* __restrict__ tot_last,
                         const int32_t* __restrict__ late_sorted, int slot0, int32_t* __restrict__ slot_of,
                         int64_t* __restrict__ counts) {
  const int64_t n_late = n - (int64_t)tot_first[255];
  if (blockIdx.x == 0 && threadIdx.x <= 2 * bsz) {
    const int t = threadIdx.x;
    counts[t] = t < bsz ? (int64_t)tot_first[t] : (t < 2 * bsz ? (int64_t)tot_last[t - bsz] : n_late);
  }
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n_late; k += (int64_t)gridDim.x * blockDim.x)
    slot_of[late_sorted[k]] = slot0 + (int32_t)k;
}

// ------------------------------------------------------------------ bitmaps
template <typename T>
__global__ void scatter_to_bit_kernel(T* __restrict__ bitmap, const int64_t* __restrict__ filter,
                                      int64_t n, int bit) {
  using U = typename std::make_unsigned<T>::type;
  const U m = (U)((U)1 << bit);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    // ids inside one filter are unique -> plain RMW, no atomics needed
    U* p = reinterpret_cast<U*>(bitmap) + filter[i];
    *p = (U)(*p | m);
  }
}

template <typename T>
__global__ void extract_ffs_kernel(const T* __restrict__ bitmap, int64_t N, uint8_t* __restrict__ ffs) {
  using U = typename std::make_unsigned<T>::type;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long v = (unsigned long long)(U)bitmap[i];
    ffs[i] = (uint8_t)(v ? (__ffsll((long long)v)) : 0);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
pair_overlap_kernel(const T* __restrict__ bitmap, int64_t N, int bsz, int32_t* __restrict__ cnt) {
  using U = typename std::make_unsigned<T>::type;
  __shared__ int sh[64];
  for (int i = threadIdx.x; i < 64; i += blockDim.x) sh[i] = 0;
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long v = (unsigned long long)(U)bitmap[i];
    // micro-batch m lives in bit (bsz-1-m); adjacent pairs = v & (v << 1)
    unsigned long long both = v & (v << 1);
    while (both) {
      const int b = __ffsll((long long)both) - 1;  // bit of micro-batch m, m+1 in bit b-1
      both &= both - 1;
      atomicAdd(&sh[bsz - 1 - b], 1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < bsz - 1; i += blockDim.x)
    if (sh[i]) atomicAdd(&cnt[i], sh[i]);
}

__global__ void set_signal_kernel(int32_t* signal, int idx, int32_t value) {
  __threadfence_system();
  __hip_atomic_store(signal + idx, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// --------------------------------------------------------------------- Adam
// update term  m / (sqrt(v) / sqrt(bc2) + eps)  with the hardware sqrt and reciprocal (1 ulp each):
// the lazy replay is VALU-bound on the ~20-instruction IEEE sqrt + divide sequences (0.92 G VALU
// instructions per launch), and a 2-ulp error of the update is far below one ulp of the parameter
// it is subtracted from.  All three Adam kernels share it, so eager, lazy and packed agree.
__device__ __forceinline__ float adam_ratio(float m, float v, float inv_sqrt_bc2, float eps) {
  return m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) * inv_sqrt_bc2 + eps);
}

// VEC = 4 when cols % 4 == 0 (the [N,48] SH rows): one 16 B access per array per thread, the
// per-step scalars are computed once per thread instead of once per element.
template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<4> { using type = float4; };
template <int VEC> __device__ __forceinline__ void vload(const float* a, float (&x)[VEC]) {
  if constexpr (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(a); x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w; }
  else x[0] = *a;
}
template <int VEC> __device__ __forceinline__ void vstore(float* a, const float (&x)[VEC]) {
  if constexpr (VEC == 4) *reinterpret_cast<float4*>(a) = make_float4(x[0], x[1], x[2], x[3]);
  else *a = x[0];
}

template <typename IdxT, int VEC>
__global__ void __launch_bounds__(256)
adam_rows_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                 float* __restrict__ v, const void* __restrict__ rows,
                 const uint8_t* __restrict__ mask, int64_t n_rows, int cols,
                 const float* __restrict__ col_lr, float beta1, float beta2, float ob1, float ob2,
                 float eps, float inv_bc1, float inv_sqrt_bc2, float grad_scale, int zero_grad) {
  const int cv = cols / VEC;
  const int64_t total = n_rows * cv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv;
    const int k = (int)(i - r * cv) * VEC;
    const int64_t row = row_of<IdxT>(rows, r);
    if (mask && !mask[row]) continue;
    const int64_t o = row * cols + k;
    float gg[VEC], mm[VEC], vv[VEC], pp[VEC], lr[VEC];
    vload<VEC>(m + o, mm); vload<VEC>(v + o, vv); vload<VEC>(p + o, pp); vload<VEC>(col_lr + k, lr);
    if (g) vload<VEC>(g + o, gg);  // g == NULL: rows known to have zero gradient
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      const float gs = g ? gg[c] * grad_scale : 0.f;
      mm[c] = beta1 * mm[c] + ob1 * gs;
      vv[c] = beta2 * vv[c] + ob2 * gs * gs;
      pp[c] -= (lr[c] * inv_bc1) * adam_ratio(mm[c], vv[c], inv_sqrt_bc2, eps);
      gg[c] = 0.f;
    }
    vstore<VEC>(m + o, mm); vstore<VEC>(v + o, vv); vstore<VEC>(p + o, pp);
    if (zero_grad && g) vstore<VEC>(g + o, gg);
  }
}

// Deferred ("lazy") dense Adam for rows that receive no gradient.  A row untouched for k steps
// only needs   m <- b1 m,  v <- b2 v,  p <- p - lr/bc1_j * m / (sqrt(v)/sqrt(bc2_j) + eps)
// k times -- nothing that depends on other rows or on data produced in between -- so the k updates
// can be REPLAYED in registers the next time the row is needed, instead of streaming all N rows
// through HBM every batch.  Same operations in the same order per element as the eager update
// (bias corrections come from a running product in float instead of a double pow: ~1e-7 relative).
// Steps older than max_replay are folded analytically (m *= b1^d, v *= b2^d): their parameter
// increments are below half an ulp of p by then.
// beta^x for the bias corrections of the deferred steps: v_exp_f32(x * v_log_f32(beta)) -- three instructions; libm's powf
// is ~80 and was called four to six times per thread (round 5: the kernel issued ~1 G wave instructions per batch,
// a third of its run time at the issue peak).  Relative error ~1e-6 at the step counts where 1 - beta^x is not yet 1.
__device__ __forceinline__ float pow_beta(float beta, float x) {
  return __builtin_amdgcn_exp2f(x * __builtin_amdgcn_logf(beta));
}

template <typename IdxT, int VEC>
__global__ void __launch_bounds__(256)
adam_catch_up_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                     const int32_t* __restrict__ last_step, const void* __restrict__ rows,
                     int64_t n_rows, int cols, const float* __restrict__ col_lr, float beta1,
                     float beta2, float eps, int to_step, int bias_correction, int max_replay,
                     float* __restrict__ g, const int32_t* __restrict__ g_step, float grad_scale,
                     float ob1, float ob2,  // 1 - beta, rounded from double like the eager kernel's
                     int keep_grad) {  // 1: the consumed gradient row is left as it is (first-touch producers)
  const int cv = cols / VEC;
  const int64_t total = n_rows * cv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv;
    const int k = (int)(i - r * cv) * VEC;
    const int64_t row = row_of<IdxT>(rows, r);
    int a = last_step[row];
    // DEFERRED gradient step: g[row] holds the gradient of optimizer step g_step[row] (> last_step: not
    // applied yet).  It is applied here, at its own step, between the zero-gradient replays before and
    // after it -- the same operations in the same order as an eager update at the end of that batch, but
    // the row's p / m / v make ONE round trip per touch instead of two (catch-up + end-of-batch Adam).
    const int gs = g_step ? g_step[row] : 0;
    const bool pending = gs > a && gs <= to_step;
    if (to_step - a <= 0) continue;
    const int64_t o = row * cols + k;
    float mm[VEC], vv[VEC], pp[VEC], lr[VEC], gg[VEC];
    // everything the row needs is requested at once (one round trip to HBM, not two): the rows handed
    // to this kernel are the ones a batch touches, and nearly all of them have work to do
    vload<VEC>(m + o, mm); vload<VEC>(v + o, vv); vload<VEC>(p + o, pp); vload<VEC>(col_lr + k, lr);
    if (pending) vload<VEC>(g + o, gg);
    if (!pending) {  // all-zero state (rows that never had a gradient): every replayed step is the identity
      // (m, v stay 0 and p -= lr * 0 / (0 + eps)), so neither the loop nor the stores are needed
      bool any_state = false;
#pragma unroll
      for (int c = 0; c < VEC; ++c) any_state |= (mm[c] != 0.f) | (vv[c] != 0.f);
      if (!any_state) continue;
    }
    // zero-gradient steps from+1 .. to: the first max_replay are replayed exactly; by then the first moment
    // has decayed by beta1^max_replay (1e-12 at 0.9^256, far less with the batch-scaled betas), the remaining
    // parameter increments are below float resolution and only the moments' decay is applied
    auto replay = [&](int from, int to) {
      const int missed = to - from;
      if (missed <= 0) return;
      const int n = min(missed, max_replay);
      float pw1 = pow_beta(beta1, (float)(from + 1)), pw2 = pow_beta(beta2, (float)(from + 1));
      for (int j = 0; j < n; ++j) {
        float inv_bc1 = 1.f, inv_sqrt_bc2 = 1.f;
        if (bias_correction) {
          inv_bc1 = 1.f / (1.f - pw1);
          inv_sqrt_bc2 = 1.f / sqrtf(1.f - pw2);
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
          mm[c] *= beta1;
          vv[c] *= beta2;
          pp[c] -= (lr[c] * inv_bc1) * adam_ratio(mm[c], vv[c], inv_sqrt_bc2, eps);
        }
        pw1 *= beta1; pw2 *= beta2;
      }
      if (missed > n) {
        const int d = missed - n;
        const float f1 = pow_beta(beta1, (float)d), f2 = pow_beta(beta2, (float)d);
#pragma unroll
        for (int c = 0; c < VEC; ++c) { mm[c] *= f1; vv[c] *= f2; }
      }
    };
    if (pending) {
      replay(a, gs - 1);
      float inv_bc1 = 1.f, inv_sqrt_bc2 = 1.f;
      if (bias_correction) {
        inv_bc1 = 1.f / (1.f - pow_beta(beta1, (float)gs));
        inv_sqrt_bc2 = 1.f / sqrtf(1.f - pow_beta(beta2, (float)gs));
      }
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        const float gsc = gg[c] * grad_scale;
        mm[c] = beta1 * mm[c] + ob1 * gsc;
        vv[c] = beta2 * vv[c] + ob2 * gsc * gsc;
        pp[c] -= (lr[c] * inv_bc1) * adam_ratio(mm[c], vv[c], inv_sqrt_bc2, eps);
        gg[c] = 0.f;
      }
      if (!keep_grad) vstore<VEC>(g + o, gg);  // consumed: cleared for producers that accumulate
      a = gs;
    }
    replay(a, to_step);
    vstore<VEC>(m + o, mm); vstore<VEC>(v + o, vv); vstore<VEC>(p + o, pp);
  }
}

// The [N,48] form of the same kernel (the SH-row tables; what the engine calls every batch): a thread walks
// float4 column k of rows r, r + S, r + 2S ... -- the grid is a multiple of 3 blocks, so the stride is a multiple of
// the 12 float4 of a row and neither k nor the learning rates change: no 64-bit division per element (the generic
// kernel's i / cv), the learning rates are loaded once; the next row id is requested a whole iteration ahead, and
// the row's stamps, p, m, v and its gradient row are all requested together (the generic form waits for the stamp
// before it asks for the data: three dependent HBM latencies per element instead of two).
template <typename IdxT>
__global__ void __launch_bounds__(256)
adam_catch_up48_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                       const int32_t* __restrict__ last_step, const void* __restrict__ rows,
                       unsigned n_rows, const float* __restrict__ col_lr, float beta1,
                       float beta2, float eps, int to_step, int bias_correction, int max_replay,
                       float* __restrict__ g, const int32_t* __restrict__ g_step, float grad_scale,
                       float ob1, float ob2, int keep_grad) {
  constexpr int VEC = 4;
  const unsigned t0 = blockIdx.x * 256u + threadIdx.x;
  const unsigned r_stride = gridDim.x * 256u / 12u;  // (grid is a multiple of 3)
  unsigned r = t0 / 12u;
  const int k = (int)(t0 - r * 12u) * VEC;
  if (r >= n_rows) return;
  float lr[VEC];
  vload<VEC>(col_lr + k, lr);
  int64_t row_next = row_of<IdxT>(rows, (int64_t)r);
  // the row id is requested TWO iterations ahead and the row's stamps one iteration ahead: when the row's turn comes it
  // is already known whether a gradient waits, so the gradient line (192 B of the 1 344 B a row costs) is requested,
  // with everything else, only for the rows that have one (round 5; a batch finds a waiting gradient on the rows the
  // batch before it touched too -- about a third of them)
  int a_next = last_step[row_next];
  int gs_next = g_step ? g_step[row_next] : 0;
  int64_t row_next2 = (r + r_stride < n_rows) ? row_of<IdxT>(rows, (int64_t)(r + r_stride)) : row_next;
  for (; r < n_rows; r += r_stride) {
    const int64_t row = row_next;
    int a = a_next;
    const int gs = gs_next;
    row_next = row_next2;
    if (r + r_stride < n_rows) {
      a_next = last_step[row_next];
      gs_next = g_step ? g_step[row_next] : 0;
      if (r + 2 * r_stride < n_rows) row_next2 = row_of<IdxT>(rows, (int64_t)(r + 2 * r_stride));
    }
    const int64_t o = row * 48 + k;
    float mm[VEC], vv[VEC], pp[VEC], gg[VEC];
    const bool pending = gs > a && gs <= to_step;
    // a row that is already current -- an earlier camera's list of the same batch held it too (the engine brings the
    // rows up to date camera by camera) -- costs its stamps only
    if (to_step - a <= 0) continue;
    vload<VEC>(m + o, mm); vload<VEC>(v + o, vv); vload<VEC>(p + o, pp);
    if (pending) vload<VEC>(g + o, gg);
    if (!pending) {
      bool any_state = false;
#pragma unroll
      for (int c = 0; c < VEC; ++c) any_state |= (mm[c] != 0.f) | (vv[c] != 0.f);
      if (!any_state) continue;
    }
    auto replay = [&](int from, int to) {
      const int missed = to - from;
      if (missed <= 0) return;
      const int n = min(missed, max_replay);
      float pw1 = pow_beta(beta1, (float)(from + 1)), pw2 = pow_beta(beta2, (float)(from + 1));
      for (int j = 0; j < n; ++j) {
        float inv_bc1 = 1.f, inv_sqrt_bc2 = 1.f;
        if (bias_correction) {
          inv_bc1 = 1.f / (1.f - pw1);
          inv_sqrt_bc2 = 1.f / sqrtf(1.f - pw2);
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
          mm[c] *= beta1;
          vv[c] *= beta2;
          pp[c] -= (lr[c] * inv_bc1) * adam_ratio(mm[c], vv[c], inv_sqrt_bc2, eps);
        }
        pw1 *= beta1; pw2 *= beta2;
      }
      if (missed > n) {
        const int d = missed - n;
        const float f1 = pow_beta(beta1, (float)d), f2 = pow_beta(beta2, (float)d);
#pragma unroll
        for (int c = 0; c < VEC; ++c) { mm[c] *= f1; vv[c] *= f2; }
      }
    };
    if (pending) {
      replay(a, gs - 1);
      float inv_bc1 = 1.f, inv_sqrt_bc2 = 1.f;
      if (bias_correction) {
        inv_bc1 = 1.f / (1.f - pow_beta(beta1, (float)gs));
        inv_sqrt_bc2 = 1.f / sqrtf(1.f - pow_beta(beta2, (float)gs));
      }
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        const float gsc = gg[c] * grad_scale;
        mm[c] = beta1 * mm[c] + ob1 * gsc;
        vv[c] = beta2 * vv[c] + ob2 * gsc * gsc;
        pp[c] -= (lr[c] * inv_bc1) * adam_ratio(mm[c], vv[c], inv_sqrt_bc2, eps);
        gg[c] = 0.f;
      }
      if (!keep_grad) vstore<VEC>(g + o, gg);
      a = gs;
    }
    replay(a, to_step);
    vstore<VEC>(m + o, mm); vstore<VEC>(v + o, vv); vstore<VEC>(p + o, pp);
  }
}

// Dense Adam of the 11 GPU-resident attributes when the engine keeps the packed [N,12] mirror
// (xyz 3 | opacity 1 | scaling 3 | rotation 4 | pad) and accumulates their gradients in a packed
// [N,12] table: one pass reads the gradient row, updates p / m / v of the four parameter tensors
// (torch.optim.Adam's own state tensors), refreshes the mirror row and zeroes the gradient row.
struct SmallAdam {
  float* p[4]; float* m[4]; float* v[4];  // xyz [N,3], opacity [N,1], scaling [N,3], rotation [N,4]
  float lr[4];
};

// One block = 256 rows.  The packed gradient rows are staged in LDS with coalesced 16 B loads;
// each of the four tensors is then walked as a FLAT float4 stream (its [256, w] slab is contiguous),
// the gradient of element (row, e) comes from LDS, the new parameter goes back to LDS, and the
// mirror rows leave with coalesced 16 B stores: every global access is a full-wave 16 B/lane run.
constexpr int SA_ROWS = 256;

__device__ __forceinline__ float adam_elem(float& m, float& v, float p, float g, float lr_bc, float beta1,
                                           float beta2, float ob1, float ob2, float eps,
                                           float inv_sqrt_bc2) {
  // explicit roundings (no contraction left to the compiler): the <false> and <true> forms of the kernel below, and
  // any future caller, produce the same bits from the same inputs
  m = __builtin_fmaf(beta1, m, __fmul_rn(ob1, g));
  v = __builtin_fmaf(beta2, v, __fmul_rn(__fmul_rn(ob2, g), g));
  const float ratio = __fmul_rn(m, __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_sqrtf(v), inv_sqrt_bc2, eps)));
  return __builtin_fmaf(-lr_bc, ratio, p);
}

// g_stamp != NULL (first-touch producers, clmgs_preprocess_bwd with sh_stamp): row r carries a gradient of
// THIS step only if g_stamp[r] == cur_step; other rows hold consumed leftovers, are not read (their
// gradient is zero) and nothing is cleared.
// RANGE (camera-DP, the small attributes computed by the owner of a row range): only rows row_begin <= r < row_end
// are stepped; the blocks that straddle the range borders pass the other rows through unchanged (same values written
// back, mirror row rewritten from the tensors).
template <bool RANGE>
__global__ void __launch_bounds__(SA_ROWS)
adam_small_packed_kernel(int64_t n, int64_t row_begin, int64_t row_end, SmallAdam t, float4* __restrict__ packed_p,
                         float4* __restrict__ packed_g, float beta1, float beta2, float ob1,
                         float ob2, float eps, float inv_bc1, float inv_sqrt_bc2, float grad_scale,
                         const int32_t* __restrict__ g_stamp, int cur_step) {
  __shared__ __attribute__((aligned(16))) float sg[SA_ROWS * 12];
  __shared__ __attribute__((aligned(16))) float sp[SA_ROWS * 12];
  const int tid = threadIdx.x;
  const int64_t n_blocks = RANGE ? (row_end + SA_ROWS - 1) / SA_ROWS : (n + SA_ROWS - 1) / SA_ROWS;
  for (int64_t blk = (RANGE ? row_begin / SA_ROWS : 0) + blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const int64_t row0 = blk * SA_ROWS;
    const int rows = (int)min((int64_t)SA_ROWS, n - row0);
    __syncthreads();
    if (g_stamp) {
      __shared__ uint8_t has_g[SA_ROWS];
      if (tid < rows) has_g[tid] = (uint8_t)(g_stamp[row0 + tid] == cur_step);
      __syncthreads();
      for (int i = tid; i < rows * 3; i += SA_ROWS)
        reinterpret_cast<float4*>(sg)[i] = has_g[i / 3] ? packed_g[row0 * 3 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (int i = tid; i < rows * 3; i += SA_ROWS)
        reinterpret_cast<float4*>(sg)[i] = packed_g[row0 * 3 + i];
    }
    if (tid < rows) sp[tid * 12 + 11] = 0.f;
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const int w = ti == 0 ? 3 : (ti == 1 ? 1 : (ti == 2 ? 3 : 4));
      const int co = ti == 0 ? 0 : (ti == 1 ? 3 : (ti == 2 ? 4 : 7));
      const int n_el = rows * w;
      float* P = t.p[ti] + row0 * w;
      float* M = t.m[ti] + row0 * w;
      float* V = t.v[ti] + row0 * w;
      const float lr_bc = t.lr[ti] * inv_bc1;
      for (int i = tid * 4; i < n_el; i += SA_ROWS * 4) {
        if (i + 3 < n_el) {
          float4 p4 = *reinterpret_cast<float4*>(P + i), m4 = *reinterpret_cast<float4*>(M + i),
                 v4 = *reinterpret_cast<float4*>(V + i);
          float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w},
                vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int idx = i + k, row = idx / w, e = idx - row * w;
            // (RANGE: the same arithmetic, then a select -- a branch around it changes how the compiler contracts the
            // moment updates, and the two forms of the kernel must agree bit for bit)
            float m2 = mm[k], v2 = vv[k];
            const float pn = adam_elem(m2, v2, pp[k], sg[row * 12 + co + e] * grad_scale, lr_bc, beta1, beta2,
                                       ob1, ob2, eps, inv_sqrt_bc2);
            const bool act = !RANGE || (row0 + row >= row_begin && row0 + row < row_end);
            pp[k] = act ? pn : pp[k]; mm[k] = act ? m2 : mm[k]; vv[k] = act ? v2 : vv[k];
            sp[row * 12 + co + e] = pp[k];
          }
          *reinterpret_cast<float4*>(P + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
          *reinterpret_cast<float4*>(M + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
          *reinterpret_cast<float4*>(V + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
          for (int idx = i; idx < n_el; ++idx) {  // ragged tail of the last block
            const int row = idx / w, e = idx - row * w;
            float mm = M[idx], vv = V[idx];
            const float p0 = P[idx];
            const float pn = adam_elem(mm, vv, p0, sg[row * 12 + co + e] * grad_scale, lr_bc, beta1, beta2, ob1, ob2,
                                       eps, inv_sqrt_bc2);
            const bool act = !RANGE || (row0 + row >= row_begin && row0 + row < row_end);
            if (act) { P[idx] = pn; M[idx] = mm; V[idx] = vv; }
            sp[row * 12 + co + e] = act ? pn : p0;
          }
        }
      }
    }
    __syncthreads();
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < rows * 3; i += SA_ROWS) {
      packed_p[row0 * 3 + i] = reinterpret_cast<const float4*>(sp)[i];
      const float4 gq = reinterpret_cast<const float4*>(sg)[i];  // rows the batch did not touch are
      if (!g_stamp && (gq.x != 0.f || gq.y != 0.f || gq.z != 0.f || gq.w != 0.f))  // zero already
        packed_g[row0 * 3 + i] = z;
    }
  }
}

// DEFERRED dense Adam of the 11 GPU-resident attributes (single GPU, round 5).  The eager pass above streams p / m / v of
// all N rows every batch (312 B x N) although a batch gives a gradient to a third of them, because a row without a
// gradient still moves (its first moment decays); and it cannot simply be postponed, because the next batch's visibility
// pass reads every row's position.  What CAN be postponed is the step of rows that provably stay invisible: rows are kept
// in Z-order, a block of 256 consecutive rows is a patch of ground, and Adam moves no element by more than
// 7.28 lr per step whatever its gradients (gaussian_model.py small_after_step).  Per block: `blk_last` = the optimizer
// step its 256 rows are current as of.  At the HEAD of a batch every block that is k = to_step - blk_last steps behind
// runs the cull's own conservative test (vis_candidate) on its STALE values with the margins k steps allow; only if some
// row may be visible in some camera of the batch -- or k has reached SD_KMAX, or the caller flushes -- are the block's k
// steps replayed: step by step in registers, the row's waiting gradient line (stamp g_stamp[row] == s) entering at ITS
// step, zero gradients at the others, every step with the constants (learning rates, bias corrections) it had -- the
// operations of k eager passes in the same order, element for element (adam_elem with g = 0 is what the eager pass
// computes for a row without a gradient).  The exact visibility pass that follows sees current values in every block
// that can matter: a row of a skipped block fails the exact cull with its stale values (the candidate test contains it)
// and with its true ones (margins), so the filters are the eager run's, bit for bit.
constexpr int SD_KMAX = 16;
struct SmallDeferred {
  // index j = to_step - s of optimizer step s (0 = the newest)
  float lr[SD_KMAX][4];
  float inv_bc1[SD_KMAX], inv_sqrt_bc2[SD_KMAX];
  // index k = number of steps a block is behind: how far a stale mean may be off / a stale largest scale may have grown
  float pos_margin[SD_KMAX + 1], scale_gain[SD_KMAX + 1];
  int to_step;
};

__global__ void __launch_bounds__(SA_ROWS)
adam_small_deferred_kernel(int64_t n, SmallAdam t, float4* __restrict__ packed_p, const float4* __restrict__ packed_g,
                           const int32_t* __restrict__ g_stamp, int32_t* __restrict__ blk_last, SmallDeferred d,
                           float beta1, float beta2, float ob1, float ob2, float eps, float grad_scale, int C,
                           const float* __restrict__ viewmats, const float* __restrict__ Ks, float W, float H, float eps2d,
                           float near_plane, float far_plane, int flush_all, uint8_t* __restrict__ blk_flag, int n_hist,
                           uint32_t* __restrict__ dev_err) {
  __shared__ __attribute__((aligned(16))) float sg[SA_ROWS * 12];
  __shared__ __attribute__((aligned(16))) float sp[SA_ROWS * 12];
  __shared__ int gstep_s[SA_ROWS];
  __shared__ float cam_s[VB_MAX_CAMS][VB_CAM_F];
  const int tid = threadIdx.x;
  if (!flush_all) {
    for (int c = tid; c < C; c += SA_ROWS) vis_store_cam(cam_s[c], viewmats + 16 * c, Ks + 9 * c, W, H);
  }
  const float near_m = near_plane - (fabsf(near_plane) * 1e-5f + 1e-6f);
  const float far_m = far_plane + fabsf(far_plane) * 1e-5f;
  const int64_t n_blocks = (n + SA_ROWS - 1) / SA_ROWS;
  for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const int last = blk_last[blk];
    int k = d.to_step - last;
    if (k <= 0 && (flush_all || !blk_flag)) continue;  // (block-uniform)
    if (k > n_hist) {
      // further behind than the recorded history reaches (the caller keeps kmax steps and this kernel forces a block at
      // k == kmax, so this cannot happen while that invariant holds): replaying with another step's constants would be
      // silently wrong -- raise bit 2 of the device error word (clmgs_device_errors) and replay what IS recorded
      if (tid == 0 && dev_err) atomicOr(dev_err, 4u);
      k = n_hist;
    }
    const int64_t row0 = blk * SA_ROWS;
    const int rows = (int)min((int64_t)SA_ROWS, n - row0);
    __syncthreads();  // cam_s written / the previous block's LDS consumed
    if (!flush_all) {
      // the candidate test: may a row of this block pass the cull of one of the batch's cameras, its values being stale by
      // k steps?  Also handed to the exact visibility pass that follows (blk_flag: it skips the blocks that cannot)
      const int kk = min(max(k, 0), SD_KMAX);
      bool cand = false;
      if (tid < rows) {
        const float* x = t.p[0] + 3 * (row0 + tid);
        const float* ls = t.p[2] + 3 * (row0 + tid);
        const float m[3] = {x[0], x[1], x[2]};
        const float sgn = __expf(fmaxf(ls[0], fmaxf(ls[1], ls[2]))) * d.scale_gain[kk];
        const float smax2g = sgn * sgn;
        cand = !(smax2g < 1e30f);  // NaN / overflowing scales: current values decide
        for (int c = 0; c < C && !cand; ++c)
          cand = vis_candidate(lds_cam(cam_s[c]), cam_s[c][16], m, smax2g, d.pos_margin[kk], W, H, eps2d, near_m, far_m);
      }
      const bool any = __syncthreads_or(cand);
      if (blk_flag && tid == 0) blk_flag[blk] = any ? 1 : 0;
      if (k <= 0 || (!any && k < SD_KMAX)) continue;
    }
    // ---- the block's k waiting steps.  Gradient lines: a row's line belongs to step g_stamp[row]; it is waiting iff
    // last < stamp <= to_step (consumed lines keep their old stamp: first-touch producers never clear)
    if (tid < rows) {
      const int gs = g_stamp[row0 + tid];
      gstep_s[tid] = (gs > last && gs <= d.to_step) ? gs : -1;
    }
    __syncthreads();
    for (int i = tid; i < rows * 3; i += SA_ROWS)
      reinterpret_cast<float4*>(sg)[i] = gstep_s[i / 3] >= 0 ? packed_g[row0 * 3 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < rows) sp[tid * 12 + 11] = 0.f;
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const int w = ti == 0 ? 3 : (ti == 1 ? 1 : (ti == 2 ? 3 : 4));
      const int co = ti == 0 ? 0 : (ti == 1 ? 3 : (ti == 2 ? 4 : 7));
      const int n_el = rows * w;
      float* P = t.p[ti] + row0 * w;
      float* M = t.m[ti] + row0 * w;
      float* V = t.v[ti] + row0 * w;
      for (int i = tid * 4; i < n_el; i += SA_ROWS * 4) {
        if (i + 3 < n_el) {
          float4 p4 = *reinterpret_cast<float4*>(P + i), m4 = *reinterpret_cast<float4*>(M + i),
                 v4 = *reinterpret_cast<float4*>(V + i);
          float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w},
                vv[4] = {v4.x, v4.y, v4.z, v4.w};
          float gg[4];
          int gst[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int idx = i + q, row = idx / w, e = idx - row * w;
            gg[q] = sg[row * 12 + co + e] * grad_scale;
            gst[q] = gstep_s[row];
          }
          for (int s_ = last + 1; s_ <= d.to_step; ++s_) {
            const int j = d.to_step - s_;
            const float lr_bc = d.lr[j][ti] * d.inv_bc1[j];
            const float isb2 = d.inv_sqrt_bc2[j];
#pragma unroll
            for (int q = 0; q < 4; ++q)
              pp[q] = adam_elem(mm[q], vv[q], pp[q], gst[q] == s_ ? gg[q] : 0.f, lr_bc, beta1, beta2, ob1, ob2, eps, isb2);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int idx = i + q, row = idx / w, e = idx - row * w;
            sp[row * 12 + co + e] = pp[q];
          }
          *reinterpret_cast<float4*>(P + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
          *reinterpret_cast<float4*>(M + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
          *reinterpret_cast<float4*>(V + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
          for (int idx = i; idx < n_el; ++idx) {  // ragged tail of the last block
            const int row = idx / w, e = idx - row * w;
            float mm = M[idx], vv = V[idx], pq = P[idx];
            const float g1 = sg[row * 12 + co + e] * grad_scale;
            for (int s_ = last + 1; s_ <= d.to_step; ++s_) {
              const int j = d.to_step - s_;
              pq = adam_elem(mm, vv, pq, gstep_s[row] == s_ ? g1 : 0.f, d.lr[j][ti] * d.inv_bc1[j], beta1, beta2, ob1, ob2,
                             eps, d.inv_sqrt_bc2[j]);
            }
            P[idx] = pq; M[idx] = mm; V[idx] = vv;
            sp[row * 12 + co + e] = pq;
          }
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < rows * 3; i += SA_ROWS) packed_p[row0 * 3 + i] = reinterpret_cast<const float4*>(sp)[i];
    if (tid == 0) blk_last[blk] = d.to_step;
  }
}

// [N,12] mirror from the four parameter tensors
__global__ void __launch_bounds__(256)
pack_small_kernel(int64_t n, const float* __restrict__ xyz, const float* __restrict__ opa,
                  const float* __restrict__ sca, const float* __restrict__ rot,
                  float4* __restrict__ packed_p) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * blockDim.x) {
    const float4 q = *reinterpret_cast<const float4*>(rot + 4 * r);
    packed_p[3 * r] = make_float4(xyz[3 * r], xyz[3 * r + 1], xyz[3 * r + 2], opa[r]);
    packed_p[3 * r + 1] = make_float4(sca[3 * r], sca[3 * r + 1], sca[3 * r + 2], q.x);
    packed_p[3 * r + 2] = make_float4(q.y, q.z, q.w, 0.f);
  }
}

// Camera-DP step S (dp.small_fetch): packed [n_rows,12] lines received from the owners -> the four parameter tensors
// and the packed mirror at `rows` (unique ids)
__global__ void __launch_bounds__(256)
small_rows_scatter_kernel(int64_t n_rows, const int64_t* __restrict__ rows, const float4* __restrict__ src,
                          float* __restrict__ xyz, float* __restrict__ opa, float* __restrict__ sca,
                          float* __restrict__ rot, float4* __restrict__ packed_p) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows[i];
    const float4 a = src[3 * i], b = src[3 * i + 1], c = src[3 * i + 2];
    xyz[3 * r] = a.x; xyz[3 * r + 1] = a.y; xyz[3 * r + 2] = a.z;
    opa[r] = a.w;
    sca[3 * r] = b.x; sca[3 * r + 1] = b.y; sca[3 * r + 2] = b.z;
    *reinterpret_cast<float4*>(rot + 4 * r) = make_float4(b.w, c.x, c.y, c.z);
    packed_p[3 * r] = a; packed_p[3 * r + 1] = b; packed_p[3 * r + 2] = make_float4(c.x, c.y, c.z, 0.f);
  }
}

// ------------------------------------------------------- densification stats
__global__ void __launch_bounds__(256)
densify_stats_kernel(int64_t n, const int64_t* __restrict__ filter,
                     const float* __restrict__ v_means2d, const int32_t* __restrict__ radii,
                     int only_visible, float half_w, float half_h, float* __restrict__ max_radii2D,
                     float* __restrict__ accum, float* __restrict__ denom) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int r = radii[i];
    if (only_visible && r <= 0) continue;
    const int64_t g = filter ? filter[i] : i;
    const float gx = v_means2d[2 * i] * half_w, gy = v_means2d[2 * i + 1] * half_h;
    max_radii2D[g] = fmaxf(max_radii2D[g], (float)r);
    accum[g] += sqrtf(gx * gx + gy * gy);
    denom[g] += 1.f;
  }
}

}  // namespace clmgs

using namespace clmgs;

extern "C" int clmgs_rows_gather(void* stream, float* dst, const float* src, const void* dst_idx,
                                 const void* src_idx, int idx_is_64, int64_t n_rows, int cols,
                                 int grid_blocks) {
  return rows_move<false>(stream, dst, src, dst_idx, src_idx, idx_is_64, n_rows, cols, grid_blocks);
}

extern "C" int clmgs_publish_pack(void* stream, float* msg, const float* table, const int64_t* own_rows,
                                  const int32_t* stamp, int step, int64_t lo, int64_t n_rows, int64_t chunk,
                                  int cols) {
  CLMGS_CHECK_ARG(n_rows >= 0 && chunk >= n_rows && cols > 0 && cols % 4 == 0);
  if (n_rows == 0) return 0;
  CLMGS_CHECK_ARG(msg && table && own_rows && (((uintptr_t)msg | (uintptr_t)table) & 15) == 0);
  const int64_t total = n_rows * (cols / 4);
  hipLaunchKernelGGL(publish_pack_kernel, dim3(min(ceil_div(total, 256), 256 * 8)), dim3(256), 0, (hipStream_t)stream,
                     msg, table, own_rows, stamp, step, lo, n_rows, chunk, cols / 4);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_rows_scatter_add(void* stream, float* dst, const float* src,
                                      const void* dst_idx, const void* src_idx, int idx_is_64,
                                      int64_t n_rows, int cols, int grid_blocks) {
  return rows_move<true>(stream, dst, src, dst_idx, src_idx, idx_is_64, n_rows, cols, grid_blocks);
}

#define DISPATCH_ELEM(bytes, CALL)                    \
  switch (bytes) {                                    \
    case 1: { typedef int8_t T; CALL; } break;        \
    case 2: { typedef int16_t T; CALL; } break;       \
    case 4: { typedef int32_t T; CALL; } break;       \
    case 8: { typedef int64_t T; CALL; } break;       \
    default: clmgs::set_error("bitmap elem_bytes must be 1,2,4,8"); return CLMGS_EINVAL; \
  }

extern "C" int clmgs_scatter_to_bit(void* stream, void* bitmap, int elem_bytes,
                                    const int64_t* filter, int64_t n, int bit) {
  CLMGS_CHECK_ARG(n >= 0 && bit >= 0 && bit < elem_bytes * 8);
  if (n == 0) return 0;
  CLMGS_CHECK_ARG(bitmap && filter);
  const int grid = min(ceil_div(n, 256), 256 * 8);
  DISPATCH_ELEM(elem_bytes, hipLaunchKernelGGL(scatter_to_bit_kernel<T>, dim3(grid), dim3(256), 0,
                                               (hipStream_t)stream, (T*)bitmap, filter, n, bit));
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t clmgs_host_groups_temp_bytes(int64_t n) {
  if (n <= 0) return 256;
  return 4 * align_up((size_t)n * 4, 256) + 2 * align_up((size_t)n * 4, 256) + 2 * radix_table_bytes(n) + 256;
}

extern "C" int clmgs_host_groups(void* stream, int64_t n, const int64_t* touched, const void* bitmap, int elem_bytes,
                                 int bsz, const uint8_t* staged, int slot0, int32_t* late_sorted,
                                 int32_t* rows_by_last, int32_t* slot_of, int64_t* counts, void* temp,
                                 size_t temp_bytes) {
  CLMGS_CHECK_ARG(n >= 0 && bsz >= 1 && bsz <= 64 && bsz <= elem_bytes * 8);
  CLMGS_CHECK_ARG(counts);
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    CLMGS_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * (2 * bsz + 1), s));
    return 0;
  }
  CLMGS_CHECK_ARG(n < ((int64_t)1 << 31) && touched && bitmap && late_sorted && rows_by_last && slot_of && temp);
  CLMGS_CHECK_ARG(temp_bytes >= clmgs_host_groups_temp_bytes(n));
  char* base = (char*)temp;
  const size_t a4 = align_up((size_t)n * 4, 256);
  uint32_t* kf = (uint32_t*)base; base += a4;
  uint32_t* kl = (uint32_t*)base; base += a4;
  uint32_t* kb = (uint32_t*)base; base += a4;   // second key buffer of the sorts
  int32_t* rows32 = (int32_t*)base; base += a4;
  int32_t* vb = (int32_t*)base; base += a4;
  int32_t* vc = (int32_t*)base; base += a4;     // (unused by one-pass sorts; kept for the sorter's interface)
  uint32_t* table1 = (uint32_t*)base; base += radix_table_bytes(n);
  uint32_t* table2 = (uint32_t*)base;
  const int grid = min(ceil_div(n, 256), 256 * 8);
  DISPATCH_ELEM(elem_bytes, hipLaunchKernelGGL(host_group_keys_kernel<T>, dim3(grid), dim3(256), 0, s, n, touched,
                                               (const T*)bitmap, bsz, staged, kf, kl, rows32));
  CLMGS_LAUNCH_CHECK();
  uint32_t* sorted = nullptr;
  int rc = radix_sort_pairs<uint32_t, int32_t>(s, n, kf, kb, rows32, vb, late_sorted, 0, 8, table1, &sorted);
  if (rc) return rc;
  rc = radix_sort_pairs<uint32_t, int32_t>(s, n, kl, kb, rows32, vc, rows_by_last, 0, 8, table2, &sorted);
  if (rc) return rc;
  const int nb = (int)((n + RS_MIN_CHUNK - 1) / RS_MIN_CHUNK);
  hipLaunchKernelGGL(host_group_finish_kernel, dim3(grid), dim3(256), 0, s, n, bsz, table1 + (size_t)nb * 256,
                     table2 + (size_t)nb * 256, late_sorted, slot0, slot_of, counts);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

// ---- storage order of the rows: Z-order (Morton) curve of (x, y), 16 bits per axis (clm_gs_amd/utils.py morton_order)
// code = spread(qx) | spread(qy) << 1,  q = rint((double(p) - lo) / max(hi - lo, 1e-30) * 65535)  -- the IEEE double
// operations of the torch form, one pass instead of ~25 elementwise passes over [N] int64 / double temporaries -- then
// the stable LSD radix sort of radix.h on (code, row) pairs and the row ids widened to int64.
__device__ __forceinline__ uint32_t spread16(uint32_t v) {
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

__global__ void __launch_bounds__(256)
morton_keys_kernel(int64_t n, const float* __restrict__ xyz, const double* __restrict__ lohi,
                   uint32_t* __restrict__ keys, int32_t* __restrict__ vals) {
  const double lx = lohi[0], ly = lohi[1];
  const double dx = fmax(lohi[2] - lx, 1e-30), dy = fmax(lohi[3] - ly, 1e-30);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double qx = rint(((double)xyz[3 * i] - lx) / dx * 65535.0);
    const double qy = rint(((double)xyz[3 * i + 1] - ly) / dy * 65535.0);
    const uint32_t ix = (uint32_t)(long long)qx & 0xFFFFu, iy = (uint32_t)(long long)qy & 0xFFFFu;
    keys[i] = spread16(ix) | (spread16(iy) << 1);
    vals[i] = (int32_t)i;
  }
}

__global__ void __launch_bounds__(256)
widen_i32_kernel(int64_t n, const int32_t* __restrict__ src, int64_t* __restrict__ dst) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (int64_t)src[i];
}

extern "C" size_t clmgs_morton_order_temp_bytes(int64_t n) {
  if (n <= 0) return 256;
  return 5 * align_up((size_t)n * 4, 256) + radix_table_bytes(n) + 256;
}

extern "C" int clmgs_morton_order(void* stream, int64_t n, const float* xyz, const double* lo_hi, int64_t* order,
                                  void* temp, size_t temp_bytes) {
  CLMGS_CHECK_ARG(n >= 0);
  if (n == 0) return 0;
  CLMGS_CHECK_ARG(n < ((int64_t)1 << 31) && xyz && lo_hi && order && temp);
  CLMGS_CHECK_ARG(temp_bytes >= clmgs_morton_order_temp_bytes(n));
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)temp;
  const size_t a4 = align_up((size_t)n * 4, 256);
  uint32_t* ka = (uint32_t*)base; base += a4;
  uint32_t* kb = (uint32_t*)base; base += a4;
  int32_t* va = (int32_t*)base; base += a4;
  int32_t* vb = (int32_t*)base; base += a4;
  int32_t* vf = (int32_t*)base; base += a4;
  uint32_t* table = (uint32_t*)base;
  const int grid = (int)min(ceil_div(n, 256), (int64_t)256 * 16);
  hipLaunchKernelGGL(morton_keys_kernel, dim3(grid), dim3(256), 0, s, n, xyz, lo_hi, ka, va);
  CLMGS_LAUNCH_CHECK();
  uint32_t* sorted = nullptr;
  int rc = radix_sort_pairs<uint32_t, int32_t>(s, n, ka, kb, va, vb, vf, 0, 32, table, &sorted);
  if (rc) return rc;
  hipLaunchKernelGGL(widen_i32_kernel, dim3(grid), dim3(256), 0, s, n, vf, order);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_extract_ffs(void* stream, const void* bitmap, int elem_bytes, int64_t N,
                                 uint8_t* ffs) {
  CLMGS_CHECK_ARG(N >= 0);
  if (N == 0) return 0;
  CLMGS_CHECK_ARG(bitmap && ffs);
  const int grid = min(ceil_div(N, 256), 256 * 8);
  DISPATCH_ELEM(elem_bytes, hipLaunchKernelGGL(extract_ffs_kernel<T>, dim3(grid), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)bitmap, N, ffs));
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_pair_overlap_count(void* stream, const void* bitmap, int elem_bytes,
                                        int64_t N, int bsz, int32_t* cnt) {
  CLMGS_CHECK_ARG(N >= 0 && bsz >= 2 && bsz <= elem_bytes * 8 && cnt);
  if (N == 0) return 0;
  CLMGS_CHECK_ARG(bitmap);
  const int grid = min(ceil_div(N, 256), 256 * 4);
  DISPATCH_ELEM(elem_bytes, hipLaunchKernelGGL(pair_overlap_kernel<T>, dim3(grid), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)bitmap, N, bsz, cnt));
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_set_signal(void* stream, int32_t* signal_pinned, int idx, int32_t value) {
  CLMGS_CHECK_ARG(signal_pinned && idx >= 0);
  hipLaunchKernelGGL(set_signal_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, signal_pinned,
                     idx, value);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_adam_rows(void* stream, float* p, float* g, float* m, float* v,
                               const void* rows, int idx_is_64, const uint8_t* mask,
                               int64_t n_rows, int cols, const float* col_lr, double beta1,
                               double beta2, double eps, int step, int bias_correction,
                               float grad_scale, int zero_grad) {
  CLMGS_CHECK_ARG(n_rows >= 0 && cols > 0 && step >= 1);
  if (n_rows == 0) return 0;
  CLMGS_CHECK_ARG(p && m && v && col_lr);  // g may be NULL (= all-zero gradient)
  float inv_bc1 = 1.f, inv_sqrt_bc2 = 1.f;
  if (bias_correction) {
    inv_bc1 = (float)(1.0 / (1.0 - pow(beta1, (double)step)));
    inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step)));
  }
  // 1 - beta in double on the host: (1.f - 0.999f) alone is off by 1.3e-5 relative
  const float ob1 = (float)(1.0 - beta1), ob2 = (float)(1.0 - beta2);
  const bool v4 = (cols % 4 == 0) && (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g |
                                        (uintptr_t)col_lr) & 15) == 0;
  const int grid = (int)min(ceil_div(n_rows * (cols / (v4 ? 4 : 1)), 256), (int64_t)256 * 16);
#define CLMGS_ADAM_ROWS(I, VEC)                                                                    \
  hipLaunchKernelGGL((adam_rows_kernel<I, VEC>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, \
                     g, m, v, rows, mask, n_rows, cols, col_lr, (float)beta1, (float)beta2, ob1,   \
                     ob2, (float)eps, inv_bc1, inv_sqrt_bc2, grad_scale, zero_grad)
  if (idx_is_64) { if (v4) CLMGS_ADAM_ROWS(int64_t, 4); else CLMGS_ADAM_ROWS(int64_t, 1); }
  else { if (v4) CLMGS_ADAM_ROWS(int32_t, 4); else CLMGS_ADAM_ROWS(int32_t, 1); }
#undef CLMGS_ADAM_ROWS
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_adam_catch_up(void* stream, float* p, float* m, float* v,
                                   const int32_t* last_step, const void* rows, int idx_is_64,
                                   int64_t n_rows, int cols, const float* col_lr, double beta1,
                                   double beta2, double eps, int to_step, int bias_correction,
                                   int max_replay, float* g, const int32_t* g_step, float grad_scale,
                                   int keep_grad) {
  CLMGS_CHECK_ARG(n_rows >= 0 && cols > 0 && to_step >= 0 && max_replay >= 1);
  if (n_rows == 0) return 0;
  CLMGS_CHECK_ARG(p && m && v && last_step && col_lr && ((g != nullptr) == (g_step != nullptr)));
  const bool v4 = (cols % 4 == 0) &&
                  (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)col_lr | (uintptr_t)g) & 15) == 0;
  if (v4 && cols == 48 && n_rows < ((int64_t)1 << 27)) {  // (12 n < 2^32)
    int grid48 = (int)min(ceil_div(n_rows * 12, 256), (int64_t)4095);
    grid48 = (grid48 + 2) / 3 * 3;
#define CLMGS_CATCH_UP48(I)                                                                          \
  hipLaunchKernelGGL((adam_catch_up48_kernel<I>), dim3(grid48), dim3(256), 0, (hipStream_t)stream, p, m, v, \
                     last_step, rows, (unsigned)n_rows, col_lr, (float)beta1, (float)beta2, (float)eps, to_step, \
                     bias_correction, max_replay, g, g_step, grad_scale, (float)(1.0 - beta1),        \
                     (float)(1.0 - beta2), keep_grad)
    if (idx_is_64) CLMGS_CATCH_UP48(int64_t); else CLMGS_CATCH_UP48(int32_t);
#undef CLMGS_CATCH_UP48
    CLMGS_LAUNCH_CHECK();
    return 0;
  }
  const int grid = (int)min(ceil_div(n_rows * (cols / (v4 ? 4 : 1)), 256), (int64_t)256 * 16);
#define CLMGS_CATCH_UP(I, VEC)                                                                     \
  hipLaunchKernelGGL((adam_catch_up_kernel<I, VEC>), dim3(grid), dim3(256), 0, (hipStream_t)stream, \
                     p, m, v, last_step, rows, n_rows, cols, col_lr, (float)beta1, (float)beta2,   \
                     (float)eps, to_step, bias_correction, max_replay, g, g_step, grad_scale,     \
                     (float)(1.0 - beta1), (float)(1.0 - beta2), keep_grad)
  if (idx_is_64) { if (v4) CLMGS_CATCH_UP(int64_t, 4); else CLMGS_CATCH_UP(int64_t, 1); }
  else { if (v4) CLMGS_CATCH_UP(int32_t, 4); else CLMGS_CATCH_UP(int32_t, 1); }
#undef CLMGS_CATCH_UP
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_pack_small(void* stream, int64_t n, const float* xyz, const float* opacity,
                                const float* scaling, const float* rotation, void* packed_p) {
  CLMGS_CHECK_ARG(n >= 0);
  if (n == 0) return 0;
  CLMGS_CHECK_ARG(xyz && opacity && scaling && rotation && packed_p && (((uintptr_t)packed_p & 15) == 0));
  hipLaunchKernelGGL(pack_small_kernel, dim3(min(ceil_div(n, 256), 256 * 16)), dim3(256), 0,
                     (hipStream_t)stream, n, xyz, opacity, scaling, rotation, (float4*)packed_p);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_small_rows_scatter(void* stream, int64_t n_rows, const int64_t* rows, const void* lines,
                                        float* xyz, float* opacity, float* scaling, float* rotation,
                                        void* packed_p) {
  CLMGS_CHECK_ARG(n_rows >= 0);
  if (n_rows == 0) return 0;
  CLMGS_CHECK_ARG(rows && lines && xyz && opacity && scaling && rotation && packed_p &&
                  (((uintptr_t)packed_p | (uintptr_t)lines | (uintptr_t)rotation) & 15) == 0);
  hipLaunchKernelGGL(small_rows_scatter_kernel, dim3(min(ceil_div(n_rows, 256), 256 * 16)), dim3(256), 0,
                     (hipStream_t)stream, n_rows, rows, (const float4*)lines, xyz, opacity, scaling, rotation,
                     (float4*)packed_p);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_adam_small_packed_range(void* stream, int64_t n, int64_t row_begin, int64_t row_end,
                                             float* const* params, float* const* exp_avg,
                                             float* const* exp_avg_sq, const double* lr4, void* packed_p,
                                             void* packed_g, double beta1, double beta2, double eps, int step,
                                             int bias_correction, float grad_scale, const int32_t* g_stamp,
                                             int cur_step);

extern "C" int clmgs_adam_small_packed(void* stream, int64_t n, float* const* params,
                                       float* const* exp_avg, float* const* exp_avg_sq,
                                       const double* lr4, void* packed_p, void* packed_g,
                                       double beta1, double beta2, double eps, int step,
                                       int bias_correction, float grad_scale, const int32_t* g_stamp,
                                       int cur_step) {
  return clmgs_adam_small_packed_range(stream, n, 0, -1, params, exp_avg, exp_avg_sq, lr4, packed_p, packed_g, beta1,
                                       beta2, eps, step, bias_correction, grad_scale, g_stamp, cur_step);
}

// row_end < 0: all n rows (the kernel without the per-row range test); otherwise only rows row_begin <= r < row_end
extern "C" int clmgs_adam_small_packed_range(void* stream, int64_t n, int64_t row_begin, int64_t row_end,
                                             float* const* params, float* const* exp_avg,
                                             float* const* exp_avg_sq, const double* lr4, void* packed_p,
                                             void* packed_g, double beta1, double beta2, double eps, int step,
                                             int bias_correction, float grad_scale, const int32_t* g_stamp,
                                             int cur_step) {
  CLMGS_CHECK_ARG(n >= 0 && step >= 1);
  const bool ranged = row_end >= 0;
  if (ranged) CLMGS_CHECK_ARG(row_begin >= 0 && row_begin <= row_end && row_end <= n);
  if (n == 0 || (ranged && row_end == row_begin)) return 0;
  CLMGS_CHECK_ARG(params && exp_avg && exp_avg_sq && lr4 && packed_p && packed_g &&
                  (((uintptr_t)packed_p | (uintptr_t)packed_g) & 15) == 0);
  SmallAdam t;
  for (int i = 0; i < 4; ++i) {
    CLMGS_CHECK_ARG(params[i] && exp_avg[i] && exp_avg_sq[i]);
    t.p[i] = params[i]; t.m[i] = exp_avg[i]; t.v[i] = exp_avg_sq[i]; t.lr[i] = (float)lr4[i];
  }
  float inv_bc1 = 1.f, inv_sqrt_bc2 = 1.f;
  if (bias_correction) {
    inv_bc1 = (float)(1.0 / (1.0 - pow(beta1, (double)step)));
    inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step)));
  }
  const float ob1 = (float)(1.0 - beta1), ob2 = (float)(1.0 - beta2);
  if (ranged) {
    const int64_t blocks = (row_end + SA_ROWS - 1) / SA_ROWS - row_begin / SA_ROWS;
    hipLaunchKernelGGL(adam_small_packed_kernel<true>, dim3(min(blocks, (int64_t)256 * 16)), dim3(SA_ROWS), 0,
                       (hipStream_t)stream, n, row_begin, row_end, t, (float4*)packed_p, (float4*)packed_g,
                       (float)beta1, (float)beta2, ob1, ob2, (float)eps, inv_bc1, inv_sqrt_bc2, grad_scale, g_stamp,
                       cur_step);
  } else {
    hipLaunchKernelGGL(adam_small_packed_kernel<false>, dim3(min(ceil_div(n, SA_ROWS), 256 * 16)), dim3(SA_ROWS), 0,
                       (hipStream_t)stream, n, (int64_t)0, n, t, (float4*)packed_p, (float4*)packed_g, (float)beta1,
                       (float)beta2, ob1, ob2, (float)eps, inv_bc1, inv_sqrt_bc2, grad_scale, g_stamp, cur_step);
  }
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_small_deferred_kmax(void) { return SD_KMAX; }

// See adam_small_deferred_kernel.  n_hist <= clmgs_small_deferred_kmax() optimizer steps of history, newest first
// (entry j describes step to_step - j): lr4_hist[j][4], step_index[j] (the Adam step count of that step, for the bias
// corrections), and how far a block that is k steps behind may be off: pos_margin[k], scale_gain[k], k = 0 .. n_hist.
// blk_last[ceil(n / 256)]: the step every block of 256 rows is current as of (updated here).  flush_all != 0: every
// block is brought to to_step (no camera needed).  Blocks more than n_hist steps behind are an error of the caller
// (it must flush at least every n_hist steps): checked on the host side by construction (a block is never left
// behind for more than kmax steps: the kernel forces it at k == kmax, and the caller trims its history to kmax entries);
// the kernel itself raises bit 2 of the device error word (clmgs_device_errors) if it ever meets one.
extern "C" int clmgs_adam_small_deferred(void* stream, int64_t n, float* const* params, float* const* exp_avg,
                                         float* const* exp_avg_sq, void* packed_p, const void* packed_g,
                                         const int32_t* g_stamp, int32_t* blk_last, int to_step, int n_hist,
                                         const double* lr4_hist, const int32_t* step_index, const float* pos_margin,
                                         const float* scale_gain, double beta1, double beta2, double eps,
                                         float grad_scale, int C, const float* viewmats, const float* Ks, int width,
                                         int height, float eps2d, float near_plane, float far_plane, int flush_all,
                                         uint8_t* blk_flag) {
  CLMGS_CHECK_ARG(n >= 0 && to_step >= 0 && n_hist >= 0 && n_hist <= SD_KMAX);
  CLMGS_CHECK_ARG(!blk_flag || !flush_all);
  if (n == 0) return 0;
  if (n_hist == 0) {  // nothing recorded yet: nothing waits, and nothing is known about visibility
    if (blk_flag) CLMGS_HIP(hipMemsetAsync(blk_flag, 1, (size_t)((n + SA_ROWS - 1) / SA_ROWS), (hipStream_t)stream));
    return 0;
  }
  CLMGS_CHECK_ARG(params && exp_avg && exp_avg_sq && packed_p && packed_g && g_stamp && blk_last && lr4_hist &&
                  step_index && pos_margin && scale_gain && (((uintptr_t)packed_p | (uintptr_t)packed_g) & 15) == 0);
  CLMGS_CHECK_ARG(flush_all || (C >= 1 && C <= VB_MAX_CAMS && viewmats && Ks && width > 0 && height > 0));
  SmallAdam t;
  for (int i = 0; i < 4; ++i) {
    CLMGS_CHECK_ARG(params[i] && exp_avg[i] && exp_avg_sq[i]);
    t.p[i] = params[i]; t.m[i] = exp_avg[i]; t.v[i] = exp_avg_sq[i]; t.lr[i] = 0.f;
  }
  SmallDeferred d;
  for (int j = 0; j < SD_KMAX; ++j) {
    const int jj = j < n_hist ? j : n_hist - 1;
    for (int i = 0; i < 4; ++i) d.lr[j][i] = (float)lr4_hist[4 * jj + i];
    CLMGS_CHECK_ARG(step_index[jj] >= 1);
    d.inv_bc1[j] = (float)(1.0 / (1.0 - pow(beta1, (double)step_index[jj])));       // as clmgs_adam_small_packed_range
    d.inv_sqrt_bc2[j] = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step_index[jj])));
  }
  for (int k = 0; k <= SD_KMAX; ++k) {
    const int kk = k <= n_hist ? k : n_hist;
    d.pos_margin[k] = pos_margin[kk];
    d.scale_gain[k] = scale_gain[kk];
  }
  d.to_step = to_step;
  const float ob1 = (float)(1.0 - beta1), ob2 = (float)(1.0 - beta2);
  hipLaunchKernelGGL(adam_small_deferred_kernel, dim3(min(ceil_div(n, SA_ROWS), 256 * 16)), dim3(SA_ROWS), 0,
                     (hipStream_t)stream, n, t, (float4*)packed_p, (const float4*)packed_g, g_stamp, blk_last, d,
                     (float)beta1, (float)beta2, ob1, ob2, (float)eps, grad_scale, C, viewmats, Ks, (float)width,
                     (float)height, eps2d, near_plane, far_plane, flush_all, blk_flag, n_hist, device_error_word());
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_densify_stats(void* stream, int64_t n, const int64_t* filter,
                                   const float* v_means2d, const int32_t* radii,
                                   int only_visible, float half_w, float half_h,
                                   float* max_radii2D, float* xyz_gradient_accum,
                                   float* denom) {
  CLMGS_CHECK_ARG(n >= 0);
  if (n == 0) return 0;
  CLMGS_CHECK_ARG(v_means2d && radii && max_radii2D && xyz_gradient_accum && denom);
  const int grid = min(ceil_div(n, 256), 256 * 8);
  hipLaunchKernelGGL(densify_stats_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, filter,
                     v_means2d, radii, only_visible, half_w, half_h, max_radii2D,
                     xyz_gradient_accum, denom);
  CLMGS_LAUNCH_CHECK();
  return 0;
}
