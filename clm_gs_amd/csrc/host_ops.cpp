// Host-side pieces of libclmgs_hip.so: error slot, pinned allocator, the row-group
// host Adam that stands in for cpu_adam.FusedCPUAdam.batched_sparse_step
// (strategies/clm_offload/engine.py:316-328, optimizer.py:130-144) and the camera
// tour heuristic that stands in for fast_tsp.find_tour (clm_offload/engine.py:179).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/clmgs.h"

namespace clmgs {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace clmgs

extern "C" int clmgs_version(void) { return 100; }
extern "C" const char* clmgs_last_error(void) { return clmgs::g_err; }

extern "C" void* clmgs_pinned_alloc(size_t bytes) {
  void* p = nullptr;
  // mapped + portable: kernels may dereference it directly (zero-copy rows)
  hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocPortable);
  if (e != hipSuccess) {
    clmgs::set_error("hipHostMalloc(%zu) -> %s", bytes, hipGetErrorString(e));
    return nullptr;
  }
  return p;
}

extern "C" int clmgs_pinned_free(void* p) {
  if (!p) return 0;
  hipError_t e = hipHostFree(p);
  if (e != hipSuccess) {
    clmgs::set_error("hipHostFree -> %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// One row group of the overlapped host optimizer.  Waits (spinning, yielding) for the
// GPU's set_signal write, then updates the listed rows with n_threads std::threads.
extern "C" int clmgs_host_adam_rows(float* p, float* g, float* m, float* v, const int32_t* rows,
                                    int64_t n_rows, int cols, const float* col_lr, double beta1d,
                                    double beta2d, double epsd, int step, int bias_correction,
                                    float grad_scale, int zero_grad,
                                    const volatile int32_t* signal, int n_threads) {
  if (n_rows < 0 || cols <= 0 || step < 1 || (n_rows > 0 && !(p && g && m && v && col_lr))) {
    clmgs::set_error("clmgs_host_adam_rows: invalid argument");
    return CLMGS_EINVAL;
  }
  if (signal) {
    while (__atomic_load_n((const int32_t*)signal, __ATOMIC_ACQUIRE) == 0) std::this_thread::yield();
  }
  if (n_rows == 0) return 0;
  const float beta1 = (float)beta1d, beta2 = (float)beta2d, eps = (float)epsd;
  float inv_bc1 = 1.f, inv_sqrt_bc2 = 1.f;
  if (bias_correction) {
    inv_bc1 = (float)(1.0 / (1.0 - pow(beta1d, (double)step)));
    inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(beta2d, (double)step)));
  }
  std::vector<float> step_lr(cols);
  for (int k = 0; k < cols; ++k) step_lr[k] = col_lr[k] * inv_bc1;
  const float ob1 = (float)(1.0 - beta1d), ob2 = (float)(1.0 - beta2d);
  auto work = [&](int64_t lo, int64_t hi) {
    for (int64_t r = lo; r < hi; ++r) {
      const int64_t row = rows ? (int64_t)rows[r] : r;
      float* pp = p + row * cols;
      float* gp = g + row * cols;
      float* mp = m + row * cols;
      float* vp = v + row * cols;
      for (int k = 0; k < cols; ++k) {
        const float gg = gp[k] * grad_scale;
        const float mm = beta1 * mp[k] + ob1 * gg;
        const float vv = beta2 * vp[k] + ob2 * gg * gg;
        mp[k] = mm; vp[k] = vv;
        pp[k] -= step_lr[k] * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
        if (zero_grad) gp[k] = 0.f;
      }
    }
  };
  int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, n_rows / 1024));
  if (nt == 1) { work(0, n_rows); return 0; }
  std::vector<std::thread> th;
  const int64_t per = (n_rows + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) {
    const int64_t lo = t * per, hi = std::min<int64_t>(n_rows, lo + per);
    if (lo < hi) th.emplace_back(work, lo, hi);
  }
  for (auto& t : th) t.join();
  return 0;
}

// Open tour: greedy nearest neighbour from every start, keep the best, then 2-opt
// (segment reversal) until no move improves.  n <= 64 cameras: exact cost is cheap.
extern "C" int clmgs_tsp_tour(int n, const int64_t* dist, int32_t* tour) {
  if (n <= 0 || !dist || !tour) { clmgs::set_error("clmgs_tsp_tour: invalid argument"); return CLMGS_EINVAL; }
  auto D = [&](int a, int b) { return dist[(int64_t)a * n + b]; };
  auto cost = [&](const std::vector<int>& t) { int64_t c = 0; for (int i = 0; i + 1 < n; ++i) c += D(t[i], t[i + 1]); return c; };
  std::vector<int> best;
  int64_t best_c = 0;
  for (int s = 0; s < n; ++s) {
    std::vector<int> t; std::vector<char> used(n, 0);
    t.push_back(s); used[s] = 1;
    for (int i = 1; i < n; ++i) {
      int cur = t.back(), nx = -1;
      for (int c = 0; c < n; ++c) if (!used[c] && (nx < 0 || D(cur, c) < D(cur, nx))) nx = c;
      t.push_back(nx); used[nx] = 1;
    }
    const int64_t c = cost(t);
    if (best.empty() || c < best_c) { best = t; best_c = c; }
  }
  bool improved = true;
  while (improved) {
    improved = false;
    for (int i = 0; i < n - 1 && !improved; ++i)
      for (int j = i + 1; j < n && !improved; ++j) {
        std::vector<int> t = best;
        std::reverse(t.begin() + i, t.begin() + j + 1);
        const int64_t c = cost(t);
        if (c < best_c) { best = t; best_c = c; improved = true; }
      }
  }
  for (int i = 0; i < n; ++i) tour[i] = best[i];
  return 0;
}
