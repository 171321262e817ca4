// Host-side pieces of libclmgs_hip.so: error slot, pinned allocator, the row-group
// host Adam that stands in for cpu_adam.FusedCPUAdam.batched_sparse_step
// (strategies/clm_offload/engine.py:316-328, optimizer.py:130-144) and the camera
// tour heuristic that stands in for fast_tsp.find_tour (clm_offload/engine.py:179).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <map>

#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
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

extern "C" int clmgs_version(void) { return 200; }
extern "C" int clmgs_host_usable_cpus(void);
extern "C" const char* clmgs_last_error(void) { return clmgs::g_err; }

// Number of NUMA nodes with memory (directories /sys/devices/system/node/nodeK).
static int numa_nodes() {
  int n = 0;
  char path[64];
  for (; n < 64; ++n) {
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d", n);
    if (access(path, F_OK) != 0) break;
  }
  return n > 0 ? n : 1;
}

// Large tables (>= 64 MB): anonymous mapping advised to transparent huge pages, touched (so the pages
// exist, interleaved over the NUMA nodes) and then registered with the runtime (pinned + device-mapped).
// The host row optimizer walks these tables with gappy 192 B accesses: with 4 KB pages every row costs
// TLB misses in four tables; 2 MB pages remove most of them.  CLMGS_PINNED_NO_THP=1 restores plain
// hipHostMalloc.  Returns nullptr (caller falls back) if any step is refused.
static std::map<void*, size_t> g_registered;
static std::mutex g_registered_mu;

static void* thp_alloc(size_t bytes, int nodes) {
  const size_t huge = (size_t)2 << 20;
  const size_t len = (bytes + huge - 1) / huge * huge;
  void* p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) return nullptr;
  madvise(p, len, MADV_HUGEPAGE);
  if (nodes > 1) {
    unsigned long mask = nodes >= 64 ? ~0ul : ((1ul << nodes) - 1ul);
    syscall(SYS_mbind, p, len, 3 /* MPOL_INTERLEAVE */, &mask, (unsigned long)(sizeof(mask) * 8), 0u);
  }
  // fault the pages in now, from several threads (first touch of 20+ GB by one thread takes seconds)
  const int nt = 8;
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([=] {
      char* c = (char*)p;
      for (size_t o = (size_t)t * 4096; o < len; o += (size_t)nt * 4096) c[o] = 0;
    });
  for (auto& t : th) t.join();
  if (hipHostRegister(p, len, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    munmap(p, len);
    return nullptr;
  }
  // kernels are handed the HOST pointer (zero-copy gradient scatter into the row tables): only valid when the
  // device address of the registered range IS the host address (HMM / unified addressing).  Otherwise fall
  // back to hipHostMalloc (the caller does), whose pointer is device-accessible by contract.
  void* dptr = nullptr;
  if (hipHostGetDevicePointer(&dptr, p, 0) != hipSuccess || dptr != p) {
    (void)hipGetLastError();
    (void)hipHostUnregister(p);
    munmap(p, len);
    return nullptr;
  }
  std::lock_guard<std::mutex> l(g_registered_mu);
  g_registered[p] = len;
  return p;
}

extern "C" void* clmgs_pinned_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes >= ((size_t)64 << 20) && !getenv("CLMGS_PINNED_NO_THP")) {
    p = thp_alloc(bytes, getenv("CLMGS_PINNED_NO_INTERLEAVE") ? 1 : numa_nodes());
    if (p) return p;
  }
  // mapped + portable: kernels may dereference it directly (zero-copy rows).  The row tables of the
  // host-resident mode are walked by threads on every socket, so large allocations are interleaved
  // over the NUMA nodes (MPOL_INTERLEAVE for the duration of the call, hipHostMallocNumaUser makes
  // the runtime honour it): one socket's memory channels would otherwise bound the host optimizer.
  const int nodes = numa_nodes();
  const bool interleave = nodes > 1 && bytes >= ((size_t)64 << 20) && !getenv("CLMGS_PINNED_NO_INTERLEAVE");
  unsigned flags = hipHostMallocMapped | hipHostMallocPortable;
  if (interleave) {
    unsigned long mask = nodes >= 64 ? ~0ul : ((1ul << nodes) - 1ul);
    if (syscall(SYS_set_mempolicy, 3 /* MPOL_INTERLEAVE */, &mask, (unsigned long)(sizeof(mask) * 8)) == 0)
      flags |= hipHostMallocNumaUser;
  }
  hipError_t e = hipHostMalloc(&p, bytes, flags);
  if (interleave) syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0ul);
  if (e != hipSuccess) {
    clmgs::set_error("hipHostMalloc(%zu) -> %s", bytes, hipGetErrorString(e));
    return nullptr;
  }
  return p;
}

// hipMemcpyAsync between pinned host memory and HBM on `stream` (the side stream of the host-resident
// mode): the copy runs on an SDMA engine and takes no compute unit away from the rendering kernels.
// kind: 1 = host -> device, 2 = device -> host.
extern "C" int clmgs_memcpy_async(void* stream, void* dst, const void* src, size_t bytes, int kind) {
  if (bytes == 0) return 0;
  if (!dst || !src || (kind != 1 && kind != 2)) { clmgs::set_error("clmgs_memcpy_async: invalid argument"); return CLMGS_EINVAL; }
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost,
                                (hipStream_t)stream);
  if (e != hipSuccess) { clmgs::set_error("hipMemcpyAsync -> %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

extern "C" int clmgs_pinned_free(void* p) {
  if (!p) return 0;
  {
    std::lock_guard<std::mutex> l(g_registered_mu);
    auto it = g_registered.find(p);
    if (it != g_registered.end()) {
      const size_t len = it->second;
      g_registered.erase(it);
      hipError_t e = hipHostUnregister(p);
      munmap(p, len);
      if (e != hipSuccess) { clmgs::set_error("hipHostUnregister -> %s", hipGetErrorString(e)); return (int)e; }
      return 0;
    }
  }
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
  // default: the CPUs the process may use (cgroup quota aware; see usable_cpus below), not the hardware
  // thread count -- 256 threads on a 16-CPU quota are throttled into a 3x slowdown
  int nt = n_threads > 0 ? n_threads : clmgs_host_usable_cpus();
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

// ---------------------------------------------------------------------------------------------
// Persistent host worker pool (the host-resident mode calls into it several times per batch; a
// fresh std::thread pool per call cost more than the work at 64+ threads).
namespace clmgs {
class HostPool {
 public:
  explicit HostPool(int n) : stop_(false), gen_(0), pending_(0) {
    for (int t = 0; t < n; ++t) workers_.emplace_back([this, t] { loop(t); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> l(mu_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  int size() const { return (int)workers_.size(); }
  // Runs fn(begin, end) over [0, n) in dynamic chunks on all workers; returns when done.
  void parallel_for(int64_t n, int64_t chunk, const std::function<void(int64_t, int64_t)>& fn) {
    if (n <= 0) return;
    std::unique_lock<std::mutex> l(mu_);
    fn_ = &fn; n_ = n; chunk_ = chunk; next_.store(0);
    pending_ = (int)workers_.size();
    ++gen_;
    cv_.notify_all();
    done_cv_.wait(l, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void loop(int) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int64_t, int64_t)>* fn;
      int64_t n, chunk;
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        fn = fn_; n = n_; chunk = chunk_;
      }
      if (fn) {
        for (;;) {
          const int64_t b = next_.fetch_add(chunk);
          if (b >= n) break;
          (*fn)(b, std::min(n, b + chunk));
        }
      }
      {
        std::lock_guard<std::mutex> l(mu_);
        if (--pending_ == 0) done_cv_.notify_one();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  bool stop_;
  uint64_t gen_;
  int pending_;
  const std::function<void(int64_t, int64_t)>* fn_ = nullptr;
  int64_t n_ = 0, chunk_ = 1;
  std::atomic<int64_t> next_{0};
};
static HostPool* g_pool = nullptr;
static std::mutex g_pool_mu;   // one job at a time (the pool has one job slot)
}  // namespace clmgs

// CPUs this process may actually use: the cgroup CPU quota (cpu.max = "<quota> <period>", cgroup v2;
// cpu.cfs_quota_us / cpu.cfs_period_us, v1) when there is one -- a container that sees 256 hardware
// threads may be entitled to 16; threads beyond the quota only get throttled -- else the hardware
// thread count.
static int usable_cpus() {
  int hw = (int)std::max(1u, std::thread::hardware_concurrency());
  long quota = -1, period = 100000;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = "";
    if (fscanf(f, "%31s %ld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atol(q);
    fclose(f);
  } else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    if (fscanf(f1, "%ld", &quota) != 1) quota = -1;
    fclose(f1);
    if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%ld", &period) != 1) period = 100000; fclose(f2); }
  }
  if (quota > 0 && period > 0) hw = (int)std::max(1l, std::min((long)hw, (quota + period - 1) / period));
  return hw;
}

extern "C" int clmgs_host_usable_cpus(void) { return usable_cpus(); }

// Starts (or resizes) the pool; n_threads <= 0: CLMGS_HOST_THREADS, else the CPUs the process may use
// (cgroup quota aware), at most one per two hardware threads (the row update is memory bound; SMT
// siblings only add contention).  Returns the pool size.
extern "C" int clmgs_host_pool_start(int n_threads) {
  std::lock_guard<std::mutex> l(clmgs::g_pool_mu);
  if (n_threads <= 0) {
    const char* e = getenv("CLMGS_HOST_THREADS");
    n_threads = e ? atoi(e) : std::min(usable_cpus(), (int)std::max(1u, std::thread::hardware_concurrency() / 2));
    // two of the quota's CPUs stay with the enqueueing thread and the feeder thread: a pool as large as a
    // cgroup quota gets the whole process throttled (16 of 16: 25.6-30.5 img/s, 14: 28.8-30.6)
    if (!e && n_threads >= 8) n_threads -= 2;
    if (n_threads <= 0) n_threads = 1;
  }
  if (clmgs::g_pool && clmgs::g_pool->size() != n_threads) { delete clmgs::g_pool; clmgs::g_pool = nullptr; }
  if (!clmgs::g_pool) clmgs::g_pool = new clmgs::HostPool(n_threads);
  return clmgs::g_pool->size();
}

// Deferred row optimizer of the host-resident mode (SH rows + Adam state in pinned host memory).
// Per row r two stamps: last_step[r] = optimizer step p/m/v are current as of, g_step[r] = step whose
// gradient is waiting in g[r] (0 = none).  For every listed row this call
//   1. replays the zero-gradient Adam steps last_step+1 .. g_step-1, applies Adam step g_step with
//      g[r] * grad_scale, then replays g_step+1 .. to_step (all of it in registers: the row's p / m / v
//      are read once and written once whatever the number of steps -- the dense reference optimizer
//      streams all N rows every batch instead, clm_offload/engine.py:316-328);
//   2. sets last_step[r] = to_step and g_step[r] = next_g_step (the step of the batch about to render
//      the row, whose gradient will land in g[r]; 0 for a flush);
//   3. copies the up-to-date parameter row into stage[k] (contiguous pinned staging buffer that a
//      hipMemcpyAsync then moves to the GPU), unless stage == NULL.
// sparse != 0 (--sparse_adam): only steps that carry a gradient exist for a row, nothing is replayed.
// Same per-element operations and order as the eager update (clmgs_host_adam_rows) would have applied.
extern "C" int clmgs_host_rows_prepare(float* p, const float* g, float* m, float* v, int32_t* last_step,
                                       int32_t* g_step, const int32_t* rows, int64_t n_rows, int cols,
                                       const float* col_lr, double beta1d, double beta2d, double epsd,
                                       int to_step, int next_g_step, int bias_correction,
                                       float grad_scale, int max_replay, float* stage, int sparse) {
  if (n_rows < 0 || cols <= 0 || cols > 64 || to_step < 0 || max_replay < 1 ||
      (n_rows > 0 && !(p && g && m && v && last_step && g_step && col_lr))) {
    clmgs::set_error("clmgs_host_rows_prepare: invalid argument");
    return CLMGS_EINVAL;
  }
  if (n_rows == 0) return 0;
  clmgs_host_pool_start(clmgs::g_pool ? clmgs::g_pool->size() : 0);
  const float beta1 = (float)beta1d, beta2 = (float)beta2d, eps = (float)epsd;
  const float ob1 = (float)(1.0 - beta1d), ob2 = (float)(1.0 - beta2d);
  // per-step scalars for steps 1 .. to_step (float, the same values the per-call variant computes)
  std::vector<float> inv_bc1(to_step + 2, 1.f), inv_sqrt_bc2(to_step + 2, 1.f);
  if (bias_correction)
    for (int s = 1; s <= to_step + 1; ++s) {
      inv_bc1[s] = (float)(1.0 / (1.0 - pow(beta1d, (double)s)));
      inv_sqrt_bc2[s] = (float)(1.0 / sqrt(1.0 - pow(beta2d, (double)s)));
    }
  // The row tables are walked in (ascending but gappy) row order: without help every row costs a chain
  // of DRAM misses (~200 ns per row and thread measured).  Each row's lines of the four tables are
  // prefetched PF rows ahead; the staged copy leaves with non-temporal stores (it is read next by the
  // DMA engine, not by this core: no read-for-ownership, no cache pollution).
  constexpr int PF = 12;
  const bool nt_ok = stage && (cols % 8 == 0) && (((uintptr_t)stage & 31) == 0);
  auto work = [&](int64_t lo, int64_t hi) {
    float pp[64], mm[64], vv[64];
    for (int64_t k = lo; k < hi; ++k) {
      if (k + PF < hi) {
        const int64_t rn = rows ? (int64_t)rows[k + PF] : k + PF;
        const size_t off = (size_t)rn * cols;
        for (int b = 0; b < cols; b += 16) {
          __builtin_prefetch(p + off + b, 1, 0);
          __builtin_prefetch(m + off + b, 1, 0);
          __builtin_prefetch(v + off + b, 1, 0);
          __builtin_prefetch(g + off + b, 0, 0);
        }
        __builtin_prefetch(last_step + rn, 1, 0);
        __builtin_prefetch(g_step + rn, 1, 0);
      }
      const int64_t r = rows ? (int64_t)rows[k] : k;
      float* pr = p + r * cols;
      int cur = last_step[r];
      const int gs = g_step[r];
      const bool pending = gs > cur && gs <= to_step;
      if (cur < to_step || pending) {
        float* mr = m + r * cols;
        float* vr = v + r * cols;
        bool any_state = pending;
        for (int c = 0; c < cols; ++c) { mm[c] = mr[c]; vv[c] = vr[c]; any_state |= (mm[c] != 0.f) | (vv[c] != 0.f); }
        if (any_state) {  // all-zero moments and no gradient: every step is the identity
          for (int c = 0; c < cols; ++c) pp[c] = pr[c];
          auto replay = [&](int from, int to) {  // zero-gradient steps from+1 .. to
            int n = to - from;
            if (n <= 0 || sparse) return;  // sparse Adam: rows without a gradient are not stepped at all
            const int exact = std::min(n, max_replay);
            for (int j = 1; j <= exact; ++j) {
              const float a = inv_bc1[from + j], b = inv_sqrt_bc2[from + j];
              for (int c = 0; c < cols; ++c) {
                mm[c] *= beta1;
                vv[c] *= beta2;
                pp[c] -= (col_lr[c] * a) * (mm[c] / (sqrtf(vv[c]) * b + eps));
              }
            }
            if (n > exact) {  // the moments have decayed below float resolution of p: decay only
              const float f1 = powf(beta1, (float)(n - exact)), f2 = powf(beta2, (float)(n - exact));
              for (int c = 0; c < cols; ++c) { mm[c] *= f1; vv[c] *= f2; }
            }
          };
          if (pending) {
            replay(cur, gs - 1);
            const float* gr = g + r * cols;
            const float a = inv_bc1[gs], b = inv_sqrt_bc2[gs];
            for (int c = 0; c < cols; ++c) {
              const float gg = gr[c] * grad_scale;
              mm[c] = beta1 * mm[c] + ob1 * gg;
              vv[c] = beta2 * vv[c] + ob2 * gg * gg;
              pp[c] -= (col_lr[c] * a) * (mm[c] / (sqrtf(vv[c]) * b + eps));
            }
            cur = gs;
          }
          replay(cur, to_step);
          for (int c = 0; c < cols; ++c) { pr[c] = pp[c]; mr[c] = mm[c]; vr[c] = vv[c]; }
        }
      }
      last_step[r] = to_step;
      g_step[r] = next_g_step;
      if (nt_ok) {
        float* dst = stage + k * cols;
        for (int c = 0; c < cols; c += 8) _mm256_stream_ps(dst + c, _mm256_loadu_ps(pr + c));
      } else if (stage) {
        memcpy(stage + k * cols, pr, sizeof(float) * cols);
      }
    }
  };
  std::lock_guard<std::mutex> l(clmgs::g_pool_mu);
  clmgs::g_pool->parallel_for(n_rows, 2048, work);
  if (nt_ok) _mm_sfence();
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
