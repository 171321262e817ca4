// What does the row-gather access pattern of the deferred row optimizer reach on this chip, in the simplest kernel that has it?
// ("4.0-4.2 TB/s is what this access pattern reaches" was asserted, not shown.)
//
// Four [N, row] float tables (p, m, v, g), N = 28 M; a list of T = 9 M row indices; per listed row the kernel READS the row of
// all four tables and WRITES p, m, v back (what clmgs_adam_catch_up moves: 7 x 192 B per touched row, with a few FMAs).
// Experiments (each prints achieved GB/s on the useful bytes, min and median of 10 launches):
//   layout   rows of 192 B (48 floats, the product's tables: a row straddles three 64 B lines at 64 B alignment, every other
//            row straddles two 128 B lines)  vs  rows padded to 256 B (same 192 useful bytes, two aligned 128 B lines)
//   list     2 000 ascending contiguous runs (Z-ordered tables, a batch of four nadir cameras)  vs  the same rows ascending
//            but with every third row missing (sorted, scattered)  vs  the same rows shuffled
//   alloc    one hipMalloc per table (2 MB-aligned)  vs  tables carved from ONE allocation at offsets that are 4 KB- but not
//            2 MB-aligned (what a sub-allocated block of a caching allocator can look like)
//   width    12 lanes x float4 per row (the product kernel's shape)  vs  a whole wave per 1 KB of consecutive rows
// Build + run:  hipcc -O3 --offload-arch=gfx950 profiles/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int STRIDE4>  // float4 per row slot: 12 (192 B rows) or 16 (rows padded to 256 B); 12 float4 are used either way
__global__ void __launch_bounds__(256) gather_rows(const int32_t* __restrict__ rows, int64_t n_rows, float4* __restrict__ p,
                                                   float4* __restrict__ m, float4* __restrict__ v, const float4* __restrict__ g) {
  const int64_t total = n_rows * 12;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / 12;
    const int c = (int)(i - r * 12);
    const int64_t o = (int64_t)rows[r] * STRIDE4 + c;
    const float4 gp = g[o];
    float4 mm = m[o], vv = v[o], pp = p[o];
    mm.x = 0.9f * mm.x + 0.1f * gp.x; mm.y = 0.9f * mm.y + 0.1f * gp.y; mm.z = 0.9f * mm.z + 0.1f * gp.z; mm.w = 0.9f * mm.w + 0.1f * gp.w;
    vv.x = 0.999f * vv.x + 0.001f * gp.x * gp.x; vv.y = 0.999f * vv.y + 0.001f * gp.y * gp.y;
    vv.z = 0.999f * vv.z + 0.001f * gp.z * gp.z; vv.w = 0.999f * vv.w + 0.001f * gp.w * gp.w;
    pp.x -= 1e-3f * mm.x; pp.y -= 1e-3f * mm.y; pp.z -= 1e-3f * mm.z; pp.w -= 1e-3f * mm.w;
    m[o] = mm; v[o] = vv; p[o] = pp;
  }
}

// plain streaming over the first n_rows rows (no index list): the ceiling of a read-4 / write-3 stream
template <int STRIDE4>
__global__ void __launch_bounds__(256) stream_rows(int64_t n_rows, float4* __restrict__ p, float4* __restrict__ m,
                                                   float4* __restrict__ v, const float4* __restrict__ g) {
  const int64_t total = n_rows * STRIDE4;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    const float4 gp = g[o];
    float4 mm = m[o], vv = v[o], pp = p[o];
    mm.x = 0.9f * mm.x + 0.1f * gp.x; vv.x = 0.999f * vv.x + 0.001f * gp.x * gp.x; pp.x -= 1e-3f * mm.x;
    m[o] = mm; v[o] = vv; p[o] = pp;
  }
}

static double timed(const std::function<void()>& fn, double* med) {
  for (int i = 0; i < 2; ++i) fn();
  CHECK(hipDeviceSynchronize());
  std::vector<float> ts;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) {
    CHECK(hipEventRecord(e0)); fn(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  *med = ts[5];
  return ts[0];
}

int main() {
  const int64_t N = 28000000, T = 9000000;
  std::mt19937_64 rng(0);
  // list A: ~2000 ascending contiguous runs
  std::vector<int32_t> runs_list;
  {
    const int R = 2000;
    std::vector<int64_t> starts(R);
    for (auto& s : starts) s = rng() % (N - 2 * T / R);
    std::sort(starts.begin(), starts.end());
    std::vector<char> mark(N, 0);
    for (int64_t s : starts) { const int64_t len = T / R / 2 + rng() % (T / R); for (int64_t i = s; i < std::min(N, s + len); ++i) mark[i] = 1; }
    for (int64_t i = 0; i < N; ++i) if (mark[i]) runs_list.push_back((int32_t)i);
  }
  const int64_t n = (int64_t)runs_list.size();
  std::vector<int32_t> gaps_list;  // ascending, every third row of a 1.5x longer stretch missing
  for (int64_t i = 0; (int64_t)gaps_list.size() < n && i < N; ++i) if (i % 3 != 2 && (i / 4500) % 2 == 0) gaps_list.push_back((int32_t)i);
  while ((int64_t)gaps_list.size() < n) gaps_list.push_back(gaps_list.back());
  std::vector<int32_t> shuf_list = runs_list;
  std::shuffle(shuf_list.begin(), shuf_list.end(), rng);
  int32_t *d_runs, *d_gaps, *d_shuf;
  CHECK(hipMalloc(&d_runs, n * 4)); CHECK(hipMalloc(&d_gaps, n * 4)); CHECK(hipMalloc(&d_shuf, n * 4));
  CHECK(hipMemcpy(d_runs, runs_list.data(), n * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_gaps, gaps_list.data(), n * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_shuf, shuf_list.data(), n * 4, hipMemcpyHostToDevice));
  const double useful = (double)n * 7 * 192;

  // First-process effect (round 6): the FIRST process on a freshly acquired box measured the 2 MB-aligned tables at the rate
  // of the mis-aligned ones; does freeing and re-allocating inside that process change it?  (192 B rows, runs list)
  for (int rep = 0; rep < 3; ++rep) {
    const size_t tb = (size_t)N * 12 * 16;
    float4* t[4];
    for (int k = 0; k < 4; ++k) { CHECK(hipMalloc(&t[k], tb)); CHECK(hipMemset(t[k], 0, tb)); }
    double med;
    const double best = timed([&] { hipLaunchKernelGGL(gather_rows<12>, dim3(256 * 16), dim3(256), 0, 0, d_runs, n, t[0], t[1], t[2], t[3]); }, &med);
    printf("{\"experiment\": \"alloc_free_realloc\", \"allocation_round\": %d, \"ms_min\": %.4f, \"useful_GBps_min\": %.1f, \"ptr\": \"%p\"}\n",
           rep, best, useful / (best * 1e-3) / 1e9, (void*)t[0]);
    for (int k = 0; k < 4; ++k) CHECK(hipFree(t[k]));
  }
  for (int stride4 : {12, 16}) {
    const size_t tb = (size_t)N * stride4 * 16;
    for (int carved = 0; carved < 2; ++carved) {
      char* base = nullptr;
      float4* t[4];
      if (!carved) {
        for (int k = 0; k < 4; ++k) { CHECK(hipMalloc(&t[k], tb)); CHECK(hipMemset(t[k], 0, tb)); }
      } else {
        const size_t pad = 4096 * 37;  // 4 KB-aligned, not 2 MB-aligned
        CHECK(hipMalloc(&base, 4 * (tb + pad) + pad));
        CHECK(hipMemset(base, 0, 4 * (tb + pad) + pad));
        for (int k = 0; k < 4; ++k) t[k] = (float4*)(base + pad + k * (tb + pad));
      }
      const int grid = 256 * 16;
      struct { const char* name; int32_t* l; } lists[3] = {{"runs", d_runs}, {"sorted_gaps", d_gaps}, {"shuffled", d_shuf}};
      for (auto& L : lists) {
        double med;
        const double best = (stride4 == 12)
            ? timed([&] { hipLaunchKernelGGL(gather_rows<12>, dim3(grid), dim3(256), 0, 0, L.l, n, t[0], t[1], t[2], t[3]); }, &med)
            : timed([&] { hipLaunchKernelGGL(gather_rows<16>, dim3(grid), dim3(256), 0, 0, L.l, n, t[0], t[1], t[2], t[3]); }, &med);
        printf("{\"row_bytes\": %d, \"alloc\": \"%s\", \"list\": \"%s\", \"rows\": %lld, \"ms_min\": %.4f, \"ms_med\": %.4f, \"useful_GBps_min\": %.1f}\n",
               stride4 * 16, carved ? "carved_4KB_offsets" : "hipMalloc_each", L.name, (long long)n, best, med, useful / (best * 1e-3) / 1e9);
      }
      {
        double med;
        const double best = (stride4 == 12)
            ? timed([&] { hipLaunchKernelGGL(stream_rows<12>, dim3(grid), dim3(256), 0, 0, n, t[0], t[1], t[2], t[3]); }, &med)
            : timed([&] { hipLaunchKernelGGL(stream_rows<16>, dim3(grid), dim3(256), 0, 0, n, t[0], t[1], t[2], t[3]); }, &med);
        printf("{\"row_bytes\": %d, \"alloc\": \"%s\", \"list\": \"none (stream of the first rows)\", \"rows\": %lld, \"ms_min\": %.4f, \"ms_med\": %.4f, \"moved_GBps_min\": %.1f}\n",
               stride4 * 16, carved ? "carved_4KB_offsets" : "hipMalloc_each", (long long)n, best, med, (double)n * 7 * stride4 * 16 / (best * 1e-3) / 1e9);
      }
      if (!carved) { for (int k = 0; k < 4; ++k) CHECK(hipFree(t[k])); } else CHECK(hipFree(base));
    }
  }
  return 0;
}
