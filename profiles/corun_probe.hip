// Do a VALU-issue-bound kernel and an HBM-streaming kernel overlap on gfx950, or do they share a budget?
// The camera pipeline of the engine co-runs the alpha-blend kernels (VALU-bound) with the HBM-bound
// ones (Adam passes, projection backward, loss); a batch nevertheless takes about the SUM of its
// kernels' solo durations.  This probe runs a long v_fma_f32 chain kernel (one-wave workgroups, 4 waves
// per SIMD, like the tile kernels) and a float4 copy kernel (read + write) alone and together on two
// streams and reports wall time, achieved rate and the effective shader clock (cycle counter / wall)
// of each.  Build + run:
//   hipcc -O3 --offload-arch=gfx950 profiles/corun_probe.hip -o /tmp/corun_probe && /tmp/corun_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(64, 4) fma_kernel(int trips, float* out, unsigned long long* cyc) {
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = 1.0f + 1e-3f * (float)(threadIdx.x + c);
  const float a = 0.999f, b = 1e-4f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < 8; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += v[c];
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n,
                                                     int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      float4 v = src[i];
      v.x += 1.f;
      dst[i] = v;
    }
}

static hipStream_t s1_plain, s2_plain;
int main() {
  hipStream_t s1lo, s2hi;
  CHECK(hipStreamCreate(&s1_plain));
  CHECK(hipStreamCreate(&s2_plain));
  int plo = 0, phi = 0;
  CHECK(hipDeviceGetStreamPriorityRange(&plo, &phi));  // (lowest, highest)
  CHECK(hipStreamCreateWithPriority(&s1lo, hipStreamNonBlocking, plo));
  CHECK(hipStreamCreateWithPriority(&s2hi, hipStreamNonBlocking, phi));
  const size_t bytes = (size_t)4 << 30;  // 4 GiB source + 4 GiB destination: far beyond the 256 MB LLC
  float4 *src, *dst;
  CHECK(hipMalloc(&src, bytes));
  CHECK(hipMalloc(&dst, bytes));
  CHECK(hipMemset(src, 0, bytes));
  float* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMalloc(&cyc, 64));
  const size_t n = bytes / 16;
  const int fma_blocks = 256 * 4 * 4 * 6;  // 6 rounds of (1024 SIMDs x 4 waves)
  const int trips = 3000;                  // 3000 x 128 FMAs per wave
  const int reps = 2;
  hipEvent_t a0, a1, b0, b1;
  CHECK(hipEventCreate(&a0)); CHECK(hipEventCreate(&a1)); CHECK(hipEventCreate(&b0)); CHECK(hipEventCreate(&b1));
  auto run = [&](bool do_fma, bool do_copy, const char* tag, int copy_block = 256, bool prio = false) {
    hipStream_t s1 = prio ? s1lo : ::s1_plain, s2 = prio ? s2hi : ::s2_plain;
    for (int it = 0; it < 3; ++it) {  // the third round is reported (clocks settled)
      CHECK(hipDeviceSynchronize());
      if (do_fma) {
        CHECK(hipEventRecord(a0, s1));
        hipLaunchKernelGGL(fma_kernel, dim3(fma_blocks), dim3(64), 0, s1, trips, out, cyc);
        CHECK(hipEventRecord(a1, s1));
      }
      if (do_copy) {
        CHECK(hipEventRecord(b0, s2));
        if (copy_block == 256) hipLaunchKernelGGL(copy_kernel<256>, dim3(256 * 16), dim3(256), 0, s2, src, dst, n, reps);
        else hipLaunchKernelGGL(copy_kernel<64>, dim3(256 * 64), dim3(64), 0, s2, src, dst, n, reps);
        CHECK(hipEventRecord(b1, s2));
      }
      CHECK(hipDeviceSynchronize());
      if (it < 2) continue;
      float ms_f = 0.f, ms_c = 0.f;
      unsigned long long c = 0;
      if (do_fma) { CHECK(hipEventElapsedTime(&ms_f, a0, a1)); CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost)); }
      if (do_copy) CHECK(hipEventElapsedTime(&ms_c, b0, b1));
      const double insts = (double)fma_blocks * trips * 128.0;
      printf("{\"case\": \"%s\"", tag);
      if (do_fma) printf(", \"fma_ms\": %.3f, \"fma_G_wave_inst_per_s\": %.1f, \"wave0_cycles\": %llu", ms_f, insts / ms_f * 1e-6, c);
      if (do_copy) printf(", \"copy_ms\": %.3f, \"copy_TBps\": %.3f", ms_c, 2.0 * bytes * reps / ms_c * 1e-9);
      printf("}\n");
    }
  };
  run(true, false, "fma alone");
  run(false, true, "copy alone, 256-thread blocks");
  run(false, true, "copy alone, 64-thread blocks", 64);
  run(true, true, "fma + copy(256-thread blocks), equal priority");
  run(true, true, "fma + copy(64-thread blocks), equal priority", 64);
  run(true, true, "fma(low priority) + copy(256-thread blocks, high priority)", 256, true);
  run(true, true, "fma(low priority) + copy(64-thread blocks, high priority)", 64, true);
  return 0;
}
