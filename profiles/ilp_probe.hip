// How much instruction-level parallelism does a wave need on gfx950?  One-wave workgroups (the tile
// kernels' shape) run v_fma_f32 in 1, 2, 4 or 8 INDEPENDENT dependency chains per lane, with the number of
// resident waves per SIMD pinned by a dynamic-LDS allocation (160 KB per CU / 4 SIMDs).  The alpha-blend
// loops are one long dependency chain per (entry, quadrant) at 5 waves/SIMD.
//   hipcc -O3 --offload-arch=gfx950 profiles/ilp_probe.hip -o /tmp/ilp_probe && /tmp/ilp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int CH>
__global__ void __launch_bounds__(64) chain_kernel(int trips, float* out) {
  extern __shared__ float dyn[];
  float v[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) v[c] = 1.0f + 1e-3f * (float)(threadIdx.x + c);
  const float a = 0.999f, b = 1e-4f;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < 128 / CH; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += v[c];
  if (s == 123.456f) { out[0] = s; dyn[threadIdx.x] = s; }
}

template <int CH>
static void run(int waves_per_simd, float* out) {
  const int lds = (160 * 1024) / (4 * waves_per_simd) - 64;  // bytes per one-wave block -> blocks per CU
  const int blocks = 256 * 4 * waves_per_simd * 4;         // 4 rounds
  const int trips = 2000;
  CHECK(hipFuncSetAttribute((const void*)chain_kernel<CH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms = 0.f;
  for (int it = 0; it < 3; ++it) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(chain_kernel<CH>, dim3(blocks), dim3(64), lds, 0, trips, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double insts = (double)blocks * trips * 128.0;
  printf("{\"chains\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"G_wave_inst_per_s\": %.1f}\n", CH, waves_per_simd, ms,
         insts / ms * 1e-6);
}

int main() {
  float* out;
  CHECK(hipMalloc(&out, 64));
  for (int w : {1, 2, 4, 5, 8}) {
    run<1>(w, out); run<2>(w, out); run<4>(w, out); run<8>(w, out);
  }
  return 0;
}
