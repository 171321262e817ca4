// VALU issue-rate calibration for gfx950 (MI355X): the compute-side roofline of the alpha-blend tile
// kernels.  Each test runs the same wave64 instruction in long unrolled independent chains on every
// SIMD of the chip (8 waves / SIMD and 4 waves / SIMD) and reports
//   G wave-instructions / s (wall clock, whole chip)   and   shader cycles per instruction per SIMD.
// Build + run:  hipcc -O3 --offload-arch=gfx950 profiles/valu_calib.hip -o /tmp/valu_calib && /tmp/valu_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int CHAINS = 8;
constexpr int UNROLL = 16;  // instructions per chain per loop trip

enum Op { FMA = 0, MUL_ADD, EXP, RCP, DPP_ADD, PERMLANE32_SWAP, CNDMASK, LDS_B128_BCAST, FMA_WITH_SALU, MIN_CMP, PK_FMA, PK_MUL, PK_ADD,
          CNDMASK_SGPR, CMP_CNDMASK_VCC, CMP_CNDMASK_SGPR, CNDMASK_VCC_2SRC, MAX_F32, AND_B32, MOV_B32, MED3_F32, CMP_ONLY, N_OPS };
static const char* NAMES[N_OPS] = {"v_fma_f32", "v_mul_f32+v_add_f32", "v_exp_f32", "v_rcp_f32", "v_add_f32_dpp(row_shr:1)",
                                   "v_permlane32_swap_b32", "v_cndmask_b32(vcc)", "ds_read_b128(broadcast)", "v_fma_f32 + 1 s_add per 2",
                                   "v_min_f32+v_cmp_ge_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
                                   "v_cndmask_b32_e64(sgpr pair)", "v_cmp_gt_f32(vcc)+v_cndmask_b32(vcc)", "v_cmp_gt_f32_e64(sgpr)+v_cndmask_b32_e64(sgpr)",
                                   "v_cndmask_b32(vcc) dst!=src", "v_max_f32", "v_and_b32", "v_mov_b32", "v_med3_f32", "v_cmp_gt_f32(vcc)"};

template <int OP>
__global__ void __launch_bounds__(256) calib(int trips, float* out, unsigned long long* cycles) {
  __shared__ float4 lds[64];
  if (threadIdx.x < 64) lds[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  float v[CHAINS];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 w[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) { v[c] = 1.0f + 1e-3f * (float)(threadIdx.x + c); w[c] = f2{v[c], v[c] * 0.5f}; }
  const f2 a2 = f2{0.999f, 0.998f}, b2 = f2{1e-4f, 2e-4f};
  const float a = 0.999f, b = 1e-4f;
  int sacc = 0;
  unsigned long long smask = 0x5555555555555555ull + (unsigned long long)(trips == 123456);
  asm volatile("s_mov_b64 vcc, %0" : : "s"(smask) : "vcc");
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));
        if (OP == MUL_ADD) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[c]) : "v"(a)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[c]) : "v"(b)); }
        if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[c]));
        if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[c]));
        if (OP == DPP_ADD) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[c]));
        if (OP == PERMLANE32_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[c]), "+v"(v[(c + 1) % CHAINS]));
        if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(a));
        if (OP == LDS_B128_BCAST) {
          float4 r;
          asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"((int)((u * 16) & 1008)));
          asm volatile("s_waitcnt lgkmcnt(4)");
          v[c] += r.x;
        }
        if (OP == FMA_WITH_SALU) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));
          if (c & 1) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc));
        }
        if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[c]) : "v"(a2), "v"(b2));
        if (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[c]) : "v"(a2));
        if (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[c]) : "v"(b2));
        if (OP == CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "s"(smask));
        if (OP == CMP_CNDMASK_VCC) { asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[c]), "v"(b) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(a) : "vcc"); }
        if (OP == CMP_CNDMASK_SGPR) { unsigned long long m_; asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m_) : "v"(v[c]), "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "s"(m_)); }
        if (OP == CNDMASK_VCC_2SRC) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[c]) : "v"(v[(c + 1) % CHAINS]), "v"(a));
        if (OP == MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[c]) : "v"(a));
        if (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[c]) : "v"(a));
        if (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(v[c]) : "v"(v[(c + 1) % CHAINS]));
        if (OP == MED3_F32) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));
        if (OP == CMP_ONLY) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[c]), "v"(b) : "vcc");
        if (OP == MIN_CMP) { asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[c]) : "v"(a)); asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(v[c]), "v"(b) : "vcc"); }
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = (float)sacc;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += v[c] + w[c].x + w[c].y;
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int OP>
void run(int waves_per_simd, float* out, unsigned long long* cyc_d) {
  const int trips = 2000;
  const int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = one per SIMD of a CU
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(calib<OP>, dim3(blocks), dim3(256), 0, 0, 10, out, cyc_d);  // warm-up
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(calib<OP>, dim3(blocks), dim3(256), 0, 0, trips, out, cyc_d);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long cyc = 0;
  CHECK(hipMemcpy(&cyc, cyc_d, sizeof(cyc), hipMemcpyDeviceToHost));
  int per = (OP == MUL_ADD || OP == MIN_CMP || OP == CMP_CNDMASK_VCC || OP == CMP_CNDMASK_SGPR) ? 2 : 1;
  const double instr_per_wave = (double)trips * UNROLL * CHAINS * per;
  const double total = instr_per_wave * blocks * 4;
  printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"G_wave_instr_per_s\": %.1f, \"cycles_per_instr_per_simd\": %.2f, \"ms\": %.3f}\n",
         NAMES[OP], waves_per_simd, total / (ms * 1e-3) / 1e9, (double)cyc / (instr_per_wave * waves_per_simd), ms);
}

int main() {
  float* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&cyc, 64));
  for (int w : {8, 5, 4}) {
    run<CNDMASK_SGPR>(w, out, cyc); run<CMP_CNDMASK_VCC>(w, out, cyc); run<CMP_CNDMASK_SGPR>(w, out, cyc);
    run<CNDMASK_VCC_2SRC>(w, out, cyc); run<MAX_F32>(w, out, cyc); run<AND_B32>(w, out, cyc); run<MOV_B32>(w, out, cyc);
    run<MED3_F32>(w, out, cyc); run<CMP_ONLY>(w, out, cyc);
    run<FMA>(w, out, cyc); run<MUL_ADD>(w, out, cyc); run<EXP>(w, out, cyc); run<RCP>(w, out, cyc);
    run<DPP_ADD>(w, out, cyc); run<PERMLANE32_SWAP>(w, out, cyc); run<CNDMASK>(w, out, cyc);
    run<LDS_B128_BCAST>(w, out, cyc); run<MIN_CMP>(w, out, cyc);
    run<PK_FMA>(w, out, cyc); run<PK_MUL>(w, out, cyc); run<PK_ADD>(w, out, cyc);
  }
  return 0;
}
