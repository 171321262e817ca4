// Fused training loss  0.8 * L1 + 0.2 * (1 - SSIM)  against a uint8 ground-truth image,
// forward and VJP in two kernels (gfx950).  Restates strategies/base_engine.py:79-103
// (FusedCompiledLoss + loss_combined): gt = clamp(u8 / 255), SSIM per utils/loss_utils.py:26-85.
//
// HBM-bound stencil.  What is fused away compared with "fused_ssim + elementwise torch ops":
// the u8 -> float ground-truth image (never materialised), the |x - y| / mean passes, the
// HWC -> CHW transposing copy (the rendered image is read through its strides) and the sign()
// backward pass.
#include "common.h"

namespace clmgs {

// Streaming separable 11-tap window, one WAVEFRONT per strip of 64 output columns x LS_ROWS output rows x one
// channel, forward AND backward (round 4; rounds 1-3: forward in macro steps of 11 rows with 26 values in flight per
// lane and a ring of 5 statistics -- 146 VGPRs, 3 waves/SIMD, 36 % of the VALU issue rate; backward tiled 32x32 with
// nine workgroup barriers per tile and ~75 instructions of halo index arithmetic per pixel -- 30 %):
//   * ROW-granular software pipeline: the values of input row r+LP are requested while row r is processed (LP = 4
//     rows ~ 2 000 cycles of one wave's time, twice an HBM miss); a lane loads its own column, lanes 0..9 the ten halo
//     columns: 4 x 4 registers in flight instead of 26;
//   * row r goes to a small LDS ring (8 rows: the L1 term reads the centre row r-5), every lane reads its 11 taps
//     and forms the HORIZONTAL sums; LDS is the only cross-lane path, ~5 KB per wave;
//   * the VERTICAL pass is a sliding window in registers: the last 12 rows of horizontal sums live in a statically
//     indexed ring (the row loop is unrolled by 12 = 3 LP), a finished output row costs 11 FMAs per statistic;
//   * FOUR statistics, not five: SSIM and its three derivative maps only ever use E[aa] + E[bb] (sigma1^2 + sigma2^2),
//     never the two terms alone -- one ring row, 11 vertical FMAs and 11 VGPRs less than (mu1, mu2, E[aa], E[bb], E[ab]);
//   * no workgroup barrier anywhere (wave-scope fences only).
// Measured at 4608x3456 (profiles/loss_microbench.py): forward 0.40 -> 0.35 ms, backward 0.43 -> 0.265 ms.  What the
// compiler needed (from the ISA): loads under a branch, or a conversion right behind its load, made every row wait
// for ALL outstanding loads (s_waitcnt vmcnt(0/1)); with unconditional loads at clamped coordinates, masks applied at
// consumption, raw ground-truth bytes and no branch around a row step it counts them (vmcnt(15-27)).  Tried and not
// kept: 4 waves/SIMD (128 VGPRs: 8-56 B of spills per lane, scratch reloads bring vmcnt(0) back: 0.35-0.42 ms), the
// products a^2 + b^2 and a b staged once per element in LDS (4 FMAs per tap instead of 7 VALU: 164 VGPRs, +-0).
// Images are read through their strides: the renderer's [H,W,3] buffer (the engine's layout) and planar [3,H,W]
// tensors measure the same in the forward, 5 % apart in the backward (0.281 vs 0.267 ms) -- not worth a second
// layout in the tile kernels.
constexpr int LR = 5;                             // window radius
constexpr int LM = 2 * LR + 1;                    // window length (11)
constexpr int LS_W = 64, LS_ROWS = 62;            // strip: output columns (= lanes), output rows (72 input rows)
constexpr int LS_IN = LS_W + 2 * LR;              // input columns per row (74)
constexpr int LS_PITCH = LS_IN + 2;               // LDS row pitch (floats)
constexpr int LS_RING = 8;                        // LDS ring rows
constexpr int LU = 12;                            // unroll of the row loop = rows of the register ring
#ifndef CLMGS_LOSS_LP
#define CLMGS_LOSS_LP 4
#endif
constexpr int LP = CLMGS_LOSS_LP;                 // prefetch distance in rows (divides LU)
constexpr int LOSS_SLOTS = 1024;                  // partial-sum slots (spreads the atomics)
constexpr float L_C1 = 0.01f * 0.01f, L_C2 = 0.03f * 0.03f;

// compile-time weights: they become instruction literals, not 11 live VGPRs
constexpr float l_win[11] = {
    0.0010283801f, 0.0075987582f, 0.0360007721f, 0.1093606895f, 0.2130055377f, 0.2660117249f,
    0.2130055377f, 0.1093606895f, 0.0360007721f, 0.0075987582f, 0.0010283801f};

struct ImgView { const float* p; int64_t sc, sy, sx; };

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// clamp(u8 / 255, 0, 1) (base_engine.py:79-103), correctly rounded like the IEEE division for all 256
// inputs (checked exhaustively) in 3 VALU instead of the ~10 of v_div_scale/fmas/fixup:
// q = v * (1/255); q += fma(-q, 255, v) * (1/255).  The clamp is a no-op on [0, 1].
__device__ __forceinline__ float gt_from_byte(unsigned byte) {
  const float v = (float)byte;
  const float r = 1.0f / 255.0f;
  const float q = v * r;
  return fmaf(fmaf(-q, 255.0f, v), r, q);
}

// block id -> (strip, channel): the three channel passes of one strip are ids i, i+8, i+16 -- workgroups go to the
// 8 XCDs round-robin by id, so the three land on the SAME XCD close in time (channel-interleaved images: every line
// is shared by the three passes and comes from HBM once; planar images: the three read disjoint planes, harmless)
__device__ __forceinline__ bool strip_of_block(int H, int W, int& c, int& x0, int& y0, int& slot) {
  const int n_sx = (W + LS_W - 1) / LS_W, n_sy = (H + LS_ROWS - 1) / LS_ROWS;
  const int grp = blockIdx.x / 24, rem = blockIdx.x - grp * 24;
  c = rem >> 3;
  const int strip = grp * 8 + (rem & 7);
  if (strip >= n_sx * n_sy) return false;
  const int by = strip / n_sx, bx = strip - by * n_sx;
  x0 = bx * LS_W; y0 = by * LS_ROWS;
  slot = ((c * n_sy + by) * n_sx + bx) & (LOSS_SLOTS - 1);
  return true;
}

#ifndef CLMGS_LOSS_FWD_WAVES
#define CLMGS_LOSS_FWD_WAVES 3
#endif
__global__ void __launch_bounds__(64, CLMGS_LOSS_FWD_WAVES)
loss_fwd_kernel(int H, int W, ImgView img, const uint8_t* __restrict__ gt, float* __restrict__ partials,
                float* __restrict__ m1, float* __restrict__ m2, float* __restrict__ m3) {
  __shared__ float la[LS_RING][LS_PITCH], lb[LS_RING][LS_PITCH];
  const int lane = threadIdx.x;
  int c, x0, y0, slot;
  if (!strip_of_block(H, W, c, x0, y0, slot)) return;
  const size_t plane = (size_t)H * W;
  const int xo = x0 + lane;                              // output column
  const int n_rows = min(LS_ROWS, H - y0) + 2 * LR;      // input rows to walk
  const int xs = x0 - LR + lane;                         // this lane's input column
  const bool xs_ok = xs >= 0 && xs < W;
  const int xh = x0 - LR + LS_W + lane;                  // lanes 0..9: one of the ten halo columns 64..73
  const bool h_ok = lane < 2 * LR && xh < W;
  const float* img_c = img.p + c * img.sc;
  const uint8_t* gt_c = gt + c * plane;
  // The loads are UNCONDITIONAL (coordinates clamped into the image, the masks are applied when a row is consumed)
  // and the ground-truth byte stays raw until then: straight-line code lets the compiler count the outstanding
  // loads exactly (s_waitcnt vmcnt(N) with N = what was issued since), whereas loads under a branch -- or a
  // conversion right behind its load -- made it wait for ALL outstanding loads at every row (vmcnt(0/1): the
  // prefetch hid nothing; rounds 1-3 had this in their macro-step fetch too).
  // (32-bit lane offsets against wave-uniform row pointers)
  const unsigned xs_c = (unsigned)min(max(xs, 0), W - 1);
  const unsigned xh_c = (unsigned)min(x0 - LR + LS_W + min(lane, 2 * LR - 1), W - 1);  // lanes >= 10 repeat lane 9's address
  const unsigned xs_off = xs_c * (unsigned)img.sx, xh_off = xh_c * (unsigned)img.sx;
  float pa[LP], ha[LP];                                  // rows in flight
  unsigned pb[LP], hb[LP];                               // (raw ground-truth bytes)
  auto fetch = [&](int r, int s) {  // request input row r into slot s
    const int y = min(max(y0 - LR + min(r, n_rows - 1), 0), H - 1);  // wave-uniform
    const float* ir = img_c + y * img.sy;
    const uint8_t* gr = gt_c + (size_t)y * W;
    pa[s] = ir[xs_off]; pb[s] = gr[xs_c];
    ha[s] = ir[xh_off]; hb[s] = gr[xh_c];
  };
#pragma unroll
  for (int s = 0; s < LP; ++s) fetch(s, s);
  float win[LU][4];   // ring of horizontal sums: mu1 mu2 E[aa]+E[bb] E[ab]
  float l1 = 0.f, ss = 0.f;
  for (int r0 = 0; r0 < n_rows; r0 += LU) {
#pragma unroll
    for (int k = 0; k < LU; ++k) {
      const int r = r0 + k;
      {  // NO branch on r: rows past the strip's end (bottom strips only) run on clamped data and store nothing --
         // every branch around a step would make the waitcnt analysis assume the step's loads may not have been issued
        __builtin_amdgcn_sched_barrier(0);  // one row at a time
        const int s = k % LP, buf = r & (LS_RING - 1);
        {
          const int yin = y0 - LR + r;
          const bool row_ok = yin >= 0 && yin < H;       // wave-uniform; zero padding outside the image
          const bool okm = row_ok && xs_ok, okh = row_ok && h_ok;
          la[buf][lane] = okm ? pa[s] : 0.f; lb[buf][lane] = okm ? gt_from_byte(pb[s]) : 0.f;
          if (lane < 2 * LR) {
            la[buf][LS_W + lane] = okh ? ha[s] : 0.f; lb[buf][LS_W + lane] = okh ? gt_from_byte(hb[s]) : 0.f;
          }
        }
        fetch(r + LP, s);
        wave_sync();
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int j = 0; j < LM; ++j) {
          const float av = la[buf][lane + j], bv = lb[buf][lane + j];
          const float wa_ = l_win[j] * av, wb_ = l_win[j] * bv;
          s0 += wa_; s1 += wb_; s2 = fmaf(wa_, av, s2); s2 = fmaf(wb_, bv, s2); s3 = fmaf(wa_, bv, s3);
        }
        win[k][0] = s0; win[k][1] = s1; win[k][2] = s2; win[k][3] = s3;
        {  // rows r-10 .. r are in the ring: output row y0 + r - 10 (r < 10: no output row yet -- computed on
           // whatever the ring holds and masked below, 14 % of the SSIM arithmetic, instead of a branch)
          float acc[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float sacc = 0.f;
#pragma unroll
            for (int j = 0; j < LM; ++j) sacc += l_win[j] * win[(k + 2 + j) % LU][q];
            acc[q] = sacc;
          }
          const int y = y0 + r - 2 * LR;
          const bool out_ok = (r >= 2 * LR) && (r < n_rows) && (xo < W) && (y < H);
          {
            const float mu1 = acc[0], mu2 = acc[1];
            const float mu1sq = mu1 * mu1, mu2sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float sgs = (acc[2] - mu1sq) - mu2sq, sg12 = acc[3] - mu12;   // sigma1^2 + sigma2^2, sigma12
            const float A = 2.f * mu12 + L_C1, B = 2.f * sg12 + L_C2;
            const float D = mu1sq + mu2sq + L_C1, E = sgs + L_C2;
            const float iDE = __builtin_amdgcn_rcpf(D * E);  // 1-ulp reciprocals: three IEEE divides
            const float val = A * B * iDE;                   // per pixel were a tenth of the kernel
            ss += out_ok ? val : 0.f;
            const int cbuf = (r - LR) & (LS_RING - 1);  // centre: input row r - 5, still in the ring
            l1 += out_ok ? fabsf(la[cbuf][lane + LR] - lb[cbuf][lane + LR]) : 0.f;
            if (m1 && out_ok) {
              const float d_mu1 = 2.f * mu2 * B * iDE - val * 2.f * mu1 * __builtin_amdgcn_rcpf(D);
              const float d_s1 = -val * __builtin_amdgcn_rcpf(E), d_s12 = 2.f * A * iDE;
              const size_t orow = c * plane + (size_t)y * W;  // wave-uniform
              (m1 + orow)[(unsigned)xo] = d_mu1 - 2.f * mu1 * d_s1 - mu2 * d_s12;
              (m2 + orow)[(unsigned)xo] = d_s1; (m3 + orow)[(unsigned)xo] = d_s12;
            }
          }
        }
      }
    }
  }
  const float tl1 = wave_sum(l1), tss = wave_sum(ss);
  if (lane == 0) {
    atomicAdd(partials + 2 * slot, tl1);
    atomicAdd(partials + 2 * slot + 1, tss);
  }
}

// Backward, same streaming shape: the three derivative maps of a channel walk through the window (3 statistics, a
// 36-register ring), the rendered / ground-truth values of the OUTPUT pixel ride in the same prefetch stream (requested
// LP rows before they are used, so every wait is for the oldest outstanding load), and the cotangent row leaves as one
// run per row (planar images: 256 contiguous bytes).
// v_img (strides of vimg) = v * ( w_l1 * sign(x - y) - w_ssim * dSSIMsum/dx ) / numel
#ifndef CLMGS_LOSS_BWD_WAVES
#define CLMGS_LOSS_BWD_WAVES 3
#endif
__global__ void __launch_bounds__(64, CLMGS_LOSS_BWD_WAVES)
loss_bwd_kernel(int H, int W, ImgView img, const uint8_t* __restrict__ gt, const float* __restrict__ v,
                float w_l1_over_numel, float w_ssim_over_numel, const float* __restrict__ m1,
                const float* __restrict__ m2, const float* __restrict__ m3, float* __restrict__ v_img,
                int64_t vsc, int64_t vsy, int64_t vsx) {
  __shared__ float lm[2][3][LS_PITCH];
  const int lane = threadIdx.x;
  int c, x0, y0, slot;
  if (!strip_of_block(H, W, c, x0, y0, slot)) return;
  const size_t plane = (size_t)H * W;
  const int xo = x0 + lane;
  const int n_out = min(LS_ROWS, H - y0);
  const int n_rows = n_out + 2 * LR;
  const int xs = x0 - LR + lane;
  const bool xs_ok = xs >= 0 && xs < W;
  // halo: lanes 0..9 fetch map 1's ten halo columns, lanes 16..25 map 2's, lanes 32..41 map 3's -- ONE load and one
  // register per row for the three maps
  const int hg = lane >> 4, hl = lane & 15;
  const int xh = x0 - LR + LS_W + hl;
  const bool h_ok = hg < 3 && hl < 2 * LR && xh < W;
  const float* mh = (hg == 0 ? m1 : (hg == 1 ? m2 : m3)) + c * plane;  // (lanes 48..63: map 3 again, never stored)
  const float* m1c = m1 + c * plane;
  const float* m2c = m2 + c * plane;
  const float* m3c = m3 + c * plane;
  const float* img_c = img.p + c * img.sc;
  const uint8_t* gt_c = gt + c * plane;
  const float vv = v[0];
  // unconditional loads at clamped coordinates, masks at consumption, raw ground-truth byte (see the forward kernel)
  const unsigned xs_c = (unsigned)min(max(xs, 0), W - 1);
  const unsigned xh_c = (unsigned)min(x0 - LR + LS_W + min(hl, 2 * LR - 1), W - 1);
  const unsigned xo_c = (unsigned)min(xo, W - 1);
  const unsigned xo_off = xo_c * (unsigned)img.sx;
  const unsigned xo_voff = (unsigned)xo * (unsigned)vsx;
  float p1[LP], p2[LP], p3[LP], ph[LP], px[LP];
  unsigned py[LP];
  auto fetch = [&](int r, int s) {
    const int rc = min(r, n_rows - 1);
    const int y = min(max(y0 - LR + rc, 0), H - 1);         // input row of the maps (wave-uniform)
    const int yo = min(max(y0 + rc - 2 * LR, 0), H - 1);    // output row finished at iteration r
    const size_t o = (size_t)y * W;   // wave-uniform
    p1[s] = (m1c + o)[xs_c]; p2[s] = (m2c + o)[xs_c]; p3[s] = (m3c + o)[xs_c];
    ph[s] = (mh + o)[xh_c];
    px[s] = (img_c + yo * img.sy)[xo_off];
    py[s] = (gt_c + (size_t)yo * W)[xo_c];
  };
#pragma unroll
  for (int s = 0; s < LP; ++s) fetch(s, s);
  float win[LU][3];
  for (int r0 = 0; r0 < n_rows; r0 += LU) {
#pragma unroll
    for (int k = 0; k < LU; ++k) {
      const int r = r0 + k;
      {  // no branch on r (see the forward kernel)
        __builtin_amdgcn_sched_barrier(0);
        const int s = k % LP, buf = r & 1;
        {
          const int yin = y0 - LR + r;
          const bool row_ok = yin >= 0 && yin < H;
          const bool okm = row_ok && xs_ok;
          lm[buf][0][lane] = okm ? p1[s] : 0.f; lm[buf][1][lane] = okm ? p2[s] : 0.f; lm[buf][2][lane] = okm ? p3[s] : 0.f;
          if (hg < 3 && hl < 2 * LR) lm[buf][hg][LS_W + hl] = (row_ok && h_ok) ? ph[s] : 0.f;
        }
        const float xv = px[s], yv = gt_from_byte(py[s]);
        fetch(r + LP, s);
        wave_sync();
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < LM; ++j) {
          s0 = fmaf(l_win[j], lm[buf][0][lane + j], s0);
          s1 = fmaf(l_win[j], lm[buf][1][lane + j], s1);
          s2 = fmaf(l_win[j], lm[buf][2][lane + j], s2);
        }
        win[k][0] = s0; win[k][1] = s1; win[k][2] = s2;
        {
          float g[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            float sacc = 0.f;
#pragma unroll
            for (int j = 0; j < LM; ++j) sacc += l_win[j] * win[(k + 2 + j) % LU][q];
            g[q] = sacc;
          }
          const int y = y0 + r - 2 * LR;
          if (r >= 2 * LR && r < n_rows && xo < W) {
            const float sgn = (xv > yv) ? 1.f : ((xv < yv) ? -1.f : 0.f);
            const float dss = g[0] + 2.f * xv * g[1] + yv * g[2];
            (v_img + (c * vsc + y * vsy))[xo_voff] = vv * (w_l1_over_numel * sgn - w_ssim_over_numel * dss);
          }
        }
        wave_sync();  // the two-row LDS ring: row r is consumed before row r+2 lands in its place
      }
    }
  }
}

}  // namespace clmgs

using namespace clmgs;

extern "C" int clmgs_loss_slots(void) { return LOSS_SLOTS; }

extern "C" int clmgs_l1_ssim_loss_fwd(void* stream, int H, int W, const float* img, int64_t stride_c,
                                      int64_t stride_y, int64_t stride_x, const uint8_t* gt_u8,
                                      float* partials, float* m1, float* m2, float* m3) {
  CLMGS_CHECK_ARG(H >= 1 && W >= 1 && img && gt_u8 && partials);
  CLMGS_CHECK_ARG((m1 && m2 && m3) || (!m1 && !m2 && !m3));
  ImgView v{img, stride_c, stride_y, stride_x};
  const int64_t strips = (int64_t)ceil_div(W, LS_W) * ceil_div(H, LS_ROWS);
  dim3 grid((unsigned)(24 * ceil_div(strips, 8)));
  hipLaunchKernelGGL(loss_fwd_kernel, grid, dim3(64), 0, (hipStream_t)stream, H, W, v, gt_u8,
                     partials, m1, m2, m3);
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_l1_ssim_loss_bwd(void* stream, int H, int W, const float* img, int64_t stride_c,
                                      int64_t stride_y, int64_t stride_x, const uint8_t* gt_u8,
                                      const float* v_loss, float lambda_dssim, const float* m1,
                                      const float* m2, const float* m3, float* v_img) {
  CLMGS_CHECK_ARG(H >= 1 && W >= 1 && img && gt_u8 && v_loss && m1 && m2 && m3 && v_img);
  ImgView v{img, stride_c, stride_y, stride_x};
  const double numel = 3.0 * (double)H * (double)W;
  const int64_t strips = (int64_t)ceil_div(W, LS_W) * ceil_div(H, LS_ROWS);
  dim3 grid((unsigned)(24 * ceil_div(strips, 8)));
  hipLaunchKernelGGL(loss_bwd_kernel, grid, dim3(64), 0, (hipStream_t)stream, H, W, v, gt_u8, v_loss,
                     (float)((1.0 - lambda_dssim) / numel), (float)(lambda_dssim / numel), m1, m2, m3,
                     v_img, stride_c, stride_y, stride_x);
  CLMGS_LAUNCH_CHECK();
  return 0;
}
