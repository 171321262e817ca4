// Fused training loss  0.8 * L1 + 0.2 * (1 - SSIM)  against a uint8 ground-truth image,
// forward and VJP in two kernels (gfx950).  Restates strategies/base_engine.py:79-103
// (FusedCompiledLoss + loss_combined): gt = clamp(u8 / 255), SSIM per utils/loss_utils.py:26-85.
//
// HBM-bound stencil.  What is fused away compared with "fused_ssim + elementwise torch ops":
// the u8 -> float ground-truth image (never materialised), the |x - y| / mean passes, the
// HWC -> CHW transposing copy (the rendered image is read through its strides) and the sign()
// backward pass.  One 256-thread block owns a 16x16 pixel tile for all three channels; the
// 26x26x3 halos are fetched in one sweep that is contiguous in memory for HWC images.
#include "common.h"

namespace clmgs {

// Streaming separable 11-tap window, one WAVEFRONT per strip of 64 output columns x LS_ROWS output
// rows (x one channel in the forward):
//   * the wave walks down the strip in MACRO steps of 11 input rows.  The 11 x 74 input values of
//     the NEXT macro step (with the +-5 column halo) are requested while the current 11 rows are
//     processed (13 loads per lane and array stay in flight for ~11 row times: HBM latency is
//     covered without relying on occupancy); at the macro boundary they are written to a 16-row LDS
//     ring;
//   * per row every lane reads its 11 taps from the ring and forms the HORIZONTAL sums;
//   * the VERTICAL pass is a sliding window in registers: the last 11 rows of horizontal sums
//     live in a statically indexed ring (the row loop is unrolled by 11), so a finished output row
//     costs 11 FMAs per statistic and no LDS at all;
//   * no workgroup barrier anywhere (wave-scope fences only), ~10 KB of LDS per wave.
// The tiled version it replaces (32x32 tile, 42 KB of LDS per 256-thread block, five barriers per
// channel) sat at ~55 % VALU utilisation with 3 waves/SIMD.
constexpr int LR = 5;                             // window radius
constexpr int LM = 2 * LR + 1;                    // rows per macro step = window length (11)
constexpr int LS_W = 64, LS_ROWS = 66;            // strip: output columns (= lanes), output rows (6 macros)
constexpr int LS_IN = LS_W + 2 * LR;              // input columns per row (74)
constexpr int LS_PITCH = LS_IN + 2;               // LDS row pitch (floats)
constexpr int LS_RING = 16;                       // LDS ring rows: 11 of the macro + 5 older (L1 centre)
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
__device__ __forceinline__ float gt_val(const uint8_t* __restrict__ gt, size_t o) {
  const float v = (float)gt[o];
  const float r = 1.0f / 255.0f;
  const float q = v * r;
  return fmaf(fmaf(-q, 255.0f, v), r, q);
}

#ifndef CLMGS_LOSS_FWD_WAVES
#define CLMGS_LOSS_FWD_WAVES 3
#endif
__global__ void __launch_bounds__(64, CLMGS_LOSS_FWD_WAVES)
loss_fwd_kernel(int H, int W, ImgView img, const uint8_t* __restrict__ gt, float* __restrict__ partials,
                float* __restrict__ m1, float* __restrict__ m2, float* __restrict__ m3) {
  __shared__ float la[LS_RING][LS_PITCH], lb[LS_RING][LS_PITCH];
  const int lane = threadIdx.x;
  // 1-D grid, block id -> (strip, channel) so that the three channel passes of one strip are ids i,
  // i+8, i+16: workgroups go to the 8 XCDs round-robin by id, so the three land on the SAME XCD close
  // in time and the channel-interleaved image lines (each pass uses a third of every line) are
  // fetched from HBM once instead of three times.
  const int n_sx = (W + LS_W - 1) / LS_W, n_sy = (H + LS_ROWS - 1) / LS_ROWS;
  const int grp = blockIdx.x / 24, rem = blockIdx.x - grp * 24;
  const int c = rem >> 3, strip = grp * 8 + (rem & 7);
  if (strip >= n_sx * n_sy) return;
  const int by = strip / n_sx, bx = strip - by * n_sx;
  const int x0 = bx * LS_W, y0 = by * LS_ROWS;
  const size_t plane = (size_t)H * W;
  const int xo = x0 + lane;                              // output column
  const int n_rows = min(LS_ROWS, H - y0) + 2 * LR;      // input rows to walk
  // Staging map: per macro step a lane requests its OWN column of the 11 rows (row pointer = scalar
  // base + lane: two VALU per load, a wave-uniform branch for rows outside the image) plus two of
  // the 110 halo values (columns 64..73 of the 11 rows).
  const int xs = x0 - LR + lane;                         // this lane's input column
  const bool xs_ok = xs >= 0 && xs < W;
  int hrow[2], hcol[2];
  bool h_ok[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int h = lane + 64 * u;
    hrow[u] = h / (2 * LR); hcol[u] = LS_W + h - hrow[u] * (2 * LR);
    const int xh = x0 - LR + hcol[u];
    h_ok[u] = h < LM * 2 * LR && xh < W;
  }
  float pa[LM], pb[LM], ha[2], hb[2];                    // the macro step in flight
  auto fetch = [&](int r0) {  // request input rows r0 .. r0+10 (zero outside the image)
#pragma unroll
    for (int k = 0; k < LM; ++k) {
      const int y = y0 - LR + r0 + k;                    // wave-uniform
      pa[k] = 0.f; pb[k] = 0.f;
      if (y >= 0 && y < H && xs_ok) {
        pa[k] = img.p[c * img.sc + y * img.sy + xs * img.sx];
        pb[k] = gt_val(gt, c * plane + (size_t)y * W + xs);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int y = y0 - LR + r0 + hrow[u], x = x0 - LR + hcol[u];
      ha[u] = 0.f; hb[u] = 0.f;
      if (h_ok[u] && y >= 0 && y < H) {
        ha[u] = img.p[c * img.sc + y * img.sy + x * img.sx];
        hb[u] = gt_val(gt, c * plane + (size_t)y * W + x);
      }
    }
  };
  fetch(0);
  float win[LM][5];   // ring of horizontal sums: mu1 mu2 E[aa] E[bb] E[ab]
  float l1 = 0.f, ss = 0.f;
  for (int r0 = 0; r0 < n_rows; r0 += LM) {
    // macro boundary: rows r0 .. r0+10 -> LDS ring, then the next macro step takes off
#pragma unroll
    for (int k = 0; k < LM; ++k) {
      const int slot = (r0 + k) & (LS_RING - 1);
      la[slot][lane] = pa[k]; lb[slot][lane] = pb[k];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (lane + 64 * u < LM * 2 * LR) {
        const int slot = (r0 + hrow[u]) & (LS_RING - 1);
        la[slot][hcol[u]] = ha[u]; lb[slot][hcol[u]] = hb[u];
      }
    }
    fetch(r0 + LM);
    wave_sync();
#pragma unroll
    for (int k = 0; k < LM; ++k) {
      const int r = r0 + k;
      if (r < n_rows) {  // wave-uniform
        const int buf = r & (LS_RING - 1);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
        for (int j = 0; j < LM; ++j) {
          const float av = la[buf][lane + j], bv = lb[buf][lane + j];
          const float wa_ = l_win[j] * av, wb_ = l_win[j] * bv;
          s0 += wa_; s1 += wb_; s2 += wa_ * av; s3 += wb_ * bv; s4 += wa_ * bv;
        }
        win[k][0] = s0; win[k][1] = s1; win[k][2] = s2; win[k][3] = s3; win[k][4] = s4;
        if (r >= 2 * LR) {  // rows r-10 .. r are in the ring: output row y0 + r - 10
          float acc[5];
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            float sacc = 0.f;
#pragma unroll
            for (int j = 0; j < LM; ++j) sacc += l_win[j] * win[(k + 1 + j) % LM][q];
            acc[q] = sacc;
          }
          const int y = y0 + r - 2 * LR;
          if (xo < W && y < H) {
            const float mu1 = acc[0], mu2 = acc[1];
            const float mu1sq = mu1 * mu1, mu2sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float sg1 = acc[2] - mu1sq, sg2 = acc[3] - mu2sq, sg12 = acc[4] - mu12;
            const float A = 2.f * mu12 + L_C1, B = 2.f * sg12 + L_C2;
            const float D = mu1sq + mu2sq + L_C1, E = sg1 + sg2 + L_C2;
            const float iDE = __builtin_amdgcn_rcpf(D * E);  // 1-ulp reciprocals: three IEEE divides
            const float val = A * B * iDE;                   // per pixel were a tenth of the kernel
            ss += val;
            const int cbuf = (r - LR) & (LS_RING - 1);  // centre: input row r - 5, still in the ring
            l1 += fabsf(la[cbuf][lane + LR] - lb[cbuf][lane + LR]);
            if (m1) {
              const float d_mu1 = 2.f * mu2 * B * iDE - val * 2.f * mu1 * __builtin_amdgcn_rcpf(D);
              const float d_s1 = -val * __builtin_amdgcn_rcpf(E), d_s12 = 2.f * A * iDE;
              const size_t oidx = c * plane + (size_t)y * W + xo;
              m1[oidx] = d_mu1 - 2.f * mu1 * d_s1 - mu2 * d_s12; m2[oidx] = d_s1; m3[oidx] = d_s12;
            }
          }
        }
      }
    }
    wave_sync();  // the ring rows of this macro step are consumed before the next one lands
  }
  const float tl1 = wave_sum(l1), tss = wave_sum(ss);
  if (lane == 0) {
    const int slot = ((c * n_sy + by) * n_sx + bx) & (LOSS_SLOTS - 1);
    atomicAdd(partials + 2 * slot, tl1);
    atomicAdd(partials + 2 * slot + 1, tss);
  }
}

// Backward: the tiled form (32x32 output pixels per 256-thread block, separable passes through LDS
// with 4-wide register blocking).  A streaming version like the forward was measured equal or
// slower (0.52-0.62 vs 0.49 ms): its output-pixel loads sit behind the prefetch in vmcnt order.
constexpr int LT = 32, LH = LT + 2 * LR;  // tile, halo edge (42)
// v_img (same strides as img) = v * ( w_l1 * sign(x - y) - w_ssim * dSSIMsum/dx ) / numel
__global__ void __launch_bounds__(256)
loss_bwd_kernel(int H, int W, ImgView img, const uint8_t* __restrict__ gt, const float* __restrict__ v,
                float w_l1_over_numel, float w_ssim_over_numel, const float* __restrict__ m1,
                const float* __restrict__ m2, const float* __restrict__ m3, float* __restrict__ v_img) {
  __shared__ float sm[3][LH][LH + 1];
  __shared__ float hz[3][LH][LT + 1];
  // XCD-aware tile order: each XCD's L2 sees a contiguous run of tiles, so the 5-pixel halos shared
  // with the neighbouring tiles are fetched from HBM once
  const int n_tx = (W + LT - 1) / LT, n_ty = (H + LT - 1) / LT;
  const int tile = (int)xcd_remap(blockIdx.x, (unsigned)(n_tx * n_ty));
  const int tyi = tile / n_tx, txi = tile - tyi * n_tx;
  const int x0 = txi * LT, y0 = tyi * LT;
  const int tid = threadIdx.x;
  const size_t plane = (size_t)H * W;
  const float vv = v[0];
  float res[3][4];  // the three channels of a pixel leave together (below): one full 12 B per pixel
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    __syncthreads();
    for (int i = tid; i < LH * LH; i += 256) {
      const int r = i / LH, cc = i - r * LH;
      const int yy = y0 + r - LR, xx = x0 + cc - LR;
      float a = 0.f, b = 0.f, d = 0.f;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const size_t o = c * plane + (size_t)yy * W + xx;
        a = m1[o]; b = m2[o]; d = m3[o];
      }
      sm[0][r][cc] = a; sm[1][r][cc] = b; sm[2][r][cc] = d;
    }
    __syncthreads();
    for (int t = tid; t < 3 * LH * (LT / 4); t += 256) {
      const int q = t / (LH * (LT / 4)), rem = t - q * (LH * (LT / 4));
      const int r = rem / (LT / 4), c4 = (rem - r * (LT / 4)) * 4;
      float a[14];
#pragma unroll
      for (int k = 0; k < 14; ++k) a[k] = sm[q][r][c4 + k];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        float sacc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) sacc += l_win[k] * a[o + k];
        hz[q][r][c4 + o] = sacc;
      }
    }
    __syncthreads();
    {
      const int col = tid & 31, r4 = (tid >> 5) * 4;
      float g[3][4];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        float vcol[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) vcol[k] = hz[q][r4 + k][col];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          float sacc = 0.f;
#pragma unroll
          for (int k = 0; k < 11; ++k) sacc += l_win[k] * vcol[o + k];
          g[q][o] = sacc;
        }
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int y = y0 + r4 + o, x = x0 + col;
        res[c][o] = 0.f;
        if (y < H && x < W) {
          const int64_t oi = c * img.sc + y * img.sy + x * img.sx;
          const float xv = img.p[oi];
          const float yv = gt_val(gt, c * plane + (size_t)y * W + x);
          const float sgn = (xv > yv) ? 1.f : ((xv < yv) ? -1.f : 0.f);
          const float dss = g[0][o] + 2.f * xv * g[1][o] + yv * g[2][o];
          res[c][o] = vv * (w_l1_over_numel * sgn - w_ssim_over_numel * dss);
        }
      }
    }
  }
  // Channel-interleaved ([H,W,3]) cotangent images: a store per channel pass touched every 64 B line
  // three times, a third of it each (WRITE_SIZE was 3x the image); back to back the three partial
  // stores of a line merge before they leave the L2.
  {
    const int col = tid & 31, r4 = (tid >> 5) * 4;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int y = y0 + r4 + o, x = x0 + col;
      if (y < H && x < W) {
        const int64_t ob = y * img.sy + x * img.sx;
#pragma unroll
        for (int c = 0; c < 3; ++c) v_img[ob + c * img.sc] = res[c][o];
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
  dim3 grid((unsigned)(ceil_div(W, LT) * ceil_div(H, LT)));
  hipLaunchKernelGGL(loss_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, v, gt_u8, v_loss,
                     (float)((1.0 - lambda_dssim) / numel), (float)(lambda_dssim / numel), m1, m2, m3,
                     v_img);
  CLMGS_LAUNCH_CHECK();
  return 0;
}
