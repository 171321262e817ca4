// Fused per-camera front end and its VJP (gfx950): for the rows of one visibility filter,
//   gather raw attributes -> activations (exp / sigmoid; the projection normalises the quaternion)
//   -> EWA projection -> SH colour -> +0.5 clamp -> packed 64 B raster record
// and, backward, from the rasterizer's packed gradient lines,
//   clamp / SH / projection / activation VJPs -> accumulate into the FULL-size gradient tensors
//   and the SH gradient rows -> densification statistics.
//
// It computes exactly what the op-by-op path computes between the row gather and the rasterizer
// (strategies/clm_offload/engine.py:650-691 forward, :703-742 + densification.py:59-102 backward)
// but in 2 kernels instead of ~35 small ones (4 gathers, 3 activations, projection, dirs, SH,
// clamp, pack | unpack, clamp', SH', dirs', projection', activation', 4 index_add, stats).
// HBM-bound: 236 B of parameters in + ~100 B of per-row outputs per visible Gaussian.
#include <stdlib.h>

#include "common.h"
#include "gs_math.h"

namespace clmgs {

// Measured and rejected for the backward (kept out of the code): requesting the SH rows before the
// projection VJP (256 VGPRs), prefetching the SH gradient rows across the SH VJP (+48 VGPRs, 0.99 vs
// 0.90 ms), 3 waves/SIMD via launch bounds (spills, 1.02 ms).
constexpr int PP_ROWS = 64;   // one wavefront owns a chunk of 64 rows; no workgroup barriers
constexpr int PP_PITCH = 52;  // floats per LDS row (see sh.hip)

struct PreArgs {
  const int64_t* filter;        // [V] row ids, or NULL (identity)
  const float* xyz;             // [N,3]
  const float* opacity_raw;     // [N]
  const float* scaling_raw;     // [N,3]
  const float* rotation_raw;    // [N,4]
  const float* sh_rows;         // [N,48] indexed by row id, or [V,48] indexed by i (sh_by_filter=0)
  int sh_by_filter;
  const int32_t* sh_index;      // != NULL: SH row of position i is sh_rows[sh_index[i]] (a staging table that
                                // holds only the rows a batch touches: host-resident mode); sh_by_filter = 1
  int packed_small;             // 1: `xyz` is the [N,12] table xyz 3 | opacity 1 | scaling 3 | rotation 4 | pad
  float viewmat[16];
  float K[9];
  float campos[3];
  int width, height, degree;
  float eps2d, near_plane, far_plane, radius_clip;
};

// The SH staging registers are 12 NAMED float4 variables, not an array: an alloca of 192 B is
// above the backend's promote-to-vector budget at this occupancy and would live in scratch.
#define CLMGS_FOR12(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)
#define CLMGS_DECL_ST(j) float4 st##j = make_float4(0.f, 0.f, 0.f, 0.f);
// (row, piece) of staging element j of this lane.  `tl` is a LAUNDERED copy of the lane id made at
// the top of every staging phase (CLMGS_LAUNDER_LANE): the index math (row, piece, LDS offset,
// 64-bit row mask, 64-bit base pointers: ~7 values x 12 elements) is loop-invariant, and LICM would
// otherwise keep all ~84 of them in VGPRs across the whole chunk loop (measured: 256 VGPRs + spills).
#define CLMGS_ELEM(j) const int e = tl + PP_ROWS * j, r = e / NF4, k = e - r * NF4;
#define CLMGS_LAUNDER_LANE int tl = t; asm volatile("" : "+v"(tl));

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }

// The 11 small attributes of row g: from four arrays (four scattered 4-16 B pieces, each costing a
// 64 B line) or from one 48 B row of the packed table (1.5 lines on average).
struct SmallRow { float m[3]; float4 q4; float s[3]; float oraw; };
template <bool PK>
__device__ __forceinline__ SmallRow load_small(const PreArgs& a, int64_t g) {
  SmallRow r;
  if (PK) {
    const float4* row = reinterpret_cast<const float4*>(a.xyz) + 3 * g;
    const float4 r0 = row[0], r1 = row[1], r2 = row[2];
    r.m[0] = r0.x; r.m[1] = r0.y; r.m[2] = r0.z; r.oraw = r0.w;
    r.s[0] = r1.x; r.s[1] = r1.y; r.s[2] = r1.z;
    r.q4 = make_float4(r1.w, r2.x, r2.y, r2.z);
  } else {
    r.m[0] = a.xyz[3 * g]; r.m[1] = a.xyz[3 * g + 1]; r.m[2] = a.xyz[3 * g + 2];
    r.q4 = *reinterpret_cast<const float4*>(a.rotation_raw + 4 * g);
    r.s[0] = a.scaling_raw[3 * g]; r.s[1] = a.scaling_raw[3 * g + 1]; r.s[2] = a.scaling_raw[3 * g + 2];
    r.oraw = a.opacity_raw[g];
  }
  return r;
}

// Camera constants live in SGPRs.  Without this, LICM hoists every camera-derived subexpression of
// the projection (limits, reciprocals, products of view-matrix entries ...) out of the chunk loop
// into ~80 long-lived VGPRs -- measured: 256 VGPRs plus scratch spills.  Laundering the SGPR copies
// inside the loop makes those subexpressions loop-variant: they are recomputed per chunk (a few
// dozen VALU) in short-lived registers.
__device__ __forceinline__ Cam launder_cam(const Cam& c) {
  Cam r = c;
#pragma unroll
  for (int i = 0; i < 9; ++i) asm volatile("" : "+s"(r.R[i]));
#pragma unroll
  for (int i = 0; i < 3; ++i) asm volatile("" : "+s"(r.t[i]));
  asm volatile("" : "+s"(r.fx)); asm volatile("" : "+s"(r.fy));
  asm volatile("" : "+s"(r.cx)); asm volatile("" : "+s"(r.cy));
  return r;
}

// LDS hand-over between the lanes of ONE wavefront: orders the compiler's memory operations and
// costs no s_waitcnt (a wave's LDS operations execute in order); unlike __syncthreads() it does
// not drain vmcnt, so gathers stay in flight across it.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Both kernels are software-pipelined around HBM latency (the row gathers are scattered 64-192 B
// pieces: latency- not bandwidth-limited at 3 waves/SIMD):
//   * each wavefront works alone on 64-row chunks (row ids travel by ds_bpermute, LDS rows are
//     wave private), so nothing ever waits on a workgroup barrier or its vmcnt(0);
//   * the row id (and radius) of the NEXT chunk is fetched while the current one computes;
//   * the lane's own small attributes are requested first, the chunk's SH rows second, so the
//     projection arithmetic runs while the SH rows are still in flight (vmcnt is in-order: the
//     wait for the attributes does not wait for the rows);
//   * all gathers are unconditional with clamped / redirected addresses (dead rows re-read row
//     0): no per-element branches, the staging registers never leave the VGPR file;
//   * backward: the current values of every read-modify-write target are requested up front
//     together with the inputs, and the SH gradient rows while the SH VJP computes.
// OVERLAP = false (no filter: most rows are culled) projects first and stages only live rows.
template <int DEG, bool OVERLAP, bool PK>
__global__ void __launch_bounds__(PP_ROWS, 3)
preprocess_fwd_kernel(int V, PreArgs a, int32_t* __restrict__ radii, float* __restrict__ means2d,
                      float* __restrict__ depths, float* __restrict__ conics,
                      float* __restrict__ colors, float* __restrict__ opacities,
                      float4* __restrict__ packed) {
  constexpr int NB = (DEG + 1) * (DEG + 1);
  constexpr int NF4 = (NB * 3 + 3) / 4;
  __shared__ __attribute__((aligned(16))) float lds[PP_ROWS * PP_PITCH];
  const Cam cam = load_cam(a.viewmat, a.K);
  const int n_chunks = (V + PP_ROWS - 1) / PP_ROWS;
  const int t = threadIdx.x;
  int chunk = blockIdx.x;
  int my_row = -1, my_sh = 0;
  if (chunk < n_chunks && chunk * PP_ROWS + t < V) {
    my_row = a.filter ? (int)a.filter[chunk * PP_ROWS + t] : chunk * PP_ROWS + t;
    my_sh = a.sh_index ? a.sh_index[chunk * PP_ROWS + t] : my_row;
  }
  for (; chunk < n_chunks; chunk += gridDim.x) {
    const int base = chunk * PP_ROWS;
    const bool mine = my_row >= 0;
    const int64_t g = mine ? my_row : 0;
    const int sh_id = mine ? my_sh : 0;  // row of the SH table (= g unless a staging index is given)
    const SmallRow sr = load_small<PK>(a, g);
    const float m[3] = {sr.m[0], sr.m[1], sr.m[2]};
    const float4 q4 = sr.q4;
    const float s[3] = {sr.s[0], sr.s[1], sr.s[2]};
    const float oraw = sr.oraw;
    CLMGS_FOR12(CLMGS_DECL_ST)
    if (OVERLAP) {
#define CLMGS_X(j)                                                                                  \
  if constexpr (j < NF4) {                                                                          \
    CLMGS_ELEM(j)                                                                                   \
    const int64_t src = a.sh_by_filter ? (int64_t)__shfl(sh_id, r) : (int64_t)min(base + r, V - 1); \
    st##j = *reinterpret_cast<const float4*>(a.sh_rows + src * 48 + 4 * k);                         \
  }
      { CLMGS_LAUNDER_LANE
      CLMGS_FOR12(CLMGS_X)
      }
#undef CLMGS_X
    }
    {  // row id of this wave's next chunk
      const int ni = (chunk + (int)gridDim.x) * PP_ROWS + t;
      my_row = -1;
      if (ni < V) { my_row = a.filter ? (int)a.filter[ni] : ni; my_sh = a.sh_index ? a.sh_index[ni] : my_row; }
    }
    Proj p;
    p.radius = 0; p.mx = p.my = p.depth = p.ca = p.cb = p.cc = 0.f;
    float o = 0.f;
    if (mine) {
      const float q[4] = {q4.x, q4.y, q4.z, q4.w};
      const float se[3] = {__expf(s[0]), __expf(s[1]), __expf(s[2])};
      o = sigmoidf(oraw);
      p = project_fwd(cam, m, q, se, (float)a.width, (float)a.height, a.eps2d, a.near_plane,
                      a.far_plane, a.radius_clip);
    }
    wave_lds_sync();  // the previous chunk's LDS rows are consumed
    if (OVERLAP) {
#define CLMGS_X(j)                                                                                  \
  if constexpr (j < NF4) {                                                                          \
    CLMGS_ELEM(j)                                                                                   \
    *reinterpret_cast<float4*>(lds + r * PP_PITCH + 4 * k) = st##j;                                 \
  }
      { CLMGS_LAUNDER_LANE
      CLMGS_FOR12(CLMGS_X)
      }
#undef CLMGS_X
    } else {
      const unsigned long long live = __ballot(p.radius > 0);
      const int first = live ? (int)__builtin_ctzll(live) : 0;
      if (live) {
#pragma unroll
        for (int j = 0; j < NF4; ++j) {
          const int e = t + PP_ROWS * j, r = e / NF4, k = e - r * NF4;
          const int rr = ((live >> r) & 1ull) ? r : first;  // dead rows re-read a live one (cache hit)
          const int64_t src = a.sh_by_filter ? (int64_t)__shfl(sh_id, rr) : (int64_t)(base + rr);
          *reinterpret_cast<float4*>(lds + r * PP_PITCH + 4 * k) =
              *reinterpret_cast<const float4*>(a.sh_rows + src * 48 + 4 * k);
        }
      }
    }
    wave_lds_sync();
    if (mine) {
      const int i = base + t;
      float cr = 0.f, cg = 0.f, cb = 0.f;
      if (p.radius > 0) {
        float x = m[0] - a.campos[0], y = m[1] - a.campos[1], z = m[2] - a.campos[2];
        const float inv = 1.0f / sqrtf(x * x + y * y + z * z);
        x *= inv; y *= inv; z *= inv;
        float B[16];
        sh_basis(DEG, x, y, z, B);
        const float* row = lds + t * PP_PITCH;
#pragma unroll
        for (int k = 0; k < NB; ++k) { cr += B[k] * row[3 * k]; cg += B[k] * row[3 * k + 1]; cb += B[k] * row[3 * k + 2]; }
        cr = fmaxf(cr + 0.5f, 0.f); cg = fmaxf(cg + 0.5f, 0.f); cb = fmaxf(cb + 0.5f, 0.f);
      }
      radii[i] = p.radius;
      *reinterpret_cast<float2*>(means2d + 2 * (size_t)i) = make_float2(p.mx, p.my);
      depths[i] = p.depth;
      if (conics) {  // optional: the packed record carries the same values for the tile kernels
        conics[3 * (size_t)i] = p.ca; conics[3 * (size_t)i + 1] = p.cb; conics[3 * (size_t)i + 2] = p.cc;
        colors[3 * (size_t)i] = cr; colors[3 * (size_t)i + 1] = cg; colors[3 * (size_t)i + 2] = cb;
        opacities[i] = o;
      }
      float4* rec = packed + 4 * (size_t)i;
      rec[0] = make_float4(p.mx, p.my, o, p.ca);
      rec[1] = make_float4(p.cb, p.cc, cr, cg);
      rec[2] = make_float4(cb, 0.f, 0.f, 0.f);
    }
  }
}

struct PreGrads {
  float* g_xyz;          // [N,3]   accumulated at row ids
  float* g_opacity;      // [N]
  float* g_scaling;      // [N,3]
  float* g_rotation;     // [N,4]
  float* g_sh_rows;      // [N,48] (sh_by_filter) or [V,48]: accumulated
  float* max_radii2D;    // [N] or NULL (no statistics)
  float* grad_accum;     // [N]
  float* denom;          // [N]
  int packed_grads;      // 1: g_xyz is the [N,12] gradient table laid out like the packed parameters
  int packed_stats;      // 1: max_radii2D is a [N,4] table  max radius | grad accum | count | pad
  float* v_means2d_out;  // [V,2] or NULL: copy of the screen-space gradient (API parity)
  int stats_only_visible;  // 1: statistics only for rows with radius > 0 (no_offload semantics)
  // != NULL: the rasterize backward left one partial-gradient line (PART_F4 float4 = 64 B) per (row, tile) intersection;
  // row i owns lines [row_cum[i-1], row_cum[i]) and its gradient line is their sum in ascending order
  // (this kernel then takes the place of the row-sum pass: no [V,16] gradient table round trip)
  const float4* partials;
  const int64_t* row_cum;
  // != NULL (with sh_by_filter): FIRST-TOUCH stores into the [N,48] gradient table.  sh_stamp[row] is the
  // optimizer step whose gradient the row holds; a row whose stamp is not cur_step is STORED (its old
  // content was consumed by the deferred row optimizer, which then need not clear it) and stamped,
  // later cameras of the batch accumulate.  Saves the consumer's clearing write and the first read.
  int32_t* sh_stamp;
  int cur_step;
};

template <int DEG, bool PK>
__global__ void __launch_bounds__(PP_ROWS, 2)
preprocess_bwd_kernel(int V, PreArgs a, const int32_t* __restrict__ radii,
                      const float4* __restrict__ packed_grad, PreGrads o) {
  constexpr int NB = (DEG + 1) * (DEG + 1);
  constexpr int NF4 = (NB * 3 + 3) / 4;
  __shared__ __attribute__((aligned(16))) float lds[PP_ROWS * PP_PITCH];
  const Cam cam = load_cam(a.viewmat, a.K);
  const int n_chunks = (V + PP_ROWS - 1) / PP_ROWS;
  const int t = threadIdx.x;
  int chunk = blockIdx.x;
  int my_row = -1, my_radius = 0, my_sh = 0, my_stamp = 0;
  int64_t my_s0 = 0;
  int my_cnt = 0;
  if (chunk < n_chunks && chunk * PP_ROWS + t < V) {
    const int i = chunk * PP_ROWS + t;
    my_row = a.filter ? (int)a.filter[i] : i;
    my_sh = a.sh_index ? a.sh_index[i] : my_row;
    my_radius = radii[i];
    if (o.sh_stamp) my_stamp = o.sh_stamp[my_sh];
    if (o.partials) { my_s0 = i ? o.row_cum[i - 1] : 0; my_cnt = (int)(o.row_cum[i] - my_s0); }
  }
  for (; chunk < n_chunks; chunk += gridDim.x) {
    const int base = chunk * PP_ROWS;
    const int radius = my_radius;
    const int64_t s0 = my_s0;
    const int cnt = my_cnt;
    const bool mine = my_row >= 0;
    const bool vis = mine && radius > 0;
    const bool stat = mine && o.max_radii2D && (radius > 0 || !o.stats_only_visible);
    const unsigned long long live = __ballot(vis);
    const int first = live ? (int)__builtin_ctzll(live) : 0;
    // dead lanes redirect every gather to the chunk's first live row: no branches, cache hits
    const int64_t g = mine ? my_row : 0;
    const int sh_id = mine ? my_sh : 0;
    // first-touch mode: is this row's gradient line still the one of an earlier (consumed) step?
    const bool fresh = o.sh_stamp && vis && (my_stamp != o.cur_step);
    const unsigned long long fresh_rows = __ballot(fresh);
    const int64_t gl = vis ? g : (int64_t)__shfl((int)g, first);
    const int i = mine ? base + t : base;
    // ---- everything this lane needs from HBM, requested at once
    float4 ga, gb;   // line: x y ca cb | cc r g b | o
    float go;
    float4 pa[4], pb[4];
    float po[4];
    if (o.partials) {  // kernel-uniform.  The row's first four partial lines are requested here, with
      // everything else the lane needs; longer ranges (rare: 2.6 lines per row on average) follow below
      ga = make_float4(0.f, 0.f, 0.f, 0.f); gb = ga; go = 0.f;
      const float4* src = o.partials + PART_F4 * (size_t)s0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        pa[u] = ga; pb[u] = ga; po[u] = 0.f;
        if (mine && u < cnt) { pa[u] = src[PART_F4 * u]; pb[u] = src[PART_F4 * u + 1]; po[u] = src[PART_F4 * u + 2].x; }
      }
    } else {
      ga = packed_grad[4 * (size_t)i];
      gb = packed_grad[4 * (size_t)i + 1];
      go = packed_grad[4 * (size_t)i + 2].x;
    }
    const SmallRow sr = load_small<PK>(a, gl);
    const float m[3] = {sr.m[0], sr.m[1], sr.m[2]};
    const float4 q4 = sr.q4;
    const float s[3] = {sr.s[0], sr.s[1], sr.s[2]};
    const float oraw = sr.oraw;
    float c_xyz[3], c_sc[3], c_op;
    float4 c_rot;
    if (PK) {  // the row's 11 accumulated gradients: one 48 B row (first touch of the step: zeros, no read)
      const float4* grow = reinterpret_cast<const float4*>(o.g_xyz) + 3 * gl;
      float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
      if (!fresh) { g0 = grow[0]; g1 = grow[1]; g2 = grow[2]; }
      c_xyz[0] = g0.x; c_xyz[1] = g0.y; c_xyz[2] = g0.z; c_op = g0.w;
      c_sc[0] = g1.x; c_sc[1] = g1.y; c_sc[2] = g1.z;
      c_rot = make_float4(g1.w, g2.x, g2.y, g2.z);
    } else {
      c_xyz[0] = o.g_xyz[3 * gl]; c_xyz[1] = o.g_xyz[3 * gl + 1]; c_xyz[2] = o.g_xyz[3 * gl + 2];
      c_sc[0] = o.g_scaling[3 * gl]; c_sc[1] = o.g_scaling[3 * gl + 1]; c_sc[2] = o.g_scaling[3 * gl + 2];
      c_rot = *reinterpret_cast<const float4*>(o.g_rotation + 4 * gl);
      c_op = o.g_opacity[gl];
    }
    float c_mr = 0.f, c_ga = 0.f, c_dn = 0.f;
    if (o.max_radii2D) {
      if (o.packed_stats) {  // one 16 B row instead of three 4 B pieces (= three 64 B lines)
        const float4 st4 = reinterpret_cast<const float4*>(o.max_radii2D)[g];
        c_mr = st4.x; c_ga = st4.y; c_dn = st4.z;
      } else {
        c_mr = o.max_radii2D[g]; c_ga = o.grad_accum[g]; c_dn = o.denom[g];
      }
    }
    CLMGS_FOR12(CLMGS_DECL_ST)
#define CLMGS_LOAD_SH(j)                                                                            \
  if constexpr (j < NF4) {                                                                          \
    CLMGS_ELEM(j)                                                                                   \
    const int rr = ((live >> r) & 1ull) ? r : first;                                                \
    const int64_t src = a.sh_by_filter ? (int64_t)__shfl(sh_id, rr) : (int64_t)(base + rr);         \
    st##j = *reinterpret_cast<const float4*>(a.sh_rows + src * 48 + 4 * k);                         \
  }
    {
      const int ni = (chunk + (int)gridDim.x) * PP_ROWS + t;
      my_row = -1; my_radius = 0; my_s0 = 0; my_cnt = 0;
      if (ni < V) {
        my_row = a.filter ? (int)a.filter[ni] : ni; my_radius = radii[ni];
        my_sh = a.sh_index ? a.sh_index[ni] : my_row;
        if (o.sh_stamp) my_stamp = o.sh_stamp[my_sh];  // next chunk's stamps travel with its row ids
        if (o.partials) { my_s0 = ni ? o.row_cum[ni - 1] : 0; my_cnt = (int)(o.row_cum[ni] - my_s0); }
      }
    }
    if (o.partials) {
      // ascending slot order = the order the row-sum kernel used -> bit-identical gradients.  Neighbouring
      // lanes own neighbouring ranges: the wave streams one contiguous stretch of the buffer.
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // lines beyond cnt were loaded as zeros
        ga.x += pa[u].x; ga.y += pa[u].y; ga.z += pa[u].z; ga.w += pa[u].w;
        gb.x += pb[u].x; gb.y += pb[u].y; gb.z += pb[u].z; gb.w += pb[u].w;
        go += po[u];
      }
      if (mine) {
        const float4* src = o.partials + PART_F4 * (size_t)s0;
        for (int tt = 4; tt < cnt; tt += 4) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int l = min(tt + u, cnt - 1);
            pa[u] = src[PART_F4 * l]; pb[u] = src[PART_F4 * l + 1]; po[u] = src[PART_F4 * l + 2].x;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (tt + u < cnt) {
              ga.x += pa[u].x; ga.y += pa[u].y; ga.z += pa[u].z; ga.w += pa[u].w;
              gb.x += pb[u].x; gb.y += pb[u].y; gb.z += pb[u].z; gb.w += pb[u].w;
              go += po[u];
            }
          }
        }
      }
    }
    // ---- statistics + projection VJP while the SH rows are in flight
    const float v_m2[2] = {ga.x, ga.y};
    if (mine && o.v_means2d_out) *reinterpret_cast<float2*>(o.v_means2d_out + 2 * (size_t)i) = make_float2(ga.x, ga.y);
    if (stat) {  // default: every filter row, as gsplat_add_densification_stats_exact_filter;
      // only_visible = the no_offload mask form
      const float gx = v_m2[0] * (0.5f * a.width), gy = v_m2[1] * (0.5f * a.height);
      if (o.packed_stats) {
        reinterpret_cast<float4*>(o.max_radii2D)[g] =
            make_float4(fmaxf(c_mr, (float)radius), c_ga + sqrtf(gx * gx + gy * gy), c_dn + 1.f, 0.f);
      } else {
        o.max_radii2D[g] = fmaxf(c_mr, (float)radius);
        o.grad_accum[g] = c_ga + sqrtf(gx * gx + gy * gy);
        o.denom[g] = c_dn + 1.f;
      }
    }
    float vm[3] = {0.f, 0.f, 0.f};
    float n_sc[3] = {0.f, 0.f, 0.f}, n_op = 0.f;
    float4 n_rot = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vis) {
      const float q[4] = {q4.x, q4.y, q4.z, q4.w};
      const float se[3] = {__expf(s[0]), __expf(s[1]), __expf(s[2])};
      const float op = sigmoidf(oraw);
      const float v_con[3] = {ga.z, ga.w, gb.x};
      float vq[4], vs[3];
      const Cam camL = launder_cam(cam);
      project_bwd(camL, m, q, se, (float)a.width, (float)a.height, a.eps2d, v_m2, 0.f, v_con, vm, vq, vs);
      n_sc[0] = c_sc[0] + vs[0] * se[0]; n_sc[1] = c_sc[1] + vs[1] * se[1]; n_sc[2] = c_sc[2] + vs[2] * se[2];
      n_rot = make_float4(c_rot.x + vq[0], c_rot.y + vq[1], c_rot.z + vq[2], c_rot.w + vq[3]);
      n_op = c_op + go * op * (1.f - op);
      if (!PK) {
        o.g_scaling[3 * g] = n_sc[0]; o.g_scaling[3 * g + 1] = n_sc[1]; o.g_scaling[3 * g + 2] = n_sc[2];
        *reinterpret_cast<float4*>(o.g_rotation + 4 * g) = n_rot;
        o.g_opacity[g] = n_op;
      }
    }
    if (!live) continue;  // wave-uniform: nothing of this chunk reached the screen
    // ---- SH rows -> LDS; their registers then prefetch the gradient rows to accumulate into
    { CLMGS_LAUNDER_LANE CLMGS_FOR12(CLMGS_LOAD_SH) }  // SH rows are requested only now (register-lean)
#undef CLMGS_LOAD_SH
    wave_lds_sync();  // the previous chunk's LDS rows are consumed
#define CLMGS_X(j)                                                                                  \
  if constexpr (j < NF4) {                                                                          \
    CLMGS_ELEM(j)                                                                                   \
    *reinterpret_cast<float4*>(lds + r * PP_PITCH + 4 * k) = st##j;                                 \
  }
    { CLMGS_LAUNDER_LANE
    CLMGS_FOR12(CLMGS_X)
    }
#undef CLMGS_X
    wave_lds_sync();
    if (vis) {
      float* row = lds + t * PP_PITCH;
      // recompute the pre-clamp colour for the clamp mask, then the VJP
      float dx = m[0] - a.campos[0], dy = m[1] - a.campos[1], dz = m[2] - a.campos[2];
      const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
      const float x = dx * inv, y = dy * inv, z = dz * inv;
      float B[16];
      sh_basis(DEG, x, y, z, B);
      // ONE pass over the row, band by band: colour P_c = sum_k B_k r_kc and, per channel, the
      // un-clamped direction cotangent U_c = sum_k r_kc grad B_k; the clamp mask is applied to
      // the three U_c afterwards.  Each coefficient is consumed as soon as it is read (the
      // two-pass form kept all 48 row values + 64 basis values live: 256 VGPRs and spills).
      float P[3] = {0.f, 0.f, 0.f};
      float U[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
      {
        float Bx[16], By[16], Bz[16];
        if (DEG > 0) sh_basis_grad(DEG, x, y, z, Bx, By, Bz);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          if (k == 1 || k == 4 || k == 9) __builtin_amdgcn_sched_barrier(0);  // band boundary
          const float rk[3] = {row[3 * k], row[3 * k + 1], row[3 * k + 2]};
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            P[c] += B[k] * rk[c];
            if (DEG > 0 && k > 0) { U[c][0] += rk[c] * Bx[k]; U[c][1] += rk[c] * By[k]; U[c][2] += rk[c] * Bz[k]; }
          }
        }
      }
      const float vc[3] = {(P[0] + 0.5f > 0.f) ? gb.y : 0.f, (P[1] + 0.5f > 0.f) ? gb.z : 0.f,
                           (P[2] + 0.5f > 0.f) ? gb.w : 0.f};
      float vdx = 0.f, vdy = 0.f, vdz = 0.f;
      if (DEG > 0) {
        const float ux = vc[0] * U[0][0] + vc[1] * U[1][0] + vc[2] * U[2][0];
        const float uy = vc[0] * U[0][1] + vc[1] * U[1][1] + vc[2] * U[2][1];
        const float uz = vc[0] * U[0][2] + vc[1] * U[1][2] + vc[2] * U[2][2];
        const float dot = ux * x + uy * y + uz * z;
        vdx = (ux - dot * x) * inv; vdy = (uy - dot * y) * inv; vdz = (uz - dot * z) * inv;
      }
#pragma unroll
      for (int k = 0; k < 4 * NF4 / 3 + 1 && k < 16; ++k) {  // this lane's LDS row becomes its SH gradient row
        const float bk = (k < NB) ? B[k] : 0.f;
        row[3 * k] = bk * vc[0]; row[3 * k + 1] = bk * vc[1]; row[3 * k + 2] = bk * vc[2];
      }
      if (PK) {
        float4* grow = reinterpret_cast<float4*>(o.g_xyz) + 3 * g;
        grow[0] = make_float4(c_xyz[0] + vm[0] + vdx, c_xyz[1] + vm[1] + vdy, c_xyz[2] + vm[2] + vdz, n_op);
        grow[1] = make_float4(n_sc[0], n_sc[1], n_sc[2], n_rot.x);
        grow[2] = make_float4(n_rot.y, n_rot.z, n_rot.w, 0.f);
      } else {
        o.g_xyz[3 * g] = c_xyz[0] + vm[0] + vdx;
        o.g_xyz[3 * g + 1] = c_xyz[1] + vm[1] + vdy;
        o.g_xyz[3 * g + 2] = c_xyz[2] + vm[2] + vdz;
      }
    }
    wave_lds_sync();
    // launder the row id: the addresses below are RE-computed here instead of 12 64-bit
    // pointers staying live across the SH VJP
    int g_l = sh_id;
    asm volatile("" : "+v"(g_l));
#define CLMGS_X(j)                                                                                  \
  if constexpr (j < NF4) {                                                                          \
    CLMGS_ELEM(j)                                                                                   \
    const int64_t dst_row = a.sh_by_filter ? (int64_t)__shfl(g_l, r) : (int64_t)(base + r);         \
    if ((live >> r) & 1ull) {                                                                       \
      const float4 v = *reinterpret_cast<const float4*>(lds + r * PP_PITCH + 4 * k);                \
      float4 c = make_float4(0.f, 0.f, 0.f, 0.f);                                                   \
      if (!((fresh_rows >> r) & 1ull))                                                              \
        c = *reinterpret_cast<const float4*>(o.g_sh_rows + dst_row * 48 + 4 * k);                   \
      c.x += v.x; c.y += v.y; c.z += v.z; c.w += v.w;                                               \
      *reinterpret_cast<float4*>(o.g_sh_rows + dst_row * 48 + 4 * k) = c;                           \
    }                                                                                               \
  }
    { CLMGS_LAUNDER_LANE
    CLMGS_FOR12(CLMGS_X)
    }
#undef CLMGS_X
    if (fresh) o.sh_stamp[sh_id] = o.cur_step;  // every row belongs to one lane of one wave of this launch
  }
}

}  // namespace clmgs

using namespace clmgs;

static void fill_args(PreArgs& a, const int64_t* filter, const float* xyz, const float* opacity_raw,
                      const float* scaling_raw, const float* rotation_raw, const float* sh_rows,
                      int sh_by_filter, const float* viewmat, const float* K, const float* campos,
                      int width, int height, int degree, float eps2d, float near_plane,
                      float far_plane, float radius_clip) {
  a.filter = filter; a.xyz = xyz; a.opacity_raw = opacity_raw; a.scaling_raw = scaling_raw;
  a.rotation_raw = rotation_raw; a.sh_rows = sh_rows; a.sh_by_filter = sh_by_filter; a.sh_index = nullptr;
  a.packed_small = (!opacity_raw && !scaling_raw && !rotation_raw) ? 1 : 0;
  for (int i = 0; i < 16; ++i) a.viewmat[i] = viewmat[i];
  for (int i = 0; i < 9; ++i) a.K[i] = K[i];
  for (int i = 0; i < 3; ++i) a.campos[i] = campos[i];
  a.width = width; a.height = height; a.degree = degree;
  a.eps2d = eps2d; a.near_plane = near_plane; a.far_plane = far_plane; a.radius_clip = radius_clip;
}

extern "C" int clmgs_preprocess_fwd(void* stream, int V, const int64_t* filter, const float* xyz,
                                    const float* opacity_raw, const float* scaling_raw,
                                    const float* rotation_raw, const float* sh_rows,
                                    int sh_by_filter, const float* viewmat_host,
                                    const float* K_host, const float* campos_host, int width,
                                    int height, int degree, float eps2d, float near_plane,
                                    float far_plane, float radius_clip, int32_t* radii,
                                    float* means2d, float* depths, float* conics, float* colors,
                                    float* opacities, void* packed, const int32_t* sh_index) {
  CLMGS_CHECK_ARG(V >= 0 && degree >= 0 && degree <= 3 && width > 0 && height > 0);  // row ids are int32 in flight
  if (V == 0) return 0;
  CLMGS_CHECK_ARG(xyz && sh_rows && viewmat_host && K_host && campos_host && radii && means2d &&
                  depths && packed);
  CLMGS_CHECK_ARG((opacity_raw && scaling_raw && rotation_raw) ||
                  (!opacity_raw && !scaling_raw && !rotation_raw && (((uintptr_t)xyz & 15) == 0)));
  CLMGS_CHECK_ARG(!conics || (colors && opacities));
  PreArgs a;
  fill_args(a, filter, xyz, opacity_raw, scaling_raw, rotation_raw, sh_rows, sh_by_filter,
            viewmat_host, K_host, campos_host, width, height, degree, eps2d, near_plane, far_plane,
            radius_clip);
  CLMGS_CHECK_ARG(!sh_index || sh_by_filter);
  a.sh_index = sh_index;
  const size_t lds = 0;
  const int grid = min(ceil_div(V, PP_ROWS), 256 * 12);
#define CLMGS_PRE_FWD(D, O)                                                                        \
  do {                                                                                             \
    if (a.packed_small)                                                                            \
      hipLaunchKernelGGL((preprocess_fwd_kernel<D, O, true>), dim3(grid), dim3(PP_ROWS), lds,      \
                         (hipStream_t)stream, V, a, radii, means2d, depths, conics, colors,        \
                         opacities, (float4*)packed);                                              \
    else                                                                                           \
      hipLaunchKernelGGL((preprocess_fwd_kernel<D, O, false>), dim3(grid), dim3(PP_ROWS), lds,     \
                         (hipStream_t)stream, V, a, radii, means2d, depths, conics, colors,        \
                         opacities, (float4*)packed);                                              \
  } while (0)
  const bool ov = filter != nullptr;  // a filter means (almost) every row is visible
  switch (degree) {
    case 0: if (ov) CLMGS_PRE_FWD(0, true); else CLMGS_PRE_FWD(0, false); break;
    case 1: if (ov) CLMGS_PRE_FWD(1, true); else CLMGS_PRE_FWD(1, false); break;
    case 2: if (ov) CLMGS_PRE_FWD(2, true); else CLMGS_PRE_FWD(2, false); break;
    default: if (ov) CLMGS_PRE_FWD(3, true); else CLMGS_PRE_FWD(3, false); break;
  }
#undef CLMGS_PRE_FWD
  CLMGS_LAUNCH_CHECK();
  return 0;
}

extern "C" int clmgs_preprocess_bwd(void* stream, int V, const int64_t* filter, const float* xyz,
                                    const float* opacity_raw, const float* scaling_raw,
                                    const float* rotation_raw, const float* sh_rows,
                                    int sh_by_filter, const float* viewmat_host,
                                    const float* K_host, const float* campos_host, int width,
                                    int height, int degree, float eps2d, const int32_t* radii,
                                    const void* packed_grad, float* g_xyz, float* g_opacity,
                                    float* g_scaling, float* g_rotation, float* g_sh_rows,
                                    float* max_radii2D, float* grad_accum, float* denom,
                                    float* v_means2d_out, int stats_only_visible,
                                    const void* partials, const int64_t* row_cum,
                                    const int32_t* sh_index, int32_t* sh_stamp, int cur_step) {
  CLMGS_CHECK_ARG(V >= 0 && degree >= 0 && degree <= 3 && width > 0 && height > 0);  // row ids are int32 in flight
  CLMGS_CHECK_ARG(!sh_stamp || (sh_by_filter && cur_step >= 1));
  if (V == 0) return 0;
  CLMGS_CHECK_ARG(xyz && sh_rows && viewmat_host && K_host && campos_host && radii && g_xyz && g_sh_rows);
  // exactly one source of the per-row raster gradient: the [V,16] table, or the partial lines + ranges
  CLMGS_CHECK_ARG((packed_grad != nullptr) != (partials != nullptr));
  CLMGS_CHECK_ARG(!partials || (row_cum && (((uintptr_t)partials & 15) == 0)));
  CLMGS_CHECK_ARG((opacity_raw && scaling_raw && rotation_raw) ||
                  (!opacity_raw && !scaling_raw && !rotation_raw && (((uintptr_t)xyz & 15) == 0)));
  const bool pg = !g_opacity && !g_scaling && !g_rotation;
  CLMGS_CHECK_ARG(pg || (g_opacity && g_scaling && g_rotation));
  // packed parameters and packed gradients go together (one kernel variant)
  CLMGS_CHECK_ARG(pg == (!opacity_raw && !scaling_raw && !rotation_raw) && (!pg || (((uintptr_t)g_xyz & 15) == 0)));
  const bool ps = max_radii2D && !grad_accum && !denom;  // [N,4] statistics table
  CLMGS_CHECK_ARG(!max_radii2D || ps || (grad_accum && denom));
  CLMGS_CHECK_ARG(!ps || (((uintptr_t)max_radii2D & 15) == 0));
  PreArgs a;
  fill_args(a, filter, xyz, opacity_raw, scaling_raw, rotation_raw, sh_rows, sh_by_filter,
            viewmat_host, K_host, campos_host, width, height, degree, eps2d, 0.f, 0.f, 0.f);
  CLMGS_CHECK_ARG(!sh_index || sh_by_filter);
  a.sh_index = sh_index;
  PreGrads o{g_xyz, g_opacity, g_scaling, g_rotation, g_sh_rows, max_radii2D, grad_accum, denom,
             pg ? 1 : 0, ps ? 1 : 0, v_means2d_out, stats_only_visible, (const float4*)partials, row_cum,
             sh_stamp, cur_step};
  const size_t lds = 0;
  const int grid = min(ceil_div(V, PP_ROWS), 256 * 12);
#define CLMGS_PRE_BWD(D)                                                                          \
  do {                                                                                             \
    if (pg)                                                                                        \
      hipLaunchKernelGGL((preprocess_bwd_kernel<D, true>), dim3(grid), dim3(PP_ROWS), lds,         \
                         (hipStream_t)stream, V, a, radii, (const float4*)packed_grad, o);         \
    else                                                                                           \
      hipLaunchKernelGGL((preprocess_bwd_kernel<D, false>), dim3(grid), dim3(PP_ROWS), lds,        \
                         (hipStream_t)stream, V, a, radii, (const float4*)packed_grad, o);         \
  } while (0)
  switch (degree) {
    case 0: CLMGS_PRE_BWD(0); break;
    case 1: CLMGS_PRE_BWD(1); break;
    case 2: CLMGS_PRE_BWD(2); break;
    default: CLMGS_PRE_BWD(3); break;
  }
#undef CLMGS_PRE_BWD
  CLMGS_LAUNCH_CHECK();
  return 0;
}
