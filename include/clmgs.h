/* clmgs.h -- C ABI of libclmgs_hip.so, the MI355X (gfx950) replacement for the
 * native operators the CLM-GS engines call.
 *
 * Conventions:
 *   - plain C types; every buffer is CALLER-ALLOCATED (device memory unless the
 *     name says host/pinned) and passed as raw pointer + extents;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return 0 on success, non-zero = hipError_t or a CLMGS_E* code; the message
 *     is available (thread-local) from clmgs_last_error();
 *   - no allocation inside, except explicit temp-storage queries: calling a
 *     `*_temp_bytes` function tells the caller how much scratch to pass;
 *   - re-entrant from two host threads on different streams (render thread and
 *     optimizer thread); nothing here holds the Python GIL.
 *
 * Each entry point cites the reference interface it replaces
 * (file:line under the reference tree).  The kernels themselves are absent
 * from the reference (empty submodules), see oracle/gs_oracle.py's header.
 */
#ifndef CLMGS_H
#define CLMGS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLMGS_EINVAL 10001
#define CLMGS_ENOMEM 10002

int clmgs_version(void);
const char* clmgs_last_error(void);

/* ---- gsplat.fully_fused_projection  (strategies/base_engine.py:36-47,139-151;
 *      strategies/no_offload/engine.py:49-60; strategies/clm_offload/engine.py:51-63)
 * means[N,3] quats[N,4] scales[N,3] viewmats[C,4,4] Ks[C,3,3] ->
 * radii[C,N] i32 (0 = culled), means2d[C,N,2], depths[C,N], conics[C,N,3].
 * Any of means2d/depths/conics may be NULL (visibility-only pass). */
int clmgs_projection_fwd(void* stream, int C, int N, const float* means, const float* quats,
                         const float* scales, const float* viewmats, const float* Ks, int width,
                         int height, float eps2d, float near_plane, float far_plane,
                         float radius_clip, int32_t* radii, float* means2d, float* depths,
                         float* conics);
/* The same cull from the RAW parameters (quaternions un-normalised, scales as logs), radii[C,N]
 * only: what calculate_filters (strategies/base_engine.py:18-76) needs, reading each Gaussian
 * once for all C cameras and skipping the exp / normalize passes over N. */
int clmgs_visibility_raw(void* stream, int C, int N, const float* means, const float* quats_raw,
                         const float* log_scales, const float* viewmats, const float* Ks, int width,
                         int height, float eps2d, float near_plane, float far_plane,
                         float radius_clip, int32_t* radii);

/* Visibility filters of a batch selected on the GPU (calculate_filters, base_engine.py:18-76, plus the
 * union "rows the batch touches"), same cull as clmgs_visibility_raw but nothing of size C*N leaves
 * the kernel: _count writes one ballot word + popcount per (camera, 64 Gaussians) into `temp`, scans
 * them and leaves cum_totals[C+1] (device, i64: set bits of cameras 0..r; row C = union over the
 * cameras); the caller reads cum_totals, allocates out[cum_totals[C]] i64 and calls _emit, which
 * writes the ascending Gaussian indices of camera 0, camera 1, ..., then of the union.  1 <= C <= 64
 * (the engines' largest batch); every pair first takes a conservative screen test, only the
 * survivors the exact projection -- the selected sets are the exact test's. */
size_t clmgs_visibility_select_temp_bytes(int C, int N);
int clmgs_visibility_select_count(void* stream, int C, int N, const float* means,
                                  const float* quats_raw, const float* log_scales,
                                  const float* viewmats, const float* Ks, int width, int height,
                                  float eps2d, float near_plane, float far_plane, float radius_clip,
                                  void* temp, size_t temp_bytes, int64_t* cum_totals);
/* The same with what the caller already knows about blocks of 256 consecutive rows: block_flags[ceil(N/256)], 0 = no row
 * of the block passes the cull in any camera (clmgs_adam_small_deferred computes such flags with a superset test).  The
 * flagged-off blocks are not read; the result is the unflagged call's, bit for bit.  NULL: as above. */
int clmgs_visibility_select_count_blocks(void* stream, int C, int N, const float* means, const float* quats_raw,
                                         const float* log_scales, const float* viewmats, const float* Ks, int width,
                                         int height, float eps2d, float near_plane, float far_plane, float radius_clip,
                                         void* temp, size_t temp_bytes, int64_t* cum_totals,
                                         const uint8_t* block_flags);
int clmgs_visibility_select_emit(void* stream, int C, int N, const void* temp, int64_t* out);
/* VJP of the above.  v_means[N,3], v_quats[N,4], v_scales[N,3] are overwritten
 * with the sum over the C cameras. */
int clmgs_projection_bwd(void* stream, int C, int N, const float* means, const float* quats,
                         const float* scales, const float* viewmats, const float* Ks, int width,
                         int height, float eps2d, const int32_t* radii, const float* v_means2d,
                         const float* v_depths, const float* v_conics, float* v_means,
                         float* v_quats, float* v_scales);

/* ---- gsplat.spherical_harmonics  (base_engine.py:161-163; no_offload/engine.py:67-69;
 *      clm_offload/engine.py:73-76)
 * dirs[n,3] (un-normalised), coeffs[n,16,3], masks[n] u8 or NULL -> colors[n,3]. */
int clmgs_sh_fwd(void* stream, int n, int degree, const float* dirs, const float* coeffs,
                 const uint8_t* masks, float* colors);
/* v_coeffs[n,16,3]: accumulate != 0 adds into the buffer (this is
 * clm_kernels.spherical_harmonics_bwd_inplace, clm_offload/engine.py:709-716),
 * else overwrites (rows beyond the active degree = 0).  v_dirs[n,3] may be NULL. */
int clmgs_sh_bwd(void* stream, int n, int degree, const float* dirs, const float* coeffs,
                 const uint8_t* masks, const float* v_colors, float* v_coeffs, int accumulate,
                 float* v_dirs);

/* ---- gsplat.isect_tiles / isect_offset_encode  (base_engine.py:175-186)
 * Phase 1: tiles_per_gauss[C*N] i32 and its inclusive prefix sum cum[C*N] i64.
 * The caller reads cum[C*N-1] (= I) and allocates isect buffers. */
size_t clmgs_isect_count_temp_bytes(int CN);
int clmgs_isect_count(void* stream, int C, int N, const float* means2d, const int32_t* radii,
                      int tile_size, int tile_width, int tile_height, int32_t* tiles_per_gauss,
                      int64_t* cum, void* temp, size_t temp_bytes);
/* Phase 2: emit + sort.  isect_ids[I] i64 = (cam << tile_bits | tile) << 32 | depth bits,
 * flatten_ids[I] i32 = cam*N + gaussian, both sorted by key (stable). */
size_t clmgs_isect_sort_temp_bytes(int64_t n_isects);
int clmgs_isect_emit_sort(void* stream, int C, int N, int64_t n_isects, const float* means2d,
                          const int32_t* radii, const float* depths, const int64_t* cum,
                          int tile_size, int tile_width, int tile_height, int64_t* isect_ids,
                          int32_t* flatten_ids, void* temp, size_t temp_bytes);
/* offsets[C*tile_h*tile_w] i32 = first sorted index of each (cam,tile). */
int clmgs_isect_offsets(void* stream, int64_t n_isects, const int64_t* isect_ids, int C,
                        int tile_width, int tile_height, int32_t* offsets);

/* ---- two-level binning (engine fast path; single camera).  Same (tile, depth, row) order as
 * clmgs_isect_emit_sort, ~1/4 of its traffic: A = stable depth sort of the V rows + tile counts
 * in that order (order[V] i32, cum[V] i64 inclusive; caller reads cum[V-1] = I);
 * B = emit in depth order + ONE stable sort on the tile-id bits -> flatten_ids[I] i32 (row ids),
 * offsets[tile_w*tile_h] i32, and isect_ids[I] i64 if non-NULL.
 * row_cum[V] (i64, optional, from order_count): inclusive counts of the EMITTED intersections in ROW
 * order; with it emit_sort also returns emit_slot[I] (i32): the slot of every sorted intersection,
 * row i owning the contiguous slots [row_cum[i-1], row_cum[i]) -- hand emit_slot to
 * clmgs_rasterize_bwd (which STORES one partial-gradient line per slot) and row_cum to whoever sums the
 * rows (clmgs_preprocess_bwd, or clmgs_rasterize_bwd's own row-sum pass): atomic-free, deterministic. */
size_t clmgs_isect2_order_temp_bytes(int V);
int clmgs_isect2_order_count(void* stream, int V, const float* means2d, const int32_t* radii,
                             const float* depths, int tile_size, int tile_width, int tile_height,
                             const void* packed, int32_t* order, int64_t* cum, uint64_t* boxes,
                             int64_t* totals, void* temp, size_t temp_bytes, int64_t* row_cum);
size_t clmgs_isect2_sort_temp_bytes(int64_t n_isects);
int clmgs_isect2_emit_sort(void* stream, int V, int64_t n_isects, const float* depths,
                           const int32_t* order, const int64_t* cum, const uint64_t* boxes,
                           int tile_width, int tile_height, int32_t* flatten_ids, int32_t* offsets,
                           int64_t* isect_ids, int32_t* emit_slot, void* temp, size_t temp_bytes,
                           const int64_t* row_cum);

/* ---- gsplat.rasterize_to_pixels  (base_engine.py:192-203)
 * means2d[C*N,2] conics[C*N,3] colors[C*N,3] opacities[C*N], backgrounds[C,3] or NULL ->
 * render_colors[C,H,W,3], render_alphas[C,H,W], last_ids[C,H,W] i32.  tile_size must be 16.
 * `packed` is caller scratch of clmgs_rasterize_pack_bytes(C,N) bytes (64 B aligned): fwd fills it
 * with one 64 B raster record per Gaussian and the tile kernels gather that one line; keep it for
 * the backward call.  means2d == NULL: `packed` already holds the records (clmgs_preprocess_fwd). */
size_t clmgs_rasterize_pack_bytes(int C, int N);
int clmgs_rasterize_fwd(void* stream, int C, int N, int64_t n_isects, const float* means2d,
                        const float* conics, const float* colors, const float* opacities,
                        const float* backgrounds, int width, int height, int tile_size,
                        int tile_width, int tile_height, const int32_t* offsets,
                        const int32_t* flatten_ids, void* packed, float* render_colors,
                        float* render_alphas, int32_t* last_ids);
/* VJP.  packed = the forward's record buffer; packed_grad = scratch of the same size, one 64 B
 * gradient line per Gaussian (x y ca cb | cc r g b | o).  Two accumulation modes:
 *   - emit_slot == NULL: packed_grad is zeroed here and the per-(Gaussian,tile) sums are added
 *     with float atomics (any flatten_ids / offsets, C >= 1);
 *   - emit_slot from clmgs_isect2_emit_sort (C == 1) + `partials` scratch of
 *     clmgs_rasterize_partials_bytes(n_isects) bytes (64 B lines, 16 B aligned): every (Gaussian,tile) sum is
 *     STORED at its slot (every slot is written exactly once, zeros included) -- no atomics (they
 *     bound the kernel: 4.05 -> ~2 ms at 12 M intersections), bitwise reproducible gradients.  With
 *     packed_grad != NULL (+ row_cum from clmgs_isect2_order_count) a row-order pass then adds each row's
 *     contiguous slot range into packed_grad; with packed_grad == NULL the partial lines are the
 *     result and clmgs_preprocess_bwd(partials, row_cum) sums them on the fly (engine path).
 * v_means2d[C*N,2] v_conics[C*N,3] v_colors[C*N,3] v_opacities[C*N] are OVERWRITTEN;
 * v_means2d == NULL skips the unpack (the caller reads packed_grad / the partial lines). */
size_t clmgs_rasterize_partials_bytes(int64_t n_isects);
int clmgs_rasterize_bwd(void* stream, int C, int N, int64_t n_isects, const void* packed,
                        const float* backgrounds, int width, int height, int tile_size,
                        int tile_width, int tile_height, const int32_t* offsets,
                        const int32_t* flatten_ids, const float* render_alphas,
                        const int32_t* last_ids, const float* v_render_colors,
                        const float* v_render_alphas, void* packed_grad, float* v_means2d,
                        float* v_conics, float* v_colors, float* v_opacities,
                        const int32_t* emit_slot, const int64_t* row_cum, void* partials);

/* ---- device-count forms (engine fast path): the data-dependent size I of a camera need not reach the host
 * before the consumers of the list are enqueued.  The reference reads it back synchronously
 * (strategies/base_engine.py:64-69 `counts.cpu()`, gsplat.isect_tiles' cum[-1].item()); here `capacity` (a
 * prediction by the caller, >= the true count or the camera is redone) sizes every buffer and launch, and
 * the TRUE count is read ON THE DEVICE from n_isects_dev = totals[0] of clmgs_isect2_order_count.
 * Intersections beyond the capacity are dropped (never written out of bounds); the caller compares the count
 * with the capacity from its asynchronous readback and, if it was exceeded, repeats the camera with the exact
 * forms above before anything was accumulated.  Same results as the exact forms when count <= capacity
 * (tests/test_gpu_ops.py).  rasterize_*_dev: slot mode only (C == 1, records in `packed`, partial lines out). */
int clmgs_isect2_emit_sort_dev(void* stream, int V, int64_t capacity, const int64_t* n_isects_dev,
                               const float* depths, const int32_t* order, const int64_t* cum,
                               const uint64_t* boxes, int tile_width, int tile_height,
                               int32_t* flatten_ids, int32_t* offsets, int64_t* isect_ids,
                               int32_t* emit_slot, void* temp, size_t temp_bytes, const int64_t* row_cum);

/* Tile-major binning (round 5; the engine's default route): the same lists as clmgs_isect2_order_count +
 * clmgs_isect2_emit_sort -- gsplat.isect_tiles + isect_offset_encode of strategies/base_engine.py:175-186, sorted by
 * (tile, depth bits, row index) -- built without a global sort: per-tile counters, an exclusive scan (= offsets), a
 * scatter of 16 B records into the tiles' segments and one LDS sort per tile (csrc/isect3.hip).  8 kernel launches and one
 * clear instead of 23.
 *  - clmgs_isect3_front: per-row tile boxes / exact tile masks (packed != NULL), tile counters, row_cum[V] (inclusive
 *    emitted counts in ROW order = the slot ranges of clmgs_rasterize_bwd's slot mode), totals[2] on the device =
 *    {intersections to emit, un-culled count}.  `temp` (clmgs_isect3_front_temp_bytes) must stay alive until _bin ran.
 *  - clmgs_isect3_bin: offsets[tile_w*tile_h], flatten_ids[I], emit_slot[I] (optional), isect_ids[I] (optional) from the
 *    front half's temp; n_isects = totals[0] read back by the caller.
 *  - clmgs_isect3_bin_dev: device-count form (see clmgs_isect2_emit_sort_dev): buffers and launches sized for
 *    `capacity`, the true count read on the device; above the capacity the lists are memory-safe but incomplete and
 *    the caller redoes the camera. */
size_t clmgs_isect3_front_temp_bytes(int V, int n_tiles);
int clmgs_isect3_front(void* stream, int V, const float* means2d, const int32_t* radii, int tile_size, int tile_width,
                       int tile_height, const void* packed, int64_t* totals, int64_t* row_cum, void* temp,
                       size_t temp_bytes);
size_t clmgs_isect3_bin_temp_bytes(int64_t n_isects, int n_tiles);
int clmgs_isect3_bin(void* stream, int V, int64_t n_isects, const float* depths, int tile_width, int tile_height,
                     const int64_t* row_cum, const void* front_temp, int32_t* flatten_ids, int32_t* offsets,
                     int64_t* isect_ids, int32_t* emit_slot, void* temp, size_t temp_bytes);
int clmgs_isect3_bin_dev(void* stream, int V, int64_t capacity, const int64_t* n_isects_dev, const float* depths,
                         int tile_width, int tile_height, const int64_t* row_cum, const void* front_temp,
                         int32_t* flatten_ids, int32_t* offsets, int64_t* isect_ids, int32_t* emit_slot, void* temp,
                         size_t temp_bytes);
int clmgs_rasterize_fwd_dev(void* stream, int C, int N, int64_t capacity, const int64_t* n_isects_dev,
                            const float* backgrounds, int width, int height, int tile_size,
                            int tile_width, int tile_height, const int32_t* offsets,
                            const int32_t* flatten_ids, void* packed, float* render_colors,
                            float* render_alphas, int32_t* last_ids);
int clmgs_rasterize_bwd_dev(void* stream, int C, int N, int64_t capacity, const int64_t* n_isects_dev,
                            const void* packed, const float* backgrounds, int width, int height,
                            int tile_size, int tile_width, int tile_height, const int32_t* offsets,
                            const int32_t* flatten_ids, const float* render_alphas,
                            const int32_t* last_ids, const float* v_render_colors,
                            const float* v_render_alphas, const int32_t* emit_slot,
                            const int64_t* row_cum, void* partials);

/* ---- fused per-camera front end (engine-internal fast path; same arithmetic as the op chain
 * strategies/clm_offload/engine.py:650-691 forward and :703-742 + densification.py:59-102 backward)
 * For i < V, row g = filter ? filter[i] : i of the RAW parameter tensors (xyz[N,3], opacity_raw[N],
 * scaling_raw[N,3], rotation_raw[N,4]); SH rows from sh_rows[g] (sh_by_filter) or sh_rows[i].
 * viewmat[16] (row-major world->camera), K[9], campos[3] are HOST pointers (copied into the launch).
 * fwd: exp / sigmoid, projection, SH colour, +0.5 clamp -> radii[V] means2d[V,2] depths[V]
 * conics[V,3] colors[V,3] opacities[V] and the packed raster records packed[V,16].
 * bwd: from packed_grad[V,16] -- or, packed_grad == NULL, from the partial lines `partials` of
 * clmgs_rasterize_bwd, row i summing the lines [row_cum[i-1], row_cum[i]) in ascending order --
 * ACCUMULATES into g_xyz[N,3] g_opacity[N] g_scaling[N,3]
 * g_rotation[N,4] (raw-parameter gradients) and g_sh_rows (indexed like sh_rows), and, if
 * max_radii2D != NULL, updates the densification statistics of every filter row (or, with
 * stats_only_visible, of the rows with radius > 0: densification.py:105-147). */
/* Packed small attributes: opacity_raw == scaling_raw == rotation_raw == NULL means `xyz` is the
 * 16 B-aligned [N,12] table  xyz 3 | opacity 1 | scaling 3 | rotation 4 | pad  (one 48 B row per
 * Gaussian instead of four scattered pieces, each of which costs a 64 B line); in the backward the
 * gradient outputs follow the same convention (g_opacity == g_scaling == g_rotation == NULL:
 * g_xyz is the packed [N,12] gradient table), and packed parameters go with packed gradients.
 * Statistics: grad_accum == denom == NULL with max_radii2D != NULL means max_radii2D is a 16 B-aligned
 * [N,4] table  max radius | grad accum | count | pad  (one row instead of three scattered floats).
 * sh_index[V] (i32, optional, with sh_by_filter = 1): position i reads SH row sh_rows[sh_index[i]] and
 * accumulates into g_sh_rows[sh_index[i]] -- sh_rows is then a staging table holding only the rows a
 * batch touches (host-resident mode: clm_offload/engine.py:494-508 moves rows host -> GPU per batch).
 * sh_stamp[N] (i32, optional, bwd with sh_by_filter = 1) + cur_step: FIRST-TOUCH stores.  sh_stamp[row]
 * is the optimizer step whose gradient g_sh_rows[row] holds; a row whose stamp differs from cur_step is
 * STORED (its previous content was consumed -- clmgs_adam_catch_up(keep_grad = 1) does not clear it) and
 * stamped cur_step, later cameras of the same step accumulate.  sh_stamp doubles as the deferred
 * optimizer's g_step table.  The reference clears the whole gradient buffer every batch
 * (clm_offload/engine.py:870-882) and accumulates (send_shs2cpu_grad_buffer_stream, accum = True). */
int clmgs_preprocess_fwd(void* stream, int V, const int64_t* filter, const float* xyz,
                         const float* opacity_raw, const float* scaling_raw,
                         const float* rotation_raw, const float* sh_rows, int sh_by_filter,
                         const float* viewmat_host, const float* K_host, const float* campos_host,
                         int width, int height, int degree, float eps2d, float near_plane,
                         float far_plane, float radius_clip, int32_t* radii, float* means2d,
                         float* depths, float* conics, float* colors, float* opacities, void* packed,
                         const int32_t* sh_index);
int clmgs_preprocess_bwd(void* stream, int V, const int64_t* filter, const float* xyz,
                         const float* opacity_raw, const float* scaling_raw,
                         const float* rotation_raw, const float* sh_rows, int sh_by_filter,
                         const float* viewmat_host, const float* K_host, const float* campos_host,
                         int width, int height, int degree, float eps2d, const int32_t* radii,
                         const void* packed_grad, float* g_xyz, float* g_opacity, float* g_scaling,
                         float* g_rotation, float* g_sh_rows, float* max_radii2D, float* grad_accum,
                         float* denom, float* v_means2d_out, int stats_only_visible,
                         const void* partials, const int64_t* row_cum, const int32_t* sh_index,
                         int32_t* sh_stamp, int cur_step);

/* ---- clm_kernels.fused_ssim  (base_engine.py:5,93; definition utils/loss_utils.py:26-85)
 * img1,img2 [B,CH,H,W].  fwd adds per-block SSIM-map sums into ssim_sum[1024] (caller zeroes
 * it; mean = sum of the 1024 slots / (B*CH*H*W)) and, when the three dm_* maps are non-NULL, the
 * partial derivatives needed by bwd.  bwd writes v_img1 = v_mean[0] * inv_numel * dSSIM_sum/dimg1
 * (v_mean is a DEVICE scalar: the upstream cotangent never visits the host). */
int clmgs_ssim_fwd(void* stream, int B, int CH, int H, int W, const float* img1,
                   const float* img2, float* ssim_sum, float* dm_dmu1, float* dm_dsigma1_sq,
                   float* dm_dsigma12);
int clmgs_ssim_bwd(void* stream, int B, int CH, int H, int W, const float* img1,
                   const float* img2, const float* v_mean, float inv_numel, const float* dm_dmu1,
                   const float* dm_dsigma1_sq, const float* dm_dsigma12, float* v_img1);

/* ---- fused training loss  (strategies/base_engine.py:79-103: FusedCompiledLoss + loss_combined)
 * loss = (1 - lambda) * mean|img - gt| + lambda * (1 - SSIM(img, gt)), gt = clamp(gt_u8 / 255).
 * img is a [3,H,W] VIEW given by its element strides (so the rasterizer's [H,W,3] output is read
 * in place); gt_u8 is planar [3,H,W].  fwd adds per-block (sum|x-y|, sum SSIM) pairs into
 * partials[2 * clmgs_loss_slots()] (caller zeroes; total = column sums) and, if m1..m3 != NULL,
 * writes the three [3,H,W] derivative maps bwd needs.  bwd writes v_img with img's strides;
 * v_loss is a DEVICE scalar. */
int clmgs_loss_slots(void);
int clmgs_l1_ssim_loss_fwd(void* stream, int H, int W, const float* img, int64_t stride_c,
                           int64_t stride_y, int64_t stride_x, const uint8_t* gt_u8,
                           float* partials, float* m1, float* m2, float* m3);
int clmgs_l1_ssim_loss_bwd(void* stream, int H, int W, const float* img, int64_t stride_c,
                           int64_t stride_y, int64_t stride_x, const uint8_t* gt_u8,
                           const float* v_loss, float lambda_dssim, const float* m1,
                           const float* m2, const float* m3, float* v_img);

/* ---- clm_kernels row movers  (clm_offload/engine.py:499-505, 622-636, 789-802, 815-822)
 * dst/src may be device memory or pinned (mapped) host memory.
 * gather:      dst[dst_idx ? dst_idx[i] : i] = src[src_idx ? src_idx[i] : i]
 * scatter_add: dst[dst_idx ? dst_idx[i] : i] += src[src_idx ? src_idx[i] : i]
 * Index arrays are i32 or i64 (idx_is_64). cols floats per row; grid_blocks = 0 -> auto. */
int clmgs_rows_gather(void* stream, float* dst, const float* src, const void* dst_idx,
                      const void* src_idx, int idx_is_64, int64_t n_rows, int cols,
                      int grid_blocks);
int clmgs_rows_scatter_add(void* stream, float* dst, const float* src, const void* dst_idx,
                           const void* src_idx, int idx_is_64, int64_t n_rows, int cols,
                           int grid_blocks);

/* ---- clm_kernels bitmap helpers  (clm_offload/engine.py:152-153, 200-204, 227-232)
 * bitmap elements are elem_bytes in {1,2,4,8}. */
int clmgs_scatter_to_bit(void* stream, void* bitmap, int elem_bytes, const int64_t* filter,
                         int64_t n, int bit);
int clmgs_extract_ffs(void* stream, const void* bitmap, int elem_bytes, int64_t N, uint8_t* ffs);
/* cnt[bsz-1] i32 (caller zeroes): cnt[i] = #{g : bit(bsz-1-i) & bit(bsz-2-i)} i.e. the
 * number of Gaussians visible in both micro-batch i and i+1. */
int clmgs_pair_overlap_count(void* stream, const void* bitmap, int elem_bytes, int64_t N,
                             int bsz, int32_t* cnt);
/* clm_kernels.set_signal (clm_offload/engine.py:807,825): stream-ordered write of `value`
 * to pinned host flag signal[idx], visible to a polling host thread. */
int clmgs_set_signal(void* stream, int32_t* signal_pinned, int idx, int32_t value);

/* ---- Adam  (optimizer.py:6-184; clm_offload/gaussian_model.py:161-211)
 * Row-wise Adam over p,g,m,v [*, cols] with per-column learning rate col_lr[cols] (device).
 * rows: i32/i64 row list or NULL (all n_rows rows in order); mask: u8[n_rows] or NULL
 * (rows with mask==0 are skipped; this is clm_kernels.selective_adam_update).
 * g == NULL means an all-zero gradient (pure moment decay; nothing read or cleared).
 * g is multiplied by grad_scale; bias_correction uses the 1-based `step`; betas/eps are
 * doubles so 1-beta and beta^step are formed in double on the host;
 * zero_grad != 0 clears consumed gradient rows. */
int clmgs_adam_rows(void* stream, float* p, float* g, float* m, float* v, const void* rows,
                    int idx_is_64, const uint8_t* mask, int64_t n_rows, int cols,
                    const float* col_lr, double beta1, double beta2, double eps, int step,
                    int bias_correction, float grad_scale, int zero_grad);
/* Deferred dense Adam: replay, for the listed rows (NULL = rows 0..n_rows-1), the zero-gradient
 * updates of steps last_step[row]+1 .. to_step (moment decay + parameter step, same per-element
 * operations as clmgs_adam_rows with g == NULL, bias corrections from a running product).  The
 * caller then sets last_step[rows] = to_step.  Only the first max_replay missed steps are replayed
 * exactly; for the rest the moments are decayed analytically (the first moment has decayed by
 * beta1^max_replay by then, the parameter increments are below float resolution).  Elements whose
 * moments are both zero are left untouched (every replayed step is the identity for them).
 * g / g_step (both or neither): DEFERRED gradient step.  g_step[row] (i32) is the optimizer step whose
 * gradient waits in g[row, cols]; if it lies in (last_step, to_step] it is applied here at its own step
 * with grad_scale, between the replays before and after it, and the gradient row is zeroed: the same
 * operations in the same order as the eager update at the end of that batch (optimizer.py:130-144 /
 * clm_offload/engine.py:316-328), in one pass over the row instead of two.  keep_grad = 1: the consumed
 * row is not cleared (producers that STORE on first touch: clmgs_preprocess_bwd with sh_stamp).
 * With an explicit row list every table is addressed as base + row * cols and ONLY the listed rows are touched: m / v
 * may therefore be bases moved back by row0 rows in front of a shard that holds rows row0.. only (camera-DP: the moments
 * of a row live at its owner, clm_gs_amd/strategies/clm_offload/gaussian_model.py moments_sharded). */
/* The packed [N,12] mirror of the four GPU-resident parameter tensors, and their dense Adam when the
 * engine accumulates gradients in a packed [N,12] table: params / exp_avg / exp_avg_sq are HOST
 * arrays of 4 device pointers (xyz [N,3], opacity [N,1], scaling [N,3], rotation [N,4]), lr4 a
 * host array of their 4 learning rates.  One pass: grad * grad_scale -> Adam (torch.optim.Adam's
 * bias-corrected formula) -> p / m / v written back, mirror row refreshed, gradient row zeroed.
 * g_stamp (i32 [n], optional) + cur_step: the first-touch form -- only rows with g_stamp[row] == cur_step
 * carry a gradient of this step (clmgs_preprocess_bwd stamped them); the others are not read, and nothing
 * is zeroed. */
int clmgs_pack_small(void* stream, int64_t n, const float* xyz, const float* opacity,
                     const float* scaling, const float* rotation, void* packed_p);
int clmgs_adam_small_packed(void* stream, int64_t n, float* const* params, float* const* exp_avg,
                            float* const* exp_avg_sq, const double* lr4, void* packed_p,
                            void* packed_g, double beta1, double beta2, double eps, int step,
                            int bias_correction, float grad_scale, const int32_t* g_stamp, int cur_step);
int clmgs_adam_catch_up(void* stream, float* p, float* m, float* v, const int32_t* last_step,
                        const void* rows, int idx_is_64, int64_t n_rows, int cols,
                        const float* col_lr, double beta1, double beta2, double eps, int to_step,
                        int bias_correction, int max_replay, float* g, const int32_t* g_step,
                        float grad_scale, int keep_grad);
/* Host (OpenMP) variant on pinned/pageable host memory: cpu_adam.FusedCPUAdam row group
 * update (clm_offload/engine.py:316-328).  If signal != NULL, busy-waits until
 * *signal != 0 before touching the rows. */
int clmgs_host_adam_rows(float* p, float* g, float* m, float* v, const int32_t* rows,
                         int64_t n_rows, int cols, const float* col_lr, double beta1,
                         double beta2, double eps, int step, int bias_correction, float grad_scale,
                         int zero_grad, const volatile int32_t* signal, int n_threads);

/* Host-resident mode, this build's form of the cpu_adam worker (clm_offload/engine.py:301-335,
 * optimizer.py:130-144): a persistent pool of host threads and a DEFERRED row optimizer.  Per row two
 * int32 stamps: last_step (step p/m/v are current as of) and g_step (step whose gradient waits in g,
 * 0 = none).  clmgs_host_rows_prepare brings the listed rows (rows == NULL: rows 0..n_rows-1) to
 * `to_step` -- zero-gradient steps are replayed in registers, the waiting gradient is applied at its
 * own step with grad_scale -- sets last_step = to_step, g_step = next_g_step, and copies each
 * up-to-date parameter row into stage[k] (contiguous pinned staging for a hipMemcpyAsync; NULL to skip).
 * sparse != 0 (sparse_adam): rows are stepped only when they carry a gradient, nothing is replayed.
 * Arithmetic per element and step = clmgs_host_adam_rows'. */
int clmgs_host_pool_start(int n_threads);  /* <= 0: sized to the CPUs the process may use (cgroup quota aware) */
int clmgs_host_usable_cpus(void);
/* hipMemcpyAsync of `bytes` between pinned host memory and HBM on `stream` (SDMA engine; kind 1 = host
 * -> device, 2 = device -> host): how the host-resident mode moves its contiguous staging chunks
 * (replaces clm_kernels.send_shs2gpu_stream's zero-copy gather, clm_offload/engine.py:499-505). */
int clmgs_memcpy_async(void* stream, void* dst, const void* src, size_t bytes, int kind);
int clmgs_host_rows_prepare(float* p, const float* g, float* m, float* v, int32_t* last_step,
                            int32_t* g_step, const int32_t* rows, int64_t n_rows, int cols,
                            const float* col_lr, double beta1, double beta2, double eps, int to_step,
                            int next_g_step, int bias_correction, float grad_scale, int max_replay,
                            float* stage, int sparse);

/* ---- densification statistics  (clm_offload/gaussian_model.py:833-851;
 *      no_offload/gaussian_model.py:767-783; densification.py:59-147)
 * For i < n (only rows with radii[i] > 0 when only_visible != 0): g = filter ? filter[i] : i;
 * max_radii2D[g] = max(., radii[i]); accum[g] += |v_means2d[i] * (W/2, H/2)|; denom[g] += 1. */
int clmgs_densify_stats(void* stream, int64_t n, const int64_t* filter, const float* v_means2d,
                        const int32_t* radii, int only_visible, float half_w, float half_h,
                        float* max_radii2D,
                        float* xyz_gradient_accum, float* denom);

/* ---- storage order of the rows (no reference counterpart; clm_gs_amd/utils.py morton_order, trainer / densification):
 * order[i] = the row that comes i-th along the Z-order curve of (x, y), 16 bits per axis, ties in row order (stable):
 * code = spread(qx) | spread(qy) << 1 with q = rint((double(p) - lo) / max(hi - lo, 1e-30) * 65535); lo_hi (device) =
 * lo_x lo_y hi_x hi_y as doubles.  xyz [n,3] f32, order [n] i64, temp: clmgs_morton_order_temp_bytes(n); n < 2^31. */
size_t clmgs_morton_order_temp_bytes(int64_t n);
int clmgs_morton_order(void* stream, int64_t n, const float* xyz, const double* lo_hi, int64_t* order, void* temp,
                       size_t temp_bytes);

/* ---- fast_tsp.find_tour  (clm_offload/engine.py:179): open-tour heuristic on an n x n
 * integer distance matrix (greedy nearest neighbour + 2-opt until no improvement). */
int clmgs_tsp_tour(int n, const int64_t* dist, int32_t* tour);

/* ---- simple_knn._C.distCUDA2  (strategies/clm_offload/gaussian_model.py:60-63; init-time only)
 * pts_sorted[n,3]: points sorted by grid cell id ((z*gy + y)*gx + x, cell = floor((p - origin)/h)
 * clamped); cell_start[gx*gy*gz + 1] i32: first sorted index of every cell.  Writes the mean
 * squared distance to the 3 nearest neighbours, in sorted order.  Exact while the answer lies
 * within max_ring cells. */
int clmgs_knn3_mean_dist2(void* stream, int n, const float* pts_sorted, const int32_t* cell_start,
                          float ox, float oy, float oz, float h, int gx, int gy, int gz,
                          int max_ring, float* mean_d2_sorted);

/* Host-resident batch (sh_residency="host"): the rows a batch touches, grouped by the camera that uses them FIRST (slot
 * order of the rows that still have to be staged) and LAST (hand-back order of the gradient rows) -- what the reference
 * derives from its bitmap with ffs / sort / tolist (clm_offload/engine.py:137-260), as two stable one-digit radix sorts.
 * touched[n] ascending row ids; bitmap[N] of elem_bytes-wide words, bit bsz-1-i = camera i; staged (u8[N], optional):
 * rows already staged for this batch (not "late").  Out: late_sorted[<= n] (late rows by first camera), rows_by_last[n],
 * slot_of[row] = slot0 + position in late_sorted, counts[2*bsz+1] (device) = late rows per first camera | rows per
 * last camera | number of late rows. */
size_t clmgs_host_groups_temp_bytes(int64_t n);
int clmgs_host_groups(void* stream, int64_t n, const int64_t* touched, const void* bitmap, int elem_bytes, int bsz,
                      const uint8_t* staged, int slot0, int32_t* late_sorted, int32_t* rows_by_last, int32_t* slot_of,
                      int64_t* counts, void* temp, size_t temp_bytes);

/* Camera-DP, locality exchange, step F (clm_gs_amd/dp.py publish_rows; net-new, the reference is single GPU): packs
 * the message an owner all-gathers -- msg[0 .. chunk*cols) = the rows table[own_rows[i]] (zeros where stamp != NULL and
 * stamp[row] != step: the row's gradient line is not of this step), msg[chunk*cols + i] = the bits of
 * int32(own_rows[i] - lo).  cols % 4 == 0, msg / table 16 B aligned, chunk >= n_rows. */
int clmgs_publish_pack(void* stream, float* msg, const float* table, const int64_t* own_rows, const int32_t* stamp,
                       int step, int64_t lo, int64_t n_rows, int64_t chunk, int cols);

/* Camera-DP, the small attributes (xyz / opacity / scaling / rotation) computed by the OWNER of a row range (net-new:
 * the reference is single GPU; its SelectiveAdam, optimizer.py:6-88, is the sparse form on one device).  Between two
 * refreshes a rank's copies of the rows it does not own are stale by a bounded amount; three entries:
 *  - clmgs_visibility_candidates: mask[N] (u8) = 1 for every row OUTSIDE [own_lo, own_hi) that may pass the cull of
 *    clmgs_visibility_select_count in any of the C cameras if its stored mean is off by up to pos_margin (Euclidean) and
 *    its largest scale by a factor up to scale_gain -- a superset of the rows the exact cull keeps for the true values;
 *    the caller fetches the candidates' current lines from their owners before the exact pass.
 *  - clmgs_small_rows_scatter: packed [n_rows,12] lines (xyz 3 | opacity 1 | scaling 3 | rotation 4 | pad) written to
 *    the four parameter tensors and the packed mirror at the unique row ids `rows`.
 *  - clmgs_adam_small_packed_range: clmgs_adam_small_packed on the rows row_begin <= r < row_end only (row_end < 0:
 *    all rows); every other row of every table is left bit for bit as it was. */
int clmgs_visibility_candidates(void* stream, int C, int N, int own_lo, int own_hi, const float* means,
                                const float* log_scales, const float* viewmats, const float* Ks, int width,
                                int height, float eps2d, float near_plane, float far_plane, float pos_margin,
                                float scale_gain, uint8_t* mask);
int clmgs_small_rows_scatter(void* stream, int64_t n_rows, const int64_t* rows, const void* lines, float* xyz,
                             float* opacity, float* scaling, float* rotation, void* packed_p);
int clmgs_adam_small_packed_range(void* stream, int64_t n, int64_t row_begin, int64_t row_end, float* const* params,
                                  float* const* exp_avg, float* const* exp_avg_sq, const double* lr4, void* packed_p,
                                  void* packed_g, double beta1, double beta2, double eps, int step,
                                  int bias_correction, float grad_scale, const int32_t* g_stamp, int cur_step);

/* Single GPU, round 5: the dense Adam of the small attributes DEFERRED per block of 256 consecutive (Z-ordered) rows
 * (replaces the eager clmgs_adam_small_packed of a batch; the reference steps them eagerly with torch's Adam,
 * strategies/clm_offload/engine.py:870-882 / optimizer.py:91-184 -- same arithmetic per step, applied later).
 * blk_last[ceil(n/256)] holds the optimizer step each block is current as of.  A call brings up to `to_step` every
 * block that (a) may hold a row visible in one of the C cameras when its values are stale by the waiting steps'
 * worth of Adam's step bound (pos_margin[k], scale_gain[k] for k waiting steps), (b) is clmgs_small_deferred_kmax()
 * steps behind, or (c) any block at all when flush_all != 0 -- replaying the waiting steps one by one with the
 * constants of THEIR step (lr4_hist[j][4], step_index[j]: entry j describes step to_step - j; n_hist entries) and the
 * row's waiting gradient line (packed_g row, stamp g_stamp[row]) at the step it belongs to.  Bit-identical to the
 * eager sequence; the packed mirror is refreshed for the blocks processed. */
int clmgs_small_deferred_kmax(void);
int clmgs_adam_small_deferred(void* stream, int64_t n, float* const* params, float* const* exp_avg,
                              float* const* exp_avg_sq, void* packed_p, const void* packed_g, const int32_t* g_stamp,
                              int32_t* blk_last, int to_step, int n_hist, const double* lr4_hist,
                              const int32_t* step_index, const float* pos_margin, const float* scale_gain, double beta1,
                              double beta2, double eps, float grad_scale, int C, const float* viewmats, const float* Ks,
                              int width, int height, float eps2d, float near_plane, float far_plane, int flush_all,
                              uint8_t* blk_flag /* optional [ceil(n/256)] out: 0 = no row of the block can be visible in
                              any of the C cameras; input of clmgs_visibility_select_count_blocks */);

/* Profiling aid: counters of the CLMGS_BWD_DEBUG=3 variant of the backward tile kernel. */
int clmgs_debug_counters(unsigned long long* out16, int reset);

/* Device error word of the single-launch scan / sort-pass kernels of the binning chain (csrc/onesweep.h: workgroups
 * of one launch hand per-chunk aggregates to each other by decoupled look-back; every poll loop is bounded).
 * *bits: 1 = a scan look-back, 2 = a sort-pass look-back gave up after its bound -- the lists of that call are
 * invalid; 4 = clmgs_adam_small_deferred met a block further behind than the step history it was given (the caller's
 * invariant was broken: those rows' parameters are wrong).  Synchronises the device; `reset` clears the word.  No reference counterpart (gsplat.isect_tiles sorts
 * with cub, strategies/base_engine.py:175-186); callers check it where they synchronise anyway (evaluation, saving,
 * the end of a benchmark). */
int clmgs_device_errors(uint32_t* bits, int reset);

/* ---- pinned host memory  (numba.cuda.pinned_array at clm_offload/gaussian_model.py:34-44) */
void* clmgs_pinned_alloc(size_t bytes);
int clmgs_pinned_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* CLMGS_H */
