/*
 * litegs_hip.h -- C ABI of liblitegs_hip.so, the MI355X (gfx950) implementation of the LiteGS
 * `litegs.render` hot path.
 *
 * This is the drop-in boundary.  The reference's boundary is the pybind11/ATen module `litegs_fused`
 * (GR/ext_cuda.cpp:9-35, GR = litegs/submodules/gaussian_raster); each entry point below is the
 * raw-pointer form of one of its 26 exports (or of one of the torch ops the reference's wrapper glues
 * between them), so kernels are testable without torch and bindable from any host language.
 * litegs_amd/fused.py is the ATen-shaped binding (same names / argument order / returned tensors as
 * GR/ext_cuda.cpp); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; tensors are C-contiguous float32 / int32 /
 *     int64 exactly as the reference lays them out (SoA, Gaussian index innermost: [C,N] or [V,C,N]);
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); launches are async;
 *   - `valid_length` (nullable) is the reference's GPU-driven bound: int32[1] on the device, work with
 *     index >= *valid_length is skipped inside the kernel (no host sync);
 *   - return value: 0 on success, otherwise the hipError_t of the failed launch / argument check.
 *   - V = views, N = Gaussians after compaction, C/chunks = number of 128-wide chunks, S = chunk size,
 *     A = allocated (predicted) visible chunks, L = tile-instance table length, H/W image, TH/TW tile.
 */
#ifndef LITEGS_HIP_H
#define LITEGS_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- compact.hip : GR/compact.h ------------------------------------------------------------- */

/* frustum_culling_aabb (GR/compact.cu:419-551, compact.h:28).  visibility bool[M]; *visible_num = count;
 * visible_chunk_id int64[M] = visible ids ASCENDING then arange() tail.  Feedback copy: lg_feedback_d2h. */
int lg_frustum_culling_aabb(const float* aabb_origin, const float* aabb_ext, const float* frustumplane, int V, int M,
                            uint8_t* visibility, int* visible_num, int64_t* visible_chunk_id, void* stream);

/* cull_compact_activate (GR/compact.cu:826-893,983-1085, compact.h:3-8) */
int lg_cull_compact_activate(int sh_degree, const int64_t* visible_chunk_id, const int* visible_chunks_num, int A,
                             const float* view_matrix, int V,
                             const float* position, const float* scale, const float* rotation,
                             const float* sh_base, const float* sh_rest, const float* opacity, int chunks, int S,
                             float* out_position /*[4,A,S]*/, float* out_scale /*[3,A,S]*/, float* out_rotation /*[4,A,S]*/,
                             float* out_color /*[V,3,A,S]*/, float* out_opacity /*[1,A,S]*/, void* stream);

/* activate_backward (GR/compact.cu:896-980,1087-1212, compact.h:10-16); R = sh_rest.shape[0] */
int lg_activate_backward(int sh_degree, const int64_t* visible_chunk_id, const int* visible_chunks_num, int A,
                         const float* view_matrix, int V,
                         const float* position, const float* scale, const float* rotation, const float* opacity,
                         int chunks, int S, int R,
                         const float* g_position /*[4,A,S]*/, const float* g_scale, const float* g_rotation,
                         const float* g_color /*[V,3,A,S]*/, const float* g_opacity,
                         float* d_position /*[3,A,S]*/, float* d_scale, float* d_rotation,
                         float* d_sh_base /*[1,3,A,S]*/, float* d_sh_rest /*[R,3,A,S]*/, float* d_opacity, void* stream);

/* adamUpdate (GR/compact.cu:320-417, compact.h:18-23): chunk form ([E,chunks,S] params, [E,A,S] compact grads)
 * and primitive form ([E,N], int64 mask[N]).  No bias correction, as in the reference. */
int lg_adam_update_chunk(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* visible_chunk_id,
                         const int* valid_length, int E, int chunks, int A, int S, int grad_dense /*grad is [E,chunks,S]*/,
                         float lr, float b1, float b2, float eps, void* stream);
int lg_adam_update_primitive(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* mask,
                             int E, int N, float lr, float b1, float b2, float eps, void* stream);

/* gpu_driven_pipeline_sparse_op (GR/compact.cu:1222-1336, compact.h:30-36).
 * dtype: 0 f32, 1 i32, 2 i64, 3 f64, 4 i16, 5 i8;  op: 0 add, 1 min, 2 max */
int lg_sparse_scatter(void* A_, const void* B, const int64_t* chunk_ids, const int* valid_count,
                      int E, int chunks, int alloc, int S, int dtype, int op, void* stream);

/* Statistic epochs of the native executor: visible_count and the moments "fragment_weight" / "fragment_err" of the statistics helper
 * (litegs/utils/statistic_helper.py) accumulated in one pass over the frame's gradient records [A*S,16] (slots 8-11: M0, fragment count,
 * fragment weight sum, err_square; lg_fused_backward in statistic mode), the packed splat records (opacity) and the tile counts: the
 * scatters gpu_driven_pipeline_sparse_op would perform, without the torch glue in between.  All destinations are [chunks, S]. */
int lg_stat_accumulate(const float* packed_grad, const float* packed, const int* alloc, const int64_t* chunk_ids, const int* valid_count,
                       int A, int chunks, int S, int* visible_count, float* w_sum, float* w_sq, int* w_cnt,
                       float* e_sum, float* e_sq, int* e_cnt, void* stream);

/* data-parallel helpers (new: the reference has no multi-GPU path): mask[ids[i]]=1 for i<*count; ordered compaction of an
 * int32 mask into (count, ids) with the output contract of lg_frustum_culling_aabb */
int lg_mark_chunks(const int64_t* ids, const int* count, int A, int* mask, void* stream);
int lg_compact_mask(const int* mask, int M, int* count, int64_t* ids, void* stream);
int lg_compact_mask_rank(const int* mask, int M, int* count, int64_t* ids, int64_t* rank /*[M]: position of each kept chunk in ids*/, void* stream);

/* the 4-byte async device->pinned-host feedback copy of GR/compact.cu:538 and GR/binning.cu:148 */
int lg_feedback_d2h(int* host_dst, const int* device_src, void* stream);

/* ---- transform.hip : GR/transform.h --------------------------------------------------------- */
int lg_mvp_transform_forward(const float* world /*[4,N]*/, const float* view, const float* proj, const int* valid_length,
                             int V, int N, float* view_pos /*[V,4,N]*/, float* ndc_pos /*[V,4,N]*/, void* stream); /* transform.cu:378-470 */
int lg_mvp_transform_backward(const float* g_ndc, const float* g_view, const float* view, const float* proj,
                              const float* view_pos, const int* valid_length, int V, int N, float* g_world /*[4,N]*/, void* stream); /* :472-598 */
int lg_create_transform_matrix_forward(const float* quat /*[4,N]*/, const float* scale /*[3,N]*/, const int* valid_length,
                                       int N, float* T /*[3,3,N]*/, void* stream);                                /* :92-149 */
int lg_create_transform_matrix_backward(const float* gT, const float* quat, const float* scale, const int* valid_length,
                                        int N, float* g_quat, float* g_scale, void* stream);                        /* :151-256 */
int lg_jacobian_rayspace(const float* view_pos, const float* proj, const int* valid_length, int V, int N, int H, int W,
                         float* J /*[V,3,3,N], fully written*/, void* stream);                                      /* :23-90 */
int lg_create_cov2d_forward(const float* J, const float* view, const float* T, const int* valid_length, int V, int N,
                            float* cov2d /*[V,2,2,N]*/, void* stream);                                              /* :737-821 */
int lg_create_cov2d_backward(const float* g_cov2d, const float* J, const float* view, const float* T, const int* valid_length,
                             int V, int N, float* gT /*[3,3,N]*/, void* stream);                                    /* :824-927 */
int lg_eigh_inv_2x2_forward(const float* cov2d, const int* valid_length, int V, int N,
                            float* eig_val /*[V,2,N] or NULL*/, float* eig_vec /*[V,2,2,N] or NULL*/, float* inv /*[V,2,2,N]*/,
                            void* stream);                                                                          /* :1365-1487 */
int lg_inv_2x2_backward(const float* inv, const float* g_inv, const int* valid_length, int V, int N, int zero_nonfinite,
                        float* g_cov2d, void* stream);                                                              /* :1425-1518 (+wrapper.py:591) */
int lg_sh2rgb_forward(int degree, const float* sh_base, const float* sh_rest, const float* dirs, int V, int N,
                      float* rgb /*[V,3,N]*/, void* stream);                                                        /* :952-1089 */
int lg_sh2rgb_backward(int degree, const float* g_rgb, const float* dirs, int V, int N, int rest_dim,
                       float* d_sh_base, float* d_sh_rest, float* d_dirs, void* stream);                            /* :1091-1363 */
int lg_world2ndc_forward(const float* world, const float* viewproj, int V, int N, float* ndc, float* recp_w, void* stream);   /* :602-669 */
int lg_world2ndc_backward(const float* viewproj, const float* ndc, const float* recp_w, const float* g_ndc, int V, int N,
                          float* g_pos, void* stream);                                                              /* :671-731 */

/* Learnable cameras (GR/compact.cu:17-316; wrapper.py:772-791).  view_params [V,7] = unit-less quaternion (r,x,y,z) + translation;
 * outputs row-vector matrices [V,4,4] and frustum planes [V,6,4].  Backward keeps the reference's integer img_w/img_h ratio
 * (compact.cu:268) and its quaternion-sphere projection (:271-277); grad_recp_tan_half_fov_x[1] is overwritten with the
 * ordered sum over views (the reference's unsynchronised "+=" is a race for V>1). */
int lg_create_viewproj_forward(const float* view_params, const float* recp_tan_half_fov_x, int V, int H, int W, float z_near, float z_far,
                               float* view_matrix, float* proj_matrix, float* viewproj_matrix, float* frustumplane, void* stream); /* compact.cu:17-135 */
int lg_create_viewproj_backward(const float* view_matrix_grad, const float* proj_matrix_grad, const float* viewproj_matrix_grad,
                                const float* view_params, const float* recp_tan_half_fov_x, int V, int H, int W, float z_near,
                                float z_far, float* grad_view_params, float* grad_recp_tan_half_fov_x, void* stream);          /* compact.cu:137-316 */

/* ---- binning.hip : GR/binning.h ------------------------------------------------------------- */
int lg_get_allocate_size(const float* ndc, const float* view_z, const float* inv_cov2d, const float* opacity,
                         const int* valid_length, int V, int N, int H, int W, int TH, int TW,
                         int32_t* left_up /*[V,2,N] or NULL*/, int32_t* right_down, int32_t* allocate_size /*[V,N]*/,
                         void* stream);                                                                             /* binning.cu:290-440 */
/* create_table, first half (binning.cu:34-110): keys must be zero-filled; sorted_id int64 (torch.sort) or int32, or NULL = ascending id
 * (prefix_sum then runs over the splats in id order: the grouped form of the table, lg_tile_group + lg_tile_depth_sort_unordered).
 * temp (lg_duplicate_with_keys_temp_bytes): queue of the splats that touch many tiles, emitted by a second launch (giants in
 * parts of 1024 tiles).  N < 2^24. */
long long lg_duplicate_with_keys_temp_bytes(int V, int N, long long table_len);
int lg_duplicate_with_keys(const float* ndc, const float* inv_cov2d, const float* opacity, const int32_t* prefix_sum,
                           const void* depth_sorted_id, int sorted_id_is_int64, int V, int N, int H, int W, int TH, int TW,
                           long long table_len, int32_t* keys, int32_t* values, void* temp, long long temp_bytes, void* stream);
/* create_table as ONE call for a single view (binning.cu:123-226): emission that also counts the sort's digits, zero padding of an
 * over-allocated table (key 0 = no tile, sorts to the front), stable sort over bits [0, end_bit).  keys/vals need no initialisation;
 * the result is in the b pair when lg_radix_sort_num_passes(0, end_bit) is odd, else in the a pair. */
long long lg_create_table_temp_bytes(int N, long long table_len, int end_bit);
int lg_create_table(const float* ndc, const float* inv_cov2d, const float* opacity, const int32_t* prefix_sum, const void* depth_sorted_id,
                    int sorted_id_is_int64, int N, int H, int W, int TH, int TW, long long table_len, int end_bit,
                    int32_t* keys_a, int32_t* vals_a, int32_t* keys_b, int32_t* vals_b, void* temp, long long temp_bytes, void* stream);
/* create_table, second half: stable LSD radix sort replacing cub::DeviceRadixSort::SortPairs (binning.cu:204-221).
 * Ping-pongs a->b->a...; result is in the b pair when lg_radix_sort_num_passes() is odd, else in the a pair. */
/* 0: the radix passes rank keys with lane-ordered returning LDS adds, verified by a device self-test on first use; 1: ballot ranking
 * (stability by construction; also forced by LITEGS_RADIX_RANK=ballot).  The first call runs the self-test (synchronous). */
int lg_radix_rank_mode(void);
int lg_radix_set_rank_mode(int mode);   /* test hook: 0 / 1 force, -1 re-run the self-test */
long long lg_radix_sort_temp_bytes(long long n);
int lg_radix_sort_num_passes(int begin_bit, int end_bit);
int lg_radix_sort_pairs(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, long long n,
                        int begin_bit, int end_bit, void* temp, long long temp_bytes, void* stream);
/* device-bounded variants for the GPU-driven pipeline: only the first min(n, *n_dev) elements are sorted / scanned for
 * tile ranges, i.e. the actual instance count rather than the 1.5x over-allocated table (n_dev may be NULL) */
int lg_radix_sort_pairs_bounded(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, long long n, const int* n_dev,
                                int begin_bit, int end_bit, void* temp, long long temp_bytes, void* stream);
int lg_tile_range_bounded(const int32_t* sorted_keys, int V, long long L, const int* n_dev, int max_tile, int32_t* out, void* stream);
/* the executor's tile sort as one call (tests): (key, value) pairs, keys in 0..max_tile, values below 2^value_bits (0: unknown) -> the values
 * grouped by key in input order (in vals_a for an even number of 8-bit passes over the key bits, else vals_b) and tileRange's table in
 * range_out[max_tile + 2].  temp: lg_radix_sort_temp_bytes(n).  *ranges_from_sort (host int) = 1 when the packed passes ran and the last one
 * left the ranges itself (binning.hip radix_onesweep_kernel PACK; the sorted keys then do not exist), 0 for key / value passes + range scan. */
int lg_tile_sort_ranges(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, long long n, const int* n_dev, int max_tile,
                        int value_bits, void* temp, long long temp_bytes, int32_t* range_out, int* ranges_from_sort, void* stream);
/* depth sort keys + gathered inclusive scan: the torch.sort / gather / cumsum glue of wrapper.py:739-745 */
int lg_depth_sort_keys(const float* depth, long long n, uint32_t* keys, uint32_t* vals, void* stream);
long long lg_scan_temp_bytes(long long n);
int lg_gather_inclusive_scan(const int32_t* src, const void* idx, int idx_is_int64, long long n, int32_t* out,
                             void* temp, long long temp_bytes, void* stream);
int lg_tile_range(const int32_t* sorted_keys, int V, long long L, int max_tile, int32_t* out /*[V,max_tile+2]*/, void* stream); /* binning.cu:228-287 */
/* tilesort.hip -- replaces the depth half of wrapper.py:739-745 (torch.sort over all visible splats) + the reliance on the stable
 * tile sort (binning.cu:205-220) to carry that order into the tiles: every tile's list in vals [V,L] (splat ids grouped by tile,
 * ascending id inside a tile; tile_start from lg_tile_range) is sorted in place by (depth[V,N] of the splat, id).
 * scratch [V,L] uint32: only touched for lists longer than 1024. */
int lg_tile_depth_sort(int32_t* vals, const int32_t* tile_start, const float* depth, int V, long long L, int N, int ntiles,
                       uint32_t* scratch, void* stream);
/* Grouping by tile WITHOUT a sort (executor, per-tile-depth-sort mode; no reference counterpart: GR/binning.cu:205-220 uses a stable radix
 * sort because its emission order must survive inside a tile, which the per-tile sort makes unnecessary): keys[L] (0 .. max_tile, unsorted)
 * / vals[L] -> tile_start [max_tile + 2] as lg_tile_range would leave it for the sorted keys, out_vals[L] grouped by key (arbitrary order
 * inside a key).  temp: 2 * (max_tile + 2) ints.  lg_tile_depth_sort_unordered then orders every list by (depth, id). */
int lg_tile_group(const int32_t* keys, const int32_t* vals, long long L, int max_tile, int32_t* tile_start, int32_t* out_vals, void* temp, void* stream);
int lg_tile_depth_sort_unordered(int32_t* vals, const int32_t* tile_start, const float* depth, int V, long long L, int N, int ntiles,
                                 uint32_t* scratch, void* stream);
int lg_memset_async(void* ptr, int value, long long bytes, void* stream);

/* ---- raster.hip : GR/raster.h --------------------------------------------------------------- */
int lg_packed_record_floats(void);   /* floats per packed splat record (16) */
int lg_packed_grad_floats(void);     /* floats per packed gradient record (16) */
int lg_pack_forward_params(const float* ndc, const float* inv_cov2d, const float* color, const float* opacity,
                           const int* valid_length, int V, int N, int H, int W, float* packed /*[V,N,16]*/, void* stream); /* raster.cu:334-356 */
/* rasterize_forward / rasterize_forward_packed (raster.cu:162-332,386-586); tiles = specific_tiles int32[V,K] or NULL */
int lg_raster_forward(const int* sorted_points /*[V,L]*/, const int* start_index /*[V,T+2]*/, const float* packed,
                      const int* tiles, int K, int V, long long L, int N, int H, int W, int TH, int TW, int enable_statistic,
                      float* img /*[V,3,Hp,Wp]*/, float* transmitance /*[V,1,Hp,Wp]*/, short* last_contributor /*[V,1,Hp,Wp]*/,
                      int* fragment_count /*[V,1,N] zeroed*/, float* fragment_weight_sum /*[V,1,N] zeroed*/,
                      const int* order /*nullable int32[V,T]: tile schedule, a permutation of 1..T*/,
                      int* tile_work /*nullable int32[V,T+1]: receives the splats walked per tile*/, void* stream);
/* heaviest-first tile schedule (no reference kernel; statistic_helper.py:77 builds the same order in torch for specific_tiles) */
int lg_tile_order(const int* tile_work /*[V,T+1]*/, int V, int ntiles, int* order /*[V,T]*/, void* stream);
int lg_tile_work_from_last(const short* last_contributor /*[V,1,Hp,Wp]*/, int V, int H, int W, int TH, int TW, int* tile_work /*[V,T+1]*/, void* stream);
/* rasterize_backward (raster.cu:600-853,917-1037): accumulates per-splat MOMENTS into packed_grad [V,N,16] (zeroed by the caller;
 * layout in raster.hip: Mx My Mxx Mxy Myy dr dg db M0); lg_unpack_gradient turns them into the reference's four gradient tensors.
 * tile_counters (nullable, int32 [V,T+1,2]): measurement hook, per tile the visited and the contributing (tile, splat) iterations. */
int lg_raster_backward(const int* sorted_points, const int* start_index, const float* packed, const int* tiles, int K,
                       const float* final_transmitance, const short* last_contributor, const float* d_img,
                       const float* d_trans /*or NULL*/, int V, long long L, int N, int H, int W, int TH, int TW,
                       int enable_statistic, float* packed_grad, float* err_square_sum /*[V,1,N] zeroed*/,
                       int* tile_counters, const int* order /*nullable int32[V,T]: tile schedule*/, void* stream);
int lg_unpack_gradient(const float* packed_grad, const float* packed /*[V,N,16] records of lg_pack_forward_params*/,
                       const float* grad_inv_scaler /*[1] or NULL*/, const int* valid_length,
                       int V, int N, int H, int W, float* d_ndc /*[V,4,N]*/, float* d_cov2d_inv /*[V,2,2,N]*/,
                       float* d_color /*[V,3,N]*/, float* d_opacity /*[1,N]*/, void* stream);                        /* raster.cu:855-886 */
/* developer hook (no reference counterpart): selects launch variants of the blend kernels for A/B measurements.
 * key 0: tiles per workgroup of the blend backward (1, 2, 4); key 1 / 2: workgroup -> tile map of the backward / forward
 * (0 = one band per XCD, 1 = identity, C >= 2 = runs of C workgroups interleaved over the XCDs); key 4: heaviest-first tile schedule
 * on / off; key 5: blend backward of 8x16 tiles without statistics (0 generic, 1 the packed two-pixel kernel = default, 2 the
 * splat-parallel formulation); key 7: packed blend forward on / off; key 8: issue priority by schedule rank; keys 10 / 11: key emission
 * (in-workgroup tile ceiling, groups on demand); key 12: fused projection with the SH loads in front of the tile walk; key 15: look-back
 * width of small radix sorts (8 | 32); key 16: L2
 * warm-up block of the blend kernels' scalar record path (0 | 8 | 16 | 32 | 64 list positions); key 17: lean blend forward on / off;
 * key 18: measurement hooks (wrong results: bit 0 blend backward without its atomics, bits 1 / 2 forward / backward read 1024 always-cached
 * records); keys 19 / 20: KB of unused dynamic LDS per workgroup of the lean forward / fast backward (caps their occupancy); key 22: segmented blend
 * backward of the executor on / off (default off); key 23: log2 of its segment length (6 .. 12); key 26: the executor's two-pass tile sort
 * with separate key / value arrays (0), second digit + value in one word between the passes (1), ... and the range table's starts left by
 * the second pass instead of the sorted keys (2, default).  Out-of-range
 * values are refused.  Defaults are the
 * measured best; see DESIGN.md section 9. */
int lg_set_tuning(int key, int value);
/* developer hook: per-wave {start, end (100 MHz), list length << 32 | tile, XCC_ID << 32 | HW_ID} of the lean blend forward / the fast
 * blend backward into device int64[slots][4] buffers (NULL, NULL = off; tools/wave_clock.py: SIMD occupancy over a launch) */
int lg_debug_wave_clock(void* fwd_buf, void* bwd_buf);
int lg_stat_in_record_supported(int TH, int TW);   /* 1: statistic renders of this tile shape carry their three statistics in gradient-record slots 9-11 (raster.hip) */

/* ---- loss.hip : fused_ssim.fused_l1_ssim_loss (litegs/training/trainer.py:145; un-vendored submodule, formula in
 * litegs_amd/loss.py).  planes = B*C image planes of H x W.  dmaps [3,planes,H,W] carries dS/dmu1, dS/dE[x^2], dS/dE[xy]
 * from forward to backward; partial holds lg_l1_ssim_partial_floats() floats; loss is a device scalar. */
long long lg_l1_ssim_partial_floats(int planes, int H, int W);
int lg_l1_ssim_forward(const float* img, const float* gt, int planes, int H, int W, float lam,
                       float* dmaps, float* partial, float* loss, void* stream);
int lg_l1_ssim_backward(const float* img, const float* gt, const float* dmaps, const float* grad_out /*[1] or NULL*/,
                        int planes, int H, int W, float lam, float* d_img, void* stream);
/* The same loss taken straight on the raw tile-padded raster output [planes][Hp][Wp]: clamp(0,1) (litegs/render/__init__.py, the
 * image returned to the trainer is clamped) is applied on load, and the backward writes the gradient w.r.t. the raw image in the
 * padded layout (0 in the padding and where the clamp saturates) -- what rasterize_backward consumes. */
int lg_l1_ssim_forward_raster(const float* img, int Hp, int Wp, const float* gt, int planes, int H, int W, float lam,
                              float* dmaps, float* partial, float* loss, void* stream);
int lg_l1_ssim_backward_raster(const float* img, int Hp, int Wp, const float* gt, const float* dmaps, const float* grad_out,
                               int planes, int H, int W, float lam, float* d_img, void* stream);
/* the training step's pair: lg_l1_ssim_forward_raster(loss = NULL) + this backward, which also writes the loss value from the forward's
 * partial sums (same summation order as the stand-alone reduction: same bits) -- one launch less per step */
int lg_l1_ssim_backward_raster_value(const float* img, int Hp, int Wp, const float* gt, const float* dmaps, const float* grad_out,
                                     int planes, int H, int W, float lam, float* d_img, const float* partial, float* loss, void* stream);

/* ---- fused.hip : native executor of the whole path (one C call enqueues a stage; same arithmetic as the operators above).
 * There is no counterpart in the reference (its executor is the Python in litegs/render/__init__.py:11-94 + wrapper.py); these
 * entry points are what litegs_amd/fast.py binds.  view_host/proj_host are HOST float[16] (row-vector 4x4, passed to kernels by value).
 * Workspace 1 holds per-Gaussian buffers for N = A*S, workspace 2 the tile-instance table of length L. */
/* Execution context of the native executor: everything the lg_fused_* entry points need beyond their tensors.  A plain HOST struct owned
 * by the caller (one per renderer), read during the call only: the library keeps NO pointer into it and has no process-wide executor
 * state, so any number of renderers / trainers / devices can live in one process.  NULL = the defaults in brackets.
 *   depth_order   [2] how each tile's list gets its depth order: 0 the reference's structure (depth sort of all visible splats before
 *                 the emission, wrapper.py:739-745), 1 per-tile depth sort after the grouping (tilesort.hip; no sort over the splats),
 *                 2 choose per frame: 1 from 1 M compacted Gaussians on, 0 below.  Identical tables either way.
 *   bound_margin_pct [100] depth-bound culling: how far beyond a tile's saturation point its bound for the frame's next visit lies.
 *   tile_scatter  [1] (depth order 1 only) group the instances by tile with per-tile counts and cursors (lg_tile_group's kernels)
 *                 instead of the stable tile radix sort; identical tables.
 *   grad_replicas [0] splats that cover >= 128 tiles get 2^k <= 64 gradient lines behind the N regular records, the blend backward adds
 *                 into replica (tile mod 2^k), the fused backward kernels fold them (removes same-line contention at the memory-side
 *                 atomic units).  The gradient accumulator then has lg_fused_grad_lines(N) lines; the value must be the same for
 *                 lg_fused_stage1 / stage2 / backward of one frame, and hot_counter must point to a persistent device int (zero at first;
 *                 the backward kernels reset it at the end of a step).  lg_fused_backward_adam is handed the assignment table
 *                 (workspace 1 + lg_fused_hot_offset(N)) as an argument.
 *   poison, poison_host, applied_host, step_id [NULL, NULL, NULL, 0] speculative depth-bound culling: with poison set, a culled
 *                 lg_fused_stage2 enqueues no gated repeat; a violated bound raises *poison (sticky device int) and its pinned mirror
 *                 *poison_host, and every lg_fused_backward_adam returns at once while it is raised, otherwise stores step_id into
 *                 *applied_host (pinned).  The caller replays the steps after *applied_host (the first one unculled) after clearing both
 *                 words.  The pinned words must outlive every launch that can store into them: take them from lg_host_words_alloc. */
typedef struct LgFusedCtx {
    int32_t struct_bytes;       /* sizeof(LgFusedCtx): checked by every entry point */
    int32_t depth_order;
    int32_t bound_margin_pct;
    int32_t tile_scatter;
    int32_t grad_replicas;
    int32_t step_id;
    int32_t debug_validate;     /* 1: lg_fused_stage2 checks the grouped table on the device before anything indexes with it (debug_words) */
    int32_t reserved;
    int* hot_counter;           /* device int32[1] */
    int* poison;                /* device int32[1] */
    int* poison_host;           /* pinned int32[1] */
    int* applied_host;          /* pinned int32[1] */
    int* debug_words;           /* pinned int32[48], zero ([8..39]: detail records of the key emission): first violation found by the table check {code, tile, position, value, bound, ...} */
} LgFusedCtx;
/* Pinned host words for everything the device stores into asynchronously (sizing feedback, speculation mirrors, exchange headers).
 * They come from an arena inside the library that is never unmapped: a kernel that is still in flight when its owner dies stores into
 * memory that is still there, and a freed word is handed out again only after a device synchronisation.  n int32 words, zeroed. */
int* lg_host_words_alloc(int n);
void lg_host_words_free(int* words, int n);
/* Always-on counters of table words that a consumer had to neutralise instead of indexing with them (csrc/lg_sanity.h; a correct table
 * never takes those branches).  out[8] receives the counts since the last reset, summed over the library's kernels: [0] emission key out
 * of range, [1] emission walk != prefix sums, [2] tile_range boundary key skipped, [3] tile count / scatter key dropped, [4] radix sort
 * scatter position outside [0, n), [5] per-tile sort id clamped, [6] truncated tables (not an error: GR/binning.cu:63), [7] big-splat queue entry that names no slot of the prefix sums.
 * Blocking (device memcpy on the null stream): call at a synchronisation point.  No reference counterpart (the reference faults). */
int lg_sanitised_counts(int* out, int reset);
long long lg_fused_hot_offset(long long N);
long long lg_fused_grad_lines(long long N);
long long lg_fused_workspace1_bytes(long long N);
long long lg_fused_workspace2_bytes(long long L, long long N, int H, int W, int TH, int TW);
long long lg_fused_total_offset(long long N);
long long lg_fused_tile_start_offset(long long L, long long N, int H, int W, int TH, int TW);   /* int32[ntiles+2] tile ranges in workspace 2 (valid after stage 2) */
long long lg_fused_sorted_points_offset(const LgFusedCtx* ctx, long long L, long long N, int H, int W, int TH, int TW); /* int32[L] tile-grouped, depth-ordered splat ids in workspace 2 (valid after stage 2) */
long long lg_fused_unit_count_offset(long long L, long long N, int H, int W, int TH, int TW);   /* int32[17] in workspace 2: unit counts of the segmented blend backward -- full segments, then the 16 length classes of the remainders (valid after a stage 2 that rendered along a tile list) */
long long lg_fused_alloc_offset(long long N);   /* int32[N] tile counts per compacted Gaussian (valid after stage 1) */
long long lg_fused_packed_offset(long long N);  /* float[N,16] packed splat records (valid after stage 1) */
int lg_fused_stage1(const LgFusedCtx* ctx, const float* aabb_origin, const float* aabb_ext, const float* planes_dev, int chunks,
                    const float* view_host, const float* proj_host, int H, int W, int TH, int TW, int degree,
                    const float* pos, const float* scale, const float* rot, const float* sh0, const float* shr, const float* opa, int S,
                    int do_cull, uint8_t* visibility, int* vis_num, int64_t* vis_ids, int A,
                    void* ws1, long long ws1_bytes, int* host_feedback_vis, int* host_feedback_total,
                    void* cull_scratch /*nullable: persistent, lg_fused_cull_scratch_bytes(chunks), zeroed once*/, uint32_t cull_epoch /*1,2,3,.. per call*/,
                    const int* sched_cull /*nullable: the frame's depth-bound block of its previous visit -> depth-bound culling (fused.hip)*/,
                    int* sched_out /*nullable: the depth-bound block the coming lg_fused_stage2 fills; its head is cleared here*/,
                    void* stream);
long long lg_fused_cull_scratch_bytes(int chunks);
int lg_fused_stage2(const LgFusedCtx* ctx, int A, int S, long long L, int H, int W, int TH, int TW, void* ws1, long long ws1_bytes,
                    void* ws2, long long ws2_bytes, const int* tiles, int K, int enable_stat,
                    float* img, float* trans, short* last, int* frag_count, float* frag_weight,
                    float* packed_grad_clear /*nullable [N,16]: zeroed on the side for the coming lg_fused_backward*/,
                    const int* order /*nullable int32[T]: the frame's heaviest-first tile schedule (raster.hip)*/,
                    int* order_out /*nullable int32[T], may alias order: the schedule recomputed from this visit (one more launch)*/,
                    const int* sched_in /*nullable: the frame's depth-bound block of its previous visit*/,
                    int* sched_out /*nullable: this visit's depth-bound block, lg_sched_words() words*/,
                    int cull_active /*stage 1 culled against sched_in: verify, and enqueue the gated fallback*/,
                    long long L_cull /*table length the culled run is sized for (<= L)*/,
                    int* host_feedback_full /*nullable pinned int: full table length, written when the fallback ran*/,
                    const float* view_host, const float* proj_host, int degree, int chunks,
                    const float* pos, const float* scale, const float* rot, const float* sh0, const float* shr, const float* opa,
                    const int64_t* vis_ids, const int* vis_num, void* stream);
long long lg_sched_words(int H, int W, int TH, int TW);   /* 32-bit words of a per-frame depth-bound block (csrc/lg_tilewalk.h) */
long long lg_fused_flags_offset(long long N);   /* int32[2] in workspace 1: {fallback flag of the last stage 2, full table length if it ran} */
int lg_fused_backward(const LgFusedCtx* ctx, int A, int S, long long L, int H, int W, int TH, int TW, const void* ws1, long long ws1_bytes,
                      const void* ws2, long long ws2_bytes, const float* view_host, const float* proj_host, int degree, int chunks, int R,
                      const int64_t* vis_ids, const int* vis_num,
                      const float* pos, const float* scale, const float* rot, const float* opa,
                      const int* tiles, int K, const float* final_T, const short* last, const float* d_img, const float* d_trans,
                      const float* grad_inv_scaler, int enable_stat, float* packed_grad, int packed_grad_is_zero, float* err_square_sum,
                      float* d_pos, float* d_scale, float* d_rot, float* d_sh0, float* d_shr, float* d_opa,
                      const int* order /*nullable int32[T]: the frame's tile schedule*/, void* stream);
int lg_fused_backward_adam(const LgFusedCtx* ctx, const int* hot_of /*nullable: this frame's replica assignment*/, int A, int S, int H, int W, const float* view_host, const float* proj_host, int degree, int chunks, int R,
                           const int64_t* vis_ids, const int* vis_num, const float* packed_grad, const float* grad_inv_scaler,
                           float* pos, float* scale, float* rot, float* sh0, float* shr, float* opa,
                           float* m_pos, float* m_scale, float* m_rot, float* m_sh0, float* m_shr, float* m_opa,
                           float* v_pos, float* v_scale, float* v_rot, float* v_sh0, float* v_shr, float* v_opa,
                           const float* lr6, float b1, float b2, float eps,
                           unsigned char* touched /* nullable uint8[chunks*S] owned by the optimizer: 0 = all Adam moments of that Gaussian are
                                                     zero; such Gaussians are skipped (exactly a no-op) while their blend moments are zero, and
                                                     flagged on their first non-zero record */,
                           const int* emitted /* nullable int32[A*S]: this frame's tile counts (workspace 1, lg_fused_alloc_offset): a splat that
                                                 was not emitted has an all-zero record, which then need not be read */, void* stream);
int lg_adam_update_multi(int ngroups, void* const* param, const void* const* grad, void* const* exp_avg, void* const* exp_avg_sq,
                         const int* rows, const float* lr, const int64_t* visible_chunk_id, const int* valid_length,
                         int chunks, int A, int S, int grad_dense, float b1, float b2, float eps, void* stream);

/* ---- dp.hip : data-parallel gradient exchange on the device (the reference is single-GPU only).  The ranks exchange the
 * blend backward's moment records (global Gaussian index + 9 floats per touched Gaussian) instead of parameter gradients; every rank
 * replays the per-Gaussian chain backward of every rank's records with that rank's camera, averages and applies Adam in one kernel.
 * A block is float[(1 + cap) * lg_dp_record_floats()]: a header of lg_dp_record_floats() words (word 0 = number of touched Gaussians, int bits)
 * followed by lg_dp_record_floats() rows of cap words: row 0 the global Gaussian indices (int bits), rows 1..9 the nine moments. */
int lg_dp_record_floats(void);
int lg_dp_compact_moments(const float* packed_grad /*[A*S,16] of lg_fused_backward (+ replica lines when hot_of is given)*/, const int64_t* vis_ids,
                          const int* vis_num, int A, int S, int cap, float* block,
                          const int* hot_of /*nullable: the frame's replica assignment (workspace 1 + lg_fused_hot_offset): folded into the records*/,
                          int* hot_counter /*required with hot_of: the renderer's replica line counter, reset here for the next frame*/, void* stream);
int lg_dp_build_slotmap(const float* gathered /*[W] blocks*/, int W, int cap, long long total /*chunks*S*/, int* slot /*[W][total], zero*/,
                        int* host_max_k /*nullable pinned int[2]: {largest count of the job, the same count if it exceeded cap (records were dropped) else 0}*/, int* overflow /*nullable device flag*/, void* stream);
int lg_dp_backward_adam(const int64_t* union_ids, const int* union_count, int chunks, int S, int H, int W_img,
                        const float* views_host /*[W][16]*/, const float* projs_host /*[W][16]*/, int world, int degree, int R,
                        const float* gathered, int cap, int* slot,
                        float* pos, float* scale, float* rot, float* sh0, float* shr, float* opa,
                        float* m_pos, float* m_scale, float* m_rot, float* m_sh0, float* m_shr, float* m_opa,
                        float* v_pos, float* v_scale, float* v_rot, float* v_sh0, float* v_shr, float* v_opa,
                        const float* lr6 /*host: xyz, sh_0, sh_rest, opacity, scale, rot*/, float b1, float b2, float eps,
                        unsigned char* touched /*nullable uint8[chunks*S], as in lg_fused_backward_adam*/, void* stream);
/* The same three steps under SPECULATIVE depth-bound culling across ranks (no gated repeat of a culled frame; litegs_amd/dp.py
 * "rank-consistent speculation").  poison: the renderer's sticky device word (LgFusedCtx.poison): a rank whose culled forward failed
 * sends header word 1 = 1 with its records; the slot map of EVERY rank derives the step's verdict from the same gathered headers --
 * some rank failed, or some block overflowed (then a failed step, not an error) -- raises its own poison word and writes
 * status_host[0] = step_id, status_host[1] = bit r: rank r's flag | bit 8: overflow (a function of the headers only); the backward +
 * Adam launch of a poisoned replica changes nothing, any other records step_id in applied_host.  All pointers nullable = the plain forms. */
int lg_dp_compact_moments_spec(const float* packed_grad, const int64_t* vis_ids, const int* vis_num, int A, int S, int cap, float* block,
                               const int* hot_of, int* hot_counter, const int* poison, void* stream);
int lg_dp_build_slotmap_spec(const float* gathered, int W, int cap, long long total, int* slot, int* host_max_k, int* overflow,
                             int* poison, int* status_host /*pinned int[2]*/, int step_id, void* stream);
int lg_dp_backward_adam_spec(const int64_t* union_ids, const int* union_count, int chunks, int S, int H, int W_img,
                             const float* views_host, const float* projs_host, int world, int degree, int R,
                             const float* gathered, int cap, int* slot,
                             float* pos, float* scale, float* rot, float* sh0, float* shr, float* opa,
                             float* m_pos, float* m_scale, float* m_rot, float* m_sh0, float* m_shr, float* m_opa,
                             float* v_pos, float* v_scale, float* v_rot, float* v_sh0, float* v_shr, float* v_opa,
                             const float* lr6, float b1, float b2, float eps, unsigned char* touched,
                             const int* poison, int* applied_host /*pinned int*/, int step_id, void* stream);

/* ---- knn.hip : simple_knn._C.distCUDA2 (litegs/submodules/simple-knn/simple_knn.cu:186-222; caller litegs/scene/point.py:8) --------
 * mean squared distance of every point to its 3 nearest neighbours (exact).  points [P,3] fp32, mean_dist2 [P]. */
long long lg_knn3_temp_bytes(int P);
int lg_knn3_mean_dist2(const float* points, int P, float* mean_dist2, void* temp, long long temp_bytes, void* stream);

/* ---- refine.hip : litegs/scene/point.py:29-154 (_gen_morton_code, spatial_refine) and the column gathers of
 * litegs/training/densify.py:75-101 (_prune_optimizer) ---------------------------------------------------------------------------
 * lg_morton_order: 3x21-bit Morton codes of xyz [3,n] (bounding-box normalised, the reference's fp32 arithmetic) and their STABLE
 * ascending argsort (torch.sort(stable=True) of point.py:94).  codes may be NULL.  n < 2^31.
 * lg_permute_columns: dst[r,i] = src[r,order[i]] -- the per-tensor gather applied to every parameter, gradient and Adam moment. */
long long lg_morton_order_temp_bytes(long long n);
int lg_morton_order(const float* xyz, long long n, int64_t* codes, int32_t* order, void* temp, long long temp_bytes, void* stream);
int lg_permute_columns(const float* src, const int32_t* order, long long rows, long long n_src, long long n_dst, float* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LITEGS_HIP_H */
