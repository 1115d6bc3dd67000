// Library-internal entry points of binning.hip used by the fused executor (fused.hip): same kernels as the C ABI, but with
// the scratch clearing ("zero duty") and the radix digit counts folded into the producer kernels instead of separate launches.
// Not part of include/litegs_hip.h.
#pragma once
#include <stdint.h>

// Digit totals are accumulated in LG_SORT_TOTALS_COPIES interleaved copies (workgroup b adds into copy b % COPIES, the sort sums them):
// a thousand persistent workgroups flushing 512 counters each into ONE copy serialise on its 32 cache lines at the end of the launch.
#define LG_SORT_TOTALS_COPIES 8
#define LG_SORT_TOTALS_STRIDE (4 * 256)
#define LG_SORT_HEADER_INTS (LG_SORT_TOTALS_COPIES * LG_SORT_TOTALS_STRIDE + 64)      // totals[copies][4][256] | ticket[4] | pad

// words of look-back status a prepared sort of n keys with `passes` passes needs (zero on entry)
long long lg_radix_table_words(long long n, int passes);

// keys/vals of the depth sort + the digit counts of all four passes into header[0..1023]; header must be zero on entry
int lg_depth_keys_hist(const float* depth, long long n, uint32_t* keys, uint32_t* vals, int* header, void* stream);

// radix sort whose header (totals filled, tickets zero) and status table (zero) were prepared by earlier kernels.
// aux_in/aux_sorted (nullable): the last pass also writes aux_sorted[g] = aux_in[sorted value at g] (a gather in sorted order)
int lg_radix_sort_prepared(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, long long n, const int* n_dev,
                           int begin_bit, int end_bit, int* header, uint32_t* table, const int32_t* aux_in, int32_t* aux_sorted, void* stream);
int lg_radix_sort_prepared_values(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, long long n, const int* n_dev,
                                  int begin_bit, int end_bit, int* header, uint32_t* table, const int32_t* aux_in, int32_t* aux_sorted,
                                  int value_bits /*0: unknown; else every value < 2^value_bits*/, int32_t* range_out /*nullable*/, int max_tile,
                                  int* ranges_done /*nullable*/, int* zero32 /*nullable*/, void* stream);

// big-splat queue: 64 sub-queue counters per view (zero on entry) and lg_dup_queue_entries(N, table_len) uint32 entries per view
long long lg_dup_queue_entries(long long N, long long table_len);

// duplicate_with_keys; totals (nullable) receives the digit counts of the emitted keys for a
// sort on bits [begin_bit, end_bit); zero_ptr/zero_words: scratch cleared on the side
int lg_dup_emit(const float* ndc, const float* inv_cov, const float* opacity, const float* packed /*nullable: [V*N][16] records instead of the SoA*/,
                const int32_t* prefix, const void* sorted_id,
                int sorted_id_is_int64, int V, int N, int H, int W, int TH, int TW, long long table_len, int32_t* keys, int32_t* values,
                int* qcount, uint32_t* qentries, int* totals, int begin_bit, int end_bit, uint32_t* zero_ptr, long long zero_words,
                uint32_t* ones_ptr /*filled with 0xffffffff*/, long long ones_words,
                uint32_t* zero2_ptr /*16-byte aligned, also cleared on the side (the backward's gradient accumulator)*/, long long zero2_words, void* stream);

// the same with a gate (nullable device int: the launches do nothing unless *gate != 0) and a truncation flag (nullable: set when the
// table turns out too short for the prefix sums)
int lg_dup_emit_gated(const float* ndc, const float* inv_cov, const float* opacity, const float* packed, const int32_t* prefix, const void* sorted_id,
                      int sorted_id_is_int64, int V, int N, int H, int W, int TH, int TW, long long table_len, int32_t* keys, int32_t* values,
                      int* qcount, uint32_t* qentries, int* totals, int begin_bit, int end_bit,
                      int* tile_counts /*nullable [V][tiles + 2], zero on entry: instances per key, for lg_tile_scatter_gated*/,
                      uint32_t* zero_ptr, long long zero_words,
                      uint32_t* ones_ptr, long long ones_words, uint32_t* zero2_ptr, long long zero2_words,
                      const int* gate, int* trunc_flag,
                      int* dbg /*nullable pinned debug words: [5] += slots whose walk disagreed with the prefix sums, [6] = last difference*/,
                      int* grp_ticket /*nullable device int, zero on entry: the persistent workgroups take their groups of 256 slots on demand*/,
                      void* stream);
int lg_binning_set_tuning(int key, int value);       // keys 10, 11 of lg_set_tuning (binning.hip)
int lg_fused_set_tuning(int key, int value);         // key 12 (fused.hip)

// grouping by tile without a sort (binning.hip "Tile scatter"): per-key counts -> range table + cursors -> values dropped at their
// tile's cursor.  Order inside a tile is arbitrary: follow with lg_tile_depth_sort_gated(any_order = 1).
int lg_tile_scatter_gated(const int32_t* keys, const int32_t* vals, long long L, const int* n_dev, int max_tile,
                          int* counts /*zero on entry if count_keys, else filled by the emission*/, int count_keys, int* cursor,
                          int32_t* tile_start /*pre-filled with -1*/, int32_t* out_vals, const int* gate, void* stream);

// gathered inclusive scan in one launch; status = lg_scan_status_words(n) zero words; host_total (nullable) = pinned host int
long long lg_scan_status_words(long long n);
int lg_gather_scan_prepared(const int32_t* src, const int32_t* idx, long long n, int32_t* out, uint32_t* status, int* host_total, void* stream);
// mode 0: source words as they are; 1 / 2: culled / full view of tile counts whose sign bit marks a culled splat.  gate (nullable):
// nothing happens unless *gate != 0 (total_out then receives 0).  total_out (nullable): device copy of the last prefix.
int lg_gather_scan_gated(const int32_t* src, const int32_t* idx, long long n, int32_t* out, uint32_t* status, int* host_total,
                         int mode, const int* gate, int* total_out, void* stream);

// per-tile depth sort of the tile-sorted value table (tilesort.hip); gate as above
// any_order: the lists do not arrive in ascending id order (tile scatter): ties in depth are ordered by id explicitly
int lg_tile_depth_sort_gated(int32_t* vals, const int32_t* tile_start, const float* depth /*[V,N] view depths*/, int V, long long L, int N, int ntiles,
                             uint32_t* scratch, int any_order, const int* gate, void* stream);

// tileRange on a table whose output was pre-filled with -1
int lg_tile_range_prefilled(const int32_t* sorted_keys, int V, long long L, const int* n_dev, int max_tile, int32_t* out, void* stream);

// frustum culling that also stores the visible-chunk count into pinned host memory (nullable)
int lg_frustum_culling_fb(const float* origin, const float* ext, const float* planes, int V, int M, uint8_t* visibility, int* visible_num,
                          int64_t* visible_chunk_id, int* host_feedback, void* stream);

// multi-workgroup ordered culling with a persistent, epoch-tagged look-back table (compact.hip)
long long lg_cull_scratch_bytes(int M);
int lg_frustum_culling_chain(const float* origin, const float* ext, const float* planes, int V, int M, uint8_t* visibility, int* visible_num,
                             int64_t* visible_chunk_id, void* scratch, unsigned int epoch, int* host_feedback, void* stream);

// blend forward reading / filling the per-frame depth-bound blocks of lg_tilewalk.h (raster.hip; see fused.hip "depth-bound culling")
int lg_raster_forward_bounds(const int* sorted_points, const int* start_index, const float* packed, const int* tiles, int K,
                             int V, long long L, int N, int H, int W, int TH, int TW, int enable_stat,
                             float* img, float* trans, short* last, int* frag_count, float* frag_weight,
                             const int* order, int* tile_work, const int* sched_in, int* sched_out, int zb_check, int* fail_flag,
                             int* fail_host /*nullable pinned mirror: receives 1 when fail_flag is raised*/, const int* gate, void* stream);

// Segmented blend backward (raster.hip): the buffers of one frame, one allocation.  ckpt: ((L >> shift) + 2) checkpoint records of 2 KB;
// ckpt_fin: (tiles + 1) records; counts: 32 words (17 used: full segments, 16 length classes); units: cap_full + 16 * cap_class words
// (tile | segment << 16: the full segments' region, then one region per length class of the remainders).
struct LgSegLayout { size_t ckpt_fin, counts, units, total; int cap_full, cap_class; };
#if defined(__HIPCC__)
__host__ __device__
#endif
inline LgSegLayout lg_seg_layout(long long L, int ntiles, int shift)
{
    LgSegLayout f;
    f.cap_full = (int)((L >> shift) + 1);
    f.cap_class = ntiles;
    f.ckpt_fin = 2048 * ((size_t)(L >> shift) + 2);
    f.counts = f.ckpt_fin + 2048 * ((size_t)ntiles + 1);
    f.units = f.counts + 128;
    f.total = f.units + 4 * ((size_t)f.cap_full + 16 * (size_t)f.cap_class);
    return f;
}
struct LgSegments { char* base; long long L; int ntiles; int shift; int counts_zeroed /*1: an earlier kernel of the stream cleared the unit counters*/; };
int lg_raster_segments_apply(int V, int TH, int TW, int enable_stat, const int* tiles, const void* sched, const void* fail, const void* gate, const void* d_trans);
int lg_raster_segment_shift();
int lg_raster_forward_segments(const int* sorted_points, const int* start_index, const float* packed, const int* tiles, int K,
                               int V, long long L, int N, int H, int W, int TH, int TW, int enable_stat,
                               float* img, float* trans, short* last, int* frag_count, float* frag_weight,
                               const int* order, int* tile_work, const int* sched_in, int* sched_out, int zb_check, int* fail_flag,
                               int* fail_host, const int* gate, const LgSegments* seg, void* stream);
int lg_raster_backward_segments(const int* sorted_points, const int* start_index, const float* packed, const int* tiles, int K,
                                const float* final_T, const short* last, const float* d_img, const float* d_trans,
                                int V, long long L, int N, int H, int W, int TH, int TW, int enable_stat,
                                float* packed_grad, float* err_square_sum, int* tile_counters, const int* order,
                                const int* hot_of, long long hot_lines, const LgSegments* seg, void* stream);

// blend backward with gradient replicas for the splats that cover many tiles (raster.hip)
int lg_raster_backward_hot(const int* sorted_points, const int* start_index, const float* packed, const int* tiles, int K,
                           const float* final_T, const short* last, const float* d_img, const float* d_trans,
                           int V, long long L, int N, int H, int W, int TH, int TW, int enable_stat,
                           float* packed_grad, float* err_square_sum, int* tile_counters, const int* order,
                           const int* hot_of, long long hot_lines, void* stream);
