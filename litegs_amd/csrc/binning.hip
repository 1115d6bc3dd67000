// Tile binning: exact ellipse/tile intersection count, key/value emission in depth order, the
// hand-written stable LSD radix sort ("the tile radix sort"), prefix scan and tile range table
// (SURVEY.md 8a rows a8-a11).  Integer/index work: outputs are BIT-EXACT against the CPU oracle,
// which is why this file is compiled with -ffp-contract=off and uses lg_logf (fixed polynomial).
//
// HBM traffic per tile instance (I of them): 8 B written by duplicate_with_keys, then per radix pass 8 B read + 8 B
// written; 13 tile-id bits at 1080p = 2 passes of 8 bits.  No MFMA: this is byte shuffling.
#include "lg_common.h"
#include <stdlib.h>
#include "lg_tilewalk.h"
#include "lg_binning_internal.h"
#include "lg_sanity.h"

LG_DEFINE_SANITY_COLLECT(binning)

#define TPB 256

// a8 get_allocate_size
template <int TH, int TW>
__global__ void __launch_bounds__(TPB) get_allocate_size_kernel(const float* __restrict__ ndc, const float* __restrict__ view_z,
                                                                const float* __restrict__ inv_cov, const float* __restrict__ opacity,
                                                                const int* __restrict__ valid_length, int N, int H, int W, int gx, int gy,
                                                                int32_t* __restrict__ left_up, int32_t* __restrict__ right_down,
                                                                int32_t* __restrict__ alloc)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    int b = blockIdx.y;
    if (i >= N) return;
    size_t ao = (size_t)b * N + i;
    if (i >= lg_valid_len(valid_length, N)) { alloc[ao] = 0; return; }
    float nx = ndc[((size_t)b * 4) * N + i], ny = ndc[((size_t)b * 4 + 1) * N + i];
    float a = inv_cov[((size_t)b * 4) * N + i], bb = inv_cov[((size_t)b * 4 + 1) * N + i], c = inv_cov[((size_t)b * 4 + 3) * N + i];
    float o = opacity[i];
    float disc = bb * bb - a * c;
    bool vis = !((nx < -1.3f) || (nx > 1.3f) || (ny < -1.3f) || (ny > 1.3f) || (view_z[ao] <= 0.2f) || (o < 1.0f / 255));
    vis = vis && (a > 0) && (c > 0) && (disc < 0);
    size_t l0 = ((size_t)b * 2) * N + i, l1 = ((size_t)b * 2 + 1) * N + i;
    if (!vis) {
        if (left_up) { left_up[l0] = -1; left_up[l1] = -1; right_down[l0] = -1; right_down[l1] = -1; }
        alloc[ao] = 0;
        return;
    }
    SplatExtent e;
    splat_extent<TH, TW>(nx, ny, a, bb, c, o, H, W, gx, gy, e);
    if (left_up) {
        left_up[l0] = lg_f2i(ceilf(e.bbox_min_x)); left_up[l1] = lg_f2i(ceilf(e.bbox_min_y));
        right_down[l0] = lg_f2i(floorf(e.bbox_max_x)); right_down[l1] = lg_f2i(floorf(e.bbox_max_y));
    }
    int n = 0;
    if ((e.rmaxy - e.rminy) * (e.rmaxx - e.rminx) > 0) n = (int)walk_tiles<TH, TW, false>(e, gx, i, 0, nullptr, nullptr);
    alloc[ao] = n;
}

LG_API int lg_get_allocate_size(const float* ndc, const float* view_z, const float* inv_cov, const float* opacity,
                                const int* valid_length, int V, int N, int H, int W, int TH, int TW,
                                int32_t* left_up, int32_t* right_down, int32_t* alloc, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(ndc, view_z, inv_cov, opacity, alloc);
    int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    dim3 grid(lg_cdiv(N, TPB), V);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_GAS(A_, B_) hipLaunchKernelGGL((get_allocate_size_kernel<A_, B_>), grid, dim3(TPB), 0, s, ndc, view_z, inv_cov, opacity, \
                                              valid_length, N, H, W, gx, gy, left_up, right_down, alloc)
    if (TH == 8 && TW == 16) LAUNCH_GAS(8, 16);
    else if (TH == 16 && TW == 16) LAUNCH_GAS(16, 16);
    else if (TH == 12 && TW == 16) LAUNCH_GAS(12, 16);
    else if (TH == 8 && TW == 8) LAUNCH_GAS(8, 8);
    else return (int)hipErrorInvalidValue;
#undef LAUNCH_GAS
    LG_RETURN_LAST();
}

// a10 (first half) duplicate_with_keys: slot j (depth order) -> point sorted_id[j]; emits at prefix[j-1].
// A 256-thread workgroup owns 256 consecutive depth slots.  Two paths, both bit-identical to walk_tiles<>:
//  * small splats (<= DUP_SMALL_HI tiles while the group's entries fit the LDS buffer, else <= DUP_SMALL): the owning thread runs the serial
//    AccuTile walk, leaving one LDS entry per tile slice (first key) and a bit per slice start; the workgroup then streams the
//    positions out -- owner thread and slice by bitmap + popcount / leading-zero count, key = first key + distance * stride -- so
//    global stores are coalesced instead of 64 scattered 4-byte stores per wave instruction and the walk has no per-tile loop;
//  * big splats (near-camera Gaussians can touch thousands of tiles) are queued and emitted by a second launch whose
//    persistent waves take one queue entry at a time (3 % of the 590 k visible splats hold a quarter of the instances at 3 M):
//    one lane per tile slice computes that slice's [min_tile_v, max_tile_v) independently (the serial walk's
//    carried intersections are pure functions of the slice index, see slice_bounds), a wave scan turns the
//    slice counts into offsets, and the 64 lanes then write the splat's contiguous output range in
//    256-byte coalesced stores.  Without this, one thread serialises a 16 200-tile splat and the launch
//    waits for it (measured: 2.3 ms of a 4.9 ms training step at 3 M Gaussians).
#ifndef LG_DUP_SMALL_WAVES
#define LG_DUP_SMALL_WAVES __attribute__((amdgpu_waves_per_eu(5)))
#endif
#ifndef LG_DUP_BIG_WAVES
#define LG_DUP_BIG_WAVES __attribute__((amdgpu_waves_per_eu(4)))
#endif
#define DUP_SMALL 32
#define DUP_SMALL_HI 256
#define DUP_LDS_ENTRIES (TPB * DUP_SMALL)
#define DUP_MAX_SLICES 256
#define DUP_MAX_RUN 32768      // tiles one splat may touch on the cooperative path (bitmap of 4 KiB per wave)
#define SORT_MAX_PASSES_DUP 4
// Big splats are appended to DUP_NQ sub-queues (group g of 256 depth slots -> sub-queue g % DUP_NQ) with ONE returning atomic
// per group: returning atomics on a single address serialise at ~8 ns each in L2, and one counter for the whole launch
// (10 k wave-level appends at 3 M Gaussians) cost 85 us -- more than the rest of the kernel.
#define DUP_NQ 64
#define DUP_GRP_BATCH 1        // groups of 256 slots a workgroup of dup_small takes per ticket (see the kernel; 2 measured slower: fewer, longer rounds)
// A queue entry is (depth slot << 8 | part): a splat with more than DUP_PART tiles is emitted in parts of DUP_PART outputs by different
// waves (every part recomputes the slices, which is cheap next to 1024 outputs) -- otherwise the launch waits for the one wave that
// owns the largest splat (11 033 tiles at 500 k Gaussians: 172 store instructions in a row).  Sub-queue capacity: one entry per slot
// of its groups plus one per DUP_PART outputs of the whole table.
#define DUP_PART 1024
#define DUP_MAX_PARTS 255
__host__ __device__ static inline int dup_num_parts(int cnt) { int p = (cnt + DUP_PART - 1) / DUP_PART; return p < 1 ? 1 : (p > DUP_MAX_PARTS ? DUP_MAX_PARTS : p); }
__host__ __device__ static inline long long dup_queue_cap(long long N, long long L)
{
    return ((N + TPB - 1) / TPB + DUP_NQ - 1) / DUP_NQ * TPB + (L + DUP_PART - 1) / DUP_PART + TPB;
}

// (WalkFrame / walk_frame / slice_bounds: lg_tilewalk.h)

// ---- helpers shared by the producers that feed the radix sort -------------------------------------------------
// Radix digit counts of the keys a kernel emits, accumulated in an LDS table h[passes][RADIX] (flushed to the sort's `totals` at
// the end of the workgroup).  Keys emitted by one wave instruction are often equal in their high digits (neighbouring tiles,
// similar depths): when the whole wave agrees on a digit, one lane adds the wave's count instead of 64 same-address atomics.
struct DigitSpec { int begin_bit, passes; uint32_t last_mask; };
// The LDS table is indexed through a bijection of the digit: tile keys emitted by one wave instruction step by the grid width (120 at
// 1080p), whose digits land on only 4 of the 32 LDS banks; d ^ (d >> 3) spreads any power-of-two-strided digit sequence over all banks.
#define HPERM(d) ((d) ^ ((d) >> 3))
__device__ __forceinline__ void digit_hist_flush(const int* __restrict__ h, int* __restrict__ totals, int passes)
{
    totals += (blockIdx.x % LG_SORT_TOTALS_COPIES) * LG_SORT_TOTALS_STRIDE;
    for (int k = threadIdx.x; k < passes * 256; k += blockDim.x) {
        const int v = h[(k & ~255) + HPERM(k & 255)];
        if (v) atomicAdd(&totals[k], v);
    }
}

__device__ __forceinline__ void digit_hist_add(int* __restrict__ h, uint32_t key, bool active, const DigitSpec& ds)
{
    const unsigned long long m = __ballot(active);
    if (m == 0ull) return;
    const int leader = __ffsll((long long)m) - 1;
    // low digits are effectively random across a wave: plain LDS atomics.  Only the top digit(s) get the wave-uniform shortcut.
    const int first_uniform = ds.passes >= 3 ? ds.passes - 2 : ds.passes - 1;
    for (int p = 0; p < ds.passes; p++) {
        const uint32_t d = (key >> (ds.begin_bit + p * 8)) & ((p == ds.passes - 1) ? ds.last_mask : 255u);
        if (p < first_uniform) {
            if (active) atomicAdd(&h[p * 256 + HPERM(d)], 1);
            continue;
        }
        const uint32_t d0 = (uint32_t)__shfl((int)d, leader);
        if (__ballot(active && d != d0) == 0ull) {
            if ((int)(threadIdx.x & 63) == leader) atomicAdd(&h[p * 256 + HPERM(d0)], __popcll(m));
        } else if (active) {
            atomicAdd(&h[p * 256 + HPERM(d)], 1);
        }
    }
}

// "zero duty": a producer kernel clears a later kernel's scratch (look-back status words, counters) on the side, which
// removes the separate fill launches (each costs ~5 us of dispatch latency on the critical path).
__device__ __forceinline__ void zero_duty(uint32_t* __restrict__ p, long long words, long long gid, long long nthreads)
{
    for (long long i = gid; i < words; i += nthreads) p[i] = 0u;
}

// The six floats the tile walk needs.  SoA (operator path: separate ndc / inv_cov / opacity tensors, six 4-byte gathers = six
// cache lines per splat) or the fused executor's 64-byte packed record (one line): at 3 M Gaussians the SoA gathers alone move
// ~450 MB of cache lines for 14 MB of useful data.
struct SplatSrc { const float* ndc; const float* inv_cov; const float* opacity; const float4* packed; };

template <bool PACKED>
__device__ __forceinline__ void load_splat(const SplatSrc& src, size_t b, int N, int idx, float& nx, float& ny, float& a, float& bb,
                                           float& cc, float& o)
{
    if (PACKED) {       // record layout: raster.hip / fused.hip (slot 5 opacity, 9..11 inverse covariance, 13..14 ndc)
        const float4* r = src.packed + ((size_t)b * N + idx) * 4;
        const float4 r1 = r[1], r2 = r[2], r3 = r[3];
        o = r1.y; a = r2.y; bb = r2.z; cc = r2.w; nx = r3.y; ny = r3.z;
    } else {
        nx = src.ndc[((size_t)b * 4) * N + idx]; ny = src.ndc[((size_t)b * 4 + 1) * N + idx];
        a = src.inv_cov[((size_t)b * 4) * N + idx]; bb = src.inv_cov[((size_t)b * 4 + 1) * N + idx]; cc = src.inv_cov[((size_t)b * 4 + 3) * N + idx];
        o = src.opacity[idx];
    }
}

// debug_words[5] = emission slots whose walk disagreed with the projection's count, [6] = the last one's (walked - counted); word 0
// is left to the table validators (fused.hip).  dbg nullable (product runs): the emission pads / drops silently.
__device__ __forceinline__ void dup_report_mismatch(int* __restrict__ dbg, int slot, int walked, int counted)
{
    lg_note_sanitised(LG_SITE_EMIT_COUNT);
    if (dbg == nullptr) return;
    __hip_atomic_fetch_add(dbg + 5, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dbg + 6, walked - counted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    (void)slot;
}

// Kernel 1 of duplicate_with_keys: one thread per depth slot.  Small splats are walked serially into a compacted LDS buffer of
// keys and streamed out coalesced; big splats (> DUP_SMALL tiles) are only QUEUED for kernel 2.
// LdsKeyT: uint16_t when every tile id + 1 fits 16 bits (anything up to ~8 MPixel at 8x16 tiles) -- halves the staging buffer, one
// more workgroup per CU for this latency-bound kernel.
template <int TH, int TW, typename IdxT, bool PACKED, typename LdsKeyT>
__global__ void __launch_bounds__(TPB) LG_DUP_SMALL_WAVES dup_small_kernel(SplatSrc src, const int32_t* __restrict__ prefix,
                                                        const IdxT* __restrict__ sorted_id, int N, int H, int W, int gx, int gy,
                                                        long long table_len, int32_t* __restrict__ keys, int32_t* __restrict__ values,
                                                        int* __restrict__ qcount /*[V][DUP_NQ], zero*/, uint32_t* __restrict__ qentries /*[V][DUP_NQ][cap]*/,
                                                        int* __restrict__ totals /*nullable [passes][256]*/, DigitSpec ds,
                                                        int* __restrict__ tile_counts /*nullable [V][gx*gy+2], zero: instances per key (tile scatter)*/,
                                                        uint32_t* __restrict__ zero_ptr, long long zero_words,
                                                        uint32_t* __restrict__ ones_ptr, long long ones_words,
                                                        uint32_t* __restrict__ zero2_ptr, long long zero2_words,
                                                        const int* __restrict__ gate, int* __restrict__ trunc_flag, int* __restrict__ dbg,
                                                        int small_hi /*largest tile count walked in the workgroup (<= DUP_SMALL_HI)*/,
                                                        int* __restrict__ grp_ticket /*nullable, zero on entry: groups are handed out dynamically*/,
                                                        int static_rounds /*>= 1: rounds dealt round robin before the tickets start*/)
{
    if (gate != nullptr && *gate == 0) return;            // fallback launch of the depth-bound culling that is not needed (fused.hip)
    __shared__ LdsKeyT buf[DUP_LDS_ENTRIES];              // 16/32 KiB: compacted keys of the small splats
    __shared__ int grp_s;
    __shared__ int t_loff[TPB + 1];                       // per-thread start in buf
    __shared__ int t_goff[TPB];                           // per-thread start in the table
    __shared__ int t_idx[TPB];                            // per-thread point id
    __shared__ int hist[SORT_MAX_PASSES_DUP * 256];
    __shared__ int wsum[TPB / 64];
    __shared__ int wbig[TPB / 64];
    __shared__ int wnz[TPB / 64];
    __shared__ int qbase_s;
    // owner lookup of the stream-out: bit p of `starts` = some thread's entries begin at buf[p]; the r-th set bit belongs to c_tid[r]
    // (one 64-bit LDS broadcast + a popcount per output instead of an 8-step binary search over t_loff)
    __shared__ unsigned long long starts[DUP_LDS_ENTRIES / 64 + 4];
    __shared__ int c_tid[TPB];
    // the walk leaves one entry per tile SLICE (first key at the slice's first position, bit in `sstarts`); the stream-out rebuilds
    // key = first + (position - slice start) * stride -- no per-tile loop in the serial, divergent walk
    __shared__ unsigned long long sstarts[DUP_LDS_ENTRIES / 64 + 4];
    __shared__ int t_stride[TPB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const long long qcap = dup_queue_cap(N, table_len);
    const int32_t* pf = prefix + (size_t)b * N;
    int32_t* kout = keys + (size_t)b * table_len;
    int32_t* vout = values + (size_t)b * table_len;
    // Zero duty, one burst of 16-byte stores at the start of the launch: the tile range table (-1), the tile sort's look-back table
    // (17 MB at 23 M instances) and the gradient accumulator of the coming blend backward (64 B per compacted Gaussian: 140 MB late in a
    // run).  (Round 5 tried to issue the two big ones a few stores per group iteration, "under the walk": +26 us on this kernel at 23 M
    // instances -- on gfx9-family parts stores count on vmcnt like loads, so every wait for the walk's loads also waited for the zero
    // stores issued in front of them.  profiles/r05_emission_ab.log.)
    {
        const long long z_gid = ((long long)b * gridDim.x + blockIdx.x) * TPB + tid, z_nth = (long long)gridDim.x * gridDim.y * TPB;
        if (ones_ptr) for (long long i = z_gid; i < ones_words; i += z_nth) ones_ptr[i] = 0xffffffffu;      // tile range table: -1 = empty
        if (zero_ptr) {
            uint4* z4 = reinterpret_cast<uint4*>(zero_ptr);
            const long long n4 = zero_words / 4;
            for (long long i = z_gid; i < n4; i += z_nth) z4[i] = make_uint4(0u, 0u, 0u, 0u);
            for (long long i = n4 * 4 + z_gid; i < zero_words; i += z_nth) zero_ptr[i] = 0u;
        }
        if (zero2_ptr) {
            uint4* z4 = reinterpret_cast<uint4*>(zero2_ptr);
            const long long n4 = zero2_words / 4;
            for (long long i = z_gid; i < n4; i += z_nth) z4[i] = make_uint4(0u, 0u, 0u, 0u);
            for (long long i = n4 * 4 + z_gid; i < zero2_words; i += z_nth) zero2_ptr[i] = 0u;
        }
    }
    if (totals) for (int k = tid; k < ds.passes * 256; k += TPB) hist[k] = 0;
    // persistent workgroups (the digit table is flushed once per workgroup, not once per 256 splats).  Groups are dealt round robin, or
    // -- grp_ticket -- the TAIL of the launch is handed out on demand, in batches of DUP_GRP_BATCH consecutive groups: groups differ in
    // cost by an order of magnitude (in depth order the near groups hold the large splats), which costs the long launches of a
    // density-control run 38 us (profiles/r05_emission_ab.log).  A returning atomic on one address is serialised in L2 (~8 ns each) and --
    // vector memory operations return in order -- its round trip sits in front of the requesting wave's next load: a ticket per group
    // cost the bench's fresh frame (3 groups per workgroup) 36 us, a ticket per batch from the first batch on still 27 us.  So every
    // workgroup's first `static_rounds` batches are static (batch blockIdx.x + r * gridDim.x), and the HOST turns tickets on only for
    // long launches (>= 5 groups per workgroup: lg_dup_emit_gated), where one static round is followed by tickets; short launches stay
    // round robin.  Measured over five schemes (profiles/r05_emission_ab.log): tickets early in a long GLOBAL-route launch -38 .. 0 us,
    // tickets only for the tail +12 us, tickets in a short launch +1.6 .. +36 us, in the long TILE-route launch of the 10 M frame +80 us.
    // Default: off (lg_set_tuning(11, 0)); the mechanism stays for the next look at this kernel.
    const int ngroups = (N + TPB - 1) / TPB;
    int batch = blockIdx.x, in_batch = 0, next_ticket = 0, round = 0;
    int grp = grp_ticket != nullptr ? batch * DUP_GRP_BATCH : (int)blockIdx.x;
    while (grp < ngroups) {
    if (grp_ticket != nullptr && in_batch == 0 && round >= static_rounds - 1 && tid == 0)
        next_ticket = atomicAdd(grp_ticket, 1) + static_rounds * (int)gridDim.x;   // the batch after this one; consumed at the end of this one
    const int j = grp * TPB + tid;
    if (tid < DUP_LDS_ENTRIES / 64 + 4) { starts[tid] = 0ull; sstarts[tid] = 0ull; }       // (barriers below separate this from the bit sets)

    // 1. size of the slot (tile count > 0 <=> non-empty tile rectangle, so no geometry is needed to classify it)
    long long off = 0;
    int cnt = 0, idx = 0;
    if (j < N) {
        off = (j == 0) ? 0 : pf[j - 1];
        const long long c = pf[j] - off;
        if (c > 0 && off + c <= table_len) cnt = (int)c;
        else if (c > 0 && off <= table_len) {           // (off == table_len: nothing left to pad, but the table IS too short -- raise the flag)
            // first splat that does not fit (GR/binning.cu:63 drops it and, prefix being monotone, every later one): the rest of
            // the table becomes key 0 = "no tile".  Values too (the table is not pre-cleared on the fused path), and the padding keys
            // are counted into the sort's digit totals (digit 0 of every pass) -- the sort then handles exactly table_len keys.
            for (long long q = off; q < table_len; q++) { kout[q] = 0; vout[q] = 0; }
            if (trunc_flag) atomicOr(trunc_flag, 1);          // the table was under-predicted: the culled run asks for the fallback
            lg_note_sanitised(LG_SITE_TRUNCATED);             // (counted, not an error: the reference truncates silently)
            if (totals)
                for (int p = 0; p < ds.passes; p++) atomicAdd(&totals[p * 256], (int)(table_len - off));
            if (tile_counts) atomicAdd(&tile_counts[(size_t)b * (gx * gy + 2)], (int)(table_len - off));
        }
    }
    // Threshold between the in-workgroup path and the queue: DUP_SMALL_HI when this group's entries still fit the LDS buffer
    // (the usual case: ~8 tiles per splat on average), DUP_SMALL otherwise (then 256 x 32 entries fit by construction).
    // The largest of DUP_SMALL_HI, /2, /4 ... for which the group still fits (groups of spatial neighbours -- emission in splat-id
    // order -- are all large or all small: halving step by step keeps most of such a group in the in-workgroup path instead of
    // demoting it wholesale to DUP_SMALL).
    const int hi1 = small_hi, hi2 = max(small_hi / 2, DUP_SMALL), hi4 = max(small_hi / 4, DUP_SMALL);
    int thr = hi1;
    {
        int c_hi = (cnt <= hi1) ? cnt : 0, c_h2 = (cnt <= hi2) ? cnt : 0, c_h4 = (cnt <= hi4) ? cnt : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { c_hi += __shfl_xor(c_hi, o); c_h2 += __shfl_xor(c_h2, o); c_h4 += __shfl_xor(c_h4, o); }
        if (lane == 0) { wbig[wave] = c_hi; wnz[wave] = c_h2; wsum[wave] = c_h4; }
        __syncthreads();
        if (wbig[0] + wbig[1] + wbig[2] + wbig[3] > DUP_LDS_ENTRIES) {
            thr = hi2;
            if (wnz[0] + wnz[1] + wnz[2] + wnz[3] > DUP_LDS_ENTRIES) {
                thr = hi4;
                if (wsum[0] + wsum[1] + wsum[2] + wsum[3] > DUP_LDS_ENTRIES) thr = DUP_SMALL;
            }
        }
        __syncthreads();
    }
    const bool small = cnt > 0 && cnt <= thr;
    const bool big = cnt > thr;

    // 2. big splats: reserve their queue slots (one returning atomic per group, issued now, consumed after the geometry below so that
    //    its L2 round trip overlaps the record loads)
    const int np = big ? dup_num_parts(cnt) : 0;           // queue entries of this slot
    int np_inc = np;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int nb = __shfl_up(np_inc, o);
        if (lane >= o) np_inc += nb;
    }
    if (lane == 63) wbig[wave] = np_inc;
    __syncthreads();
    int qb = 0;
    if (tid == 0) {
        const int nent = wbig[0] + wbig[1] + wbig[2] + wbig[3];
        if (nent) qb = atomicAdd(qcount + (size_t)b * DUP_NQ + (grp % DUP_NQ), nent);
    }

    // 3. geometry of the small splats
    SplatExtent e;
    if (small) {
        idx = sorted_id ? (int)sorted_id[(size_t)b * N + j] : j;          // no order given: slots are the splats themselves
        float nx, ny, a, bb, cc, o;
        load_splat<PACKED>(src, b, N, idx, nx, ny, a, bb, cc, o);
        splat_extent<TH, TW>(nx, ny, a, bb, cc, o, H, W, gx, gy, e);
    }
    if (tid == 0) qbase_s = qb;
    __syncthreads();
    if (big) {
        int pos = qbase_s + np_inc - np;
        for (int w = 0; w < wave; w++) pos += wbig[w];
        uint32_t* q = qentries + ((size_t)b * DUP_NQ + (grp % DUP_NQ)) * qcap + pos;
        // (the capacity bound of dup_queue_cap makes an overflow impossible; if it ever happens the entries beyond the sub-queue are
        // dropped and counted -- dup_big refuses the unwritten entries it then finds -- instead of landing in the next sub-queue)
        for (int p = 0; p < np; p++) {
            if ((long long)pos + p < qcap) q[p] = ((uint32_t)j << 8) | (uint32_t)p;
            else lg_note_sanitised(LG_SITE_QUEUE_ENTRY);
        }

    }

    // ---- small splats: exclusive block scan of their counts -> compacted LDS layout ----
    const int scnt = small ? cnt : 0;
    int incl = scnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int nb = __shfl_up(incl, o);
        if (lane >= o) incl += nb;
    }
    const unsigned long long nzb = __ballot(scnt > 0);
    if (lane == 63) wsum[wave] = incl;
    if (lane == 0) wnz[wave] = __popcll(nzb);
    __syncthreads();
    int wbase = 0, rbase = 0;
    for (int w = 0; w < wave; w++) { wbase += wsum[w]; rbase += wnz[w]; }
    const int loff = wbase + incl - scnt;
    if (scnt > 0) {
        c_tid[rbase + __popcll(nzb & ((1ull << lane) - 1ull))] = tid;
        atomicOr(reinterpret_cast<unsigned int*>(starts) + (loff >> 5), 1u << (loff & 31));
    }
    const int total_small = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    t_loff[tid] = loff;
    t_goff[tid] = (int)off;
    t_idx[tid] = idx;
    if (tid == 0) t_loff[TPB] = total_small;
    if (small) {
        t_stride[tid] = ((e.rmaxy - e.rminy) < (e.rmaxx - e.rminx)) ? 1 : gx;       // walk_tiles' isY rule
        // the slot owns buf[loff, loff + cnt): cnt is the projection's count of the same walk over the same six floats.  The walk is
        // nevertheless bounded by it, and the stream-out below writes exactly cnt entries per slot, so the table is fully written and
        // nothing outside the slot's range is touched even if the two counts ever disagree (reported through dbg when validating).
        const int walked = (int)walk_tiles<TH, TW, true, LdsKeyT>(e, gx, idx, loff, (int32_t*)nullptr, (int32_t*)nullptr, buf,
                                                                  reinterpret_cast<unsigned int*>(sstarts), (long long)loff + cnt);
        if (walked != cnt) {
            if (walked <= 0) {                            // no slice start of its own: the slot becomes padding (key 0 = "no tile")
                buf[loff] = (LdsKeyT)0; t_stride[tid] = 0;
                atomicOr(reinterpret_cast<unsigned int*>(sstarts) + (loff >> 5), 1u << (loff & 31));
            }
            dup_report_mismatch(dbg, j, walked, cnt);
        }
    }
    __syncthreads();
    int sbase = 0;                                        // set bits below position p0
    int scarry = 0;                                       // position of the last slice start below p0
    int bad_keys = 0;
    for (int p0 = 0; p0 < total_small; p0 += TPB) {
        const int p = p0 + tid;
        const bool act = p < total_small;
        // the four 64-position blocks of this round (one per wave): uniform addresses, LDS broadcasts
        const unsigned long long q0 = starts[(p0 >> 6)], q1 = starts[(p0 >> 6) + 1], q2 = starts[(p0 >> 6) + 2], q3 = starts[(p0 >> 6) + 3];
        const int c0 = __popcll(q0), c1 = __popcll(q1), c2 = __popcll(q2);
        const unsigned long long word = wave == 0 ? q0 : (wave == 1 ? q1 : (wave == 2 ? q2 : q3));
        const int before = sbase + (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
        sbase += c0 + c1 + c2 + __popcll(q3);
        // same four blocks of the slice-start bitmap; l_k = last slice start at or below the end of block k (uniform)
        const unsigned long long s0 = sstarts[(p0 >> 6)], s1 = sstarts[(p0 >> 6) + 1], s2 = sstarts[(p0 >> 6) + 2], s3 = sstarts[(p0 >> 6) + 3];
        const int l0 = s0 ? p0 + 63 - __clzll(s0) : scarry;
        const int l1 = s1 ? p0 + 127 - __clzll(s1) : l0;
        const int l2 = s2 ? p0 + 191 - __clzll(s2) : l1;
        const unsigned long long sword = wave == 0 ? s0 : (wave == 1 ? s1 : (wave == 2 ? s2 : s3));
        const int sprev = wave == 0 ? scarry : (wave == 1 ? l0 : (wave == 2 ? l1 : l2));
        scarry = s3 ? p0 + 255 - __clzll(s3) : l2;
        int32_t key = 0;
        if (act) {
            // owner = the thread whose start bit is the last one at or below p (threads without entries set no bit)
            const int t = c_tid[before + __popcll(word & ((2ull << lane) - 1ull)) - 1];
            const unsigned long long sm = sword & ((2ull << lane) - 1ull);
            const int ps = sm ? p0 + wave * 64 + 63 - __clzll(sm) : sprev;
            key = (int32_t)buf[ps] + (p - ps) * t_stride[t];
            // cannot happen while walk and count agree; a key is an index downstream.  A select here, the count behind the loop: a branch
            // with an atomic in this loop cost the kernel 15 us per frame (profiles/r05_emission_ab.log)
            const bool oob = (unsigned)key > (unsigned)(gx * gy);
            key = oob ? 0 : key;
            bad_keys += oob ? 1 : 0;
            const int g = t_goff[t] + (p - t_loff[t]);
            kout[g] = key;
            vout[g] = t_idx[t];
            if (tile_counts) atomicAdd(&tile_counts[(size_t)b * (gx * gy + 2) + key], 1);
        }
        if (totals) digit_hist_add(hist, (uint32_t)key, act, ds);
    }
    if (bad_keys) lg_note_sanitised(LG_SITE_EMIT_KEY, bad_keys);
    if (grp_ticket != nullptr && in_batch == DUP_GRP_BATCH - 1 && round >= static_rounds - 1 && tid == 0) grp_s = next_ticket;
    __syncthreads();                                      // buf / t_* are reused by the next group
    if (grp_ticket == nullptr) grp += (int)gridDim.x;
    else if (++in_batch < DUP_GRP_BATCH) grp++;
    else {
        in_batch = 0;
        batch = (round < static_rounds - 1) ? batch + (int)gridDim.x : grp_s;
        round++;
        grp = batch * DUP_GRP_BATCH;
    }
    }

    if (totals) {
        __syncthreads();
        digit_hist_flush(hist, totals, ds.passes);
    }
}

// first-claim of a detail record in the pinned debug words (host memory: system scope)
__device__ __forceinline__ bool dbg_claim(int* word)
{
    int expected = 0;
    return __hip_atomic_compare_exchange_strong(word, &expected, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float bcast_f(float v, int src) { return __shfl(v, src); }
__device__ __forceinline__ int bcast_i(int v, int src) { return __shfl(v, src); }

// Kernel 2: persistent waves drain the queue of big splats, DUP_BATCH at a time: lanes 0..DUP_BATCH-1 each fetch one queued
// splat and compute its extent (so the dependent loads slot -> point id -> record are paid once per batch, not once per splat),
// then the wave emits the batch one splat at a time: one lane per tile slice computes that slice's [min_tile_v, max_tile_v)
// independently (slice_bounds), a wave scan turns the slice counts into offsets and the 64 lanes write the splat's contiguous
// output range in 256-byte coalesced stores.
#define DUP_BATCH 16
template <int TH, int TW, typename IdxT, bool PACKED>
__global__ void __launch_bounds__(TPB) LG_DUP_BIG_WAVES dup_big_kernel(SplatSrc src, const int32_t* __restrict__ prefix,
                                                      const IdxT* __restrict__ sorted_id, int N, int H, int W, int gx, int gy,
                                                      long long table_len, int32_t* __restrict__ keys, int32_t* __restrict__ values,
                                                      const int* __restrict__ qcount, const uint32_t* __restrict__ qentries,
                                                      int* __restrict__ totals, DigitSpec ds, int* __restrict__ tile_counts, const int* __restrict__ gate,
                                                      int* __restrict__ dbg)
{
    if (gate != nullptr && *gate == 0) return;
    __shared__ int w_minv[TPB / 64][DUP_MAX_SLICES];      // per-wave slice scratch
    __shared__ int w_off[TPB / 64][DUP_MAX_SLICES + 1];
    __shared__ int c_idx[TPB / 64][DUP_MAX_SLICES];       // r-th non-empty slice
    __shared__ uint32_t bitmap[TPB / 64][DUP_MAX_RUN / 32 + 2];   // bit k = an output run starts at k; all-zero between splats
    __shared__ int hist[SORT_MAX_PASSES_DUP * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    for (int k = tid; k < (TPB / 64) * (DUP_MAX_RUN / 32 + 2); k += TPB) (&bitmap[0][0])[k] = 0u;
    const int32_t* pf = prefix + (size_t)b * N;
    int32_t* kout = keys + (size_t)b * table_len;
    int32_t* vout = values + (size_t)b * table_len;
    const long long qcap = dup_queue_cap(N, table_len);
    const uint32_t* q = qentries + (size_t)b * DUP_NQ * qcap;
    __shared__ int qstart[DUP_NQ + 1];                     // exclusive prefix of the sub-queue lengths
    if (tid < 64) {
        const int c = qcount[(size_t)b * DUP_NQ + tid];
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int nbv = __shfl_up(inc, o);
            if (lane >= o) inc += nbv;
        }
        qstart[tid] = inc - c;
        if (tid == 63) qstart[64] = inc;
    }
    if (totals) for (int k = tid; k < ds.passes * 256; k += TPB) hist[k] = 0;
    __syncthreads();
    const int nq = qstart[DUP_NQ];
    const int nwaves = gridDim.x * (TPB / 64);
    int bad_keys = 0;
    // entries are dealt round-robin (entry = slot * nwaves + wave): the queues are roughly in depth order and the giant
    // near-camera splats sit together -- contiguous batches would hand all of them to a few waves
    const int gw = blockIdx.x * (TPB / 64) + wave;
    for (int r0 = 0; r0 * nwaves + gw < nq; r0 += DUP_BATCH) {
        const int left = (nq - gw - r0 * nwaves + nwaves - 1) / nwaves;          // entries of this wave from slot r0 on
        const int nb = left < DUP_BATCH ? left : DUP_BATCH;
        SplatExtent e;
        int my_off = 0, my_idx = 0, my_part = 0, my_cnt = 0;
        if (lane < nb) {
            const int t = (r0 + lane) * nwaves + gw;               // flat entry -> (sub-queue, position)
            int lo = 0, hi = DUP_NQ - 1;
            while (lo < hi) {
                int mid = (lo + hi + 1) >> 1;
                if (qstart[mid] <= t) lo = mid; else hi = mid - 1;
            }
            const uint32_t ent = q[(size_t)lo * qcap + (t - qstart[lo])];
            int j = (int)(ent >> 8);
            my_part = (int)(ent & 255u);
            // A queue entry is an index (depth slot -> prefix sums -> splat id -> 64-byte record -> a range of the table).  An entry that
            // dup_small did not write -- stale memory -- must never be followed: slot and part are checked against what the prefix sums say.
            bool ok = j < N;
            if (ok) {
                my_off = (j == 0) ? 0 : pf[j - 1];
                my_cnt = pf[j] - my_off;                           // the slot's share of the table (what dup_small sized the parts by)
                ok = my_cnt > 0 && (long long)my_off + my_cnt <= table_len && my_part < dup_num_parts(my_cnt);
            }
            if (!ok) {
                lg_note_sanitised(LG_SITE_QUEUE_ENTRY);
                if (dbg != nullptr && dbg_claim(dbg + 8)) {          // first bad entry of the run: where it sat and what it held
                    dbg[9] = lo; dbg[10] = t - qstart[lo]; dbg[11] = (int)ent; dbg[12] = qstart[lo + 1] - qstart[lo]; dbg[13] = nq; dbg[14] = (int)qcap; dbg[15] = N;
                }
                j = 0; my_off = 0; my_cnt = 0; my_part = 0;
            }
            my_idx = sorted_id ? (int)sorted_id[(size_t)b * N + j] : j;
            if ((unsigned)my_idx >= (unsigned)N) { my_idx = 0; my_cnt = 0; lg_note_sanitised(LG_SITE_QUEUE_ENTRY); }
            float nx, ny, a, bb, cc, o;
            load_splat<PACKED>(src, b, N, my_idx, nx, ny, a, bb, cc, o);
            splat_extent<TH, TW>(nx, ny, a, bb, cc, o, H, W, gx, gy, e);
        }
        for (int srcl = 0; srcl < nb; srcl++) {
            SplatExtent s;
            s.a = bcast_f(e.a, srcl); s.b = bcast_f(e.b, srcl); s.c = bcast_f(e.c, srcl); s.disc = bcast_f(e.disc, srcl); s.t = bcast_f(e.t, srcl);
            s.px = bcast_f(e.px, srcl); s.py = bcast_f(e.py, srcl);
            s.bbox_min_x = bcast_f(e.bbox_min_x, srcl); s.bbox_min_y = bcast_f(e.bbox_min_y, srcl);
            s.bbox_max_x = bcast_f(e.bbox_max_x, srcl); s.bbox_max_y = bcast_f(e.bbox_max_y, srcl);
            s.argmin_x = bcast_f(e.argmin_x, srcl); s.argmin_y = bcast_f(e.argmin_y, srcl);
            s.argmax_x = bcast_f(e.argmax_x, srcl); s.argmax_y = bcast_f(e.argmax_y, srcl);
            s.rminx = bcast_i(e.rminx, srcl); s.rminy = bcast_i(e.rminy, srcl); s.rmaxx = bcast_i(e.rmaxx, srcl); s.rmaxy = bcast_i(e.rmaxy, srcl);
            const int sidx = bcast_i(my_idx, srcl);
            const int sgoff = bcast_i(my_off, srcl);
            const int part = bcast_i(my_part, srcl);
            const int scnt = bcast_i(my_cnt, srcl);
            if (scnt <= 0) continue;                                       // a neutralised queue entry (above): nothing to emit
            const WalkFrame f = walk_frame<TH, TW>(s);
            const int nsl = f.rect_max_u - f.rect_min_u;                   // <= min(grid.x, grid.y) slices
            if (nsl > DUP_MAX_SLICES) {                                    // > 4K-class images: the owner lane walks serially
                if (part != 0) continue;                                   // (the whole splat, by the wave that holds its first part)
                int c = 0;
                if (lane == srcl) c = (int)walk_tiles<TH, TW, true>(e, gx, my_idx, my_off, kout, vout);
                if (totals || tile_counts) {                               // count what was just written
                    c = __shfl(c, srcl);
                    __threadfence();
                    for (int k0 = 0; k0 < c; k0 += 64) {
                        const bool act = k0 + lane < c;
                        const int32_t key = act ? __hip_atomic_load(kout + sgoff + k0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                        if (totals) digit_hist_add(hist, (uint32_t)key, act, ds);
                        if (tile_counts && act) atomicAdd(&tile_counts[(size_t)b * (gx * gy + 2) + key], 1);
                    }
                }
                continue;
            }
            // K = number of leading slices whose upper line is <= bmax_u
            int K = 0;
            for (int i0 = 0; i0 < nsl; i0 += 64) {
                int i = i0 + lane;
                bool c = (i < nsl) && ((float)(f.rect_min_u + i) * f.BLOCK_U + f.BLOCK_U <= f.bmax_u);
                K += __popcll(__ballot(c));
            }
            // slice i -> (first tile w_minv[i], output offset w_off[i]); the r-th NON-EMPTY slice is c_idx[r] and sets bit w_off[i] of a
            // bitmap over the splat's output range: the owner of output k is then "number of set bits at positions <= k" - 1, i.e.
            // one uniform 64-bit LDS read and a popcount per 64 outputs instead of a binary search per output
            // A slice's tile count max_tile_v - min_tile_v can be NEGATIVE (both of its lines unselected and neither extreme point inside
            // it -- a degenerate, measure-zero configuration of the reference's arithmetic, GR/speedy_splat.cuh:118-125, which ADDS that
            // negative number into the count and emits nothing for the slice).  The slot's share of the table, scnt, is that signed sum
            // (the projection's count); the layout below must be built from max(n, 0): a negative term in the offsets would put a slice
            // start at a negative position (no owner for the first outputs -> an LDS read in front of c_idx -> a garbage key).  The share
            // is then shorter than the tiles the slices hold and the last tiles are dropped, exactly as when the table is truncated.
            int run = 0, nne = 0, run_signed = 0;
            for (int i0 = 0; i0 < nsl; i0 += 64) {
                int i = i0 + lane;
                int mn = 0, n = 0;
                if (i < nsl) {
                    int mx;
                    slice_bounds(s, f, i, K, mn, mx);
                    n = mx - mn;
                }
#ifndef LG_REPRO_NEGATIVE_SLICE_BUG        // (tools/repro_negative_slice.py builds this file once WITHOUT the clamp and the key check below:
                                            //  the round-3 / round-4 memory access fault, profiles/r05_fault_root_cause.md)
                if (__ballot(n < 0) != 0ull) {                 // never in a healthy cloud: one ballot per 64 slices
                    int ng = n < 0 ? n : 0;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) ng += __shfl_xor(ng, o);
                    run_signed += ng;                          // the negative counts; the positive ones are added below (run)
                }
                n = n > 0 ? n : 0;
#endif
                int inc = n;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    int nbv = __shfl_up(inc, o);
                    if (lane >= o) inc += nbv;
                }
                const unsigned long long ne = __ballot(n > 0);
                if (n > 0) {
                    const int off = run + inc - n;
                    w_minv[wave][i] = mn; w_off[wave][i] = off;
                    c_idx[wave][nne + __popcll(ne & ((1ull << lane) - 1ull))] = i;
                    if (off < DUP_MAX_RUN) atomicOr(&bitmap[wave][off >> 5], 1u << (off & 31));
                }
                nne += __popcll(ne);
                run += __shfl(inc, 63);
            }
            __builtin_amdgcn_wave_barrier();      // LDS operations of one wave execute in order: no fence (a fence would also wait for the global stores)
            if (run > DUP_MAX_RUN) {              // cannot happen below ~8K x 8K images; keep the table consistent and walk serially
                for (int wq = lane; wq < DUP_MAX_RUN / 32; wq += 64) bitmap[wave][wq] = 0u;
                if (part != 0) continue;
                if (lane == srcl) walk_tiles<TH, TW, true>(e, gx, my_idx, my_off, kout, vout);
                if (totals || tile_counts) {
                    __threadfence();
                    for (int k0 = 0; k0 < run; k0 += 64) {
                        const bool act = k0 + lane < run;
                        const int32_t key = act ? __hip_atomic_load(kout + sgoff + k0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                        if (totals) digit_hist_add(hist, (uint32_t)key, act, ds);
                        if (tile_counts && act) atomicAdd(&tile_counts[(size_t)b * (gx * gy + 2) + key], 1);
                    }
                }
                continue;
            }
            // this wave's part of the output range (whole splat when it has DUP_PART outputs or fewer).  Parts and range follow the
            // slot's share of the table (scnt, from the prefix sums -- the projection's count), not the count of this kernel's own walk:
            // the two are the same function of the same six floats, but every entry of [off, off + scnt) is written whatever happens
            // (tiles beyond scnt dropped, entries beyond run padded with key 0 = "no tile"), because an entry left unwritten is stale
            // memory that the sort carries to the range scan, and a word of garbage there is a wild store.
            run_signed += run;                                 // signed sum of the slice counts = what the projection counted
            if (run_signed != scnt && part == 0 && lane == 0) dup_report_mismatch(dbg, -1 - sidx, run_signed, scnt);
            const int nparts = dup_num_parts(scnt);
            const int k_begin = part * DUP_PART;
            const int k_end = (part == nparts - 1) ? scnt : (k_begin + DUP_PART < scnt ? k_begin + DUP_PART : scnt);
            int before = 0;                       // non-empty slices that start before k_begin
            if (k_begin > 0) {
                const int kb = k_begin < run ? k_begin : run;
                for (int wq = lane; wq < (kb >> 5); wq += 64) before += __popc(bitmap[wave][wq]);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
            }
            for (int k0 = k_begin; k0 < k_end; k0 += 64) {
                const int k = k0 + lane;
                const bool act = k < k_end;
                const int bw = (k0 < run ? k0 : run) >> 5;         // (run <= DUP_MAX_RUN here: inside the bitmap)
                const uint32_t wlo = bitmap[wave][bw], whi = bitmap[wave][bw + 1];
                const unsigned long long word = ((unsigned long long)whi << 32) | wlo;
                int32_t key = 0;
                if (act && k >= run) {                             // padding (only if the counts disagree)
                    kout[sgoff + k] = 0;
                    vout[sgoff + k] = 0;
                    if (tile_counts) atomicAdd(&tile_counts[(size_t)b * (gx * gy + 2)], 1);
                } else if (act) {
                    const int r = before + __popcll(word & ((2ull << lane) - 1ull)) - 1;
                    const int sl = c_idx[wave][r];
                    const int u = f.rect_min_u + sl;
                    const int v = w_minv[wave][sl] + (k - w_off[wave][sl]);
                    const uint32_t tk = f.isY ? (uint32_t)(u * gx + v) : (uint32_t)(v * gx + u);
                    key = (int32_t)(tk + 1);
#ifndef LG_REPRO_NEGATIVE_SLICE_BUG
                    const bool oob = (unsigned)key > (unsigned)(gx * gy);       // cannot happen for consistent slices; a key is an index downstream
                    bad_keys += oob ? 1 : 0;
                    if (dbg != nullptr && oob) {
                        if (dbg_claim(dbg + 16)) {
                            dbg[17] = sidx; dbg[18] = part; dbg[19] = k; dbg[20] = r; dbg[21] = sl; dbg[22] = w_off[wave][sl]; dbg[23] = w_minv[wave][sl];
                            dbg[24] = run; dbg[25] = scnt; dbg[26] = sgoff; dbg[27] = nsl; dbg[28] = nne; dbg[29] = key; dbg[30] = before; dbg[31] = f.isY ? 1 : 0;
                        }
                    }
                    key = oob ? 0 : key;
#endif
                    kout[sgoff + k] = key;
                    vout[sgoff + k] = sidx;
                    if (tile_counts) atomicAdd(&tile_counts[(size_t)b * (gx * gy + 2) + key], 1);
                }
                before += __popcll(word);
                if (totals) digit_hist_add(hist, (uint32_t)key, act, ds);
            }
            __builtin_amdgcn_wave_barrier();
            // leave the bitmap clean for the next splat: every non-empty slice clears the word that holds its start bit
            for (int r = lane; r < nne; r += 64) bitmap[wave][w_off[wave][c_idx[wave][r]] >> 5] = 0u;
            __builtin_amdgcn_wave_barrier();      // LDS operations of one wave execute in order: no fence (a fence would also wait for the global stores)
        }
    }
    if (bad_keys) lg_note_sanitised(LG_SITE_EMIT_KEY, bad_keys);
    if (totals) {
        __syncthreads();
        digit_hist_flush(hist, totals, ds.passes);
    }
}

// Validation aid (dbg != NULL: LgFusedCtx.debug_validate): the big-splat queue as dup_small left it, checked before dup_big follows it.
// dbg[32] = 1 and dbg[33..39] = {kind, sub-queue, position, entry, sub-queue length, capacity, N}; kind 1: sub-queue longer than its
// capacity, 2: entry names no slot / part of the prefix sums.
__global__ void __launch_bounds__(TPB) dup_queue_check_kernel(const int32_t* __restrict__ prefix, int N, long long table_len, const int* __restrict__ qcount,
                                                              const uint32_t* __restrict__ qentries, const int* __restrict__ gate, int* __restrict__ dbg)
{
    if (gate != nullptr && *gate == 0) return;
    const long long qcap = dup_queue_cap(N, table_len);
    const int sub = blockIdx.x;
    const int cnt = qcount[sub];
    if (cnt > qcap && threadIdx.x == 0 && dbg_claim(dbg + 32)) { dbg[33] = 1; dbg[34] = sub; dbg[35] = 0; dbg[36] = 0; dbg[37] = cnt; dbg[38] = (int)qcap; dbg[39] = N; }
    const long long lim = cnt < qcap ? cnt : qcap;
    for (long long e = threadIdx.x; e < lim; e += TPB) {
        const uint32_t ent = qentries[(size_t)sub * qcap + e];
        const int j = (int)(ent >> 8), part = (int)(ent & 255u);
        bool ok = j < N;
        if (ok) {
            const long long off = j == 0 ? 0 : prefix[j - 1];
            const long long c = prefix[j] - off;
            ok = c > 0 && off + c <= table_len && part < dup_num_parts((int)c);
        }
        if (!ok && dbg_claim(dbg + 32)) { dbg[33] = 2; dbg[34] = sub; dbg[35] = (int)e; dbg[36] = (int)ent; dbg[37] = cnt; dbg[38] = (int)qcap; dbg[39] = N; }
    }
}

// (Round 6 built a wave-autonomous form of this emission -- a wave owns 64 depth slots and never waits for another wave; small splats staged
// in LDS by the owning lane, medium ones one at a time with one lane per slice and a 2-D lane map for the runs, giants queued for
// dup_big_kernel -- and measured it in situ: 330 + 22 us against 207 + 143 us for dup_small + dup_big at 23.7 M instances (no gain), 245 us
// against 54 us on the bench's fresh frame in splat-id order, where whole wave groups are made of large splats.  SQ counters: 3.0 vector
// instructions per key against a budget of ~1 for the 100 us the round-5 verdict asked for; what binds either form is the instruction
// count per key -- extent 250 and slice bounds ~100 per slice of ~3.5 keys -- not the barriers.  Removed; profiles/r06_binning_ab.log,
// profiles/r06_emission_fresh_state_ab.log, git history of this file.)
// launch variants of the key emission (lg_set_tuning keys 10 / 11: A/B hooks of tools/, plain ints as in raster.hip)
static int g_depth_hist_blocks = 1024;        // workgroups of depth_keys_hist_kernel (lg_set_tuning(24, .))

static int g_sort_pack = 2;                   // two-pass tile sorts: 1 = second digit + value in one word between the passes; 2 = ... and the second pass leaves the
                                              // range table's starts instead of the sorted keys (lg_set_tuning(26, 0 | 1 | 2))
static int g_small_sort_lb = 8;                // look-back width of radix sorts with < 1024 key tiles (lg_set_tuning(15, 8 | 32); radix_onesweep_kernel).  32 measured SLOWER
                                               // (36-37 against 30 us per pass of the 2.2 M-key splat sort, profiles/r06_binning_ab.log): the passes are not bound by the look-back chain
static int g_dup_small_hi = DUP_SMALL_HI;      // largest tile count the owning thread walks itself; larger splats go to dup_big
static int g_dup_dynamic = 0;                  // 0 (default): groups dealt round robin; 1: long launches (>= 5 groups per workgroup) hand their groups out
                                               // on demand after one static round (needs the caller's zeroed ticket word); 2: always.  Five schemes measured,
                                               // none adopted: -38 .. +12 us on a 23 M-instance frame, +2 .. +36 us on the bench's fresh frame, +80 us on the
                                               // 10 M @1600x1200 frame (profiles/r05_emission_ab.log)
int lg_binning_set_tuning(int key, int value)
{
    if (key == 10) { if (value < DUP_SMALL || value > DUP_SMALL_HI) return (int)hipErrorInvalidValue; g_dup_small_hi = value; return 0; }
    if (key == 11) { if (value < 0 || value > 2) return (int)hipErrorInvalidValue; g_dup_dynamic = value; return 0; }
    if (key == 24) { if (value < 64 || value > 4096) return (int)hipErrorInvalidValue; g_depth_hist_blocks = value; return 0; }
    if (key == 26) { if (value < 0 || value > 2) return (int)hipErrorInvalidValue; g_sort_pack = value; return 0; }
    if (key == 15) { if (value != 8 && value != 32) return (int)hipErrorInvalidValue; g_small_sort_lb = value; return 0; }
    return (int)hipErrorInvalidValue;
}

// qcount: int32 [V][DUP_NQ], zero on entry; qentries: uint32 [V][lg_dup_queue_entries(N, table_len)].  totals (nullable): the tile sort's
// digit counts, accumulated here.
int lg_dup_emit(const float* ndc, const float* inv_cov, const float* opacity, const float* packed, const int32_t* prefix, const void* sorted_id,
                int sorted_id_is_int64, int V, int N, int H, int W, int TH, int TW, long long table_len, int32_t* keys, int32_t* values,
                int* qcount, uint32_t* qentries, int* totals, int begin_bit, int end_bit, uint32_t* zero_ptr, long long zero_words,
                uint32_t* ones_ptr, long long ones_words, uint32_t* zero2_ptr, long long zero2_words, void* stream)
{
    return lg_dup_emit_gated(ndc, inv_cov, opacity, packed, prefix, sorted_id, sorted_id_is_int64, V, N, H, W, TH, TW, table_len, keys, values,
                             qcount, qentries, totals, begin_bit, end_bit, nullptr, zero_ptr, zero_words, ones_ptr, ones_words, zero2_ptr, zero2_words,
                             nullptr, nullptr, nullptr, nullptr, stream);
}

int lg_dup_emit_gated(const float* ndc, const float* inv_cov, const float* opacity, const float* packed, const int32_t* prefix, const void* sorted_id,
                      int sorted_id_is_int64, int V, int N, int H, int W, int TH, int TW, long long table_len, int32_t* keys, int32_t* values,
                      int* qcount, uint32_t* qentries, int* totals, int begin_bit, int end_bit, int* tile_counts,
                      uint32_t* zero_ptr, long long zero_words,
                      uint32_t* ones_ptr, long long ones_words, uint32_t* zero2_ptr, long long zero2_words,
                      const int* gate, int* trunc_flag, int* dbg, int* grp_ticket, void* stream)
{
    if (N <= 0) return 0;
    if (packed && sorted_id_is_int64) return (int)hipErrorInvalidValue;     // packed records: fused executor only (int32 order)
    if (N >= (1 << 24)) return (int)hipErrorInvalidValue;                   // queue entries carry the depth slot in 24 bits
    int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    const int ngroups = lg_cdiv(N, TPB);
    dim3 grid(ngroups < 1280 ? ngroups : 1280, V);        // persistent workgroups (5 per CU), 256 depth slots at a time
    // groups on demand (after one static round): long launches only (auto), never, or always -- lg_set_tuning(11, 1 / 0 / 2)
    const bool use_ticket = grp_ticket != nullptr && (g_dup_dynamic == 2 || (g_dup_dynamic == 1 && ngroups >= 5 * (int)grid.x));
    dim3 grid_big(1024, V);                               // persistent: 4096 waves drain the queue
    hipStream_t s = (hipStream_t)stream;
    DigitSpec ds = { begin_bit, 0, 0u };
    if (totals) {
        ds.passes = (end_bit - begin_bit + 7) / 8;
        if (ds.passes < 1 || ds.passes > SORT_MAX_PASSES_DUP) return (int)hipErrorInvalidValue;
        ds.last_mask = (1u << ((end_bit - begin_bit) - (ds.passes - 1) * 8)) - 1u;
    }
    SplatSrc src = { ndc, inv_cov, opacity, (const float4*)packed };
#define LAUNCH_DUP(A_, B_, T_, P_)                                                                                                          \
    do {                                                                                                                                   \
        if (gx * gy + 1 <= 0xffff)                                                                                                         \
            hipLaunchKernelGGL((dup_small_kernel<A_, B_, T_, P_, uint16_t>), grid, dim3(TPB), 0, s, src, prefix, (const T_*)sorted_id, N,  \
                               H, W, gx, gy, table_len, keys, values, qcount, qentries, totals, ds, tile_counts, zero_ptr, zero_words, ones_ptr, ones_words, zero2_ptr, zero2_words, gate, trunc_flag, dbg, g_dup_small_hi, use_ticket ? grp_ticket : (int*)nullptr, 1); \
        else                                                                                                                               \
            hipLaunchKernelGGL((dup_small_kernel<A_, B_, T_, P_, int32_t>), grid, dim3(TPB), 0, s, src, prefix, (const T_*)sorted_id, N,   \
                               H, W, gx, gy, table_len, keys, values, qcount, qentries, totals, ds, tile_counts, zero_ptr, zero_words, ones_ptr, ones_words, zero2_ptr, zero2_words, gate, trunc_flag, dbg, g_dup_small_hi, use_ticket ? grp_ticket : (int*)nullptr, 1); \
        if (dbg != nullptr && V == 1)                                                                                                      \
            hipLaunchKernelGGL(dup_queue_check_kernel, dim3(DUP_NQ), dim3(TPB), 0, s, prefix, N, table_len, (const int*)qcount,             \
                               (const uint32_t*)qentries, gate, dbg);                                                                       \
        hipLaunchKernelGGL((dup_big_kernel<A_, B_, T_, P_>), grid_big, dim3(TPB), 0, s, src, prefix, (const T_*)sorted_id,                 \
                           N, H, W, gx, gy, table_len, keys, values, (const int*)qcount, (const uint32_t*)qentries, totals, ds, tile_counts, gate, dbg); \
    } while (0)
#define DISPATCH_DUP(A_, B_)                                              \
    do {                                                                  \
        if (packed) LAUNCH_DUP(A_, B_, int32_t, true);                    \
        else if (sorted_id_is_int64) LAUNCH_DUP(A_, B_, int64_t, false);  \
        else LAUNCH_DUP(A_, B_, int32_t, false);                          \
    } while (0)
    if (TH == 8 && TW == 16) DISPATCH_DUP(8, 16);
    else if (TH == 16 && TW == 16) DISPATCH_DUP(16, 16);
    else if (TH == 12 && TW == 16) DISPATCH_DUP(12, 16);
    else if (TH == 8 && TW == 8) DISPATCH_DUP(8, 8);
    else return (int)hipErrorInvalidValue;
#undef DISPATCH_DUP
#undef LAUNCH_DUP
    LG_RETURN_LAST();
}

long long lg_dup_queue_entries(long long N, long long table_len) { return DUP_NQ * dup_queue_cap(N, table_len); }

// temp: per view DUP_NQ counters followed by the queue entries
LG_API long long lg_duplicate_with_keys_temp_bytes(int V, int N, long long table_len)
{
    return (long long)sizeof(int) * (long long)V * (DUP_NQ + lg_dup_queue_entries(N, table_len));
}

LG_API int lg_duplicate_with_keys(const float* ndc, const float* inv_cov, const float* opacity, const int32_t* prefix,
                                  const void* sorted_id, int sorted_id_is_int64, int V, int N, int H, int W, int TH, int TW,
                                  long long table_len, int32_t* keys, int32_t* values, void* temp, long long temp_bytes, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(ndc, inv_cov, opacity, prefix, keys, values);            // sorted_id == NULL: the slots are the splats themselves (ascending id)
    if (temp == nullptr || temp_bytes < lg_duplicate_with_keys_temp_bytes(V, N, table_len)) return (int)hipErrorInvalidValue;
    int* qcount = (int*)temp;                                                                        // [V][DUP_NQ]
    uint32_t* qentries = (uint32_t*)(qcount + (size_t)V * DUP_NQ);
    hipError_t err = hipMemsetAsync(qcount, 0, sizeof(int) * (size_t)V * DUP_NQ, (hipStream_t)stream);
    if (err != hipSuccess) return (int)err;
    return lg_dup_emit(ndc, inv_cov, opacity, nullptr, prefix, sorted_id, sorted_id_is_int64, V, N, H, W, TH, TW, table_len, keys, values,
                       qcount, qentries, nullptr, 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, stream);
}

// ---------------------------------------------------------------------------------------------
// Stable LSD radix sort of (u32 key, u32 value) pairs on key bits [begin_bit, end_bit), 8 bits per pass.
// Replaces cub::DeviceRadixSort::SortPairs (GR/binning.cu:204-221) and torch.sort of the depth keys
// (litegs/utils/wrapper.py:739).  Two code paths: sorts of a few key tiles (tile = 4096 keys per 256-thread workgroup)
// run histogram + scatter launches where every workgroup scans the raw histogram rows itself (radix_hist_kernel +
// radix_scatter_kernel<true>: match-any ranking with 8 wave ballots); everything else runs ONE launch per pass
// (radix_onesweep_kernel: decoupled look-back, lane-ordered LDS ranking).  Equal keys keep their input order in both
// (stability is load-bearing: it preserves depth order inside a tile).  Digit totals for all passes are counted once up
// front (they are permutation invariant) -- by radix_totals_kernel here, or by the kernel that produced the keys
// (lg_radix_sort_prepared).
// ---------------------------------------------------------------------------------------------
#define RADIX_BITS 8
#define RADIX (1 << RADIX_BITS)
#ifndef SORT_ITEMS
#define SORT_ITEMS 16
#endif
#define SORT_TILE (TPB * SORT_ITEMS)
#define SORT_MAX_PASSES 4

__device__ __forceinline__ long long bounded_n(long long n, const int* __restrict__ n_dev)
{
    if (n_dev == nullptr) return n;
    long long m = n_dev[0];
    return m < n ? (m < 0 ? 0 : m) : n;
}

__global__ void __launch_bounds__(TPB) radix_totals_kernel(const uint32_t* __restrict__ keys, long long n, const int* __restrict__ n_dev,
                                                           int begin_bit, int passes,
                                                           uint32_t last_mask, int* __restrict__ totals /*[passes][RADIX]*/)
{
    __shared__ int h[SORT_MAX_PASSES * RADIX];
    n = bounded_n(n, n_dev);
    for (int k = threadIdx.x; k < passes * RADIX; k += TPB) h[k] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
        uint32_t key = keys[i];
        for (int p = 0; p < passes; p++) {
            uint32_t d = (key >> (begin_bit + p * RADIX_BITS)) & ((p == passes - 1) ? last_mask : (uint32_t)(RADIX - 1));
            atomicAdd(&h[p * RADIX + d], 1);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < passes * RADIX; k += TPB)
        if (h[k]) atomicAdd(&totals[(blockIdx.x % LG_SORT_TOTALS_COPIES) * LG_SORT_TOTALS_STRIDE + k], h[k]);
}

__global__ void __launch_bounds__(TPB) radix_hist_kernel(const uint32_t* __restrict__ keys, long long n, const int* __restrict__ n_dev,
                                                         int shift, uint32_t mask,
                                                         int ntiles, int* __restrict__ hist /*[RADIX][ntiles]*/)
{
    __shared__ int h[RADIX];
    n = bounded_n(n, n_dev);
    long long base = (long long)blockIdx.x * SORT_TILE;
    if (base >= n) { hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = 0; return; }
    h[threadIdx.x] = 0;                              // TPB == RADIX
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        long long i = base + j * TPB + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & mask], 1);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// Scatter with an LDS-local reorder: (1) 16 rounds of 256 keys compute each key's stable rank inside the tile (wave
// "match-any" ballots + per-wave digit counts, two barriers per round); (2) keys/values are placed in LDS in sorted order;
// (3) the tile is streamed out: consecutive LDS slots with the same digit go to consecutive global addresses, so the
// stores are coalesced runs instead of 4-byte scatters over 256 destinations (1.9 -> >3 TB/s effective on the tile sort).
// INLINE_SCAN (few tiles, e.g. the depth sort of ~1 M keys): `offsets` is the RAW histogram table and every workgroup derives
// its own global offsets from it (digit d: sum of row d before this tile + exclusive scan over digits of the row totals), which
// removes the totals and scan launches from each pass -- those small sorts are launch-latency bound.
template <bool INLINE_SCAN>
__global__ void __launch_bounds__(TPB) radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                            const int* __restrict__ offsets /*[RADIX][ntiles]*/, long long n,
                                                            const int* __restrict__ n_dev, int shift, uint32_t mask, int ntiles)
{
    __shared__ uint32_t lds_k[SORT_TILE];
    __shared__ uint32_t lds_v[SORT_TILE];
    __shared__ int wave_cnt[2][TPB / 64][RADIX];
    __shared__ int digit_run[RADIX];          // running count per digit, then exclusive local base
    __shared__ int global_base[RADIX];
    __shared__ int wsum[TPB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    n = bounded_n(n, n_dev);
    const long long base = (long long)blockIdx.x * SORT_TILE;
    if (base >= n) return;
    const int cnt_tile = (int)((n - base) < SORT_TILE ? (n - base) : SORT_TILE);
    digit_run[tid] = 0;
#pragma unroll
    for (int w = 0; w < TPB / 64; w++) { wave_cnt[0][w][tid] = 0; wave_cnt[1][w][tid] = 0; }
    if (INLINE_SCAN) {
        const int* row = offsets + (size_t)tid * ntiles;
        int before = 0, total = 0;
        for (int t = 0; t < ntiles; t++) { int c = row[t]; total += c; before += (t < (int)blockIdx.x) ? c : 0; }
        int inc = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int nb = __shfl_up(inc, o);
            if (lane >= o) inc += nb;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int wb0 = 0;
        for (int w = 0; w < wave; w++) wb0 += wsum[w];
        global_base[tid] = wb0 + inc - total + before;
        __syncthreads();
    } else {
        global_base[tid] = offsets[(size_t)tid * ntiles + blockIdx.x];
    }
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS];
    int lrank[SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        const int e = j * TPB + tid;
        const bool ok = e < cnt_tile;
        key[j] = ok ? keys_in[base + e] : 0u;
        val[j] = ok ? vals_in[base + e] : 0u;
    }
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        const int cur = j & 1;
        const bool ok = (j * TPB + tid) < cnt_tile;
        const uint32_t d = (key[j] >> shift) & mask;
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int bit = 0; bit < RADIX_BITS; bit++) {
            const bool set = (d >> bit) & 1u;
            unsigned long long bal = __ballot(set);
            peers &= set ? bal : ~bal;
        }
        const int rank = __popcll(peers & lt_mask);
        if (ok && rank == 0) wave_cnt[cur][wave][d] = __popcll(peers);
        __syncthreads();
        int off = digit_run[d] + rank;
        for (int w = 0; w < wave; w++) off += wave_cnt[cur][w][d];
        lrank[j] = off;
        __syncthreads();
        {
            int add = 0;
#pragma unroll
            for (int w = 0; w < TPB / 64; w++) { add += wave_cnt[cur][w][tid]; wave_cnt[cur][w][tid] = 0; }
            digit_run[tid] += add;
        }
    }
    __syncthreads();
    // exclusive scan of the per-digit tile counts -> local base (TPB == RADIX: one digit per thread)
    const int dcount = digit_run[tid];
    int incl = dcount;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int nb = __shfl_up(incl, o);
        if (lane >= o) incl += nb;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wb = 0;
    for (int w = 0; w < wave; w++) wb += wsum[w];
    const int lbase = wb + incl - dcount;
    __syncthreads();
    digit_run[tid] = lbase;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        if ((j * TPB + tid) < cnt_tile) {
            const uint32_t d = (key[j] >> shift) & mask;
            const int pos = digit_run[d] + lrank[j];
            lds_k[pos] = key[j];
            lds_v[pos] = val[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        const int p = j * TPB + tid;
        if (p < cnt_tile) {
            const uint32_t k = lds_k[p];
            const uint32_t d = (k >> shift) & mask;
            const int g = global_base[d] + (p - digit_run[d]);
            if ((unsigned long long)(long long)g >= (unsigned long long)n) { lg_note_sanitised(LG_SITE_RADIX_INDEX); continue; }     // (few-tile sorts: not a hot loop)
            keys_out[g] = k;
            vals_out[g] = lds_v[p];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Single-launch radix pass ("onesweep"): the per-workgroup digit counts are chained through a status table with
// decoupled look-back instead of a histogram launch + a scan launch per pass.  Workgroups take a ticket (so a workgroup's
// logical predecessors have all started), each wave ranks a contiguous run of 1024 keys (returning LDS adds), they publish their 256 digit
// counts (flag AGG), thread d then walks back over the predecessors' words for digit d until it meets an inclusive prefix
// (flag INC), publishes its own inclusive prefix and the tile is streamed out.  One status word carries flag and value, so
// no ordering between separate flag/value stores is needed.  status[] and ticket[] must be zero on entry.
// ---------------------------------------------------------------------------------------------
#define ST_AGG 0x40000000u
#define ST_INC 0x80000000u
#define ST_VAL 0x3fffffffu

// TILES key tiles per workgroup (256 threads each, processed side by side) share ONE ticket: the ticket is a returning atomic on a
// single address (~8 ns each, serialised in L2); at 2865 tiles per pass it delayed workgroup starts by ~16 us per pass.
// PACK (two-pass sorts of keys below 2^16 whose values fit beside the second digit: the tile sort -- 15 key bits at 1080p, 22 value bits
// at 3 M Gaussians): the first pass (PACK == 1) writes ONE word per element, second digit << vbits | value; the second pass (PACK == 2)
// reads that word alone and rebuilds the full key -- its first digit is the bucket of the first pass the element's position lies in,
// found against the exclusive scan of the first pass's digit totals (`totals_prev`).  8 bytes per element less through HBM (28 instead
// of 36 over the two passes and the range scan), same table bit for bit.  PACK == 3: the second pass does not write the keys at all but
// the range table's starts -- the first element of a key inside a workgroup's tile proposes its output position with an unsigned
// atomicMin on the table (pre-filled with -1 = the largest unsigned); tile_range_close_kernel then adds the words tile_range_kernel
// derives from gaps and the table length.  20 bytes per instance, one 16 000-word kernel instead of a pass over the keys.
template <int TILES, bool BALLOT_RANK, int LB, int PACK = 0>
__global__ void __launch_bounds__(TPB * TILES) radix_onesweep_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                             uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                             const int* __restrict__ totals /*[RADIX] of this pass*/,
                                                             uint32_t* __restrict__ status /*[ntiles][RADIX], zero*/, int* __restrict__ ticket,
                                                             long long n, const int* __restrict__ n_dev, int shift, uint32_t mask,
                                                             const int32_t* __restrict__ aux_in, int32_t* __restrict__ aux_out,
                                                             int vbits /*PACK*/, const int* __restrict__ totals_prev /*PACK >= 2: [RADIX] of the first pass*/,
                                                             int32_t* __restrict__ range_out /*PACK == 3: the tile range table, pre-filled with -1*/, int max_tile)
{
    constexpr int NW = TPB / 64;
    constexpr int WAVE_KEYS = SORT_ITEMS * 64;       // each wave ranks a contiguous run of 1024 keys on its own (no block barriers)
    __shared__ uint32_t lds_k_[TILES][SORT_TILE];
    __shared__ uint32_t lds_v_[PACK ? 1 : TILES][PACK ? 1 : SORT_TILE];
    __shared__ unsigned char lds_b_[PACK ? TILES : 1][PACK ? SORT_TILE : 1];      // PACK: the first digit of the element (its value rides in the packed word)
    __shared__ int wave_cnt_[TILES][NW][RADIX];      // running per-wave digit counts, then exclusive offset of the wave inside the digit
    __shared__ int digit_base_[TILES][RADIX];        // exclusive local base of the digit in the sorted tile
    __shared__ int global_base_[TILES][RADIX];
    __shared__ int wsum_[TILES][NW];
    __shared__ int wsum_g_[TILES][NW];
    __shared__ int bid_s;
    __shared__ int low_base_[PACK >= 2 ? TILES : 1][PACK >= 2 ? RADIX + 1 : 1];      // PACK == 2: first position of every bucket of the first pass
    const int half = threadIdx.x / TPB;              // which of the workgroup's tiles this thread works on
    const int tid = threadIdx.x % TPB, lane = tid & 63, wave = tid >> 6;
    uint32_t* lds_k = lds_k_[half]; uint32_t* lds_v = lds_v_[PACK ? 0 : half];
    unsigned char* lds_b = lds_b_[PACK ? half : 0];
    int (*wave_cnt)[RADIX] = wave_cnt_[half];
    int* digit_base = digit_base_[half]; int* global_base = global_base_[half];
    int* wsum = wsum_[half]; int* wsum_g = wsum_g_[half];
    n = bounded_n(n, n_dev);
    if (n <= 0) return;                              // nothing to sort (a gated fallback pass that is not needed): no ticket traffic
    if (threadIdx.x == 0) bid_s = atomicAdd(ticket, 1);
#pragma unroll
    for (int w = 0; w < NW; w++) wave_cnt[w][tid] = 0;
    __syncthreads();
    if ((long long)bid_s * TILES * SORT_TILE >= n) return;          // uniform: none of this workgroup's tiles holds keys
    const int bid = bid_s * TILES + half;
    const long long base = (long long)bid * SORT_TILE;
    const bool act = base < n;                       // a tile past the end stays in the barriers but neither publishes nor looks back
    const int cnt_tile = act ? (int)((n - base) < SORT_TILE ? (n - base) : SORT_TILE) : 0;
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS];
    int lrank[SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        const int e = wave * WAVE_KEYS + j * 64 + lane;
        const bool ok = e < cnt_tile;
        key[j] = ok ? keys_in[base + e] : 0u;
        if (PACK < 2) val[j] = ok ? vals_in[base + e] : 0u;
    }
    if constexpr (PACK >= 2) {
        // exclusive scan of the first pass's digit totals -> bucket boundaries; an element's first digit is the bucket its position is in
        int* low_base = low_base_[half];
        int tl = 0;
#pragma unroll
        for (int c = 0; c < LG_SORT_TOTALS_COPIES; c++) tl += totals_prev[c * LG_SORT_TOTALS_STRIDE + tid];
        int inc = tl;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int wb = 0;
        for (int w = 0; w < wave; w++) wb += wsum[w];
        low_base[tid] = wb + inc - tl;
        if (tid == TPB - 1) low_base[RADIX] = 0x7fffffff;
        __syncthreads();
        // a thread's elements lie 64 positions apart in ascending order: one binary search, then a walk
        int lo = 0;
        {
            const long long p0 = base + wave * WAVE_KEYS + lane;
            int a = 0, b = RADIX;                                           // largest d with low_base[d] <= p0
            while (b - a > 1) { const int m = (a + b) >> 1; if ((long long)low_base[m] <= p0) a = m; else b = m; }
            lo = a;
        }
        const uint32_t vmask = (1u << vbits) - 1u;
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; j++) {
            const long long pj = base + wave * WAVE_KEYS + j * 64 + lane;
            while ((long long)low_base[lo + 1] <= pj) lo++;
            val[j] = (uint32_t)lo;                                          // (the packed word stays in key[j]: second digit << vbits | value)
        }
        (void)vmask;
        __syncthreads();                                                    // (wsum is reused below)
    }
    // Rank of a key among the wave's earlier keys with the same digit.
    // Fast path: the value a returning LDS add hands back -- rounds are sequential, and inside one ds_add_rtn instruction the LDS
    // serves the lanes that hit the same address in increasing lane order, so equal digits keep their input order.  That order is a
    // property of the CDNA LDS pipeline, not of the programming model, so it is never ASSUMED: lg_radix_rank_selftest() (below) runs
    // once per process before the first sort and checks it on this very device; if a single rank comes back out of lane order the
    // sorts use BALLOT_RANK instead, where stability follows from the code alone: the lanes that share a digit are found with eight
    // ballots (one per digit bit), a key's rank is the number of LOWER lanes in that set, and the set's lowest lane advances the
    // wave-private counter (+~10 us per pass of the 11.7 M-key tile sort; measured when it replaced the first version).
    int* my_cnt = wave_cnt[wave];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        const bool ok = (wave * WAVE_KEYS + j * 64 + lane) < cnt_tile;
        const uint32_t d = PACK >= 2 ? ((key[j] >> vbits) & mask) : ((key[j] >> shift) & mask);
        if constexpr (!BALLOT_RANK) {
            lrank[j] = ok ? atomicAdd(&my_cnt[d], 1) : 0;
        } else {
            unsigned long long same = __ballot(ok);
#pragma unroll
            for (int b = 0; b < RADIX_BITS; b++) {
                const unsigned long long bal = __ballot((d >> b) & 1u);
                same &= ((d >> b) & 1u) ? bal : ~bal;
            }
            int r = 0;
            if (ok) {
                const int leader = __ffsll((long long)same) - 1;
                const int below = __popcll(same & ((1ull << lane) - 1ull));
                int base = 0;
                if (lane == leader) { base = my_cnt[d]; my_cnt[d] = base + __popcll(same); }
                base = __shfl(base, leader);
                r = base + below;
            }
            lrank[j] = r;
        }
    }
    __syncthreads();
    // thread d: counts of digit d per wave -> wave offsets inside the digit, tile count of the digit
    int dcount = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) { int c = wave_cnt[w][tid]; wave_cnt[w][tid] = dcount; dcount += c; }
    uint32_t* my = status + (size_t)bid * RADIX + tid;
    if (act) {
        if (bid == 0) __hip_atomic_store(my, ST_INC | (uint32_t)dcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(my, ST_AGG | (uint32_t)dcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // digit base in the output = exclusive scan over digits of the global totals; local base = same scan of the tile counts
    int total = 0;
#pragma unroll
    for (int c = 0; c < LG_SORT_TOTALS_COPIES; c++) total += totals[c * LG_SORT_TOTALS_STRIDE + tid];
    int inc_g = total, inc_l = dcount;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int ng = __shfl_up(inc_g, o), nl = __shfl_up(inc_l, o);
        if (lane >= o) { inc_g += ng; inc_l += nl; }
    }
    if (lane == 63) { wsum_g[wave] = inc_g; wsum[wave] = inc_l; }
    __syncthreads();
    int wbg = 0, wbl = 0;
    for (int w = 0; w < wave; w++) { wbg += wsum_g[w]; wbl += wsum[w]; }
    int gbase = wbg + inc_g - total;
    digit_base[tid] = wbl + inc_l - dcount;
    __syncthreads();
    // place the tile in LDS in sorted order while the predecessors' counts arrive
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        if ((wave * WAVE_KEYS + j * 64 + lane) < cnt_tile) {
            const uint32_t d = PACK >= 2 ? ((key[j] >> vbits) & mask) : ((key[j] >> shift) & mask);
            const int pos = digit_base[d] + wave_cnt[wave][d] + lrank[j];
            if constexpr (PACK == 1) {
                lds_k[pos] = ((key[j] >> RADIX_BITS) << vbits) | (val[j] & ((1u << vbits) - 1u));    // second digit | value; the first digit is implied by the position
                lds_b[pos] = (unsigned char)d;
            } else if constexpr (PACK >= 2) {
                lds_k[pos] = key[j];                                        // the packed word
                lds_b[pos] = (unsigned char)val[j];                         // the first digit, rebuilt from the element's position
            } else {
                lds_k[pos] = key[j];
                lds_v[pos] = val[j];
            }
        }
    }
    if (act && bid != 0) {
        // look-back, LB predecessors per step: the loads of one step are independent, so the walk costs one L2 round trip per
        // LB workgroups instead of one per workgroup (matters when ~1000 resident workgroups start together).  (LB = 32 for the splat
        // depth sort, whose ~540 tiles are all resident at once, was tried in round 6: slower, lg_set_tuning(15, 32).)
        uint32_t excl = 0;
        int b = bid - 1;
        bool done = false;
        while (!done) {
            uint32_t v[LB];
#pragma unroll
            for (int k = 0; k < LB; k++)
                v[k] = (b - k >= 0) ? __hip_atomic_load(status + (size_t)(b - k) * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : ST_INC;
            int used = 0;
#pragma unroll
            for (int k = 0; k < LB; k++) {
                if (!done && used == k) {
                    if ((v[k] & (ST_AGG | ST_INC)) != 0u) {
                        excl += v[k] & ST_VAL;
                        used = k + 1;
                        if (v[k] & ST_INC) done = true;
                    }
                }
            }
            b -= used;
            if (!done && used < LB) __builtin_amdgcn_s_sleep(1);
        }
        __hip_atomic_store(my, ST_INC | (excl + (uint32_t)dcount), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gbase += (int)excl;
    }
    global_base[tid] = gbase;
    __syncthreads();
    int bad_pos = 0;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        const int p = j * TPB + tid;
        if (p < cnt_tile) {
            const uint32_t kw = lds_k[p];
            const uint32_t lb = PACK ? (uint32_t)lds_b[p] : 0u;
            const uint32_t d = PACK == 1 ? lb : (PACK >= 2 ? ((kw >> vbits) & mask) : ((kw >> shift) & mask));
            const uint32_t k = PACK >= 2 ? ((d << RADIX_BITS) | lb) : kw;                       // PACK >= 2: the full key again
            const uint32_t v = PACK >= 2 ? (kw & ((1u << vbits) - 1u)) : (PACK == 1 ? 0u : lds_v[p]);
            const int g = global_base[d] + (p - digit_base[d]);
            // g is built from the producer's digit totals and the predecessors' look-back words: if either is inconsistent with the keys
            // (a total that over-counts, a status word that was not zero on entry) g leaves [0, n) -- or lands on another key's slot,
            // which leaves a hole of stale memory elsewhere.  The first is caught here and counted; never a wild store.
            // (a predicated store and a count behind the loop: no branch with an atomic in the streaming loop)
            const bool in_range = (unsigned long long)(long long)g < (unsigned long long)n;
            bad_pos += in_range ? 0 : 1;
            if (in_range) {
                if constexpr (PACK == 3) {
                    // equal keys are adjacent inside a digit's run of the tile (the input is ordered by the first digit, the ranking is stable)
                    if (p == digit_base[d] || (lds_k[p - 1] >> vbits) != (kw >> vbits) || lds_b[p - 1] != (unsigned char)lb) {
                        if ((unsigned)k <= (unsigned)max_tile) atomicMin(reinterpret_cast<unsigned int*>(range_out) + k, (unsigned int)g);
                        else bad_pos++;
                    }
                } else keys_out[g] = k;
                if (PACK != 1) vals_out[g] = v;
                if (PACK == 0 && aux_in) aux_out[g] = aux_in[v];      // last pass of the depth sort: tile counts gathered into depth order on the way out
            }
        }
    }
    if (bad_pos) lg_note_sanitised(LG_SITE_RADIX_INDEX, bad_pos);
}

// ---------------------------------------------------------------------------------------------
// One-off self-test of the LDS property the fast ranking relies on (see radix_onesweep_kernel): for several digit patterns, 64 lanes
// of a wave issue a returning LDS add on counters chosen by the pattern, repeatedly; every returned value must equal the counter's
// previous total plus the number of LOWER lanes that hit the same counter.  Runs on 64 workgroups x 4 waves so that several CUs and
// all SIMDs of a CU are exercised, with neighbouring waves hammering the same LDS at the same time.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) radix_rank_selftest_kernel(int* __restrict__ bad)
{
    __shared__ int cnt[TPB / 64][RADIX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = lane; k < RADIX; k += 64) cnt[wave][k] = 0;
    __syncthreads();
    int expect[RADIX / 64 + 1];
    uint32_t seed = 0x9e3779b9u * (blockIdx.x * 4 + wave + 1);
    int errors = 0;
    for (int round = 0; round < 512; round++) {
        uint32_t d;
        const int kind = round & 7;
        if (kind == 0) d = 0;                                   // all lanes on one counter
        else if (kind == 1) d = lane & 1;
        else if (kind == 2) d = (lane >> 3) & 7;
        else if (kind == 3) d = lane % 3;
        else if (kind == 4) d = (uint32_t)lane;                  // all different
        else {                                                  // pseudo-random, few distinct values
            seed = seed * 1664525u + 1013904223u + (uint32_t)lane * 2654435761u;
            d = (seed >> 24) & (kind == 5 ? 3u : (kind == 6 ? 15u : 255u));
            seed = __shfl((int)seed, 0) ^ (uint32_t)round;
        }
        (void)expect;
        const int before = cnt[wave][d];                        // plain read: nothing else touches this wave's counters
        unsigned long long same = ~0ull;
#pragma unroll
        for (int b = 0; b < RADIX_BITS; b++) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const int want = before + __popcll(same & ((1ull << lane) - 1ull));
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // the read above completes before the atomics are issued
        const int got = atomicAdd(&cnt[wave][d], 1);
        if (got != want) errors++;
    }
    if (errors) atomicAdd(bad, errors);
}

static int g_rank_mode = -1;          // -1 unknown, 0 lane-ordered LDS returns verified on this device, 1 ballot ranking

LG_API int lg_radix_rank_mode(void)
{
    if (g_rank_mode >= 0) return g_rank_mode;
    const char* force = getenv("LITEGS_RADIX_RANK");
    if (force && force[0] == 'b') { g_rank_mode = 1; return 1; }
    int* bad = nullptr;
    int host = -1;
    if (hipMalloc(&bad, sizeof(int)) != hipSuccess) { g_rank_mode = 1; return 1; }
    (void)hipMemset(bad, 0, sizeof(int));
    hipLaunchKernelGGL(radix_rank_selftest_kernel, dim3(64), dim3(TPB), 0, 0, bad);
    if (hipMemcpy(&host, bad, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) host = -1;
    (void)hipFree(bad);
    g_rank_mode = (host == 0) ? 0 : 1;
    return g_rank_mode;
}

LG_API int lg_radix_set_rank_mode(int mode)          // test hook: 0 / 1 force a ranking, -1 = decide again by the self-test
{
    g_rank_mode = mode < 0 ? -1 : (mode ? 1 : 0);
    return 0;
}

// Two tiles per workgroup once there are more tiles than fit on the chip at once (the tile sort); small sorts keep one tile per
// workgroup so that every CU gets work.
static void launch_onesweep(int ntiles, hipStream_t s, const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, const int* totals,
                            uint32_t* status, int* ticket, long long n, const int* n_dev, int shift, uint32_t mask, const int32_t* aux_in,
                            int32_t* aux_out, int pack = 0, int vbits = 0, const int* totals_prev = nullptr, int32_t* range_out = nullptr, int max_tile = 0)
{
    const bool ballot = lg_radix_rank_mode() != 0;
#define LAUNCH_OSP(T_, B_, L_, G_, P_) hipLaunchKernelGGL((radix_onesweep_kernel<T_, B_, L_, P_>), dim3(G_), dim3(TPB * T_), 0, s, kin, vin, kout, vout, totals, status, \
                                                      ticket, n, n_dev, shift, mask, aux_in, aux_out, vbits, totals_prev, range_out, max_tile)
#define LAUNCH_OS(T_, B_, L_, G_) LAUNCH_OSP(T_, B_, L_, G_, 0)
    // (one key tile per workgroup for the packed passes -- 27 KB of LDS, five workgroups per CU instead of two: +20 us per step, profiles/r06_packed_tile_sort_ab.log)
    if (pack == 1) { if (ballot) LAUNCH_OSP(2, true, 8, (ntiles + 1) / 2, 1); else LAUNCH_OSP(2, false, 8, (ntiles + 1) / 2, 1); }
    else if (pack == 2) { if (ballot) LAUNCH_OSP(2, true, 8, (ntiles + 1) / 2, 2); else LAUNCH_OSP(2, false, 8, (ntiles + 1) / 2, 2); }
    else if (pack == 3) { if (ballot) LAUNCH_OSP(2, true, 8, (ntiles + 1) / 2, 3); else LAUNCH_OSP(2, false, 8, (ntiles + 1) / 2, 3); }
    else if (ntiles >= 1024) {                             // (4 tiles per workgroup measured slower: 97 vs 88 us per pass)
        if (ballot) LAUNCH_OS(2, true, 8, (ntiles + 1) / 2); else LAUNCH_OS(2, false, 8, (ntiles + 1) / 2);
    } else if (g_small_sort_lb == 8) {
        if (ballot) LAUNCH_OS(1, true, 8, ntiles); else LAUNCH_OS(1, false, 8, ntiles);
    } else {
        if (ballot) LAUNCH_OS(1, true, 32, ntiles); else LAUNCH_OS(1, false, 32, ntiles);
    }
#undef LAUNCH_OS
#undef LAUNCH_OSP
}

LG_API int lg_radix_sort_pairs_bounded(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, long long n, const int* n_dev,
                                       int begin_bit, int end_bit, void* temp, long long temp_bytes, void* stream);

// temp layout: totals[SORT_MAX_PASSES][RADIX] | ticket[SORT_MAX_PASSES] (+pad to 64 ints) | table
// table = per-pass histogram [RADIX][ntiles] (small sorts) or look-back status [SORT_MAX_PASSES][ntiles][RADIX] (onesweep)
#define SORT_HEADER_INTS LG_SORT_HEADER_INTS
LG_API long long lg_radix_sort_temp_bytes(long long n)
{
    long long ntiles = (n + SORT_TILE - 1) / SORT_TILE;
    if (ntiles < 1) ntiles = 1;
    return (long long)sizeof(int) * (SORT_HEADER_INTS + (long long)SORT_MAX_PASSES * RADIX * ntiles);
}

LG_API int lg_radix_sort_num_passes(int begin_bit, int end_bit)
{
    int bits = end_bit - begin_bit;
    return bits <= 0 ? 0 : (bits + RADIX_BITS - 1) / RADIX_BITS;
}

// Ping-pongs a -> b -> a ...; the sorted result is in (keys_b, vals_b) when the pass count is odd, else in
// (keys_a, vals_a) (lg_radix_sort_num_passes tells the caller which).  Both buffer pairs hold n elements.
LG_API int lg_radix_sort_pairs(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, long long n,
                               int begin_bit, int end_bit, void* temp, long long temp_bytes, void* stream)
{
    return lg_radix_sort_pairs_bounded(keys_a, vals_a, keys_b, vals_b, n, nullptr, begin_bit, end_bit, temp, temp_bytes, stream);
}

// n_dev (nullable): device int32 holding the number of leading elements to sort (min(n, *n_dev)); the rest of the buffers is
// left untouched.  Lets the GPU-driven pipeline sort the ACTUAL instance count instead of the 1.5x over-allocated table.
LG_API int lg_radix_sort_pairs_bounded(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, long long n, const int* n_dev,
                                       int begin_bit, int end_bit, void* temp, long long temp_bytes, void* stream)
{
    int passes = lg_radix_sort_num_passes(begin_bit, end_bit);
    if (n <= 0 || passes == 0) return 0;
    LG_REQUIRE(keys_a, vals_a, keys_b, vals_b);
    if (passes > SORT_MAX_PASSES || n > 0x3fffffffLL) return (int)hipErrorInvalidValue;   // look-back status words carry 30-bit counts
    if (temp_bytes < lg_radix_sort_temp_bytes(n)) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    int ntiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
    int* totals = (int*)temp;
    int* ticket = totals + LG_SORT_TOTALS_COPIES * LG_SORT_TOTALS_STRIDE;
    int* table = totals + SORT_HEADER_INTS;
    int last_bits = (end_bit - begin_bit) - (passes - 1) * RADIX_BITS;
    uint32_t last_mask = (1u << last_bits) - 1u;
    const bool inline_scan = ntiles <= 8;      // tiny sorts: every workgroup scans the raw histogram rows itself (no look-back chain)
    if (!inline_scan) {
        hipError_t err = hipMemsetAsync(totals, 0, sizeof(int) * (SORT_HEADER_INTS + (size_t)passes * RADIX * ntiles), s);
        if (err != hipSuccess) return (int)err;
        int tot_grid = ntiles < 512 ? ntiles : 512;
        hipLaunchKernelGGL(radix_totals_kernel, dim3(tot_grid), dim3(TPB), 0, s, keys_a, n, n_dev, begin_bit, passes, last_mask, totals);
    }
    uint32_t *kin = keys_a, *vin = vals_a, *kout = keys_b, *vout = vals_b;
    for (int p = 0; p < passes; p++) {
        int shift = begin_bit + p * RADIX_BITS;
        uint32_t mask = (p == passes - 1) ? last_mask : (uint32_t)(RADIX - 1);
        if (inline_scan) {
            hipLaunchKernelGGL(radix_hist_kernel, dim3(ntiles), dim3(TPB), 0, s, kin, n, n_dev, shift, mask, ntiles, table);
            hipLaunchKernelGGL(radix_scatter_kernel<true>, dim3(ntiles), dim3(TPB), 0, s, kin, vin, kout, vout, table, n, n_dev, shift, mask, ntiles);
        } else {
            launch_onesweep(ntiles, s, kin, vin, kout, vout, totals + p * RADIX, (uint32_t*)table + (size_t)p * RADIX * ntiles, ticket + p, n,
                            n_dev, shift, mask, nullptr, nullptr);
        }
        uint32_t* t;
        t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    LG_RETURN_LAST();
}

long long lg_radix_table_words(long long n, int passes)
{
    long long ntiles = (n + SORT_TILE - 1) / SORT_TILE;
    if (ntiles < 1) ntiles = 1;
    return (long long)passes * RADIX * ntiles;
}

int lg_radix_sort_prepared(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, long long n, const int* n_dev,
                           int begin_bit, int end_bit, int* header, uint32_t* table, const int32_t* aux_in, int32_t* aux_sorted, void* stream)
{
    return lg_radix_sort_prepared_values(keys_a, vals_a, keys_b, vals_b, n, n_dev, begin_bit, end_bit, header, table, aux_in, aux_sorted, 0, nullptr, 0, nullptr, nullptr, stream);
}

// The words tile_range_kernel writes besides the starts (which the sort's last pass left by atomicMin): a tile without entries behind one
// with entries gets the end of that one's run (= the start of the next tile that has entries), the table's last word its length.
__global__ void __launch_bounds__(1024) tile_range_close_kernel(int32_t* __restrict__ out, int max_tile, long long L, const int* __restrict__ n_dev,
                                                                int* __restrict__ zero32 /*nullable: 32 words cleared on the side (the blend's unit counters)*/)
{
    if (zero32 != nullptr && threadIdx.x < 32) zero32[threadIdx.x] = 0;
    constexpr int PER = 16;
    __shared__ unsigned int wmin[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nb = max_tile + 1;                       // bins 0 .. max_tile
    const long long total = bounded_n(L, n_dev);
    if (total <= 0) return;                            // nothing was sorted (a gated fallback pass that is not needed): the table is not this launch's
    unsigned int carry = 0xffffffffu;                  // smallest start among the bins behind the current chunk
    for (int top = ((nb + 1024 * PER - 1) / (1024 * PER)) * (1024 * PER); top > 0; top -= 1024 * PER) {
        const int b0 = top - 1024 * PER + t * PER;     // this thread's bins b0 .. b0 + PER - 1
        unsigned int v[PER + 1];
#pragma unroll
        for (int k = 0; k < PER; k++) v[k + 1] = (b0 + k < nb) ? (unsigned int)out[b0 + k] : 0xffffffffu;
        v[0] = (b0 - 1 >= 0 && b0 - 1 < nb) ? (unsigned int)out[b0 - 1] : 0xffffffffu;
        unsigned int m = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < PER; k++) m = min(m, v[k + 1]);
        // suffix minimum over the threads behind this one (higher t)
        unsigned int suf = m;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned int u = __shfl_down(suf, o); if (lane + o < 64) suf = min(suf, u); }
        if (lane == 0) wmin[wave] = suf;
        __syncthreads();
        unsigned int behind = carry;
        for (int w = wave + 1; w < 16; w++) behind = min(behind, wmin[w]);
        const unsigned int nxt = __shfl_down(suf, 1);
        unsigned int after = (lane < 63) ? min(behind, nxt) : behind;          // smallest start in the bins behind this thread's
        unsigned int chunk_min = carry;
        for (int w = 0; w < 16; w++) chunk_min = min(chunk_min, wmin[w]);
        __syncthreads();
#pragma unroll
        for (int k = PER - 1; k >= 0; k--) {
            if (b0 + k < nb && v[k + 1] == 0xffffffffu && v[k] != 0xffffffffu && after != 0xffffffffu) out[b0 + k] = (int32_t)after;
            after = min(after, v[k + 1]);
        }
        carry = chunk_min;
    }
    if (t == 0) out[max_tile + 1] = (int32_t)total;
}

// value_bits > 0: every value is below 2^value_bits (the caller's promise); a two-pass sort whose second digit fits beside such a value in
// one word moves 8 bytes per element less (radix_onesweep_kernel PACK).  Same result in the same buffers.
int lg_radix_sort_prepared_values(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, long long n, const int* n_dev,
                                  int begin_bit, int end_bit, int* header, uint32_t* table, const int32_t* aux_in, int32_t* aux_sorted,
                                  int value_bits, int32_t* range_out /*nullable: tile range table pre-filled with -1*/, int max_tile,
                                  int* ranges_done /*nullable: set to 1 when the sort wrote range_out itself (the sorted KEYS then do not exist)*/,
                                  int* zero32 /*nullable: 32 words the range kernel clears on the side when ranges_done*/, void* stream)
{
    if (ranges_done) *ranges_done = 0;
    int passes = lg_radix_sort_num_passes(begin_bit, end_bit);
    if (n <= 0 || passes == 0) return 0;
    if (passes > SORT_MAX_PASSES || n > 0x3fffffffLL) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    int ntiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
    int* totals = header;
    int* ticket = header + LG_SORT_TOTALS_COPIES * LG_SORT_TOTALS_STRIDE;
    int last_bits = (end_bit - begin_bit) - (passes - 1) * RADIX_BITS;
    uint32_t last_mask = (1u << last_bits) - 1u;
    uint32_t *kin = keys_a, *vin = vals_a, *kout = keys_b, *vout = vals_b;
    if (g_sort_pack && passes == 2 && begin_bit == 0 && value_bits > 0 && last_bits + value_bits <= 32 && aux_in == nullptr) {
        launch_onesweep(ntiles, s, keys_a, vals_a, keys_b, vals_b, totals, table, ticket, n, n_dev, 0, (uint32_t)(RADIX - 1), nullptr, nullptr, 1, value_bits);
        const bool ranges = g_sort_pack == 2 && range_out != nullptr && ranges_done != nullptr && max_tile < (1 << (RADIX_BITS + last_bits));
        launch_onesweep(ntiles, s, keys_b, vals_b, keys_a, vals_a, totals + RADIX, table + (size_t)RADIX * ntiles, ticket + 1, n, n_dev, RADIX_BITS, last_mask,
                        nullptr, nullptr, ranges ? 3 : 2, value_bits, totals, range_out, max_tile);
        if (ranges) {
            hipLaunchKernelGGL(tile_range_close_kernel, dim3(1), dim3(1024), 0, s, range_out, max_tile, n, n_dev, zero32);
            *ranges_done = 1;
        }
        LG_RETURN_LAST();
    }
    for (int p = 0; p < passes; p++) {
        int shift = begin_bit + p * RADIX_BITS;
        uint32_t mask = (p == passes - 1) ? last_mask : (uint32_t)(RADIX - 1);
        const bool last = p == passes - 1;
        launch_onesweep(ntiles, s, kin, vin, kout, vout, totals + p * RADIX, table + (size_t)p * RADIX * ntiles, ticket + p, n, n_dev, shift, mask,
                        last ? aux_in : (const int32_t*)nullptr, last ? aux_sorted : (int32_t*)nullptr);
        uint32_t* t;
        t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    LG_RETURN_LAST();
}

// The executor's tile sort as an operator of its own (tests; what lg_fused_stage2 runs between the key emission and the blend): (key,
// value) pairs with keys in 0..max_tile and values below 2^value_bits -> values grouped by key in their input order (vals_a when the
// number of passes is even, else vals_b) and the tile range table of tileRange (GR/binning.cu:228-287) in range_out[max_tile + 2].  With two
// passes and a second digit that fits beside the value the passes are packed and the last one leaves the ranges itself (radix_onesweep_kernel
// PACK); *ranges_from_sort says which route ran.  The sorted KEYS exist only when it is 0.
LG_API int lg_tile_sort_ranges(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, long long n, const int* n_dev, int max_tile,
                               int value_bits, void* temp, long long temp_bytes, int32_t* range_out, int* ranges_from_sort, void* stream)
{
    LG_REQUIRE(range_out, ranges_from_sort);
    *ranges_from_sort = 0;
    if (max_tile < 0) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    hipError_t err = hipMemsetAsync(range_out, 0xFF, sizeof(int32_t) * ((size_t)max_tile + 2), s);
    if (err != hipSuccess) return (int)err;
    if (n <= 0) return 0;
    LG_REQUIRE(keys_a, vals_a, keys_b, vals_b, temp);
    int end_bit = 1;
    while ((1ll << end_bit) <= (long long)max_tile) end_bit++;
    const int passes = lg_radix_sort_num_passes(0, end_bit);
    if (passes > SORT_MAX_PASSES || n > 0x3fffffffLL || temp_bytes < lg_radix_sort_temp_bytes(n)) return (int)hipErrorInvalidValue;
    const int ntiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
    int* totals = (int*)temp;
    err = hipMemsetAsync(totals, 0, sizeof(int) * (SORT_HEADER_INTS + (size_t)passes * RADIX * ntiles), s);
    if (err != hipSuccess) return (int)err;
    const int last_bits = end_bit - (passes - 1) * RADIX_BITS;
    hipLaunchKernelGGL(radix_totals_kernel, dim3(ntiles < 512 ? ntiles : 512), dim3(TPB), 0, s, keys_a, n, n_dev, 0, passes, (1u << last_bits) - 1u, totals);
    int rc = lg_radix_sort_prepared_values(keys_a, vals_a, keys_b, vals_b, n, n_dev, 0, end_bit, totals, (uint32_t*)(totals + SORT_HEADER_INTS), nullptr, nullptr,
                                           value_bits, range_out, max_tile, ranges_from_sort, nullptr, stream);
    if (rc) return rc;
    if (!*ranges_from_sort)
        rc = lg_tile_range_prefilled((const int32_t*)(passes % 2 == 1 ? keys_b : keys_a), 1, n, n_dev, max_tile, range_out, stream);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// create_table (GR/binning.cu:123-226) as one entry point for a single view: key/value emission that also counts the sort's digits,
// zero padding of an over-allocated table, then the prepared radix sort -- no counting pass over the keys, no pre-cleared key table.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) table_pad_kernel(const int32_t* __restrict__ total_ptr, long long L, int32_t* __restrict__ keys,
                                                        int* __restrict__ totals, int passes)
{
    const long long total = *total_ptr;
    if (total >= L) return;                                  // exact or truncated table: dup_small already closed it
    const long long gid = (long long)blockIdx.x * TPB + threadIdx.x, nth = (long long)gridDim.x * TPB;
    for (long long i = total + gid; i < L; i += nth) keys[i] = 0;             // key 0 = "no tile": sorts to the front (binning.cu:139-150)
    if (gid == 0)
        for (int p = 0; p < passes; p++) atomicAdd(&totals[p * RADIX], (int)(L - total));
}

struct TableLayout { size_t header, table, qcount, cleared, qentries, total; };
static TableLayout table_layout(int N, long long L, int passes)
{
    TableLayout f;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
    f.header = take(sizeof(int) * SORT_HEADER_INTS);
    f.table = take(sizeof(int) * (size_t)lg_radix_table_words(L, passes));
    f.qcount = take(sizeof(int) * DUP_NQ);
    f.cleared = o;                                           // everything up to here must be zero on entry: one memset
    f.qentries = take(sizeof(int) * (size_t)lg_dup_queue_entries(N, L));
    f.total = o;
    return f;
}

LG_API long long lg_create_table_temp_bytes(int N, long long table_len, int end_bit)
{
    return (long long)table_layout(N, table_len, lg_radix_sort_num_passes(0, end_bit)).total;
}

// Result in (keys_b, vals_b) when lg_radix_sort_num_passes(0, end_bit) is odd, else in (keys_a, vals_a).  keys/vals need no initialisation.
LG_API int lg_create_table(const float* ndc, const float* inv_cov, const float* opacity, const int32_t* prefix, const void* sorted_id,
                           int sorted_id_is_int64, int N, int H, int W, int TH, int TW, long long table_len, int end_bit,
                           int32_t* keys_a, int32_t* vals_a, int32_t* keys_b, int32_t* vals_b, void* temp, long long temp_bytes, void* stream)
{
    LG_REQUIRE(ndc, inv_cov, opacity, prefix, sorted_id, keys_a, vals_a, keys_b, vals_b);
    const int passes = lg_radix_sort_num_passes(0, end_bit);
    if (N <= 0 || table_len <= 0 || passes < 1 || passes > SORT_MAX_PASSES_DUP) return (int)hipErrorInvalidValue;
    const TableLayout f = table_layout(N, table_len, passes);
    if (temp == nullptr || temp_bytes < (long long)f.total) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)temp;
    hipError_t err = hipMemsetAsync(w, 0, f.cleared, s);
    if (err != hipSuccess) return (int)err;
    int* header = (int*)(w + f.header);
    int rc = lg_dup_emit(ndc, inv_cov, opacity, nullptr, prefix, sorted_id, sorted_id_is_int64, 1, N, H, W, TH, TW, table_len, keys_a, vals_a,
                         (int*)(w + f.qcount), (uint32_t*)(w + f.qentries), header, 0, end_bit, nullptr, 0, nullptr, 0, nullptr, 0, stream);
    if (rc) return rc;
    long long pad_blocks = lg_cdiv(table_len, (long long)TPB * 16);
    if (pad_blocks > 1024) pad_blocks = 1024;
    hipLaunchKernelGGL(table_pad_kernel, dim3((unsigned)pad_blocks), dim3(TPB), 0, s, prefix + (N - 1), table_len, keys_a, header, passes);
    return lg_radix_sort_prepared((uint32_t*)keys_a, (uint32_t*)vals_a, (uint32_t*)keys_b, (uint32_t*)vals_b, table_len, nullptr, 0, end_bit,
                                  header, (uint32_t*)(w + f.table), nullptr, nullptr, stream);
}

// depth keys: monotone float -> uint32 map (sign flip) + identity payload; replaces the key side of torch.sort
__global__ void __launch_bounds__(TPB) depth_keys_kernel(const float* __restrict__ depth, long long n, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals)
{
    long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    uint32_t u = __float_as_uint(depth[i]);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    keys[i] = u;
    vals[i] = (uint32_t)i;
}

// same, plus the digit counts of the four 8-bit passes (saves the sort's own counting launch)
__global__ void __launch_bounds__(TPB) depth_keys_hist_kernel(const float* __restrict__ depth, long long n, uint32_t* __restrict__ keys,
                                                              uint32_t* __restrict__ vals, int* __restrict__ totals)
{
    __shared__ int hist[4 * 256];
    for (int k = threadIdx.x; k < 4 * 256; k += TPB) hist[k] = 0;
    __syncthreads();
    const DigitSpec ds = { 0, 4, 255u };
    const long long stride = (long long)gridDim.x * TPB;
    for (long long i0 = (long long)blockIdx.x * TPB; i0 < n; i0 += stride) {
        const long long i = i0 + threadIdx.x;
        const bool act = i < n;
        uint32_t u = 0;
        if (act) {
            u = __float_as_uint(depth[i]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            keys[i] = u;
            vals[i] = (uint32_t)i;
        }
        digit_hist_add(hist, u, act, ds);
    }
    __syncthreads();
    digit_hist_flush(hist, totals, 4);
}

int lg_depth_keys_hist(const float* depth, long long n, uint32_t* keys, uint32_t* vals, int* header, void* stream)
{
    if (n <= 0) return 0;
    long long blocks = lg_cdiv(n, (long long)TPB * 16);         // >= 16 keys per thread: amortises the flush of the 1024-entry LDS table
    if (blocks > g_depth_hist_blocks) blocks = g_depth_hist_blocks;
    hipLaunchKernelGGL(depth_keys_hist_kernel, dim3((unsigned)blocks), dim3(TPB), 0, (hipStream_t)stream, depth, n, keys, vals, header);
    LG_RETURN_LAST();
}

LG_API int lg_depth_sort_keys(const float* depth, long long n, uint32_t* keys, uint32_t* vals, void* stream)
{
    if (n <= 0) return 0;
    LG_REQUIRE(depth, keys, vals);
    hipLaunchKernelGGL(depth_keys_kernel, dim3(lg_cdiv(n, TPB)), dim3(TPB), 0, (hipStream_t)stream, depth, n, keys, vals);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// Inclusive int32 prefix sum with an optional gather: out[j] = sum_{k<=j} src[idx ? idx[k] : k]
// (litegs/utils/wrapper.py:740-745: allocate_size gathered into depth order, then cumsum).
// Three launches: tile sums, spine scan (single workgroup), tile scan + base.
// ---------------------------------------------------------------------------------------------
template <typename IdxT>
__device__ __forceinline__ int scan_load(const int32_t* __restrict__ src, const IdxT* __restrict__ idx, long long k, long long n)
{
    if (k >= n) return 0;
    return idx ? src[(long long)idx[k]] : src[k];
}

template <typename IdxT>
__global__ void __launch_bounds__(TPB) scan_tile_sums_kernel(const int32_t* __restrict__ src, const IdxT* __restrict__ idx, long long n,
                                                             int* __restrict__ tile_sums)
{
    __shared__ int wsum[TPB / 64];
    long long base = (long long)blockIdx.x * SORT_TILE + (long long)threadIdx.x * SORT_ITEMS;
    int s = 0;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) s += scan_load(src, idx, base + j, n);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void __launch_bounds__(1024) scan_spine_kernel(int* __restrict__ tile_sums, int ntiles)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int carry = 0;
    for (int start = 0; start < ntiles; start += 1024) {
        int k = start + tid;
        int v = (k < ntiles) ? tile_sums[k] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int nb = __shfl_up(incl, off);
            if (lane >= off) incl += nb;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; w++) wbase += wsum[w];
        if (k < ntiles) tile_sums[k] = carry + wbase + incl - v;    // exclusive
        if (tid == 1023) carry_s = carry + wbase + incl;
        __syncthreads();
        carry = carry_s;
        __syncthreads();
    }
}

template <typename IdxT>
__global__ void __launch_bounds__(TPB) scan_apply_kernel(const int32_t* __restrict__ src, const IdxT* __restrict__ idx, long long n,
                                                         const int* __restrict__ tile_base, int32_t* __restrict__ out)
{
    __shared__ int wsum[TPB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long base = (long long)blockIdx.x * SORT_TILE + (long long)tid * SORT_ITEMS;
    int v[SORT_ITEMS];
    int s = 0;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) { v[j] = scan_load(src, idx, base + j, n); s += v[j]; }
    int incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int nb = __shfl_up(incl, off);
        if (lane >= off) incl += nb;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = tile_base[blockIdx.x] + incl - s;
    for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        run += v[j];
        if (base + j < n) out[base + j] = run;
    }
}

// Single-launch variant for the fused executor: tile sums are chained with decoupled look-back (status words as in the radix
// sort; wave 0 inspects 64 predecessors per step), so the gathered values are read once and there is no spine launch.
// status[ntiles] and ticket[1] must be zero on entry.  host_total (nullable): pinned host int that receives out[n-1]
// (the GPU-driven sizing feedback, litegs/data.py:238) -- stored by the kernel itself instead of a copy launch.
// mode: how a source word is read -- 0 as is; 1 "culled view" of an encoded tile count (sign bit = splat removed by the depth-bound
// culling: counts 0); 2 "full view" (sign bit masked off).  gate (nullable): the launch does nothing unless *gate != 0 (then
// total_out, if given, receives 0); total_out (nullable): device copy of out[n-1].
template <typename IdxT>
__global__ void __launch_bounds__(TPB) scan_lookback_kernel(const int32_t* __restrict__ src, const IdxT* __restrict__ idx, long long n,
                                                            int32_t* __restrict__ out, uint32_t* __restrict__ status,
                                                            int* __restrict__ ticket, int* __restrict__ host_total,
                                                            int mode, const int* __restrict__ gate, int* __restrict__ total_out)
{
    __shared__ int wsum[TPB / 64];
    __shared__ int bid_s, excl_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (gate != nullptr && *gate == 0) {
        if (total_out != nullptr && blockIdx.x == 0 && tid == 0) *total_out = 0;
        return;
    }
    if (tid == 0) bid_s = atomicAdd(ticket, 1);
    __syncthreads();
    const int bid = bid_s;
    const long long base = (long long)bid * SORT_TILE + (long long)tid * SORT_ITEMS;
    int v[SORT_ITEMS];
    int s = 0;
    // a thread owns SORT_ITEMS consecutive words: four 16-byte loads instead of sixteen 4-byte ones whose lanes sit 64 bytes apart
    // (the scalar form re-touches every cache line of the wave sixteen times: 45 us for 3 M words, latency of the L1 miss path)
    const bool vec = (idx == nullptr) && (base + SORT_ITEMS <= n) && ((reinterpret_cast<uintptr_t>(src + base) & 15) == 0);
    if (vec) {
        const int4* __restrict__ s4 = reinterpret_cast<const int4*>(src + base);
#pragma unroll
        for (int q = 0; q < SORT_ITEMS / 4; q++) {
            const int4 x4 = s4[q];
            v[4 * q] = x4.x; v[4 * q + 1] = x4.y; v[4 * q + 2] = x4.z; v[4 * q + 3] = x4.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; j++) v[j] = scan_load(src, idx, base + j, n);
    }
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        int x = v[j];
        if (mode == 1) x = x < 0 ? 0 : x;
        else if (mode == 2) x &= 0x7fffffff;
        v[j] = x; s += x;
    }
    int incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int nb = __shfl_up(incl, off);
        if (lane >= off) incl += nb;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    const int tile_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (wave == 0) {
        uint32_t excl = 0;
        if (bid == 0) {
            if (lane == 0) __hip_atomic_store(status, ST_INC | (uint32_t)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(status + bid, ST_AGG | (uint32_t)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int b = bid - 1;
            while (true) {
                const uint32_t w = (b - lane >= 0) ? __hip_atomic_load(status + (b - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ST_INC;
                const unsigned long long inc_m = __ballot((w & ST_INC) != 0u);
                const unsigned long long nr_m = __ballot((w & (ST_AGG | ST_INC)) == 0u);
                const int first_inc = inc_m ? __ffsll((long long)inc_m) - 1 : 64;
                const int first_nr = nr_m ? __ffsll((long long)nr_m) - 1 : 64;
                const int take = first_nr < first_inc ? first_nr : (first_inc < 64 ? first_inc + 1 : 64);   // lanes [0, take) are usable
                uint32_t part = (lane < take) ? (w & ST_VAL) : 0u;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
                excl += __shfl(part, 0);
                if (first_inc < first_nr) break;                   // reached an inclusive prefix
                b -= take;
                if (take < 64) __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) __hip_atomic_store(status + bid, ST_INC | (excl + (uint32_t)tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) excl_s = (int)excl;
    }
    __syncthreads();
    int run = excl_s + incl - s;
    for (int w = 0; w < wave; w++) run += wsum[w];
    const bool vec_out = (base + SORT_ITEMS <= n) && ((reinterpret_cast<uintptr_t>(out + base) & 15) == 0);
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        run += v[j];
        v[j] = run;
        if (!vec_out && base + j < n) out[base + j] = run;
        if (host_total && base + j == n - 1) __hip_atomic_store(host_total, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (total_out && base + j == n - 1) *total_out = run;
    }
    if (vec_out) {
        int4* __restrict__ o4 = reinterpret_cast<int4*>(out + base);
#pragma unroll
        for (int q = 0; q < SORT_ITEMS / 4; q++) o4[q] = make_int4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
}

// status: lg_scan_status_words(n) zero words followed by one zero ticket word
long long lg_scan_status_words(long long n) { return (n + SORT_TILE - 1) / SORT_TILE + 1; }

int lg_gather_scan_prepared(const int32_t* src, const int32_t* idx, long long n, int32_t* out, uint32_t* status, int* host_total, void* stream)
{
    return lg_gather_scan_gated(src, idx, n, out, status, host_total, 0, nullptr, nullptr, stream);
}

int lg_gather_scan_gated(const int32_t* src, const int32_t* idx, long long n, int32_t* out, uint32_t* status, int* host_total,
                         int mode, const int* gate, int* total_out, void* stream)
{
    if (n <= 0) return 0;
    int ntiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
    hipLaunchKernelGGL(scan_lookback_kernel<int32_t>, dim3(ntiles), dim3(TPB), 0, (hipStream_t)stream, src, idx, n, out, status,
                       (int*)(status + ntiles), host_total, mode, gate, total_out);
    LG_RETURN_LAST();
}

LG_API long long lg_scan_temp_bytes(long long n)
{
    long long ntiles = (n + SORT_TILE - 1) / SORT_TILE;
    return sizeof(int) * (ntiles < 1 ? 1 : ntiles);
}

LG_API int lg_gather_inclusive_scan(const int32_t* src, const void* idx, int idx_is_int64, long long n, int32_t* out,
                                    void* temp, long long temp_bytes, void* stream)
{
    if (n <= 0) return 0;
    LG_REQUIRE(src, out);
    if (temp_bytes < lg_scan_temp_bytes(n)) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    int ntiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
    int* sums = (int*)temp;
    if (idx_is_int64) {
        hipLaunchKernelGGL(scan_tile_sums_kernel<int64_t>, dim3(ntiles), dim3(TPB), 0, s, src, (const int64_t*)idx, n, sums);
        hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(1024), 0, s, sums, ntiles);
        hipLaunchKernelGGL(scan_apply_kernel<int64_t>, dim3(ntiles), dim3(TPB), 0, s, src, (const int64_t*)idx, n, sums, out);
    } else {
        hipLaunchKernelGGL(scan_tile_sums_kernel<int32_t>, dim3(ntiles), dim3(TPB), 0, s, src, (const int32_t*)idx, n, sums);
        hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(1024), 0, s, sums, ntiles);
        hipLaunchKernelGGL(scan_apply_kernel<int32_t>, dim3(ntiles), dim3(TPB), 0, s, src, (const int32_t*)idx, n, sums, out);
    }
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// a11 tileRange (GR/binning.cu:228-287): out[V, max_tile+2]; start of each tile's run, -1 if empty,
// out[cur+1] closes a run that is followed by a gap, out[max_tile+1] = table length.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) tile_range_kernel(const int32_t* __restrict__ sorted_keys, long long L, const int* __restrict__ n_dev,
                                                         int max_tile, int32_t* __restrict__ out)
{
    // 4 consecutive keys per thread (one 16-byte load) + the key after them
    const long long i0 = ((long long)blockIdx.x * TPB + threadIdx.x) * 4;
    const int b = blockIdx.y;
    const long long stride = L;
    L = bounded_n(L, n_dev);
    if (L <= 0 || i0 >= L) return;
    const int32_t* k = sorted_keys + (size_t)b * stride;
    int32_t* o = out + (size_t)b * (max_tile + 2);
    int key[5];
    if (i0 + 4 < L && ((stride & 3) == 0)) {
        const int4 q = *reinterpret_cast<const int4*>(k + i0);
        key[0] = q.x; key[1] = q.y; key[2] = q.z; key[3] = q.w; key[4] = k[i0 + 4];
    } else {
#pragma unroll
        for (int j = 0; j < 5; j++) key[j] = (i0 + j < L) ? k[i0 + j] : 0;
    }
    // A key outside 0..max_tile cannot come out of a correct table; if one does (an entry the emission left unwritten, DESIGN.md section 9
    // "memory access fault"), it must not become a store address: such boundaries are skipped, the tile keeps "empty".
    int skipped = 0;
    if (i0 == 0) { if ((unsigned)key[0] <= (unsigned)max_tile) o[key[0]] = 0; else skipped++; }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const long long i = i0 + j;
        if (i == L - 1) o[max_tile + 1] = (int32_t)L;
        if (i < L - 1) {
            const int cur = key[j], nxt = key[j + 1];
            if (cur != nxt && (unsigned)nxt <= (unsigned)max_tile) {
                if (cur + 1 < nxt && cur >= -1) o[cur + 1] = (int32_t)(i + 1);
                o[nxt] = (int32_t)(i + 1);
            } else if (cur != nxt) skipped++;
        }
    }
    if (skipped) lg_note_sanitised(LG_SITE_RANGE_KEY, skipped);
}

// `out` already filled with -1 (by a producer kernel's fill duty)
int lg_tile_range_prefilled(const int32_t* sorted_keys, int V, long long L, const int* n_dev, int max_tile, int32_t* out, void* stream)
{
    if (L <= 0) return 0;
    hipLaunchKernelGGL(tile_range_kernel, dim3(lg_cdiv(L, TPB * 4), V), dim3(TPB), 0, (hipStream_t)stream, sorted_keys, L, n_dev, max_tile, out);
    LG_RETURN_LAST();
}

LG_API int lg_tile_range_bounded(const int32_t* sorted_keys, int V, long long L, const int* n_dev, int max_tile, int32_t* out, void* stream);

LG_API int lg_tile_range(const int32_t* sorted_keys, int V, long long L, int max_tile, int32_t* out, void* stream)
{
    return lg_tile_range_bounded(sorted_keys, V, L, nullptr, max_tile, out, stream);
}

// n_dev (nullable): only the first min(L, *n_dev) sorted entries are a valid table (see lg_radix_sort_pairs_bounded)
LG_API int lg_tile_range_bounded(const int32_t* sorted_keys, int V, long long L, const int* n_dev, int max_tile, int32_t* out, void* stream)
{
    LG_REQUIRE(out);
    if (L > 0) LG_REQUIRE(sorted_keys);
    hipStream_t s = (hipStream_t)stream;
    hipError_t err = hipMemsetAsync(out, 0xFF, sizeof(int32_t) * (size_t)V * (max_tile + 2), s);
    if (err != hipSuccess) return (int)err;
    if (L <= 0) return 0;
    hipLaunchKernelGGL(tile_range_kernel, dim3(lg_cdiv(L, TPB * 4), V), dim3(TPB), 0, s, sorted_keys, L, n_dev, max_tile, out);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// Tile scatter (executor, per-tile-depth-sort mode; no reference counterpart).  The reference groups the instances by tile with a
// STABLE radix sort because its emission order (depth) must survive inside every tile (GR/binning.cu:205-220).  When every tile's list is
// re-ordered by (depth, id) afterwards (tilesort.hip) the order inside a tile is irrelevant, and grouping needs no sort at all: the
// emission kernels count the instances per key (one fire-and-forget atomic each, lg_dup_emit_gated's tile_counts), one workgroup turns
// the counts into the range table -- the very words tile_range_kernel would derive from the sorted keys -- and into write cursors, and
// one pass drops every value at its tile's cursor.  Two radix passes (4 x 16 B per instance, rank + look-back chains) and the range
// scan are replaced by 8 B read + 4 B written per instance.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) tile_offsets_kernel(const int* __restrict__ counts /*[max_tile + 2], bins 0..max_tile*/, int max_tile,
                                                            const int* __restrict__ n_dev, long long L, int* __restrict__ cursor /*[max_tile + 2]*/,
                                                            int32_t* __restrict__ out /*[max_tile + 2], pre-filled with -1*/, const int* __restrict__ gate)
{
    if (gate != nullptr && *gate == 0) return;
    // One workgroup; the counts pass through LDS so that every global access is coalesced (a thread scanning 16 consecutive bins
    // straight from memory touches a cache line of its own per load: 23 us for 16 201 bins; this form: a third of that).
    constexpr int PER = 8, CHUNK = 1024 * PER;
    __shared__ int sc[CHUNK + 1];          // sc[1 + i] = count of bin base + i; sc[0] = count of the bin in front of the chunk
    __shared__ int so[CHUNK];              // range-table word of bin base + i (or -1)
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nb = max_tile + 1;
    const int total = (int)bounded_n(L, n_dev);              // == the sum of the counts: every emitted entry (padding included) was counted
    int carry = 0;
    for (int base = 0; base < nb; base += CHUNK) {
        for (int i = t; i < CHUNK; i += 1024) sc[1 + i] = (base + i < nb) ? counts[base + i] : 0;
        if (t == 0) sc[0] = base > 0 ? counts[base - 1] : 0;
        __syncthreads();
        int c[PER], s = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) { c[k] = sc[1 + t * PER + k]; s += c[k]; }
        int prev = sc[t * PER];
        int incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int nbv = __shfl_up(incl, o); if (lane >= o) incl += nbv; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0, round_total = 0;
        for (int w = 0; w < 16; w++) { if (w < wave) wbase += wsum[w]; round_total += wsum[w]; }
        int p = carry + wbase + incl - s;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = t * PER + k;
            // tileRange (GR/binning.cu:228-264): a run's start; the word after a run that is followed by a gap and a later run
            so[i] = (c[k] > 0 || (prev > 0 && p < total)) ? p : -1;
            sc[1 + i] = p;                                    // becomes the cursor (each thread rewrites only its own slots)
            p += c[k];
            prev = c[k];
        }
        carry += round_total;
        __syncthreads();
        for (int i = t; i < CHUNK && base + i < nb; i += 1024) {
            cursor[base + i] = sc[1 + i];
            if (so[i] >= 0) out[base + i] = so[i];
        }
        __syncthreads();
    }
    if (t == 0 && total > 0) out[max_tile + 1] = total;
}

// Global atomics are the expensive primitive here (measured on MI355X, profiles/r03_scatter_ab.log: one atomic per instance for the
// counts plus one returning atomic per instance for the cursors -- 8 M at 3 M Gaussians -- cost 110 us MORE than the two radix passes
// they replaced).  So both passes aggregate in LDS first: a workgroup takes TG_ITEMS consecutive instances -- neighbours in emission
// order are spatial neighbours and share their tiles -- counts them per key in a 16-bit LDS histogram (two keys per 32-bit word; a
// workgroup's count of one key is at most TG_ITEMS < 65536, so the halves never carry into each other) and goes to global memory once
// per (workgroup, key): a few hundred atomics per 4096 instances.
#define TG_BINS 16384                      // keys 0 .. TG_BINS - 1 (1080p at 8x16: 16 201)
#define TG_PER_THREAD 16
#define TG_ITEMS (TPB * TG_PER_THREAD)     // 4096 instances per workgroup

__device__ __forceinline__ void tg_load(const int32_t* __restrict__ src, long long i0, long long n, int out[TG_PER_THREAD], int fill)
{
    // thread t takes items i0 + t * 4 + r * (TPB * 4) + {0..3}: 16-byte loads, a wave covers 1 KB per round
#pragma unroll
    for (int r = 0; r < TG_PER_THREAD / 4; r++) {
        const long long i = i0 + (long long)r * (TPB * 4) + (long long)threadIdx.x * 4;
        if (i + 3 < n) {
            const int4 q = *reinterpret_cast<const int4*>(src + i);
            out[4 * r] = q.x; out[4 * r + 1] = q.y; out[4 * r + 2] = q.z; out[4 * r + 3] = q.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) out[4 * r + j] = (i + j < n) ? src[i + j] : fill;
        }
    }
}

__global__ void __launch_bounds__(TPB) tile_count_lds_kernel(const int32_t* __restrict__ keys, long long L, const int* __restrict__ n_dev,
                                                             int* __restrict__ counts, const int* __restrict__ gate, int max_tile)
{
    if (gate != nullptr && *gate == 0) return;
    __shared__ unsigned int hist[TG_BINS / 2];
    const long long n = bounded_n(L, n_dev);
    const long long i0 = (long long)blockIdx.x * TG_ITEMS;
    if (i0 >= n) return;
    for (int w = threadIdx.x; w < TG_BINS / 2; w += TPB) hist[w] = 0u;
    __syncthreads();
    int k[TG_PER_THREAD];
    tg_load(keys, i0, n, k, -1);
    int dropped = 0;
#pragma unroll
    for (int j = 0; j < TG_PER_THREAD; j++)
        if ((unsigned)k[j] <= (unsigned)max_tile) atomicAdd(&hist[k[j] >> 1], 1u << ((k[j] & 1) * 16));      // (a key is an index: range-checked)
        else dropped += (i0 + (long long)(j >> 2) * (TPB * 4) + (long long)threadIdx.x * 4 + (j & 3) < n) ? 1 : 0;
    if (dropped) lg_note_sanitised(LG_SITE_SCATTER_KEY, dropped);
    __syncthreads();
    for (int w = threadIdx.x; w < TG_BINS / 2; w += TPB) {
        const unsigned int c = hist[w];
        if (c & 0xffffu) atomicAdd(&counts[2 * w], (int)(c & 0xffffu));
        if (c >> 16) atomicAdd(&counts[2 * w + 1], (int)(c >> 16));
    }
}

__global__ void __launch_bounds__(TPB) tile_scatter_lds_kernel(const int32_t* __restrict__ keys, const int32_t* __restrict__ vals, long long L,
                                                               const int* __restrict__ n_dev, int* __restrict__ cursor, int32_t* __restrict__ out_vals,
                                                               const int* __restrict__ gate, int max_tile)
{
    if (gate != nullptr && *gate == 0) return;
    __shared__ unsigned int hist[TG_BINS / 2];     // per key: first the workgroup's count, then the index of the key's entry in base[]
    __shared__ int base[TG_ITEMS];                 // start of this workgroup's run inside the key's segment (at most one entry per instance)
    __shared__ int nlist;
    const long long n = bounded_n(L, n_dev);
    const long long i0 = (long long)blockIdx.x * TG_ITEMS;
    if (i0 >= n) return;
    for (int w = threadIdx.x; w < TG_BINS / 2; w += TPB) hist[w] = 0u;
    if (threadIdx.x == 0) nlist = 0;
    __syncthreads();
    int k[TG_PER_THREAD], v[TG_PER_THREAD], rank[TG_PER_THREAD];
    tg_load(keys, i0, n, k, -1);
    tg_load(vals, i0, n, v, 0);
#pragma unroll
    for (int j = 0; j < TG_PER_THREAD; j++) {
        rank[j] = 0;
        if ((unsigned)k[j] > (unsigned)max_tile) k[j] = -1;        // a key is an index (LDS bin, cursor): outside 0..max_tile it is dropped
        if (k[j] >= 0) {
            const int sh = (k[j] & 1) * 16;
            rank[j] = (int)((atomicAdd(&hist[k[j] >> 1], 1u << sh) >> sh) & 0xffffu);       // position inside the workgroup's run of this key
        }
    }
    __syncthreads();
    // the keys this workgroup holds, compacted: base[e] = key << 16 | count for now (key < 2^14, count <= 4096)
    for (int w = threadIdx.x; w < TG_BINS / 2; w += TPB) {
        const unsigned int c = hist[w];
        if (c == 0u) continue;
        unsigned int packed = 0u;
        if (c & 0xffffu) {
            const int e = atomicAdd(&nlist, 1);
            base[e] = (int)(((unsigned int)(2 * w) << 16) | (c & 0xffffu));
            packed |= (unsigned int)e;
        }
        if (c >> 16) {
            const int e = atomicAdd(&nlist, 1);
            base[e] = (int)(((unsigned int)(2 * w + 1) << 16) | (c >> 16));
            packed |= (unsigned int)e << 16;
        }
        hist[w] = packed;                           // entry indices are < TG_ITEMS <= 65535
    }
    __syncthreads();
    // one returning atomic per (workgroup, key), spread evenly over the threads so that their round trips overlap (issued from the
    // scan loop above, a thread that happens to own several non-empty words would pay them one after the other)
    const int nl = nlist;
    for (int e = threadIdx.x; e < nl; e += TPB) {
        const unsigned int kc = (unsigned int)base[e];
        base[e] = atomicAdd(&cursor[kc >> 16], (int)(kc & 0xffffu));
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TG_PER_THREAD; j++)
        if (k[j] >= 0) {
            const int e = (int)((hist[k[j] >> 1] >> ((k[j] & 1) * 16)) & 0xffffu);
            out_vals[base[e] + rank[j]] = v[j];
        }
}

__global__ void __launch_bounds__(TPB) tile_scatter_kernel(const int32_t* __restrict__ keys, const int32_t* __restrict__ vals, long long L,
                                                           const int* __restrict__ n_dev, int* __restrict__ cursor, int32_t* __restrict__ out_vals,
                                                           const int* __restrict__ gate, int max_tile)
{
    if (gate != nullptr && *gate == 0) return;
    const long long i0 = ((long long)blockIdx.x * TPB + threadIdx.x) * 4;
    const long long n = bounded_n(L, n_dev);
    if (i0 >= n) return;
    int k[4], v[4];
    if (i0 + 3 < n) {
        const int4 kq = *reinterpret_cast<const int4*>(keys + i0), vq = *reinterpret_cast<const int4*>(vals + i0);
        k[0] = kq.x; k[1] = kq.y; k[2] = kq.z; k[3] = kq.w; v[0] = vq.x; v[1] = vq.y; v[2] = vq.z; v[3] = vq.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) { k[j] = (i0 + j < n) ? keys[i0 + j] : -1; v[j] = (i0 + j < n) ? vals[i0 + j] : 0; }
    }
    int pos[4];
#pragma unroll
    for (int j = 0; j < 4; j++) pos[j] = (unsigned)k[j] <= (unsigned)max_tile ? atomicAdd(&cursor[k[j]], 1) : -1;       // four independent returning atomics in flight
#pragma unroll
    for (int j = 0; j < 4; j++) if (pos[j] >= 0) out_vals[pos[j]] = v[j];
}

// counts [max_tile + 2] (filled by the emission), cursor [max_tile + 2] scratch, tile_start [max_tile + 2] pre-filled with -1;
// keys / vals: the emitted table (capacity L, valid entries min(L, *n_dev)); out_vals: values grouped by tile (any order inside a tile)
__global__ void __launch_bounds__(TPB) tile_count_kernel(const int32_t* __restrict__ keys, long long L, const int* __restrict__ n_dev,
                                                         int* __restrict__ counts, const int* __restrict__ gate, int max_tile)
{
    if (gate != nullptr && *gate == 0) return;
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i < bounded_n(L, n_dev) && (unsigned)keys[i] <= (unsigned)max_tile) atomicAdd(&counts[keys[i]], 1);
}

// counts [max_tile + 2] zero on entry when count_keys != 0 (then counted here from the emitted keys, in LDS-aggregated form), else filled by
// the caller; cursor [max_tile + 2] scratch, tile_start [max_tile + 2] pre-filled with -1; keys / vals: the emitted table (capacity L, valid
// entries min(L, *n_dev)); out_vals: values grouped by tile (any order inside a tile)
int lg_tile_scatter_gated(const int32_t* keys, const int32_t* vals, long long L, const int* n_dev, int max_tile, int* counts, int count_keys,
                          int* cursor, int32_t* tile_start, int32_t* out_vals, const int* gate, void* stream)
{
    if (L <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool lds = max_tile + 1 <= TG_BINS;
    if (count_keys) {
        if (lds) hipLaunchKernelGGL(tile_count_lds_kernel, dim3(lg_cdiv(L, TG_ITEMS)), dim3(TPB), 0, s, keys, L, n_dev, counts, gate, max_tile);
        else hipLaunchKernelGGL(tile_count_kernel, dim3(lg_cdiv(L, TPB)), dim3(TPB), 0, s, keys, L, n_dev, counts, gate, max_tile);
    }
    hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(1024), 0, s, counts, max_tile, n_dev, L, cursor, tile_start, gate);
    if (lds) hipLaunchKernelGGL(tile_scatter_lds_kernel, dim3(lg_cdiv(L, TG_ITEMS)), dim3(TPB), 0, s, keys, vals, L, n_dev, cursor, out_vals, gate, max_tile);
    else hipLaunchKernelGGL(tile_scatter_kernel, dim3(lg_cdiv(L, TPB * 4)), dim3(TPB), 0, s, keys, vals, L, n_dev, cursor, out_vals, gate, max_tile);
    LG_RETURN_LAST();
}

// Stand-alone form of the tile scatter (tests, tools): an UNSORTED table keys[L] (0 = padding ... max_tile) / vals[L] -> tile_start
// [max_tile + 2] exactly as lg_tile_range leaves it for the sorted table, and out_vals[L] = the values grouped by key, in arbitrary
// order inside a key (follow with lg_tile_depth_sort_unordered).  temp: 2 * (max_tile + 2) ints.
LG_API int lg_tile_group(const int32_t* keys, const int32_t* vals, long long L, int max_tile, int32_t* tile_start, int32_t* out_vals,
                         void* temp, void* stream)
{
    LG_REQUIRE(tile_start, temp);
    hipStream_t s = (hipStream_t)stream;
    int* counts = (int*)temp;
    int* cursor = counts + (max_tile + 2);
    hipError_t err = hipMemsetAsync(tile_start, 0xFF, sizeof(int32_t) * (size_t)(max_tile + 2), s);
    if (err != hipSuccess) return (int)err;
    if (L <= 0) return 0;
    LG_REQUIRE(keys, vals, out_vals);
    err = hipMemsetAsync(counts, 0, sizeof(int) * (size_t)(max_tile + 2), s);
    if (err != hipSuccess) return (int)err;
    return lg_tile_scatter_gated(keys, vals, L, nullptr, max_tile, counts, 1, cursor, tile_start, out_vals, nullptr, stream);
}

LG_API int lg_memset_async(void* ptr, int value, long long bytes, void* stream)
{
    if (bytes <= 0) return 0;
    return (int)hipMemsetAsync(ptr, value, (size_t)bytes, (hipStream_t)stream);
}
