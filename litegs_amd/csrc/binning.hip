// Tile binning: exact ellipse/tile intersection count, key/value emission in depth order, the
// hand-written stable LSD radix sort ("the tile radix sort"), prefix scan and tile range table
// (SURVEY.md 8a rows a8-a11).  Integer/index work: outputs are BIT-EXACT against the CPU oracle,
// which is why this file is compiled with -ffp-contract=off and uses lg_logf (fixed polynomial).
//
// HBM traffic per tile instance (I of them): 8 B written by duplicate_with_keys, then per radix pass
// 4 B (histogram read) + 8 B (scatter read) + 8 B (scatter write); 14 tile-id bits at 1080p = 2 passes
// of 8 bits.  No MFMA: this is byte shuffling.
#include "lg_common.h"

#define TPB 256

// ---------------------------------------------------------------------------------------------
// Ellipse extent + AccuTile walk (reference: GR/binning.cu:310-373 and GR/speedy_splat.cuh:16-149).
// ---------------------------------------------------------------------------------------------
struct SplatExtent {
    float a, b, c, disc, t;
    float px, py;
    float bbox_min_x, bbox_min_y, bbox_max_x, bbox_max_y;
    float argmin_x, argmin_y, argmax_x, argmax_y;   // .x = along y, .y = along x (reference naming)
    int rminx, rminy, rmaxx, rmaxy;
};

__device__ __forceinline__ void ellipse_cut(const SplatExtent& e, bool isY, float coord, float& lo, float& hi)
{
    float p_u = isY ? e.py : e.px;
    float p_v = isY ? e.px : e.py;
    float coeff = isY ? e.a : e.c;
    float h = coord - p_u;
    float sq = sqrtf(e.disc * h * h + e.t * coeff);
    lo = (-e.b * h - sq) / coeff + p_v;
    hi = (-e.b * h + sq) / coeff + p_v;
}

template <int TH, int TW>
__device__ __forceinline__ void splat_extent(float ndcx, float ndcy, float ic00, float ic01, float ic11, float opacity,
                                             int H, int W, int gx, int gy, SplatExtent& e)
{
    e.a = ic00; e.b = ic01; e.c = ic11;
    e.disc = ic01 * ic01 - ic00 * ic11;
    float u = ndcx * 0.5f + 0.5f, v = ndcy * 0.5f + 0.5f;
    e.px = u * W - 0.5f;
    e.py = v * H - 0.5f;
    float t = 2.0f * lg_logf(opacity * 255.0f);
    e.t = t;
    float x_term = sqrtf(-(ic01 * ic01 * t) / (e.disc * ic00));
    x_term = (ic01 < 0) ? x_term : -x_term;
    float y_term = sqrtf(-(ic01 * ic01 * t) / (e.disc * ic11));
    y_term = (ic01 < 0) ? y_term : -y_term;
    e.argmin_x = e.py - y_term; e.argmin_y = e.px - x_term;
    e.argmax_x = e.py + y_term; e.argmax_y = e.px + x_term;
    float lo, hi;
    ellipse_cut(e, true, e.argmin_x, lo, hi);  e.bbox_min_x = lo;
    ellipse_cut(e, false, e.argmin_y, lo, hi); e.bbox_min_y = lo;
    ellipse_cut(e, true, e.argmax_x, lo, hi);  e.bbox_max_x = hi;
    ellipse_cut(e, false, e.argmax_y, lo, hi); e.bbox_max_y = hi;
    e.rminx = max(0, min(gx, lg_f2i(e.bbox_min_x / TW)));
    e.rminy = max(0, min(gy, lg_f2i(e.bbox_min_y / TH)));
    e.rmaxx = max(0, min(gx, lg_f2i((e.bbox_max_x + TW - 1) / TW)));
    e.rmaxy = max(0, min(gy, lg_f2i((e.bbox_max_y + TH - 1) / TH)));
}

// Walks tile slices along the shorter rect axis; returns tiles touched; EMIT writes (tile_id+1, idx).
template <int TH, int TW, bool EMIT>
__device__ __forceinline__ uint32_t walk_tiles(const SplatExtent& e, int gx, int32_t idx, long long off,
                                               int32_t* __restrict__ keys, int32_t* __restrict__ values)
{
    const int ys = e.rmaxy - e.rminy, xs = e.rmaxx - e.rminx;
    const bool isY = ys < xs;
    const float BLOCK_U = isY ? (float)TH : (float)TW;
    const float BLOCK_V = isY ? (float)TW : (float)TH;
    // (u,v) frame: u = slicing axis
    const int rect_min_u = isY ? e.rminy : e.rminx, rect_max_u = isY ? e.rmaxy : e.rmaxx;
    const int rect_min_v = isY ? e.rminx : e.rminy, rect_max_v = isY ? e.rmaxx : e.rmaxy;
    const float bmin_u = isY ? e.bbox_min_y : e.bbox_min_x, bmin_v = isY ? e.bbox_min_x : e.bbox_min_y;
    const float bmax_u = isY ? e.bbox_max_y : e.bbox_max_x, bmax_v = isY ? e.bbox_max_x : e.bbox_max_y;
    const float argmin_v = isY ? e.argmin_x : e.argmin_y;   // coordinate along u where v is minimal
    const float argmax_v = isY ? e.argmax_x : e.argmax_y;

    uint32_t count = 0;
    float imax_lo = bmax_v, imax_hi = bmin_v;              // "never selected" sentinels
    float imin_lo, imin_hi;
    float min_line = rect_min_u * BLOCK_U;
    if (bmin_u <= min_line) ellipse_cut(e, isY, rect_min_u * BLOCK_U, imin_lo, imin_hi);
    else { imin_lo = imax_lo; imin_hi = imax_hi; }

    for (int u = rect_min_u; u < rect_max_u; ++u) {
        float max_line = min_line + BLOCK_U;
        if (max_line <= bmax_u) ellipse_cut(e, isY, max_line, imax_lo, imax_hi);
        float ellipse_min, ellipse_max;
        if (min_line <= argmin_v && argmin_v < max_line) ellipse_min = bmin_v;
        else ellipse_min = fminf(imin_lo, imax_lo);
        if (min_line <= argmax_v && argmax_v < max_line) ellipse_max = bmax_v;
        else ellipse_max = fmaxf(imin_hi, imax_hi);
        int min_tile_v = max(rect_min_v, min(rect_max_v, lg_f2i(ellipse_min / BLOCK_V)));
        int max_tile_v = min(rect_max_v, max(rect_min_v, lg_f2i(ellipse_max / BLOCK_V + 1)));
        count += (uint32_t)(max_tile_v - min_tile_v);
        if (EMIT) {
            for (int v = min_tile_v; v < max_tile_v; v++) {
                uint32_t key = isY ? (uint32_t)(u * gx + v) : (uint32_t)(v * gx + u);
                keys[off] = (int32_t)(key + 1);
                values[off] = idx;
                off++;
            }
        }
        imin_lo = imax_lo; imin_hi = imax_hi;
        min_line = max_line;
    }
    return count;
}

// fminf/fmaxf above must behave like the oracle's (a<b?a:b): identical for non-NaN operands, and a NaN
// intersection only arises for degenerate ellipses that the visibility test already rejects.

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
// A 256-thread workgroup owns 256 consecutive depth slots, whose output ranges are CONTIGUOUS in the table
// (prefix order).  Each thread runs the serial AccuTile walk, but instead of scattering 4-byte stores (64 cache
// lines per wave instruction) it writes into an LDS window of DUP_WIN entries that the workgroup then streams
// out with coalesced stores; the walk is suspended / resumed at window boundaries.  Arithmetic is the walk of
// walk_tiles<> verbatim (same cuts, same order) so the emitted keys stay bit-identical to the oracle.
#define DUP_WIN 4096
template <int TH, int TW, typename IdxT>
__global__ void __launch_bounds__(TPB) duplicate_with_keys_kernel(const float* __restrict__ ndc, const float* __restrict__ inv_cov,
                                                                  const float* __restrict__ opacity, const int32_t* __restrict__ prefix,
                                                                  const IdxT* __restrict__ sorted_id, int N, int H, int W, int gx, int gy,
                                                                  long long table_len, int32_t* __restrict__ keys, int32_t* __restrict__ values)
{
    __shared__ int2 buf[DUP_WIN];
    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * TPB;
    const int j = j0 + tid;
    const int b = blockIdx.y;
    const int32_t* pf = prefix + (size_t)b * N;
    int32_t* kout = keys + (size_t)b * table_len;
    int32_t* vout = values + (size_t)b * table_len;
    const int jl = min(j0 + TPB, N) - 1;
    long long block_start = (j0 == 0) ? 0 : pf[j0 - 1];
    long long block_end = pf[jl];
    if (block_end > table_len) block_end = table_len;
    if (block_start >= block_end) return;

    // ---- per-thread walk state ----
    bool done = true;
    long long cur = 0;
    int idx = 0;
    SplatExtent e;
    bool isY = false;
    float BLOCK_U = 0.f, BLOCK_V = 0.f, bmax_u = 0.f, bmin_v = 0.f, bmax_v = 0.f, argmin_v = 0.f, argmax_v = 0.f;
    int u = 0, u_end = 0, rect_min_v = 0, rect_max_v = 0, v = 0, v_end = 0, cur_u = 0;
    float min_line = 0.f, imin_lo = 0.f, imin_hi = 0.f, imax_lo = 0.f, imax_hi = 0.f;
    if (j < N) {
        long long off = (j == 0) ? 0 : pf[j - 1];
        long long cnt = pf[j] - off;
        if (cnt > 0 && off + cnt <= table_len) {
            idx = (int)sorted_id[(size_t)b * N + j];
            float nx = ndc[((size_t)b * 4) * N + idx], ny = ndc[((size_t)b * 4 + 1) * N + idx];
            float a = inv_cov[((size_t)b * 4) * N + idx], bb = inv_cov[((size_t)b * 4 + 1) * N + idx], c = inv_cov[((size_t)b * 4 + 3) * N + idx];
            splat_extent<TH, TW>(nx, ny, a, bb, c, opacity[idx], H, W, gx, gy, e);
            const int ys = e.rmaxy - e.rminy, xs = e.rmaxx - e.rminx;
            if (ys * xs > 0) {
                done = false;
                cur = off;
                isY = ys < xs;
                BLOCK_U = isY ? (float)TH : (float)TW;
                BLOCK_V = isY ? (float)TW : (float)TH;
                u = isY ? e.rminy : e.rminx; u_end = isY ? e.rmaxy : e.rmaxx;
                rect_min_v = isY ? e.rminx : e.rminy; rect_max_v = isY ? e.rmaxx : e.rmaxy;
                const float bmin_u = isY ? e.bbox_min_y : e.bbox_min_x;
                bmin_v = isY ? e.bbox_min_x : e.bbox_min_y;
                bmax_u = isY ? e.bbox_max_y : e.bbox_max_x; bmax_v = isY ? e.bbox_max_x : e.bbox_max_y;
                argmin_v = isY ? e.argmin_x : e.argmin_y;
                argmax_v = isY ? e.argmax_x : e.argmax_y;
                imax_lo = bmax_v; imax_hi = bmin_v;
                min_line = u * BLOCK_U;
                if (bmin_u <= min_line) ellipse_cut(e, isY, u * BLOCK_U, imin_lo, imin_hi);
                else { imin_lo = imax_lo; imin_hi = imax_hi; }
            }
        }
    }
    // advance to the next non-empty slice (or finish)
    auto advance = [&]() {
        while (true) {
            if (u >= u_end) { done = true; return; }
            float max_line = min_line + BLOCK_U;
            if (max_line <= bmax_u) ellipse_cut(e, isY, max_line, imax_lo, imax_hi);
            float ellipse_min, ellipse_max;
            if (min_line <= argmin_v && argmin_v < max_line) ellipse_min = bmin_v;
            else ellipse_min = fminf(imin_lo, imax_lo);
            if (min_line <= argmax_v && argmax_v < max_line) ellipse_max = bmax_v;
            else ellipse_max = fmaxf(imin_hi, imax_hi);
            v = max(rect_min_v, min(rect_max_v, lg_f2i(ellipse_min / BLOCK_V)));
            v_end = min(rect_max_v, max(rect_min_v, lg_f2i(ellipse_max / BLOCK_V + 1)));
            cur_u = u;
            imin_lo = imax_lo; imin_hi = imax_hi;
            min_line = max_line;
            u++;
            if (v < v_end) return;
        }
    };
    if (!done) advance();

    for (long long w0 = block_start; w0 < block_end; w0 += DUP_WIN) {
        const long long w1 = (w0 + DUP_WIN < block_end) ? (w0 + DUP_WIN) : block_end;
        for (int k = tid; k < (int)(w1 - w0); k += TPB) buf[k] = make_int2(0, 0);      // holes of dropped splats stay padding
        __syncthreads();
        while (!done && cur < w1) {
            uint32_t key = isY ? (uint32_t)(cur_u * gx + v) : (uint32_t)(v * gx + cur_u);
            buf[cur - w0] = make_int2((int)(key + 1), idx);
            cur++;
            v++;
            if (v >= v_end) advance();
        }
        __syncthreads();
        for (int k = tid; k < (int)(w1 - w0); k += TPB) {
            int2 kv = buf[k];
            kout[w0 + k] = kv.x;
            vout[w0 + k] = kv.y;
        }
        __syncthreads();
    }
}

LG_API int lg_duplicate_with_keys(const float* ndc, const float* inv_cov, const float* opacity, const int32_t* prefix,
                                  const void* sorted_id, int sorted_id_is_int64, int V, int N, int H, int W, int TH, int TW,
                                  long long table_len, int32_t* keys, int32_t* values, void* stream)
{
    if (N <= 0) return 0;
    int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    dim3 grid(lg_cdiv(N, TPB), V);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_DUP(A_, B_)                                                                                                              \
    do {                                                                                                                                \
        if (sorted_id_is_int64)                                                                                                         \
            hipLaunchKernelGGL((duplicate_with_keys_kernel<A_, B_, int64_t>), grid, dim3(TPB), 0, s, ndc, inv_cov, opacity, prefix,     \
                               (const int64_t*)sorted_id, N, H, W, gx, gy, table_len, keys, values);                                    \
        else                                                                                                                            \
            hipLaunchKernelGGL((duplicate_with_keys_kernel<A_, B_, int32_t>), grid, dim3(TPB), 0, s, ndc, inv_cov, opacity, prefix,     \
                               (const int32_t*)sorted_id, N, H, W, gx, gy, table_len, keys, values);                                    \
    } while (0)
    if (TH == 8 && TW == 16) LAUNCH_DUP(8, 16);
    else if (TH == 16 && TW == 16) LAUNCH_DUP(16, 16);
    else if (TH == 12 && TW == 16) LAUNCH_DUP(12, 16);
    else if (TH == 8 && TW == 8) LAUNCH_DUP(8, 8);
    else return (int)hipErrorInvalidValue;
#undef LAUNCH_DUP
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// Stable LSD radix sort of (u32 key, u32 value) pairs on key bits [begin_bit, end_bit), 8 bits per pass.
// Replaces cub::DeviceRadixSort::SortPairs (GR/binning.cu:204-221) and torch.sort of the depth keys
// (litegs/utils/wrapper.py:739).  Per pass: (1) per-tile digit histogram (tile = 4096 keys per 256-thread
// workgroup, LDS atomics), (2) one workgroup per digit turns its histogram row into global offsets,
// (3) scatter: 16 rounds of 256 keys; within a round a lane's rank among equal digits comes from
// 8 wave ballots ("match-any"), wave totals are combined through LDS, so equal keys keep their input
// order (stability is load-bearing: it preserves depth order inside a tile).
// Digit totals for all passes are counted once up front (they are permutation invariant).
// ---------------------------------------------------------------------------------------------
#define RADIX_BITS 8
#define RADIX (1 << RADIX_BITS)
#define SORT_ITEMS 16
#define SORT_TILE (TPB * SORT_ITEMS)
#define SORT_MAX_PASSES 4

__global__ void __launch_bounds__(TPB) radix_totals_kernel(const uint32_t* __restrict__ keys, long long n, int begin_bit, int passes,
                                                           uint32_t last_mask, int* __restrict__ totals /*[passes][RADIX]*/)
{
    __shared__ int h[SORT_MAX_PASSES * RADIX];
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
        if (h[k]) atomicAdd(&totals[k], h[k]);
}

__global__ void __launch_bounds__(TPB) radix_hist_kernel(const uint32_t* __restrict__ keys, long long n, int shift, uint32_t mask,
                                                         int ntiles, int* __restrict__ hist /*[RADIX][ntiles]*/)
{
    __shared__ int h[RADIX];
    h[threadIdx.x] = 0;                              // TPB == RADIX
    __syncthreads();
    long long base = (long long)blockIdx.x * SORT_TILE;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        long long i = base + j * TPB + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & mask], 1);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// one workgroup per digit: base = sum of totals of smaller digits; exclusive scan of this digit's row
__global__ void __launch_bounds__(TPB) radix_scan_kernel(int* __restrict__ hist, const int* __restrict__ totals, int ntiles)
{
    __shared__ int wsum[TPB / 64];
    __shared__ int carry_s;
    const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // base
    int t = (tid < d) ? totals[tid] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
    if (lane == 0) wsum[wave] = t;
    __syncthreads();
    int carry = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    int* row = hist + (size_t)d * ntiles;
    for (int start = 0; start < ntiles; start += TPB) {
        int k = start + tid;
        int v = (k < ntiles) ? row[k] : 0;
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
        if (k < ntiles) row[k] = carry + wbase + incl - v;
        if (tid == TPB - 1) carry_s = carry + wbase + incl;
        __syncthreads();
        carry = carry_s;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(TPB) radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                            const int* __restrict__ offsets /*[RADIX][ntiles]*/, long long n,
                                                            int shift, uint32_t mask, int ntiles)
{
    __shared__ int digit_base[RADIX];
    __shared__ int wave_cnt[TPB / 64][RADIX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (long long)blockIdx.x * SORT_TILE;
    digit_base[tid] = offsets[(size_t)tid * ntiles + blockIdx.x];
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
        long long i = base + j * TPB + tid;
        bool ok = i < n;
        key[j] = ok ? keys_in[i] : 0u;
        val[j] = ok ? vals_in[i] : 0u;
    }
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; j++) {
#pragma unroll
        for (int w = 0; w < TPB / 64; w++) wave_cnt[w][tid] = 0;
        __syncthreads();
        const bool ok = (base + j * TPB + tid) < n;
        const uint32_t d = (key[j] >> shift) & mask;
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int bit = 0; bit < RADIX_BITS; bit++) {
            const bool set = (d >> bit) & 1u;
            unsigned long long bal = __ballot(set);
            peers &= set ? bal : ~bal;
        }
        const int rank = __popcll(peers & lt_mask);
        if (ok && rank == 0) wave_cnt[wave][d] = __popcll(peers);
        __syncthreads();
        if (ok) {
            int off = digit_base[d] + rank;
            for (int w = 0; w < wave; w++) off += wave_cnt[w][d];
            keys_out[off] = key[j];
            vals_out[off] = val[j];
        }
        __syncthreads();
        digit_base[tid] += wave_cnt[0][tid] + wave_cnt[1][tid] + wave_cnt[2][tid] + wave_cnt[3][tid];
        __syncthreads();
    }
}

LG_API long long lg_radix_sort_temp_bytes(long long n)
{
    long long ntiles = (n + SORT_TILE - 1) / SORT_TILE;
    if (ntiles < 1) ntiles = 1;
    return (long long)sizeof(int) * (SORT_MAX_PASSES * RADIX + (long long)RADIX * ntiles);
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
    int passes = lg_radix_sort_num_passes(begin_bit, end_bit);
    if (n <= 0 || passes == 0) return 0;
    if (passes > SORT_MAX_PASSES || n > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (temp_bytes < lg_radix_sort_temp_bytes(n)) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    int ntiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
    int* totals = (int*)temp;
    int* hist = totals + SORT_MAX_PASSES * RADIX;
    int last_bits = (end_bit - begin_bit) - (passes - 1) * RADIX_BITS;
    uint32_t last_mask = (1u << last_bits) - 1u;
    hipError_t err = hipMemsetAsync(totals, 0, sizeof(int) * SORT_MAX_PASSES * RADIX, s);
    if (err != hipSuccess) return (int)err;
    int tot_grid = ntiles < 512 ? ntiles : 512;
    hipLaunchKernelGGL(radix_totals_kernel, dim3(tot_grid), dim3(TPB), 0, s, keys_a, n, begin_bit, passes, last_mask, totals);
    uint32_t *kin = keys_a, *vin = vals_a, *kout = keys_b, *vout = vals_b;
    for (int p = 0; p < passes; p++) {
        int shift = begin_bit + p * RADIX_BITS;
        uint32_t mask = (p == passes - 1) ? last_mask : (uint32_t)(RADIX - 1);
        hipLaunchKernelGGL(radix_hist_kernel, dim3(ntiles), dim3(TPB), 0, s, kin, n, shift, mask, ntiles, hist);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(RADIX), dim3(TPB), 0, s, hist, totals + p * RADIX, ntiles);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(ntiles), dim3(TPB), 0, s, kin, vin, kout, vout, hist, n, shift, mask, ntiles);
        uint32_t* t;
        t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    LG_RETURN_LAST();
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

LG_API int lg_depth_sort_keys(const float* depth, long long n, uint32_t* keys, uint32_t* vals, void* stream)
{
    if (n <= 0) return 0;
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

LG_API long long lg_scan_temp_bytes(long long n)
{
    long long ntiles = (n + SORT_TILE - 1) / SORT_TILE;
    return sizeof(int) * (ntiles < 1 ? 1 : ntiles);
}

LG_API int lg_gather_inclusive_scan(const int32_t* src, const void* idx, int idx_is_int64, long long n, int32_t* out,
                                    void* temp, long long temp_bytes, void* stream)
{
    if (n <= 0) return 0;
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
__global__ void __launch_bounds__(TPB) tile_range_kernel(const int32_t* __restrict__ sorted_keys, long long L, int max_tile,
                                                         int32_t* __restrict__ out)
{
    long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    const int b = blockIdx.y;
    const int32_t* k = sorted_keys + (size_t)b * L;
    int32_t* o = out + (size_t)b * (max_tile + 2);
    if (i == 0) o[k[0]] = 0;
    if (i == L - 1) o[max_tile + 1] = (int32_t)L;
    if (i < L - 1) {
        int cur = k[i], nxt = k[i + 1];
        if (cur != nxt) {
            if (cur + 1 < nxt) o[cur + 1] = (int32_t)(i + 1);
            o[nxt] = (int32_t)(i + 1);
        }
    }
}

LG_API int lg_tile_range(const int32_t* sorted_keys, int V, long long L, int max_tile, int32_t* out, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    hipError_t err = hipMemsetAsync(out, 0xFF, sizeof(int32_t) * (size_t)V * (max_tile + 2), s);
    if (err != hipSuccess) return (int)err;
    if (L <= 0) return 0;
    hipLaunchKernelGGL(tile_range_kernel, dim3(lg_cdiv(L, TPB), V), dim3(TPB), 0, s, sorted_keys, L, max_tile, out);
    LG_RETURN_LAST();
}

LG_API int lg_memset_async(void* ptr, int value, long long bytes, void* stream)
{
    if (bytes <= 0) return 0;
    return (int)hipMemsetAsync(ptr, value, (size_t)bytes, (hipStream_t)stream);
}
