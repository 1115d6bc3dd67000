// Tile binning: exact ellipse/tile intersection count, key/value emission in depth order, the
// hand-written stable LSD radix sort ("the tile radix sort"), prefix scan and tile range table
// (SURVEY.md 8a rows a8-a11).  Integer/index work: outputs are BIT-EXACT against the CPU oracle,
// which is why this file is compiled with -ffp-contract=off and uses lg_logf (fixed polynomial).
//
// HBM traffic per tile instance (I of them): 8 B written by duplicate_with_keys, then per radix pass
// 4 B (histogram read) + 8 B (scatter read) + 8 B (scatter write); 14 tile-id bits at 1080p = 2 passes
// of 8 bits.  No MFMA: this is byte shuffling.
#include "lg_common.h"
#include "lg_tilewalk.h"

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
//  * small splats (<= DUP_SMALL tiles): the owning thread runs the serial AccuTile walk into a compacted LDS
//    buffer; the workgroup then streams the buffer out (entry -> owner by binary search over 256 offsets), so
//    global stores are coalesced instead of 64 scattered 4-byte stores per wave instruction;
//  * big splats (near-camera Gaussians can touch thousands of tiles): the owning WAVE emits them cooperatively:
//    one lane per tile slice computes that slice's [min_tile_v, max_tile_v) independently (the serial walk's
//    carried intersections are pure functions of the slice index, see slice_bounds), a wave scan turns the
//    slice counts into offsets, and the 64 lanes then write the splat's contiguous output range in
//    256-byte coalesced stores.  Without this, one thread serialises a 16 200-tile splat and the launch
//    waits for it (measured: 2.3 ms of a 4.9 ms training step at 3 M Gaussians).
#define DUP_SMALL 32
#define DUP_LDS_ENTRIES (TPB * DUP_SMALL)
#define DUP_MAX_SLICES 256

struct WalkFrame {          // per-splat constants of the (u,v) walk, derivable from SplatExtent
    bool isY;
    float BLOCK_U, BLOCK_V, bmin_u, bmax_u, bmin_v, bmax_v, argmin_v, argmax_v;
    int rect_min_u, rect_max_u, rect_min_v, rect_max_v;
};

template <int TH, int TW>
__device__ __forceinline__ WalkFrame walk_frame(const SplatExtent& e)
{
    WalkFrame f;
    const int ys = e.rmaxy - e.rminy, xs = e.rmaxx - e.rminx;
    f.isY = ys < xs;
    f.BLOCK_U = f.isY ? (float)TH : (float)TW;
    f.BLOCK_V = f.isY ? (float)TW : (float)TH;
    f.rect_min_u = f.isY ? e.rminy : e.rminx; f.rect_max_u = f.isY ? e.rmaxy : e.rmaxx;
    f.rect_min_v = f.isY ? e.rminx : e.rminy; f.rect_max_v = f.isY ? e.rmaxx : e.rmaxy;
    f.bmin_u = f.isY ? e.bbox_min_y : e.bbox_min_x; f.bmin_v = f.isY ? e.bbox_min_x : e.bbox_min_y;
    f.bmax_u = f.isY ? e.bbox_max_y : e.bbox_max_x; f.bmax_v = f.isY ? e.bbox_max_x : e.bbox_max_y;
    f.argmin_v = f.isY ? e.argmin_x : e.argmin_y;
    f.argmax_v = f.isY ? e.argmax_x : e.argmax_y;
    return f;
}

// The serial walk carries intersect_max_line from slice to slice: it is cut(max_line_i) while max_line_i <= bmax_u and
// then sticks at the last such cut (or at the sentinel if there is none).  Lines are (rect_min_u + i) * BLOCK_U exactly
// (small integers), so "intersection at the upper line of slice i" is a pure function of i and K, where K = number of
// slices whose upper line is <= bmax_u (a prefix).  Same for the lower line (= upper line of slice i-1, or the special
// first-slice rule).  => each slice can be evaluated independently and still match the serial walk bit for bit.
__device__ __forceinline__ void upper_cut(const SplatExtent& e, const WalkFrame& f, int i, int K, float& lo, float& hi)
{
    // intersect_max_line after processing slice i (i >= 0); i == -1 -> sentinel
    int j = (i < K) ? i : (K - 1);
    if (j < 0) { lo = f.bmax_v; hi = f.bmin_v; return; }
    ellipse_cut(e, f.isY, (float)(f.rect_min_u + j + 1) * f.BLOCK_U, lo, hi);
}

__device__ __forceinline__ void slice_bounds(const SplatExtent& e, const WalkFrame& f, int i, int K, int& min_tile_v, int& max_tile_v)
{
    const float min_line = (float)(f.rect_min_u + i) * f.BLOCK_U;
    const float max_line = min_line + f.BLOCK_U;
    float imin_lo, imin_hi, imax_lo, imax_hi;
    if (i == 0) {
        if (f.bmin_u <= min_line) ellipse_cut(e, f.isY, (float)f.rect_min_u * f.BLOCK_U, imin_lo, imin_hi);
        else { imin_lo = f.bmax_v; imin_hi = f.bmin_v; }
    } else {
        upper_cut(e, f, i - 1, K, imin_lo, imin_hi);
    }
    upper_cut(e, f, i, K, imax_lo, imax_hi);
    float ellipse_min, ellipse_max;
    if (min_line <= f.argmin_v && f.argmin_v < max_line) ellipse_min = f.bmin_v;
    else ellipse_min = fminf(imin_lo, imax_lo);
    if (min_line <= f.argmax_v && f.argmax_v < max_line) ellipse_max = f.bmax_v;
    else ellipse_max = fmaxf(imin_hi, imax_hi);
    min_tile_v = max(f.rect_min_v, min(f.rect_max_v, lg_f2i(ellipse_min / f.BLOCK_V)));
    max_tile_v = min(f.rect_max_v, max(f.rect_min_v, lg_f2i(ellipse_max / f.BLOCK_V + 1)));
}

__device__ __forceinline__ float bcast_f(float v, int src) { return __shfl(v, src); }
__device__ __forceinline__ int bcast_i(int v, int src) { return __shfl(v, src); }

template <int TH, int TW, typename IdxT>
__global__ void __launch_bounds__(TPB) duplicate_with_keys_kernel(const float* __restrict__ ndc, const float* __restrict__ inv_cov,
                                                                  const float* __restrict__ opacity, const int32_t* __restrict__ prefix,
                                                                  const IdxT* __restrict__ sorted_id, int N, int H, int W, int gx, int gy,
                                                                  long long table_len, int32_t* __restrict__ keys, int32_t* __restrict__ values)
{
    __shared__ int2 buf[DUP_LDS_ENTRIES];                 // 64 KiB: compacted (key, idx) of the small splats
    __shared__ int t_loff[TPB + 1];                       // per-thread start in buf
    __shared__ int t_goff[TPB];                           // per-thread start in the table
    __shared__ int w_minv[TPB / 64][DUP_MAX_SLICES];      // per-wave slice scratch for the cooperative path
    __shared__ int w_off[TPB / 64][DUP_MAX_SLICES + 1];
    __shared__ int wsum[TPB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = blockIdx.x * TPB + tid;
    const int b = blockIdx.y;
    const int32_t* pf = prefix + (size_t)b * N;
    int32_t* kout = keys + (size_t)b * table_len;
    int32_t* vout = values + (size_t)b * table_len;

    long long off = 0;
    int cnt = 0, idx = 0;
    bool live = false;
    SplatExtent e;
    if (j < N) {
        off = (j == 0) ? 0 : pf[j - 1];
        long long c = pf[j] - off;
        if (c > 0 && off + c <= table_len) {
            idx = (int)sorted_id[(size_t)b * N + j];
            float nx = ndc[((size_t)b * 4) * N + idx], ny = ndc[((size_t)b * 4 + 1) * N + idx];
            float a = inv_cov[((size_t)b * 4) * N + idx], bb = inv_cov[((size_t)b * 4 + 1) * N + idx], cc = inv_cov[((size_t)b * 4 + 3) * N + idx];
            splat_extent<TH, TW>(nx, ny, a, bb, cc, opacity[idx], H, W, gx, gy, e);
            if ((e.rmaxy - e.rminy) * (e.rmaxx - e.rminx) > 0) { live = true; cnt = (int)c; }
        }
    }
    const bool small = live && cnt <= DUP_SMALL;
    const bool big = live && cnt > DUP_SMALL;

    // ---- small splats: exclusive block scan of their counts -> compacted LDS layout ----
    const int scnt = small ? cnt : 0;
    int incl = scnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int nb = __shfl_up(incl, o);
        if (lane >= o) incl += nb;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; w++) wbase += wsum[w];
    const int loff = wbase + incl - scnt;
    const int total_small = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    t_loff[tid] = loff;
    t_goff[tid] = (int)off;
    if (tid == 0) t_loff[TPB] = total_small;
    if (small) {
        walk_tiles<TH, TW, true>(e, gx, idx, loff, (int32_t*)nullptr, (int32_t*)nullptr, buf);
    }
    __syncthreads();
    for (int p = tid; p < total_small; p += TPB) {
        // owner = last thread t with t_loff[t] <= p  (threads with no small entries have empty ranges)
        int lo = 0, hi = TPB - 1;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (t_loff[mid] <= p) lo = mid; else hi = mid - 1;
        }
        int2 kv = buf[p];
        int g = t_goff[lo] + (p - t_loff[lo]);
        kout[g] = kv.x;
        vout[g] = kv.y;
    }

    // ---- big splats: wave-cooperative, one splat at a time ----
    unsigned long long bigmask = __ballot(big);
    while (bigmask) {
        const int src = __ffsll((long long)bigmask) - 1;
        bigmask &= bigmask - 1;
        SplatExtent s;
        s.a = bcast_f(e.a, src); s.b = bcast_f(e.b, src); s.c = bcast_f(e.c, src); s.disc = bcast_f(e.disc, src); s.t = bcast_f(e.t, src);
        s.px = bcast_f(e.px, src); s.py = bcast_f(e.py, src);
        s.bbox_min_x = bcast_f(e.bbox_min_x, src); s.bbox_min_y = bcast_f(e.bbox_min_y, src);
        s.bbox_max_x = bcast_f(e.bbox_max_x, src); s.bbox_max_y = bcast_f(e.bbox_max_y, src);
        s.argmin_x = bcast_f(e.argmin_x, src); s.argmin_y = bcast_f(e.argmin_y, src);
        s.argmax_x = bcast_f(e.argmax_x, src); s.argmax_y = bcast_f(e.argmax_y, src);
        s.rminx = bcast_i(e.rminx, src); s.rminy = bcast_i(e.rminy, src); s.rmaxx = bcast_i(e.rmaxx, src); s.rmaxy = bcast_i(e.rmaxy, src);
        const int sidx = bcast_i(idx, src);
        const int sgoff = bcast_i((int)off, src);
        const WalkFrame f = walk_frame<TH, TW>(s);
        const int nsl = f.rect_max_u - f.rect_min_u;                   // <= min(grid.x, grid.y) slices
        if (nsl > DUP_MAX_SLICES) {                                    // > 4K-class images: owner lane walks serially
            if (lane == src) walk_tiles<TH, TW, true>(e, gx, idx, off, kout, vout);
            continue;
        }
        // K = number of leading slices whose upper line is <= bmax_u
        int K = 0;
        for (int i0 = 0; i0 < nsl; i0 += 64) {
            int i = i0 + lane;
            bool c = (i < nsl) && ((float)(f.rect_min_u + i) * f.BLOCK_U + f.BLOCK_U <= f.bmax_u);
            K += __popcll(__ballot(c));
        }
        int run = 0;
        for (int i0 = 0; i0 < nsl; i0 += 64) {
            int i = i0 + lane;
            int mn = 0, n = 0;
            if (i < nsl) {
                int mx;
                slice_bounds(s, f, i, K, mn, mx);
                n = mx - mn;
            }
            int inc = n;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                int nb = __shfl_up(inc, o);
                if (lane >= o) inc += nb;
            }
            if (i < nsl && i < DUP_MAX_SLICES) { w_minv[wave][i] = mn; w_off[wave][i] = run + inc - n; }
            run += __shfl(inc, 63);
        }
        const int nsl_c = nsl < DUP_MAX_SLICES ? nsl : DUP_MAX_SLICES;
        if (lane == 0) w_off[wave][nsl_c] = run;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        for (int k = lane; k < run; k += 64) {
            int lo = 0, hi = nsl_c - 1;
            while (lo < hi) {
                int mid = (lo + hi + 1) >> 1;
                if (w_off[wave][mid] <= k) lo = mid; else hi = mid - 1;
            }
            const int u = f.rect_min_u + lo;
            const int v = w_minv[wave][lo] + (k - w_off[wave][lo]);
            const uint32_t key = f.isY ? (uint32_t)(u * gx + v) : (uint32_t)(v * gx + u);
            kout[sgoff + k] = (int32_t)(key + 1);
            vout[sgoff + k] = sidx;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
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
        if (h[k]) atomicAdd(&totals[k], h[k]);
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
            keys_out[g] = k;
            vals_out[g] = lds_v[p];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Single-launch radix pass ("onesweep"): the per-workgroup digit counts are chained through a status table with
// decoupled look-back instead of a histogram launch + a scan launch per pass.  Workgroups take a ticket (so a workgroup's
// logical predecessors have all started), rank their 4096 keys exactly like radix_scatter_kernel, publish their 256 digit
// counts (flag AGG), thread d then walks back over the predecessors' words for digit d until it meets an inclusive prefix
// (flag INC), publishes its own inclusive prefix and the tile is streamed out.  One status word carries flag and value, so
// no ordering between separate flag/value stores is needed.  status[] and ticket[] must be zero on entry.
// ---------------------------------------------------------------------------------------------
#define ST_AGG 0x40000000u
#define ST_INC 0x80000000u
#define ST_VAL 0x3fffffffu

__global__ void __launch_bounds__(TPB) radix_onesweep_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                             uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                             const int* __restrict__ totals /*[RADIX] of this pass*/,
                                                             uint32_t* __restrict__ status /*[ntiles][RADIX], zero*/, int* __restrict__ ticket,
                                                             long long n, const int* __restrict__ n_dev, int shift, uint32_t mask)
{
    __shared__ uint32_t lds_k[SORT_TILE];
    __shared__ uint32_t lds_v[SORT_TILE];
    __shared__ int wave_cnt[2][TPB / 64][RADIX];
    __shared__ int digit_run[RADIX];          // running count per digit, then exclusive local base
    __shared__ int global_base[RADIX];
    __shared__ int wsum[TPB / 64];
    __shared__ int bid_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    n = bounded_n(n, n_dev);
    if (tid == 0) bid_s = atomicAdd(ticket, 1);
    digit_run[tid] = 0;
#pragma unroll
    for (int w = 0; w < TPB / 64; w++) { wave_cnt[0][w][tid] = 0; wave_cnt[1][w][tid] = 0; }
    __syncthreads();
    const int bid = bid_s;
    const long long base = (long long)bid * SORT_TILE;
    if (base >= n) return;
    const int cnt_tile = (int)((n - base) < SORT_TILE ? (n - base) : SORT_TILE);
    // digit base = exclusive scan over digits of the global totals (TPB == RADIX)
    {
        const int total = totals[tid];
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
        global_base[tid] = wb0 + inc - total;
        __syncthreads();
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
    const int dcount = digit_run[tid];
    // publish this workgroup's count of digit `tid`, then look back
    uint32_t* my = status + (size_t)bid * RADIX + tid;
    if (bid == 0) {
        __hip_atomic_store(my, ST_INC | (uint32_t)dcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __hip_atomic_store(my, ST_AGG | (uint32_t)dcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        for (int b = bid - 1; b >= 0; b--) {
            const uint32_t* pw = status + (size_t)b * RADIX + tid;
            uint32_t v;
            while (((v = __hip_atomic_load(pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & (ST_AGG | ST_INC)) == 0u)
                __builtin_amdgcn_s_sleep(1);
            excl += v & ST_VAL;
            if (v & ST_INC) break;
        }
        __hip_atomic_store(my, ST_INC | (excl + (uint32_t)dcount), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        global_base[tid] += (int)excl;
    }
    // exclusive scan of the per-digit tile counts -> local base
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
            keys_out[g] = k;
            vals_out[g] = lds_v[p];
        }
    }
}

LG_API int lg_radix_sort_pairs_bounded(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, long long n, const int* n_dev,
                                       int begin_bit, int end_bit, void* temp, long long temp_bytes, void* stream);

// temp layout: totals[SORT_MAX_PASSES][RADIX] | ticket[SORT_MAX_PASSES] (+pad to 64 ints) | table
// table = per-pass histogram [RADIX][ntiles] (small sorts) or look-back status [SORT_MAX_PASSES][ntiles][RADIX] (onesweep)
#define SORT_HEADER_INTS (SORT_MAX_PASSES * RADIX + 64)
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
    if (passes > SORT_MAX_PASSES || n > 0x3fffffffLL) return (int)hipErrorInvalidValue;   // look-back status words carry 30-bit counts
    if (temp_bytes < lg_radix_sort_temp_bytes(n)) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    int ntiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
    int* totals = (int*)temp;
    int* ticket = totals + SORT_MAX_PASSES * RADIX;
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
            hipLaunchKernelGGL(radix_onesweep_kernel, dim3(ntiles), dim3(TPB), 0, s, kin, vin, kout, vout, totals + p * RADIX,
                               (uint32_t*)table + (size_t)p * RADIX * ntiles, ticket + p, n, n_dev, shift, mask);
        }
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
__global__ void __launch_bounds__(TPB) tile_range_kernel(const int32_t* __restrict__ sorted_keys, long long L, const int* __restrict__ n_dev,
                                                         int max_tile, int32_t* __restrict__ out)
{
    long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    const int b = blockIdx.y;
    const long long stride = L;
    L = bounded_n(L, n_dev);
    if (L <= 0) return;
    const int32_t* k = sorted_keys + (size_t)b * stride;
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

LG_API int lg_tile_range_bounded(const int32_t* sorted_keys, int V, long long L, const int* n_dev, int max_tile, int32_t* out, void* stream);

LG_API int lg_tile_range(const int32_t* sorted_keys, int V, long long L, int max_tile, int32_t* out, void* stream)
{
    return lg_tile_range_bounded(sorted_keys, V, L, nullptr, max_tile, out, stream);
}

// n_dev (nullable): only the first min(L, *n_dev) sorted entries are a valid table (see lg_radix_sort_pairs_bounded)
LG_API int lg_tile_range_bounded(const int32_t* sorted_keys, int V, long long L, const int* n_dev, int max_tile, int32_t* out, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    hipError_t err = hipMemsetAsync(out, 0xFF, sizeof(int32_t) * (size_t)V * (max_tile + 2), s);
    if (err != hipSuccess) return (int)err;
    if (L <= 0) return 0;
    hipLaunchKernelGGL(tile_range_kernel, dim3(lg_cdiv(L, TPB), V), dim3(TPB), 0, s, sorted_keys, L, n_dev, max_tile, out);
    LG_RETURN_LAST();
}

LG_API int lg_memset_async(void* ptr, int value, long long bytes, void* stream)
{
    if (bytes <= 0) return 0;
    return (int)hipMemsetAsync(ptr, value, (size_t)bytes, (hipStream_t)stream);
}
