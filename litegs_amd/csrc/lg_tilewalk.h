// Ellipse extent and AccuTile tile walk shared by binning.hip (get_allocate_size / duplicate_with_keys) and fused.hip
// (tile counting fused into the projection kernel).  Exact arithmetic: include only from translation units built with
// -ffp-contract=off, so the results stay bit-identical to the CPU checker.
#pragma once
#include "lg_common.h"

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
// lds_slice_starts (with lds_keys): instead of one key per tile, ONE entry per non-empty slice -- the slice's first key at the slice's
// first position plus a bit in a bitmap over positions; the consumer rebuilds key = first + (position - start) * stride, stride = 1
// when slices run along y (isY: consecutive tiles in x) and gx otherwise.  Removes the per-tile loop from the serial walk.
template <int TH, int TW, bool EMIT, typename LdsKeyT = int32_t>
__device__ __forceinline__ uint32_t walk_tiles(const SplatExtent& e, int gx, int32_t idx, long long off,
                                               int32_t* __restrict__ keys, int32_t* __restrict__ values, LdsKeyT* lds_keys = nullptr,
                                               unsigned int* lds_slice_starts = nullptr, long long lds_limit = 0x7fffffffffffffffLL)
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
        if (EMIT && lds_slice_starts) {
            if (max_tile_v > min_tile_v && off < lds_limit) {     // (lds_limit: end of the caller's share of the LDS buffer)
                const uint32_t key = isY ? (uint32_t)(u * gx + min_tile_v) : (uint32_t)(min_tile_v * gx + u);
                lds_keys[off] = (LdsKeyT)(key + 1);
                atomicOr(lds_slice_starts + (off >> 5), 1u << (off & 31));
                off += max_tile_v - min_tile_v;
            }
        } else if (EMIT) {
            for (int v = min_tile_v; v < max_tile_v; v++) {
                uint32_t key = isY ? (uint32_t)(u * gx + v) : (uint32_t)(v * gx + u);
                if (lds_keys) lds_keys[off] = (LdsKeyT)(key + 1);
                else { keys[off] = (int32_t)(key + 1); values[off] = idx; }
                off++;
            }
        }
        imin_lo = imax_lo; imin_hi = imax_hi;
        min_line = max_line;
    }
    return count;
}

// fminf/fmaxf above drop a NaN operand (v_min_f32 / CUDA min()): a cut taken exactly on the ellipse's extreme line can have a
// slightly negative discriminant (sqrt -> NaN) and the slice then uses the other line's intersection, as in the reference.


// visibility test + exact tile count of one splat (reference: GR/binning.cu:310-373); rect (nullable) receives the tile rectangle
// [x0, x1) x [y0, y1) the walk stays inside
template <int TH, int TW>
__device__ __forceinline__ int lg_tile_count(float nx, float ny, float view_z, float a, float bb, float c, float o, int H, int W, int gx, int gy,
                                             int* rect = nullptr)
{
    float disc = bb * bb - a * c;
    bool vis = !((nx < -1.3f) || (nx > 1.3f) || (ny < -1.3f) || (ny > 1.3f) || (view_z <= 0.2f) || (o < 1.0f / 255));
    vis = vis && (a > 0) && (c > 0) && (disc < 0);
    if (!vis) return 0;
    SplatExtent e;
    splat_extent<TH, TW>(nx, ny, a, bb, c, o, H, W, gx, gy, e);
    if ((e.rmaxy - e.rminy) * (e.rmaxx - e.rminx) <= 0) return 0;
    if (rect) { rect[0] = e.rminx; rect[1] = e.rmaxx; rect[2] = e.rminy; rect[3] = e.rmaxy; }
    return (int)walk_tiles<TH, TW, false>(e, gx, 0, 0, nullptr, nullptr);
}

// ---- the same walk, one slice at a time (binning.hip dup_big_kernel: one lane per slice) ----
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

// ---------------------------------------------------------------------------------------------
// Per-frame depth-bound block ("bounds", no reference counterpart; see fused.hip "depth-bound culling").  Written by the blend forward
// of one visit of a frame (plain stores + atomicMax, no extra launch), read by the next visit of the same frame.  Floats:
//   [0, U)          pyramid levels 1..3: max of level 0 over 2^k x 2^k tile blocks (positive floats, atomicMax on their bit patterns)
//   [U, U + T)      level 0: one view-depth bound per tile (T = gx * gy)
// The first U words must be zero before the forward that fills the block (cleared by the projection kernel's zero duty).
// ---------------------------------------------------------------------------------------------
#define LG_PYR_LEVELS 4
__host__ __device__ static inline int lg_pyr_w(int g, int k) { return (g + (1 << k) - 1) >> k; }
__host__ __device__ static inline long long lg_sched_upper_words(int gx, int gy)
{
    long long o = 0;
    for (int k = 1; k < LG_PYR_LEVELS; k++) o += (long long)lg_pyr_w(gx, k) * lg_pyr_w(gy, k);
    return o;
}
__host__ __device__ static inline long long lg_sched_level_offset(int gx, int gy, int k)      // word offset of pyramid level k
{
    if (k == 0) return lg_sched_upper_words(gx, gy);
    long long o = 0;
    for (int j = 1; j < k; j++) o += (long long)lg_pyr_w(gx, j) * lg_pyr_w(gy, j);
    return o;
}
__host__ __device__ static inline long long lg_sched_words_total(int gx, int gy) { return lg_sched_upper_words(gx, gy) + (long long)gx * gy; }
__host__ __device__ static inline long long lg_sched_clear_words(int gx, int gy) { return lg_sched_upper_words(gx, gy); }

// upper bound of the maximum of the level-0 bounds over the tile rectangle [x0, x1) x [y0, y1): cells of the coarsest stored level
// that keeps the lookup at <= 16 cells; larger rectangles are not culled (+inf)
__device__ __forceinline__ float lg_bound_query(const float* __restrict__ sched, int gx, int gy, int x0, int x1, int y0, int y1)
{
    const int span = max(x1 - x0, y1 - y0);                       // >= 1
    int k = span <= 1 ? 0 : 32 - __clz(span - 1);                 // 2^k >= span: at most 2 cells per axis
    if (k > LG_PYR_LEVELS - 1) k = LG_PYR_LEVELS - 1;
    const int cx0 = x0 >> k, cx1 = (x1 - 1) >> k, cy0 = y0 >> k, cy1 = (y1 - 1) >> k;
    if ((cx1 - cx0 + 1) * (cy1 - cy0 + 1) > 16) return __builtin_inff();
    const int wk = lg_pyr_w(gx, k);
    const float* __restrict__ lv = sched + lg_sched_level_offset(gx, gy, k);
    float m = 0.0f;
    for (int cy = cy0; cy <= cy1; cy++)
        for (int cx = cx0; cx <= cx1; cx++) m = fmaxf(m, lv[cy * wk + cx]);
    return m;
}
