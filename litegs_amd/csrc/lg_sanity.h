// Always-on "sanitised" counters (DESIGN.md section 9 "memory access fault").
//
// A word of the tile-instance table becomes an address two kernels later (a key selects a range-table slot, a counter, a cursor; a
// splat id a depth word).  Every consumer range-checks such a word and neutralises it instead of indexing with it -- which keeps a
// run alive, but also turns what would have been a crash into a silently wrong tile boundary.  So the cold branch that neutralises
// a word also COUNTS it: one relaxed device atomic, never taken by a correct table.  The counts are collected by the host at the
// boundaries that synchronise anyway (FrameTrainer.flush(), bench.py, the convergence scripts) and reported; a non-zero count means
// "a table word was garbage" and names the consumer that saw it.
//
// The library is built without relocatable device code, so every translation unit that includes this header has its own copy of the
// counter block; LG_DEFINE_SANITY_COLLECT(tu) defines that unit's host-side reader and fused.hip sums them (lg_sanitised_counts).
#pragma once
#include "lg_common.h"

#define LG_SANITY_SITES 8
#define LG_SITE_EMIT_KEY 0        // key emission rebuilt a key outside 0..tiles (written as 0)
#define LG_SITE_EMIT_COUNT 1      // key emission: a splat walked to another tile count than the prefix sums hold (padded / dropped)
#define LG_SITE_RANGE_KEY 2       // tile_range: boundary key outside 0..tiles skipped
#define LG_SITE_SCATTER_KEY 3     // tile count / tile scatter: key outside 0..tiles dropped
#define LG_SITE_RADIX_INDEX 4     // radix sort: scatter position outside [0, n) (digit totals / look-back words inconsistent): store skipped
#define LG_SITE_TILESORT_ID 5     // per-tile depth sort: splat id outside 0..N-1 clamped
#define LG_SITE_TRUNCATED 6       // (not an error) tables that turned out too short for the prefix sums and were truncated (GR/binning.cu:63)
#define LG_SITE_QUEUE_ENTRY 7     // key emission: a big-splat queue entry that names no slot / part the prefix sums know (stale memory): not followed

static __device__ int lg_sanity_dev[LG_SANITY_SITES];

__device__ __forceinline__ void lg_note_sanitised(int site, int n = 1)
{
    __hip_atomic_fetch_add(&lg_sanity_dev[site], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// out[site] += this unit's counts; reset != 0 clears them.  Blocking (a memcpy on the null stream): call at a synchronisation point.
#define LG_DEFINE_SANITY_COLLECT(tu)                                                                                        \
    int lg_sanity_collect_##tu(int* out, int reset)                                                                         \
    {                                                                                                                       \
        int h[LG_SANITY_SITES];                                                                                             \
        hipError_t e = hipMemcpyFromSymbol(h, HIP_SYMBOL(lg_sanity_dev), sizeof(h));                                        \
        if (e != hipSuccess) return (int)e;                                                                                 \
        int any = 0;                                                                                                        \
        for (int i = 0; i < LG_SANITY_SITES; i++) { out[i] += h[i]; any |= h[i]; }                                          \
        if (reset && any) {                          /* (all zero -- the normal case -- needs no second copy) */             \
            for (int i = 0; i < LG_SANITY_SITES; i++) h[i] = 0;                                                             \
            e = hipMemcpyToSymbol(HIP_SYMBOL(lg_sanity_dev), h, sizeof(h));                                                 \
        }                                                                                                                   \
        return (int)e;                                                                                                      \
    }
