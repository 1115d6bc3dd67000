// Body of the per-tile depth sort (tilesort.hip), written so that the SAME text compiles as device code and as a sequential host
// emulation (tests/host/bitonic_check.cpp defines LG_TILESORT_HOST): every barrier-separated phase is a TS_PHASE loop over the
// cooperating threads -- one trip per thread on the device, all threads one after the other on the host (valid because the pairs of
// one network step are disjoint) -- and no per-thread state crosses a TS_SYNC.
//
// A tile's list is sorted by the 64-bit key (depth key << 32 | splat id): ascending view depth, ties by ascending id.  That is the
// order a STABLE depth sort of the splats followed by a STABLE tile sort of the instances produces (the reference's pipeline,
// litegs/utils/wrapper.py:739-745 + GR/binning.cu:205-220), because instances are emitted in ascending id order here.
#pragma once
#include <stdint.h>
#include "lg_bitonic.h"

#ifdef LG_TILESORT_HOST
#define TS_FN static inline
#define TS_PHASE(tid, nt) for (int tid = 0; tid < (nt); tid++)
#define TS_SYNC(wave) ((void)0)
#define TS_TID_ARG
#define TS_TID_PASS
#else
#define TS_FN __device__ __forceinline__
// tid0 = index of this thread in the cooperating group (lane in the wave regime, threadIdx.x in the workgroup regimes)
#define TS_PHASE(tid, nt) for (int tid = tid0; tid < (nt); tid += (nt))
#define TS_SYNC(wave) ts_sync(wave)
#define TS_TID_ARG , int tid0
#define TS_TID_PASS , tid0
__device__ __forceinline__ void ts_sync(bool wave)
{
    if (wave) {       // one wave: its LDS operations execute in order; the wavefront-scope fences only pin the compiler's ordering
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}
#endif

#define TS_SMALL 512          // regime S: one wave per tile, list in a 4 KB LDS slice
#define TS_CHUNK 2048         // regime M: one workgroup per tile, list in the 16 KB LDS buffer; regime L: chunk size
#define TS_LOG_CHUNK 11

// monotone float -> uint32 map of the depth sort (binning.hip depth_keys_kernel)
TS_FN uint32_t ts_depth_key(uint32_t float_bits)
{
    return (float_bits & 0x80000000u) ? ~float_bits : (float_bits | 0x80000000u);
}

TS_FN void ts_cex(uint64_t* sk, int i, int p)
{
    const uint64_t a = sk[i], b = sk[p];
    if (a > b) { sk[i] = b; sk[p] = a; }
}

// sorts sk[0, m) (m >= 2) with nt cooperating threads.  first_lj < 0: the whole network; otherwise only the half-cleaner steps with
// strides 2^first_lj ... 1 (the tail of a merge stage whose large strides were done elsewhere).
TS_FN void ts_sort_local(uint64_t* sk, int m, int first_lj, int nt, bool wave TS_TID_ARG)
{
    const int P = lg_pow2_ceil(m), lp = lg_log2_exact(P), half = P >> 1;
    if (first_lj < 0) {
        for (int lk = 1; lk <= lp; lk++) {
            TS_PHASE(tid, nt) {
                for (int t = tid; t < half; t += nt) {
                    int i, p;
                    lg_bitonic_flip_pair(t, lk, i, p);
                    if (p < m) ts_cex(sk, i, p);
                }
            }
            TS_SYNC(wave);
            for (int lj = lk - 2; lj >= 0; lj--) {
                TS_PHASE(tid, nt) {
                    for (int t = tid; t < half; t += nt) {
                        int i, p;
                        lg_bitonic_step_pair(t, lj, i, p);
                        if (p < m) ts_cex(sk, i, p);
                    }
                }
                TS_SYNC(wave);
            }
        }
    } else {
        for (int lj = first_lj; lj >= 0; lj--) {
            TS_PHASE(tid, nt) {
                for (int t = tid; t < half; t += nt) {
                    int i, p;
                    lg_bitonic_step_pair(t, lj, i, p);
                    if (p < m) ts_cex(sk, i, p);
                }
            }
            TS_SYNC(wave);
        }
    }
}

// one tile list v[0, n) (splat ids, ascending on entry), n >= 2; depth_bits(id) = raw float bits of the splat's view depth.
// sk: LDS buffer of at least min(n, TS_CHUNK) (regimes M, L) or TS_SMALL (regime S) 64-bit words, private to the cooperating group.
// dk: n words of global scratch aligned with v (regime L only).
template <class DepthBits>
TS_FN void ts_sort_tile(int* v, int n, uint64_t* sk, uint32_t* dk, DepthBits depth_bits, int nt, bool wave TS_TID_ARG)
{
    if (n <= TS_CHUNK) {                          // regimes S (nt = 64, wave) and M (nt = 256)
        TS_PHASE(tid, nt) {
            for (int e = tid; e < n; e += nt) {
                const int id = v[e];
                sk[e] = ((uint64_t)ts_depth_key(depth_bits(id)) << 32) | (uint32_t)id;
            }
        }
        TS_SYNC(wave);
        ts_sort_local(sk, n, -1, nt, wave TS_TID_PASS);
        TS_PHASE(tid, nt) {
            for (int e = tid; e < n; e += nt) v[e] = (int)(uint32_t)sk[e];
        }
        TS_SYNC(wave);
        return;
    }
    // regime L: depth keys to global scratch, every chunk sorted in LDS, then the merge stages k = 2*CHUNK ... P: strides >= CHUNK as
    // steps on global memory, strides < CHUNK as one pass over LDS-resident chunks
    TS_PHASE(tid, nt) {
        for (int e = tid; e < n; e += nt) dk[e] = ts_depth_key(depth_bits(v[e]));
    }
    TS_SYNC(wave);
    for (int c0 = 0; c0 < n; c0 += TS_CHUNK) {
        const int m = (n - c0 < TS_CHUNK) ? n - c0 : TS_CHUNK;
        TS_PHASE(tid, nt) {
            for (int e = tid; e < m; e += nt) sk[e] = ((uint64_t)dk[c0 + e] << 32) | (uint32_t)v[c0 + e];
        }
        TS_SYNC(wave);
        if (m >= 2) ts_sort_local(sk, m, -1, nt, wave TS_TID_PASS);
        TS_PHASE(tid, nt) {
            for (int e = tid; e < m; e += nt) { dk[c0 + e] = (uint32_t)(sk[e] >> 32); v[c0 + e] = (int)(uint32_t)sk[e]; }
        }
        TS_SYNC(wave);
    }
    const int P = lg_pow2_ceil(n), lp = lg_log2_exact(P), half = P >> 1;
    for (int lk = TS_LOG_CHUNK + 1; lk <= lp; lk++) {
        for (int step = 0; ; step++) {            // step 0: flip; then strides 2^(lk-2) ... CHUNK
            const int lj = lk - 1 - step;         // stride of the half-cleaner step (unused for the flip)
            if (step > 0 && lj < TS_LOG_CHUNK) break;
            TS_PHASE(tid, nt) {
                for (int t = tid; t < half; t += nt) {
                    int i, p;
                    if (step == 0) lg_bitonic_flip_pair(t, lk, i, p); else lg_bitonic_step_pair(t, lj, i, p);
                    if (p < n) {
                        const uint32_t ka = dk[i], kb = dk[p];
                        const int ia = v[i], ib = v[p];
                        if (ka > kb || (ka == kb && (uint32_t)ia > (uint32_t)ib)) { dk[i] = kb; dk[p] = ka; v[i] = ib; v[p] = ia; }
                    }
                }
            }
            TS_SYNC(wave);
        }
        for (int c0 = 0; c0 < n; c0 += TS_CHUNK) {
            const int m = (n - c0 < TS_CHUNK) ? n - c0 : TS_CHUNK;
            if (m < 2) continue;
            TS_PHASE(tid, nt) {
                for (int e = tid; e < m; e += nt) sk[e] = ((uint64_t)dk[c0 + e] << 32) | (uint32_t)v[c0 + e];
            }
            TS_SYNC(wave);
            ts_sort_local(sk, m, TS_LOG_CHUNK - 1, nt, wave TS_TID_PASS);
            TS_PHASE(tid, nt) {
                for (int e = tid; e < m; e += nt) { dk[c0 + e] = (uint32_t)(sk[e] >> 32); v[c0 + e] = (int)(uint32_t)sk[e]; }
            }
            TS_SYNC(wave);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Regime R: one wave sorts a list of up to TS_RADIX_MAX entries with a stable LSD radix sort on the 32-bit depth key, 8 bits per pass,
// the elements held in registers (element e = c * 64 + lane, c < TS_RCHUNKS) and LDS used for the 256 digit counters and a 4-byte
// exchange buffer.  The list arrives in ascending id order and every pass is stable, so ties in depth keep ascending ids: the same
// order as the 64-bit bitonic key, with ~6x less LDS traffic (per pass and element: one returning add, one 4-byte store and load for
// the key and for the id; the bitonic network moves 16 B per element in each of its ~36 steps).  Passes whose digit is the same for
// every key of the list (typically the sign/exponent byte) are skipped.
// The rank of an element inside its digit is what a returning LDS add hands back: rounds (chunks) are sequential, and inside one
// instruction the LDS serves same-address lanes in lane order -- the property lg_radix_rank_mode() verifies on the device (binning.hip);
// when it does not hold, the kernel is instantiated with BALLOT ranking, where the order follows from the code alone.
// ---------------------------------------------------------------------------------------------
#define TS_RADIX_MAX 1024
#define TS_RCHUNKS (TS_RADIX_MAX / 64)

#ifdef LG_TILESORT_HOST
#define TS_LOCAL(type, name, n) type name[64][n]
#define TS_L(name, c) name[tid][c]
#else
#define TS_LOCAL(type, name, n) type name[n]
#define TS_L(name, c) name[c]
#endif

// RCH: chunks of 64 the instantiation holds in registers (n <= 64 * RCH); the loops below are unrolled RCH times
// ANY_ORDER: the list arrives in arbitrary order (the tile scatter of the executor's tile mode places instances through atomic cursors):
// the stable passes then leave equal depth keys in arrival order, and a repair phase behind the last pass orders every run of equal
// keys by ascending id -- an odd-even transposition on the ids of neighbours with equal keys, repeated until a sweep swaps nothing
// (runs of equal depth are short: duplicated Gaussians right after a clone).  One neighbour comparison per element when there is no tie.
template <bool BALLOT, int RCH, bool ANY_ORDER, class DepthBits>
TS_FN void ts_radix_sort_tile_n(int* v, int n, uint32_t* exch /*[64 * RCH]*/, int* cnt /*[256]*/, DepthBits depth_bits TS_TID_ARG)
{
    const int C = (n + 63) >> 6;
    TS_LOCAL(uint32_t, key, RCH);
    TS_LOCAL(int, id, RCH);
    TS_LOCAL(int, pos, RCH);
    uint32_t or_all = 0u, and_all = 0xffffffffu;
#ifdef LG_TILESORT_HOST
    uint32_t or_t[64], and_t[64];
    int s_t[64];
#endif
    TS_PHASE(tid, 64) {
        uint32_t o = 0u, a = 0xffffffffu;
        // two batches of independent loads (ids, then their depths), no branches in between: a guarded load per chunk would serialise
        // 2 * C dependent global round trips.  Positions beyond the list re-read its last element (same address: one cache line).
#pragma unroll
        for (int c = 0; c < RCH; c++) {
            const int e = c * 64 + tid;
            TS_L(id, c) = v[e < n ? e : n - 1];
        }
#pragma unroll
        for (int c = 0; c < RCH; c++) TS_L(key, c) = ts_depth_key(depth_bits(TS_L(id, c)));
#pragma unroll
        for (int c = 0; c < RCH; c++)
            if (c * 64 + tid < n) { o |= TS_L(key, c); a &= TS_L(key, c); }
#ifdef LG_TILESORT_HOST
        or_t[tid] = o; and_t[tid] = a;
#else
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { o |= (uint32_t)__shfl_xor((int)o, off); a &= (uint32_t)__shfl_xor((int)a, off); }
        or_all = o; and_all = a;
#endif
    }
#ifdef LG_TILESORT_HOST
    for (int t = 0; t < 64; t++) { or_all |= or_t[t]; and_all &= and_t[t]; }
#endif
    const uint32_t varies = or_all ^ and_all;                // bits that differ somewhere in the list (wave-uniform)
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 8 * pass;
        if (((varies >> shift) & 255u) == 0u) continue;      // every key has the same digit here: the pass would be the identity
        TS_PHASE(tid, 64) {
            cnt[4 * tid] = 0; cnt[4 * tid + 1] = 0; cnt[4 * tid + 2] = 0; cnt[4 * tid + 3] = 0;
        }
        TS_SYNC(true);
        // rank inside the digit (stable): chunks in order (the loop), lanes in order (inside one returning add / the ballot ranking)
#pragma unroll
        for (int c = 0; c < RCH; c++) {
            if (c < C) {
                TS_PHASE(tid, 64) {
                    const bool ok = c * 64 + tid < n;
                    const uint32_t d = ok ? ((TS_L(key, c) >> shift) & 255u) : 0u;
#ifdef LG_TILESORT_HOST
                    if (ok) { TS_L(pos, c) = cnt[d]; cnt[d] += 1; }
#else
                    if constexpr (!BALLOT) {
                        if (ok) TS_L(pos, c) = atomicAdd(&cnt[d], 1);
                    } else {
                        unsigned long long same = __ballot(ok);
#pragma unroll
                        for (int b = 0; b < 8; b++) {
                            const unsigned long long bal = __ballot((d >> b) & 1u);
                            same &= ((d >> b) & 1u) ? bal : ~bal;
                        }
                        if (ok) {
                            const int leader = __ffsll((long long)same) - 1;
                            int base = 0;
                            if (tid == leader) { base = cnt[d]; cnt[d] = base + __popcll(same); }
                            base = __shfl(base, leader);
                            TS_L(pos, c) = base + __popcll(same & ((1ull << tid) - 1ull));
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
#endif
                }
            }
        }
        TS_SYNC(true);
        // exclusive scan of the 256 counters (4 per lane)
#ifdef LG_TILESORT_HOST
        TS_PHASE(tid, 64) { s_t[tid] = cnt[4 * tid] + cnt[4 * tid + 1] + cnt[4 * tid + 2] + cnt[4 * tid + 3]; }
        TS_PHASE(tid, 64) {
            int base = 0;
            for (int t = 0; t < tid; t++) base += s_t[t];
            const int c0 = cnt[4 * tid], c1 = cnt[4 * tid + 1], c2 = cnt[4 * tid + 2];
            cnt[4 * tid] = base; cnt[4 * tid + 1] = base + c0; cnt[4 * tid + 2] = base + c0 + c1; cnt[4 * tid + 3] = base + c0 + c1 + c2;
        }
#else
        TS_PHASE(tid, 64) {
            const int c0 = cnt[4 * tid], c1 = cnt[4 * tid + 1], c2 = cnt[4 * tid + 2], c3 = cnt[4 * tid + 3];
            const int s = c0 + c1 + c2 + c3;
            int incl = s;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int nb = __shfl_up(incl, off);
                if (tid >= off) incl += nb;
            }
            const int base = incl - s;
            cnt[4 * tid] = base; cnt[4 * tid + 1] = base + c0; cnt[4 * tid + 2] = base + c0 + c1; cnt[4 * tid + 3] = base + c0 + c1 + c2;
        }
#endif
        TS_SYNC(true);
        // destination of every element, then keys and ids through the exchange buffer
        TS_PHASE(tid, 64) {
#pragma unroll
            for (int c = 0; c < RCH; c++)
                if (c < C && c * 64 + tid < n) {
                    TS_L(pos, c) += cnt[(TS_L(key, c) >> shift) & 255u];
                    exch[TS_L(pos, c)] = TS_L(key, c);
                }
        }
        TS_SYNC(true);
        TS_PHASE(tid, 64) {
#pragma unroll
            for (int c = 0; c < RCH; c++)
                if (c < C && c * 64 + tid < n) TS_L(key, c) = exch[c * 64 + tid];
        }
        TS_SYNC(true);
        TS_PHASE(tid, 64) {
#pragma unroll
            for (int c = 0; c < RCH; c++)
                if (c < C && c * 64 + tid < n) exch[TS_L(pos, c)] = (uint32_t)TS_L(id, c);
        }
        TS_SYNC(true);
        TS_PHASE(tid, 64) {
#pragma unroll
            for (int c = 0; c < RCH; c++)
                if (c < C && c * 64 + tid < n) TS_L(id, c) = (int)exch[c * 64 + tid];
        }
        TS_SYNC(true);
    }
    if (ANY_ORDER) {
        // keys are final (only ids inside runs of equal keys may still move): element e is tied with e + 1 iff their keys are equal
        TS_LOCAL(bool, eq, RCH);
        TS_PHASE(tid, 64) {
#pragma unroll
            for (int c = 0; c < RCH; c++)
                if (c < C && c * 64 + tid < n) exch[c * 64 + tid] = TS_L(key, c);
        }
        TS_SYNC(true);
        bool any_tie = false;
#ifdef LG_TILESORT_HOST
        bool tie_t[64];
#endif
        TS_PHASE(tid, 64) {
            bool t = false;
#pragma unroll
            for (int c = 0; c < RCH; c++) {
                const int e = c * 64 + tid;
                TS_L(eq, c) = (c < C && e + 1 < n) && exch[e + 1] == TS_L(key, c);
                t |= TS_L(eq, c);
            }
#ifdef LG_TILESORT_HOST
            tie_t[tid] = t;
#else
            any_tie = __any(t);
#endif
        }
#ifdef LG_TILESORT_HOST
        for (int t = 0; t < 64; t++) any_tie |= tie_t[t];
#endif
        if (any_tie) {
            TS_SYNC(true);
            TS_PHASE(tid, 64) {
#pragma unroll
                for (int c = 0; c < RCH; c++)
                    if (c < C && c * 64 + tid < n) exch[c * 64 + tid] = (uint32_t)TS_L(id, c);
            }
            TS_SYNC(true);
            for (int sweep = 0; sweep < 2 * n + 2; sweep++) {           // bounded; ends after the first pair of sweeps without a swap
                bool swapped = false;
                for (int parity = 0; parity < 2; parity++) {
#ifdef LG_TILESORT_HOST
                    bool sw_t[64];
#endif
                    TS_PHASE(tid, 64) {
                        bool sw = false;
#pragma unroll
                        for (int c = 0; c < RCH; c++) {
                            const int e = c * 64 + tid;
                            if (c < C && (e & 1) == parity && TS_L(eq, c)) {
                                const uint32_t a = exch[e], b = exch[e + 1];
                                if (a > b) { exch[e] = b; exch[e + 1] = a; sw = true; }
                            }
                        }
#ifdef LG_TILESORT_HOST
                        sw_t[tid] = sw;
#else
                        swapped |= __any(sw);
#endif
                    }
#ifdef LG_TILESORT_HOST
                    for (int t = 0; t < 64; t++) swapped |= sw_t[t];
#endif
                    TS_SYNC(true);
                }
                if (!swapped) break;
            }
            TS_PHASE(tid, 64) {
#pragma unroll
                for (int c = 0; c < RCH; c++)
                    if (c < C && c * 64 + tid < n) TS_L(id, c) = (int)exch[c * 64 + tid];
            }
        }
    }
    TS_PHASE(tid, 64) {
#pragma unroll
        for (int c = 0; c < RCH; c++)
            if (c < C && c * 64 + tid < n) v[c * 64 + tid] = TS_L(id, c);
    }
}

// picks the smallest instantiation that holds the list
template <bool BALLOT, bool ANY_ORDER = false, class DepthBits>
TS_FN void ts_radix_sort_tile(int* v, int n, uint32_t* exch /*[TS_RADIX_MAX]*/, int* cnt /*[256]*/, DepthBits depth_bits TS_TID_ARG)
{
    if (n <= 256) ts_radix_sort_tile_n<BALLOT, 4, ANY_ORDER>(v, n, exch, cnt, depth_bits TS_TID_PASS);
    else if (n <= 512) ts_radix_sort_tile_n<BALLOT, 8, ANY_ORDER>(v, n, exch, cnt, depth_bits TS_TID_PASS);
    else ts_radix_sort_tile_n<BALLOT, TS_RCHUNKS, ANY_ORDER>(v, n, exch, cnt, depth_bits TS_TID_PASS);
}
