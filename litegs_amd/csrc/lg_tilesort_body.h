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
