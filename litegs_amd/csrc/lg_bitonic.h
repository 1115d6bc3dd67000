// Ascending-only bitonic sorting network over n elements (any n >= 2), shared by the device kernels of tilesort.hip and the host
// emulation in tests/host/bitonic_check.cpp (the network is pure index arithmetic: the same header is compiled for both).
//
// P = smallest power of two >= n.  Stage k = 2, 4, ..., P:
//   flip step:        the P/2 pairs (i, p) = (blk*k + o, blk*k + k-1-o), o < k/2     -- each block of k is made of two sorted halves;
//                                                                                      comparing mirrored positions splits it into
//                                                                                      (all small | all large)
//   half cleaners:    for j = k/4, k/8, ..., 1 the P/2 pairs (i, i + j), i = insert a zero bit at bit log2(j) of the pair index
// Every compare-exchange puts the minimum at the LOWER index, so elements at indices >= n behave as +infinity padding that never
// moves: pairs whose upper index p >= n are simply skipped and no padding is stored.  (The textbook form with alternating directions
// would need real padding.)
//
// Steps with j < C operate inside aligned blocks of 2*C elements, which is what lets a long list be processed as global-memory steps
// for the large strides and one pass over LDS-resident chunks for the small ones (tilesort.hip, regime L).
#pragma once

#if defined(__HIPCC__)
#define LG_HD __host__ __device__ __forceinline__
#else
#define LG_HD inline
#endif

LG_HD int lg_pow2_ceil(int n)            // n >= 1
{
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

LG_HD int lg_log2_exact(int p)           // p a power of two
{
    int l = 0;
    while ((1 << l) < p) l++;
    return l;
}

// flip step of stage k = 1 << lk: pair index t in [0, P/2)
LG_HD void lg_bitonic_flip_pair(int t, int lk, int& i, int& p)
{
    const int half = 1 << (lk - 1);
    const int blk = t >> (lk - 1);
    const int o = t & (half - 1);
    i = (blk << lk) + o;
    p = (blk << lk) + ((1 << lk) - 1 - o);
}

// half-cleaner step with stride j = 1 << lj: pair index t in [0, P/2)
LG_HD void lg_bitonic_step_pair(int t, int lj, int& i, int& p)
{
    const int j = 1 << lj;
    i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    p = i + j;
}
