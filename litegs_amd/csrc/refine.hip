// spatial_refine (litegs/scene/point.py:29-154): every few epochs the trainer re-sorts ALL Gaussians, their gradients and both Adam
// moments along a 63-bit Morton curve so that the 128-Gaussian chunks stay spatially compact (tight chunk AABBs = effective culling).
// The reference does it with torch glue (bit-by-bit interleave loop, torch.sort on int64, one fancy-index gather per tensor: ~2 GB of
// traffic through Python indexing at 3 M Gaussians).  Here:
//   (1) bounding box by a two-level min/max reduction,
//   (2) 3 x 21-bit Morton codes with the reference's exact fp32 arithmetic ((p - min) / max(ext, 1e-12) * (2^21 - 1), truncate, clamp),
//   (3) stable argsort of the 63-bit codes = two rounds of the library's 32-bit LSD radix sort (low word, then high word),
//   (4) one gather kernel that permutes any number of [rows, N] tensors in a single launch per tensor.
// Integer outputs (codes, order) are bit-exact against the reference's torch implementation (tests/golden: morton_*).
#include "lg_common.h"
#include "litegs_hip.h"

#define RF_TPB 256

struct RfBounds { float mn[3], mx[3]; };

__global__ void __launch_bounds__(RF_TPB) rf_minmax_partial_kernel(const float* __restrict__ xyz /*[3,N]*/, long long N, float* __restrict__ partial /*[blocks][6]*/)
{
    __shared__ float red[6][RF_TPB / 64];
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (long long i = (long long)blockIdx.x * RF_TPB + threadIdx.x; i < N; i += (long long)gridDim.x * RF_TPB)
#pragma unroll
        for (int k = 0; k < 3; k++) { const float v = xyz[(size_t)k * N + i]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], off)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off)); }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 3; k++) { red[k][threadIdx.x >> 6] = mn[k]; red[3 + k][threadIdx.x >> 6] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int w = 1; w < RF_TPB / 64; w++) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
        partial[6 * blockIdx.x + threadIdx.x] = v;
    }
}

__global__ void __launch_bounds__(64) rf_minmax_final_kernel(const float* __restrict__ partial, int nblocks, RfBounds* __restrict__ out)
{
    const int k = threadIdx.x;
    if (k >= 6) return;
    float v = partial[k];
    for (int b = 1; b < nblocks; b++) v = k < 3 ? fminf(v, partial[6 * b + k]) : fmaxf(v, partial[6 * b + k]);
    if (k < 3) out->mn[k] = v; else out->mx[k - 3] = v;
}

__device__ __forceinline__ unsigned long long spread21(unsigned long long x)      // 21 bits -> every third bit of 63
{
    x &= 0x1fffffull;
    x = (x | (x << 32)) & 0x001f00000000ffffull;
    x = (x | (x << 16)) & 0x001f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

// code = interleave(X, Y, Z) with X in bits 0,3,6,... (point.py:66-72).  Also splits it into the two sort words and writes iota.
__global__ void __launch_bounds__(RF_TPB) rf_morton_kernel(const float* __restrict__ xyz, long long N, const RfBounds* __restrict__ bnd,
                                                           int64_t* __restrict__ codes /*nullable*/, uint32_t* __restrict__ lo,
                                                           uint32_t* __restrict__ hi, uint32_t* __restrict__ iota)
{
    const long long i = (long long)blockIdx.x * RF_TPB + threadIdx.x;
    if (i >= N) return;
    const float scale = 2097151.0f;                      // 2^21 - 1
    unsigned long long q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float mn = bnd->mn[k];
        const float denom = fmaxf(bnd->mx[k] - mn, 1e-12f);          // clamp_min(1e-12) (point.py:53)
        const float t = ((xyz[(size_t)k * N + i] - mn) / denom) * scale;
        long long v = (t == t) ? (long long)t : 0;                    // .long(): truncation
        v = v < 0 ? 0 : (v > 2097151 ? 2097151 : v);
        q[k] = (unsigned long long)v;
    }
    const unsigned long long c = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
    if (codes) codes[i] = (int64_t)c;
    lo[i] = (uint32_t)c;
    hi[i] = (uint32_t)(c >> 32);
    iota[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(RF_TPB) rf_gather_u32_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, long long N,
                                                               uint32_t* __restrict__ dst)
{
    const long long i = (long long)blockIdx.x * RF_TPB + threadIdx.x;
    if (i < N) dst[i] = src[idx[i]];
}

__global__ void __launch_bounds__(RF_TPB) rf_copy_order_kernel(const uint32_t* __restrict__ src, long long N, int32_t* __restrict__ dst)
{
    const long long i = (long long)blockIdx.x * RF_TPB + threadIdx.x;
    if (i < N) dst[i] = (int32_t)src[i];
}

struct RfLayout { size_t bnd, partial, lo, hi, ka, va, kb, vb, sort_temp, total; long long sort_bytes; };

static RfLayout rf_layout(long long N)
{
    RfLayout f;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
    f.sort_bytes = lg_radix_sort_temp_bytes(N);
    f.bnd = take(sizeof(RfBounds));
    f.partial = take(6 * 4 * 1024);
    f.lo = take(4 * (size_t)N); f.hi = take(4 * (size_t)N);
    f.ka = take(4 * (size_t)N); f.va = take(4 * (size_t)N); f.kb = take(4 * (size_t)N); f.vb = take(4 * (size_t)N);
    f.sort_temp = take((size_t)f.sort_bytes);
    f.total = o;
    return f;
}

LG_API long long lg_morton_order_temp_bytes(long long n) { return n <= 0 ? 0 : (long long)rf_layout(n).total; }

LG_API int lg_morton_order(const float* xyz /*[3,n]*/, long long n, int64_t* codes /*nullable [n]*/, int32_t* order /*[n]*/,
                           void* temp, long long temp_bytes, void* stream)
{
    if (n <= 0) return 0;
    if (n >= (1ll << 31)) return (int)hipErrorInvalidValue;
    const RfLayout f = rf_layout(n);
    if (temp == nullptr || temp_bytes < (long long)f.total) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)temp;
    RfBounds* bnd = (RfBounds*)(w + f.bnd);
    float* partial = (float*)(w + f.partial);
    uint32_t *lo = (uint32_t*)(w + f.lo), *hi = (uint32_t*)(w + f.hi);
    uint32_t *ka = (uint32_t*)(w + f.ka), *va = (uint32_t*)(w + f.va), *kb = (uint32_t*)(w + f.kb), *vb = (uint32_t*)(w + f.vb);
    void* sort_temp = w + f.sort_temp;
    const int blocks = lg_cdiv(n, RF_TPB);
    const int rblocks = blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(rf_minmax_partial_kernel, dim3(rblocks), dim3(RF_TPB), 0, s, xyz, n, partial);
    hipLaunchKernelGGL(rf_minmax_final_kernel, dim3(1), dim3(64), 0, s, partial, rblocks, bnd);
    hipLaunchKernelGGL(rf_morton_kernel, dim3(blocks), dim3(RF_TPB), 0, s, xyz, n, bnd, codes, lo, hi, va);
    // round 1: by the low word (lo is consumed as keys_a)
    int rc = lg_radix_sort_pairs(lo, va, kb, vb, n, 0, 32, sort_temp, f.sort_bytes, stream);
    if (rc) return rc;
    const uint32_t* perm1 = (lg_radix_sort_num_passes(0, 32) % 2 == 1) ? vb : va;
    uint32_t* perm1_other = (perm1 == vb) ? va : vb;
    // round 2: by the high word (31 bits), carried along in round-1 order -- LSD: stable, so ties keep the low-word order
    hipLaunchKernelGGL(rf_gather_u32_kernel, dim3(blocks), dim3(RF_TPB), 0, s, hi, perm1, n, ka);
    rc = lg_radix_sort_pairs(ka, (uint32_t*)perm1, kb, perm1_other, n, 0, 31, sort_temp, f.sort_bytes, stream);
    if (rc) return rc;
    const uint32_t* perm2 = (lg_radix_sort_num_passes(0, 31) % 2 == 1) ? perm1_other : perm1;
    hipLaunchKernelGGL(rf_copy_order_kernel, dim3(blocks), dim3(RF_TPB), 0, s, perm2, n, order);
    LG_RETURN_LAST();
}

// dst[r, i] = src[r, order[i]] for r < rows: the gather of spatial_refine / prune (point.py:103-140, densify.py:75-101).
// One thread per destination column and RF_ROWS rows: the index is loaded once per RF_ROWS gathers.
#define RF_ROWS 8
__global__ void __launch_bounds__(RF_TPB) rf_permute_kernel(const float* __restrict__ src, const int32_t* __restrict__ order, long long rows,
                                                            long long n_src, long long n_dst, float* __restrict__ dst)
{
    const long long i = (long long)blockIdx.x * RF_TPB + threadIdx.x;
    if (i >= n_dst) return;
    const long long j = order[i];
    const long long r0 = (long long)blockIdx.y * RF_ROWS;
    float v[RF_ROWS];
#pragma unroll
    for (int r = 0; r < RF_ROWS; r++)
        if (r0 + r < rows) v[r] = src[(size_t)(r0 + r) * n_src + j];
#pragma unroll
    for (int r = 0; r < RF_ROWS; r++)
        if (r0 + r < rows) dst[(size_t)(r0 + r) * n_dst + i] = v[r];
}

LG_API int lg_permute_columns(const float* src /*[rows,n_src]*/, const int32_t* order /*[n_dst], values < n_src*/, long long rows,
                              long long n_src, long long n_dst, float* dst /*[rows,n_dst]*/, void* stream)
{
    if (rows <= 0 || n_dst <= 0) return 0;
    LG_REQUIRE(src, dst);
    if (src == (const float*)dst) return (int)hipErrorInvalidValue;         // out of place only
    dim3 grid(lg_cdiv(n_dst, RF_TPB), lg_cdiv(rows, RF_ROWS));
    if (grid.y > 65535) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(rf_permute_kernel, grid, dim3(RF_TPB), 0, (hipStream_t)stream, src, order, rows, n_src, n_dst, dst);
    LG_RETURN_LAST();
}
