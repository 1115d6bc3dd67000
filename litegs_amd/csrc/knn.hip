// distCUDA2: mean squared distance of every point to its 3 nearest neighbours (scene initialisation, litegs/scene/point.py:8;
// reference: litegs/submodules/simple-knn/simple_knn.cu:186-222 -- Morton order, boxes of consecutive points, box pruning).
// Exact (every box whose distance bound beats the current third-best is scanned), so any exact 3-NN method gives the same numbers.
// Design: (1) bounding box by a two-level min/max reduction, (2) 30-bit Morton keys, (3) the library's own stable radix sort,
// (4) points gathered into Morton order as float4 (xyz + original index) with one AABB per 256-point box, (5) one thread per
// sorted point: seed the bound from its +-3 Morton neighbours, then walk the boxes -- threads of a wave are spatial neighbours,
// so they accept the same boxes and their loads of a box's points are wave-uniform broadcasts.
#include "lg_common.h"
#include "litegs_hip.h"

#define KNN_TPB 256
#define KNN_BOX 256

struct KnnBounds { float mn[3], mx[3]; };

__global__ void __launch_bounds__(KNN_TPB) knn_minmax_partial_kernel(const float* __restrict__ pts, int P, float* __restrict__ partial /*[blocks][6]*/)
{
    __shared__ float red[6][KNN_TPB / 64];
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    for (int i = blockIdx.x * KNN_TPB + threadIdx.x; i < P; i += gridDim.x * KNN_TPB)
#pragma unroll
        for (int k = 0; k < 3; k++) { float v = pts[3 * (size_t)i + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn[k] = fminf(mn[k], __shfl_down(mn[k], off)); mx[k] = fmaxf(mx[k], __shfl_down(mx[k], off)); }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 3; k++) { red[k][threadIdx.x >> 6] = mn[k]; red[3 + k][threadIdx.x >> 6] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int w = 1; w < KNN_TPB / 64; w++) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
        partial[6 * blockIdx.x + threadIdx.x] = v;
    }
}

__global__ void __launch_bounds__(64) knn_minmax_final_kernel(const float* __restrict__ partial, int nblocks, KnnBounds* __restrict__ out)
{
    const int k = threadIdx.x;
    if (k >= 6) return;
    float v = partial[k];
    for (int b = 1; b < nblocks; b++) v = k < 3 ? fminf(v, partial[6 * b + k]) : fmaxf(v, partial[6 * b + k]);
    if (k < 3) out->mn[k] = v; else out->mx[k - 3] = v;
}

__device__ __forceinline__ uint32_t spread10(uint32_t x)      // 10 bits -> every third bit
{
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ void __launch_bounds__(KNN_TPB) knn_morton_kernel(const float* __restrict__ pts, int P, const KnnBounds* __restrict__ bnd,
                                                             uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int i = blockIdx.x * KNN_TPB + threadIdx.x;
    if (i >= P) return;
    uint32_t c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float ext = bnd->mx[k] - bnd->mn[k];
        const float t = ext > 0.0f ? (pts[3 * (size_t)i + k] - bnd->mn[k]) / ext : 0.0f;
        c[k] = (uint32_t)fminf(fmaxf(t * 1023.0f, 0.0f), 1023.0f);
    }
    keys[i] = spread10(c[0]) | (spread10(c[1]) << 1) | (spread10(c[2]) << 2);
    vals[i] = (uint32_t)i;
}

// sorted[j] = (xyz of the j-th point in Morton order, its original index); one AABB per box of KNN_BOX sorted points
__global__ void __launch_bounds__(KNN_BOX) knn_gather_boxes_kernel(const float* __restrict__ pts, const uint32_t* __restrict__ order, int P,
                                                                   float4* __restrict__ sorted, float* __restrict__ boxes /*[nboxes][6]*/)
{
    __shared__ float red[6][KNN_BOX / 64];
    const int j = blockIdx.x * KNN_BOX + threadIdx.x;
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    if (j < P) {
        const uint32_t i = order[j];
        const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
        sorted[j] = make_float4(x, y, z, __uint_as_float(i));
        mn[0] = mx[0] = x; mn[1] = mx[1] = y; mn[2] = mx[2] = z;
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn[k] = fminf(mn[k], __shfl_down(mn[k], off)); mx[k] = fmaxf(mx[k], __shfl_down(mx[k], off)); }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 3; k++) { red[k][threadIdx.x >> 6] = mn[k]; red[3 + k][threadIdx.x >> 6] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int w = 1; w < KNN_BOX / 64; w++) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
        boxes[6 * blockIdx.x + threadIdx.x] = v;
    }
}

__device__ __forceinline__ void knn_update3(float d, float (&best)[3])
{
    if (d < best[2]) {
        if (d < best[1]) {
            best[2] = best[1];
            if (d < best[0]) { best[1] = best[0]; best[0] = d; } else best[1] = d;
        } else best[2] = d;
    }
}

__device__ __forceinline__ float knn_dist2(const float4& a, const float4& b)
{
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
}

__global__ void __launch_bounds__(KNN_TPB) knn_search_kernel(const float4* __restrict__ sorted, const float* __restrict__ boxes, int P, int nboxes,
                                                             float* __restrict__ out)
{
    const int j = blockIdx.x * KNN_TPB + threadIdx.x;
    if (j >= P) return;
    const float4 p = sorted[j];
    float best[3] = { 3.0e38f, 3.0e38f, 3.0e38f };
    // upper bound of the third-nearest distance from the Morton neighbours (they are re-found by the box scan)
    for (int i = max(0, j - 3); i <= min(P - 1, j + 3); i++)
        if (i != j) knn_update3(knn_dist2(p, sorted[i]), best);
    const float reject = best[2];
    best[0] = best[1] = best[2] = 3.0e38f;
    for (int b = 0; b < nboxes; b++) {
        const float* bx = boxes + 6 * (size_t)b;
        float d = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float c = k == 0 ? p.x : (k == 1 ? p.y : p.z);
            const float below = bx[k] - c, above = c - bx[3 + k];
            const float gap = fmaxf(fmaxf(below, above), 0.0f);
            d += gap * gap;
        }
        if (d > reject || d > best[2]) continue;
        const int lo = b * KNN_BOX, hi = min(P, lo + KNN_BOX);
        for (int i = lo; i < hi; i++)
            if (i != j) knn_update3(knn_dist2(p, sorted[i]), best);
    }
    const int n = P - 1 < 3 ? P - 1 : 3;                 // fewer than 4 points: average over the neighbours that exist
    float s = 0.0f;
    for (int k = 0; k < n; k++) s += best[k];
    out[__float_as_uint(p.w)] = n > 0 ? s / (float)n : 0.0f;
}

struct KnnLayout { size_t bnd, partial, ka, va, kb, vb, sort_temp, sorted, boxes, total; long long sort_bytes; int nboxes; };

static KnnLayout knn_layout(int P)
{
    KnnLayout f;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
    f.nboxes = (P + KNN_BOX - 1) / KNN_BOX;
    f.sort_bytes = lg_radix_sort_temp_bytes(P);
    f.bnd = take(sizeof(KnnBounds));
    f.partial = take(6 * 4 * 1024);                      // partial min/max of <= 1024 workgroups
    f.ka = take(4 * (size_t)P); f.va = take(4 * (size_t)P); f.kb = take(4 * (size_t)P); f.vb = take(4 * (size_t)P);
    f.sort_temp = take((size_t)f.sort_bytes);
    f.sorted = take(16 * (size_t)P);
    f.boxes = take(24 * (size_t)f.nboxes);
    f.total = o;
    return f;
}

LG_API long long lg_knn3_temp_bytes(int P) { return P <= 0 ? 0 : (long long)knn_layout(P).total; }

LG_API int lg_knn3_mean_dist2(const float* points /*[P,3]*/, int P, float* mean_dist2 /*[P]*/, void* temp, long long temp_bytes, void* stream)
{
    if (P <= 0) return 0;
    LG_REQUIRE(points, mean_dist2);
    const KnnLayout f = knn_layout(P);
    if (temp == nullptr || temp_bytes < (long long)f.total) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const int nboxes = f.nboxes;
    char* w = (char*)temp;
    KnnBounds* bnd = (KnnBounds*)(w + f.bnd);
    float* partial = (float*)(w + f.partial);
    uint32_t* ka = (uint32_t*)(w + f.ka); uint32_t* va = (uint32_t*)(w + f.va);
    uint32_t* kb = (uint32_t*)(w + f.kb); uint32_t* vb = (uint32_t*)(w + f.vb);
    const long long sort_bytes = f.sort_bytes;
    void* sort_temp = w + f.sort_temp;
    float4* sorted = (float4*)(w + f.sorted);
    float* boxes = (float*)(w + f.boxes);
    int rblocks = lg_cdiv(P, KNN_TPB);
    if (rblocks > 1024) rblocks = 1024;
    hipLaunchKernelGGL(knn_minmax_partial_kernel, dim3(rblocks), dim3(KNN_TPB), 0, s, points, P, partial);
    hipLaunchKernelGGL(knn_minmax_final_kernel, dim3(1), dim3(64), 0, s, partial, rblocks, bnd);
    hipLaunchKernelGGL(knn_morton_kernel, dim3(lg_cdiv(P, KNN_TPB)), dim3(KNN_TPB), 0, s, points, P, bnd, ka, va);
    int rc = lg_radix_sort_pairs(ka, va, kb, vb, P, 0, 30, sort_temp, sort_bytes, stream);
    if (rc) return rc;
    const uint32_t* order = (lg_radix_sort_num_passes(0, 30) % 2 == 1) ? vb : va;
    hipLaunchKernelGGL(knn_gather_boxes_kernel, dim3(nboxes), dim3(KNN_BOX), 0, s, points, order, P, sorted, boxes);
    hipLaunchKernelGGL(knn_search_kernel, dim3(lg_cdiv(P, KNN_TPB)), dim3(KNN_TPB), 0, s, (const float4*)sorted, boxes, P, nboxes, mean_dist2);
    LG_RETURN_LAST();
}
