// Per-tile depth sort of the tile-instance table ("fuse the depth sort into the tile sort ... or segmented sort").
//
// The reference orders the splats by view depth with a full-length torch.sort (litegs/utils/wrapper.py:739-745), emits the tile
// instances in that order and relies on the STABILITY of the tile radix sort (GR/binning.cu:205-220) to keep each tile's list in
// depth order.  The native executor used to do the same on the device: depth keys + digit counts, four radix passes over all visible
// splats (most of which emit nothing once depth-bound culling is active) -- ~100 us per frame at 3 M Gaussians, latency bound.
//
// Here the instances are emitted in ascending splat-id order instead (no splat sort at all), the stable tile sort groups them per
// tile, and each tile's list -- a few hundred entries -- is sorted by (depth key, splat id) in LDS.  The resulting table is
// bit-identical to the reference pipeline's (lg_tilesort_body.h), but the work is proportional to the EMITTED instances and it is
// 16 200 independent small problems instead of four dependent passes chained by look-back.
//
//   regime S  2 <= n <= 512   : one wave per tile (four tiles per workgroup), the list in the wave's 4 KB slice of LDS, bitonic network
//                               without workgroup barriers (a wave's LDS operations execute in order)
//   regime M  n <= 2048       : the workgroup's four waves together, 16 KB of LDS
//   regime L  n  > 2048       : depth keys in global scratch; 2048-element chunks sorted in LDS, merge stages with the large strides on
//                               global memory (one workgroup per tile: workgroup-scope visibility) and the small strides per chunk in LDS
// One launch; 16 KB of LDS per workgroup keeps 8 workgroups (32 waves) resident per CU, which is what hides the gather of the depth
// words (one 4-byte read out of each splat's 64-byte record, the line the blend is about to read anyway).
#include "lg_common.h"
#include "lg_binning_internal.h"
#include "lg_tilesort_body.h"

#define TS_REC 16             // floats per packed splat record (raster.hip REC); the view depth is word 12

__global__ void __launch_bounds__(256) tile_depth_sort_kernel(int* __restrict__ vals, const int* __restrict__ tile_start,
                                                              const float* __restrict__ packed, int ntiles, long long L, int N,
                                                              uint32_t* __restrict__ scratch, const int* __restrict__ gate)
{
    if (gate != nullptr && *gate == 0) return;              // fallback launch of the depth-bound culling that is not needed
    __shared__ uint64_t sk[TS_CHUNK];
    const int view = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int t0 = blockIdx.x * 4;                          // this workgroup's tiles: ids t0 + 1 ... t0 + 4
    const int* __restrict__ ts = tile_start + (size_t)view * (ntiles + 2);
    int* __restrict__ v = vals + (size_t)view * L;
    const float* __restrict__ pk = packed + (size_t)view * N * TS_REC;
    auto depth_bits = [pk](int id) -> uint32_t { return __float_as_uint(pk[(size_t)id * TS_REC + 12]); };

    {   // regime S: every wave its own tile
        const int tile = t0 + wave + 1;
        if (tile <= ntiles) {
            const int start = ts[tile], end = ts[tile + 1];
            const int n = (start >= 0 && end > start) ? end - start : 0;
            if (n >= 2 && n <= TS_SMALL) ts_sort_tile(v + start, n, sk + wave * TS_SMALL, (uint32_t*)nullptr, depth_bits, 64, true, lane);
        }
    }
    for (int w = 0; w < 4; w++) {                           // regimes M and L: the whole workgroup, tile after tile
        const int tile = t0 + w + 1;
        if (tile > ntiles) break;
        const int start = ts[tile], end = ts[tile + 1];
        const int n = (start >= 0 && end > start) ? end - start : 0;
        if (n > TS_SMALL) {
            __syncthreads();                                // the LDS slices of regime S / the previous long list are free
            ts_sort_tile(v + start, n, sk, scratch ? scratch + (size_t)view * L + start : (uint32_t*)nullptr, depth_bits, 256, false, (int)threadIdx.x);
        }
    }
}

// vals [V, L] int32 splat ids grouped by tile (ascending id inside a tile), tile_start [V, ntiles + 2] (lg_tile_range), packed [V*N, 16].
// scratch [V, L] uint32 (any content; only touched for lists longer than 2048 -- nullable only if such lists cannot occur).
// gate (nullable device int): nothing runs unless *gate != 0.
int lg_tile_depth_sort_gated(int32_t* vals, const int32_t* tile_start, const float* packed, int V, long long L, int N, int ntiles,
                             uint32_t* scratch, const int* gate, void* stream)
{
    if (ntiles <= 0 || L <= 0 || V <= 0) return 0;
    hipLaunchKernelGGL(tile_depth_sort_kernel, dim3(lg_cdiv(ntiles, 4), V), dim3(256), 0, (hipStream_t)stream, vals, tile_start, packed, ntiles, L, N,
                       scratch, gate);
    LG_RETURN_LAST();
}

LG_API int lg_tile_depth_sort(int32_t* vals, const int32_t* tile_start, const float* packed, int V, long long L, int N, int ntiles,
                              uint32_t* scratch, void* stream)
{
    if (vals == nullptr || tile_start == nullptr || packed == nullptr || scratch == nullptr) return (int)hipErrorInvalidValue;
    return lg_tile_depth_sort_gated(vals, tile_start, packed, V, L, N, ntiles, scratch, nullptr, stream);
}
