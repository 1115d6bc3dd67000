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
//   regime R  2 <= n <= 1024  : one wave per tile (four tiles per workgroup): stable LSD radix sort on the 32-bit depth key with the elements
//                               in registers, 5 KB of LDS per wave (digit counters + a 4-byte exchange buffer), no workgroup barriers;
//                               ties keep the ascending id order the list arrives in (lg_tilesort_body.h)
//   regime M  n <= 2048       : the workgroup's four waves together, bitonic network on (depth key, id) in 16 KB of LDS
//   regime L  n  > 2048       : depth keys in global scratch; 2048-element chunks sorted in LDS, merge stages with the large strides on
//                               global memory (one workgroup per tile: workgroup-scope visibility) and the small strides per chunk in LDS
// One launch; 20 KB of LDS per workgroup keeps 8 workgroups (32 waves) resident per CU, which is what hides the gather of the depth
// words.
#include "lg_common.h"
#include "litegs_hip.h"
#include "lg_binning_internal.h"
#include "lg_tilesort_body.h"
#include "lg_sanity.h"

LG_DEFINE_SANITY_COLLECT(tilesort)

// LDS of one workgroup: four wave-private regions of { exchange buffer 4 KB, digit counters 1 KB } for regime R; the 16 KB bitonic
// buffer of regimes M / L aliases them (the regimes are separated by workgroup barriers).
#define TS_WAVE_WORDS (TS_RADIX_MAX + 256)
template <bool BALLOT, bool ANY_ORDER>
__global__ void __launch_bounds__(256) tile_depth_sort_kernel(int* __restrict__ vals, const int* __restrict__ tile_start,
                                                              const float* __restrict__ depth, int ntiles, long long L, int N,
                                                              uint32_t* __restrict__ scratch, const int* __restrict__ gate)
{
    if (gate != nullptr && *gate == 0) return;              // fallback launch of the depth-bound culling that is not needed
    __shared__ uint64_t lds[(4 * TS_WAVE_WORDS * 4) / 8];   // 20 KB
    static_assert(sizeof(lds) >= sizeof(uint64_t) * TS_CHUNK, "the bitonic buffer must fit");
    const int view = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int t0 = blockIdx.x * 4;                          // this workgroup's tiles: ids t0 + 1 ... t0 + 4
    const int* __restrict__ ts = tile_start + (size_t)view * (ntiles + 2);
    int* __restrict__ v = vals + (size_t)view * L;
    // the depth of a splat comes from the SoA array (4 B x N: 3.5 MB at 0.9 M visible splats, resident in every XCD's L2), not from
    // word 12 of its 64-byte record: 4 M random reads of whole record lines (56 MB of records: L2 misses) cost more than the sort itself
    const float* __restrict__ dz = depth + (size_t)view * N;
    // (an id outside 0..N-1 cannot come out of a correct table; it must not become a gather address: clamped, the blend clamps the same way)
    auto depth_bits = [dz, N](int id) -> uint32_t {
        if ((unsigned)id >= (unsigned)N) lg_note_sanitised(LG_SITE_TILESORT_ID);                 // cold: counted (lg_sanity.h)
        return __float_as_uint(dz[min((unsigned)id, (unsigned)(N - 1))]);
    };

    {   // regime R: every wave its own tile
        const int tile = t0 + wave + 1;
        if (tile <= ntiles) {
            const int start = ts[tile], end = ts[tile + 1];
            const int n = (start >= 0 && end > start) ? end - start : 0;
            if (n >= 2 && n <= TS_RADIX_MAX) {
                uint32_t* w32 = reinterpret_cast<uint32_t*>(lds) + wave * TS_WAVE_WORDS;
                ts_radix_sort_tile<BALLOT, ANY_ORDER>(v + start, n, w32, reinterpret_cast<int*>(w32 + TS_RADIX_MAX), depth_bits, lane);
            }
        }
    }
    for (int w = 0; w < 4; w++) {                           // regimes M and L: the whole workgroup, tile after tile
        const int tile = t0 + w + 1;
        if (tile > ntiles) break;
        const int start = ts[tile], end = ts[tile + 1];
        const int n = (start >= 0 && end > start) ? end - start : 0;
        if (n > TS_RADIX_MAX) {
            __syncthreads();                                // the wave-private regions / the previous long list are free
            ts_sort_tile(v + start, n, lds, scratch ? scratch + (size_t)view * L + start : (uint32_t*)nullptr, depth_bits, 256, false, (int)threadIdx.x);
        }
    }
}

// vals [V, L] int32 splat ids grouped by tile (ascending id inside a tile unless any_order), tile_start [V, ntiles + 2] (lg_tile_range), depth [V, N] view depths.
// scratch [V, L] uint32 (any content; only touched for lists longer than 2048 -- nullable only if such lists cannot occur).
// gate (nullable device int): nothing runs unless *gate != 0.
int lg_tile_depth_sort_gated(int32_t* vals, const int32_t* tile_start, const float* depth, int V, long long L, int N, int ntiles,
                             uint32_t* scratch, int any_order, const int* gate, void* stream)
{
    if (ntiles <= 0 || L <= 0 || V <= 0) return 0;
    // ranking inside a digit: the verified lane-ordered LDS add, or the ballot ranking when the device self-test says otherwise (binning.hip)
    const bool ballot = lg_radix_rank_mode() != 0;
    const dim3 grid(lg_cdiv(ntiles, 4), V), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_TDS(B_, A_) hipLaunchKernelGGL((tile_depth_sort_kernel<B_, A_>), grid, block, 0, s, vals, tile_start, depth, ntiles, L, N, scratch, gate)
    if (!ballot && !any_order) LAUNCH_TDS(false, false);
    else if (!ballot) LAUNCH_TDS(false, true);
    else if (!any_order) LAUNCH_TDS(true, false);
    else LAUNCH_TDS(true, true);
#undef LAUNCH_TDS
    LG_RETURN_LAST();
}

LG_API int lg_tile_depth_sort(int32_t* vals, const int32_t* tile_start, const float* depth, int V, long long L, int N, int ntiles,
                              uint32_t* scratch, void* stream)
{
    if (vals == nullptr || tile_start == nullptr || depth == nullptr || scratch == nullptr) return (int)hipErrorInvalidValue;
    return lg_tile_depth_sort_gated(vals, tile_start, depth, V, L, N, ntiles, scratch, 0, nullptr, stream);
}

// the same for lists that arrive in arbitrary order (lg_tile_group): equal depths are ordered by id explicitly
LG_API int lg_tile_depth_sort_unordered(int32_t* vals, const int32_t* tile_start, const float* depth, int V, long long L, int N, int ntiles,
                                        uint32_t* scratch, void* stream)
{
    if (vals == nullptr || tile_start == nullptr || depth == nullptr || scratch == nullptr) return (int)hipErrorInvalidValue;
    return lg_tile_depth_sort_gated(vals, tile_start, depth, V, L, N, ntiles, scratch, 1, nullptr, stream);
}
