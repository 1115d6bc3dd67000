// Chunk-level culling/compaction, fused gather+activate(+SH) forward/backward, sparse Adam and the
// sparse statistics scatter.
// Parameters are stored [C, chunks, S] (S = 128 Gaussians per chunk): every (channel, chunk) row is a
// contiguous 512-byte line, so a 128-thread workgroup per chunk issues perfectly coalesced loads on each
// of the 59 channels.  These kernels are pure HBM streams (236 B/Gaussian gather, 1652 B/Gaussian Adam).
// Compiled with -ffp-contract=off (bit-comparable visibility decisions vs the CPU oracle).
#include "lg_common.h"
#include "lg_binning_internal.h"
#include "lg_sh.h"

// ---------------------------------------------------------------------------------------------
// a1 frustum_culling_aabb (reference: GR/compact.cu:419-551).
// One 1024-thread workgroup; ORDERED (ascending chunk id) compaction: wave ballots -> LDS table of
// per-(pass,wave) counts -> scan -> second sweep writes ids.  The reference's order is whatever its
// atomics produce; ascending order keeps the gather in activate_forward monotone in HBM and makes
// every downstream tensor reproducible.  M <= ~1M chunks (LDS table).  visible_chunk_id[j>=count] = j
// reproduces the reference's arange() tail.
// ---------------------------------------------------------------------------------------------
#define CULL_TPB 1024
#define CULL_WAVES (CULL_TPB / 64)

__device__ __forceinline__ bool aabb_visible(const float* __restrict__ planes_lds, int V, float ox, float oy, float oz,
                                             float ex, float ey, float ez)
{
    bool gv = false;
    for (int v = 0; v < V; v++) {
        bool vis = true;
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const float* pl = planes_lds + (v * 6 + p) * 4;
            float d_o = pl[0] * ox + pl[1] * oy + pl[2] * oz + pl[3];
            float d_e = fabsf(pl[0]) * ex + fabsf(pl[1]) * ey + fabsf(pl[2]) * ez;
            vis &= ((d_o + d_e) >= 0.0f);
        }
        gv |= vis;
    }
    return gv;
}

template <bool FROM_MASK>
__global__ void __launch_bounds__(CULL_TPB) frustum_culling_kernel(const float* __restrict__ origin, const float* __restrict__ ext,
                                                                   const float* __restrict__ planes, const int* __restrict__ mask, int V, int M,
                                                                   uint8_t* __restrict__ visibility, int* __restrict__ visible_num,
                                                                   int64_t* __restrict__ visible_chunk_id, int* __restrict__ host_feedback,
                                                                   int64_t* __restrict__ rank_out /*nullable: rank_out[m] = position of chunk m*/)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int passes = (M + CULL_TPB - 1) / CULL_TPB;
    float* planes_lds = reinterpret_cast<float*>(smem);                                  // V*24 floats
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(smem + ((V * 24 * 4 + 15) & ~15));  // passes*CULL_WAVES
    int* counts = reinterpret_cast<int*>(masks + (size_t)passes * CULL_WAVES);           // passes*CULL_WAVES (+1)
    __shared__ int wave_tot[CULL_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int k = tid; k < V * 24; k += CULL_TPB) planes_lds[k] = planes[k];
    __syncthreads();

    for (int p = 0; p < passes; p++) {
        int m = p * CULL_TPB + tid;
        bool vis = false;
        if (m < M) {
            if (FROM_MASK) vis = mask[m] != 0;
            else {
                vis = aabb_visible(planes_lds, V, origin[m], origin[(size_t)M + m], origin[2 * (size_t)M + m],
                                   ext[m], ext[(size_t)M + m], ext[2 * (size_t)M + m]);
                visibility[m] = vis ? 1 : 0;
            }
        }
        unsigned long long mask = __ballot(vis);
        if (lane == 0) {
            masks[p * CULL_WAVES + wave] = mask;
            counts[p * CULL_WAVES + wave] = __popcll(mask);
        }
    }
    __syncthreads();
    // exclusive scan of counts[0 .. passes*CULL_WAVES): each thread owns a contiguous slice
    const int total = passes * CULL_WAVES;
    const int per = (total + CULL_TPB - 1) / CULL_TPB;
    int lo = tid * per, hi = min(lo + per, total), local = 0;
    for (int k = lo; k < hi; k++) local += counts[k];
    // block scan of `local`
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int n = __shfl_up(incl, off);
        if (lane >= off) incl += n;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int wave_base = 0;
    for (int w = 0; w < wave; w++) wave_base += wave_tot[w];
    int run = wave_base + incl - local;
    for (int k = lo; k < hi; k++) { int c = counts[k]; counts[k] = run; run += c; }
    __syncthreads();
    int count = 0;
    for (int w = 0; w < CULL_WAVES; w++) count += wave_tot[w];
    if (tid == 0) {
        visible_num[0] = count;
        // GPU-driven sizing feedback (litegs/data.py:238): stored straight into the pinned host buffer, no copy launch
        if (host_feedback) __hip_atomic_store(host_feedback, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }

    for (int p = 0; p < passes; p++) {
        int m = p * CULL_TPB + tid;
        unsigned long long mask = masks[p * CULL_WAVES + wave];
        if ((mask >> lane) & 1ull) {
            int rank = __popcll(mask & ((1ull << lane) - 1ull));
            visible_chunk_id[counts[p * CULL_WAVES + wave] + rank] = m;
            if (rank_out) rank_out[m] = counts[p * CULL_WAVES + wave] + rank;
        }
        if (m < M && m >= count) visible_chunk_id[m] = m;   // arange() tail (never overlaps the compacted prefix)
    }
}

LG_API int lg_frustum_culling_aabb(const float* origin, const float* ext, const float* planes, int V, int M,
                                   uint8_t* visibility, int* visible_num, int64_t* visible_chunk_id, void* stream)
{
    if (M <= 0) return 0;
    LG_REQUIRE(origin, ext, planes, visible_num, visible_chunk_id);
    int passes = (M + CULL_TPB - 1) / CULL_TPB;
    size_t lds = ((V * 24 * 4 + 15) & ~15) + (size_t)passes * CULL_WAVES * (8 + 4) + 16;
    if (lds > 150 * 1024) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(frustum_culling_kernel<false>, dim3(1), dim3(CULL_TPB), lds, (hipStream_t)stream,
                       origin, ext, planes, (const int*)nullptr, V, M, visibility, visible_num, visible_chunk_id, (int*)nullptr, (int64_t*)nullptr);
    LG_RETURN_LAST();
}

int lg_frustum_culling_fb(const float* origin, const float* ext, const float* planes, int V, int M, uint8_t* visibility, int* visible_num,
                          int64_t* visible_chunk_id, int* host_feedback, void* stream)
{
    if (M <= 0) return 0;
    int passes = (M + CULL_TPB - 1) / CULL_TPB;
    size_t lds = ((V * 24 * 4 + 15) & ~15) + (size_t)passes * CULL_WAVES * (8 + 4) + 16;
    if (lds > 150 * 1024) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(frustum_culling_kernel<false>, dim3(1), dim3(CULL_TPB), lds, (hipStream_t)stream,
                       origin, ext, planes, (const int*)nullptr, V, M, visibility, visible_num, visible_chunk_id, host_feedback, (int64_t*)nullptr);
    LG_RETURN_LAST();
}

// Multi-workgroup variant for the native executor: 256 chunks per workgroup, the ordered compaction offsets are chained with
// decoupled look-back (one 64-bit status word per workgroup = launch epoch | flag | count, so the table never needs clearing;
// `scratch` is a persistent zero-initialised buffer owned by the caller: [0] ticket counter, [2..] status words).
// 34 us (one workgroup, 23 sequential sweeps at 3 M Gaussians) -> a few us.
#define CHAIN_TPB 256
__global__ void __launch_bounds__(CHAIN_TPB) frustum_culling_chain_kernel(const float* __restrict__ origin, const float* __restrict__ ext,
                                                                          const float* __restrict__ planes, int V, int M,
                                                                          uint8_t* __restrict__ visibility, int* __restrict__ visible_num,
                                                                          int64_t* __restrict__ visible_chunk_id,
                                                                          unsigned long long* __restrict__ status, int* __restrict__ ticket,
                                                                          unsigned int epoch, int ticket_base, int* __restrict__ host_feedback)
{
    __shared__ float planes_lds[8 * 24];
    __shared__ int wcnt[CHAIN_TPB / 64];
    __shared__ int bid_s, excl_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblocks = gridDim.x;
    if (tid == 0) bid_s = atomicAdd(ticket, 1) - ticket_base;
    for (int k = tid; k < V * 24; k += CHAIN_TPB) planes_lds[k] = planes[k];
    __syncthreads();
    const int bid = bid_s;
    const int m = bid * CHAIN_TPB + tid;
    bool vis = false;
    if (m < M) {
        vis = aabb_visible(planes_lds, V, origin[m], origin[(size_t)M + m], origin[2 * (size_t)M + m],
                           ext[m], ext[(size_t)M + m], ext[2 * (size_t)M + m]);
        visibility[m] = vis ? 1 : 0;
    }
    const unsigned long long bm = __ballot(vis);
    if (lane == 0) wcnt[wave] = __popcll(bm);
    __syncthreads();
    const int block_total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    const unsigned long long tag = (unsigned long long)epoch << 34;          // bits 34..63 epoch, 32/33 flags, 0..31 count
    const unsigned long long F_AGG = 1ull << 32, F_INC = 1ull << 33;
    if (wave == 0) {
        unsigned int excl = 0;
        if (bid == 0) {
            if (lane == 0) __hip_atomic_store(status, tag | F_INC | (unsigned long long)block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(status + bid, tag | F_AGG | (unsigned long long)block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int b = bid - 1;
            while (true) {
                unsigned long long w = (b - lane >= 0) ? __hip_atomic_load(status + (b - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (tag | F_INC);
                const bool fresh = (w >> 34) == (unsigned long long)epoch;   // words of earlier launches count as "not ready"
                const unsigned long long inc_m = __ballot(fresh && (w & F_INC));
                const unsigned long long nr_m = __ballot(!fresh || (w & (F_AGG | F_INC)) == 0ull);
                const int first_inc = inc_m ? __ffsll((long long)inc_m) - 1 : 64;
                const int first_nr = nr_m ? __ffsll((long long)nr_m) - 1 : 64;
                const int take = first_nr < first_inc ? first_nr : (first_inc < 64 ? first_inc + 1 : 64);
                unsigned int part = (lane < take) ? (unsigned int)(w & 0xffffffffull) : 0u;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
                excl += __shfl(part, 0);
                if (first_inc < first_nr) break;
                b -= take;
                if (take < 64) __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) __hip_atomic_store(status + bid, tag | F_INC | (unsigned long long)(excl + (unsigned int)block_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) excl_s = (int)excl;
    }
    __syncthreads();
    int pos = excl_s + __popcll(bm & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) pos += wcnt[w];
    if (vis) visible_chunk_id[pos] = m;
    if (bid == nblocks - 1) {                                            // the last workgroup knows the total
        const int count = excl_s + block_total;
        if (tid == 0) {
            visible_num[0] = count;
            if (host_feedback) __hip_atomic_store(host_feedback, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        for (int k = count + tid; k < M; k += CHAIN_TPB) visible_chunk_id[k] = k;      // arange() tail, as the reference
    }
}

long long lg_cull_scratch_bytes(int M) { return 16 + 8LL * ((M + CHAIN_TPB - 1) / CHAIN_TPB); }

// scratch: lg_cull_scratch_bytes(M) bytes, zero-initialised ONCE by the caller and then only passed to this function with the same M;
// epoch = 1, 2, 3, ... (one per call on this scratch)
int lg_frustum_culling_chain(const float* origin, const float* ext, const float* planes, int V, int M, uint8_t* visibility, int* visible_num,
                             int64_t* visible_chunk_id, void* scratch, unsigned int epoch, int* host_feedback, void* stream)
{
    if (M <= 0) return 0;
    if (V > 8 || epoch == 0 || epoch >= (1u << 30)) return (int)hipErrorInvalidValue;
    const int nblocks = (M + CHAIN_TPB - 1) / CHAIN_TPB;
    const long long base = (long long)(epoch - 1) * nblocks;
    if (base > 0x7fffffffLL - nblocks) return (int)hipErrorInvalidValue;      // caller re-creates the scratch long before this
    hipLaunchKernelGGL(frustum_culling_chain_kernel, dim3(nblocks), dim3(CHAIN_TPB), 0, (hipStream_t)stream, origin, ext, planes, V, M,
                       visibility, visible_num, visible_chunk_id, (unsigned long long*)((char*)scratch + 16), (int*)scratch, epoch,
                       (int)base, host_feedback);
    LG_RETURN_LAST();
}

// Ordered compaction of an int32 mask (non-zero = keep) with the same output contract as frustum_culling_aabb:
// used by the data-parallel path to turn the all-reduced visibility mask into the union chunk list.
LG_API int lg_compact_mask_rank(const int* mask, int M, int* count, int64_t* ids, int64_t* rank, void* stream);

LG_API int lg_compact_mask(const int* mask, int M, int* count, int64_t* ids, void* stream)
{
    return lg_compact_mask_rank(mask, M, count, ids, nullptr, stream);
}

// same, plus the inverse map: rank[m] = position of chunk m in ids for every kept chunk (other entries untouched)
LG_API int lg_compact_mask_rank(const int* mask, int M, int* count, int64_t* ids, int64_t* rank, void* stream)
{
    if (M <= 0) return 0;
    int passes = (M + CULL_TPB - 1) / CULL_TPB;
    size_t lds = 16 + (size_t)passes * CULL_WAVES * (8 + 4) + 16;
    if (lds > 150 * 1024) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(frustum_culling_kernel<true>, dim3(1), dim3(CULL_TPB), lds, (hipStream_t)stream,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, mask, 0, M, (uint8_t*)nullptr, count, ids,
                       (int*)nullptr, rank);
    LG_RETURN_LAST();
}

// mask[ids[i]] = 1 for i < *count  (mask pre-zeroed)
__global__ void __launch_bounds__(256) mark_chunks_kernel(const int64_t* __restrict__ ids, const int* __restrict__ count, int A, int* __restrict__ mask)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < A && i < count[0]) mask[ids[i]] = 1;
}

LG_API int lg_mark_chunks(const int64_t* ids, const int* count, int A, int* mask, void* stream)
{
    if (A <= 0) return 0;
    hipLaunchKernelGGL(mark_chunks_kernel, dim3(lg_cdiv(A, 256)), dim3(256), 0, (hipStream_t)stream, ids, count, A, mask);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// a2 cull_compact_activate (GR/compact.cu:826-893, 983-1085)
// grid = allocated chunks, block = S (chunk size).  Blocks >= *visible_chunks_num only zero opacity.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void camera_center(const float* __restrict__ V, float& cx, float& cy, float& cz)
{
    float ix = -V[12], iy = -V[13], iz = -V[14];
    cx = ix * V[0] + iy * V[1] + iz * V[2];
    cy = ix * V[4] + iy * V[5] + iz * V[6];
    cz = ix * V[8] + iy * V[9] + iz * V[10];
}

template <int DEG>
__global__ void activate_forward_kernel(const int64_t* __restrict__ visible_chunk_id, const int* __restrict__ visible_chunks_num,
                                        const float* __restrict__ view, int V,
                                        const float* __restrict__ pos, const float* __restrict__ scale, const float* __restrict__ rot,
                                        const float* __restrict__ sh0, const float* __restrict__ shr, const float* __restrict__ opa,
                                        int C, int S, int A,
                                        float* __restrict__ o_pos, float* __restrict__ o_scale, float* __restrict__ o_rot,
                                        float* __restrict__ o_color, float* __restrict__ o_opa)
{
    const int a = blockIdx.x, i = threadIdx.x;
    const size_t od = (size_t)a * S + i;
    const size_t CS = (size_t)C * S, AS = (size_t)A * S;
    if (a >= visible_chunks_num[0]) { o_opa[od] = 0.0f; return; }
    const size_t sd = (size_t)visible_chunk_id[a] * S + i;

    float px = pos[sd], py = pos[CS + sd], pz = pos[2 * CS + sd];
    o_pos[od] = px; o_pos[AS + od] = py; o_pos[2 * AS + od] = pz; o_pos[3 * AS + od] = 1.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) o_scale[k * AS + od] = __expf(scale[k * CS + sd]);
    float w = rot[sd], x = rot[CS + sd], y = rot[2 * CS + sd], z = rot[3 * CS + sd];
    float rn = rsqrtf(w * w + x * x + y * y + z * z + 1e-12f);
    o_rot[od] = w * rn; o_rot[AS + od] = x * rn; o_rot[2 * AS + od] = y * rn; o_rot[3 * AS + od] = z * rn;
    o_opa[od] = 1.0f / (1.0f + __expf(-opa[sd]));

    constexpr int NB = (DEG + 1) * (DEG + 1);
    if (V == 1) {
        float cx, cy, cz;
        camera_center(view, cx, cy, cz);
        float dx = px - cx, dy = py - cy, dz = pz - cz;
        float nr = rsqrtf(dx * dx + dy * dy + dz * dz + 1e-12f);
        float b[16];
        lg_sh_basis<DEG>(dx * nr, dy * nr, dz * nr, b);
        float r0 = b[0] * sh0[sd], r1 = b[0] * sh0[CS + sd], r2 = b[0] * sh0[2 * CS + sd];
#pragma unroll
        for (int k = 1; k < NB; k++) {
            const float* s = shr + ((size_t)(k - 1) * 3) * CS + sd;
            r0 += b[k] * s[0]; r1 += b[k] * s[CS]; r2 += b[k] * s[2 * CS];
        }
        o_color[od] = r0 + 0.5f; o_color[AS + od] = r1 + 0.5f; o_color[2 * AS + od] = r2 + 0.5f;
    } else {
        for (int v = 0; v < V; v++) {
            float cx, cy, cz;
            camera_center(view + v * 16, cx, cy, cz);
            float dx = px - cx, dy = py - cy, dz = pz - cz;
            float nr = rsqrtf(dx * dx + dy * dy + dz * dz + 1e-12f);
            float b[16];
            lg_sh_basis<DEG>(dx * nr, dy * nr, dz * nr, b);
            for (int ch = 0; ch < 3; ch++) {
                float r = b[0] * sh0[ch * CS + sd];
#pragma unroll
                for (int k = 1; k < NB; k++) r += b[k] * shr[((size_t)(k - 1) * 3 + ch) * CS + sd];
                o_color[((size_t)v * 3 + ch) * AS + od] = r + 0.5f;
            }
        }
    }
}

LG_API int lg_cull_compact_activate(int degree, const int64_t* visible_chunk_id, const int* visible_chunks_num, int A,
                                    const float* view, int V,
                                    const float* pos, const float* scale, const float* rot, const float* sh0, const float* shr,
                                    const float* opa, int C, int S,
                                    float* o_pos, float* o_scale, float* o_rot, float* o_color, float* o_opa, void* stream)
{
    if (A <= 0) return 0;
    LG_REQUIRE(visible_chunk_id, visible_chunks_num, view, pos, scale, rot, sh0, opa, o_pos, o_scale, o_rot, o_color, o_opa);
    if (S > 1024 || S <= 0) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_ACT(D) hipLaunchKernelGGL(activate_forward_kernel<D>, dim3(A), dim3(S), 0, s, visible_chunk_id, visible_chunks_num, view, V, \
                                         pos, scale, rot, sh0, shr, opa, C, S, A, o_pos, o_scale, o_rot, o_color, o_opa)
    switch (degree) {
    case 0: LAUNCH_ACT(0); break;
    case 1: LAUNCH_ACT(1); break;
    case 2: LAUNCH_ACT(2); break;
    case 3: LAUNCH_ACT(3); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef LAUNCH_ACT
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// a19 activate_backward (GR/compact.cu:896-980, 1087-1212).  d_opacity_raw = g * sigmoid(x) (sic,
// compact.cu:952 -- reproduced for training parity).  The whole sh_rest gradient is
// written here (inactive degrees = 0), so no separate zero-fill pass is needed.
// ---------------------------------------------------------------------------------------------
template <int DEG>
__global__ void activate_backward_kernel(const int64_t* __restrict__ visible_chunk_id, const int* __restrict__ visible_chunks_num,
                                         const float* __restrict__ view, int V,
                                         const float* __restrict__ pos, const float* __restrict__ scale, const float* __restrict__ rot,
                                         const float* __restrict__ opa, int C, int S, int A, int R,
                                         const float* __restrict__ g_pos, const float* __restrict__ g_scale, const float* __restrict__ g_rot,
                                         const float* __restrict__ g_color, const float* __restrict__ g_opa,
                                         float* __restrict__ d_pos, float* __restrict__ d_scale, float* __restrict__ d_rot,
                                         float* __restrict__ d_sh0, float* __restrict__ d_shr, float* __restrict__ d_opa)
{
    const int a = blockIdx.x, i = threadIdx.x;
    const size_t od = (size_t)a * S + i;
    const size_t CS = (size_t)C * S, AS = (size_t)A * S;
    constexpr int NB = (DEG + 1) * (DEG + 1);
    if (a >= visible_chunks_num[0]) {
        // the tail is "dirty" in the reference except sh_rest (pre-zeroed); keep sh_rest zero there too
        for (int k = 0; k < R * 3; k++) d_shr[(size_t)k * AS + od] = 0.0f;
        return;
    }
    const size_t sd = (size_t)visible_chunk_id[a] * S + i;
#pragma unroll
    for (int k = 0; k < 3; k++) d_pos[k * AS + od] = g_pos[k * AS + od];
#pragma unroll
    for (int k = 0; k < 3; k++) d_scale[k * AS + od] = __expf(scale[k * CS + sd]) * g_scale[k * AS + od];
    float w = rot[sd], x = rot[CS + sd], y = rot[2 * CS + sd], z = rot[3 * CS + sd];
    float rn = rsqrtf(w * w + x * x + y * y + z * z + 1e-12f);
    float ow = w * rn, ox = x * rn, oy = y * rn, oz = z * rn;
    float g0 = g_rot[od], g1 = g_rot[AS + od], g2 = g_rot[2 * AS + od], g3 = g_rot[3 * AS + od];
    float dot = g0 * ow + g1 * ox + g2 * oy + g3 * oz;
    d_rot[od] = rn * (g0 - dot * ow);
    d_rot[AS + od] = rn * (g1 - dot * ox);
    d_rot[2 * AS + od] = rn * (g2 - dot * oy);
    d_rot[3 * AS + od] = rn * (g3 - dot * oz);
    d_opa[od] = g_opa[od] * (1.0f - 1.0f / (1.0f + __expf(opa[sd])));

    float px = pos[sd], py = pos[CS + sd], pz = pos[2 * CS + sd];
    if (V == 1) {
        float cx, cy, cz;
        camera_center(view, cx, cy, cz);
        float dx = px - cx, dy = py - cy, dz = pz - cz;
        float nr = rsqrtf(dx * dx + dy * dy + dz * dz + 1e-12f);
        float b[16];
        lg_sh_basis<DEG>(dx * nr, dy * nr, dz * nr, b);
        float c0 = g_color[od], c1 = g_color[AS + od], c2 = g_color[2 * AS + od];
        d_sh0[od] = b[0] * c0; d_sh0[AS + od] = b[0] * c1; d_sh0[2 * AS + od] = b[0] * c2;
#pragma unroll
        for (int k = 1; k < NB; k++) {
            float* d = d_shr + ((size_t)(k - 1) * 3) * AS + od;
            d[0] = b[k] * c0; d[AS] = b[k] * c1; d[2 * AS] = b[k] * c2;
        }
    } else {
        float acc[NB * 3];
#pragma unroll
        for (int k = 0; k < NB * 3; k++) acc[k] = 0.0f;
        for (int v = 0; v < V; v++) {
            float cx, cy, cz;
            camera_center(view + v * 16, cx, cy, cz);
            float dx = px - cx, dy = py - cy, dz = pz - cz;
            float nr = rsqrtf(dx * dx + dy * dy + dz * dz + 1e-12f);
            float b[16];
            lg_sh_basis<DEG>(dx * nr, dy * nr, dz * nr, b);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                float g = g_color[((size_t)v * 3 + ch) * AS + od];
#pragma unroll
                for (int k = 0; k < NB; k++) acc[k * 3 + ch] += b[k] * g;
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++) d_sh0[ch * AS + od] = acc[ch];
#pragma unroll
        for (int k = 1; k < NB; k++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) d_shr[((size_t)(k - 1) * 3 + ch) * AS + od] = acc[k * 3 + ch];
    }
    for (int k = NB - 1; k < R; k++)
        for (int ch = 0; ch < 3; ch++) d_shr[((size_t)k * 3 + ch) * AS + od] = 0.0f;
}

LG_API int lg_activate_backward(int degree, const int64_t* visible_chunk_id, const int* visible_chunks_num, int A,
                                const float* view, int V,
                                const float* pos, const float* scale, const float* rot, const float* opa, int C, int S, int R,
                                const float* g_pos, const float* g_scale, const float* g_rot, const float* g_color, const float* g_opa,
                                float* d_pos, float* d_scale, float* d_rot, float* d_sh0, float* d_shr, float* d_opa, void* stream)
{
    if (A <= 0) return 0;
    LG_REQUIRE(visible_chunk_id, visible_chunks_num, view, pos, scale, rot, opa, d_pos, d_scale, d_rot, d_sh0, d_opa);
    if (S > 1024 || S <= 0) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_ACTB(D) hipLaunchKernelGGL(activate_backward_kernel<D>, dim3(A), dim3(S), 0, s, visible_chunk_id, visible_chunks_num, view, V, \
                                          pos, scale, rot, opa, C, S, A, R, g_pos, g_scale, g_rot, g_color, g_opa,                           \
                                          d_pos, d_scale, d_rot, d_sh0, d_shr, d_opa)
    switch (degree) {
    case 0: LAUNCH_ACTB(0); break;
    case 1: LAUNCH_ACTB(1); break;
    case 2: LAUNCH_ACTB(2); break;
    case 3: LAUNCH_ACTB(3); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef LAUNCH_ACTB
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// a20 adamUpdate (GR/compact.cu:320-417): m=b1 m+(1-b1)g; v=b2 v+(1-b2)g^2; p -= lr*m/(sqrt(v)+eps);
// no bias correction.  Chunk form: param/m/v [E, chunks, S], grad [E, A, S] (compact), rows a < *valid.
// One workgroup = (compact chunk a, 8 channels); a lane owns 4 consecutive Gaussians -> dwordx4 on all
// seven streams (28 B/element, the largest HBM term of a training iteration).
// ---------------------------------------------------------------------------------------------
#define ADAM_TPB 256
__global__ void __launch_bounds__(ADAM_TPB) adam_chunk_kernel_v4(float* __restrict__ param, const float* __restrict__ grad,
                                                                 float* __restrict__ m, float* __restrict__ v,
                                                                 const int64_t* __restrict__ visible_chunk_id, const int* __restrict__ valid_length,
                                                                 int E, int chunks, int A, int S, int grad_dense,
                                                                 float lr, float b1, float b2, float eps)
{
    const int a = blockIdx.x;
    if (valid_length != nullptr && a >= valid_length[0]) return;
    const int quads = S >> 2;                       // float4 per row
    const int rows_per_block = ADAM_TPB / quads;    // channels handled by one block
    const int e = blockIdx.y * rows_per_block + threadIdx.x / quads;
    const int q = threadIdx.x % quads;
    if (e >= E || (int)threadIdx.x >= rows_per_block * quads) return;
    const size_t chunk = (size_t)visible_chunk_id[a];
    const size_t po = (((size_t)e * chunks + chunk) * S) / 4 + q;
    const size_t go = grad_dense ? po : (((size_t)e * A + a) * S) / 4 + q;
    float4 g = reinterpret_cast<const float4*>(grad)[go];
    float4 mm = reinterpret_cast<float4*>(m)[po];
    float4 vv = reinterpret_cast<float4*>(v)[po];
    // exact no-op: zero gradient on zero moments gives m' = v' = 0 and p' = p - lr*0/(0+eps) = p, i.e. what memory already holds --
    // the parameter is not read and nothing is written (28 -> 12 B per element; most Gaussians of the visible chunks of a frame
    // receive no gradient, and until one does their moments are zero).  NaN / inf gradients are not "zero" and take the update.
    {
        const unsigned int any = (__float_as_uint(g.x) | __float_as_uint(g.y) | __float_as_uint(g.z) | __float_as_uint(g.w) |
                                  __float_as_uint(mm.x) | __float_as_uint(mm.y) | __float_as_uint(mm.z) | __float_as_uint(mm.w) |
                                  __float_as_uint(vv.x) | __float_as_uint(vv.y) | __float_as_uint(vv.z) | __float_as_uint(vv.w)) << 1;
        if (any == 0u) return;
    }
    float4 p = reinterpret_cast<float4*>(param)[po];
#define ADAM1(c)                                                      \
    mm.c = b1 * mm.c + (1.0f - b1) * g.c;                             \
    vv.c = b2 * vv.c + (1.0f - b2) * g.c * g.c;                       \
    p.c += -lr * mm.c / (sqrtf(vv.c) + eps);
    ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
    reinterpret_cast<float4*>(param)[po] = p;
    reinterpret_cast<float4*>(m)[po] = mm;
    reinterpret_cast<float4*>(v)[po] = vv;
}

__global__ void adam_chunk_kernel_generic(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m,
                                          float* __restrict__ v, const int64_t* __restrict__ visible_chunk_id,
                                          const int* __restrict__ valid_length, int E, int chunks, int A, int S, int grad_dense,
                                          float lr, float b1, float b2, float eps)
{
    const int a = blockIdx.x, e = blockIdx.y, i = threadIdx.x;
    if (valid_length != nullptr && a >= valid_length[0]) return;
    const size_t po = ((size_t)e * chunks + (size_t)visible_chunk_id[a]) * S + i;
    const size_t go = grad_dense ? po : ((size_t)e * A + a) * S + i;
    float g = grad[go];
    float mm = b1 * m[po] + (1.0f - b1) * g;
    float vv = b2 * v[po] + (1.0f - b2) * g * g;
    param[po] += -lr * mm / (sqrtf(vv) + eps);
    m[po] = mm; v[po] = vv;
}

LG_API int lg_adam_update_chunk(float* param, const float* grad, float* m, float* v, const int64_t* visible_chunk_id,
                                const int* valid_length, int E, int chunks, int A, int S, int grad_dense,
                                float lr, float b1, float b2, float eps, void* stream)
{
    if (A <= 0 || E <= 0) return 0;
    LG_REQUIRE(param, grad, m, v, visible_chunk_id);
    hipStream_t s = (hipStream_t)stream;
    if (S % 4 == 0 && (S / 4) <= ADAM_TPB && ADAM_TPB % (S / 4) == 0) {
        int rows = ADAM_TPB / (S / 4);
        hipLaunchKernelGGL(adam_chunk_kernel_v4, dim3(A, lg_cdiv(E, rows)), dim3(ADAM_TPB), 0, s, param, grad, m, v, visible_chunk_id,
                           valid_length, E, chunks, A, S, grad_dense, lr, b1, b2, eps);
    } else {
        if (S > 1024) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(adam_chunk_kernel_generic, dim3(A, E), dim3(S), 0, s, param, grad, m, v, visible_chunk_id, valid_length,
                           E, chunks, A, S, grad_dense, lr, b1, b2, eps);
    }
    LG_RETURN_LAST();
}

// primitive form (GR/compact.cu:348-375): param/grad/m/v [E,N]; updated where mask[i] != 0
__global__ void __launch_bounds__(256) adam_primitive_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m,
                                                             float* __restrict__ v, const int64_t* __restrict__ mask, int E, int N,
                                                             float lr, float b1, float b2, float eps)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N || mask[i] == 0) return;
    for (int e = 0; e < E; e++) {
        size_t o = (size_t)e * N + i;
        float g = grad[o];
        float mm = b1 * m[o] + (1.0f - b1) * g;
        float vv = b2 * v[o] + (1.0f - b2) * g * g;
        param[o] += -lr * mm / (sqrtf(vv) + eps);
        m[o] = mm; v[o] = vv;
    }
}

LG_API int lg_adam_update_primitive(float* param, const float* grad, float* m, float* v, const int64_t* mask, int E, int N,
                                    float lr, float b1, float b2, float eps, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(param, grad, m, v, mask);
    hipLaunchKernelGGL(adam_primitive_kernel, dim3(lg_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, mask, E, N, lr, b1, b2, eps);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// a21 gpu_driven_pipeline_sparse_op (GR/compact.cu:1222-1336): A[:, chunk_ids[i], :] op= B[:, i, :], i < *count
// dtype: 0 = float32, 1 = int32, 2 = int64, 3 = float64, 4 = int16, 5 = uint8/int8(add only as int8)
// ---------------------------------------------------------------------------------------------
template <typename T, int OP>
__global__ void sparse_scatter_kernel(T* __restrict__ A, const T* __restrict__ B, const int64_t* __restrict__ chunk_ids,
                                      const int* __restrict__ valid_count, int chunks, int alloc, int S)
{
    const int src = blockIdx.x, e = blockIdx.y;
    if (src >= valid_count[0]) return;
    const size_t dst = (size_t)chunk_ids[src];
    if (dst >= (size_t)chunks) return;                  // id outside the destination (e.g. a truncated union list): dropped
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        T b = B[((size_t)e * alloc + src) * S + i];
        T* p = A + ((size_t)e * chunks + dst) * S + i;
        if (OP == 0) *p = *p + b;
        else if (OP == 1) *p = (*p < b) ? *p : b;
        else *p = (*p > b) ? *p : b;
    }
}

template <typename T>
static int launch_scatter(void* A, const void* B, const int64_t* ids, const int* cnt, int E, int chunks, int alloc, int S, int op, hipStream_t s)
{
    dim3 grid(alloc, E), block(S > 256 ? 256 : S);
    switch (op) {
    case 0: hipLaunchKernelGGL((sparse_scatter_kernel<T, 0>), grid, block, 0, s, (T*)A, (const T*)B, ids, cnt, chunks, alloc, S); break;
    case 1: hipLaunchKernelGGL((sparse_scatter_kernel<T, 1>), grid, block, 0, s, (T*)A, (const T*)B, ids, cnt, chunks, alloc, S); break;
    case 2: hipLaunchKernelGGL((sparse_scatter_kernel<T, 2>), grid, block, 0, s, (T*)A, (const T*)B, ids, cnt, chunks, alloc, S); break;
    default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

LG_API int lg_sparse_scatter(void* A, const void* B, const int64_t* chunk_ids, const int* valid_count,
                             int E, int chunks, int alloc, int S, int dtype, int op, void* stream)
{
    if (alloc <= 0 || E <= 0 || S <= 0) return 0;
    LG_REQUIRE(A, B, chunk_ids, valid_count);
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
    case 0: return launch_scatter<float>(A, B, chunk_ids, valid_count, E, chunks, alloc, S, op, s);
    case 1: return launch_scatter<int32_t>(A, B, chunk_ids, valid_count, E, chunks, alloc, S, op, s);
    case 2: return launch_scatter<int64_t>(A, B, chunk_ids, valid_count, E, chunks, alloc, S, op, s);
    case 3: return launch_scatter<double>(A, B, chunk_ids, valid_count, E, chunks, alloc, S, op, s);
    case 4: return launch_scatter<int16_t>(A, B, chunk_ids, valid_count, E, chunks, alloc, S, op, s);
    case 5: return launch_scatter<int8_t>(A, B, chunk_ids, valid_count, E, chunks, alloc, S, op, s);
    default: return (int)hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// Statistic epochs of the executor: everything the statistics helper accumulates per frame (litegs/utils/statistic_helper.py:
// visible_count, the moments "fragment_weight" and "fragment_err") in ONE pass over the frame's gradient records, instead of ~15 torch
// elementwise / scatter launches around the blend kernels.  Per compacted Gaussian i of visible chunk a:
//   visible += (tile count != 0)                                        wrapper.py:733-736 (b_visible = allocate_size != 0)
//   fragment_weight: sum += w, square_sum += w * w, count += n          n = fragment count (record slot 9), w = weight sum (slot 10)
//   fragment_err:    sum += M0 / opacity, square_sum += e2, count += n  M0 = record slot 8, e2 = err_square (slot 11)
// The adds are per-frame scatters into [chunks, S] arrays exactly as gpu_driven_pipeline_sparse_op performs them (one writer per word).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128) stat_accumulate_kernel(const float4* __restrict__ packed_grad, const float4* __restrict__ packed,
                                                              const int* __restrict__ alloc, const int64_t* __restrict__ chunk_ids,
                                                              const int* __restrict__ valid_count, int chunks, int S,
                                                              int* __restrict__ visible_count,
                                                              float* __restrict__ w_sum, float* __restrict__ w_sq, int* __restrict__ w_cnt,
                                                              float* __restrict__ e_sum, float* __restrict__ e_sq, int* __restrict__ e_cnt)
{
    const int a = blockIdx.x;
    if (a >= valid_count[0]) return;
    const size_t dstc = (size_t)chunk_ids[a];
    if (dstc >= (size_t)chunks) return;
    for (int t = threadIdx.x; t < S; t += blockDim.x) {
        const size_t i = (size_t)a * S + t, d = dstc * S + t;
        const float4 g2 = packed_grad[i * 4 + 2];                 // slots 8 (M0), 9 (count), 10 (weight), 11 (err_square)
        const float o = packed[i * 4 + 1].y;                      // record dword 5: activated opacity
        const int n = (int)__builtin_rintf(g2.y);
        if (alloc[i] != 0) visible_count[d] += 1;
        w_sum[d] += g2.z; w_sq[d] += g2.z * g2.z; w_cnt[d] += n;
        e_sum[d] += (o > 0.0f) ? g2.x / o : 0.0f; e_sq[d] += g2.w; e_cnt[d] += n;
    }
}

LG_API int lg_stat_accumulate(const float* packed_grad, const float* packed, const int* alloc, const int64_t* chunk_ids, const int* valid_count,
                              int A, int chunks, int S, int* visible_count, float* w_sum, float* w_sq, int* w_cnt,
                              float* e_sum, float* e_sq, int* e_cnt, void* stream)
{
    if (A <= 0 || S <= 0) return 0;
    LG_REQUIRE(packed_grad, packed, alloc, chunk_ids, valid_count, visible_count, w_sum, w_sq, w_cnt, e_sum, e_sq, e_cnt);
    hipLaunchKernelGGL(stat_accumulate_kernel, dim3(A), dim3(128), 0, (hipStream_t)stream, (const float4*)packed_grad, (const float4*)packed, alloc,
                       chunk_ids, valid_count, chunks, S, visible_count, w_sum, w_sq, w_cnt, e_sum, e_sq, e_cnt);
    LG_RETURN_LAST();
}

// 4-byte device -> pinned-host feedback copy on the caller's stream (GR/compact.cu:538, GR/binning.cu:148)
LG_API int lg_feedback_d2h(int* host_dst, const int* device_src, void* stream)
{
    return (int)hipMemcpyAsync(host_dst, device_src, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
}
