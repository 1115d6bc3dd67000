// Data-parallel gradient exchange on the device (the reference is single-GPU only -- no counterpart).
//
// One camera frame per GPU, full replicas.  What a frame contributes to the parameter gradients is fully described by the blend
// backward's MOMENT records (raster.hip: nine floats per Gaussian that received any gradient) plus that frame's camera: the
// per-Gaussian chain backward is a cheap function of (raw parameters, camera, moments).  So the ranks exchange the records of the
// Gaussians they touched -- 40 bytes each instead of the 236-byte parameter gradient (59 floats) -- and every rank replays the
// chain backward of EVERY rank's records with that rank's camera, sums the parameter gradients in rank order in registers
// (deterministic: replicas stay bit-identical), scales by 1/W and applies Adam in the same kernel.  The parameter gradients never
// exist in HBM, on any rank; backward + Adam stay fused exactly as on one GPU.
//
//   lg_dp_compact_moments : packed_grad [N,16] -> block of (1 + cap) * 10 words: a 10-word header (word 0: number of touched Gaussians K)
//                           followed by ten ROWS of cap words -- row 0 the global Gaussian indices (int bits), rows 1..9 the nine moments
//                           (structure of arrays: the compaction's stores and the consumer's loads are coalesced 4-byte streams; the
//                           first version's 40-byte records cost ten 2.5 KB-strided stores / loads per wave).  Unordered; one returning
//                           atomic per 1024 records (one per wave serialised 16 k atomics on the header word: 140 us in the trained
//                           state, profiles/r04_dp_glue.log).
//   (RCCL all_gather of the W blocks -- torch.distributed, litegs_amd/dp.py)
//   lg_dp_build_slotmap   : for every rank r and record k: slot[r][gid] = k + 1; also the job's largest K to a pinned host word
//                           (sizes the next visit's blocks) and an overflow flag if some K exceeded the capacity.
//   lg_dp_backward_adam   : over the union of the ranks' visible chunks: per Gaussian, for r = 0..W-1 in order, the record (if any)
//                           through gaussian_backward with camera r; mean; Adam on param / exp_avg / exp_avg_sq; the slot entries
//                           read are reset to 0 (the map needs no clearing launch).
// Compiled with -ffp-contract=off (same chain arithmetic as fused.hip).
#include "lg_common.h"
#include "lg_chain.h"
#include "lg_gaussian_bwd.h"
#include "litegs_hip.h"

#define DP_REC 10              // floats per exchanged record
#define DP_MAX_WORLD 8

// ---------------------------------------------------------------------------------------------
#define DP_BATCH 1024          // records per workgroup: one returning atomic on the header word per batch
__global__ void __launch_bounds__(256) dp_compact_kernel(const float4* __restrict__ packed_grad, const int64_t* __restrict__ vis_ids,
                                                         const int* __restrict__ vis_num, int A, int S, int cap, float* __restrict__ block,
                                                         const int* __restrict__ hot_of /*nullable: gradient replicas to fold (raster.hip)*/,
                                                         int* __restrict__ hot_counter /*nullable: reset for the next frame's projection*/,
                                                         const int* __restrict__ poison /*nullable: speculative culling, see dp_slotmap_kernel*/)
{
    __shared__ int cnt[4][4];          // [j][wave]: touched records of round j in that wave
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long N = (long long)A * S;
    const long long i0 = (long long)blockIdx.x * DP_BATCH;
    const int nvis = vis_num[0];
    if (hot_counter != nullptr && blockIdx.x == 0 && tid == 0) *hot_counter = 0;
    // header word 1: "this rank's culled forward failed (or an earlier step did)" travels with the records, so that every rank learns it
    // from the same gathered headers
    if (poison != nullptr && blockIdx.x == 0 && tid == 0) reinterpret_cast<int*>(block)[1] = (*poison != 0) ? 1 : 0;
    // thread t takes records i0 + j * 256 + t, j = 0..3: every wave instruction reads 64 consecutive 64-byte lines
    float mom[4][9];
    int gid[4], rank[4];
    bool nz[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const long long i = i0 + (long long)j * 256 + tid;
        nz[j] = false; gid[j] = 0;
        if (i < N) {
            const int a = (int)(i / S), t = (int)(i % S);
            if (a < nvis) {
                load_moments_folded(packed_grad, (size_t)i, N, hot_of, mom[j]);
#pragma unroll
                for (int k = 0; k < 9; k++) nz[j] |= (__float_as_uint(mom[j][k]) & 0x7fffffffu) != 0u;     // +-0 adds nothing to a sum; NaN/inf travel
                gid[j] = (int)(vis_ids[a] * S + t);
            }
        }
        const unsigned long long m = __ballot(nz[j]);
        rank[j] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) cnt[j][wave] = __popcll(m);
    }
    __syncthreads();
    // Records leave in ASCENDING index order inside a batch (round-major, then wave, then lane): the consumer's workgroup -- one chunk of
    // S consecutive Gaussians -- then reads consecutive columns of the block (coalesced); batches arrive in any order.
    int total = 0, before[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w == wave) before[j] = total;
            total += cnt[j][w];
        }
    if (total == 0) return;
    if (tid == 0) base_s = atomicAdd(reinterpret_cast<int*>(block), total);          // header word 0 = K
    __syncthreads();
    float* __restrict__ rows = block + DP_REC;                // row q: rows + q * cap
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = base_s + before[j] + rank[j];
        if (nz[j] && k < cap) {                               // beyond the capacity: dropped, the header still counts it (overflow)
            rows[k] = __int_as_float(gid[j]);
#pragma unroll
            for (int q = 0; q < 9; q++) rows[(size_t)(1 + q) * cap + k] = mom[j][q];
        }
    }
}

LG_API int lg_dp_record_floats(void) { return DP_REC; }

LG_API int lg_dp_compact_moments(const float* packed_grad, const int64_t* vis_ids, const int* vis_num, int A, int S, int cap,
                                 float* block /*[(1 + cap) * 10]*/, const int* hot_of, int* hot_counter, void* stream)
{
    return lg_dp_compact_moments_spec(packed_grad, vis_ids, vis_num, A, S, cap, block, hot_of, hot_counter, nullptr, stream);
}

LG_API int lg_dp_compact_moments_spec(const float* packed_grad, const int64_t* vis_ids, const int* vis_num, int A, int S, int cap,
                                      float* block, const int* hot_of, int* hot_counter, const int* poison /*nullable device int*/, void* stream)
{
    if (A <= 0 || cap <= 0 || (hot_of != nullptr && hot_counter == nullptr)) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(block, 0, sizeof(float) * DP_REC, s);          // the header row
    if (e != hipSuccess) return (int)e;
    const long long N = (long long)A * S;
    hipLaunchKernelGGL(dp_compact_kernel, dim3(lg_cdiv(N, DP_BATCH)), dim3(256), 0, s, (const float4*)packed_grad, vis_ids, vis_num, A, S, cap, block, hot_of, hot_counter, poison);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// Speculative culling across ranks (no reference counterpart; litegs_amd/dp.py "rank-consistent speculation").  With `poison` given, a step
// FAILS when some rank's header carries the failure flag (its culled forward violated a depth bound, or an earlier step failed: the flag is
// sticky) or when some rank's record count outgrew the block capacity.  Every rank derives that verdict from the SAME gathered headers, so
// all replicas raise their sticky poison word at the same step -- from which on no backward + Adam launch changes anything on any rank
// -- and report it in the same per-step status words: status[0] = step number, status[1] = bit r: rank r's flag (its forward failed in this step or it was
// poisoned before), bit 8: a block overflowed -- nothing a rank knows on its own enters the status.  The hosts read a step's status behind that step's event, at a fixed lag, and so
// all start the replay at the same step of their enqueue sequence: the collectives stay matched.
__global__ void __launch_bounds__(256) dp_slotmap_kernel(const float* __restrict__ gathered, int W, int cap, long long total,
                                                         int* __restrict__ slot, int* __restrict__ host_max_k, int* __restrict__ overflow,
                                                         int* __restrict__ poison, int* __restrict__ status_host, int step_id)
{
    const int r = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const float* __restrict__ blk = gathered + (size_t)r * (1 + cap) * DP_REC;
    const int ktrue = __float_as_int(blk[0]);
    const int kr = ktrue < cap ? ktrue : cap;
    if (k < kr) {
        const int gid = __float_as_int(blk[DP_REC + k]);               // row 0 of the block: the global indices
        if (gid >= 0 && gid < total) slot[(size_t)r * total + gid] = k + 1;
    }
    if (r == 0 && k == 0) {
        int mx = 0;
        for (int q = 0; q < W; q++) mx = max(mx, __float_as_int(gathered[(size_t)q * (1 + cap) * DP_REC]));
        if (poison != nullptr) {  // speculative step: an overflow is a failed step (replayed with an exact capacity), not an error
            int flags = 0;
            for (int q = 0; q < W; q++) flags |= (__float_as_int(gathered[(size_t)q * (1 + cap) * DP_REC + 1]) != 0) ? (1 << q) : 0;
            if (mx > cap) flags |= 1 << 8;
            if (flags != 0) __hip_atomic_store(poison, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (host_max_k) {
                __hip_atomic_store(host_max_k, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(host_max_k + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (status_host) {
                __hip_atomic_store(status_host + 1, flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(status_host, step_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        if (host_max_k) {         // word 0: the job's largest count (sizes the slot's next visit); word 1: that count when it outgrew THIS step's capacity
            __hip_atomic_store(host_max_k, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_max_k + 1, mx > cap ? mx : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (mx > cap && overflow) atomicOr(overflow, 1);
    }
}

LG_API int lg_dp_build_slotmap(const float* gathered /*[W][(1 + cap) * 10]*/, int W, int cap, long long total /*chunks * S*/,
                               int* slot /*[W][total], all zero between steps*/, int* host_max_k /*nullable pinned int[2]: {largest count, overflow marker}*/,
                               int* overflow /*nullable device flag, sticky*/, void* stream)
{
    return lg_dp_build_slotmap_spec(gathered, W, cap, total, slot, host_max_k, overflow, nullptr, nullptr, 0, stream);
}

// poison (nullable device int, sticky), status_host (nullable pinned int[2]), step_id: see dp_slotmap_kernel
LG_API int lg_dp_build_slotmap_spec(const float* gathered, int W, int cap, long long total, int* slot, int* host_max_k, int* overflow,
                                    int* poison, int* status_host, int step_id, void* stream)
{
    if (W <= 0 || W > DP_MAX_WORLD || cap <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(dp_slotmap_kernel, dim3(lg_cdiv(cap, 256), W), dim3(256), 0, (hipStream_t)stream, gathered, W, cap, total, slot,
                       host_max_k, overflow, poison, status_host, step_id);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
struct CameraSet { Camera c[DP_MAX_WORLD]; };

template <int DEG>
__global__ void dp_backward_adam_kernel(const int64_t* __restrict__ union_ids, const int* __restrict__ union_count, CameraSet cams, int W,
                                        AdamRates ar, int C, int S, int R, const float* __restrict__ gathered, int cap,
                                        int* __restrict__ slot,
                                        float* __restrict__ pos, float* __restrict__ scale, float* __restrict__ rot,
                                        float* __restrict__ sh0, float* __restrict__ shr, float* __restrict__ opa,
                                        float* __restrict__ m_pos, float* __restrict__ m_scale, float* __restrict__ m_rot,
                                        float* __restrict__ m_sh0, float* __restrict__ m_shr, float* __restrict__ m_opa,
                                        float* __restrict__ v_pos, float* __restrict__ v_scale, float* __restrict__ v_rot,
                                        float* __restrict__ v_sh0, float* __restrict__ v_shr, float* __restrict__ v_opa,
                                        unsigned char* __restrict__ touched,
                                        const int* __restrict__ poison, int* __restrict__ applied_host, int step_id)
{
    const int a = blockIdx.x, t = threadIdx.x;
    if (poison != nullptr) {                 // speculative culling: a failed step (this one or an earlier one) -> nothing is updated, on any rank
        if (*poison != 0) return;
        if (a == 0 && t == 0) __hip_atomic_store(applied_host, step_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (a >= union_count[0]) return;
    constexpr int NB = (DEG + 1) * (DEG + 1);
    const size_t CS = (size_t)C * S;
    const size_t sd = (size_t)union_ids[a] * S + t;
    // exact skip of no-op updates (see project_backward_adam_kernel, fused.hip): no rank sent a record for this Gaussian and it has no
    // Adam history -> gradient 0 on zero moments leaves parameter and moments bit for bit as they are
    if (touched != nullptr && touched[sd] == 0) {
        int any = 0;
        for (int r = 0; r < W; r++) any |= slot[(size_t)r * CS + sd];
        if (any == 0) return;
        touched[sd] = 1;
    }
    const float px = pos[sd], py = pos[CS + sd], pz = pos[2 * CS + sd];
    const float s0 = scale[sd], s1 = scale[CS + sd], s2 = scale[2 * CS + sd];
    const float rw = rot[sd], rx = rot[CS + sd], ry = rot[2 * CS + sd], rz = rot[3 * CS + sd];
    const float oraw = opa[sd];
    float g_pos[3] = { 0.f, 0.f, 0.f }, g_scale[3] = { 0.f, 0.f, 0.f }, g_rot[4] = { 0.f, 0.f, 0.f, 0.f }, g_opa = 0.f;
    float g_sh[NB * 3];
#pragma unroll
    for (int k = 0; k < NB * 3; k++) g_sh[k] = 0.f;
    for (int r = 0; r < W; r++) {                           // rank order: the sum is the same on every replica
        int* __restrict__ sp = slot + (size_t)r * CS + sd;
        const int k1 = *sp;
        if (k1 == 0) continue;
        *sp = 0;                                            // leave the map clean for the next step
        const float* __restrict__ rec = gathered + (size_t)r * (1 + cap) * DP_REC + DP_REC + (k1 - 1);     // rank r's block, column k1 - 1
        float mom[9];
#pragma unroll
        for (int q = 0; q < 9; q++) mom[q] = rec[(size_t)(1 + q) * cap];
        GaussGrads G;
        gaussian_backward<DEG>(cams.c[r], mom, 1.0f, px, py, pz, s0, s1, s2, rw, rx, ry, rz, oraw, G);
#pragma unroll
        for (int k = 0; k < 3; k++) { g_pos[k] += G.pos[k]; g_scale[k] += G.scale[k]; }
#pragma unroll
        for (int k = 0; k < 4; k++) g_rot[k] += G.rot[k];
        g_opa += G.opa;
#pragma unroll
        for (int k = 0; k < NB; k++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) g_sh[k * 3 + ch] += G.basis[k] * G.gc[ch];
    }
    const float inv_w = 1.0f / (float)W;                    // MEAN over the ranks' frames
    {
        const size_t o3[3] = { sd, CS + sd, 2 * CS + sd };
        const size_t o4[4] = { sd, CS + sd, 2 * CS + sd, 3 * CS + sd };
        const size_t o1[1] = { sd };
        const float gp[3] = { g_pos[0] * inv_w, g_pos[1] * inv_w, g_pos[2] * inv_w };
        const float gs[3] = { g_scale[0] * inv_w, g_scale[1] * inv_w, g_scale[2] * inv_w };
        const float gr[4] = { g_rot[0] * inv_w, g_rot[1] * inv_w, g_rot[2] * inv_w, g_rot[3] * inv_w }, go[1] = { g_opa * inv_w };
        const float g0[3] = { g_sh[0] * inv_w, g_sh[1] * inv_w, g_sh[2] * inv_w };
        adam_rows<3>(pos, m_pos, v_pos, o3, gp, ar.lr_pos, ar.b1, ar.b2, ar.eps);
        adam_rows<3>(scale, m_scale, v_scale, o3, gs, ar.lr_scale, ar.b1, ar.b2, ar.eps);
        adam_rows<4>(rot, m_rot, v_rot, o4, gr, ar.lr_rot, ar.b1, ar.b2, ar.eps);
        adam_rows<1>(opa, m_opa, v_opa, o1, go, ar.lr_opa, ar.b1, ar.b2, ar.eps);
        adam_rows<3>(sh0, m_sh0, v_sh0, o3, g0, ar.lr_sh0, ar.b1, ar.b2, ar.eps);
    }
    constexpr int KB = (DEG == 2) ? 4 : 3;
    static_assert(NB == 1 || (NB - 1) % KB == 0, "SH-rest batches must tile the active coefficients");
#pragma unroll
    for (int k0 = 1; k0 < NB; k0 += KB) {
        size_t o9[KB * 3];
        float g9[KB * 3];
#pragma unroll
        for (int kk = 0; kk < KB; kk++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                o9[kk * 3 + ch] = ((size_t)(k0 + kk - 1) * 3 + ch) * CS + sd;
                g9[kk * 3 + ch] = g_sh[(k0 + kk) * 3 + ch] * inv_w;
            }
        adam_rows<KB * 3>(shr, m_shr, v_shr, o9, g9, ar.lr_shr, ar.b1, ar.b2, ar.eps);
    }
    for (int k = NB - 1; k < R; k++)                         // inactive SH degrees: zero gradient, moments still decay (as adamUpdate)
        for (int ch = 0; ch < 3; ch++) adam_row(shr, m_shr, v_shr, ((size_t)k * 3 + ch) * CS + sd, 0.0f, ar.lr_shr, ar.b1, ar.b2, ar.eps);
}

// views_host / projs_host: HOST float[W][16], the cameras of the W ranks' frames of this step, in rank order
LG_API int lg_dp_backward_adam(const int64_t* union_ids, const int* union_count, int chunks, int S, int H, int W_img,
                               const float* views_host, const float* projs_host, int world, int degree, int R,
                               const float* gathered, int cap, int* slot,
                               float* pos, float* scale, float* rot, float* sh0, float* shr, float* opa,
                               float* m_pos, float* m_scale, float* m_rot, float* m_sh0, float* m_shr, float* m_opa,
                               float* v_pos, float* v_scale, float* v_rot, float* v_sh0, float* v_shr, float* v_opa,
                               const float* lr6 /*xyz, sh_0, sh_rest, opacity, scale, rot*/, float b1, float b2, float eps,
                               unsigned char* touched /*nullable, as in lg_fused_backward_adam*/, void* stream)
{
    return lg_dp_backward_adam_spec(union_ids, union_count, chunks, S, H, W_img, views_host, projs_host, world, degree, R, gathered, cap, slot,
                                    pos, scale, rot, sh0, shr, opa, m_pos, m_scale, m_rot, m_sh0, m_shr, m_opa, v_pos, v_scale, v_rot, v_sh0, v_shr, v_opa,
                                    lr6, b1, b2, eps, touched, nullptr, nullptr, 0, stream);
}

// poison (nullable device int) raised -> the launch changes nothing; otherwise it records step_id in the pinned word applied_host
LG_API int lg_dp_backward_adam_spec(const int64_t* union_ids, const int* union_count, int chunks, int S, int H, int W_img,
                                    const float* views_host, const float* projs_host, int world, int degree, int R,
                                    const float* gathered, int cap, int* slot,
                                    float* pos, float* scale, float* rot, float* sh0, float* shr, float* opa,
                                    float* m_pos, float* m_scale, float* m_rot, float* m_sh0, float* m_shr, float* m_opa,
                                    float* v_pos, float* v_scale, float* v_rot, float* v_sh0, float* v_shr, float* v_opa,
                                    const float* lr6, float b1, float b2, float eps, unsigned char* touched,
                                    const int* poison, int* applied_host, int step_id, void* stream)
{
    if (poison != nullptr && applied_host == nullptr) return (int)hipErrorInvalidValue;
    if (world <= 0 || world > DP_MAX_WORLD || chunks <= 0 || S <= 0 || S > 1024) return (int)hipErrorInvalidValue;
    CameraSet cams;
    for (int r = 0; r < DP_MAX_WORLD; r++) {
        const int q = r < world ? r : 0;
        for (int k = 0; k < 16; k++) { cams.c[r].V[k] = views_host[q * 16 + k]; cams.c[r].P[k] = projs_host[q * 16 + k]; }
        cams.c[r].H = H; cams.c[r].W = W_img;
    }
    AdamRates ar = { lr6[0], lr6[1], lr6[2], lr6[3], lr6[4], lr6[5], b1, b2, eps };
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_DP(D) hipLaunchKernelGGL(dp_backward_adam_kernel<D>, dim3(chunks), dim3(S), 0, s, union_ids, union_count, cams, world, ar, chunks, S, R, \
                                        gathered, cap, slot, pos, scale, rot, sh0, shr, opa, m_pos, m_scale, m_rot, m_sh0, m_shr, m_opa,               \
                                        v_pos, v_scale, v_rot, v_sh0, v_shr, v_opa, touched, poison, applied_host, step_id)
    switch (degree) {
    case 0: LAUNCH_DP(0); break;
    case 1: LAUNCH_DP(1); break;
    case 2: LAUNCH_DP(2); break;
    case 3: LAUNCH_DP(3); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef LAUNCH_DP
    LG_RETURN_LAST();
}
