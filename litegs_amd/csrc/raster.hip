// Per-tile front-to-back alpha blending, forward and backward.
//
// CDNA4 design (not a translation of GR/raster.cu):
//  * one wave64 owns one tile (8x16 = 128 px -> 2 px per lane; 16x16 -> 4; 8x8 -> 1); four independent
//    waves per 256-thread workgroup, no LDS and no barriers in the blend loops;
//  * the per-tile splat list and the 64-byte splat records are WAVE-UNIFORM, so the records are fetched through
//    the scalar memory path (one s_load_dwordx16 per splat into SGPRs): zero VGPRs, zero LDS bandwidth, and VALU
//    ops read the splat constants straight from SGPRs.  This is the CDNA-native replacement for the
//    "stage the splat list in shared memory" idiom of warp-32 rasterisers;
//  * measured on gfx950 (tools/ubench, profiles/r02_sq_counters.md): a SIMD issues one plain VALU instruction per 4 cycles
//    (packed fp32 ~5, transcendentals and v_permlane*_swap ~7.5), a single wave at most one instruction of any kind per ~9
//    cycles, and SQ_ACTIVE_INST_VALU of the backward equals its duration: the blend kernels are bound by VALU ISSUE, then by
//    the per-wave instruction latency in the tail (only ~2 tiles per wave slot).  Hence: two pixels per lane as packed 2-vectors;
//    nothing on the VALU that another pipe can do -- splat ids are loaded 64 at a time with one vector load and handed out by
//    v_readlane, the record of the NEXT splat is requested by a scalar load (32-bit scalar offset form) before the current one
//    is blended and waited for once per iteration, the loops are unrolled by two over a ping-pong pair of record registers
//    (no register rotation), the atomic's lane mask is applied by two s_mov of exec (no branch);
//  * fp32 blend (the reference blends in half2 with a x128 transmittance scale; 1e-4 parity needs fp32);
//    the exponent is pre-scaled by log2(e) and log2(opacity) is folded into it at pack time, so alpha is
//    2 FMA + v_exp_f32 per pixel;
//  * wave-level early exit through a 64-bit ballot; heaviest-first tile schedule (below);
//  * backward: reverse traversal; nine per-splat sums reduced across the wave by a transposing butterfly made of
//    v_permlane32/16_swap and bank-masked DPP adds (the ninth value crosses rows through the LDS crossbar) that leaves the 9
//    totals in 9 different lanes, which issue ONE coalesced global_atomic_add_f32 instruction into a 64-byte-aligned
//    gradient record (the reference issues 9 serial atomics from lane 0).
//
// Lane -> pixel map: x = lane % TW; q = lane / TW; strip = q >> 1; p = q & 1; row(k) = strip*2*PPL + 2k + p.
// This gives each lane exactly the pixel set of one reference (thread, half2-lane) pair, which is what
// the statistic-mode err_square running sum (GR/raster.cu:781-783) is defined over.
#include "lg_common.h"
#include "lg_chain.h"
#include "lg_tilewalk.h"
#include "lg_binning_internal.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#define REC 16                 // floats per packed splat record (64 B, one cache line)
#define GREC 16                // floats per packed gradient record
#define LOG2E 1.4426950408889634f

// record layout (dwords): 0 px, 1 py, 2 A2=-0.5*a*log2e, 3 B2=-b*log2e, 4 C2=-0.5*c*log2e, 5 opacity, 6 r, 7 g | 8 b,
//                          9 a=ic00, 10 b=ic01, 11 c=ic11 | 12 depth, 13 ndc.x, 14 ndc.y (fused executor only), 15 log2(opacity)
// alpha = min(255/256, exp2(A2*dx^2 + B2*dx*dy + C2*dy^2 + log2(opacity)))     (power as in GR/raster.cu:237-240)
// The blend kernels read 0-4, 6-8 and 15; the key emission 5, 9-11, 13-14; lg_unpack_gradient 5 and 9-11.
#define R_PX 0
#define R_PY 1
#define R_A2 2
#define R_B2 3
#define R_C2 4
#define R_O 5
#define R_CR 6
#define R_CG 7
#define R_CB 8
#define R_LO 15

// ---------------------------------------------------------------------------------------------
// a12 pack_forward_params (reference: GR/raster.cu:334-356), fp32 colours (no half2 rounding)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_params_kernel(const float* __restrict__ ndc, const float* __restrict__ inv_cov,
                                                          const float* __restrict__ color, const float* __restrict__ opacity,
                                                          const int* __restrict__ valid_length, int N, int H, int W,
                                                          float4* __restrict__ packed)
{
#pragma clang fp contract(off)      // same record bits as the fused executor (fused.hip is built with -ffp-contract=off)
    int i = blockIdx.x * 256 + threadIdx.x;
    int b = blockIdx.y;
    if (i >= lg_valid_len(valid_length, N)) return;
    float px = (ndc[((size_t)b * 4) * N + i] + 1.0f) * 0.5f * W - 0.5f;
    float py = (ndc[((size_t)b * 4 + 1) * N + i] + 1.0f) * 0.5f * H - 0.5f;
    float depth = ndc[((size_t)b * 4 + 2) * N + i];
    float a = inv_cov[((size_t)b * 4) * N + i], bb = inv_cov[((size_t)b * 4 + 1) * N + i], c = inv_cov[((size_t)b * 4 + 3) * N + i];
    float r = color[((size_t)b * 3) * N + i], g = color[((size_t)b * 3 + 1) * N + i], bl = color[((size_t)b * 3 + 2) * N + i];
    float o = opacity[i];
    float4* rec = packed + ((size_t)b * N + i) * (REC / 4);
    rec[0] = make_float4(px, py, -0.5f * a * LOG2E, -bb * LOG2E);
    rec[1] = make_float4(-0.5f * c * LOG2E, o, r, g);
    rec[2] = make_float4(bl, a, bb, c);
    rec[3] = make_float4(depth, 0.0f, 0.0f, lg_log2_opacity(o));
}

LG_API int lg_pack_forward_params(const float* ndc, const float* inv_cov, const float* color, const float* opacity,
                                  const int* valid_length, int V, int N, int H, int W, float* packed, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(ndc, inv_cov, color, opacity, packed);
    hipLaunchKernelGGL(pack_params_kernel, dim3(lg_cdiv(N, 256), V), dim3(256), 0, (hipStream_t)stream,
                       ndc, inv_cov, color, opacity, valid_length, N, H, W, (float4*)packed);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ float xor_dpp1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float xor_dpp2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float xor_swz4(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x101F)); }
__device__ __forceinline__ float xor_swz8(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x201F)); }
__device__ __forceinline__ float xor_swz16(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)); }
__device__ __forceinline__ float xor_32(float v) { return __shfl_xor(v, 32); }

__device__ __forceinline__ float wave_sum(float v)
{
    v += xor_dpp1(v); v += xor_dpp2(v); v += xor_swz4(v); v += xor_swz8(v); v += xor_swz16(v); v += xor_32(v);
    return v;
}
// Sum over the wave, total in lane 63, on the DPP path alone (no LDS crossbar, no bpermute): in-row prefix by row_shr 1, 2, 4, 8 (lane 15
// of every row holds its row total), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.  Six add-with-DPP slots.
#define DPP_ADD(v, ctrl, row_mask, bank_mask) ((v) + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, row_mask, bank_mask, false)))
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = DPP_ADD(v, 0x111, 0xF, 0xF);       // row_shr:1  (lanes shifted in from outside the row contribute the `old` operand: 0)
    v = DPP_ADD(v, 0x112, 0xF, 0xF);       // row_shr:2
    v = DPP_ADD(v, 0x114, 0xF, 0xF);       // row_shr:4
    v = DPP_ADD(v, 0x118, 0xF, 0xF);       // row_shr:8
    v = DPP_ADD(v, 0x142, 0xA, 0xF);       // row_bcast:15 -> rows 1, 3
    v = DPP_ADD(v, 0x143, 0xC, 0xF);       // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m));
    return v;
}

// XCD-aware slot order: hardware places workgroup b on XCD b % 8; give XCD k a contiguous run of blocks.
__device__ __forceinline__ int xcd_remap(int b, int nb)
{
    const int cpx = nb >> 3, rem = nb & 7;
    const int k = b & 7, j = b >> 3;
    return k * cpx + (k < rem ? k : rem) + j;
}
// mode 0: one contiguous band of tiles per XCD (xcd_remap); mode 1: identity; mode C >= 2: XCD k takes runs of C consecutive blocks,
// the eight XCDs interleaved (L2 locality inside a run, the image's work spread over all XCDs); the tail that does not fill a
// group of 8*C blocks keeps the identity map.  All modes are bijections of [0, nb).
__device__ __forceinline__ int block_remap(int b, int nb, int mode)
{
    if (mode == 0) return xcd_remap(b, nb);
    if (mode == 1) return b;
    const int C = mode, k = b & 7, j = b >> 3;
    const int groups = nb / (8 * C);
    if (j >= groups * C) return b;
    return (j / C) * (8 * C) + k * C + (j % C);
}

template <int TH, int TW>
struct TileMap {
    static constexpr int PPL = TH * TW / 64;       // pixels per lane
    static_assert(TH * TW % 64 == 0 && 64 % TW == 0 && (64 / TW) % 2 == 0, "unsupported tile");
    static_assert(TH % (2 * PPL) == 0, "unsupported tile");
};

// The whole 64-byte record of splat `byte_off / 64` into 16 SGPRs: one scalar load, 32-bit scalar offset (records of one view span
// < 4 GiB: N < 2^26, checked by the launchers).  Written as asm because the compiler neither forms the soffset addressing from a
// 64-bit pointer sum nor keeps the request in flight across the loop body; the matching wait is rec_wait().
// FRAGILE BY CONSTRUCTION, and checked by every parity test: the load completes asynchronously, somewhere before the matching wait, while
// the compiler takes the output operand for defined at this statement.  That is sound as long as the register allocator never COPIES the
// block between request and wait.  It does not while every record variable keeps one register block around the loops -- the shape all
// kernels below have (two variables, ping-pong).  Round 6 tried to keep two more 64-bit masks alive across the requests of the lean
// forward (last_contributor on the scalar unit): the allocator gave the four requests of a group four different blocks and moved words of
// in-flight records with s_mov_b32 at the back edge -- a wrong image in every tile, caught by tests/test_gpu_ops.py.  Making the operand
// read-write ("+s") did not pin the block either (and cost an occupancy step).  Reverted; anything that raises the scalar register
// pressure of these loops has to be checked against the ISA (`grep s_load_dwordx16`: two destination blocks per loop).
__device__ __forceinline__ void rec_request(f32x16& rec, const float* __restrict__ pk, unsigned byte_off)
{
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(rec) : "s"(pk), "s"(byte_off));
}
__device__ __forceinline__ void rec_wait(f32x16& rec) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(rec)); }
// byte offset of a splat's 64-byte record.  An id outside 0..N-1 cannot come out of a correct table; if one does (an entry the binning left
// unwritten), it must not become a load / atomic address: clamped (one scalar min per splat; the scalar unit has the slack).
__device__ __forceinline__ unsigned rec_off(int id, int N) { return min((unsigned)id, (unsigned)(N - 1)) << 6; }

// the id of a list position through the scalar path as well (one s_load_dword, two splats ahead of its use): no vector load of the
// list, no v_readlane per splat
__device__ __forceinline__ void id_request(int& id, const int* __restrict__ sp, unsigned byte_off)
{
    asm volatile("s_load_dword %0, %1, %2" : "=s"(id) : "s"(sp), "s"(byte_off));
}
__device__ __forceinline__ void rec_id_wait(f32x16& rec, int& id) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(rec), "+s"(id)); }
// v_min_f32 without the canonicalising v_max the compiler puts in front of fminf when it cannot prove its operand quiet
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// L2 warm-up of the scalar path (round 6).  A scalar load can only be waited for with lgkmcnt(0) (scalar loads return out of order), so
// the record of splat p + 1 is requested while splat p is blended and waited for at the end of that splat: ONE splat of latency cover,
// whatever the register budget.  That is enough while the records sit in L2 (the fresh cloud: ~80 splats per tile) and not when they
// do not: in the training state (1 400 instances per tile, 138 MB of records in per-tile depth order) SQ counters show the blend
// forward parked at s_waitcnt for 45 % of its wave cycles with the vector pipes active in 41 % of the launch, the backward 24 % / 57 %
// (profiles/r06_sq_training_state_before_prefetch.md).  The vector memory path has in-order counters and nothing else to do in these
// kernels: every `pfb` list positions, `pfb` lanes load the splat ids two blocks ahead (one coalesced line of the list) and touch
// one dword of each record one block ahead (a gather of `pfb` lines), with the ids that arrived a block earlier.  Nothing is done
// with the data; the lines are in L2 when the scalar loads ask for them.  Lists shorter than 2 * pfb + 64 positions skip it.
struct PfState { int ids; unsigned g; };
__device__ __forceinline__ unsigned pf_touch(const float* __restrict__ pk, int id, int N)
{
    return reinterpret_cast<const unsigned*>(pk)[(size_t)(min((unsigned)id, (unsigned)(N - 1))) << 4];
}
// consume the previous block's gather (it was issued a block ago: no stall) so that its register is free again
__device__ __forceinline__ void pf_retire(unsigned g) { asm volatile("" : : "v"(g)); }

static int g_bwd_map = 0, g_fwd_map = 0, g_use_order = 1, g_bwd_fast = 1, g_fwd_fast = 1;
// Measurement aid (tools/wave_clock.py): when set, every wave of the lean blend forward / the fast blend backward leaves
// {start, end (s_memrealtime: 100 MHz), list length << 32 | tile, XCC_ID << 32 | HW_ID} in its slot's four words -- the occupancy of every
// SIMD over the launch, i.e. how much of it is tail.  NULL (default): one scalar compare per wave.
static long long* g_wclk_fwd = nullptr;
static long long* g_wclk_bwd = nullptr;
LG_API int lg_debug_wave_clock(void* fwd_buf /*nullable device int64[slots][4]*/, void* bwd_buf) { g_wclk_fwd = (long long*)fwd_buf; g_wclk_bwd = (long long*)bwd_buf; return 0; }
__device__ __forceinline__ void wclk_store(long long* __restrict__ w, int slot, long long t0, int n, int tile, int lane)
{
    if (w == nullptr || lane != 0) return;
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    w[(size_t)slot * 4 + 0] = t0; w[(size_t)slot * 4 + 1] = t1;
    w[(size_t)slot * 4 + 2] = ((long long)n << 32) | (unsigned)tile; w[(size_t)slot * 4 + 3] = ((long long)xcc << 32) | hw;
}
// Segmented blend backward (round 6; VERDICT round 5 item 4, "transmittance checkpoints"): the lean forward leaves a checkpoint of every
// pixel's {T, C} each 2^g_seg_shift list positions, the backward runs one wave per (tile, segment) -- see LG_UNIT_CLASSES.  Parity-green
// (tests/test_gpu_trained_cloud.py) and OFF by default: the blend backward gets 4-6 % shorter (1170 -> 1102-1125 us in the training state;
// the probe without checkpoint traffic promised 7-12 %), the forward pays 21-28 us for it (stores and scalar register pressure in the
// loop, one returning atomic per wave, the counters' fill launch outside the global route), the step gains 2-20 us of 3.3 ms
// (profiles/r06_bwd_segments_ab.log).
// lg_set_tuning(22, 0 | 1) / (23, log2 of the segment length).
static int g_bwd_segments = 0;
static int g_seg_shift = 9;
static int g_blend_lds_fwd = 0, g_blend_lds_bwd = 0;    // KB of (unused) dynamic LDS per workgroup: caps the resident workgroups per CU (160 KB / value); lg_set_tuning(19 / 20, KB)
static int g_fwd_lean = 1;        // 1: renders without statistics / depth bounds / gates take the lean blend forward (lg_set_tuning(17, 0 | 1))
static int g_bwd_probe = 0;       // measurement hook (wrong gradients!): 1 = the blend backward's atomics are issued with an empty lane mask (lg_set_tuning(18, .))
static int g_pf_block = 16;       // list positions per L2 warm-up block of the fast blend kernels (0: off; lg_set_tuning(16, 0 | 8 | 16 | 32 | 64))       // launch variants (lg_set_tuning: A/B hooks of tools/ and tests/, plain ints)
static int g_rank_prio = 0;       // 1: waves of the heaviest tiles of a heavy-first schedule raise their issue priority (wave_rank_priority)

// Tail of the blend launches (profiles/r04_blend_critical_path.log): the heaviest tile of a late-phase frame walks 1127 splats; alone on
// its SIMD it needs 142 us (forward) / 403 us (backward), but scheduled first -- as the heavy-first order does -- it SHARES its SIMD with
// seven average waves for most of the launch and runs at an eighth of the issue rate, so it is still running when the machine has
// drained: the launch ends with a few long tiles at the lone-wave rate and ~20-35 % of it is tail.  The hardware's answer is the wave
// priority (s_setprio, 0..3: the SIMD's arbiter issues from the highest-priority ready wave): the first eighth of a heavy-first schedule
// runs at priority 3, the next eighth at 2, the next quarter at 1.  The long tiles then finish early at nearly their lone rate, and the
// light tiles, which have slack, fill the issue slots they leave.  Nothing about any wave's arithmetic changes.
// (prio_mode travels in bits 8.. of the map_mode launch argument.)
__device__ __forceinline__ void wave_rank_priority(int prio_mode, int slot, int nslots, bool heavy_first)
{
    if (prio_mode == 0 || !heavy_first) return;
    if (slot * 8 < nslots) __builtin_amdgcn_s_setprio(3);
    else if (slot * 4 < nslots) __builtin_amdgcn_s_setprio(2);
    else if (slot * 2 < nslots) __builtin_amdgcn_s_setprio(1);
}

// ---------------------------------------------------------------------------------------------
// a13 rasterize_forward (reference: GR/raster.cu:162-332)
// ---------------------------------------------------------------------------------------------
template <int PPL>
struct FwdState {
    float X, Y[PPL], T[PPL], Cr[PPL], Cg[PPL], Cb[PPL];
    int lc[PPL];
};

// one splat blended into the lane's pixels; i1 = (list position + 1).  Returns false when no pixel of the wave is active any more
// (checked BEFORE blending, as the reference does).
template <int PPL, bool STAT>
__device__ __forceinline__ bool fwd_splat(FwdState<PPL>& st, const f32x16& rec, int i1, int lane, unsigned pid_off,
                                          int* __restrict__ frag_count, float* __restrict__ frag_weight)
{
    bool any_act = false;
#pragma unroll
    for (int k = 0; k < PPL; k++) any_act |= (st.T[k] > 1.0f / 8192);
    if (!__any(any_act)) return false;
    const float dx = rec[R_PX] - st.X;
    const float t1 = rec[R_B2] * dx;
    const float t0 = __builtin_fmaf(rec[R_A2] * dx, dx, rec[R_LO]);
    int fc = 0;
    float ws = 0.0f;
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const bool active = st.T[k] > 1.0f / 8192;
        const float dy = rec[R_PY] - st.Y[k];
        const float E = __builtin_amdgcn_exp2f(__builtin_fmaf(dy, __builtin_fmaf(rec[R_C2], dy, t1), t0));
        const bool valid = active && (E >= 1.0f / 256);
        st.lc[k] = active ? i1 : st.lc[k];               // == number of splats visited while active (activity is monotone)
        const float alpha = valid ? fminf(255.0f / 256, E) : 0.0f;
        const float w = st.T[k] * alpha;
        if (STAT) { fc += valid ? 1 : 0; ws += w; }
        st.Cr[k] = __builtin_fmaf(rec[R_CR], w, st.Cr[k]);
        st.Cg[k] = __builtin_fmaf(rec[R_CG], w, st.Cg[k]);
        st.Cb[k] = __builtin_fmaf(rec[R_CB], w, st.Cb[k]);
        st.T[k] -= w;                       // T*(1-alpha)
    }
    if (STAT) {
        unsigned long long m = __ballot(fc != 0);
        if (m) {
            int fct = fc;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) fct += __shfl_xor(fct, s);
            float wst = wave_sum(ws);
            if (lane == 0) {
                atomicAdd(&frag_count[pid_off >> 6], fct);
                unsafeAtomicAdd(&frag_weight[pid_off >> 6], wst);
            }
        }
    }
    return true;
}

// the blend loop of the generic kernel (any tile shape, statistics): ids 64 at a time in a VGPR, handed out by v_readlane
template <int PPL, bool STAT>
__device__ __forceinline__ void fwd_generic_loop(FwdState<PPL>& st, const int* __restrict__ sp, const float* __restrict__ pk, int n, int lane,
                                                 int* __restrict__ frag_count, float* __restrict__ frag_weight, int& visited, bool& live, int N)
{
    // ids of the list, 64 at a time: lane l of `nxt` holds the id at position c0 + l + 1, i.e. the splat to REQUEST while
    // position c0 + l is blended (clamped at the list end: the surplus request is never used)
    unsigned off_a = rec_off(rfl(sp[0]), N), off_b = 0;
    f32x16 ra, rb;
    rec_request(ra, pk, off_a);
    int nxt = sp[min(lane + 1, n - 1)];
    rec_wait(ra);
    for (int c0 = 0; c0 < n && live; c0 += 64) {
        const int cnt = min(64, n - c0);                            // positions c0 .. c0 + cnt - 1 in this chunk
        const int nxt_next = sp[min(c0 + 64 + lane + 1, n - 1)];    // next chunk's ids, in flight while this chunk is blended
        for (int j = 0; j < cnt; j += 2) {                          // cnt is even except possibly in the last chunk
            off_b = rec_off(__builtin_amdgcn_readlane(nxt, j), N);
            rec_request(rb, pk, off_b);
            live = fwd_splat<PPL, STAT>(st, ra, c0 + j + 1, lane, off_a, frag_count, frag_weight);
            rec_wait(rb);
            if (!live) break;
            visited = c0 + j + 1;
            if (j + 1 >= cnt) break;                                // odd tail: the list ends here
            off_a = rec_off(__builtin_amdgcn_readlane(nxt, j + 1), N);
            rec_request(ra, pk, off_a);
            live = fwd_splat<PPL, STAT>(st, rb, c0 + j + 2, lane, off_b, frag_count, frag_weight);
            rec_wait(ra);
            if (!live) break;
            visited = c0 + j + 2;
        }
        nxt = nxt_next;
    }
}

// The default 8x16 tile without statistics, forward: the lane's two pixels as packed 2-vectors (exponent, weight, colour accumulation,
// transmittance: v_pk_* do two pixels per issue slot), splat ids through the scalar path, last_contributor as an add-with-carry of the
// activity mask, and a splat that no pixel takes (alpha < 1/256 everywhere) leaves after the alpha evaluation -- with alpha = 0 the
// colour sums and the transmittance keep their values.  36 -> 24 VALU instructions per (tile, splat).
struct FwdFast {
    float X;
    v2f Y, T, Cr, Cg, Cb;
    int lc0, lc1;
};
// lc += (lane's bit of mask): one add-with-carry, the carry-in being the activity mask itself
__device__ __forceinline__ void add_mask_bit(int& lc, unsigned long long mask)
{
    unsigned long long carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %0, 0, %2" : "+v"(lc), "=s"(carry_out) : "s"(mask));
}
// lanes of mask: min(a, b); other lanes: 0 -- the mask is the SGPR pair a ballot left, used as it is
__device__ __forceinline__ float min_where(float a, float b, unsigned long long mask)
{
    float r;
    asm("v_min_f32 %0, %1, %2\n\tv_cndmask_b32_e64 %0, 0, %0, %3" : "=&v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
// STAT (statistic epochs, GR/raster.cu:283-302): fragment count and weight sum of the splat.  The count is the popcount of the two validity
// masks (scalar unit), the weight sum one DPP reduction; one lane issues the two atomics.
template <bool STAT>
__device__ __forceinline__ bool fwd_splat_fast(FwdFast& st, const f32x16& rec, unsigned id, int lane, int* __restrict__ frag_count, float* __restrict__ frag_weight)
{
    const unsigned long long act0 = __builtin_amdgcn_ballot_w64(st.T.x > 1.0f / 8192), act1 = __builtin_amdgcn_ballot_w64(st.T.y > 1.0f / 8192);
    if ((act0 | act1) == 0ull) return false;         // checked BEFORE blending, as the reference does
    const float dx = rec[R_PX] - st.X;
    const float t1 = rec[R_B2] * dx;
    const float t0 = __builtin_fmaf(rec[R_A2] * dx, dx, rec[R_LO]);
    const v2f dyv = rec[R_PY] - st.Y;
    const v2f qv = dyv * (rec[R_C2] * dyv + t1) + t0;
    const float E0 = __builtin_amdgcn_exp2f(qv.x);
    const float E1 = __builtin_amdgcn_exp2f(qv.y);
    add_mask_bit(st.lc0, act0);                      // == number of splats visited while active (activity is monotone)
    add_mask_bit(st.lc1, act1);
    const unsigned long long val0 = act0 & __builtin_amdgcn_ballot_w64(E0 >= 1.0f / 256), val1 = act1 & __builtin_amdgcn_ballot_w64(E1 >= 1.0f / 256);
    if ((val0 | val1) == 0ull) return true;          // nobody takes this splat: alpha = 0 everywhere
    const float amax = 255.0f / 256;
    const v2f alpha = { min_where(E0, amax, val0), min_where(E1, amax, val1) };
    const v2f w = st.T * alpha;
    st.Cr = rec[R_CR] * w + st.Cr;
    st.Cg = rec[R_CG] * w + st.Cg;
    st.Cb = rec[R_CB] * w + st.Cb;
    st.T = st.T - w;                                 // T * (1 - alpha)
    if constexpr (STAT) {
        const int fct = __popcll(val0) + __popcll(val1);           // s_bcnt1: no vector instruction
        const float wst = wave_sum_to_lane63(w.x + w.y);
        if (lane == 63) {
            atomicAdd(&frag_count[id], fct);
            unsafeAtomicAdd(&frag_weight[id], wst);
        }
    }
    return true;
}

#define RF_ARGS const int* __restrict__ sorted_points, const int* __restrict__ start_index,                                              \
                const float* __restrict__ packed, const int* __restrict__ tiles, int K,                                                \
                float* __restrict__ img, float* __restrict__ trans, short* __restrict__ last,                                          \
                int* __restrict__ frag_count, float* __restrict__ frag_weight,                                                         \
                const int* __restrict__ order, int* __restrict__ tile_work,                                                            \
                const int* __restrict__ sched_in, int* __restrict__ sched_out, int zb_check,                                           \
                int* __restrict__ fail_flag, int* __restrict__ fail_host, const int* __restrict__ gate,                                \
                int gx, int ntiles, long long L, int N, int Hp, int Wp, int nslots, int map_mode, int fast
#define RF_PASS sorted_points, start_index, packed, tiles, K, img, trans, last, frag_count, frag_weight, order, tile_work, sched_in, sched_out, zb_check, \
                fail_flag, fail_host, gate, gx, ntiles, L, N, Hp, Wp, nslots, map_mode, fast
template <int TH, int TW, bool STAT>
__device__ __forceinline__ void raster_forward_body(RF_ARGS)
{
    if (gate != nullptr && *gate == 0) return;              // fallback launch of the depth-bound culling that is not needed
    // speculative executor (fused.hip): a failure raised earlier in this step (a truncated table) reaches the host mirror here
    if (fail_host != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && *fail_flag != 0)
        __hip_atomic_store(fail_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    constexpr int PPL = TileMap<TH, TW>::PPL;
    const int lane = threadIdx.x & 63;
    const int view = blockIdx.y;
    const int nb = gridDim.x;
    const int prio_mode = map_mode >> 8;
    map_mode &= 0xff;
    int blk = (tiles == nullptr && order == nullptr) ? block_remap(blockIdx.x, nb, map_mode) : (int)blockIdx.x;
    const int slot = rfl(blk * 4 + (int)(threadIdx.x >> 6));
    if (slot >= nslots) return;
    wave_rank_priority(prio_mode, slot, nslots, tiles != nullptr || order != nullptr);
    int tile = (tiles != nullptr) ? tiles[(size_t)view * K + slot] : (order != nullptr ? order[(size_t)view * ntiles + slot] : slot + 1);
    tile = rfl(tile);
    if (tile <= 0 || tile > ntiles) return;
    const int* __restrict__ si = start_index + (size_t)view * (ntiles + 2);
    const int start = rfl(si[tile]);
    const int end = rfl(si[tile + 1]);
    const float* __restrict__ pk = packed + (size_t)view * N * REC;
    if (STAT) { frag_count += (size_t)view * N; frag_weight += (size_t)view * N; }

    const int tx = (tile - 1) % gx, ty = (tile - 1) / gx;
    const int x = tx * TW + lane % TW;
    const int q = lane / TW;
    const int y0 = ty * TH + (q >> 1) * (2 * PPL) + (q & 1);
    FwdState<PPL> st;
    st.X = (float)x;
#pragma unroll
    for (int k = 0; k < PPL; k++) { st.Y[k] = (float)(y0 + 2 * k); st.T[k] = 1.0f; st.Cr[k] = st.Cg[k] = st.Cb[k] = 0.0f; st.lc[k] = 0; }

    int visited = 0;
    bool live = true;
    const int n = (start >= 0 && end > start) ? end - start : 0;
    const int* __restrict__ sp = sorted_points + (size_t)view * L + (start >= 0 ? start : 0);
    if constexpr (PPL == 2) {
        if (n > 0 && (fast & 0xff)) {
            FwdFast f;
            f.X = st.X; f.Y = v2f{ st.Y[0], st.Y[1] }; f.T = v2f{ 1.0f, 1.0f };
            f.Cr = f.Cg = f.Cb = v2f{ 0.0f, 0.0f };
            f.lc0 = f.lc1 = 0;
            // position p's record is requested while position p - 1 is blended, its id one step earlier (clamped at the list end: the
            // surplus requests are never used)
            int id_a, id_b;
            f32x16 ra, rb;
            id_request(id_a, sp, 0u);
            id_request(id_b, sp, (unsigned)min(1, n - 1) << 2);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(id_a), "+s"(id_b));
            rec_request(ra, pk, rec_off(id_a, N));
            rec_wait(ra);
            const int pfb = fast >> 8;                                         // L2 warm-up block (PfState above); 0: off
            const bool pf_on = pfb > 0 && n >= 2 * pfb + 64;
            PfState pf = { 0, 0u };
            if (pf_on && lane < pfb) {
                pf.g = pf_touch(pk, sp[min(pfb + lane, n - 1)], N);          // block 1 (one dependent round trip before the walk)
                pf.ids = sp[min(2 * pfb + lane, n - 1)];                       // ids of block 2, left in flight
            }
            for (int pos = 0; pos < n; pos += 2) {                            // `ra` holds position pos
                if (pf_on && pos > 0 && (pos & (pfb - 1)) == 0) {             // uniform: start of a block
                    pf_retire(pf.g);
                    if (lane < pfb) {
                        pf.g = pf_touch(pk, pf.ids, N);                        // records of the next block
                        pf.ids = sp[min(pos + 2 * pfb + lane, n - 1)];         // ids of the block after it
                    }
                }
                const unsigned cur_a = rec_off(id_a, N) >> 6, cur_b = rec_off(id_b, N) >> 6;  // splat ids of `ra` / `rb` (statistics)
                rec_request(rb, pk, rec_off(id_b, N));
                id_request(id_a, sp, (unsigned)min(pos + 2, n - 1) << 2);
                live = fwd_splat_fast<STAT>(f, ra, cur_a, lane, frag_count, frag_weight);
                rec_id_wait(rb, id_a);
                if (!live) break;
                visited = pos + 1;
                if (pos + 1 >= n) break;                                       // odd tail: the list ends here
                rec_request(ra, pk, rec_off(id_a, N));
                id_request(id_b, sp, (unsigned)min(pos + 3, n - 1) << 2);
                live = fwd_splat_fast<STAT>(f, rb, cur_b, lane, frag_count, frag_weight);
                rec_id_wait(ra, id_b);
                if (!live) break;
                visited = pos + 2;
            }
            pf_retire(pf.g);
            st.T[0] = f.T.x; st.T[1] = f.T.y;
            st.Cr[0] = f.Cr.x; st.Cr[1] = f.Cr.y; st.Cg[0] = f.Cg.x; st.Cg[1] = f.Cg.y; st.Cb[0] = f.Cb.x; st.Cb[1] = f.Cb.y;
            st.lc[0] = f.lc0; st.lc[1] = f.lc1;
        } else if (n > 0) {
            fwd_generic_loop<PPL, STAT>(st, sp, pk, n, lane, frag_count, frag_weight, visited, live, N);
        }
    } else if (n > 0) {
        fwd_generic_loop<PPL, STAT>(st, sp, pk, n, lane, frag_count, frag_weight, visited, live, N);
    }
    // work done for this tile (splats walked before every pixel saturated): the schedule key of the backward and of the next visit
    if (tile_work != nullptr && lane == 0) tile_work[(size_t)view * (ntiles + 1) + tile] = visited;
    // Depth bounds of this visit (single view; "bounds" block of lg_tilewalk.h).
    // Depth-bound culling (fused.hip): level 0 of sched_in holds, per tile, the view depth beyond which this frame's previous visit
    // predicted the tile to be saturated; splats deeper than the bound of EVERY tile of their rectangle were not emitted.  The list
    // walked here is then complete up to the bound, so the result is exact if the tile saturated at or before it (or if nothing was
    // culled for it); otherwise the fail flag makes the executor's gated fallback re-run the binning without culling.  The new bound
    // is the depth `margin` % (default 50, at least 16 splats) further down the list than where the tile saturated.
    if (sched_out != nullptr) {
        const int gy = ntiles / gx;
        const float INF = __builtin_inff();
        const int margin_pct = (zb_check >> 8) > 0 ? (zb_check >> 8) : 50;        // bits 8..: how far beyond the saturation point the new bound lies
        zb_check &= 1;
        const float zused = (zb_check && sched_in != nullptr) ? reinterpret_cast<const float*>(sched_in)[lg_sched_level_offset(gx, gy, 0) + tile - 1] : INF;
        bool sat = !live;
        if (live) {                                          // the list ran out: saturated exactly at its end?
            bool any_act = false;
#pragma unroll
            for (int k = 0; k < PPL; k++) any_act |= (st.T[k] > 1.0f / 8192);
            sat = !__any(any_act);
        }
        float znew = INF;
        bool ok = !(zused < INF);
        if (sat && visited > 0) {
            const float stop_z = pk[(size_t)(rec_off(rfl(sp[visited - 1]), N) >> 6) * REC + 12];
            ok = stop_z <= zused;
            const int p = visited - 1 + max(16, (int)(((long long)visited * margin_pct) / 100));
            if (p <= n - 1) znew = pk[(size_t)(rec_off(rfl(sp[p]), N) >> 6) * REC + 12];
            else if (zused < INF) znew = fmaxf(zused, pk[(size_t)(rec_off(rfl(sp[n - 1]), N) >> 6) * REC + 12]) * 1.25f;
        }
        if (lane == 0) {
            reinterpret_cast<float*>(sched_out)[lg_sched_level_offset(gx, gy, 0) + tile - 1] = znew;
            const int tx0 = (tile - 1) % gx, ty0 = (tile - 1) / gx;
#pragma unroll
            for (int k = 1; k < LG_PYR_LEVELS; k++)             // positive floats order like their bit patterns
                atomicMax(reinterpret_cast<unsigned int*>(sched_out) + lg_sched_level_offset(gx, gy, k) + (ty0 >> k) * lg_pyr_w(gx, k) + (tx0 >> k),
                          __float_as_uint(znew));
            if (!ok && fail_flag != nullptr) {
                atomicOr(fail_flag, 1);
                if (fail_host != nullptr) __hip_atomic_store(fail_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    const size_t plane = (size_t)Hp * Wp;
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const size_t o = (size_t)(y0 + 2 * k) * Wp + x;
        img[((size_t)view * 3) * plane + o] = fminf(st.Cr[k], 1.0f);
        img[((size_t)view * 3 + 1) * plane + o] = fminf(st.Cg[k], 1.0f);
        img[((size_t)view * 3 + 2) * plane + o] = fminf(st.Cb[k], 1.0f);
        trans[(size_t)view * plane + o] = st.T[k];
        last[(size_t)view * plane + o] = (short)st.lc[k];
    }
}

template <int TH, int TW, bool STAT>
__global__ void __launch_bounds__(256) raster_forward_kernel(RF_ARGS) { raster_forward_body<TH, TW, STAT>(RF_PASS); }
#undef RF_ARGS
#undef RF_PASS

// ---------------------------------------------------------------------------------------------
// a13, lean form (round 6): the 8x16 blend forward without statistics, depth bounds or gates, with the SCALAR instruction count cut.
//
// What bounds the packed loop of raster_forward_kernel in the training state is not the vector pipe: per (tile, splat) it issues ~21
// vector and ~35 scalar-type instructions (scalar ALU, scalar memory, branches: profiles/r06_forward_isa_mix.md), a SIMD takes one
// instruction of each type per 4-cycle issue slot, so the scalar side needs ~140 cycles per splat against ~95 for the vector side --
// and the launch runs at 157 cycles per splat and SIMD whether 4 or 8 waves share it (profiles/r05_blend_tail_priority_ab.log), with the
// vector pipes busy in 44 % of it (SQ_INSTS_VALU x 4.4 clocks, profiles/r06_sq_training_state_before_prefetch.md).  Batching the
// scalar loads two or three records per wait made it slower (more scalar work: profiles/r06_forward_lean_ab.log).  This kernel:
//   * list ids four at a time (one s_load_dwordx4 per four splats instead of one s_load_dword per splat, no per-splat index clamp);
//   * groups of four splats per loop trip: one loop test, one warm-up test, one saturation test per group instead of per splat -- a
//     splat blended after every pixel of the tile has stopped changes nothing (its validity masks are empty), and the tile's work
//     count is the largest last_contributor, taken once at the end;
//   * no "nobody takes this splat" branch: 97.5 % of the walked entries of a trained cloud contribute; alpha = 0 blends to the same bits.
// ~9 scalar-type instructions per splat.  Same arithmetic in the same order as fwd_splat_fast: identical images, T and last_contributor.
// ---------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ids4_request(i32x4& q, const int* __restrict__ sp, unsigned byte_off)
{
    asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(q) : "s"(sp), "s"(byte_off));
}

// Work units of the segmented blend backward: (tile | segment << 16) words, a segment = 2^shift list positions of a tile's walked range
// (the deepest one shorter), dispatched LONGEST FIRST: all full segments, then the remainders by length class.  Why
// (profiles/r06_bwd_segments_probe.log): the issue arbiter serves a SIMD's oldest wave first, so a launch ends with every SIMD's youngest
// waves finishing alone at a third of the SIMD's rate; with whole tiles the last units dispatched are 150-300 entries long (a 1080p frame
// of a trained cloud has no short tiles), with remainders sorted by length they are a few entries long and the drain goes away: 954 ->
// 836 us and 1161 -> 1078 us on two frames with units of 512.  No sort: every forward wave appends its tile's units to the region of their
// class with one returning atomic (unit_counts: [0] full segments, [1 + c] class c = remainders of (len - length) * 16 / len, zeroed before
// the launch), and a backward wave finds its unit from its slot and the 17 counts (first a prefix over them, then one load).  A sorted list
// built by a kernel of its own between the two launches cost 31 us (one workgroup: 16 000 tiles on one compute unit).
#define LG_UNIT_CLASSES 16
// checkpoint record: 64 lanes x {T.x T.y Cr.x Cr.y | Cg.x Cg.y Cb.x Cb.y}, 2 KB per (tile, boundary); slot (start >> shift) + boundary
// index -- distinct for all boundaries of all tiles (a tile's boundaries lie inside its own list range, a range apart from each other)
__device__ __forceinline__ void ckpt_store(float* __restrict__ ckpt, size_t slot, int lane, const FwdFast& f)
{
    float4* o = reinterpret_cast<float4*>(ckpt + (slot * 64 + lane) * 8);
    o[0] = float4{ f.T.x, f.T.y, f.Cr.x, f.Cr.y };
    o[1] = float4{ f.Cg.x, f.Cg.y, f.Cb.x, f.Cb.y };
}

__device__ __forceinline__ void fwd_splat_lean(FwdFast& st, const f32x16& rec, unsigned long long& act0, unsigned long long& act1)
{
    act0 = __builtin_amdgcn_ballot_w64(st.T.x > 1.0f / 8192); act1 = __builtin_amdgcn_ballot_w64(st.T.y > 1.0f / 8192);
    const float dx = rec[R_PX] - st.X;
    const float t1 = rec[R_B2] * dx;
    const float t0 = __builtin_fmaf(rec[R_A2] * dx, dx, rec[R_LO]);
    const v2f dyv = rec[R_PY] - st.Y;
    const v2f qv = dyv * (rec[R_C2] * dyv + t1) + t0;
    const float E0 = __builtin_amdgcn_exp2f(qv.x);
    const float E1 = __builtin_amdgcn_exp2f(qv.y);
    add_mask_bit(st.lc0, act0);
    add_mask_bit(st.lc1, act1);
    const unsigned long long val0 = act0 & __builtin_amdgcn_ballot_w64(E0 >= 1.0f / 256), val1 = act1 & __builtin_amdgcn_ballot_w64(E1 >= 1.0f / 256);
    const float amax = 255.0f / 256;
    const v2f alpha = { min_where(E0, amax, val0), min_where(E1, amax, val1) };
    const v2f w = st.T * alpha;
    st.Cr = rec[R_CR] * w + st.Cr;
    st.Cg = rec[R_CG] * w + st.Cg;
    st.Cb = rec[R_CB] * w + st.Cb;
    st.T = st.T - w;
}

#define LEAN_ARGS const int* __restrict__ sorted_points, const int* __restrict__ start_index,                                             \
                  const float* __restrict__ packed, const int* __restrict__ tiles, int K,                                               \
                  float* __restrict__ img, float* __restrict__ trans, short* __restrict__ last,                                         \
                  const int* __restrict__ order, int* __restrict__ tile_work,                                                           \
                  int gx, int ntiles, long long L, int N, int Hp, int Wp, int nslots, int map_mode, int pfb, long long* __restrict__ wclk, \
                  char* __restrict__ segbase /*nullable: the frame's segment buffers (LgSegLayout); their shift rides in bits 16-23 of pfb*/
#define LEAN_PASS sorted_points, start_index, packed, tiles, K, img, trans, last, order, tile_work, gx, ntiles, L, N, Hp, Wp, nslots, map_mode, pfb, wclk, segbase
__device__ __forceinline__ void raster_forward_lean_body(LEAN_ARGS)
{
    const int blk = (tiles == nullptr && order == nullptr) ? block_remap(blockIdx.x, gridDim.x, map_mode & 0xff) : (int)blockIdx.x;
    const int slot = rfl(blk * 4 + (int)(threadIdx.x >> 6));
    if (slot >= nslots) return;
    constexpr int TH = 8, TW = 16;
    const long long wclk_t0 = wclk != nullptr ? __builtin_amdgcn_s_memrealtime() : 0;
    const int lane = threadIdx.x & 63;
    const int view = blockIdx.y;
    int tile = (tiles != nullptr) ? tiles[(size_t)view * K + slot] : (order != nullptr ? order[(size_t)view * ntiles + slot] : slot + 1);
    tile = rfl(tile);
    if (tile <= 0 || tile > ntiles) return;
    const int* __restrict__ si = start_index + (size_t)view * (ntiles + 2);
    const int start = rfl(si[tile]);
    const int end = rfl(si[tile + 1]);
    const float* __restrict__ pk = packed + (size_t)view * N * REC;
    const int tx = (tile - 1) % gx, ty = (tile - 1) / gx;
    const int x = tx * TW + lane % TW;
    const int q = lane / TW;
    const int y0 = ty * TH + (q >> 1) * 4 + (q & 1);
    FwdFast f;
    f.X = (float)x; f.Y = v2f{ (float)y0, (float)(y0 + 2) }; f.T = v2f{ 1.0f, 1.0f };
    f.Cr = f.Cg = f.Cb = v2f{ 0.0f, 0.0f };
    f.lc0 = f.lc1 = 0;
    const int n = (start >= 0 && end > start) ? end - start : 0;
    const int* __restrict__ sp = sorted_points + (size_t)view * L + (start >= 0 ? start : 0);
    unsigned long long act0 = ~0ull, act1 = ~0ull;
    const int ng = n >> 2;                               // full groups of four list positions
    const unsigned idmask = (pfb & 0x100) ? 0x3ffu : 0xffffffffu;      // measurement hook (wrong image): every tile reads the same 1024 records (always cached)
    const int seg_shift = (pfb >> 16) & 0xff;
    pfb &= 0xff;
#define rec_off(id_, N_) rec_off((int)((unsigned)(id_) & idmask), N_)
    f32x16 ra, rb;                                       // (one pair for the group loop and the tail: two destination blocks in all)
    if (ng > 0) {
        i32x4 qa, qb;                                    // ids of the current group / of the next one
        ids4_request(qa, sp, 0u);
        ids4_request(qb, sp, (unsigned)min(1, ng - 1) << 4);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(qa), "+s"(qb));
        rec_request(ra, pk, rec_off(qa[0], N));
        rec_wait(ra);
        const bool pf_on = pfb > 0 && n >= 2 * pfb + 64;
        PfState pf = { 0, 0u };
        int pf_next = pfb;                               // first position of the next block to warm
        if (pf_on && lane < pfb) {
            pf.g = pf_touch(pk, sp[min(pfb + lane, n - 1)], N);
            pf.ids = sp[min(2 * pfb + lane, n - 1)];
        }
        // one group: `ra` holds the record of its first position, qcur its ids, qnxt the ids of the next group (requested a group ago);
        // on exit `ra` holds the first record of the next group and qcur the ids of the group after it (in flight until the last wait)
#define LEAN_GROUP(qcur, qnxt, g_)                                                                                           \
        {                                                                                                                   \
            if (pf_on && (g_) * 4 >= pf_next) {                                                                             \
                pf_retire(pf.g);                                                                                            \
                if (lane < pfb) {                                                                                           \
                    pf.g = pf_touch(pk, pf.ids, N);                                                                         \
                    pf.ids = sp[min(pf_next + 2 * pfb + lane, n - 1)];                                                      \
                }                                                                                                           \
                pf_next += pfb;                                                                                             \
            }                                                                                                               \
            rec_request(rb, pk, rec_off(qcur[1], N)); fwd_splat_lean(f, ra, act0, act1); rec_wait(rb);                      \
            rec_request(ra, pk, rec_off(qcur[2], N)); fwd_splat_lean(f, rb, act0, act1); rec_wait(ra);                      \
            rec_request(rb, pk, rec_off(qcur[3], N)); fwd_splat_lean(f, ra, act0, act1); rec_wait(rb);                      \
            rec_request(ra, pk, rec_off(qnxt[0], N));                                                                       \
            ids4_request(qcur, sp, (unsigned)min((g_) + 2, ng - 1) << 4);                                                   \
            fwd_splat_lean(f, rb, act0, act1);                                                                              \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ra), "+s"(qcur));                                                    \
        }
        const int seg_mask = (1 << seg_shift) - 1;
        for (int g = 0; g < ng; g += 2) {
            LEAN_GROUP(qa, qb, g)
            if ((act0 | act1) == 0ull || g + 1 >= ng) break;     // (masks of the group's last splat; activity is monotone)
            LEAN_GROUP(qb, qa, g + 1)
            // checkpoint (segmented backward): the pixels' state after (g + 2) * 4 list positions, every 2^seg_shift of them (a multiple of 8)
            if (segbase != nullptr && ((((g + 2) << 2) & seg_mask) == 0))
                ckpt_store(reinterpret_cast<float*>(segbase), (size_t)(start >> seg_shift) + (size_t)(((g + 2) << 2) >> seg_shift), lane, f);
            if ((act0 | act1) == 0ull) break;
        }
#undef LEAN_GROUP
        pf_retire(pf.g);
    }
    if ((act0 | act1) != 0ull) {
        for (int p = ng << 2; p < n; p++) {              // the last n % 4 positions, one at a time
            int id;
            id_request(id, sp, (unsigned)p << 2);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(id));
            rec_request(ra, pk, rec_off(id, N));
            rec_wait(ra);
            fwd_splat_lean(f, ra, act0, act1);
        }
    }
#undef rec_off
    // work done for this tile = splats walked while some pixel was active = the largest last_contributor
    if (tile_work != nullptr) {
        const int visited = wave_max_i(max(f.lc0, f.lc1));
        if (lane == 0) tile_work[(size_t)view * (ntiles + 1) + tile] = visited;
        const LgSegLayout sl = lg_seg_layout(L, ntiles, seg_shift);
        if (segbase != nullptr && visited > (1 << seg_shift)) {           // more than one segment: their Bd needs the unclamped final colour
            float4* o = reinterpret_cast<float4*>(reinterpret_cast<float*>(segbase + sl.ckpt_fin) + ((size_t)tile * 64 + lane) * 8);
            o[0] = float4{ f.Cr.x, f.Cr.y, f.Cg.x, f.Cg.y };
            o[1] = float4{ f.Cb.x, f.Cb.y, 0.0f, 0.0f };
        }
        // the backward's work units of this tile (see the comment at LG_UNIT_CLASSES): its full segments into the first region, its
        // deepest, shorter one into the region of its length class
        if (segbase != nullptr && visited > 0 && lane == 0) {
            int* __restrict__ unit_counts = reinterpret_cast<int*>(segbase + sl.counts);
            int* __restrict__ units = reinterpret_cast<int*>(segbase + sl.units);
            const int len = 1 << seg_shift, nseg = (visited + len - 1) >> seg_shift, rem = visited - ((nseg - 1) << seg_shift);
            const int cls = ((len - rem) * LG_UNIT_CLASSES) >> seg_shift;
            const int at = atomicAdd(&unit_counts[1 + cls], 1);
            if (at < sl.cap_class) units[(size_t)sl.cap_full + (size_t)cls * sl.cap_class + at] = tile | ((nseg - 1) << 16);
            if (nseg > 1) {
                const int af = atomicAdd(&unit_counts[0], nseg - 1);
                for (int j = 0; j < nseg - 1; j++) if (af + j < sl.cap_full) units[af + j] = tile | (j << 16);
            }
        }
    }
    const size_t plane = (size_t)Hp * Wp;
    const size_t o0 = (size_t)y0 * Wp + x, o1 = (size_t)(y0 + 2) * Wp + x;
    img[((size_t)view * 3) * plane + o0] = fminf(f.Cr.x, 1.0f); img[((size_t)view * 3) * plane + o1] = fminf(f.Cr.y, 1.0f);
    img[((size_t)view * 3 + 1) * plane + o0] = fminf(f.Cg.x, 1.0f); img[((size_t)view * 3 + 1) * plane + o1] = fminf(f.Cg.y, 1.0f);
    img[((size_t)view * 3 + 2) * plane + o0] = fminf(f.Cb.x, 1.0f); img[((size_t)view * 3 + 2) * plane + o1] = fminf(f.Cb.y, 1.0f);
    trans[(size_t)view * plane + o0] = f.T.x; trans[(size_t)view * plane + o1] = f.T.y;
    last[(size_t)view * plane + o0] = (short)f.lc0; last[(size_t)view * plane + o1] = (short)f.lc1;
    wclk_store(wclk, slot, wclk_t0, n, tile, lane);
}

// Eight waves per SIMD need <= 80 scalar registers here, not the 96 the compiler's occupancy figure assumes: the runtime installs a trap
// handler, which costs every wave 16 more (measured: with 90 registers the launch never holds more than 7 waves per SIMD,
// profiles/r06_wave_clock_training_state_final.log).  The cap spills kernel arguments to lanes of one vector register outside the loop.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) raster_forward_lean_kernel(LEAN_ARGS) { raster_forward_lean_body(LEAN_PASS); }
__global__ void __launch_bounds__(256) raster_forward_lean_s96_kernel(LEAN_ARGS) { raster_forward_lean_body(LEAN_PASS); }     // A/B: lg_set_tuning(17, 2)
#undef LEAN_ARGS
#undef LEAN_PASS

LG_API int lg_raster_forward(const int* sorted_points, const int* start_index, const float* packed, const int* tiles, int K,
                             int V, long long L, int N, int H, int W, int TH, int TW, int enable_stat,
                             float* img, float* trans, short* last, int* frag_count, float* frag_weight,
                             const int* order /*nullable [V,T]: tile schedule (a permutation of 1..T)*/, int* tile_work /*nullable [V,T+1]*/, void* stream)
{
    return lg_raster_forward_bounds(sorted_points, start_index, packed, tiles, K, V, L, N, H, W, TH, TW, enable_stat, img, trans, last,
                                    frag_count, frag_weight, order, tile_work, nullptr, nullptr, 0, nullptr, nullptr, nullptr, stream);
}

// The executor's entry: the same blend, reading / filling the per-frame depth-bound blocks (single view; record dword 12 must hold the
// VIEW depth, as the fused projection writes it), with an optional gate (device int; nothing runs unless it is non-zero).
int lg_raster_forward_bounds(const int* sorted_points, const int* start_index, const float* packed, const int* tiles, int K,
                             int V, long long L, int N, int H, int W, int TH, int TW, int enable_stat,
                             float* img, float* trans, short* last, int* frag_count, float* frag_weight,
                             const int* order, int* tile_work, const int* sched_in /*nullable: previous visit's block (the bounds used)*/,
                             int* sched_out /*nullable: this visit's block; its first lg_sched_clear_words() words are zero*/,
                             int zb_check /*bit 0 clear: nothing was culled, only produce new bounds; bits 8..: bound margin in percent (0 = 50)*/,
                             int* fail_flag, int* fail_host /*nullable pinned mirror of fail_flag*/, const int* gate, void* stream)
{
    return lg_raster_forward_segments(sorted_points, start_index, packed, tiles, K, V, L, N, H, W, TH, TW, enable_stat, img, trans, last, frag_count, frag_weight,
                                      order, tile_work, sched_in, sched_out, zb_check, fail_flag, fail_host, gate, nullptr, stream);
}

// 1 if a render with these arguments takes the lean forward and may leave checkpoints for a segmented backward (the executor asks
// before it hands over the buffers; the blend backward of the same frame asks again)
int lg_raster_segments_apply(int V, int TH, int TW, int enable_stat, const int* tiles, const void* sched, const void* fail, const void* gate, const void* d_trans)
{
    return g_bwd_segments && g_fwd_lean && g_fwd_fast && g_bwd_fast == 1 && V == 1 && TH == 8 && TW == 16 && !enable_stat && tiles != nullptr && sched == nullptr &&
           fail == nullptr && gate == nullptr && d_trans == nullptr;
}
int lg_raster_segment_shift() { return g_seg_shift; }

int lg_raster_forward_segments(const int* sorted_points, const int* start_index, const float* packed, const int* tiles, int K,
                               int V, long long L, int N, int H, int W, int TH, int TW, int enable_stat,
                               float* img, float* trans, short* last, int* frag_count, float* frag_weight,
                               const int* order, int* tile_work, const int* sched_in, int* sched_out, int zb_check,
                               int* fail_flag, int* fail_host, const int* gate, const LgSegments* seg /*nullable*/, void* stream)
{
    const int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    const int ntiles = gx * gy, Hp = gy * TH, Wp = gx * TW;
    const int nslots = tiles ? K : ntiles;
    if (nslots <= 0) return 0;
    if (N >= (1 << 26)) return (int)hipErrorInvalidValue;          // 32-bit record offsets
    if ((sched_in != nullptr || sched_out != nullptr) && (V != 1 || tiles != nullptr)) return (int)hipErrorInvalidValue;
    if (!g_use_order) order = nullptr;
    dim3 grid(lg_cdiv(nslots, 4), V), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (seg != nullptr && !(lg_raster_segments_apply(V, TH, TW, enable_stat, tiles, sched_in ? (const void*)sched_in : (const void*)sched_out,
                                                     fail_flag ? (const void*)fail_flag : (const void*)fail_host, gate, nullptr) && tile_work != nullptr))
        return (int)hipErrorInvalidValue;
    if (g_fwd_lean && g_fwd_fast && TH == 8 && TW == 16 && !enable_stat && sched_in == nullptr && sched_out == nullptr && fail_flag == nullptr &&
        fail_host == nullptr && gate == nullptr) {
        if (seg != nullptr) {          // the unit counters the waves add to (one 128-byte fill)
            if (seg->shift != g_seg_shift || seg->ntiles != ntiles || seg->L != L) return (int)hipErrorInvalidValue;
            if (!seg->counts_zeroed) {
                const int rcm = (int)hipMemsetAsync(seg->base + lg_seg_layout(L, ntiles, g_seg_shift).counts, 0, sizeof(int) * 32, s);
                if (rcm) return rcm;
            }
        }
        hipLaunchKernelGGL(g_fwd_lean == 2 ? raster_forward_lean_s96_kernel : raster_forward_lean_kernel, grid, block, (size_t)g_blend_lds_fwd << 10, s,
                           sorted_points, start_index, packed, tiles, K, img, trans, last, order,
                           tile_work, gx, ntiles, L, N, Hp, Wp, nslots, g_fwd_map, g_pf_block | ((g_bwd_probe & 2) << 7) | (g_seg_shift << 16), g_wclk_fwd,
                           seg ? seg->base : (char*)nullptr);
        LG_RETURN_LAST();
    }
#define LAUNCH_RF(A_, B_, S_) hipLaunchKernelGGL((raster_forward_kernel<A_, B_, S_>), grid, block, 0, s, sorted_points, start_index, packed, \
                                                 tiles, K, img, trans, last, frag_count, frag_weight, order, tile_work, sched_in, sched_out, zb_check, fail_flag, fail_host, gate, gx, ntiles, L, N, Hp, Wp, nslots, g_fwd_map | (g_rank_prio << 8), (g_fwd_fast ? 1 : 0) | (g_pf_block << 8))
#define DISPATCH_RF(A_, B_) do { if (enable_stat) LAUNCH_RF(A_, B_, true); else LAUNCH_RF(A_, B_, false); } while (0)
    if (TH == 8 && TW == 16) DISPATCH_RF(8, 16);
    else if (TH == 16 && TW == 16) DISPATCH_RF(16, 16);
    else if (TH == 12 && TW == 16) DISPATCH_RF(12, 16);
    else if (TH == 8 && TW == 8) DISPATCH_RF(8, 8);
    else return (int)hipErrorInvalidValue;
#undef DISPATCH_RF
#undef LAUNCH_RF
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// Tile schedule: heaviest tiles first.  A frame has only ~2 tiles per resident wave slot (16 200 tiles, 8 192 slots), so the
// blend kernels' duration is set by which tiles happen to be started last; started in descending order of work the tail is made
// of the lightest tiles (longest-processing-time-first list scheduling).  Work = splats walked before the tile saturated, written
// by the forward (or derived from last_contributor); the order is a counting sort on min(work, 1023).  The schedule only
// permutes independent tiles: results do not depend on it.  (The reference has the same idea for its statistic epochs:
// litegs/utils/statistic_helper.py:77 sorts tiles by blend count and passes them as specific_tiles.)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) tile_order_kernel(const int* __restrict__ tile_work, int ntiles, int* __restrict__ order)
{
    __shared__ int hist[1024];
    __shared__ int wsum[16];
    const int view = blockIdx.x, t = threadIdx.x;
    const int* __restrict__ w = tile_work + (size_t)view * (ntiles + 1) + 1;
    int* __restrict__ o = order + (size_t)view * ntiles;
    hist[t] = 0;
    __syncthreads();
    // 16 tiles per thread and round: the loads of a round are independent (issued together), the bins stay in registers for the
    // scatter pass.  One round covers 16 384 tiles (1080p at 8x16: 16 200); larger grids take more rounds and re-read the keys.
    constexpr int R = 16;
    int bin[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int i = t + k * 1024;
        bin[k] = i < ntiles ? 1023 - min(max(w[i], 0), 1023) : -1;
    }
#pragma unroll
    for (int k = 0; k < R; k++) if (bin[k] >= 0) atomicAdd(&hist[bin[k]], 1);
    for (int i = t + R * 1024; i < ntiles; i += 1024) atomicAdd(&hist[1023 - min(max(w[i], 0), 1023)], 1);
    __syncthreads();
    // exclusive scan of the 1024 bins
    const int mine = hist[t];
    int v = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int u = __shfl_up(v, d); if ((t & 63) >= d) v += u; }
    if ((t & 63) == 63) wsum[t >> 6] = v;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < (t >> 6); k++) base += wsum[k];
    __syncthreads();
    hist[t] = base + v - mine;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < R; k++) if (bin[k] >= 0) o[atomicAdd(&hist[bin[k]], 1)] = t + k * 1024 + 1;
    for (int i = t + R * 1024; i < ntiles; i += 1024) o[atomicAdd(&hist[1023 - min(max(w[i], 0), 1023)], 1)] = i + 1;
}

LG_API int lg_tile_order(const int* tile_work /*[V,T+1]*/, int V, int ntiles, int* order /*[V,T]*/, void* stream)
{
    if (V <= 0 || ntiles <= 0) return 0;
    hipLaunchKernelGGL(tile_order_kernel, dim3(V), dim3(1024), 0, (hipStream_t)stream, tile_work, ntiles, order);
    LG_RETURN_LAST();
}

LG_API long long lg_sched_words(int H, int W, int TH, int TW)
{
    return lg_sched_words_total((W + TW - 1) / TW, (H + TH - 1) / TH);
}

// work per tile from last_contributor (operator path: the forward and the backward are separate calls): one wave per tile
template <int TH, int TW>
__global__ void __launch_bounds__(256) tile_work_kernel(const short* __restrict__ last, int gx, int ntiles, int Hp, int Wp, int* __restrict__ tile_work)
{
    constexpr int PPL = TileMap<TH, TW>::PPL;
    const int lane = threadIdx.x & 63, view = blockIdx.y;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6) + 1;
    if (tile > ntiles) return;
    const int tx = (tile - 1) % gx, ty = (tile - 1) / gx;
    const int x = tx * TW + lane % TW, q = lane / TW;
    int m = 0;
#pragma unroll
    for (int k = 0; k < PPL; k++) m = max(m, (int)last[(size_t)view * Hp * Wp + (size_t)(ty * TH + q * PPL + k) * Wp + x]);
    m = wave_max_i(m);
    if (lane == 0) tile_work[(size_t)view * (ntiles + 1) + tile] = m;
}

LG_API int lg_tile_work_from_last(const short* last, int V, int H, int W, int TH, int TW, int* tile_work /*[V,T+1]*/, void* stream)
{
    const int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    const int ntiles = gx * gy, Hp = gy * TH, Wp = gx * TW;
    dim3 grid(lg_cdiv(ntiles, 4), V), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (TH == 8 && TW == 16) hipLaunchKernelGGL((tile_work_kernel<8, 16>), grid, block, 0, s, last, gx, ntiles, Hp, Wp, tile_work);
    else if (TH == 16 && TW == 16) hipLaunchKernelGGL((tile_work_kernel<16, 16>), grid, block, 0, s, last, gx, ntiles, Hp, Wp, tile_work);
    else if (TH == 12 && TW == 16) hipLaunchKernelGGL((tile_work_kernel<12, 16>), grid, block, 0, s, last, gx, ntiles, Hp, Wp, tile_work);
    else if (TH == 8 && TW == 8) hipLaunchKernelGGL((tile_work_kernel<8, 8>), grid, block, 0, s, last, gx, ntiles, Hp, Wp, tile_work);
    else return (int)hipErrorInvalidValue;
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// a14 rasterize_backward (reference: GR/raster.cu:600-853)
//
// packed_grad record (GREC floats), MOMENT form.  With E = exp2(power*log2e + log2 opacity) (= opacity * G, the unclamped alpha),
// m = dL/dalpha * E per (pixel, splat) (the clamp at 255/256 passes the gradient, as the reference does), dx = px - X, dy = py - Y,
// and w = alpha * T the blend weight:
//   0 Mx = sum m dx     1 My = sum m dy     2 Mxx = sum m dx^2     3 Mxy = sum m dx dy     4 Myy = sum m dy^2
//   5 dr = sum w dL/dR  6 dg                7 db                   8 M0 = sum m  (== opacity * d_opacity)
// The reference kernel (raster.cu:826-841) multiplies the conic coefficients into these sums per (tile, splat); they are linear in
// the moments with per-splat constants, so the consumer does it once per splat instead (lg_moments_to_grads in lg_chain.h):
//   d_px = -(a Mx + b My)   d_py = -(c My + b Mx)   d_a = -Mxx/2   d_b01 = d_b10 = -Mxy/2   d_c = -Myy/2   d_opacity = M0 / opacity
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void swap32(float& u, float& w)       // lanes 32-63 of u <-> lanes 0-31 of w (v_permlane32_swap)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(w), false, false);
    u = __uint_as_float(r[0]); w = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& u, float& w)       // odd 16-lane rows of u <-> even rows of w (v_permlane16_swap)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(u), __float_as_uint(w), false, false);
    u = __uint_as_float(r[0]); w = __uint_as_float(r[1]);
}
// Transposing reduction steps.  After BF32(u, w): lanes 0-31 hold u[l] + u[l+32], lanes 32-63 hold w[l-32] + w[l] -- two values
// reduced by one level in swap + add, no select.  BF16: even rows hold the u sums, odd rows the w sums.
#define BF32(u, w) { swap32(u, w); (u) += (w); }
#define BF16(u, w) { swap16(u, w); (u) += (w); }
__device__ __forceinline__ float sum32(float v) { float t = v; swap32(t, v); return t + v; }
__device__ __forceinline__ float sum16(float v) { float t = v; swap16(t, v); return t + v; }
__device__ __forceinline__ float mirror16(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true)); }
// Within a row of 16: lanes 0-7 get u[l] + u[15-l], lanes 8-15 get w[l] + w[15-l]  (two bank-masked DPP adds, no select)
__device__ __forceinline__ float bfly_mirror8(float u, float w)
{
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xc" : "+v"(u) : "v"(w));
    return u;
}
// Within a half row of 8: lanes with bit 2 clear get u[l] + u[l^7], lanes with bit 2 set get w[l] + w[l^7]
__device__ __forceinline__ float bfly_hmirror4(float u, float w)
{
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa" : "+v"(u) : "v"(w));
    return u;
}

// Sum nine per-lane values over the wave; the nine totals end up in nine different lanes (wave_slot() tells which).
// The kernel is bound by VALU issue (SQ_ACTIVE_INST_VALU == its duration; a plain VALU op holds the SIMD 4 cycles, a
// v_permlane*_swap ~7.5), so: eight values go through 6 swap + add steps and 2 bank-masked DPP pairs; the ninth, which has no
// partner to be transposed with, takes its two cross-row levels through the LDS crossbar (ds_swizzle / ds_bpermute: no VALU
// issue slot, and its ~58-cycle latency overlaps the other eight values' steps).
__device__ __forceinline__ float reduce9(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7, float v8)
{
    v8 += xor_32(v8);
    BF32(v0, v1) BF32(v2, v3) BF32(v4, v5) BF32(v6, v7)
    v8 += xor_swz16(v8);
    BF16(v0, v2) BF16(v4, v6)
    v0 = bfly_mirror8(v0, v4);
    v8 += mirror16(v8);
    v0 = bfly_hmirror4(v0, v8);
    v0 += xor_dpp2(v0);
    v0 += xor_dpp1(v0);
    return v0;
}
// record slot whose total lane `lane` holds after reduce9 (-1: none).  Values v0..v7 -> lanes with (lane & 7) == 0:
// bit 3 picks v4.. over v0.., row parity picks the BF16 partner, the upper half picks the BF32 partner; v8 -> lane 4.
__device__ __forceinline__ int wave_slot(int lane)
{
    if (lane == 4) return 8;
    if (lane & 7) return -1;
    const int row = lane >> 4;
    return ((lane >> 3) & 1) * 4 + (row & 1) * 2 + (row >> 1);
}

template <int PPL>
struct BwdState {
    float X, Y[PPL], T[PPL], Bd[PPL], gR[PPL], gG[PPL], gB[PPL], gT[PPL];
    int lcrel[PPL];                  // last_contributor minus the first list position of the current 64-chunk
};

// one splat of the reverse traversal.  j = list position - chunk base (compared with lcrel); writers = exec mask of the nine lanes
// that own a record slot; slot_off = 4 * slot of this lane.
template <int PPL, bool STAT, bool TRANS, bool COUNT>
__device__ __forceinline__ void bwd_splat(BwdState<PPL>& st, const f32x16& rec, int j, unsigned pid_off, unsigned slot_off,
                                          unsigned long long writers, float* __restrict__ pg, float* __restrict__ err_square_sum,
                                          int lane, int& contributing)
{
    const float dx = rec[R_PX] - st.X;
    const float t1 = rec[R_B2] * dx;
    const float t0 = __builtin_fmaf(rec[R_A2] * dx, dx, rec[R_LO]);
    float v_r = 0.f, v_g = 0.f, v_b = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, esq = 0.f;
    if constexpr (PPL == 2 && !STAT) {
        // the lane's two pixels as one 2-vector: packed fp32 ops do two pixels per issue slot
        const v2f dyv = { rec[R_PY] - st.Y[0], rec[R_PY] - st.Y[1] };
        const v2f qv = dyv * (rec[R_C2] * dyv + t1) + t0;              // contracted to two v_pk_fma_f32
        const float E0 = __builtin_amdgcn_exp2f(qv.x);
        const float E1 = __builtin_amdgcn_exp2f(qv.y);
        const bool val0 = (E0 >= 1.0f / 256) && (j < st.lcrel[0]);
        const bool val1 = (E1 >= 1.0f / 256) && (j < st.lcrel[1]);
        if (COUNT) contributing += __any(val0 || val1) ? 1 : 0;
        const v2f Ev = { val0 ? E0 : 0.0f, val1 ? E1 : 0.0f };
        const v2f am = { fminf(255.0f / 256, Ev.x), fminf(255.0f / 256, Ev.y) };
        const v2f om = 1.0f - am;
        const v2f rc = { __builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y) };
        v2f Tv = { st.T[0], st.T[1] };
        Tv = Tv * rc;
        Tv.x = fminf(1.0f, Tv.x); Tv.y = fminf(1.0f, Tv.y);
        st.T[0] = Tv.x; st.T[1] = Tv.y;
        const v2f gRv = { st.gR[0], st.gR[1] }, gGv = { st.gG[0], st.gG[1] }, gBv = { st.gB[0], st.gB[1] };
        const v2f w = am * Tv;
        const v2f ar = w * gRv, ag = w * gGv, ab = w * gBv;
        const v2f cdot = rec[R_CR] * gRv + rec[R_CG] * gGv + rec[R_CB] * gBv;
        v2f Bdv = { st.Bd[0], st.Bd[1] };
        const v2f diff = cdot - Bdv;
        v2f d_alpha = diff * Tv;
        Bdv = Bdv + am * diff;
        st.Bd[0] = Bdv.x; st.Bd[1] = Bdv.y;
        if (TRANS) { const v2f gTv = { st.gT[0], st.gT[1] }; d_alpha = d_alpha - gTv * rc; }
        const v2f m = d_alpha * Ev;
        const v2f my = m * dyv;
        const v2f myy = my * dyv;
        v_r = ar.x + ar.y; v_g = ag.x + ag.y; v_b = ab.x + ab.y;
        s0 = m.x + m.y; s1 = my.x + my.y; s2 = myy.x + myy.y;
    } else {
        float E[PPL], dy[PPL];
        bool valid[PPL];
        bool anyv = false;
#pragma unroll
        for (int k = 0; k < PPL; k++) {
            dy[k] = rec[R_PY] - st.Y[k];
            E[k] = __builtin_amdgcn_exp2f(__builtin_fmaf(dy[k], __builtin_fmaf(rec[R_C2], dy[k], t1), t0));
            valid[k] = (E[k] >= 1.0f / 256) && (j < st.lcrel[k]);
            anyv |= valid[k];
        }
        if (COUNT) contributing += __any(anyv) ? 1 : 0;
        const float inv_o = STAT ? __builtin_amdgcn_rcpf(rec[R_O]) : 0.0f;
#pragma unroll
        for (int k = 0; k < PPL; k++) {
            if (STAT && !__any(valid[k])) continue;         // reference's per-row-group gate (raster.cu:753)
            const float Ev = valid[k] ? E[k] : 0.0f;
            const float am = fminf(255.0f / 256, Ev);
            const float rc = __builtin_amdgcn_rcpf(1.0f - am);
            st.T[k] = fminf(1.0f, st.T[k] * rc);
            const float w = am * st.T[k];
            v_r += w * st.gR[k]; v_g += w * st.gG[k]; v_b += w * st.gB[k];
            // reference (raster.cu:757-776) keeps the three blended-behind colours; only their dot product with the pixel's
            // colour gradient is ever used, and it obeys the same recurrence: B.g <- B.g + alpha * (c.g - B.g)
            const float cdot = rec[R_CR] * st.gR[k] + rec[R_CG] * st.gG[k] + rec[R_CB] * st.gB[k];
            const float diff = cdot - st.Bd[k];
            float d_alpha = diff * st.T[k];
            st.Bd[k] += am * diff;
            if (TRANS) d_alpha -= st.gT[k] * rc;
            const float m = d_alpha * Ev;
            s0 += m;
            if (STAT) { const float vo = s0 * inv_o; esq += vo * vo; }      // running-sum quirk of d_opacity, raster.cu:781-783
            s1 += m * dy[k]; s2 += m * dy[k] * dy[k];
        }
    }
    const float mx = dx * s0;
    const float tot = reduce9(mx, s1, dx * mx, dx * s1, s2, v_r, v_g, v_b, s0);
    // ONE atomic instruction from the nine lanes that hold a total: exec is narrowed around it without a branch
    const unsigned voff = pid_off + slot_off;
    asm volatile("s_mov_b64 exec, %2\n\t"
                 "global_atomic_add_f32 %0, %1, %3\n\t"
                 "s_mov_b64 exec, -1" : : "v"(voff), "v"(tot), "s"(writers), "s"(pg) : "memory");
    if (STAT) {
        esq = wave_sum(esq);
        if (lane == 0) unsafeAtomicAdd(&err_square_sum[pid_off >> 6], esq);
    }
}

template <int TH, int TW, bool STAT, bool TRANS, bool COUNT>
__global__ void __launch_bounds__(256) raster_backward_kernel(const int* __restrict__ sorted_points, const int* __restrict__ start_index,
                                                              const float* __restrict__ packed, const int* __restrict__ tiles, int K,
                                                              const float* __restrict__ final_T, const short* __restrict__ last,
                                                              const float* __restrict__ d_img, const float* __restrict__ d_trans,
                                                              float* __restrict__ packed_grad, float* __restrict__ err_square_sum,
                                                              int* __restrict__ tile_counters, const int* __restrict__ order,
                                                              int gx, int ntiles, long long L, int N, int Hp, int Wp, int nslots, int map_mode)
{
    constexpr int PPL = TileMap<TH, TW>::PPL;
    const int lane = threadIdx.x & 63;
    const int view = blockIdx.y;
    const int nb = gridDim.x;
    const int prio_mode = map_mode >> 8;
    map_mode &= 0xff;
    int blk = (tiles == nullptr && order == nullptr) ? block_remap(blockIdx.x, nb, map_mode) : (int)blockIdx.x;
    const int slot = rfl(blk * 4 + (int)(threadIdx.x >> 6));
    if (slot >= nslots) return;
    wave_rank_priority(prio_mode, slot, nslots, tiles != nullptr || order != nullptr);
    int tile = (tiles != nullptr) ? tiles[(size_t)view * K + slot] : (order != nullptr ? order[(size_t)view * ntiles + slot] : slot + 1);
    tile = rfl(tile);
    if (tile <= 0 || tile > ntiles) return;
    const int* __restrict__ si = start_index + (size_t)view * (ntiles + 2);
    const int start = rfl(si[tile]);
    const int end = rfl(si[tile + 1]);
    if (start < 0 || start >= end) return;          // empty tile: nothing to attribute (reference bug not reproduced)
    const int* __restrict__ sp = sorted_points + (size_t)view * L + start;
    const float* __restrict__ pk = packed + (size_t)view * N * REC;
    float* __restrict__ pg = packed_grad + (size_t)view * N * GREC;
    if (STAT) err_square_sum += (size_t)view * N;

    const int tx = (tile - 1) % gx, ty = (tile - 1) / gx;
    const int x = tx * TW + lane % TW;
    const int q = lane / TW;
    const int y0 = ty * TH + (q >> 1) * (2 * PPL) + (q & 1);
    const size_t plane = (size_t)Hp * Wp;
    BwdState<PPL> st;
    st.X = (float)x;
    int lc[PPL];
    int maxlast = 0;
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const size_t o = (size_t)(y0 + 2 * k) * Wp + x;
        st.Y[k] = (float)(y0 + 2 * k);
        st.T[k] = final_T[(size_t)view * plane + o];
        lc[k] = last[(size_t)view * plane + o];
        st.gR[k] = d_img[((size_t)view * 3) * plane + o];
        st.gG[k] = d_img[((size_t)view * 3 + 1) * plane + o];
        st.gB[k] = d_img[((size_t)view * 3 + 2) * plane + o];
        st.gT[k] = TRANS ? st.T[k] * d_trans[(size_t)view * plane + o] : 0.0f;     // T_final * dL/dT (raster.cu:665)
        st.Bd[k] = 0.0f;                 // (colour blended BEHIND the current splat) . dL/dC of the pixel
        maxlast = max(maxlast, lc[k]);
    }
    maxlast = rfl(wave_max_i(maxlast));
    const int n = min(maxlast, end - start);        // list positions n-1 .. 0 are walked
    int contributing = 0;
    if (n > 0) {
        const int myslot = wave_slot(lane);
        const unsigned long long writers = __ballot(myslot >= 0);
        const unsigned slot_off = (unsigned)max(myslot, 0) * 4u;
        // An even number of iterations (two per trip over a ping-pong pair of record registers): if n is odd the walk starts one
        // position early, at `n`, with the record of position n-1 -- no pixel has last_contributor > n, so that splat adds zeros.
        const int top = (n + 1) & ~1;
        int c0 = (top - 1) & ~63;                                    // first position of the current 64-chunk
        // lane l of `prv` holds the id at position c0 + l - 1: the splat to REQUEST while position c0 + l is processed
        int prv = sp[min(max(c0 + lane - 1, 0), n - 1)];
        unsigned off_a = rec_off(rfl(sp[min(top - 1, n - 1)]), N), off_b = 0;
        f32x16 ra, rb;
        rec_request(ra, pk, off_a);
        rec_wait(ra);
        for (; c0 >= 0; c0 -= 64) {
            const int prv_next = sp[min(max(c0 - 64 + lane - 1, 0), n - 1)];       // next (lower) chunk's ids, in flight during this chunk
#pragma unroll
            for (int k = 0; k < PPL; k++) st.lcrel[k] = lc[k] - c0;
            for (int j = min(63, top - 1 - c0); j >= 1; j -= 2) {          // j is odd here: positions c0 + j and c0 + j - 1
                off_b = rec_off(__builtin_amdgcn_readlane(prv, j), N);
                rec_request(rb, pk, off_b);
                bwd_splat<PPL, STAT, TRANS, COUNT>(st, ra, j, off_a, slot_off, writers, pg, err_square_sum, lane, contributing);
                rec_wait(rb);
                off_a = rec_off(__builtin_amdgcn_readlane(prv, j - 1), N);
                rec_request(ra, pk, off_a);
                bwd_splat<PPL, STAT, TRANS, COUNT>(st, rb, j - 1, off_b, slot_off, writers, pg, err_square_sum, lane, contributing);
                rec_wait(ra);
            }
            prv = prv_next;
        }
    }
    if (COUNT && lane == 0) {          // measurement hook (bench.py roofline): visited / contributing (tile, splat) iterations
        tile_counters[((size_t)view * (ntiles + 1) + tile) * 2] = n;
        tile_counters[((size_t)view * (ntiles + 1) + tile) * 2 + 1] = contributing;
    }
}

// ---------------------------------------------------------------------------------------------
// a14, the default 8x16 tile without statistics: the same reverse traversal with fewer VALU issue slots per (tile, splat)
// (the kernel is bound by VALU issue; profiles/r03_sq_counters.md):
//  * splat ids come through the scalar path as well (one s_load_dword two splats ahead of its use) -- no vector load of the
//    list, no v_readlane per splat;
//  * a splat that no pixel of the tile takes (alpha < 1/256 everywhere, or every pixel stopped before it) is dropped after the
//    alpha evaluation by one ballot: nothing changes for such a splat (alpha = 0: T, the blended-behind dot product and all nine
//    sums keep their values) -- the reference gates the same way (GR/raster.cu:752, 795);
//  * list positions below the tile's smallest last_contributor need no per-pixel "j < last_contributor" compare: the walk is split
//    at that position into a checked and an unchecked phase;
//  * the three colour sums use colour-packed copies of the pixel gradients ({dR, dG} of each pixel as one 2-vector): 4 instead of 6
//    instructions; the adds behind the permlane swaps of the transposing reduction are issued as packed adds (two totals per slot);
//  * the atomic's address is a scalar base (record of the splat) + a constant per-lane offset.
// Results: the same nine sums (the additions inside a lane associate differently: fp32 rounding only).
// ---------------------------------------------------------------------------------------------

// The nine totals from SIX per-lane values (round 6).  Every moment with a dx in it is a column-weighted sum of a dx-free one --
// Mx = sum_c dx_c S0_c, Mxx = sum_c dx_c^2 S0_c, Mxy = sum_c dx_c S1_c, with S0 / S1 the sums of m / m dy over the four lanes of a pixel
// column (dx depends on the column only) -- so the two cross-ROW levels, the expensive ones (v_permlane*_swap issues at half rate), need
// to carry S0, S1, S2 and the three colour sums only: five swaps instead of six, and no ninth value that has to cross the rows through
// the LDS crossbar (ds_bpermute + ds_swizzle: two dependent ~100-cycle round trips per splat, and an lgkmcnt(0) wait in the middle of
// the splat that also waits for the next record's scalar load).  After the row levels every row holds column totals of a different
// quantity (A: S0 | S2 | S1 | Cr; C: Cg | Cg | Cb | Cb); the weights are applied per column (X1 = A dx, X2 = X1 dx: the three
// per-lane multiplies in front of the old reduction become two behind the row levels) and the four registers go through the transposing
// in-row butterfly together: 8 DPP adds.  Totals: lane 16 r + 4 g holds, for row r, g = 0: A, 1: X1, 2: C, 3: X2 -- wave_slot_cols().
__device__ __forceinline__ float reduce9_cols(float s0, float s1, float s2, float cr, float cg, float cb, float dx)
{
    swap32(s0, s1); swap32(s2, cr); swap32(cg, cb);
    v2f a = { s0, s2 }, b = { s1, cr };
    a += b;                                           // a.x: rows 0-1 S0, rows 2-3 S1;  a.y: rows 0-1 S2, rows 2-3 Cr
    float c = cg + cb;                                // rows 0-1 Cg, rows 2-3 Cb
    float a0 = a.x, a1 = a.y;
    swap16(a0, a1);
    const float A = a0 + a1;                          // row 0 S0, row 1 S2, row 2 S1, row 3 Cr (column totals)
    const float C = sum16(c);                         // rows 0-1 Cg, rows 2-3 Cb
    const float X1 = A * dx, X2 = X1 * dx;
    float P = bfly_mirror8(A, C);
    const float Q = bfly_mirror8(X1, X2);
    P = bfly_hmirror4(P, Q);
    P += xor_dpp2(P);
    P += xor_dpp1(P);
    return P;
}
// The same with two more per-lane values (statistic epochs: the splat's fragment weight and err_square partials).  After the row levels
// the colour register of reduce9_cols holds Cg in rows 0 AND 1, Cb in rows 2 AND 3: half of it is redundant.  One more swap32 pairs the two
// extra values, and the swap16 that sum16() spends on duplicating the colour rows transposes them in instead: row 1 carries the weight
// sum, row 3 err_square; their totals come out in lanes 24 and 56 (the C column of rows 1 and 3).  Two instructions more than
// reduce9_cols, where a reduction of their own cost nine.
__device__ __forceinline__ float reduce11_cols(float s0, float s1, float s2, float cr, float cg, float cb, float b, float e, float dx)
{
    swap32(s0, s1); swap32(s2, cr); swap32(cg, cb); swap32(b, e);
    v2f a = { s0, s2 }, bb = { s1, cr };
    a += bb;                                          // a.x: rows 0-1 S0, rows 2-3 S1;  a.y: rows 0-1 S2, rows 2-3 Cr
    float c = cg + cb;                                // rows 0-1 Cg, rows 2-3 Cb
    float be = b + e;                                 // rows 0-1 weight, rows 2-3 err_square
    float a0 = a.x, a1 = a.y;
    swap16(a0, a1);
    const float A = a0 + a1;                          // row 0 S0, row 1 S2, row 2 S1, row 3 Cr (column totals)
    swap16(c, be);
    const float C = c + be;                           // row 0 Cg, row 1 weight, row 2 Cb, row 3 err_square
    const float X1 = A * dx, X2 = X1 * dx;
    float P = bfly_mirror8(A, C);
    const float Q = bfly_mirror8(X1, X2);
    P = bfly_hmirror4(P, Q);
    P += xor_dpp2(P);
    P += xor_dpp1(P);
    return P;
}
// record slot (Mx My Mxx Mxy Myy dr dg db M0 = 0..8) of the total lane `lane` holds after reduce9_cols (-1: none)
__device__ __forceinline__ int wave_slot_cols(int lane)
{
    switch (lane) {
    case 0: return 8;        // row 0, A: S0 = M0
    case 4: return 0;        // row 0, X1: Mx
    case 8: return 6;        // row 0, C: dg
    case 12: return 2;       // row 0, X2: Mxx
    case 16: return 4;       // row 1, A: S2 = Myy
    case 32: return 1;       // row 2, A: S1 = My
    case 36: return 3;       // row 2, X1: Mxy
    case 40: return 7;       // row 2, C: db
    case 48: return 5;       // row 3, A: dr
    default: return -1;
    }
}

struct BwdFast {
    float X;
    v2f Y, T, Bd, gR, gG, gB, gT;      // the lane's two pixels
    v2f G0, G1;                        // {dL/dR, dL/dG} of pixel 0 / pixel 1 (colour-packed copies)
    int lc0, lc1;
};

// STAT (statistic epochs): the per-splat sum of squares of the running d_opacity of GR/raster.cu:781-783 -- per lane, after its first pixel
// (m0 / opacity)^2 and after its second ((m0 + m1) / opacity)^2, the second term only if some pixel of the wave takes the splat in that
// pixel row group (the reference's per-row-group gate, raster.cu:753).
//   STAT == 1 (operator surface: rasterize_backward's err_square_sum output): reduced on the DPP path, added by one lane to the array.
//   STAT == 2 (executor): the three per-splat statistics of an epoch -- fragment count and fragment weight sum (what the reference's forward
//     accumulates, raster.cu:283-302: the backward sees the same validity mask and recovers the same weights) and err_square -- are reduced
//     (the count on the scalar unit, the two sums by one swap and five DPP adds) and travel in slots 9, 10, 11 of the splat's gradient record, inside the ONE atomic instruction
//     the nine moments already cost.  Measured motive (profiles/r04_stat_epoch_timeline.md): every additional per-splat atomic instruction
//     costs ~0.4 ms per 1080p frame of the late-phase cloud (9.8 M contributing (tile, splat) pairs, same-line contention), the three of the
//     separate-array form 1.2 ms.
#define STAT_SLOT_COUNT 9
#define STAT_SLOT_WEIGHT 10
#define STAT_SLOT_ERRSQ 11
template <bool TRANS, bool CHECK, int STAT>
__device__ __forceinline__ void bwd_splat_fast(BwdFast& st, const f32x16& rec, int pos, unsigned pid_off, unsigned slot_off,
                                               unsigned long long writers, float* __restrict__ pg, float* __restrict__ err_square_sum, int lane)
{
    const float dx = rec[R_PX] - st.X;
    const float t1 = rec[R_B2] * dx;
    const float t0 = __builtin_fmaf(rec[R_A2] * dx, dx, rec[R_LO]);
    const v2f dyv = rec[R_PY] - st.Y;
    const v2f qv = dyv * (rec[R_C2] * dyv + t1) + t0;
    const float E0 = __builtin_amdgcn_exp2f(qv.x);
    const float E1 = __builtin_amdgcn_exp2f(qv.y);
    bool val0 = E0 >= 1.0f / 256, val1 = E1 >= 1.0f / 256;
    if (CHECK) { val0 = val0 && (pos < st.lc0); val1 = val1 && (pos < st.lc1); }
    if (!__any(val0 || val1)) return;                   // nobody takes this splat: every quantity below keeps its value
    const v2f Ev = { val0 ? E0 : 0.0f, val1 ? E1 : 0.0f };
    const float amax = 255.0f / 256, one = 1.0f;
    const v2f am = { vmin(Ev.x, amax), vmin(Ev.y, amax) };
    const v2f om = 1.0f - am;
    const v2f rc = { __builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y) };
    v2f Tv = st.T * rc;
    Tv.x = vmin(Tv.x, one); Tv.y = vmin(Tv.y, one);
    st.T = Tv;
    const v2f w = am * Tv;
    const v2f crg = w.x * st.G0 + w.y * st.G1;          // {sum w dR, sum w dG} over the lane's two pixels
    const float v_b = __builtin_fmaf(w.x, st.gB.x, w.y * st.gB.y);
    const v2f cdot = rec[R_CR] * st.gR + rec[R_CG] * st.gG + rec[R_CB] * st.gB;
    const v2f diff = cdot - st.Bd;
    v2f d_alpha = diff * Tv;
    st.Bd = st.Bd + am * diff;
    if (TRANS) d_alpha = d_alpha - st.gT * rc;
    const v2f m = d_alpha * Ev;
    const v2f my = m * dyv;
    const float s0 = m.x + m.y;
    const float s1 = my.x + my.y;
    const float s2 = __builtin_fmaf(my.x, dyv.x, my.y * dyv.y);
    float tot;
    if constexpr (STAT == 2) {
        const float inv_o = __builtin_amdgcn_rcpf(rec[R_O]);
        const float vo0 = m.x * inv_o, vo1 = s0 * inv_o;
        const float b = w.x + w.y;                                                               // blend weights of this lane's fragments
        const float c = __builtin_fmaf(vo0, vo0, __any(val1) ? vo1 * vo1 : 0.0f);
        // the fragment COUNT is the popcount of the two validity masks: scalar unit, no reduction.  Weight and err_square ride in the
        // redundant half of the colour register of the moment reduction (reduce11_cols): totals in lanes 24 and 56
        const float fcount = (float)(__popcll(__ballot(val0)) + __popcll(__ballot(val1)));
        tot = reduce11_cols(s0, s1, s2, crg.x, crg.y, v_b, b, c, dx);
        tot = (lane == 15) ? fcount : tot;
    } else tot = reduce9_cols(s0, s1, s2, crg.x, crg.y, v_b, dx);
    // ONE atomic instruction from the lanes that hold a total: scalar base = the splat's gradient record
    const float* base = reinterpret_cast<const float*>(reinterpret_cast<const char*>(pg) + pid_off);
    asm volatile("s_mov_b64 exec, %2\n\t"
                 "global_atomic_add_f32 %0, %1, %3\n\t"
                 "s_mov_b64 exec, -1" : : "v"(slot_off), "v"(tot), "s"(writers), "s"(base) : "memory");
    if constexpr (STAT == 1) {
        const float inv_o = __builtin_amdgcn_rcpf(rec[R_O]);
        const float vo0 = m.x * inv_o, vo1 = s0 * inv_o;
        const float esq = wave_sum_to_lane63(__builtin_fmaf(vo0, vo0, __any(val1) ? vo1 * vo1 : 0.0f));
        if (lane == 63) unsafeAtomicAdd(&err_square_sum[pid_off >> 6], esq);
    }
}

// record slot of the statistics total that lane 15 / 24 / 56 holds at the end of bwd_splat_fast<.., 2> (-1: none)
__device__ __forceinline__ int stat_lane_slot(int lane)
{
    return lane == 15 ? STAT_SLOT_COUNT : (lane == 24 ? STAT_SLOT_WEIGHT : (lane == 56 ? STAT_SLOT_ERRSQ : -1));
}

#define RBF_ARGS const int* __restrict__ sorted_points, const int* __restrict__ start_index,                                             \
                 const float* __restrict__ packed, const int* __restrict__ tiles, int K,                                               \
                 const float* __restrict__ final_T, const short* __restrict__ last,                                                    \
                 const float* __restrict__ d_img, const float* __restrict__ d_trans,                                                   \
                 float* __restrict__ packed_grad, float* __restrict__ err_square_sum /*STAT*/,                                         \
                 const int* __restrict__ order,                                                                                        \
                 int gx, int ntiles, long long L, int N, int Hp, int Wp, int nslots, int map_mode_in,                                  \
                 const int* __restrict__ hot_of, long long* __restrict__ wclk,                                                         \
                 const char* __restrict__ segbase /*SEG: the frame's segment buffers (LgSegLayout)*/, int seg_shift
#define RBF_PASS sorted_points, start_index, packed, tiles, K, final_T, last, d_img, d_trans, packed_grad, err_square_sum, order, gx, ntiles, L, N, Hp, Wp, \
                 nslots, map_mode_in, hot_of, wclk, segbase, seg_shift
// -> true: `slot` lies beyond the last work unit (SEG: the caller's stride loop ends)
template <bool TRANS, int STAT, bool SEG>
__device__ __forceinline__ bool raster_backward_fast_body(int slot, RBF_ARGS)
{
    int unit = 0;                                             // tile | segment << 16
    const LgSegLayout sl = lg_seg_layout(L, ntiles, SEG ? seg_shift : 0);
    if (SEG) {
        const int* __restrict__ unit_counts = reinterpret_cast<const int*>(segbase + sl.counts);
        const int* __restrict__ units = reinterpret_cast<const int*>(segbase + sl.units);
        const int cap_full = sl.cap_full, cap_class = sl.cap_class;
        // slot -> unit: the regions in dispatch order (full segments, then the length classes, longest first), each as full as its count says
        int rest = slot, region = -1;
        size_t at = 0;
#pragma unroll
        for (int c = 0; c <= LG_UNIT_CLASSES; c++) {
            const int cnt = min(rfl(unit_counts[c]), c == 0 ? cap_full : cap_class);
            if (region < 0) {
                if (rest < cnt) { region = c; at = (c == 0 ? (size_t)0 : (size_t)cap_full + (size_t)(c - 1) * cap_class) + rest; }
                else rest -= cnt;
            }
        }
        if (region < 0) return true;
        unit = rfl(units[at]);
    }
    wave_rank_priority((map_mode_in >> 8) & 0xff, slot, nslots, tiles != nullptr || order != nullptr);
    constexpr int TH = 8, TW = 16;
    const long long wclk_t0 = wclk != nullptr ? __builtin_amdgcn_s_memrealtime() : 0;
    const int lane = threadIdx.x & 63;
    const int view = blockIdx.y;
    int tile = SEG ? (unit & 0xffff)
                   : ((tiles != nullptr) ? tiles[(size_t)view * K + slot] : (order != nullptr ? order[(size_t)view * ntiles + slot] : slot + 1));
    tile = rfl(tile);
    if (tile <= 0 || tile > ntiles) return false;
    const int* __restrict__ si = start_index + (size_t)view * (ntiles + 2);
    const int start = rfl(si[tile]);
    const int end = rfl(si[tile + 1]);
    if (start < 0 || start >= end) return false;
    const int* __restrict__ sp = sorted_points + (size_t)view * L + start;
    const float* __restrict__ pk = packed + (size_t)view * N * REC;
    float* __restrict__ pg = packed_grad + (size_t)view * N * GREC;
    if (STAT == 1) err_square_sum += (size_t)view * N;

    const int tx = (tile - 1) % gx, ty = (tile - 1) / gx;
    const int x = tx * TW + lane % TW;
    const int q = lane / TW;
    const int y0 = ty * TH + (q >> 1) * 4 + (q & 1);
    const size_t plane = (size_t)Hp * Wp;
    const size_t o0 = (size_t)y0 * Wp + x, o1 = (size_t)(y0 + 2) * Wp + x;
    BwdFast st;
    st.X = (float)x;
    st.Y = v2f{ (float)y0, (float)(y0 + 2) };
    st.T = v2f{ final_T[(size_t)view * plane + o0], final_T[(size_t)view * plane + o1] };
    st.lc0 = last[(size_t)view * plane + o0];
    st.lc1 = last[(size_t)view * plane + o1];
    st.gR = v2f{ d_img[((size_t)view * 3) * plane + o0], d_img[((size_t)view * 3) * plane + o1] };
    st.gG = v2f{ d_img[((size_t)view * 3 + 1) * plane + o0], d_img[((size_t)view * 3 + 1) * plane + o1] };
    st.gB = v2f{ d_img[((size_t)view * 3 + 2) * plane + o0], d_img[((size_t)view * 3 + 2) * plane + o1] };
    st.G0 = v2f{ st.gR.x, st.gG.x };
    st.G1 = v2f{ st.gR.y, st.gG.y };
    st.gT = v2f{ 0.0f, 0.0f };
    if (TRANS) st.gT = st.T * v2f{ d_trans[(size_t)view * plane + o0], d_trans[(size_t)view * plane + o1] };      // T_final * dL/dT (raster.cu:665)
    st.Bd = v2f{ 0.0f, 0.0f };
    const int maxlast = rfl(wave_max_i(max(st.lc0, st.lc1)));
    const int minlast = -rfl(wave_max_i(-min(st.lc0, st.lc1)));
    int n = min(maxlast, end - start);              // list positions n-1 .. 0 are walked
    int lo = 0;                                     // SEG: this wave walks positions hi-1 .. lo of the tile's n
    if (SEG) {
        const int seg = (int)((unsigned)unit >> 16);
        lo = seg << seg_shift;
        if (lo >= n) return false;
        const int hi = min(lo + (1 << seg_shift), n);
        if (hi < n) {
            // not the tile's deepest segment: the state behind position hi-1 comes from the forward's checkpoint at hi -- T as the forward
            // had it (exact, where the deepest segment's walk recovers it by divisions), and the colour blended behind, dotted with the
            // pixel's gradient, from the colour accumulated by then: sum_{j >= hi} T_j alpha_j c_j = C_final - C_hi = T_hi * (behind colour)
            const float4* c = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(segbase) + (((size_t)(start >> seg_shift) + seg + 1) * 64 + lane) * 8);
            const float4* fc = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(segbase + sl.ckpt_fin) + ((size_t)tile * 64 + lane) * 8);
            const float4 a = c[0], b = c[1], fa = fc[0], fb = fc[1];
            st.T = v2f{ a.x, a.y };
            const v2f num = v2f{ fa.x - a.z, fa.y - a.w } * st.gR + v2f{ fa.z - b.x, fa.w - b.y } * st.gG + v2f{ fb.x - b.z, fb.y - b.w } * st.gB;
            st.Bd = v2f{ a.x > 0.0f ? num.x / a.x : 0.0f, a.y > 0.0f ? num.y / a.y : 0.0f };
        }
        n = hi;
    }
    if (n <= 0) return false;
    int myslot = wave_slot_cols(lane);
    if (STAT == 2 && stat_lane_slot(lane) >= 0) myslot = stat_lane_slot(lane);
    const unsigned long long writers = ((map_mode_in >> 24) & 1) ? 0ull : __ballot(myslot >= 0);      // (bit 24: measurement hook, no atomics)

    const unsigned idmask = ((map_mode_in >> 25) & 1) ? 0x3ffu : 0xffffffffu;                          // (bit 25: measurement hook, every tile reads the same 1024 records)
#define rec_off(id_, N_) rec_off((int)((unsigned)(id_) & idmask), N_)
    const unsigned slot_off = (unsigned)max(myslot, 0) * 4u;
    // An even number of iterations (two per trip over a ping-pong pair of record registers): if n is odd the walk starts one
    // position early, at `n`, with the record of position n-1 -- no pixel has last_contributor > n, so that splat adds nothing.
    const int top = (n + 1) & ~1;
    int pos = top - 1;                               // position of the splat in `ra`
    int id_a, id_b;
    f32x16 ra, rb;
    id_request(id_a, sp, (unsigned)min(pos, n - 1) << 2);
    id_request(id_b, sp, (unsigned)(pos - 1) << 2);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(id_a), "+s"(id_b));
    unsigned off_a = rec_off(id_a, N), off_b = rec_off(id_b, N);
    // replicas: a splat that covers many tiles has R = 2^k gradient lines behind the N regular ones (hot_of: first line << 6 | k, -1: none;
    // assigned by the projection, fused.hip); this tile adds into replica (tile mod R) -- same-line contention at the memory-side atomic
    // units was 70-80 % of what the atomics cost (profiles/r03_bwd_ab.log).  The word travels with the splat's record (scalar path).
    const int* __restrict__ hot = hot_of != nullptr ? hot_of : sp;        // without replicas the loads below still go somewhere valid
    const unsigned hot_on = hot_of != nullptr ? 1u : 0u;
    int hot_a = -1, hot_b = -1;
    rec_request(ra, pk, off_a);
    id_request(hot_a, hot, hot_on ? off_a >> 4 : 0u);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ra), "+s"(hot_a));
    // (an L2 warm-up like the forward's -- PfState -- changes nothing here: 3.604 / 3.591 ms per step with, 3.596 / 3.592 without, profiles/r06_l2_warmup_ab.log)
    // phase 1: positions that some pixel of the tile had already stopped before (pos >= minlast): per-pixel last_contributor test
#define HOT_TARGET(h, off) ((hot_on && (h) >= 0) ? (((unsigned)N + ((unsigned)(h) >> 6) + ((unsigned)tile & ((1u << ((h) & 63)) - 1u))) << 6) : (off))
#define BWD_PAIR(CHK)                                                                                         \
    {                                                                                                         \
        rec_request(rb, pk, off_b);                                                                           \
        id_request(hot_b, hot, hot_on ? off_b >> 4 : 0u);                                                     \
        id_request(id_a, sp, (unsigned)max(pos - 2, 0) << 2);                                                 \
        bwd_splat_fast<TRANS, CHK, STAT>(st, ra, pos, HOT_TARGET(hot_a, off_a), slot_off, writers, pg, err_square_sum, lane);  \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(rb), "+s"(id_a), "+s"(hot_b));                             \
        off_a = rec_off(id_a, N);                                                                          \
        rec_request(ra, pk, off_a);                                                                           \
        id_request(hot_a, hot, hot_on ? off_a >> 4 : 0u);                                                     \
        id_request(id_b, sp, (unsigned)max(pos - 3, 0) << 2);                                                 \
        bwd_splat_fast<TRANS, CHK, STAT>(st, rb, pos - 1, HOT_TARGET(hot_b, off_b), slot_off, writers, pg, err_square_sum, lane);  \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ra), "+s"(id_b), "+s"(hot_a));                             \
        off_b = rec_off(id_b, N);                                                                          \
        pos -= 2;                                                                                             \
    }
    for (; pos >= lo + 1 && pos >= minlast; ) BWD_PAIR(true)
    for (; pos >= lo + 1; ) BWD_PAIR(false)
    wclk_store(wclk, slot, wclk_t0, n - lo, tile, lane);
#undef rec_off
#undef BWD_PAIR
#undef HOT_TARGET
    return false;
}

// map_mode: bits 0-7 workgroup -> tile map, 8-15 priority switch, 24 / 25 measurement hooks.  SEG: one wave per (tile, segment) unit; the
// grid covers an upper bound of the unit count (tiles + table length / segment length), waves beyond the count leave at once.
template <bool TRANS, int STAT, bool SEG = false>
__global__ void __launch_bounds__(256) raster_backward_fast_kernel(RBF_ARGS)
{
    const int blk = (SEG || tiles != nullptr || order != nullptr) ? (int)blockIdx.x : block_remap(blockIdx.x, gridDim.x, map_mode_in & 0xff);
    int slot = rfl(blk * 4 + (int)(threadIdx.x >> 6));
    if (!SEG) {
        if (slot < nslots) raster_backward_fast_body<TRANS, STAT, false>(slot, RBF_PASS);
        return;
    }
    // (a stride loop over the units -- a grid sized by an estimate -- was tried: 106 scalar / 71 vector registers, six waves per SIMD)
    if (slot < nslots) raster_backward_fast_body<TRANS, STAT, true>(slot, RBF_PASS);
}
#undef RBF_ARGS
#undef RBF_PASS



// ---------------------------------------------------------------------------------------------
// a14, SPLAT-PARALLEL formulation ("measure both"; the shape of the reference's backward, GR/raster.cu:696-849, where
// threads own Gaussians).  A/B variant, lg_set_tuning(5, 2); not the default -- the numbers are in DESIGN.md section 9 (round 5).
//
// One wave per 8x16 tile.  The tile's list is walked from its deep end in batches of 64 positions; lane l of a batch owns position
// hi - 1 - l (lane 0 = the deepest splat of the batch) and keeps that splat's record and its nine moment sums in registers.  The tile's
// 128 pixels are LOOPED: a pixel's constants (position, dL/dC, last_contributor) and its running state -- T behind the batch and Bd, the
// colour blended behind it dotted with dL/dC -- sit in LDS and are read by broadcast.  What the pixel-parallel kernel gets for free from
// its sequential walk has to be rebuilt across the lanes: the transmittance in front of splat l is T_state / prod_{i <= l} (1 - alpha_i)
// and the behind-colour recursion Bd <- (1 - alpha) Bd + alpha (c . g) is an affine map per splat, so one inclusive scan of affine maps
// over the 64 lanes (six DPP steps: row_shr 1, 2, 4, 8, row_bcast 15, 31; a multiply and a multiply-add each) yields both, its exclusive
// form (wave_shr 1) gives every lane the Bd it sees, and lane 63 holds the pixel's new state.  No nine-value reduction per (tile, splat);
// the 64 x 9 sums of a batch are transposed through LDS into nine atomic instructions of consecutive words -- against two 64-lane scans
// per (pixel, batch).
// ---------------------------------------------------------------------------------------------
#define DPP_MOV_F(old_, v_, ctrl_, rm_) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old_), __float_as_int(v_), ctrl_, rm_, 0xF, false))
// One step of the inclusive scan of affine maps: (A, B) <- (A, B) o (A_low, B_low) -- first the lower lanes' map, then this lane's.  With the
// DPP operand on the instruction itself a lane WITHOUT a source is disabled and keeps (A, B), which is the composition with the identity:
//   B += B_low * A   (v_fmac_f32 with the DPP-shifted B as first operand; A is still this lane's own factor)
//   A *= A_low
// (the compiler does not fold update_dpp into these two: as v_mov_dpp + op + a re-initialised `old` every step the scan was 40 instructions
// instead of 12; s_nop 0: a VGPR written by a VALU instruction needs two wait states before a DPP read)
#define AFFINE_STEP(A_, B_, CTRL_)                                                       \
    asm volatile("s_nop 0\n\t"                                                          \
                 "v_fmac_f32_dpp %1, %1, %0 " CTRL_ " bank_mask:0xf\n\t"                \
                 "v_mul_f32_dpp %0, %0, %0 " CTRL_ " bank_mask:0xf" : "+v"(A_), "+v"(B_))

__global__ void __launch_bounds__(256) raster_backward_sp_kernel(const int* __restrict__ sorted_points, const int* __restrict__ start_index,
                                                                 const float* __restrict__ packed, const int* __restrict__ tiles, int K,
                                                                 const float* __restrict__ final_T, const short* __restrict__ last,
                                                                 const float* __restrict__ d_img, float* __restrict__ packed_grad,
                                                                 const int* __restrict__ order,
                                                                 int gx, int ntiles, long long L, int N, int Hp, int Wp, int nslots, int map_mode)
{
    constexpr int TH = 8, TW = 16, NPX = TH * TW;
    __shared__ float4 px_a[4][NPX];       // per pixel: T behind the current batch, Bd, dL/dR, dL/dG
    __shared__ float4 px_b[4][NPX];       // dL/dB, last_contributor (int bits), x, y
    __shared__ float tr9[4][64 * 9];      // a batch's 64 x 9 sums, transposed for the atomics
    __shared__ int trid[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int view = blockIdx.y;
    const int nb = gridDim.x;
    const int prio_mode = map_mode >> 8;
    map_mode &= 0xff;
    int blk = (tiles == nullptr && order == nullptr) ? block_remap(blockIdx.x, nb, map_mode) : (int)blockIdx.x;
    const int slot = rfl(blk * 4 + wave);
    if (slot >= nslots) return;
    wave_rank_priority(prio_mode, slot, nslots, tiles != nullptr || order != nullptr);
    int tile = (tiles != nullptr) ? tiles[(size_t)view * K + slot] : (order != nullptr ? order[(size_t)view * ntiles + slot] : slot + 1);
    tile = rfl(tile);
    if (tile <= 0 || tile > ntiles) return;
    const int* __restrict__ si = start_index + (size_t)view * (ntiles + 2);
    const int start = rfl(si[tile]);
    const int end = rfl(si[tile + 1]);
    if (start < 0 || start >= end) return;
    const int* __restrict__ sp = sorted_points + (size_t)view * L + start;
    const float4* __restrict__ pk4 = reinterpret_cast<const float4*>(packed + (size_t)view * N * REC);
    float* __restrict__ pg = packed_grad + (size_t)view * N * GREC;
    const int tx = (tile - 1) % gx, ty = (tile - 1) / gx;
    const size_t plane = (size_t)Hp * Wp;
    int maxlast = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {                                  // pixel p = lane + 64 k: row p / 16, column p % 16 (coalesced image rows)
        const int p = lane + 64 * k;
        const int x = tx * TW + (p % TW), y = ty * TH + (p / TW);
        const size_t o = (size_t)y * Wp + x;
        const int lc = last[(size_t)view * plane + o];
        px_a[wave][p] = make_float4(final_T[(size_t)view * plane + o], 0.0f, d_img[((size_t)view * 3) * plane + o], d_img[((size_t)view * 3 + 1) * plane + o]);
        px_b[wave][p] = make_float4(d_img[((size_t)view * 3 + 2) * plane + o], __int_as_float(lc), (float)x, (float)y);
        maxlast = max(maxlast, lc);
    }
    maxlast = rfl(wave_max_i(maxlast));
    const int n = min(maxlast, end - start);                        // list positions n-1 .. 0 are walked
    if (n <= 0) return;
    __builtin_amdgcn_wave_barrier();                                // (LDS operations of one wave execute in order)
    const float amax = 255.0f / 256;
    for (int hi = n; hi > 0; hi -= 64) {
        const int lo = hi > 64 ? hi - 64 : 0;
        const int j = hi - 1 - lane;                                // this lane's list position (lane 0: the deepest of the batch)
        const bool act = j >= lo;
        const int id = sp[act ? j : lo];
        const unsigned rid = min((unsigned)id, (unsigned)(N - 1));
        const float4 r0 = pk4[(size_t)rid * 4], r1 = pk4[(size_t)rid * 4 + 1], r2 = pk4[(size_t)rid * 4 + 2], r3 = pk4[(size_t)rid * 4 + 3];
        const float spx = r0.x, spy = r0.y, A2 = r0.z, B2 = r0.w, C2 = r1.x, cr = r1.z, cg = r1.w, cb = r2.x, LO = r3.w;
        float Mx = 0.f, My = 0.f, Mxx = 0.f, Mxy = 0.f, Myy = 0.f, dR = 0.f, dG = 0.f, dB = 0.f, M0 = 0.f;
        for (int p = 0; p < NPX; p++) {
            const float4 sa = px_a[wave][p], sb = px_b[wave][p];      // broadcast reads
            const int lc = rfl(__float_as_int(sb.y));
            if (lc <= lo) continue;                                  // the pixel stopped in front of this batch: nothing changes for it
            const float dx = spx - sb.z, dy = spy - sb.w;
            const float q = __builtin_fmaf(__builtin_fmaf(C2, dy, B2 * dx), dy, __builtin_fmaf(A2 * dx, dx, LO));
            const float E = __builtin_amdgcn_exp2f(q);
            const bool val = act && (E >= 1.0f / 256) && (j < lc);
            const float Ev = val ? E : 0.0f;
            const float am = vmin(Ev, amax);
            const float cdot = __builtin_fmaf(cr, sa.z, __builtin_fmaf(cg, sa.w, cb * sb.x));
            float A = 1.0f - am, B = am * cdot;                      // this splat's map on Bd: Bd <- A Bd + B; A is also its factor on T
            asm volatile("s_nop 0" : "+v"(A), "+v"(B));              // (second wait state in front of the first DPP read)
            AFFINE_STEP(A, B, "row_shr:1 row_mask:0xf");
            AFFINE_STEP(A, B, "row_shr:2 row_mask:0xf");
            AFFINE_STEP(A, B, "row_shr:4 row_mask:0xf");
            AFFINE_STEP(A, B, "row_shr:8 row_mask:0xf");
            AFFINE_STEP(A, B, "row_bcast:15 row_mask:0xa");
            AFFINE_STEP(A, B, "row_bcast:31 row_mask:0xc");
            // exclusive form: what the deeper lanes of the batch did to Bd before this splat (lane 0: nothing)
            const float Ae = DPP_MOV_F(1.0f, A, 0x138, 0xF), Be = DPP_MOV_F(0.0f, B, 0x138, 0xF);      // wave_shr:1
            const float Tb = vmin(sa.x * __builtin_amdgcn_rcpf(A), 1.0f);                               // T in front of this splat
            const float Bd = __builtin_fmaf(Ae, sa.y, Be);
            const float w = am * Tb;
            dR = __builtin_fmaf(w, sa.z, dR); dG = __builtin_fmaf(w, sa.w, dG); dB = __builtin_fmaf(w, sb.x, dB);
            const float m = (cdot - Bd) * Tb * Ev;
            const float mx = m * dx, my = m * dy;
            M0 += m; Mx += mx; My += my;
            Mxx = __builtin_fmaf(mx, dx, Mxx); Mxy = __builtin_fmaf(mx, dy, Mxy); Myy = __builtin_fmaf(my, dy, Myy);
            if (lane == 63) {                                        // the pixel's state behind the NEXT (shallower) batch
                float2 st2 = make_float2(Tb, __builtin_fmaf(A, sa.y, B));
                *reinterpret_cast<float2*>(&px_a[wave][p]) = st2;
            }
        }
        // The 64 x 9 sums leave through an LDS transpose: as nine atomics per lane every instruction touched 64 records (576 line requests
        // per batch, measured 4x the whole kernel's issue time); transposed, consecutive lanes hold consecutive words of a record and an
        // instruction touches 7-8 of them -- the request count of the pixel-parallel kernel's one nine-lane atomic per (tile, splat).
        float* t9 = tr9[wave];
        t9[lane * 9 + 0] = Mx; t9[lane * 9 + 1] = My; t9[lane * 9 + 2] = Mxx; t9[lane * 9 + 3] = Mxy; t9[lane * 9 + 4] = Myy;
        t9[lane * 9 + 5] = dR; t9[lane * 9 + 6] = dG; t9[lane * 9 + 7] = dB; t9[lane * 9 + 8] = M0;
        trid[wave][lane] = (int)rid;
        __builtin_amdgcn_wave_barrier();
        const int cnt = hi - lo;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int f = k * 64 + lane;
            const int sidx = (f * 7282) >> 16;                       // f / 9 for f < 576
            const int c = f - 9 * sidx;
            const float v = t9[f];
            if (sidx < cnt && v != 0.0f) unsafeAtomicAdd(pg + (size_t)trid[wave][sidx] * GREC + c, v);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

LG_API int lg_set_tuning(int key, int value)
{
    switch (key) {
    // (maps travel in the low byte of the kernels' map_mode argument, the priority switch above it: a map beyond 255 would turn priorities on)
    case 1: if (value < 0 || value > 255) return (int)hipErrorInvalidValue; g_bwd_map = value; return 0;      // workgroup -> tile map of the blend backward (block_remap)
    case 2: if (value < 0 || value > 255) return (int)hipErrorInvalidValue; g_fwd_map = value; return 0;      // ... of the blend forward
    case 4: g_use_order = value; return 0;                                    // 0: ignore the heaviest-first tile schedule
    case 7: g_fwd_fast = value; return 0;                                     // 0: the generic blend loop also for 8x16 tiles without statistics
    case 5: g_bwd_fast = value; return 0;                                     // 0: the generic blend backward also for 8x16 tiles without statistics; 2: the splat-parallel variant (A/B)
    case 16: if (value != 0 && value != 8 && value != 16 && value != 32 && value != 64) return (int)hipErrorInvalidValue; g_pf_block = value; return 0;   // L2 warm-up block of the fast blend kernels
    case 22: if (value < 0 || value > 1) return (int)hipErrorInvalidValue; g_bwd_segments = value; return 0;       // segmented blend backward on / off
    case 23: if (value < 6 || value > 12) return (int)hipErrorInvalidValue; g_seg_shift = value; return 0;        // log2 of the segment length (64 .. 4096 list positions)
    case 17: if (value < 0 || value > 2) return (int)hipErrorInvalidValue; g_fwd_lean = value; return 0;      // lean blend forward on / off
    case 19: if (value < 0 || value > 64) return (int)hipErrorInvalidValue; g_blend_lds_fwd = value; return 0;   // occupancy cap of the lean blend forward (KB of dynamic LDS per workgroup)
    case 20: if (value < 0 || value > 64) return (int)hipErrorInvalidValue; g_blend_lds_bwd = value; return 0;   // ... of the fast blend backward
    case 18: g_bwd_probe = value; return 0;                                      // measurement hook: blend backward without its atomics
    case 8: g_rank_prio = value ? 1 : 0; return 0;                            // 1: issue priority by rank in a heavy-first schedule (wave_rank_priority)
    case 10: case 11: case 15: case 24: case 26: return lg_binning_set_tuning(key, value);   // binning.hip: key emission variants (10, 11), look-back width of small sorts (15)
    case 12: return lg_fused_set_tuning(key, value);                           // fused projection (fused.hip): SH loads in front of the tile walk
    default: return (int)hipErrorInvalidValue;
    }
}

// 1 if a statistics render of this tile shape carries fragment count / weight / err_square inside the gradient record (slots 9-11, the
// 8x16 moment-form kernel; off when that kernel was disabled through lg_set_tuning(5, 0)): the executor's caller then passes no
// separate statistics arrays (err_square_sum == NULL); otherwise it must allocate them.
LG_API int lg_stat_in_record_supported(int TH, int TW) { return (TH == 8 && TW == 16 && g_bwd_fast) ? 1 : 0; }

LG_API int lg_raster_backward(const int* sorted_points, const int* start_index, const float* packed, const int* tiles, int K,
                              const float* final_T, const short* last, const float* d_img, const float* d_trans,
                              int V, long long L, int N, int H, int W, int TH, int TW, int enable_stat,
                              float* packed_grad /*[V,N,16] zeroed*/, float* err_square_sum /*[V,1,N] zeroed*/,
                              int* tile_counters /*nullable [V,T+1,2]*/, const int* order /*nullable [V,T]*/, void* stream)
{
    return lg_raster_backward_hot(sorted_points, start_index, packed, tiles, K, final_T, last, d_img, d_trans, V, L, N, H, W, TH, TW, enable_stat,
                                  packed_grad, err_square_sum, tile_counters, order, nullptr, 0, stream);
}

// The executor's entry: hot_of (nullable int32[N], V == 1) assigns replica lines to the splats that cover many tiles; packed_grad then has
// N + hot_lines lines (the consumer folds them: lg_gaussian_bwd.h load_moments_folded).  Only the 8x16 fast kernel uses them.
int lg_raster_backward_hot(const int* sorted_points, const int* start_index, const float* packed, const int* tiles, int K,
                           const float* final_T, const short* last, const float* d_img, const float* d_trans,
                           int V, long long L, int N, int H, int W, int TH, int TW, int enable_stat,
                           float* packed_grad, float* err_square_sum, int* tile_counters, const int* order,
                           const int* hot_of, long long hot_lines, void* stream)
{
    return lg_raster_backward_segments(sorted_points, start_index, packed, tiles, K, final_T, last, d_img, d_trans, V, L, N, H, W, TH, TW, enable_stat,
                                       packed_grad, err_square_sum, tile_counters, order, hot_of, hot_lines, nullptr, stream);
}

int lg_raster_backward_segments(const int* sorted_points, const int* start_index, const float* packed, const int* tiles, int K,
                                const float* final_T, const short* last, const float* d_img, const float* d_trans,
                                int V, long long L, int N, int H, int W, int TH, int TW, int enable_stat,
                                float* packed_grad, float* err_square_sum, int* tile_counters, const int* order,
                                const int* hot_of, long long hot_lines, const LgSegments* seg /*nullable: what the frame's forward filled*/, void* stream)
{
    LG_REQUIRE(sorted_points, start_index, packed, final_T, last, d_img, packed_grad);
    const int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    const int ntiles = gx * gy, Hp = gy * TH, Wp = gx * TW;
    const int nslots = tiles ? K : ntiles;
    if (nslots <= 0) return 0;
    if ((long long)N + hot_lines >= (1 << 26)) return (int)hipErrorInvalidValue;          // 32-bit record offsets
    if (hot_of != nullptr && V != 1) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    if (!g_use_order) order = nullptr;
    dim3 grid(lg_cdiv(nslots, 4), V), block(256);
#define LAUNCH_RB(A_, B_, S_, T_, C_) hipLaunchKernelGGL((raster_backward_kernel<A_, B_, S_, T_, C_>), grid, block, 0, s, sorted_points, start_index, \
                                                         packed, tiles, K, final_T, last, d_img, d_trans, packed_grad, err_square_sum, tile_counters, order, \
                                                         gx, ntiles, L, N, Hp, Wp, nslots, g_bwd_map | (g_rank_prio << 8))
#define DISPATCH_RB(A_, B_)                                                   \
    do {                                                                      \
        if (enable_stat) { if (d_trans) LAUNCH_RB(A_, B_, true, true, false); else LAUNCH_RB(A_, B_, true, false, false); } \
        else { if (d_trans) LAUNCH_RB(A_, B_, false, true, false); else LAUNCH_RB(A_, B_, false, false, false); }           \
    } while (0)
    if (tile_counters != nullptr) {          // measurement variant: the plain 8x16 kernel with the two counters
        if (TH != 8 || TW != 16 || enable_stat || d_trans) return (int)hipErrorInvalidValue;
        LAUNCH_RB(8, 16, false, false, true);
    }
    else if (enable_stat && err_square_sum == nullptr && !(TH == 8 && TW == 16 && g_bwd_fast && hot_of == nullptr))
        return (int)hipErrorInvalidValue;          // statistics inside the gradient record exist in the 8x16 moment-form kernel only
    else if (TH == 8 && TW == 16 && g_bwd_fast == 2 && !enable_stat && d_trans == nullptr) {
        // splat-parallel A/B variant (adds into the splats' main records; a frame's gradient replicas, if any, stay zero and fold to nothing)
        hipLaunchKernelGGL(raster_backward_sp_kernel, grid, block, 0, s, sorted_points, start_index, packed, tiles, K, final_T, last, d_img, packed_grad, order,
                           gx, ntiles, L, N, Hp, Wp, nslots, g_bwd_map | (g_rank_prio << 8));
    }
    else if (TH == 8 && TW == 16 && g_bwd_fast && !(enable_stat && hot_of != nullptr)) {
#define LAUNCH_RBF(T_, S_) hipLaunchKernelGGL((raster_backward_fast_kernel<T_, S_>), grid, block, (size_t)g_blend_lds_bwd << 10, s, sorted_points, start_index, packed, tiles, K, final_T, last, \
                                              d_img, d_trans, packed_grad, err_square_sum, order, gx, ntiles, L, N, Hp, Wp, nslots,                                                             \
                                              g_bwd_map | (g_rank_prio << 8) | ((g_bwd_probe & 1) << 24) | (((g_bwd_probe >> 2) & 1) << 25), hot_of, g_wclk_bwd, nullptr, 0)
        if (seg != nullptr) {          // one wave per (tile, segment) unit of the list the forward's bwd_units_kernel left; the grid covers the bound
            if (!lg_raster_segments_apply(V, TH, TW, enable_stat, tiles, nullptr, nullptr, nullptr, d_trans)) return (int)hipErrorInvalidValue;
            if (seg->shift != g_seg_shift || seg->ntiles != ntiles || seg->L != L) return (int)hipErrorInvalidValue;
            // units = tiles + full segments <= tiles + table length / segment length; the waves beyond the count cost nothing measurable
            // (a grid of 2 x tiles instead of 5 x: 3.404 / 3.401 against 3.390 / 3.395 ms per step, profiles/r06_bwd_segments_ab.log)
            const int est = nslots + (int)((L >> g_seg_shift) + 1);
            dim3 ugrid(lg_cdiv(est, 4), 1);
            hipLaunchKernelGGL((raster_backward_fast_kernel<false, 0, true>), ugrid, block, 0, s, sorted_points, start_index, packed, tiles, K, final_T, last,
                               d_img, d_trans, packed_grad, err_square_sum, order, gx, ntiles, L, N, Hp, Wp, est,
                               g_bwd_map | ((g_bwd_probe & 1) << 24), hot_of, g_wclk_bwd, (const char*)seg->base, g_seg_shift);
        }
        else if (enable_stat && err_square_sum == nullptr) { if (d_trans) LAUNCH_RBF(true, 2); else LAUNCH_RBF(false, 2); }       // executor: statistics in the record
        else if (enable_stat) { if (d_trans) LAUNCH_RBF(true, 1); else LAUNCH_RBF(false, 1); }
        else { if (d_trans) LAUNCH_RBF(true, 0); else LAUNCH_RBF(false, 0); }
#undef LAUNCH_RBF
    }
    else if (TH == 8 && TW == 16) DISPATCH_RB(8, 16);
    else if (TH == 16 && TW == 16) DISPATCH_RB(16, 16);
    else if (TH == 12 && TW == 16) DISPATCH_RB(12, 16);
    else if (TH == 8 && TW == 8) DISPATCH_RB(8, 8);
    else return (int)hipErrorInvalidValue;
#undef DISPATCH_RB
#undef LAUNCH_RB
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// unpack_gradient (reference: GR/raster.cu:855-886): moments -> d_ndc / d_cov2d_inv / d_color / d_opacity.  inv_scaler =
// *grad_inv_scaler (the reference's extra 1/128 undoes its fp16 transmittance scale, which does not exist here).
// d_opacity is summed over views.  `packed` supplies the per-splat constants (conic a, b, c and opacity) of the moment combination.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) unpack_gradient_kernel(const float4* __restrict__ packed_grad, const float4* __restrict__ packed,
                                                              const float* __restrict__ grad_inv_scaler,
                                                              const int* __restrict__ valid_length, int V, int N, int H, int W,
                                                              float* __restrict__ d_ndc, float* __restrict__ d_inv_cov,
                                                              float* __restrict__ d_color, float* __restrict__ d_opacity)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const bool live = i < lg_valid_len(valid_length, N);
    const float sc = grad_inv_scaler ? grad_inv_scaler[0] : 1.0f;
    float dop = 0.0f;
    for (int b = 0; b < V; b++) {
        float g[9] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        if (live) {
            const float4* rec = packed_grad + ((size_t)b * N + i) * (GREC / 4);
            const float4* pr = packed + ((size_t)b * N + i) * (REC / 4);
            const float4 m0 = rec[0], m1 = rec[1];
            const float M0 = rec[2].x;
            const float4 p1 = pr[1], p2 = pr[2];
            lg_moments_to_grads(m0.x, m0.y, m0.z, m0.w, m1.x, M0, p2.y, p2.z, p2.w, p1.y, g);
            g[5] = m1.y; g[6] = m1.z; g[7] = m1.w;
        }
        d_ndc[((size_t)b * 4) * N + i] = g[0] * 0.5f * W * sc;
        d_ndc[((size_t)b * 4 + 1) * N + i] = g[1] * 0.5f * H * sc;
        d_ndc[((size_t)b * 4 + 2) * N + i] = 0.0f;
        d_ndc[((size_t)b * 4 + 3) * N + i] = 0.0f;
        d_inv_cov[((size_t)b * 4) * N + i] = g[2] * sc;
        d_inv_cov[((size_t)b * 4 + 1) * N + i] = g[3] * sc;
        d_inv_cov[((size_t)b * 4 + 2) * N + i] = g[3] * sc;
        d_inv_cov[((size_t)b * 4 + 3) * N + i] = g[4] * sc;
        d_color[((size_t)b * 3) * N + i] = g[5] * sc;
        d_color[((size_t)b * 3 + 1) * N + i] = g[6] * sc;
        d_color[((size_t)b * 3 + 2) * N + i] = g[7] * sc;
        dop += g[8] * sc;
    }
    d_opacity[i] = dop;
}

LG_API int lg_unpack_gradient(const float* packed_grad, const float* packed, const float* grad_inv_scaler, const int* valid_length, int V, int N, int H, int W,
                              float* d_ndc, float* d_inv_cov, float* d_color, float* d_opacity, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(packed_grad, packed, d_ndc, d_inv_cov, d_color, d_opacity);
    hipLaunchKernelGGL(unpack_gradient_kernel, dim3(lg_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)packed_grad, (const float4*)packed, grad_inv_scaler, valid_length, V, N, H, W, d_ndc, d_inv_cov, d_color, d_opacity);
    LG_RETURN_LAST();
}

LG_API int lg_packed_record_floats(void) { return REC; }
LG_API int lg_packed_grad_floats(void) { return GREC; }
