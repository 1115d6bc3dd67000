// Fused hot path: the native executor of one render forward / backward / optimizer step.
//
// The drop-in surface (GR/ext_cuda.cpp's 26 functions) forces one launch per operator and ~60 host-side
// calls per training iteration; on MI355X the blend kernels are fast enough that this host work (>2 ms)
// was the bottleneck.  This file provides what a CDNA-first design wants instead:
//   * project_fused_kernel   : gather + activate + SH->RGB + MVP + cov2d + inverse + tile count + record pack in
//                              ONE pass over the visible Gaussians (reads 236 B, writes ~100 B per Gaussian;
//                              replaces 8 launches and ~300 B/Gaussian of intermediate traffic);
//   * project_fused_backward : packed gradient -> six compact parameter gradients in ONE pass (recomputes the
//                              cheap forward chain from the raw parameters instead of saving intermediates);
//   * adam_multi_kernel      : all six parameter groups (59 rows) in one launch;
//   * lg_fused_stage1/2, lg_fused_backward : C entry points that enqueue the whole forward / backward
//                              sequence on the caller's stream using one workspace arena (no allocation, no
//                              host sync; GPU-driven sizes exactly as the reference's feedback protocol).
// All arithmetic goes through lg_chain.h / the binning walk, i.e. it is the same arithmetic as the
// op-by-op kernels.  Compiled with -ffp-contract=off.
#include "lg_common.h"
#include "lg_chain.h"
#include "lg_gaussian_bwd.h"
#include "lg_tilewalk.h"
#include "litegs_hip.h"
#include "lg_binning_internal.h"

#define REC 16
#define GREC 16

// How each tile's list gets its depth order.  GLOBAL (default): the reference's structure (wrapper.py:739-745): stable
// depth sort of all visible splats first (4 radix passes), emission in that order, stable tile sort.  TILE: no sort over the splats --
// instances are emitted in splat-id order, the stable tile sort groups them and tilesort.hip sorts every tile's list by (depth, id).
// Both give the same table bit for bit (tests/test_gpu_tilesort.py) -- except when a table was under-predicted: the truncation of
// GR/binning.cu:63 then drops the splats that come last in EMISSION order, the deepest ones in GLOBAL mode (the reference's behaviour),
// the highest ids in TILE mode.  Measured at 3 M @1080p (profiles/r02_margin_ab.log): TILE removes 93 us of
// splat sort per frame and pays 42 us for the per-tile sort (latency: ~10 us per wave-tile at 4 waves per SIMD) plus 32 us in the
// emission, whose adaptive small/big split and output locality were tuned for depth order (Morton neighbours are all large or all
// small) -- 0.787 vs 0.793 ms per step: a tie at this size, so GLOBAL is kept there (reference truncation semantics).
// AUTO (default): TILE for frames with at least LG_AUTO_TILE_N compacted Gaussians (N = A*S), GLOBAL below -- the splat sort grows
// with N (hist + 4 passes + gather: 93 us at 0.9 M, 215 us at 3 M) while what the tile mode pays is bounded by the emitted instances:
// at 6 M Gaussians @1080p (N = 1.75 M) TILE is 0.81 ms per step against 0.92 ms, at 10 M @1600x1200 (N = 3.0 M) 1.07 against 1.16 ms
// (forward only 0.70 against 0.88 ms); at the 3 M bench frame (N = 0.9 M) the two tie.
#define LG_DEPTH_ORDER_GLOBAL 0
#define LG_DEPTH_ORDER_TILE 1
#define LG_DEPTH_ORDER_AUTO 2
#define LG_AUTO_TILE_N 1000000
#define LG_HOT_MIN_TILES 128               // a splat with at least this many tile instances gets R = 2^k <= 64 lines, ~64 instances per line
__host__ __device__ static inline long long hot_capacity(long long N) { return N / 4 + 1024; }
#define LG_TILE_BINS_MAX (1 << 17)          // count words reserved (and cleared) per frame in workspace 1; frames with more tiles keep the radix sort

// The executor has NO process-wide state: every entry point resolves its options and its per-renderer device / pinned words from the
// caller's LgFusedCtx (include/litegs_hip.h) for the duration of the call.
//  * depth_order / tile_scatter: see above.  bound_margin_pct: how far (percent of the splats walked, at least 16 splats) beyond a tile's
//    saturation point its next depth bound lies -- a wider margin emits more instances but survives more drift of the scene between two
//    visits of a frame (litegs_amd/fast.py adapts it per frame).
//  * grad_replicas + hot_counter: gradient replicas for splats that cover many tiles (raster.hip).  The replica line counter is a
//    persistent device int owned by the renderer (the projection that assigns lines cannot also clear its own counter); it is reset by the
//    kernel that consumes the records at the end of a training step (project_backward_adam / project_fused_backward).
//  * Speculative culling (no reference counterpart).  The exactness of the depth-bound culling rests on a repeat of the frame without
//    culling whenever a bound was violated.  Enqueued unconditionally as gated launches that repeat costs eight empty dependent launches
//    (~39 us) in EVERY step.  With `poison` set, a culled training step enqueues no repeat at all: a violated bound (or a truncated culled
//    table) raises the sticky device word `poison` (mirrored into pinned host memory), every fused backward + Adam launch that finds it
//    raised returns without touching anything -- so from the failed step on NO parameter, moment or flag changes -- and every launch that
//    does run records its step number in the pinned word `applied`.  The host (litegs_amd/trainer.py) notices the mirror one or two steps
//    later, synchronises, and replays the steps after `applied` in order, the first of them unculled: the parameter sequence is exactly
//    the one the gated repeat would have produced.  Renders that are not followed by a fused Adam step (evaluation, gradient-hook data
//    parallelism) keep the gated repeat.
// Breadcrumbs for fault hunting (LITEGS_CRUMBS=1, with HIP_LAUNCH_BLOCKING=1): the name of the launch group a stage call is in and the sizes
// it runs with, printed from a SIGABRT handler -- a GPU memory access fault aborts the process from the runtime's own thread, and with
// blocking launches the group on record is the one that faulted.  Debugging aid; costs one getenv per process.
#include <csignal>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
namespace {
char g_crumb[512] = "(none)";
char g_crumb_ctx[512] = "";
int g_crumbs_on = -1;
void (*g_prev_abort)(int) = nullptr;
void crumb_abort_handler(int sig)
{
    const char* head = "\n[litegs crumbs] last launch group: ";
    (void)!write(2, head, strlen(head)); (void)!write(2, g_crumb, strlen(g_crumb));
    (void)!write(2, "\n[litegs crumbs] ", 18); (void)!write(2, g_crumb_ctx, strlen(g_crumb_ctx)); (void)!write(2, "\n", 1);
    if (g_prev_abort != nullptr && g_prev_abort != SIG_DFL && g_prev_abort != SIG_IGN) g_prev_abort(sig);
    signal(SIGABRT, SIG_DFL);
    abort();
}
bool crumbs_on()
{
    if (g_crumbs_on < 0) {
        const char* e = getenv("LITEGS_CRUMBS");
        g_crumbs_on = (e != nullptr && e[0] == '1') ? 1 : 0;
        if (g_crumbs_on) g_prev_abort = signal(SIGABRT, crumb_abort_handler);
    }
    return g_crumbs_on == 1;
}
}
#define CRUMB(name) do { if (crumbs_on()) snprintf(g_crumb, sizeof(g_crumb), "%s", name); } while (0)

struct Exec {
    int depth_order, margin_pct, tile_scatter, replicas, step_id, validate;
    int *hot_counter, *poison, *poison_host, *applied_host, *debug_words;
};
static int resolve_ctx(const LgFusedCtx* c, Exec& x)
{
    x = Exec{ LG_DEPTH_ORDER_AUTO, 100, 1, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr };
    if (c == nullptr) return 0;
    if (c->struct_bytes != (int32_t)sizeof(LgFusedCtx)) return (int)hipErrorInvalidValue;          // caller built against another header
    if (c->depth_order < 0 || c->depth_order > 2 || c->bound_margin_pct < 1 || c->bound_margin_pct > 100000) return (int)hipErrorInvalidValue;
    if (c->grad_replicas && c->hot_counter == nullptr) return (int)hipErrorInvalidValue;
    if (c->poison != nullptr && (c->poison_host == nullptr || c->applied_host == nullptr)) return (int)hipErrorInvalidValue;
    if (c->debug_validate && c->debug_words == nullptr) return (int)hipErrorInvalidValue;
    x.depth_order = c->depth_order; x.margin_pct = c->bound_margin_pct; x.tile_scatter = c->tile_scatter ? 1 : 0;
    x.replicas = c->grad_replicas ? 1 : 0; x.step_id = c->step_id; x.validate = c->debug_validate ? 1 : 0;
    x.hot_counter = c->hot_counter; x.poison = c->poison; x.poison_host = c->poison_host; x.applied_host = c->applied_host;
    x.debug_words = c->debug_words;
    return 0;
}
static bool use_tile_order(const Exec& x, long long N)
{
    return x.depth_order == LG_DEPTH_ORDER_TILE || (x.depth_order == LG_DEPTH_ORDER_AUTO && N >= LG_AUTO_TILE_N);
}
static bool use_tile_scatter(const Exec& x, long long N, int ntiles) { return x.tile_scatter && use_tile_order(x, N) && ntiles + 2 <= LG_TILE_BINS_MAX; }

// ---------------------------------------------------------------------------------------------
// Pinned host words (lg_host_words_alloc): one arena per process, never unmapped.  The device stores sizing feedback, speculation
// mirrors and exchange headers into pinned words asynchronously; handing those words out from the torch host allocator ties their
// lifetime to a Python tensor the allocator may recycle or unmap while a launch is still in flight.  Here a freed range is quarantined
// and re-issued only after a device synchronisation, and the pages themselves stay mapped for the life of the process.
// ---------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>
#include <unordered_map>
namespace {
struct HostRange { int* p; int n; int dev; };      // dev: the device that was current when the range was handed out (its kernels store into it)
std::mutex g_hw_mutex;
std::vector<HostRange> g_hw_free, g_hw_quarantine;
std::unordered_map<int*, int> g_hw_owner;        // live range -> device
int* g_hw_chunk = nullptr;
int g_hw_chunk_left = 0;
constexpr int HW_CHUNK_WORDS = 1 << 16;          // 256 KB of pinned memory per arena chunk

// free ranges are kept coalesced: adjacent ranges (same chunk by construction: chunks are separate allocations) merge
void hw_add_free(HostRange r)
{
    for (size_t i = 0; i < g_hw_free.size();) {
        HostRange& f = g_hw_free[i];
        if (f.p + f.n == r.p) { r.p = f.p; r.n += f.n; g_hw_free[i] = g_hw_free.back(); g_hw_free.pop_back(); i = 0; continue; }
        if (r.p + r.n == f.p) { r.n += f.n; g_hw_free[i] = g_hw_free.back(); g_hw_free.pop_back(); i = 0; continue; }
        i++;
    }
    g_hw_free.push_back(r);
}
}
LG_API int* lg_host_words_alloc(int n)
{
    if (n <= 0) return nullptr;
    n = (n + 15) & ~15;                          // 64-byte granules: no two owners share a cache line
    std::lock_guard<std::mutex> lock(g_hw_mutex);
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) cur = -1;
    auto take_free = [&]() -> int* {
        for (size_t i = 0; i < g_hw_free.size(); i++)
            if (g_hw_free[i].n >= n) {
                int* p = g_hw_free[i].p;
                if (g_hw_free[i].n > n) { g_hw_free[i].p += n; g_hw_free[i].n -= n; }
                else { g_hw_free[i] = g_hw_free.back(); g_hw_free.pop_back(); }
                return p;
            }
        return nullptr;
    };
    int* p = take_free();
    if (p == nullptr && !g_hw_quarantine.empty()) {
        // nothing can still be storing into a quarantined range once everything enqueued so far ON ITS OWNER'S DEVICE has completed (the
        // arena is process-wide and Portable: ranges of several devices may sit here; devices this process never used are not touched)
        bool drained = true;
        std::vector<int> devs;
        for (const HostRange& r : g_hw_quarantine) {
            bool seen = false;
            for (int d : devs) seen = seen || d == r.dev;
            if (!seen) devs.push_back(r.dev);
        }
        for (int d : devs) {
            if (d >= 0 && d != cur) drained = drained && hipSetDevice(d) == hipSuccess;
            drained = drained && hipDeviceSynchronize() == hipSuccess;
        }
        if (cur >= 0) (void)hipSetDevice(cur);
        if (drained) {
            for (const HostRange& r : g_hw_quarantine) hw_add_free(r);
            g_hw_quarantine.clear();
            p = take_free();
        }
    }
    if (p == nullptr) {
        if (n > g_hw_chunk_left) {
            const int words = n > HW_CHUNK_WORDS ? n : HW_CHUNK_WORDS;
            void* mem = nullptr;
            if (hipHostMalloc(&mem, sizeof(int) * (size_t)words, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return nullptr;
            if (g_hw_chunk_left > 0) hw_add_free(HostRange{ g_hw_chunk, g_hw_chunk_left, -1 });      // the rest of the previous chunk stays usable
            g_hw_chunk = (int*)mem; g_hw_chunk_left = words;
        }
        p = g_hw_chunk; g_hw_chunk += n; g_hw_chunk_left -= n;
    }
    for (int i = 0; i < n; i++) p[i] = 0;
    g_hw_owner[p] = cur;
    return p;
}
LG_API void lg_host_words_free(int* words, int n)
{
    if (words == nullptr || n <= 0) return;
    n = (n + 15) & ~15;
    std::lock_guard<std::mutex> lock(g_hw_mutex);
    int dev = -1;
    auto it = g_hw_owner.find(words);
    if (it != g_hw_owner.end()) { dev = it->second; g_hw_owner.erase(it); }
    g_hw_quarantine.push_back(HostRange{ words, n, dev });
}

// ---------------------------------------------------------------------------------------------
// Always-on sanitised-word counters (lg_sanity.h): one block per translation unit that can neutralise a table word
// ---------------------------------------------------------------------------------------------
#include "lg_sanity.h"
int lg_sanity_collect_binning(int* out, int reset);
int lg_sanity_collect_tilesort(int* out, int reset);
LG_API int lg_sanitised_counts(int* out, int reset)
{
    if (out == nullptr) return (int)hipErrorInvalidValue;
    for (int i = 0; i < LG_SANITY_SITES; i++) out[i] = 0;
    int rc = lg_sanity_collect_binning(out, reset); if (rc) return rc;
    return lg_sanity_collect_tilesort(out, reset);
}

#define LOG2E 1.4426950408889634f


// ---------------------------------------------------------------------------------------------
// forward: one workgroup per allocated visible chunk, one thread per Gaussian (S <= 1024)
// outputs are SoA over N = A*S:  ndc[4,N] (rows 0,1,2 written), view_z[N], inv_cov[4,N], opacity[N],
// alloc[N] is produced by a second tiny kernel (tile walk lives in binning.hip), packed[N,16]
// ---------------------------------------------------------------------------------------------
// EARLY (A/B variant, lg_set_tuning(12, 1); DEG 3, 8x16 tiles): the SH coefficients are requested as soon as the cheap part of the fine test
// has passed, BEFORE the tile walk, so that their round trip overlaps the walk instead of following it -- at the price of 48 live
// registers across the walk.  Numbers: DESIGN.md section 9 (round 5).
int g_proj_early = 0;
int lg_fused_set_tuning(int key, int value) { if (key == 12) { g_proj_early = value ? 1 : 0; return 0; } return (int)hipErrorInvalidValue; }

template <int DEG, int TH, int TW, bool EARLY = false>
__global__ void project_fused_kernel(const int64_t* __restrict__ visible_chunk_id, const int* __restrict__ visible_chunks_num,
                                     Camera cam,
                                     const float* __restrict__ pos, const float* __restrict__ scale, const float* __restrict__ rot,
                                     const float* __restrict__ sh0, const float* __restrict__ shr, const float* __restrict__ opa,
                                     int C, int S, int A,
                                     float* __restrict__ view_z, int* __restrict__ alloc, float4* __restrict__ packed, int gx, int gy,
                                     uint32_t* __restrict__ zero_ptr, long long zero_words,
                                     uint32_t* __restrict__ zero3_ptr, long long zero3_words,
                                     const float* __restrict__ bound_pyr, const int* __restrict__ gate,
                                     int* __restrict__ hot_of /*nullable*/, int* __restrict__ hot_lines, int hot_cap)
{
    if (gate != nullptr && *gate == 0) return;          // fallback launch of the depth-bound culling that is not needed
    const int a = blockIdx.x, t = threadIdx.x;
    const size_t N = (size_t)A * S;
    const size_t i = (size_t)a * S + t;
    // zero duty: sort headers, look-back table and the big-splat counter of the kernels that follow (no fill launches), and the head
    // (bucket counts, upper bound levels) of the frame's sched block the coming blend forward fills
    for (long long z = (long long)i; z < zero_words; z += (long long)N) zero_ptr[z] = 0u;
    for (long long z = (long long)i; z < zero3_words; z += (long long)N) zero3_ptr[z] = 0u;
    if (a >= visible_chunks_num[0]) {
        alloc[i] = 0;
        view_z[i] = 3.0e38f;            // sorts last and emits nothing
        if (hot_of != nullptr) hot_of[i] = -1;
        return;
    }
    const size_t CS = (size_t)C * S;
    const size_t sd = (size_t)visible_chunk_id[a] * S + t;
    // ---- activation
    const float px = pos[sd], py = pos[CS + sd], pz = pos[2 * CS + sd];
    float s3[3], q[4];
#pragma unroll
    for (int k = 0; k < 3; k++) s3[k] = lg_act_scale(scale[k * CS + sd]);
    lg_act_quat(rot[sd], rot[CS + sd], rot[2 * CS + sd], rot[3 * CS + sd], q);
    const float o = lg_act_opacity(opa[sd]);
    // ---- projection chain
    float v[4], n[4], T9[9], j4[4], J6[6], c4[4], i4[4];
    lg_mvp(cam.V, cam.P, px, py, pz, 1.0f, v, n);
    lg_transform_matrix(q, s3, T9);
    lg_jacobian(cam.P, cam.H, cam.W, v[0], v[1], v[2], j4);
    J6[0] = j4[0]; J6[1] = 0.0f; J6[2] = 0.0f; J6[3] = j4[1]; J6[4] = j4[2]; J6[5] = j4[3];
    lg_cov2d(T9, cam.V, J6, c4);
    lg_inv2x2(c4[0], c4[1], c4[2], c4[3], i4);
    constexpr int NBE = (DEG + 1) * (DEG + 1);
    float shv[EARLY ? NBE * 3 : 1];
    if (EARLY) {                                          // the tests lg_tile_count starts with: everything it can count passes them
        const bool pre = !((n[0] < -1.3f) || (n[0] > 1.3f) || (n[1] < -1.3f) || (n[1] > 1.3f) || (v[2] <= 0.2f) || (o < 1.0f / 255)) &&
                         (i4[0] > 0) && (i4[3] > 0) && (i4[1] * i4[1] - i4[0] * i4[3] < 0);
        if (pre) {
            shv[0] = sh0[sd]; shv[1] = sh0[CS + sd]; shv[2] = sh0[2 * CS + sd];
#pragma unroll
            for (int k = 1; k < NBE; k++) {
                const float* s = shr + ((size_t)(k - 1) * 3) * CS + sd;
                shv[3 * k] = s[0]; shv[3 * k + 1] = s[CS]; shv[3 * k + 2] = s[2 * CS];
            }
        }
        asm volatile("" ::: "memory");                   // (keeps the compiler from sinking the loads into the branch that uses them)
    }
    int rect[4];
    int tiles = lg_tile_count<TH, TW>(n[0], n[1], v[2], i4[0], i4[1], i4[3], o, cam.H, cam.W, gx, gy, rect);     // a8, fused
    // ---- depth-bound culling (see "depth-bound culling" below): deeper than the saturation bound of every tile of its rectangle ->
    // keeps its count (sign bit set: the culled view of the prefix sum counts it as 0) but emits nothing
    if (bound_pyr != nullptr && tiles > 0 && v[2] > lg_bound_query(bound_pyr, gx, gy, rect[0], rect[1], rect[2], rect[3]))
        tiles |= (int)0x80000000u;
    // ---- gradient replicas (raster.hip): R = 2^k lines behind the N regular ones for a splat that many tiles will add to
    if (hot_of != nullptr) {
        int hot = -1;
        if (tiles >= LG_HOT_MIN_TILES) {
            int k = 31 - __clz(tiles >> 6);                  // ~64 instances per line
            k = k > 6 ? 6 : k;
            const int base = atomicAdd(hot_lines, 1 << k);
            if (base + (1 << k) <= hot_cap) hot = (base << 6) | k;
        }
        hot_of[i] = hot;
    }
    // ---- SH -> RGB (+0.5, no clamp).  Only for splats that are emitted: a third of the Gaussians of the visible CHUNKS fail
    // the fine test (frustum / opacity / degenerate), most of the rest are culled by depth, and the 48 SH coefficients are 76 % of
    // a Gaussian's bytes
    float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
    if (tiles > 0) {
        float cx, cy, cz, dx, dy, dz;
        lg_camera_center(cam.V, cx, cy, cz);
        lg_view_dir(px, py, pz, cx, cy, cz, dx, dy, dz);
        float b[16];
        lg_sh_basis<DEG>(dx, dy, dz, b);
        constexpr int NB = (DEG + 1) * (DEG + 1);
        if (EARLY) {
            r0 = b[0] * shv[0]; r1 = b[0] * shv[1]; r2 = b[0] * shv[2];
#pragma unroll
            for (int k = 1; k < NB; k++) { r0 += b[k] * shv[3 * k]; r1 += b[k] * shv[3 * k + 1]; r2 += b[k] * shv[3 * k + 2]; }
        } else {
            r0 = b[0] * sh0[sd]; r1 = b[0] * sh0[CS + sd]; r2 = b[0] * sh0[2 * CS + sd];
#pragma unroll
            for (int k = 1; k < NB; k++) {
                const float* s = shr + ((size_t)(k - 1) * 3) * CS + sd;
                r0 += b[k] * s[0]; r1 += b[k] * s[CS]; r2 += b[k] * s[2 * CS];
            }
        }
        r0 += 0.5f; r1 += 0.5f; r2 += 0.5f;
    }
    // ---- outputs
    view_z[i] = v[2];
    alloc[i] = tiles;
    const float ppx = (n[0] + 1.0f) * 0.5f * cam.W - 0.5f;
    const float ppy = (n[1] + 1.0f) * 0.5f * cam.H - 0.5f;
    // the 64-byte record is only ever read for splats that are emitted (key emission, blend, depth bounds): two thirds of the
    // Gaussians of the visible chunks fail the fine test or are culled by depth and skip the store
    if (tiles <= 0) return;
    float4* rec = packed + i * (REC / 4);
    rec[0] = make_float4(ppx, ppy, -0.5f * i4[0] * LOG2E, -i4[1] * LOG2E);       // layout: raster.hip
    rec[1] = make_float4(-0.5f * i4[3] * LOG2E, o, r0, r1);
    rec[2] = make_float4(r2, i4[0], i4[1], i4[3]);
    rec[3] = make_float4(v[2], n[0], n[1], lg_log2_opacity(o));                 // 12: VIEW depth (depth bounds); 13,14: ndc for the tile walk (binning.hip load_splat); 15: blend exponent offset
}

// ---------------------------------------------------------------------------------------------
// backward: packed_grad[N,16] -> compact grads d_pos[3,A,S] d_scale[3,A,S] d_rot[4,A,S] d_sh0[3,A,S]
// d_shr[R*3,A,S] d_opa[1,A,S]   (rasterize_backward's unpack + wrapper.py's chain + activate_backward)
// ---------------------------------------------------------------------------------------------
template <int DEG>
__global__ void project_fused_backward_kernel(const int64_t* __restrict__ visible_chunk_id, const int* __restrict__ visible_chunks_num,
                                              Camera cam,
                                              const float* __restrict__ pos, const float* __restrict__ scale, const float* __restrict__ rot,
                                              const float* __restrict__ opa, int C, int S, int A, int R,
                                              const float4* __restrict__ packed_grad, const float* __restrict__ grad_inv_scaler,
                                              float* __restrict__ d_pos, float* __restrict__ d_scale, float* __restrict__ d_rot,
                                              float* __restrict__ d_sh0, float* __restrict__ d_shr, float* __restrict__ d_opa,
                                              const int* __restrict__ hot_of, int* __restrict__ hot_counter)
{
    const int a = blockIdx.x, t = threadIdx.x;
    if (hot_counter != nullptr && a == 0 && t == 0) *hot_counter = 0;        // the next frame's projection assigns replica lines from 0
    const size_t AS = (size_t)A * S;
    const size_t od = (size_t)a * S + t;
    constexpr int NB = (DEG + 1) * (DEG + 1);
    if (a >= visible_chunks_num[0]) {
        for (int k = 0; k < R * 3; k++) d_shr[(size_t)k * AS + od] = 0.0f;
        return;
    }
    const size_t CS = (size_t)C * S;
    const size_t sd = (size_t)visible_chunk_id[a] * S + t;
    const float sc = grad_inv_scaler ? grad_inv_scaler[0] : 1.0f;
    GaussGrads G;
    float mom[9];
    load_moments_folded(packed_grad, od, (long long)AS, hot_of, mom);
    gaussian_backward<DEG>(cam, mom, sc, pos[sd], pos[CS + sd], pos[2 * CS + sd],
                           scale[sd], scale[CS + sd], scale[2 * CS + sd], rot[sd], rot[CS + sd], rot[2 * CS + sd], rot[3 * CS + sd], opa[sd], G);
#pragma unroll
    for (int k = 0; k < 3; k++) { d_pos[k * AS + od] = G.pos[k]; d_scale[k * AS + od] = G.scale[k]; }
#pragma unroll
    for (int k = 0; k < 4; k++) d_rot[k * AS + od] = G.rot[k];
    d_opa[od] = G.opa;
    d_sh0[od] = G.basis[0] * G.gc[0]; d_sh0[AS + od] = G.basis[0] * G.gc[1]; d_sh0[2 * AS + od] = G.basis[0] * G.gc[2];
#pragma unroll
    for (int k = 1; k < NB; k++) {
        float* d = d_shr + ((size_t)(k - 1) * 3) * AS + od;
        d[0] = G.basis[k] * G.gc[0]; d[AS] = G.basis[k] * G.gc[1]; d[2 * AS] = G.basis[k] * G.gc[2];
    }
    for (int k = NB - 1; k < R; k++)
        for (int ch = 0; ch < 3; ch++) d_shr[((size_t)k * 3 + ch) * AS + od] = 0.0f;
}

// Same per-Gaussian backward, but the gradient never goes to HBM: each row is fed straight into the Adam update of the
// parameter it belongs to (single-GPU training, no gradient exchange).  Saves the 236 B/Gaussian gradient write and its
// re-read by the optimizer: 2188 -> 1480 B per visible Gaussian for backward+Adam.  Update rule = adam_multi_kernel's.
template <int DEG>
__global__ void project_backward_adam_kernel(const int64_t* __restrict__ visible_chunk_id, const int* __restrict__ visible_chunks_num,
                                             Camera cam, AdamRates ar, int C, int S, int A, int R,
                                             const float4* __restrict__ packed_grad, const float* __restrict__ grad_inv_scaler,
                                             float* __restrict__ pos, float* __restrict__ scale, float* __restrict__ rot,
                                             float* __restrict__ sh0, float* __restrict__ shr, float* __restrict__ opa,
                                             float* __restrict__ m_pos, float* __restrict__ m_scale, float* __restrict__ m_rot,
                                             float* __restrict__ m_sh0, float* __restrict__ m_shr, float* __restrict__ m_opa,
                                             float* __restrict__ v_pos, float* __restrict__ v_scale, float* __restrict__ v_rot,
                                             float* __restrict__ v_sh0, float* __restrict__ v_shr, float* __restrict__ v_opa,
                                             unsigned char* __restrict__ touched, const int* __restrict__ emitted,
                                             const int* __restrict__ poison, int* __restrict__ applied_host, int step_id,
                                             const int* __restrict__ hot_of, int* __restrict__ hot_counter)
{
    const int a = blockIdx.x, t = threadIdx.x;
    if (hot_counter != nullptr && a == 0 && t == 0) *hot_counter = 0;        // the next frame's projection assigns replica lines from 0
    if (poison != nullptr) {                 // speculative culling: a failed step (this one or an earlier one) -> nothing is updated
        if (*poison != 0) return;
        if (a == 0 && t == 0) __hip_atomic_store(applied_host, step_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (a >= visible_chunks_num[0]) return;
    const size_t od = (size_t)a * S + t;
    constexpr int NB = (DEG + 1) * (DEG + 1);
    const size_t CS = (size_t)C * S;
    const size_t sd = (size_t)visible_chunk_id[a] * S + t;
    const float sc = grad_inv_scaler ? grad_inv_scaler[0] : 1.0f;
    GaussGrads G;
    float mom[9];
    // emitted (nullable): the projection's tile counts of this frame (> 0: the splat was emitted; <= 0: failed the fine test or culled
    // by depth).  A splat that was not emitted was never blended: its gradient record is the zeros it was cleared to, and for the
    // no-op test below 4 bytes tell as much as the 64-byte record (two thirds of the Gaussians of the visible chunks)
    if (emitted != nullptr && touched != nullptr && emitted[od] <= 0 && touched[sd] == 0) return;
    load_moments_folded(packed_grad, od, (long long)A * S, hot_of, mom);
    // Exact skip of no-op updates.  touched[g] == 0 asserts that both Adam moments of every row of Gaussian g are (+-)0 -- it has never
    // received a gradient.  If this frame's nine blend moments are all zero as well, every parameter gradient is 0 and the update is
    // m' = b1*0 + (1-b1)*0 = 0, v' = 0, p' = p - lr*0/(sqrt(0)+eps) = p: bit for bit what is already in memory, so the 708 B of
    // parameters/moments are neither read nor written (1480 -> 65 B for such a Gaussian).  The first non-zero record sets the flag for
    // good.  The reference's semantics (Adam over all Gaussians of the visible chunks, GR/compact.cu:333-342) are untouched: Gaussians
    // with a history keep decaying their moments on zero gradients.
    if (touched != nullptr) {
        unsigned int any = 0u;
#pragma unroll
        for (int k = 0; k < 9; k++) any |= __float_as_uint(mom[k]) << 1;          // +-0 -> 0; NaN / inf count as a gradient
        if (touched[sd] == 0) {
            if (any == 0u) return;
            touched[sd] = 1;
        }
    }
    gaussian_backward<DEG>(cam, mom, sc, pos[sd], pos[CS + sd], pos[2 * CS + sd],
                           scale[sd], scale[CS + sd], scale[2 * CS + sd], rot[sd], rot[CS + sd], rot[2 * CS + sd], rot[3 * CS + sd], opa[sd], G);
    {
        const size_t o3[3] = { sd, CS + sd, 2 * CS + sd };
        const size_t o4[4] = { sd, CS + sd, 2 * CS + sd, 3 * CS + sd };
        const size_t o1[1] = { sd };
        const float gp[3] = { G.pos[0], G.pos[1], G.pos[2] }, gs[3] = { G.scale[0], G.scale[1], G.scale[2] };
        const float gr[4] = { G.rot[0], G.rot[1], G.rot[2], G.rot[3] }, go[1] = { G.opa };
        const float g0[3] = { G.basis[0] * G.gc[0], G.basis[0] * G.gc[1], G.basis[0] * G.gc[2] };
        adam_rows<3>(pos, m_pos, v_pos, o3, gp, ar.lr_pos, ar.b1, ar.b2, ar.eps);
        adam_rows<3>(scale, m_scale, v_scale, o3, gs, ar.lr_scale, ar.b1, ar.b2, ar.eps);
        adam_rows<4>(rot, m_rot, v_rot, o4, gr, ar.lr_rot, ar.b1, ar.b2, ar.eps);
        adam_rows<1>(opa, m_opa, v_opa, o1, go, ar.lr_opa, ar.b1, ar.b2, ar.eps);
        adam_rows<3>(sh0, m_sh0, v_sh0, o3, g0, ar.lr_sh0, ar.b1, ar.b2, ar.eps);
    }
    // SH rest: KB coefficients (3*KB rows, 9*KB loads) per batch
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
                g9[kk * 3 + ch] = G.basis[k0 + kk] * G.gc[ch];
            }
        adam_rows<KB * 3>(shr, m_shr, v_shr, o9, g9, ar.lr_shr, ar.b1, ar.b2, ar.eps);
    }
    for (int k = NB - 1; k < R; k++)                         // inactive SH degrees: zero gradient, moments still decay (as adamUpdate)
        for (int ch = 0; ch < 3; ch++) adam_row(shr, m_shr, v_shr, ((size_t)k * 3 + ch) * CS + sd, 0.0f, ar.lr_shr, ar.b1, ar.b2, ar.eps);
}

// ---------------------------------------------------------------------------------------------
// workspace layout (bytes, 256-aligned), identical in stage1 / stage2 / backward
// ---------------------------------------------------------------------------------------------
struct Layout1 {      // sized by N = A*S (per-Gaussian buffers)
    size_t view_z, alloc, packed, dk_a, dv_a, dk_b, dv_b, prefix, temp, total, temp_bytes;
    // [zeroed, zeroed + zero_bytes) must be zero after the projection: depth-sort header | tile-sort header | depth-sort look-back
    // table | scan look-back words | head of dup_queue (the big-splat sub-queue counters).  Cleared on the side by the culling kernel ("zero duty").
    size_t zeroed, zero_bytes, dsort_hdr, tsort_hdr, dsort_table, scan_status, dup_queue;
    // second set for the gated fallback of the depth-bound culling + its two device flags (fail flag | full total), same zeroed region
    size_t tsort_hdr2, scan_status2, dup_queue2, flags;
    // per-key instance counts of the tile scatter, for the culled run and for its gated fallback (inside the zeroed region)
    size_t tcount, tcount2;
    size_t hot_of;                        // int32[N]: replica assignment per compacted splat (-1: none)
};
struct Layout2 {      // sized by the tile-instance table length L
    size_t tk_a, tv_a, tk_b, tv_b, tsort_table, tsort_table_words, tile_start, tile_work, dup_entries, tile_cursor, total;
    size_t seg, seg_counts;                            // segmented blend backward (raster.hip LgSegments / LgSegLayout); behind everything else
    int seg_shift;
};

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static Layout1 layout1(long long N)
{
    Layout1 f;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes); return at; };
    f.view_z = take(sizeof(float) * N);
    f.alloc = take(sizeof(int) * N);
    f.packed = take(sizeof(float) * REC * N);
    f.dk_a = take(4 * N); f.dv_a = take(4 * N); f.dk_b = take(4 * N); f.dv_b = take(4 * N);
    f.prefix = take(4 * N);
    f.temp_bytes = 0; f.temp = o;
    const size_t hdr = sizeof(int) * LG_SORT_HEADER_INTS;
    f.zeroed = take(2 * hdr + 4 * (size_t)lg_radix_table_words(N, 4) + 4 * (size_t)lg_scan_status_words(N));
    f.dsort_hdr = f.zeroed; f.tsort_hdr = f.zeroed + hdr; f.dsort_table = f.zeroed + 2 * hdr;
    f.scan_status = f.dsort_table + 4 * (size_t)lg_radix_table_words(N, 4);
    f.dup_queue = take(4 * 64);                                // the 64 sub-queue counters (the entries live in workspace 2: their number depends on L)
    f.tsort_hdr2 = take(hdr);
    f.scan_status2 = take(4 * (size_t)lg_scan_status_words(N));
    f.dup_queue2 = take(4 * 64);
    f.flags = take(4 * 64);
    f.tcount = take(4 * (size_t)LG_TILE_BINS_MAX);
    f.tcount2 = take(4 * (size_t)LG_TILE_BINS_MAX);
    f.zero_bytes = f.tcount2 + 4 * (size_t)LG_TILE_BINS_MAX - f.zeroed;
    f.hot_of = take(4 * (size_t)N);
    f.total = o;
    return f;
}

static Layout2 layout2(long long L, int ntiles, long long N)
{
    Layout2 f;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes); return at; };
    f.tk_a = take(4 * (size_t)L); f.tv_a = take(4 * (size_t)L); f.tk_b = take(4 * (size_t)L); f.tv_b = take(4 * (size_t)L);
    f.tsort_table_words = (size_t)lg_radix_table_words(L, 4);
    f.tsort_table = take(4 * f.tsort_table_words);
    f.tile_start = take(sizeof(int) * ((size_t)ntiles + 2));
    f.tile_work = take(sizeof(int) * ((size_t)ntiles + 1));
    f.dup_entries = take(4 * (size_t)lg_dup_queue_entries(N, L));
    f.tile_cursor = take(sizeof(int) * ((size_t)ntiles + 2));
    // segmented blend backward: checkpoints of the lean forward (2 KB per segment boundary), the tiles' unclamped final colours, the units
    f.seg_shift = lg_raster_segment_shift();
    const LgSegLayout sl = lg_seg_layout(L, ntiles, f.seg_shift);
    f.seg = take(sl.total);
    f.seg_counts = f.seg + sl.counts;
    f.total = o;
    return f;
}

static LgSegments segments_of(char* w, const Layout2& f, long long L, int ntiles)
{
    return LgSegments{ w + f.seg, L, ntiles, f.seg_shift, 0 };
}

// where the blend kernels find the tile-grouped splat ids in workspace 2
static size_t sorted_points_offset(const Exec& x, const Layout2& f, long long N, int ntiles)
{
    if (use_tile_scatter(x, N, ntiles)) return f.tv_b;                 // tile scatter: emitted into tv_a, dropped at the cursors into tv_b
    int bits = 0;
    for (unsigned int mt = (unsigned int)ntiles; mt >>= 1;) bits++;
    bits++;
    return (lg_radix_sort_num_passes(0, bits) % 2 == 1) ? f.tv_b : f.tv_a;
}

LG_API long long lg_fused_workspace1_bytes(long long N) { return (long long)layout1(N > 0 ? N : 1).total; }

// lines (16 floats each) of the gradient accumulator of a frame with N compacted Gaussians: the N regular records + the replica region
LG_API long long lg_fused_grad_lines(long long N) { return N + hot_capacity(N); }

LG_API long long lg_fused_workspace2_bytes(long long L, long long N, int H, int W, int TH, int TW)
{
    int ntiles = ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    return (long long)layout2(L > 0 ? L : 1, ntiles, N).total;
}

LG_API long long lg_fused_cull_scratch_bytes(int chunks) { return lg_cull_scratch_bytes(chunks); }

// byte offset in workspace 2 of the tile range table int32[ntiles + 2] (valid after stage 2) -- measurement tools
LG_API long long lg_fused_tile_start_offset(long long L, long long N, int H, int W, int TH, int TW)
{
    int ntiles = ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    return (long long)layout2(L > 0 ? L : 1, ntiles, N).tile_start;
}

// byte offset in workspace 2 of the unit count of the segmented blend backward (int32; valid after a stage 2 that rendered along a tile
// list) -- tests: more units than non-empty tiles means the segments were really used
LG_API long long lg_fused_unit_count_offset(long long L, long long N, int H, int W, int TH, int TW)
{
    int ntiles = ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    return (long long)layout2(L > 0 ? L : 1, ntiles, N).seg_counts;
}

LG_API long long lg_fused_sorted_points_offset(const LgFusedCtx* ctx, long long L, long long N, int H, int W, int TH, int TW)
{
    Exec x;
    if (resolve_ctx(ctx, x)) return -1;
    int ntiles = ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    return (long long)sorted_points_offset(x, layout2(L > 0 ? L : 1, ntiles, N), N, ntiles);
}

// byte offset in workspace 1 of the exact instance total (prefix[N-1]) -- for the blocking first-visit path
LG_API long long lg_fused_total_offset(long long N) { return (long long)(layout1(N).prefix + 4 * (size_t)(N - 1)); }

// byte offset in workspace 1 of the packed splat records float[N,16] (raster.hip layout): read by the statistics hook
LG_API long long lg_fused_packed_offset(long long N) { return (long long)layout1(N).packed; }

// byte offset in workspace 1 of the replica assignment int32[N] (valid after a stage 1 that ran with option 3 on)
LG_API long long lg_fused_hot_offset(long long N) { return (long long)layout1(N).hot_of; }

// byte offset in workspace 1 of the per-Gaussian tile counts int32[N] (allocate_size, wrapper.py:726-733): read by the statistics hook
LG_API long long lg_fused_alloc_offset(long long N) { return (long long)layout1(N).alloc; }

static Camera make_camera(const float* view_host, const float* proj_host, int H, int W)
{
    Camera c;
    for (int k = 0; k < 16; k++) { c.V[k] = view_host[k]; c.P[k] = proj_host[k]; }
    c.H = H; c.W = W;
    return c;
}

// ---------------------------------------------------------------------------------------------
// Depth-bound culling (no reference counterpart; the reference emits, sorts and range-scans every tile instance).
//
// At 3 M Gaussians @1080p a tile's list holds ~720 splats of which the blend walks ~85 before every pixel is saturated: 88 % of the
// emitted instances are never read.  Each visit of a frame therefore records, per tile, a view depth a safe margin BEHIND the point
// where the tile saturated (raster.hip, end of the blend forward) and the maxima of these bounds over 2^k x 2^k tile blocks (a
// pyramid).  The next visit of the same frame drops, in the projection kernel, every splat that lies deeper than the bound of every
// tile its rectangle touches (one 2 x 2 pyramid lookup): it keeps its depth-sort slot but counts zero instances.  The lists that
// are built are exact prefixes (in depth) of the full lists up to each tile's bound, so the blend is bit-identical to the unculled
// one PROVIDED every tile saturates at or before its bound.  The forward checks exactly that per tile; a violated bound (or a table
// that turned out too short) raises a device flag, and a second, gated copy of the pipeline -- projection without culling, prefix
// sum, emission, sort, ranges, blend -- that is always enqueued but exits at once while the flag is clear, redoes the frame in
// full.  No host decision, no synchronisation; the common case pays seven empty launches.
// ---------------------------------------------------------------------------------------------
static int validate_chunk_ids(const struct Exec& x, int64_t* vis_ids, const int* vis_num, int A, int chunks, int* lock, hipStream_t s);
static int validate_zero(const struct Exec& x, const void* p, long long words, int* lock, const int* gate, int range_id, hipStream_t s);
static int validate_order(const struct Exec& x, const int32_t* ids, const int32_t* prefix, int N, int* lock, const int* gate, hipStream_t s);
#define LG_VALIDATE_LOCK_WORD 8            // int index inside Layout1::flags (cleared with the frame's scratch by the projection)
#define LG_DUP_TICKET_WORD 16              // ... group tickets of the key emission: [16] the frame's (culled) run, [17] its gated fallback

struct Scene {            // what the projection kernel reads (raw parameters + the frame's visible chunks)
    const float *pos, *scale, *rot, *sh0, *shr, *opa;
    const int64_t* vis_ids;
    const int* vis_num;
    int chunks, S, A, degree;
};

static int launch_projection(const Scene& sc, const Camera& cam, int TH, int TW, char* w, const Layout1& f, bool zero_duty,
                             const int* sched_in /*nullable: cull against its depth bounds*/, int* sched_out /*nullable: head cleared*/,
                             const int* gate, hipStream_t s, int* hot_counter /*nullable: assign gradient replica lines*/)
{
    const float* bound_pyr = reinterpret_cast<const float*>(sched_in);
    float* view_z = (float*)(w + f.view_z); float4* packed = (float4*)(w + f.packed);
    int* alloc = (int*)(w + f.alloc);
    const int gx = (cam.W + TW - 1) / TW, gy = (cam.H + TH - 1) / TH;
    const bool hot = hot_counter != nullptr;
#define LAUNCH_PF(D, A_, B_) hipLaunchKernelGGL((project_fused_kernel<D, A_, B_>), dim3(sc.A), dim3(sc.S), 0, s, sc.vis_ids, sc.vis_num, cam,   \
                                                sc.pos, sc.scale, sc.rot, sc.sh0, sc.shr, sc.opa, sc.chunks, sc.S, sc.A, view_z, alloc, packed, \
                                                gx, gy, (uint32_t*)(w + f.zeroed), zero_duty ? (long long)(f.zero_bytes / 4) : 0LL,             \
                                                (uint32_t*)sched_out, sched_out ? lg_sched_clear_words(gx, gy) : 0LL, bound_pyr, gate,     \
                                                hot ? (int*)(w + f.hot_of) : (int*)nullptr, hot_counter, (int)hot_capacity((long long)sc.A * sc.S))
#define DISPATCH_PF(A_, B_)                                                  \
    switch (sc.degree) {                                                     \
    case 0: LAUNCH_PF(0, A_, B_); break;                                     \
    case 1: LAUNCH_PF(1, A_, B_); break;                                     \
    case 2: LAUNCH_PF(2, A_, B_); break;                                     \
    case 3: LAUNCH_PF(3, A_, B_); break;                                     \
    default: return (int)hipErrorInvalidValue;                               \
    }
    if (TH == 8 && TW == 16 && sc.degree == 3 && g_proj_early) {
        hipLaunchKernelGGL((project_fused_kernel<3, 8, 16, true>), dim3(sc.A), dim3(sc.S), 0, s, sc.vis_ids, sc.vis_num, cam,
                           sc.pos, sc.scale, sc.rot, sc.sh0, sc.shr, sc.opa, sc.chunks, sc.S, sc.A, view_z, alloc, packed,
                           gx, gy, (uint32_t*)(w + f.zeroed), zero_duty ? (long long)(f.zero_bytes / 4) : 0LL,
                           (uint32_t*)sched_out, sched_out ? lg_sched_clear_words(gx, gy) : 0LL, bound_pyr, gate,
                           hot ? (int*)(w + f.hot_of) : (int*)nullptr, hot_counter, (int)hot_capacity((long long)sc.A * sc.S));
    }
    else if (TH == 8 && TW == 16) { DISPATCH_PF(8, 16) }
    else if (TH == 16 && TW == 16) { DISPATCH_PF(16, 16) }
    else if (TH == 12 && TW == 16) { DISPATCH_PF(12, 16) }
    else if (TH == 8 && TW == 8) { DISPATCH_PF(8, 8) }
    else return (int)hipErrorInvalidValue;
#undef DISPATCH_PF
#undef LAUNCH_PF
    return (int)hipGetLastError();
}

// Stage 1: cull (unless vis_ids already computed) -> fused projection + tile counts -> depth order -> prefix sums.
// view_host / proj_host are HOST copies of the 4x4 matrices (passed by value to the kernels: no device reads of them).
// Afterwards prefix[N-1] (device) is the table length; it is copied to host_feedback_total if given.
// sched_cull (nullable): this frame's sched block of its previous visit -> depth-bound culling against its bounds.
// sched_out (nullable): the block this visit's blend forward will fill; its head is cleared here.
LG_API int lg_fused_stage1(const LgFusedCtx* ctx, const float* aabb_origin, const float* aabb_ext, const float* planes_dev, int chunks,
                           const float* view_host, const float* proj_host, int H, int W, int TH, int TW, int degree,
                           const float* pos, const float* scale, const float* rot, const float* sh0, const float* shr, const float* opa, int S,
                           int do_cull, uint8_t* visibility, int* vis_num, int64_t* vis_ids, int A,
                           void* ws1, long long ws1_bytes,
                           int* host_feedback_vis, int* host_feedback_total,
                           void* cull_scratch /*nullable: lg_fused_cull_scratch_bytes(chunks), zeroed once*/, unsigned int cull_epoch,
                           const int* sched_cull, int* sched_out, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    Exec x;
    int rc = resolve_ctx(ctx, x);
    if (rc) return rc;
    if (do_cull) {
        if (cull_scratch)
            rc = lg_frustum_culling_chain(aabb_origin, aabb_ext, planes_dev, 1, chunks, visibility, vis_num, vis_ids, cull_scratch, cull_epoch,
                                          host_feedback_vis, stream);
        else
            rc = lg_frustum_culling_fb(aabb_origin, aabb_ext, planes_dev, 1, chunks, visibility, vis_num, vis_ids, host_feedback_vis, stream);
        if (rc) return rc;
    }
    if (A <= 0) return 0;
    if (S > 1024 || S <= 0) return (int)hipErrorInvalidValue;
    const long long N = (long long)A * S;
    Layout1 f = layout1(N);
    if ((long long)f.total > ws1_bytes) return (int)hipErrorInvalidValue;
    char* w = (char*)ws1;
    Camera cam = make_camera(view_host, proj_host, H, W);
    Scene sc = { pos, scale, rot, sh0, shr, opa, vis_ids, vis_num, chunks, S, A, degree };
    if (x.validate) {      // (the lock word lives in the scratch the projection clears: taken from the previous frame's state of this workspace)
        rc = validate_chunk_ids(x, vis_ids, vis_num, A, chunks, nullptr, s); if (rc) return rc;
    }
    rc = launch_projection(sc, cam, TH, TW, w, f, true, sched_cull, sched_out, nullptr, s, x.replicas ? x.hot_counter : nullptr); if (rc) return rc;
    int* vlock = (int*)(w + f.flags) + LG_VALIDATE_LOCK_WORD;
    if (x.validate) {      // code 8: the scratch the projection clears on the side (sort headers, look-back tables, tickets, counters) IS clear
        rc = validate_zero(x, w + f.zeroed, (long long)(f.zero_bytes / 4), vlock, nullptr, 1, s); if (rc) return rc;
    }
    if (use_tile_order(x, N)) {
        // no splat sort: instances are emitted in splat-id order and every tile's list is depth-sorted after the tile sort
        // (tilesort.hip).  Inclusive scan of the tile counts in id order; prefix[N-1] (the table length) also goes to the host feedback slot
        rc = lg_gather_scan_gated((const int32_t*)(w + f.alloc), (const int32_t*)nullptr, N, (int32_t*)(w + f.prefix),
                                  (uint32_t*)(w + f.scan_status), host_feedback_total, sched_cull ? 1 : 0, nullptr, nullptr, stream);
        if (rc == 0 && x.validate) rc = validate_order(x, nullptr, (const int32_t*)(w + f.prefix), (int)N, vlock, nullptr, s);
        return rc;
    }
    float* view_z = (float*)(w + f.view_z);
    rc = lg_depth_keys_hist(view_z, N, (uint32_t*)(w + f.dk_a), (uint32_t*)(w + f.dv_a), (int*)(w + f.dsort_hdr), stream); if (rc) return rc;
    // the last pass also gathers the tile counts into depth order (into the prefix buffer, scanned in place below)
    rc = lg_radix_sort_prepared((uint32_t*)(w + f.dk_a), (uint32_t*)(w + f.dv_a), (uint32_t*)(w + f.dk_b), (uint32_t*)(w + f.dv_b), N, nullptr, 0, 32,
                                (int*)(w + f.dsort_hdr), (uint32_t*)(w + f.dsort_table), (const int32_t*)(w + f.alloc), (int32_t*)(w + f.prefix),
                                stream);
    if (rc) return rc;
    // depth-ordered inclusive scan of the tile counts (culled splats count 0); prefix[N-1] (the table length) also goes to the host
    // feedback slot
    rc = lg_gather_scan_gated((const int32_t*)(w + f.prefix), (const int32_t*)nullptr, N, (int32_t*)(w + f.prefix),
                              (uint32_t*)(w + f.scan_status), host_feedback_total, sched_cull ? 1 : 0, nullptr, nullptr, stream);
    if (rc == 0 && x.validate)      // codes 10 / 11: the splat sort's output is a table of ids, the prefix sums are monotone
        rc = validate_order(x, (const int32_t*)(w + (lg_radix_sort_num_passes(0, 32) % 2 == 1 ? f.dv_b : f.dv_a)), (const int32_t*)(w + f.prefix), (int)N, vlock, nullptr, s);
    return rc;
}

static int tile_key_bits(int ntiles)
{
    int bits = 0;
    for (unsigned int mt = (unsigned int)ntiles; mt >>= 1;) bits++;
    return bits + 1;
}

// ---------------------------------------------------------------------------------------------
// Table validators (LgFusedCtx.debug_validate; debugging aid, off in production: two extra launches per frame).  Everything behind the
// key emission indexes memory with what the table holds -- a key selects a counter and a cursor, a splat id a depth word and a 64-byte
// record through the scalar path -- so one word of garbage in the table is a wild access two kernels later.  With the flag on, the
// emitted keys and the grouped table are checked on the device BEFORE anything indexes with them; an offending word is reported to the
// caller's pinned debug_words {code, where, value, bound, valid entries, -, -, reports so far} (code 1: key outside 0..tiles at table
// position `where`; 2: range of tile `where` ends at `value` beyond the valid entries; 3: splat id `value` at position `where` outside
// 0..N-1) and neutralised (key 0 / empty range / id 0), so that a long run survives to tell where it happened.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void validate_report(int* __restrict__ lock, int* __restrict__ dbg, int code, int where, int value, int bound, int n)
{
    if (atomicCAS(lock, 0, 1) != 0) return;               // first report of this frame wins (the lock word is cleared with the frame's scratch)
    const int seen = __hip_atomic_load(dbg + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dbg + 1, where, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dbg + 2, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dbg + 3, bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dbg + 4, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dbg + 7, seen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dbg + 0, code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(256) validate_keys_kernel(int32_t* __restrict__ keys, long long L, const int* __restrict__ n_dev, int ntiles,
                                                            int* __restrict__ lock, int* __restrict__ dbg, const int* __restrict__ gate, int code)
{
    if (gate != nullptr && *gate == 0) return;
    long long n = L;
    if (n_dev != nullptr && (long long)*n_dev < n) n = *n_dev;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int k = keys[i];
        if (k < 0 || k > ntiles) { validate_report(lock, dbg, code, (int)i, k, ntiles, (int)n); keys[i] = 0; }
    }
}

__global__ void __launch_bounds__(256) validate_table_kernel(int32_t* __restrict__ vals, int32_t* __restrict__ tile_start, long long L,
                                                             const int* __restrict__ n_dev, int ntiles, int N,
                                                             int* __restrict__ lock, int* __restrict__ dbg, const int* __restrict__ gate)
{
    if (gate != nullptr && *gate == 0) return;
    long long n = L;
    if (n_dev != nullptr && (long long)*n_dev < n) n = *n_dev;
    const long long stride = (long long)gridDim.x * 256, gid = (long long)blockIdx.x * 256 + threadIdx.x;
    for (long long t = gid + 1; t <= ntiles; t += stride) {
        const int start = tile_start[t], end = tile_start[t + 1];
        if (start >= 0 && end > start && (long long)end > n) { validate_report(lock, dbg, 2, (int)t, end, start, (int)n); tile_start[t] = -1; }
    }
    for (long long i = gid; i < n; i += stride) {
        const int id = vals[i];
        if ((unsigned)id >= (unsigned)N) { validate_report(lock, dbg, 3, (int)i, id, N, (int)n); vals[i] = 0; }
    }
}

// code 4: visible_chunk_id[where] = value outside 0..chunks-1 for a position the kernels will use (where < min(A, visible count))
__global__ void __launch_bounds__(256) validate_chunk_ids_kernel(int64_t* __restrict__ vis_ids, const int* __restrict__ vis_num, int A, int chunks,
                                                                 int* __restrict__ dbg)
{
    const int a = blockIdx.x * 256 + threadIdx.x;
    const int n = vis_num[0];
    if (a >= A || a >= n) return;
    const int64_t id = vis_ids[a];
    if (id < 0 || id >= (int64_t)chunks) {
        const int seen = __hip_atomic_load(dbg + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dbg + 1, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dbg + 2, (int)id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dbg + 3, chunks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dbg + 4, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dbg + 7, seen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dbg + 0, 4, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        vis_ids[a] = 0;
    }
    if (a == 0 && (n < 0 || n > chunks)) {
        __hip_atomic_store(dbg + 1, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dbg + 2, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dbg + 3, chunks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dbg + 0, 5, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static int validate_chunk_ids(const Exec& x, int64_t* vis_ids, const int* vis_num, int A, int chunks, int* /*lock*/, hipStream_t s)
{
    hipLaunchKernelGGL(validate_chunk_ids_kernel, dim3((A + 255) / 256), dim3(256), 0, s, vis_ids, vis_num, A, chunks, x.debug_words);
    return (int)hipGetLastError();
}

// code 8: a word that must be zero on entry to the kernel that follows (look-back status, ticket, counter, digit total) is not: `where` =
// word index inside the checked range, `value` = the word, `bound` = which range (1 the projection's cleared scratch in workspace 1, checked
// before anything counts into it; 2 the tile sort's look-back table in workspace 2; 3 the tile sort's tickets).  A status word that
// escaped its clear and holds a negative float has both flag bits set and is taken for a finished prefix (binning.hip ST_INC | ST_AGG).
__global__ void __launch_bounds__(256) validate_zero_kernel(const uint32_t* __restrict__ p, long long words, int* __restrict__ lock, int* __restrict__ dbg,
                                                            const int* __restrict__ gate, int range_id)
{
    if (gate != nullptr && *gate == 0) return;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < words; i += (long long)gridDim.x * 256)
        if (p[i] != 0u) validate_report(lock, dbg, 8, (int)i, (int)p[i], range_id, (int)words);
}
static int validate_zero(const Exec& x, const void* p, long long words, int* lock, const int* gate, int range_id, hipStream_t s)
{
    if (words <= 0) return 0;
    long long blocks = (words + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(validate_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t*)p, words, lock, x.debug_words, gate, range_id);
    return (int)hipGetLastError();
}
// code 10: depth-sorted splat id `value` at slot `where` outside 0..N-1 (the splat sort left a hole of stale memory); code 11: the prefix
// sums are not non-decreasing at `where` (value = prefix[where], bound = prefix[where - 1]) -- a slot with a negative share of the table
__global__ void __launch_bounds__(256) validate_order_kernel(const int32_t* __restrict__ ids /*nullable*/, const int32_t* __restrict__ prefix, int N,
                                                             int* __restrict__ lock, int* __restrict__ dbg, const int* __restrict__ gate)
{
    if (gate != nullptr && *gate == 0) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        if (ids != nullptr && (unsigned)ids[i] >= (unsigned)N) validate_report(lock, dbg, 10, i, ids[i], N, N);
        const int a = i > 0 ? prefix[i - 1] : 0, b = prefix[i];
        if (b < a) validate_report(lock, dbg, 11, i, b, a, N);
    }
}
static int validate_order(const Exec& x, const int32_t* ids, const int32_t* prefix, int N, int* lock, const int* gate, hipStream_t s)
{
    hipLaunchKernelGGL(validate_order_kernel, dim3(1024), dim3(256), 0, s, ids, prefix, N, lock, x.debug_words, gate);
    return (int)hipGetLastError();
}

static int validate_keys(const Exec& x, int32_t* keys, long long L, const int* n_dev, int ntiles, int* lock, const int* gate, hipStream_t s, int code = 1)
{
    hipLaunchKernelGGL(validate_keys_kernel, dim3(2048), dim3(256), 0, s, keys, L, n_dev, ntiles, lock, x.debug_words, gate, code);
    return (int)hipGetLastError();
}
// code 7: the radix digit totals the emission counted for pass `where` add up to `value`, not to the `bound` entries the sort will move
__global__ void __launch_bounds__(256) validate_totals_kernel(const int* __restrict__ totals, int passes, long long L, const int* __restrict__ n_dev,
                                                              int* __restrict__ lock, int* __restrict__ dbg, const int* __restrict__ gate)
{
    if (gate != nullptr && *gate == 0) return;
    long long n = L;
    if (n_dev != nullptr && (long long)*n_dev < n) n = *n_dev;
    __shared__ long long part[4];
    for (int p = 0; p < passes; p++) {
        long long v = 0;                                  // (the totals live in LG_SORT_TOTALS_COPIES interleaved copies, lg_binning_internal.h)
        for (int c = 0; c < LG_SORT_TOTALS_COPIES; c++) v += totals[c * LG_SORT_TOTALS_STRIDE + p * 256 + threadIdx.x];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
        __syncthreads();
        const long long sum = part[0] + part[1] + part[2] + part[3];
        if (threadIdx.x == 0 && sum != n) validate_report(lock, dbg, 7, p, (int)sum, (int)n, (int)n);
        __syncthreads();
    }
}
static int validate_totals(const Exec& x, const int* totals, int passes, long long L, const int* n_dev, int* lock, const int* gate, hipStream_t s)
{
    hipLaunchKernelGGL(validate_totals_kernel, dim3(1), dim3(256), 0, s, totals, passes, L, n_dev, lock, x.debug_words, gate);
    return (int)hipGetLastError();
}
static int validate_table(const Exec& x, int32_t* vals, int32_t* tile_start, long long L, const int* n_dev, int ntiles, int N, int* lock, const int* gate, hipStream_t s)
{
    hipLaunchKernelGGL(validate_table_kernel, dim3(2048), dim3(256), 0, s, vals, tile_start, L, n_dev, ntiles, N, lock, x.debug_words, gate);
    return (int)hipGetLastError();
}

// key/value emission -> stable tile sort -> tile ranges -> blend forward over the table described by `prefix`; Ls = table length this
// run is sized for (<= the capacity L of the layout).  hdr / qcount: the zeroed scratch set to use.
static int binning_and_blend(const Exec& x, char* w1, const Layout1& f1, char* w, const Layout2& f, long long N, long long L, long long Ls, int H, int W, int TH, int TW,
                             int* tsort_hdr, int* qcount, int* tcount, const int* tiles, int K, int enable_stat,
                             float* img, float* trans, short* last, int* frag_count, float* frag_weight, float* packed_grad_clear,
                             const int* order, int* tile_work, const int* sched_in, int* sched_out, int zb_check, int* fail_flag, int* fail_host, const int* gate,
                             const int* total_dev, hipStream_t s)
{
    const int ntiles = ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    const bool odd32 = lg_radix_sort_num_passes(0, 32) % 2 == 1;
    const bool tile_mode = use_tile_order(x, N);
    const void* depth_order = tile_mode ? (const void*)nullptr : (odd32 ? (w1 + f1.dv_b) : (w1 + f1.dv_a));     // nullptr: emission in splat-id order
    const int bits = tile_key_bits(ntiles);
    bool seg_counts_zeroed = false;                      // the range kernel of the global route clears the segmented backward's unit counters on the side
    // a render along a tile list leaves checkpoints and the unit list of the segmented blend backward (whether or not a backward follows:
    // lg_fused_backward decides by the same rule and must find them)
    const bool seg_on = K <= ntiles && ntiles < 65536 &&
                        lg_raster_segments_apply(1, TH, TW, enable_stat, tiles, nullptr, fail_flag ? (const void*)fail_flag : (const void*)fail_host, gate, nullptr);
    const int32_t* sorted_pts = (const int32_t*)(w + sorted_points_offset(x, f, N, ntiles));
    int rc;
    if (crumbs_on())
        snprintf(g_crumb_ctx, sizeof(g_crumb_ctx), "stage2: N=%lld L=%lld Ls=%lld ntiles=%d tile_mode=%d scatter=%d replicas=%d stat=%d tiles=%p K=%d gate=%p ws1=%p ws2=%p ws2_bytes=%zu",
                 N, L, Ls, ntiles, (int)tile_mode, (int)use_tile_scatter(x, N, ntiles), x.replicas, enable_stat, (const void*)tiles, K, (const void*)gate, (void*)w1, (void*)w, f.total);
    if (use_tile_scatter(x, N, ntiles)) {
        // TILE mode without a sort: the emitted keys are counted per key (LDS-aggregated), one workgroup turns the counts into the range
        // table and write cursors, one pass drops the values at their cursors, and the per-tile sort orders every list by (depth, id)
        CRUMB("tile route: key emission (dup_small + dup_big)");
        if (x.validate) { rc = (int)hipMemsetAsync(w + f.tk_a, 0xff, sizeof(int32_t) * (size_t)Ls, s); if (rc) return rc; }      // an entry the emission leaves out reads -1
        rc = lg_dup_emit_gated(nullptr, nullptr, nullptr, (const float*)(w1 + f1.packed),
                               (const int32_t*)(w1 + f1.prefix), depth_order, 0, 1, (int)N, H, W, TH, TW, Ls, (int32_t*)(w + f.tk_a), (int32_t*)(w + f.tv_a),
                               qcount, (uint32_t*)(w + f.dup_entries), nullptr, 0, bits, nullptr, nullptr, 0,
                               (uint32_t*)(w + f.tile_start), (long long)ntiles + 2,
                               (uint32_t*)packed_grad_clear, packed_grad_clear ? (long long)GREC * (x.replicas ? lg_fused_grad_lines(N) : N) : 0, gate, fail_flag,
                               x.validate ? x.debug_words : nullptr, (int*)(w1 + f1.flags) + (gate != nullptr ? LG_DUP_TICKET_WORD + 1 : LG_DUP_TICKET_WORD), s);
        if (rc) return rc;
        if (x.validate) { rc = validate_keys(x, (int32_t*)(w + f.tk_a), Ls, total_dev, ntiles, (int*)(w1 + f1.flags) + LG_VALIDATE_LOCK_WORD, gate, s); if (rc) return rc; }
        CRUMB("tile route: count + offsets + scatter");
        rc = lg_tile_scatter_gated((const int32_t*)(w + f.tk_a), (const int32_t*)(w + f.tv_a), Ls, total_dev, ntiles, tcount, 1, (int*)(w + f.tile_cursor),
                                   (int32_t*)(w + f.tile_start), (int32_t*)(w + f.tv_b), gate, s);
        if (rc) return rc;
        if (x.validate) { rc = validate_table(x, (int32_t*)(w + f.tv_b), (int32_t*)(w + f.tile_start), Ls, total_dev, ntiles, (int)N, (int*)(w1 + f1.flags) + LG_VALIDATE_LOCK_WORD, gate, s); if (rc) return rc; }
        CRUMB("tile route: per-tile depth sort");
        rc = lg_tile_depth_sort_gated((int32_t*)(w + f.tv_b), (const int32_t*)(w + f.tile_start), (const float*)(w1 + f1.view_z), 1, L, (int)N, ntiles,
                                      (uint32_t*)(w + f.tk_b), 1, gate, s);
        if (rc) return rc;
    } else {
    // key/value emission; on the side it counts the tile sort's radix digits (into the header the projection kernel cleared) and
    // clears the sort's look-back table.  No table memset: the bounded sort only reads the first prefix[N-1] entries, and a
    // truncated table (Ls < total) gets its tail zeroed by the first splat that does not fit.
    CRUMB("global route: key emission (dup_small + dup_big)");
    if (x.validate) { rc = (int)hipMemsetAsync(w + f.tk_a, 0xff, sizeof(int32_t) * (size_t)Ls, s); if (rc) return rc; }          // an entry the emission leaves out reads -1
    rc = lg_dup_emit_gated(nullptr, nullptr, nullptr, (const float*)(w1 + f1.packed),
                               (const int32_t*)(w1 + f1.prefix), depth_order, 0, 1, (int)N, H, W, TH, TW, Ls, (int32_t*)(w + f.tk_a), (int32_t*)(w + f.tv_a),
                               qcount, (uint32_t*)(w + f.dup_entries), tsort_hdr, 0, bits, nullptr, (uint32_t*)(w + f.tsort_table),
                               (long long)lg_radix_table_words(Ls, lg_radix_sort_num_passes(0, bits)),
                               (uint32_t*)(w + f.tile_start), (long long)ntiles + 2,
                               (uint32_t*)packed_grad_clear, packed_grad_clear ? (long long)GREC * (x.replicas ? lg_fused_grad_lines(N) : N) : 0, gate, fail_flag,
                               x.validate ? x.debug_words : nullptr, (int*)(w1 + f1.flags) + (gate != nullptr ? LG_DUP_TICKET_WORD + 1 : LG_DUP_TICKET_WORD), s);
    if (rc) return rc;
    if (x.validate) {           // emitted keys (code 1: value -1 = never written) and the digit totals counted on the side (code 7)
        int* lock = (int*)(w1 + f1.flags) + LG_VALIDATE_LOCK_WORD;
        rc = validate_keys(x, (int32_t*)(w + f.tk_a), Ls, total_dev, ntiles, lock, gate, s); if (rc) return rc;
        rc = validate_totals(x, tsort_hdr, lg_radix_sort_num_passes(0, bits), Ls, total_dev, lock, gate, s); if (rc) return rc;
        // code 8: what the sort needs zero on entry -- its look-back table (cleared by the emission's zero duty) and its tickets
        rc = validate_zero(x, w + f.tsort_table, (long long)lg_radix_table_words(Ls, lg_radix_sort_num_passes(0, bits)), lock, gate, 2, s); if (rc) return rc;
        rc = validate_zero(x, tsort_hdr + LG_SORT_TOTALS_COPIES * LG_SORT_TOTALS_STRIDE, 4, lock, gate, 3, s); if (rc) return rc;
    }
    // instance count on the device: only that many entries are sorted and range-scanned
    CRUMB("global route: tile radix sort");
    int value_bits = 1;                                  // the values are compacted Gaussian indices below N; only total_dev entries exist (no padding)
    while (value_bits < 32 && (1ll << value_bits) < N) value_bits++;
    int ranges_done = 0;                                 // (the validators read the sorted keys: they keep the range scan)
    rc = lg_radix_sort_prepared_values((uint32_t*)(w + f.tk_a), (uint32_t*)(w + f.tv_a), (uint32_t*)(w + f.tk_b), (uint32_t*)(w + f.tv_b), Ls,
                                       total_dev, 0, bits, tsort_hdr, (uint32_t*)(w + f.tsort_table), nullptr, nullptr, value_bits,
                                       x.validate ? (int32_t*)nullptr : (int32_t*)(w + f.tile_start), ntiles, x.validate ? (int*)nullptr : &ranges_done,
                                       seg_on ? (int*)(w + f.seg_counts) : (int*)nullptr, s);
    seg_counts_zeroed = seg_on && ranges_done != 0;
    if (rc) return rc;
    const bool odd = lg_radix_sort_num_passes(0, bits) % 2 == 1;
    const int32_t* sorted_keys = (const int32_t*)(w + (odd ? f.tk_b : f.tk_a));
    if (x.validate) {           // code 6: a key outside 0..tiles in the SORTED table (the sort moved stale memory)
        rc = validate_keys(x, (int32_t*)(w + (odd ? f.tk_b : f.tk_a)), Ls, total_dev, ntiles, (int*)(w1 + f1.flags) + LG_VALIDATE_LOCK_WORD, gate, s, 6);
        if (rc) return rc;
    }
    CRUMB("global route: tile ranges");
    if (!ranges_done) { rc = lg_tile_range_prefilled(sorted_keys, 1, Ls, total_dev, ntiles, (int32_t*)(w + f.tile_start), s); if (rc) return rc; }
    CRUMB("global route: validators / per-tile sort");
    if (x.validate) { rc = validate_table(x, (int32_t*)(w + (odd ? f.tv_b : f.tv_a)), (int32_t*)(w + f.tile_start), Ls, total_dev, ntiles, (int)N, (int*)(w1 + f1.flags) + LG_VALIDATE_LOCK_WORD, gate, s); if (rc) return rc; }
    if (tile_mode) {      // depth order inside every tile; scratch for lists beyond 2048: the key buffer the tile sort did not end in
        rc = lg_tile_depth_sort_gated((int32_t*)(w + (odd ? f.tv_b : f.tv_a)), (const int32_t*)(w + f.tile_start), (const float*)(w1 + f1.view_z), 1, L,
                                      (int)N, ntiles, (uint32_t*)(w + (odd ? f.tk_a : f.tk_b)), 0, gate, s);
        if (rc) return rc;
    }
    }
    CRUMB("blend forward");
    // statistic epochs: the executor's blend backward accumulates the per-splat statistics inside the gradient record (raster.hip, STAT == 2);
    // the forward then is the plain one (frag_count == NULL).  A caller that wants the forward's own counters passes the two arrays.
    LgSegments seg = segments_of(w, f, L, ntiles);
    seg.counts_zeroed = seg_counts_zeroed ? 1 : 0;
    return lg_raster_forward_segments(sorted_pts, (const int*)(w + f.tile_start), (const float*)(w1 + f1.packed), tiles, K, 1, L, (int)N, H, W, TH, TW,
                                      (enable_stat && frag_count != nullptr && frag_weight != nullptr) ? 1 : 0, img, trans, last, frag_count, frag_weight, tiles ? nullptr : order,
                                      seg_on ? (int*)(w + f.tile_work) : (tiles ? nullptr : tile_work),
                                      tiles ? nullptr : sched_in, tiles ? nullptr : sched_out, (zb_check & 1) | (x.margin_pct << 8), fail_flag, fail_host, gate,
                                      seg_on ? &seg : nullptr, s);
}

static int culling_fallback(const Exec& x, char* w1, const Layout1& f1, char* w, const Layout2& f, long long N, long long L, int H, int W, int TH, int TW,
                            float* img, float* trans, short* last, float* packed_grad_clear, const int* order, int* tile_work,
                            const int* sched_in, int* sched_out, int* host_feedback_full,
                            const float* view_host, const float* proj_host, int degree, int chunks,
                            const float* pos, const float* scale, const float* rot, const float* sh0, const float* shr, const float* opa,
                            const int64_t* vis_ids, const int* vis_num, int A, int S, hipStream_t s);

// Stage 2: key/value emission -> stable tile sort -> tile ranges -> blend forward.  L = table capacity of the layout.
// order (nullable [T]): heaviest-first tile schedule of the frame (raster.hip); order_out (nullable [T], may alias order): recomputed from
// this visit's work per tile (one more launch -- callers refresh it every few visits).  sched_in (nullable): the frame's depth-bound
// block of its previous visit (the bounds stage 1 culled with when cull_active).  sched_out (nullable): receives this visit's bounds.
// cull_active: stage 1 culled against sched_in -> the bounds are verified and the gated fallback (which needs the projection's inputs
// again) is enqueued; L_cull <= L then sizes the culled run.  host_feedback_full (nullable, pinned): receives the full table length
// when the fallback ran.
LG_API int lg_fused_stage2(const LgFusedCtx* ctx, int A, int S, long long L, int H, int W, int TH, int TW, void* ws1, long long ws1_bytes,
                           void* ws2, long long ws2_bytes, const int* tiles, int K, int enable_stat,
                           float* img, float* trans, short* last, int* frag_count, float* frag_weight,
                           float* packed_grad_clear /*nullable: [N,16] gradient accumulator of the coming backward, zeroed on the side*/,
                           const int* order, int* order_out, const int* sched_in, int* sched_out, int cull_active, long long L_cull,
                           int* host_feedback_full, const float* view_host, const float* proj_host, int degree, int chunks,
                           const float* pos, const float* scale, const float* rot, const float* sh0, const float* shr, const float* opa,
                           const int64_t* vis_ids, const int* vis_num, void* stream)
{
    if (A <= 0 || L <= 0) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    Exec x;
    { const int rcx = resolve_ctx(ctx, x); if (rcx) return rcx; }
    const long long N = (long long)A * S;
    const int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH, ntiles = gx * gy;
    Layout1 f1 = layout1(N);
    Layout2 f = layout2(L, ntiles, N);
    if ((long long)f1.total > ws1_bytes || (long long)f.total > ws2_bytes) return (int)hipErrorInvalidValue;
    char* w1 = (char*)ws1;            // stage 2 updates the tile-sort header and the big-splat queue that live in workspace 1
    char* w = (char*)ws2;
    if (tiles != nullptr || enable_stat) { sched_in = nullptr; sched_out = nullptr; order_out = nullptr; if (cull_active) return (int)hipErrorInvalidValue; }
    int* tile_work = order_out ? (int*)(w + f.tile_work) : nullptr;
    if (cull_active && (sched_in == nullptr || sched_out == nullptr)) return (int)hipErrorInvalidValue;
    const bool spec = cull_active && x.poison != nullptr;          // no gated repeat: a failure poisons the following Adam launches instead
    int* fail_flag = spec ? x.poison : (int*)(w1 + f1.flags);
    const long long Ls = (cull_active && L_cull > 0 && L_cull < L) ? L_cull : L;
    const int* total_dev = (const int*)(w1 + f1.prefix) + (N - 1);
    int rc = binning_and_blend(x, w1, f1, w, f, N, L, Ls, H, W, TH, TW, (int*)(w1 + f1.tsort_hdr), (int*)(w1 + f1.dup_queue), (int*)(w1 + f1.tcount), tiles, K, enable_stat,
                               img, trans, last, frag_count, frag_weight, packed_grad_clear, order, tile_work, sched_in, sched_out,
                               cull_active, cull_active ? fail_flag : nullptr, spec ? x.poison_host : nullptr, nullptr, total_dev, s);
    if (rc) return rc;
    if (cull_active && !spec) { rc = culling_fallback(x, w1, f1, w, f, N, L, H, W, TH, TW, img, trans, last, packed_grad_clear, order, tile_work, sched_in, sched_out,
                                              host_feedback_full, view_host, proj_host, degree, chunks, pos, scale, rot, sh0, shr, opa, vis_ids, vis_num, A, S, s);
                       if (rc) return rc; }
    if (order_out != nullptr) return lg_tile_order(tile_work, 1, ntiles, order_out, s);
    return 0;
}

static int culling_fallback(const Exec& x, char* w1, const Layout1& f1, char* w, const Layout2& f, long long N, long long L, int H, int W, int TH, int TW,
                            float* img, float* trans, short* last, float* packed_grad_clear, const int* order, int* tile_work,
                            const int* sched_in, int* sched_out, int* host_feedback_full,
                            const float* view_host, const float* proj_host, int degree, int chunks,
                            const float* pos, const float* scale, const float* rot, const float* sh0, const float* shr, const float* opa,
                            const int64_t* vis_ids, const int* vis_num, int A, int S, hipStream_t s)
{
    int* fail_flag = (int*)(w1 + f1.flags);
    int* full_total = fail_flag + 1;
    int rc;
    // the gated fallback: everything below returns at once unless the culled run raised the flag.  The projection is repeated without
    // culling (records of the culled splats carry no colour yet) and clears the head of sched_out again.
    Camera cam = make_camera(view_host, proj_host, H, W);
    Scene sc = { pos, scale, rot, sh0, shr, opa, vis_ids, vis_num, chunks, S, A, degree };
    rc = launch_projection(sc, cam, TH, TW, w1, f1, false, nullptr, sched_out, fail_flag, s, nullptr); if (rc) return rc;
    const bool odd32 = lg_radix_sort_num_passes(0, 32) % 2 == 1;
    const int32_t* depth_order = use_tile_order(x, N) ? (const int32_t*)nullptr : (const int32_t*)(odd32 ? (w1 + f1.dv_b) : (w1 + f1.dv_a));
    rc = lg_gather_scan_gated((const int32_t*)(w1 + f1.alloc), depth_order, N, (int32_t*)(w1 + f1.prefix), (uint32_t*)(w1 + f1.scan_status2),
                              host_feedback_full, 0, fail_flag, full_total, s);
    if (rc) return rc;
    return binning_and_blend(x, w1, f1, w, f, N, L, L, H, W, TH, TW, (int*)(w1 + f1.tsort_hdr2), (int*)(w1 + f1.dup_queue2), (int*)(w1 + f1.tcount2), nullptr, 0, 0,
                             img, trans, last, nullptr, nullptr, packed_grad_clear, order, tile_work, sched_in, sched_out, 0, nullptr, nullptr, fail_flag,
                             full_total, s);
}

// 1 if the culled run of the last lg_fused_stage2 on this workspace raised the fail flag (device word; read it after a sync) -- tests
LG_API long long lg_fused_flags_offset(long long N) { return (long long)layout1(N).flags; }

// Backward: blend backward (atomics into packed_grad) -> fused per-Gaussian backward -> six compact gradients.
LG_API int lg_fused_backward(const LgFusedCtx* ctx, int A, int S, long long L, int H, int W, int TH, int TW, const void* ws1, long long ws1_bytes,
                             const void* ws2, long long ws2_bytes, const float* view_host, const float* proj_host, int degree, int chunks, int R,
                             const int64_t* vis_ids, const int* vis_num,
                             const float* pos, const float* scale, const float* rot, const float* opa,
                             const int* tiles, int K, const float* final_T, const short* last, const float* d_img, const float* d_trans,
                             const float* grad_inv_scaler, int enable_stat,
                             float* packed_grad /*[N,16] scratch*/, int packed_grad_is_zero /*cleared by lg_fused_stage2*/, float* err_square_sum,
                             float* d_pos /*NULL: blend backward only (gradients consumed later by lg_fused_backward_adam)*/,
                             float* d_scale, float* d_rot, float* d_sh0, float* d_shr, float* d_opa,
                             const int* order /*nullable [T]: the frame's tile schedule*/, void* stream)
{
    if (A <= 0 || L <= 0) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    Exec x;
    { const int rcx = resolve_ctx(ctx, x); if (rcx) return rcx; }
    const long long N = (long long)A * S;
    const int ntiles = ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    Layout1 f1 = layout1(N);
    Layout2 f = layout2(L, ntiles, N);
    if ((long long)f1.total > ws1_bytes || (long long)f.total > ws2_bytes) return (int)hipErrorInvalidValue;
    const char* w1 = (const char*)ws1;
    const char* w = (const char*)ws2;
    const int32_t* sorted_pts = (const int32_t*)(w + sorted_points_offset(x, f, N, ntiles));
    int rc = 0;
    const bool hot = x.replicas != 0;           // the context holds what this frame's lg_fused_stage1 ran with
    const int* hot_of = hot ? (const int*)(w1 + f1.hot_of) : nullptr;
    if (!packed_grad_is_zero) { rc = lg_memset_async(packed_grad, 0, (long long)sizeof(float) * GREC * (hot ? lg_fused_grad_lines(N) : N), stream); if (rc) return rc; }
    // the frame's forward (lg_fused_stage2 -> binning_and_blend) left checkpoints and work units under exactly this condition
    const bool seg_on = K <= ntiles && ntiles < 65536 && lg_raster_segments_apply(1, TH, TW, enable_stat, tiles, nullptr, nullptr, nullptr, d_trans);
    const LgSegments seg = segments_of(const_cast<char*>(w), f, L, ntiles);
    rc = lg_raster_backward_segments(sorted_pts, (const int*)(w + f.tile_start), (const float*)(w1 + f1.packed), tiles, K, final_T, last, d_img, d_trans,
                                     1, L, (int)N, H, W, TH, TW, enable_stat, packed_grad, err_square_sum, nullptr, tiles ? nullptr : order,
                                     hot_of, hot ? hot_capacity(N) : 0, seg_on ? &seg : nullptr, stream);
    if (rc) return rc;
    if (d_pos == nullptr) return 0;
    Camera cam = make_camera(view_host, proj_host, H, W);
#define LAUNCH_PB(D) hipLaunchKernelGGL(project_fused_backward_kernel<D>, dim3(A), dim3(S), 0, s, vis_ids, vis_num, cam, pos, scale, rot, opa, \
                                        chunks, S, A, R, (const float4*)packed_grad, grad_inv_scaler, d_pos, d_scale, d_rot, d_sh0, d_shr, d_opa, \
                                        hot_of, hot ? x.hot_counter : (int*)nullptr)
    switch (degree) {
    case 0: LAUNCH_PB(0); break;
    case 1: LAUNCH_PB(1); break;
    case 2: LAUNCH_PB(2); break;
    case 3: LAUNCH_PB(3); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef LAUNCH_PB
    LG_RETURN_LAST();
}

// Second half of the backward fused with the optimizer: packed_grad (left by lg_fused_backward with d_pos == NULL) ->
// per-Gaussian gradients in registers -> Adam on param / exp_avg / exp_avg_sq of the visible chunks.  lr6 (host) =
// {xyz, sh_0, sh_rest, opacity, scale, rot}, the reference's group order (litegs/training/optimizer.py:80-87).
LG_API int lg_fused_backward_adam(const LgFusedCtx* ctx, const int* hot_of, int A, int S, int H, int W, const float* view_host, const float* proj_host, int degree, int chunks, int R,
                                  const int64_t* vis_ids, const int* vis_num, const float* packed_grad, const float* grad_inv_scaler,
                                  float* pos, float* scale, float* rot, float* sh0, float* shr, float* opa,
                                  float* m_pos, float* m_scale, float* m_rot, float* m_sh0, float* m_shr, float* m_opa,
                                  float* v_pos, float* v_scale, float* v_rot, float* v_sh0, float* v_shr, float* v_opa,
                                  const float* lr6, float b1, float b2, float eps,
                                  unsigned char* touched /*nullable [chunks*S]: 0 = both moments of every row of that Gaussian are zero*/,
                                  const int* emitted /*nullable [A*S]: the tile counts stage 1 left in workspace 1 (lg_fused_alloc_offset)*/, void* stream)
{
    if (A <= 0) return 0;
    Exec x;
    { const int rcx = resolve_ctx(ctx, x); if (rcx) return rcx; }
    if (hot_of != nullptr && x.hot_counter == nullptr) return (int)hipErrorInvalidValue;
    if (S > 1024 || S <= 0) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    Camera cam = make_camera(view_host, proj_host, H, W);
    AdamRates ar = { lr6[0], lr6[1], lr6[2], lr6[3], lr6[4], lr6[5], b1, b2, eps };
#define LAUNCH_PA(D) hipLaunchKernelGGL(project_backward_adam_kernel<D>, dim3(A), dim3(S), 0, s, vis_ids, vis_num, cam, ar, chunks, S, A, R, \
                                        (const float4*)packed_grad, grad_inv_scaler, pos, scale, rot, sh0, shr, opa,                          \
                                        m_pos, m_scale, m_rot, m_sh0, m_shr, m_opa, v_pos, v_scale, v_rot, v_sh0, v_shr, v_opa, touched, emitted, \
                                        (const int*)x.poison, x.applied_host, x.step_id, hot_of, hot_of ? x.hot_counter : (int*)nullptr)
    switch (degree) {
    case 0: LAUNCH_PA(0); break;
    case 1: LAUNCH_PA(1); break;
    case 2: LAUNCH_PA(2); break;
    case 3: LAUNCH_PA(3); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef LAUNCH_PA
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// Adam over several parameter groups in one launch (same update as adam_chunk_kernel_v4)
// ---------------------------------------------------------------------------------------------
#define ADAM_MAX_GROUPS 8
struct AdamGroups {
    float* param[ADAM_MAX_GROUPS];
    const float* grad[ADAM_MAX_GROUPS];
    float* m[ADAM_MAX_GROUPS];
    float* v[ADAM_MAX_GROUPS];
    float lr[ADAM_MAX_GROUPS];
    int row_start[ADAM_MAX_GROUPS + 1];      // prefix of rows (E) per group
    int ngroups;
};

__global__ void __launch_bounds__(256) adam_multi_kernel(AdamGroups G, const int64_t* __restrict__ visible_chunk_id,
                                                         const int* __restrict__ valid_length, int chunks, int A, int S, int grad_dense,
                                                         float b1, float b2, float eps)
{
    const int a = blockIdx.x;
    if (valid_length != nullptr && a >= valid_length[0]) return;
    const int quads = S >> 2;
    const int rows_per_block = 256 / quads;
    const int row = blockIdx.y * rows_per_block + threadIdx.x / quads;
    const int q = threadIdx.x % quads;
    if (row >= G.row_start[G.ngroups] || (int)threadIdx.x >= rows_per_block * quads) return;
    int g = 0;
#pragma unroll
    for (int k = 1; k < ADAM_MAX_GROUPS; k++) g += (k < G.ngroups && row >= G.row_start[k]) ? 1 : 0;
    const int e = row - G.row_start[g];
    const size_t chunk = (size_t)visible_chunk_id[a];
    const size_t po = (((size_t)e * chunks + chunk) * S) / 4 + q;
    const size_t go = grad_dense ? po : (((size_t)e * A + a) * S) / 4 + q;
    const float lr = G.lr[g];
    float4 gr = reinterpret_cast<const float4*>(G.grad[g])[go];
    float4 mm = reinterpret_cast<float4*>(G.m[g])[po];
    float4 vv = reinterpret_cast<float4*>(G.v[g])[po];
    {   // exact no-op (zero gradient on zero moments): see adam_chunk_kernel_v4, compact.hip
        const unsigned int any = (__float_as_uint(gr.x) | __float_as_uint(gr.y) | __float_as_uint(gr.z) | __float_as_uint(gr.w) |
                                  __float_as_uint(mm.x) | __float_as_uint(mm.y) | __float_as_uint(mm.z) | __float_as_uint(mm.w) |
                                  __float_as_uint(vv.x) | __float_as_uint(vv.y) | __float_as_uint(vv.z) | __float_as_uint(vv.w)) << 1;
        if (any == 0u) return;
    }
    float4 p = reinterpret_cast<float4*>(G.param[g])[po];
#define ADAM1(c)                                  \
    mm.c = b1 * mm.c + (1.0f - b1) * gr.c;        \
    vv.c = b2 * vv.c + (1.0f - b2) * gr.c * gr.c; \
    p.c += -lr * mm.c / (sqrtf(vv.c) + eps);
    ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
    reinterpret_cast<float4*>(G.param[g])[po] = p;
    reinterpret_cast<float4*>(G.m[g])[po] = mm;
    reinterpret_cast<float4*>(G.v[g])[po] = vv;
}

LG_API int lg_adam_update_multi(int ngroups, void* const* param, const void* const* grad, void* const* exp_avg, void* const* exp_avg_sq,
                                const int* rows, const float* lr, const int64_t* visible_chunk_id, const int* valid_length,
                                int chunks, int A, int S, int grad_dense, float b1, float b2, float eps, void* stream)
{
    if (ngroups <= 0 || A <= 0) return 0;
    if (ngroups > ADAM_MAX_GROUPS || S % 4 != 0 || (S / 4) > 256 || 256 % (S / 4) != 0) return (int)hipErrorInvalidValue;
    AdamGroups G;
    G.ngroups = ngroups;
    G.row_start[0] = 0;
    for (int k = 0; k < ADAM_MAX_GROUPS; k++) {
        bool on = k < ngroups;
        G.param[k] = on ? (float*)param[k] : nullptr;
        G.grad[k] = on ? (const float*)grad[k] : nullptr;
        G.m[k] = on ? (float*)exp_avg[k] : nullptr;
        G.v[k] = on ? (float*)exp_avg_sq[k] : nullptr;
        G.lr[k] = on ? lr[k] : 0.0f;
        G.row_start[k + 1] = G.row_start[k] + (on ? rows[k] : 0);
    }
    const int rows_per_block = 256 / (S / 4);
    dim3 grid(A, lg_cdiv(G.row_start[ngroups], rows_per_block));
    hipLaunchKernelGGL(adam_multi_kernel, grid, dim3(256), 0, (hipStream_t)stream, G, visible_chunk_id, valid_length, chunks, A, S, grad_dense, b1, b2, eps);
    LG_RETURN_LAST();
}
