// Per-tile front-to-back alpha blending, forward and backward.
//
// CDNA4 design (not a translation of GR/raster.cu):
//  * one wave64 owns one tile (8x16 = 128 px -> 2 px per lane; 16x16 -> 4; 8x8 -> 1); four independent
//    waves per 256-thread workgroup, no LDS and no barriers in the blend loops;
//  * the per-tile splat list and the 64-byte splat records are WAVE-UNIFORM, so they are fetched through
//    the scalar memory path (s_load_dwordx8/x4 into SGPRs): zero VGPRs, zero LDS bandwidth, and VALU
//    ops read the splat constants straight from SGPRs.  This is the CDNA-native replacement for the
//    "stage the splat list in shared memory" idiom of warp-32 rasterisers;
//  * fp32 blend (the reference blends in half2 with a x128 transmittance scale; 1e-4 parity needs fp32);
//    the exponent is pre-scaled by log2(e) at pack time so the inner loop is 2 FMA + v_exp_f32 per pixel;
//  * wave-level early exit through a 64-bit ballot; XCD-aware tile order (each XCD renders a contiguous
//    band of tiles so vertically adjacent tiles hit the same 4 MiB L2 for their shared splats);
//  * backward: reverse traversal, per-splat gradients reduced across the wave with a multi-value
//    butterfly (8 values in 3+3 exchange levels, DPP / ds_swizzle / permlane32) that leaves the 9
//    results in 9 different lanes, which then issue ONE coalesced global_atomic_add_f32 instruction
//    into a 64-byte-aligned gradient record (the reference issues 9 serial atomics from lane 0).
//
// Lane -> pixel map: x = lane % TW; q = lane / TW; strip = q >> 1; p = q & 1; row(k) = strip*2*PPL + 2k + p.
// This gives each lane exactly the pixel set of one reference (thread, half2-lane) pair, which is what
// the statistic-mode err_square running sum (GR/raster.cu:781-783) is defined over.
#include "lg_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#define REC 16                 // floats per packed splat record (64 B, one cache line)
#define GREC 16                // floats per packed gradient record
#define LOG2E 1.4426950408889634f

// record layout (dwords): 0 px, 1 py, 2 A2=-0.5*a*log2e, 3 B2=-b*log2e, 4 C2=-0.5*c*log2e, 5 opacity, 6 r, 7 g | 8 b,
//                          9 a=ic00, 10 b=ic01, 11 c=ic11 | 12 depth, 13..15 = 0
// forward reads dwords 0-8 (one s_load_dwordx8 + one s_load_dword), backward 0-11 (x8 + x4): tuples land in adjacent SGPRs,
// which is what lets the compiler feed v_pk_* ops straight from SGPR pairs.
// power*log2e = A2*dx^2 + B2*dx*dy + C2*dy^2   (power as in GR/raster.cu:237-240)

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
    rec[3] = make_float4(depth, 0.0f, 0.0f, 0.0f);
}

LG_API int lg_pack_forward_params(const float* ndc, const float* inv_cov, const float* color, const float* opacity,
                                  const int* valid_length, int V, int N, int H, int W, float* packed, void* stream)
{
    if (N <= 0) return 0;
    hipLaunchKernelGGL(pack_params_kernel, dim3(lg_cdiv(N, 256), V), dim3(256), 0, (hipStream_t)stream,
                       ndc, inv_cov, color, opacity, valid_length, N, H, W, (float4*)packed);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
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

template <int TH, int TW>
struct TileMap {
    static constexpr int PPL = TH * TW / 64;       // pixels per lane
    static_assert(TH * TW % 64 == 0 && 64 % TW == 0 && (64 / TW) % 2 == 0, "unsupported tile");
    static_assert(TH % (2 * PPL) == 0, "unsupported tile");
};

// ---------------------------------------------------------------------------------------------
// a13 rasterize_forward (reference: GR/raster.cu:162-332)
// ---------------------------------------------------------------------------------------------
template <int TH, int TW, bool STAT>
__global__ void __launch_bounds__(256) raster_forward_kernel(const int* __restrict__ sorted_points, const int* __restrict__ start_index,
                                                             const float* __restrict__ packed, const int* __restrict__ tiles, int K,
                                                             float* __restrict__ img, float* __restrict__ trans, short* __restrict__ last,
                                                             int* __restrict__ frag_count, float* __restrict__ frag_weight,
                                                             int gx, int ntiles, long long L, int N, int Hp, int Wp, int nslots)
{
    constexpr int PPL = TileMap<TH, TW>::PPL;
    const int lane = threadIdx.x & 63;
    const int view = blockIdx.y;
    const int nb = gridDim.x;
    int blk = (tiles == nullptr) ? xcd_remap(blockIdx.x, nb) : (int)blockIdx.x;
    const int slot = rfl(blk * 4 + (int)(threadIdx.x >> 6));
    if (slot >= nslots) return;
    int tile = (tiles != nullptr) ? tiles[(size_t)view * K + slot] : slot + 1;
    tile = rfl(tile);
    if (tile <= 0 || tile > ntiles) return;
    const int* __restrict__ si = start_index + (size_t)view * (ntiles + 2);
    const int start = rfl(si[tile]);
    const int end = rfl(si[tile + 1]);
    const int* __restrict__ sp = sorted_points + (size_t)view * L;
    const float* __restrict__ pk = packed + (size_t)view * N * REC;

    const int tx = (tile - 1) % gx, ty = (tile - 1) / gx;
    const int x = tx * TW + lane % TW;
    const int q = lane / TW;
    const int y0 = ty * TH + (q >> 1) * (2 * PPL) + (q & 1);
    const float X = (float)x;
    float Y[PPL], T[PPL], Cr[PPL], Cg[PPL], Cb[PPL];
    int lc[PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) { Y[k] = (float)(y0 + 2 * k); T[k] = 1.0f; Cr[k] = Cg[k] = Cb[k] = 0.0f; lc[k] = 0; }

    if (start >= 0) {
        for (int i = start; i < end; i++) {
            bool any_act = false;
#pragma unroll
            for (int k = 0; k < PPL; k++) any_act |= (T[k] > 1.0f / 8192);
            if (!__any(any_act)) break;
            const int pid = rfl(sp[i]);
            const float* __restrict__ r = pk + (size_t)pid * REC;
            const float spx = r[0], spy = r[1], A2 = r[2], B2 = r[3], C2 = r[4], o = r[5], cr = r[6], cg = r[7], cb = r[8];
            const float dx = spx - X;
            const float t0 = A2 * dx * dx, t1 = B2 * dx;
            int fc = 0;
            float ws = 0.0f;
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                const bool active = T[k] > 1.0f / 8192;
                const float dy = spy - Y[k];
                const float p2 = t0 + dy * (t1 + C2 * dy);
                float alpha = o * __builtin_amdgcn_exp2f(p2);
                const bool valid = active && (alpha >= 1.0f / 256);
                alpha = fminf(255.0f / 256, alpha);
                lc[k] = active ? (i - start + 1) : lc[k];       // == number of splats visited while active (activity is monotone)
                alpha = valid ? alpha : 0.0f;
                const float w = T[k] * alpha;
                if (STAT) { fc += valid ? 1 : 0; ws += w; }
                Cr[k] += cr * w; Cg[k] += cg * w; Cb[k] += cb * w;
                T[k] -= w;                       // T*(1-alpha)
            }
            if (STAT) {
                unsigned long long m = __ballot(fc != 0);
                if (m) {
                    int fct = fc;
#pragma unroll
                    for (int s = 1; s < 64; s <<= 1) fct += __shfl_xor(fct, s);
                    float wst = wave_sum(ws);
                    if (lane == 0) {
                        atomicAdd(&frag_count[(size_t)view * N + pid], fct);
                        unsafeAtomicAdd(&frag_weight[(size_t)view * N + pid], wst);
                    }
                }
            }
        }
    }
    const size_t plane = (size_t)Hp * Wp;
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const size_t o = (size_t)(y0 + 2 * k) * Wp + x;
        img[((size_t)view * 3) * plane + o] = fminf(Cr[k], 1.0f);
        img[((size_t)view * 3 + 1) * plane + o] = fminf(Cg[k], 1.0f);
        img[((size_t)view * 3 + 2) * plane + o] = fminf(Cb[k], 1.0f);
        trans[(size_t)view * plane + o] = T[k];
        last[(size_t)view * plane + o] = (short)lc[k];
    }
}

LG_API int lg_raster_forward(const int* sorted_points, const int* start_index, const float* packed, const int* tiles, int K,
                             int V, long long L, int N, int H, int W, int TH, int TW, int enable_stat,
                             float* img, float* trans, short* last, int* frag_count, float* frag_weight, void* stream)
{
    const int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    const int ntiles = gx * gy, Hp = gy * TH, Wp = gx * TW;
    const int nslots = tiles ? K : ntiles;
    if (nslots <= 0) return 0;
    dim3 grid(lg_cdiv(nslots, 4), V), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_RF(A_, B_, S_) hipLaunchKernelGGL((raster_forward_kernel<A_, B_, S_>), grid, block, 0, s, sorted_points, start_index, packed, \
                                                 tiles, K, img, trans, last, frag_count, frag_weight, gx, ntiles, L, N, Hp, Wp, nslots)
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
// a14 rasterize_backward (reference: GR/raster.cu:600-853)
// packed_grad record (GREC floats): 0 dpx, 1 dpy, 2 da, 3 db(01), 4 dc, 5 dr, 6 dg, 7 db, 8 dopacity
// ---------------------------------------------------------------------------------------------
// One butterfly level: lanes with `bit` clear keep u (own + partner's), lanes with it set keep w.
#define BFLY(u, w, bit, XOR)                      \
    {                                             \
        const float send_ = (bit) ? (u) : (w);    \
        const float keep_ = (bit) ? (w) : (u);    \
        (u) = keep_ + XOR(send_);                 \
    }

template <int TH, int TW, bool STAT, bool TRANS>
__global__ void __launch_bounds__(256) raster_backward_kernel(const int* __restrict__ sorted_points, const int* __restrict__ start_index,
                                                              const float* __restrict__ packed, const int* __restrict__ tiles, int K,
                                                              const float* __restrict__ final_T, const short* __restrict__ last,
                                                              const float* __restrict__ d_img, const float* __restrict__ d_trans,
                                                              float* __restrict__ packed_grad, float* __restrict__ err_square_sum,
                                                              int gx, int ntiles, long long L, int N, int Hp, int Wp, int nslots)
{
    constexpr int PPL = TileMap<TH, TW>::PPL;
    const int lane = threadIdx.x & 63;
    const int view = blockIdx.y;
    const int nb = gridDim.x;
    int blk = (tiles == nullptr) ? xcd_remap(blockIdx.x, nb) : (int)blockIdx.x;
    const int slot = rfl(blk * 4 + (int)(threadIdx.x >> 6));
    if (slot >= nslots) return;
    int tile = (tiles != nullptr) ? tiles[(size_t)view * K + slot] : slot + 1;
    tile = rfl(tile);
    if (tile <= 0 || tile > ntiles) return;
    const int* __restrict__ si = start_index + (size_t)view * (ntiles + 2);
    const int start = rfl(si[tile]);
    const int end = rfl(si[tile + 1]);
    if (start < 0 || start >= end) return;          // empty tile: nothing to attribute (reference bug not reproduced)
    const int* __restrict__ sp = sorted_points + (size_t)view * L + start;
    const float* __restrict__ pk = packed + (size_t)view * N * REC;
    float* __restrict__ pg = packed_grad + (size_t)view * N * GREC;

    const int tx = (tile - 1) % gx, ty = (tile - 1) / gx;
    const int x = tx * TW + lane % TW;
    const int q = lane / TW;
    const int y0 = ty * TH + (q >> 1) * (2 * PPL) + (q & 1);
    const float X = (float)x;
    const size_t plane = (size_t)Hp * Wp;
    float Y[PPL], T[PPL], Bd[PPL], gR[PPL], gG[PPL], gB[PPL], gT[PPL];
    int lc[PPL];
    int maxlast = 0;
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const size_t o = (size_t)(y0 + 2 * k) * Wp + x;
        Y[k] = (float)(y0 + 2 * k);
        T[k] = final_T[(size_t)view * plane + o];
        lc[k] = last[(size_t)view * plane + o];
        gR[k] = d_img[((size_t)view * 3) * plane + o];
        gG[k] = d_img[((size_t)view * 3 + 1) * plane + o];
        gB[k] = d_img[((size_t)view * 3 + 2) * plane + o];
        gT[k] = TRANS ? T[k] * d_trans[(size_t)view * plane + o] : 0.0f;     // T_final * dL/dT (raster.cu:665)
        Bd[k] = 0.0f;                    // (colour blended BEHIND the current splat) . dL/dC of the pixel
        maxlast = max(maxlast, lc[k]);
    }
    maxlast = rfl(wave_max_i(maxlast));
    maxlast = min(maxlast, end - start);
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;

    for (int idx = maxlast - 1; idx >= 0; idx--) {
        const int pid = rfl(sp[idx]);
        const float* __restrict__ r = pk + (size_t)pid * REC;
        const float spx = r[0], spy = r[1], A2 = r[2], B2 = r[3], C2 = r[4], o = r[5], cr = r[6], cg = r[7], cb = r[8];
        const float a = r[9], b = r[10], c = r[11];
        const float dx = spx - X;
        const float t0 = A2 * dx * dx, t1 = B2 * dx;
        float G[PPL], alpha[PPL], dy[PPL];
        bool valid[PPL];
        bool anyv = false;
#pragma unroll
        for (int k = 0; k < PPL; k++) {
            dy[k] = spy - Y[k];
            G[k] = __builtin_amdgcn_exp2f(t0 + dy[k] * (t1 + C2 * dy[k]));
            alpha[k] = fminf(255.0f / 256, o * G[k]);
            valid[k] = (alpha[k] >= 1.0f / 256) && (idx < lc[k]);
            anyv |= valid[k];
        }
        if (!__any(anyv)) continue;

        float v_r = 0.f, v_g = 0.f, v_b = 0.f, v_o = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, esq = 0.f;
        if constexpr (PPL == 2 && !STAT) {
            // the lane's two pixels as one 2-vector: packed fp32 (v_pk_fma/mul/add_f32) halves the VALU issue of this block,
            // which is what bounds the kernel (measured: ~150 VALU per (tile, splat), 8 waves/SIMD all issue-limited)
            typedef float v2f __attribute__((ext_vector_type(2)));
            const v2f am = { valid[0] ? alpha[0] : 0.0f, valid[1] ? alpha[1] : 0.0f };
            const v2f Gm = { valid[0] ? G[0] : 0.0f, valid[1] ? G[1] : 0.0f };
            const v2f om = 1.0f - am;
            const v2f rc = { __builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y) };
            v2f Tv = { T[0], T[1] };
            Tv = Tv * rc;
            Tv.x = fminf(1.0f, Tv.x); Tv.y = fminf(1.0f, Tv.y);
            T[0] = Tv.x; T[1] = Tv.y;
            const v2f gRv = { gR[0], gR[1] }, gGv = { gG[0], gG[1] }, gBv = { gB[0], gB[1] }, dyv = { dy[0], dy[1] };
            const v2f w = am * Tv;
            const v2f ar = w * gRv, ag = w * gGv, ab = w * gBv;
            const v2f cdot = cr * gRv + cg * gGv + cb * gBv;
            v2f Bdv = { Bd[0], Bd[1] };
            const v2f diff = cdot - Bdv;
            v2f d_alpha = diff * Tv;
            Bdv = Bdv + am * diff;
            Bd[0] = Bdv.x; Bd[1] = Bdv.y;
            if (TRANS) { const v2f gTv = { gT[0], gT[1] }; d_alpha = d_alpha - gTv * rc; }
            const v2f vo = d_alpha * Gm;
            const v2f dP = (Gm * o) * d_alpha;
            const v2f dPy = dP * dyv;
            const v2f dPyy = dPy * dyv;
            v_r = ar.x + ar.y; v_g = ag.x + ag.y; v_b = ab.x + ab.y; v_o = vo.x + vo.y;
            s0 = dP.x + dP.y; s1 = dPy.x + dPy.y; s2 = dPyy.x + dPyy.y;
        } else {
#pragma unroll
        for (int k = 0; k < PPL; k++) {
            if (STAT && !__any(valid[k])) continue;         // reference's per-row-group gate (raster.cu:753)
            const float am = valid[k] ? alpha[k] : 0.0f;
            const float Gm = valid[k] ? G[k] : 0.0f;
            T[k] = fminf(1.0f, T[k] * __builtin_amdgcn_rcpf(1.0f - am));
            const float w = am * T[k];
            v_r += w * gR[k]; v_g += w * gG[k]; v_b += w * gB[k];
            // reference (raster.cu:757-776) keeps the three blended-behind colours; only their dot product with the pixel's colour
            // gradient is ever used, and it obeys the same recurrence: B.g <- B.g + alpha * (c.g - B.g)
            const float cdot = cr * gR[k] + cg * gG[k] + cb * gB[k];
            const float diff = cdot - Bd[k];
            float d_alpha = diff * T[k];
            Bd[k] += am * diff;
            if (TRANS) d_alpha -= gT[k] * __builtin_amdgcn_rcpf(1.0f - am);
            v_o += d_alpha * Gm;
            if (STAT) esq += v_o * v_o;                      // running-sum quirk, raster.cu:781-783
            const float dP = Gm * o * d_alpha;
            s0 += dP; s1 += dP * dy[k]; s2 += dP * dy[k] * dy[k];
        }
        }
        // gradients of the quadratic form (GR/raster.cu:826-841 restated without forward differences)
        float v_a = -0.5f * dx * dx * s0;
        float v_bq = -0.5f * dx * s1;
        float v_c = -0.5f * s2;
        float v_px = -(a * dx * s0 + b * s1);
        float v_py = -(c * s1 + b * dx * s0);

        // 8-value butterfly: after xor 1,2,4 lane (l&7) holds one value; xor 8,16,32 complete the sums.
        BFLY(v_px, v_py, b0, xor_dpp1)
        BFLY(v_a, v_bq, b0, xor_dpp1)
        BFLY(v_c, v_r, b0, xor_dpp1)
        BFLY(v_g, v_b, b0, xor_dpp1)
        BFLY(v_px, v_a, b1, xor_dpp2)
        BFLY(v_c, v_g, b1, xor_dpp2)
        BFLY(v_px, v_c, b2, xor_swz4)
        // ninth value (opacity): reduced over xor 1,2,4 on its own, then merged into the xor-8 level -- lanes with bit 3 set carry it
        // through the last two levels, so lane 8 ends up with the opacity sum next to the eight others in lanes 0..7
        v_o += xor_dpp1(v_o); v_o += xor_dpp2(v_o); v_o += xor_swz4(v_o);
        BFLY(v_px, v_o, b3, xor_swz8)
        v_px += xor_swz16(v_px); v_px += xor_32(v_px);
        // lane (l&7) -> record slot: bit0 picks second of pair, bit1 second pair-of-pairs, bit2 second quad
        // pairs: (px,py) (a,bq) (c,r) (g,b) -> slots (0,1) (2,3) (4,5) (6,7)
        if (lane < 9) {
            const int sl = (lane == 8) ? 8 : (((lane >> 2) & 1) * 4 + ((lane >> 1) & 1) * 2 + (lane & 1));
            unsafeAtomicAdd(pg + (size_t)pid * GREC + sl, v_px);
        }
        if (STAT) {
            esq = wave_sum(esq);
            if (lane == 0) unsafeAtomicAdd(&err_square_sum[(size_t)view * N + pid], esq);
        }
    }
}

LG_API int lg_raster_backward(const int* sorted_points, const int* start_index, const float* packed, const int* tiles, int K,
                              const float* final_T, const short* last, const float* d_img, const float* d_trans,
                              int V, long long L, int N, int H, int W, int TH, int TW, int enable_stat,
                              float* packed_grad /*[V,N,16] zeroed*/, float* err_square_sum /*[V,1,N] zeroed*/, void* stream)
{
    const int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    const int ntiles = gx * gy, Hp = gy * TH, Wp = gx * TW;
    const int nslots = tiles ? K : ntiles;
    if (nslots <= 0) return 0;
    dim3 grid(lg_cdiv(nslots, 4), V), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_RB(A_, B_, S_, T_) hipLaunchKernelGGL((raster_backward_kernel<A_, B_, S_, T_>), grid, block, 0, s, sorted_points, start_index, \
                                                     packed, tiles, K, final_T, last, d_img, d_trans, packed_grad, err_square_sum,              \
                                                     gx, ntiles, L, N, Hp, Wp, nslots)
#define DISPATCH_RB(A_, B_)                                                   \
    do {                                                                      \
        if (enable_stat) { if (d_trans) LAUNCH_RB(A_, B_, true, true); else LAUNCH_RB(A_, B_, true, false); } \
        else { if (d_trans) LAUNCH_RB(A_, B_, false, true); else LAUNCH_RB(A_, B_, false, false); }           \
    } while (0)
    if (TH == 8 && TW == 16) DISPATCH_RB(8, 16);
    else if (TH == 16 && TW == 16) DISPATCH_RB(16, 16);
    else if (TH == 12 && TW == 16) DISPATCH_RB(12, 16);
    else if (TH == 8 && TW == 8) DISPATCH_RB(8, 8);
    else return (int)hipErrorInvalidValue;
#undef DISPATCH_RB
#undef LAUNCH_RB
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// unpack_gradient (reference: GR/raster.cu:855-886).  inv_scaler = *grad_inv_scaler (the reference's extra
// 1/128 undoes its fp16 transmittance scale, which does not exist here).  d_opacity is summed over views.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) unpack_gradient_kernel(const float4* __restrict__ packed_grad, const float* __restrict__ grad_inv_scaler,
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
        float4 g0 = make_float4(0, 0, 0, 0), g1 = g0;
        float g8 = 0.0f;
        if (live) {
            const float4* rec = packed_grad + ((size_t)b * N + i) * (GREC / 4);
            g0 = rec[0]; g1 = rec[1]; g8 = rec[2].x;
        }
        d_ndc[((size_t)b * 4) * N + i] = g0.x * 0.5f * W * sc;
        d_ndc[((size_t)b * 4 + 1) * N + i] = g0.y * 0.5f * H * sc;
        d_ndc[((size_t)b * 4 + 2) * N + i] = 0.0f;
        d_ndc[((size_t)b * 4 + 3) * N + i] = 0.0f;
        d_inv_cov[((size_t)b * 4) * N + i] = g0.z * sc;
        d_inv_cov[((size_t)b * 4 + 1) * N + i] = g0.w * sc;
        d_inv_cov[((size_t)b * 4 + 2) * N + i] = g0.w * sc;
        d_inv_cov[((size_t)b * 4 + 3) * N + i] = g1.x * sc;
        d_color[((size_t)b * 3) * N + i] = g1.y * sc;
        d_color[((size_t)b * 3 + 1) * N + i] = g1.z * sc;
        d_color[((size_t)b * 3 + 2) * N + i] = g1.w * sc;
        dop += g8 * sc;
    }
    d_opacity[i] = dop;
}

LG_API int lg_unpack_gradient(const float* packed_grad, const float* grad_inv_scaler, const int* valid_length, int V, int N, int H, int W,
                              float* d_ndc, float* d_inv_cov, float* d_color, float* d_opacity, void* stream)
{
    if (N <= 0) return 0;
    hipLaunchKernelGGL(unpack_gradient_kernel, dim3(lg_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)packed_grad, grad_inv_scaler, valid_length, V, N, H, W, d_ndc, d_inv_cov, d_color, d_opacity);
    LG_RETURN_LAST();
}

LG_API int lg_packed_record_floats(void) { return REC; }
LG_API int lg_packed_grad_floats(void) { return GREC; }
