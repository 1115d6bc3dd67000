// Fused L1 + SSIM training loss, forward and backward (caller: litegs/training/trainer.py:145).
// The reference takes this from the un-vendored submodule kemchenj/fused-ssim (.gitmodules:1-4): arithmetic
// is not pinned by the reference tree; formula assumed (see litegs_amd/loss.py):
//   loss = (1-lam) * mean|x-y| + lam * (1 - mean SSIM),  11x11 Gaussian window sigma 1.5, zero "same" padding.
// Design: one 256-thread workgroup per 32x32 output tile per channel; the 42x42 halo of x and y is staged in
// LDS once (coalesced), the separable 11-tap blur of the five moments (x, y, x^2, y^2, xy) runs LDS->LDS
// (horizontal, 8-output register sliding window) then LDS->registers (vertical, 4-output sliding window).  The forward also emits the three partial-derivative maps
// (dS/dmu1, dS/dE[x^2], dS/dE[xy]) so the backward is a second separable blur of three maps.
// HBM traffic: forward 8 B in + 12 B out, backward 20 B in + 4 B out per pixel-channel -- bandwidth bound.
// Measured split of the backward at 1080p (launches with parts switched off, 53 us in total): the halo fetch of the three derivative
// maps 25 us (12 us for the same number of perfectly coalesced loads; the rest is the 168-byte halo row that straddles three 128-byte
// lines), the pixel's own x / y 7 us, the store 4 us; both blur passes hide behind them.  Occupancy (4, 6 or 7 workgroups per CU) and the
// order of the loads do not move it: the L1 miss path is the limit, not latency.  Forward: all 14 halo loads of a thread are issued
// before the first LDS store and the two divisions are v_rcp_f32 (57 -> 48 us).
// Loss partial sums are written per workgroup and reduced in a fixed order (deterministic scalar).
#include "lg_common.h"

typedef float v2f __attribute__((ext_vector_type(2)));

#define TS 32                   // output tile edge
#define HALO 5
#define TIN (TS + 2 * HALO)     // 42
#define HSEG 8                  // horizontal pass: outputs per work item (one row segment)
#define VSEG 4                  // vertical pass: outputs per thread (one column strip); TS*TS/VSEG == 256 threads

__constant__ float c_gauss[11] = { 0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f, 0.10936068743467331f,
                                   0.21300552785396576f, 0.26601171493530273f, 0.21300552785396576f, 0.10936068743467331f,
                                   0.036000773310661316f, 0.0075987582094967365f, 0.001028380123898387f };

// Sliding-window separable blur: a work item that produces SEG consecutive outputs reads SEG+10 inputs once into registers and
// feeds each into the (up to 11) accumulators it contributes to, so the LDS read count per output drops from 11 to (SEG+10)/SEG.
// Tap order per output is t = 0..10, as in a plain 11-tap loop.
// img may be a padded raster image (row stride img_rs, plane stride img_ps) that is clamped to [0,1] on load (clamp01): this is the
// executor's "render -> clamp(0,1) -> loss" without the separate clamp launch; gt / dmaps are dense [planes][H][W].
__global__ void __launch_bounds__(256) l1_ssim_forward_kernel(const float* __restrict__ img, long long img_ps, int img_rs, int clamp01,
                                                              const float* __restrict__ gt, int H, int W,
                                                              float* __restrict__ dmaps /*[3][B*C][H][W]*/, float* __restrict__ partial /*[blocks][2]*/)
{
    // LDS: the x/y halo tiles (2 x 42 x 43 floats) and the five horizontally blurred maps (5 x 42 x 33 floats) share one buffer --
    // every row segment first pulls its 18 + 18 inputs into registers, the workgroup synchronises, and only then are the blurred
    // values written over the inputs.  27.7 KB instead of 42 KB per workgroup: one more resident workgroup per CU for a kernel whose
    // time is the latency of its load -> blur -> blur -> store chain.
    constexpr int SH_FLOATS = 4 * TIN * (TS + 1), SXY_FLOATS = 2 * TIN * (TIN + 1);
    __shared__ float lds[SH_FLOATS > SXY_FLOATS ? SH_FLOATS : SXY_FLOATS];
    float (*sx)[TIN + 1] = reinterpret_cast<float (*)[TIN + 1]>(lds);
    float (*sy)[TIN + 1] = reinterpret_cast<float (*)[TIN + 1]>(lds + TIN * (TIN + 1));
    float (*sh)[TIN][TS + 1] = reinterpret_cast<float (*)[TIN][TS + 1]>(lds);       // 4 maps: blur x, blur y, blur (x+y)^2, blur (x-y)^2
    __shared__ float red[2][4];
    const int plane_id = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const float* x = img + plane_id * img_ps;
    const float* y = gt + plane_id * plane;
    const int bx = blockIdx.x * TS, by = blockIdx.y * TS;
    const int tid = threadIdx.x;
    // halo load: all 7 x 2 loads of a thread are issued before the first LDS store (a rolled loop waits for each pair in turn and the
    // kernel's time is this latency times the number of workgroup rounds per CU)
    {
        constexpr int NLD = (TIN * TIN + 255) / 256;
        float xv[NLD], yv[NLD];
#pragma unroll
        for (int it = 0; it < NLD; it++) {
            const int k = tid + it * 256;
            const int r = k / TIN, c = k - r * TIN;
            const int gy = by + r - HALO, gx = bx + c - HALO;
            const bool in = (k < TIN * TIN) && (gy >= 0 && gy < H && gx >= 0 && gx < W);
            xv[it] = in ? x[(size_t)gy * img_rs + gx] : 0.0f;
            yv[it] = in ? y[(size_t)gy * W + gx] : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < NLD; it++) {
            const int k = tid + it * 256;
            const int r = k / TIN, c = k - r * TIN;
            if (k < TIN * TIN) {
                sx[r][c] = clamp01 ? fminf(fmaxf(xv[it], 0.0f), 1.0f) : xv[it];
                sy[r][c] = yv[it];
            }
        }
    }
    __syncthreads();
    const bool hwork = tid < TIN * (TS / HSEG);          // 168 row segments
    const int hr = tid / (TS / HSEG), hc0 = (tid % (TS / HSEG)) * HSEG;
    float xin[HSEG + 10], yin[HSEG + 10];
    float l1_sum = 0.0f;
    if (hwork) {
#pragma unroll
        for (int u = 0; u < HSEG + 10; u++) { xin[u] = sx[hr][hc0 + u]; yin[u] = sy[hr][hc0 + u]; }
        // L1 term of the segment's own 8 output pixels (centre rows only; out-of-image pixels hold x = y = 0)
        if (hr >= HALO && hr < HALO + TS) {
#pragma unroll
            for (int j = 0; j < HSEG; j++) {
                const int gy = by + hr - HALO, gx = bx + hc0 + j;
                if (gx < W && gy < H) l1_sum += fabsf(xin[j + HALO] - yin[j + HALO]);
            }
        }
    }
    __syncthreads();                                     // all inputs are in registers: the buffer may be overwritten
    if (hwork) {
        // Four blurred maps instead of five: SSIM needs E[x^2]+E[y^2] and E[xy] only, and both follow from P = blur((x+y)^2) and
        // Q = blur((x-y)^2):  E[x^2]+E[y^2] = (P+Q)/2,  E[xy] = (P-Q)/4  (the blur is linear).  -20 % of the FMAs and LDS traffic.
        // The kernel is bound by VALU issue (25 M wave instructions, 4 cycles each): the four maps are accumulated as two packed
        // 2-vectors (v_pk_fma_f32), which halves the multiply-add instructions.
        v2f a01[HSEG], a23[HSEG];
#pragma unroll
        for (int j = 0; j < HSEG; j++) { a01[j] = v2f{ 0.0f, 0.0f }; a23[j] = v2f{ 0.0f, 0.0f }; }
#pragma unroll
        for (int u = 0; u < HSEG + 10; u++) {
            const float xv = xin[u], yv = yin[u];
            const float sp = xv + yv, sm = xv - yv;
            const v2f xy = { xv, yv }, pq = { sp * sp, sm * sm };
#pragma unroll
            for (int j = 0; j < HSEG; j++) {
                const int t = u - j;
                if (t >= 0 && t <= 10) {
                    const float w = c_gauss[t];
                    a01[j] += w * xy; a23[j] += w * pq;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < HSEG; j++) {
            sh[0][hr][hc0 + j] = a01[j].x; sh[1][hr][hc0 + j] = a01[j].y; sh[2][hr][hc0 + j] = a23[j].x;
            sh[3][hr][hc0 + j] = a23[j].y;
        }
    }
    __syncthreads();
    const int tx = tid % TS, r0 = (tid / TS) * VSEG;
    v2f m01[VSEG], m23[VSEG];
#pragma unroll
    for (int j = 0; j < VSEG; j++) { m01[j] = v2f{ 0.0f, 0.0f }; m23[j] = v2f{ 0.0f, 0.0f }; }
#pragma unroll
    for (int u = 0; u < VSEG + 10; u++) {
        const v2f v01 = { sh[0][r0 + u][tx], sh[1][r0 + u][tx] }, v23 = { sh[2][r0 + u][tx], sh[3][r0 + u][tx] };
#pragma unroll
        for (int j = 0; j < VSEG; j++) {
            const int t = u - j;
            if (t >= 0 && t <= 10) {
                const float w = c_gauss[t];
                m01[j] += w * v01; m23[j] += w * v23;
            }
        }
    }
    float m[4][VSEG];
#pragma unroll
    for (int j = 0; j < VSEG; j++) { m[0][j] = m01[j].x; m[1][j] = m01[j].y; m[2][j] = m23[j].x; m[3][j] = m23[j].y; }
    const int gx = bx + tx;
    float s_sum = 0.0f;
    const size_t stride = (size_t)gridDim.z * plane;
#pragma unroll
    for (int j = 0; j < VSEG; j++) {
        const int gy = by + r0 + j;
        if (gx < W && gy < H) {
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            float mu1 = m[0][j], mu2 = m[1][j];
            float e2sum = 0.5f * (m[2][j] + m[3][j]), exy = 0.25f * (m[2][j] - m[3][j]);       // E[x^2]+E[y^2], E[xy]
            float s12 = exy - mu1 * mu2;
            float A1 = 2.0f * mu1 * mu2 + C1, A2 = 2.0f * s12 + C2;
            float B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = (e2sum - mu1 * mu1 - mu2 * mu2) + C2;
            const float rB2 = __builtin_amdgcn_rcpf(B2);               // B1, B2 >= C1, C2 > 0: v_rcp_f32 (1 ulp) instead of two IEEE divisions
            float inv = __builtin_amdgcn_rcpf(B1) * rB2;
            float s_val = A1 * A2 * inv;
            // partial derivatives of S wrt the three blurred moments that depend on x
            float dA = 2.0f * mu2 * A2 - 2.0f * mu2 * A1;                 // d(A1*A2)/dmu1
            float dB = 2.0f * mu1 * B2 - 2.0f * mu1 * B1;                 // d(B1*B2)/dmu1
            float dmu1 = (dA - s_val * dB) * inv;
            float dex2 = -s_val * rB2;                                     // d/dE[x^2] : only B2
            float dexy = 2.0f * A1 * inv;                                  // d/dE[xy]  : only A2
            size_t o = plane_id * plane + (size_t)gy * W + gx;
            dmaps[o] = dmu1;
            dmaps[stride + o] = dex2;
            dmaps[2 * stride + o] = dexy;
            s_sum += s_val;
        }
    }
    // block reduce (fixed order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s_sum += __shfl_down(s_sum, off); l1_sum += __shfl_down(l1_sum, off); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s_sum; red[1][tid >> 6] = l1_sum; }
    __syncthreads();
    if (tid == 0) {
        size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * b] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[2 * b + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// deterministic final reduction: loss = (1-lam)*sum_l1/n + lam*(1 - sum_ssim/n)
__global__ void __launch_bounds__(1024) l1_ssim_reduce_kernel(const float* __restrict__ partial, int nblocks, float inv_n, float lam,
                                                              float* __restrict__ loss)
{
    __shared__ double rs[16], rl[16];
    double s = 0.0, l = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += 1024) { s += partial[2 * k]; l += partial[2 * k + 1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off); l += __shfl_down(l, off); }
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rl[threadIdx.x >> 6] = l; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0, tl = 0;
        for (int w = 0; w < 16; w++) { ts += rs[w]; tl += rl[w]; }
        loss[0] = (float)((1.0 - lam) * tl * inv_n + lam * (1.0 - ts * inv_n));
    }
}

static int l1_ssim_forward_launch(const float* img, long long img_ps, int img_rs, int clamp01, const float* gt, int planes, int H, int W,
                                  float lam, float* dmaps, float* partial, float* loss /*NULL: the backward reduces the partial sums*/, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(lg_cdiv(W, TS), lg_cdiv(H, TS), planes);
    hipLaunchKernelGGL(l1_ssim_forward_kernel, grid, dim3(256), 0, s, img, img_ps, img_rs, clamp01, gt, H, W, dmaps, partial);
    int nblocks = grid.x * grid.y * grid.z;
    if (loss != nullptr)
        hipLaunchKernelGGL(l1_ssim_reduce_kernel, dim3(1), dim3(1024), 0, s, partial, nblocks, 1.0f / ((float)planes * H * W), lam, loss);
    LG_RETURN_LAST();
}

LG_API int lg_l1_ssim_forward(const float* img, const float* gt, int planes, int H, int W, float lam,
                              float* dmaps, float* partial, float* loss, void* stream)
{
    LG_REQUIRE(img, gt, dmaps, partial, loss);
    return l1_ssim_forward_launch(img, (long long)H * W, W, 0, gt, planes, H, W, lam, dmaps, partial, loss, stream);
}

// img: raw raster output [planes][Hp][Wp] (tile-padded); the loss is taken on clamp(img[:, :H, :W], 0, 1)
LG_API int lg_l1_ssim_forward_raster(const float* img, int Hp, int Wp, const float* gt, int planes, int H, int W, float lam,
                                     float* dmaps, float* partial, float* loss /*NULL: left to lg_l1_ssim_backward_raster_value*/, void* stream)
{
    LG_REQUIRE(img, gt, dmaps, partial);
    if (Hp < H || Wp < W) return (int)hipErrorInvalidValue;
    return l1_ssim_forward_launch(img, (long long)Hp * Wp, Wp, 1, gt, planes, H, W, lam, dmaps, partial, loss, stream);
}

LG_API long long lg_l1_ssim_partial_floats(int planes, int H, int W)
{
    return 2LL * lg_cdiv(W, TS) * lg_cdiv(H, TS) * planes;
}

// backward: d_img = g * [ lam*(-1/n) * ( blur(M1) + 2x*blur(M2) + y*blur(M3) ) + (1-lam)/n * sign(x-y) ]
// With clamp01 the gradient is taken through clamp(img, 0, 1) (zero where the raw value lies outside [0,1], as torch's clamp backward)
// and d_img has the padded raster layout [planes][Hp][Wp], padding written as 0 -- directly consumable by the blend backward.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) l1_ssim_backward_kernel(const float* __restrict__ img, long long img_ps, int img_rs, int clamp01,
                                                               const float* __restrict__ gt,
                                                               const float* __restrict__ dmaps, const float* __restrict__ grad_out,
                                                               int H, int W, int Hp, int Wp, float lam, float inv_n, float* __restrict__ d_img,
                                                               const float* __restrict__ partial /*nullable*/, int nblocks, float* __restrict__ loss)
{
    // The loss VALUE on the side (training step: nothing reads it between the forward and the backward, and a launch of its own costs
    // ~5 us of dependent-launch latency): the first workgroup sums the forward's per-workgroup partial sums exactly as
    // l1_ssim_reduce_kernel does -- same strided partial sums per (virtual) thread, same shuffle tree, same final order: same bits.
    if (partial != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
        __shared__ double rs[16], rl[16];
        for (int q = 0; q < 4; q++) {
            double sacc = 0.0, lacc = 0.0;
            for (int k = q * 256 + (int)threadIdx.x; k < nblocks; k += 1024) { sacc += partial[2 * k]; lacc += partial[2 * k + 1]; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { sacc += __shfl_down(sacc, off); lacc += __shfl_down(lacc, off); }
            if ((threadIdx.x & 63) == 0) { rs[4 * q + (threadIdx.x >> 6)] = sacc; rl[4 * q + (threadIdx.x >> 6)] = lacc; }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double ts = 0, tl = 0;
            for (int w = 0; w < 16; w++) { ts += rs[w]; tl += rl[w]; }
            loss[0] = (float)((1.0 - lam) * tl * inv_n + lam * (1.0 - ts * inv_n));
        }
    }
    // one LDS buffer for the three input halo tiles and their horizontally blurred versions (see the forward kernel)
    constexpr int SM_FLOATS = 3 * TIN * (TIN + 1), SHB_FLOATS = 3 * TIN * (TS + 1);
    __shared__ float lds[SM_FLOATS > SHB_FLOATS ? SM_FLOATS : SHB_FLOATS];
    float (*sm)[TIN][TIN + 1] = reinterpret_cast<float (*)[TIN][TIN + 1]>(lds);
    float (*sh)[TIN][TS + 1] = reinterpret_cast<float (*)[TIN][TS + 1]>(lds);
    const int plane_id = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const size_t stride = (size_t)gridDim.z * plane;
    const int bx = blockIdx.x * TS, by = blockIdx.y * TS;
    const int tid = threadIdx.x;
    // the 7 x 3 halo loads are all in flight before the first LDS store
    const int tx = tid % TS, r0 = (tid / TS) * VSEG;
    const int gx = bx + tx;
    {
        constexpr int NLD = (TIN * TIN + 255) / 256;
        float v0[NLD], v1[NLD], v2[NLD];
#pragma unroll
        for (int it = 0; it < NLD; it++) {
            const int k = tid + it * 256;
            const int r = k / TIN, c = k - r * TIN;
            const int gy = by + r - HALO, gxx = bx + c - HALO;
            const bool in = (k < TIN * TIN) && (gy >= 0 && gy < H && gxx >= 0 && gxx < W);
            const size_t o = plane_id * plane + (size_t)gy * W + gxx;
            v0[it] = in ? dmaps[o] : 0.0f;
            v1[it] = in ? dmaps[stride + o] : 0.0f;
            v2[it] = in ? dmaps[2 * stride + o] : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < NLD; it++) {
            const int k = tid + it * 256;
            const int r = k / TIN, c = k - r * TIN;
            if (k < TIN * TIN) { sm[0][r][c] = v0[it]; sm[1][r][c] = v1[it]; sm[2][r][c] = v2[it]; }
        }
    }
    __syncthreads();
    const bool hwork = tid < TIN * (TS / HSEG);
    const int hr = tid / (TS / HSEG), hc0 = (tid % (TS / HSEG)) * HSEG;
    float vin[3][HSEG + 10];
    if (hwork) {
#pragma unroll
        for (int u = 0; u < HSEG + 10; u++) { vin[0][u] = sm[0][hr][hc0 + u]; vin[1][u] = sm[1][hr][hc0 + u]; vin[2][u] = sm[2][hr][hc0 + u]; }
    }
    __syncthreads();                                     // inputs are in registers: the buffer may be overwritten
    if (hwork) {
        v2f a01[HSEG];
        float a2[HSEG];
#pragma unroll
        for (int j = 0; j < HSEG; j++) { a01[j] = v2f{ 0.0f, 0.0f }; a2[j] = 0.0f; }
#pragma unroll
        for (int u = 0; u < HSEG + 10; u++) {
            const v2f v01 = { vin[0][u], vin[1][u] };
            const float v2 = vin[2][u];
#pragma unroll
            for (int j = 0; j < HSEG; j++) {
                const int t = u - j;
                if (t >= 0 && t <= 10) { const float w = c_gauss[t]; a01[j] += w * v01; a2[j] += w * v2; }
            }
        }
#pragma unroll
        for (int j = 0; j < HSEG; j++) { sh[0][hr][hc0 + j] = a01[j].x; sh[1][hr][hc0 + j] = a01[j].y; sh[2][hr][hc0 + j] = a2[j]; }
    }
    // the pixel's own x / y are needed last: issued here (the row-segment registers are dead), they land during the vertical pass
    float xraw[VSEG], yown[VSEG];
#pragma unroll
    for (int j = 0; j < VSEG; j++) {
        const int gy = by + r0 + j;
        const bool in = gx < W && gy < H;
        xraw[j] = in ? img[plane_id * img_ps + (size_t)gy * img_rs + gx] : 0.0f;
        yown[j] = in ? gt[plane_id * plane + (size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();
    v2f b01[VSEG];
    float b2[VSEG];
#pragma unroll
    for (int j = 0; j < VSEG; j++) { b01[j] = v2f{ 0.0f, 0.0f }; b2[j] = 0.0f; }
#pragma unroll
    for (int u = 0; u < VSEG + 10; u++) {
        const v2f v01 = { sh[0][r0 + u][tx], sh[1][r0 + u][tx] };
        const float v2 = sh[2][r0 + u][tx];
#pragma unroll
        for (int j = 0; j < VSEG; j++) {
            const int t = u - j;
            if (t >= 0 && t <= 10) { const float w = c_gauss[t]; b01[j] += w * v01; b2[j] += w * v2; }
        }
    }
    float b[3][VSEG];
#pragma unroll
    for (int j = 0; j < VSEG; j++) { b[0][j] = b01[j].x; b[1][j] = b01[j].y; b[2][j] = b2[j]; }
    const float g = grad_out ? grad_out[0] : 1.0f;
#pragma unroll
    for (int j = 0; j < VSEG; j++) {
        const int gy = by + r0 + j;
        if (gx < W && gy < H) {
            const size_t oi = plane_id * img_ps + (size_t)gy * img_rs + gx;
            float xr = xraw[j], yv = yown[j];
            float xv = clamp01 ? fminf(fmaxf(xr, 0.0f), 1.0f) : xr;
            float d = xv - yv;
            float sgn = (d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f);
            float gr = g * (-lam * inv_n * (b[0][j] + 2.0f * xv * b[1][j] + yv * b[2][j]) + (1.0f - lam) * inv_n * sgn);
            if (clamp01 && !(xr >= 0.0f && xr <= 1.0f)) gr = 0.0f;
            d_img[oi] = gr;
        } else if (gx < Wp && gy < Hp) {
            d_img[plane_id * img_ps + (size_t)gy * img_rs + gx] = 0.0f;
        }
    }
}

LG_API int lg_l1_ssim_backward(const float* img, const float* gt, const float* dmaps, const float* grad_out, int planes, int H, int W,
                               float lam, float* d_img, void* stream)
{
    LG_REQUIRE(img, gt, dmaps, grad_out, d_img);
    dim3 grid(lg_cdiv(W, TS), lg_cdiv(H, TS), planes);
    hipLaunchKernelGGL(l1_ssim_backward_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, (long long)H * W, W, 0, gt, dmaps, grad_out,
                       H, W, H, W, lam, 1.0f / ((float)planes * H * W), d_img, (const float*)nullptr, 0, (float*)nullptr);
    LG_RETURN_LAST();
}

// gradient w.r.t. the RAW raster image (see lg_l1_ssim_forward_raster); d_img [planes][Hp][Wp]
LG_API int lg_l1_ssim_backward_raster_value(const float* img, int Hp, int Wp, const float* gt, const float* dmaps, const float* grad_out,
                                            int planes, int H, int W, float lam, float* d_img, const float* partial, float* loss, void* stream);

LG_API int lg_l1_ssim_backward_raster(const float* img, int Hp, int Wp, const float* gt, const float* dmaps, const float* grad_out,
                                      int planes, int H, int W, float lam, float* d_img, void* stream)
{
    return lg_l1_ssim_backward_raster_value(img, Hp, Wp, gt, dmaps, grad_out, planes, H, W, lam, d_img, nullptr, nullptr, stream);
}

// The training step's pair: lg_l1_ssim_forward_raster with loss == NULL leaves only the per-workgroup partial sums, and this backward
// also writes the loss value (partial / loss both given) -- one launch less per step; identical value (same summation order).
LG_API int lg_l1_ssim_backward_raster_value(const float* img, int Hp, int Wp, const float* gt, const float* dmaps, const float* grad_out,
                                            int planes, int H, int W, float lam, float* d_img, const float* partial, float* loss, void* stream)
{
    LG_REQUIRE(img, gt, dmaps, grad_out, d_img);
    if (Hp < H || Wp < W || ((partial == nullptr) != (loss == nullptr))) return (int)hipErrorInvalidValue;
    dim3 grid(lg_cdiv(Wp, TS), lg_cdiv(Hp, TS), planes);
    const int nblocks = lg_cdiv(W, TS) * lg_cdiv(H, TS) * planes;                  // the FORWARD's grid
    hipLaunchKernelGGL(l1_ssim_backward_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, (long long)Hp * Wp, Wp, 1, gt, dmaps, grad_out,
                       H, W, Hp, Wp, lam, 1.0f / ((float)planes * H * W), d_img, partial, nblocks, loss);
    LG_RETURN_LAST();
}
