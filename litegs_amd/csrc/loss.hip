// Fused L1 + SSIM training loss, forward and backward (caller: litegs/training/trainer.py:145).
// The reference takes this from the un-vendored submodule kemchenj/fused-ssim (.gitmodules:1-4): arithmetic
// is not pinned by the reference tree; formula assumed (see litegs_amd/loss.py):
//   loss = (1-lam) * mean|x-y| + lam * (1 - mean SSIM),  11x11 Gaussian window sigma 1.5, zero "same" padding.
// Design: one 256-thread workgroup per 16x16 output tile per channel; the 26x26 halo of x and y is staged in
// LDS once (coalesced), the separable 11-tap blur of the five moments (x, y, x^2, y^2, xy) runs LDS->LDS
// (horizontal) then LDS->registers (vertical).  The forward also emits the three partial-derivative maps
// (dS/dmu1, dS/dE[x^2], dS/dE[xy]) so the backward is a second separable blur of three maps.
// HBM traffic: forward 8 B in + 12 B out, backward 20 B in + 4 B out per pixel-channel -- bandwidth bound.
// Loss partial sums are written per workgroup and reduced in a fixed order (deterministic scalar).
#include "lg_common.h"

#define TS 16
#define HALO 5
#define TIN (TS + 2 * HALO)     // 26

__constant__ float c_gauss[11] = { 0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f, 0.10936068743467331f,
                                   0.21300552785396576f, 0.26601171493530273f, 0.21300552785396576f, 0.10936068743467331f,
                                   0.036000773310661316f, 0.0075987582094967365f, 0.001028380123898387f };

__global__ void __launch_bounds__(256) l1_ssim_forward_kernel(const float* __restrict__ img, const float* __restrict__ gt, int H, int W,
                                                              float* __restrict__ dmaps /*[3][B*C][H][W]*/, float* __restrict__ partial /*[blocks][2]*/)
{
    __shared__ float sx[TIN][TIN + 1], sy[TIN][TIN + 1];
    __shared__ float sh[5][TIN][TS + 1];
    __shared__ float red[2][4];
    const int plane_id = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const float* x = img + plane_id * plane;
    const float* y = gt + plane_id * plane;
    const int bx = blockIdx.x * TS, by = blockIdx.y * TS;
    const int tid = threadIdx.x;
    for (int k = tid; k < TIN * TIN; k += 256) {
        int r = k / TIN, c = k % TIN;
        int gy = by + r - HALO, gx = bx + c - HALO;
        bool in = (gy >= 0 && gy < H && gx >= 0 && gx < W);
        sx[r][c] = in ? x[(size_t)gy * W + gx] : 0.0f;
        sy[r][c] = in ? y[(size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();
    for (int k = tid; k < TIN * TS; k += 256) {
        int r = k / TS, c = k % TS;
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
        for (int t = 0; t < 11; t++) {
            float w = c_gauss[t], xv = sx[r][c + t], yv = sy[r][c + t];
            a0 += w * xv; a1 += w * yv; a2 += w * xv * xv; a3 += w * yv * yv; a4 += w * xv * yv;
        }
        sh[0][r][c] = a0; sh[1][r][c] = a1; sh[2][r][c] = a2; sh[3][r][c] = a3; sh[4][r][c] = a4;
    }
    __syncthreads();
    const int tx = tid % TS, ty = tid / TS;
    float mu1 = 0, mu2 = 0, ex2 = 0, ey2 = 0, exy = 0;
#pragma unroll
    for (int t = 0; t < 11; t++) {
        float w = c_gauss[t];
        mu1 += w * sh[0][ty + t][tx]; mu2 += w * sh[1][ty + t][tx]; ex2 += w * sh[2][ty + t][tx];
        ey2 += w * sh[3][ty + t][tx]; exy += w * sh[4][ty + t][tx];
    }
    const int gx = bx + tx, gy = by + ty;
    float s_val = 0.0f, l1 = 0.0f;
    if (gx < W && gy < H) {
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        float s11 = ex2 - mu1 * mu1, s22 = ey2 - mu2 * mu2, s12 = exy - mu1 * mu2;
        float A1 = 2.0f * mu1 * mu2 + C1, A2 = 2.0f * s12 + C2;
        float B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
        float inv = 1.0f / (B1 * B2);
        s_val = A1 * A2 * inv;
        // partial derivatives of S wrt the three blurred moments that depend on x
        float dA = 2.0f * mu2 * A2 - 2.0f * mu2 * A1;                 // d(A1*A2)/dmu1
        float dB = 2.0f * mu1 * B2 - 2.0f * mu1 * B1;                 // d(B1*B2)/dmu1
        float dmu1 = (dA - s_val * dB) * inv;
        float dex2 = -s_val / B2;                                      // d/dE[x^2] : only B2
        float dexy = 2.0f * A1 * inv;                                  // d/dE[xy]  : only A2
        size_t o = (size_t)gy * W + gx;
        size_t stride = (size_t)gridDim.z * plane;
        dmaps[plane_id * plane + o] = dmu1;
        dmaps[stride + plane_id * plane + o] = dex2;
        dmaps[2 * stride + plane_id * plane + o] = dexy;
        l1 = fabsf(sx[ty + HALO][tx + HALO] - sy[ty + HALO][tx + HALO]);
    }
    // block reduce (fixed order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s_val += __shfl_down(s_val, off); l1 += __shfl_down(l1, off); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s_val; red[1][tid >> 6] = l1; }
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

LG_API int lg_l1_ssim_forward(const float* img, const float* gt, int planes, int H, int W, float lam,
                              float* dmaps, float* partial, float* loss, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(lg_cdiv(W, TS), lg_cdiv(H, TS), planes);
    hipLaunchKernelGGL(l1_ssim_forward_kernel, grid, dim3(256), 0, s, img, gt, H, W, dmaps, partial);
    int nblocks = grid.x * grid.y * grid.z;
    hipLaunchKernelGGL(l1_ssim_reduce_kernel, dim3(1), dim3(1024), 0, s, partial, nblocks, 1.0f / ((float)planes * H * W), lam, loss);
    LG_RETURN_LAST();
}

LG_API long long lg_l1_ssim_partial_floats(int planes, int H, int W)
{
    return 2LL * lg_cdiv(W, TS) * lg_cdiv(H, TS) * planes;
}

// backward: d_img = g * [ lam*(-1/n) * ( blur(M1) + 2x*blur(M2) + y*blur(M3) ) + (1-lam)/n * sign(x-y) ]
__global__ void __launch_bounds__(256) l1_ssim_backward_kernel(const float* __restrict__ img, const float* __restrict__ gt,
                                                               const float* __restrict__ dmaps, const float* __restrict__ grad_out,
                                                               int H, int W, float lam, float inv_n, float* __restrict__ d_img)
{
    __shared__ float sm[3][TIN][TIN + 1];
    __shared__ float sh[3][TIN][TS + 1];
    const int plane_id = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const size_t stride = (size_t)gridDim.z * plane;
    const int bx = blockIdx.x * TS, by = blockIdx.y * TS;
    const int tid = threadIdx.x;
    for (int k = tid; k < TIN * TIN; k += 256) {
        int r = k / TIN, c = k % TIN;
        int gy = by + r - HALO, gx = bx + c - HALO;
        bool in = (gy >= 0 && gy < H && gx >= 0 && gx < W);
        size_t o = plane_id * plane + (size_t)gy * W + gx;
        sm[0][r][c] = in ? dmaps[o] : 0.0f;
        sm[1][r][c] = in ? dmaps[stride + o] : 0.0f;
        sm[2][r][c] = in ? dmaps[2 * stride + o] : 0.0f;
    }
    __syncthreads();
    for (int k = tid; k < TIN * TS; k += 256) {
        int r = k / TS, c = k % TS;
        float a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
        for (int t = 0; t < 11; t++) {
            float w = c_gauss[t];
            a0 += w * sm[0][r][c + t]; a1 += w * sm[1][r][c + t]; a2 += w * sm[2][r][c + t];
        }
        sh[0][r][c] = a0; sh[1][r][c] = a1; sh[2][r][c] = a2;
    }
    __syncthreads();
    const int tx = tid % TS, ty = tid / TS;
    const int gx = bx + tx, gy = by + ty;
    if (gx >= W || gy >= H) return;
    float b0 = 0, b1 = 0, b2 = 0;
#pragma unroll
    for (int t = 0; t < 11; t++) {
        float w = c_gauss[t];
        b0 += w * sh[0][ty + t][tx]; b1 += w * sh[1][ty + t][tx]; b2 += w * sh[2][ty + t][tx];
    }
    size_t o = plane_id * plane + (size_t)gy * W + gx;
    float xv = img[o], yv = gt[o];
    float g = grad_out ? grad_out[0] : 1.0f;
    float d = xv - yv;
    float sgn = (d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f);
    d_img[o] = g * (-lam * inv_n * (b0 + 2.0f * xv * b1 + yv * b2) + (1.0f - lam) * inv_n * sgn);
}

LG_API int lg_l1_ssim_backward(const float* img, const float* gt, const float* dmaps, const float* grad_out, int planes, int H, int W,
                               float lam, float* d_img, void* stream)
{
    dim3 grid(lg_cdiv(W, TS), lg_cdiv(H, TS), planes);
    hipLaunchKernelGGL(l1_ssim_backward_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, gt, dmaps, grad_out, H, W, lam,
                       1.0f / ((float)planes * H * W), d_img);
    LG_RETURN_LAST();
}
