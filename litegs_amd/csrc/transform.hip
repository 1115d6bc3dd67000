// Per-Gaussian projection chain, forward and backward.
// All tensors are SoA with the Gaussian index innermost ([C,N] / [V,C,N]), so one thread per Gaussian
// gives fully coalesced 256-byte wave accesses on every component; each kernel is a pure HBM stream
// (16-72 B in, 16-64 B out per Gaussian) -- no LDS, no MFMA (a 3x3.3x3.3x2 product per Gaussian is
// ~100 flops on ~100 B: bandwidth bound by >10x, see DESIGN.md "MFMA").
// Compiled with -ffp-contract=off so results are bit-comparable with the CPU oracle's op order.
#include "lg_common.h"
#include "lg_chain.h"

#define TPB 256

// ---------------------------------------------------------------------------------------------
// a3 mvp_transform_forward (reference: GR/transform.cu:378-470): view = p.V, ndc = (view.P)/w
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) mvp_forward_kernel(const float* __restrict__ world, const float* __restrict__ view,
                                                          const float* __restrict__ proj, const int* __restrict__ valid_length,
                                                          int N, float* __restrict__ view_pos, float* __restrict__ ndc_pos)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    int b = blockIdx.y;
    if (i >= lg_valid_len(valid_length, N)) return;
    const float* V = view + b * 16;
    const float* P = proj + b * 16;
    float v[4], n[4];
    lg_mvp(V, P, world[i], world[(size_t)N + i], world[2 * (size_t)N + i], world[3 * (size_t)N + i], v, n);
    size_t o = (size_t)b * 4 * N + i;
#pragma unroll
    for (int k = 0; k < 4; k++) { view_pos[o + (size_t)k * N] = v[k]; ndc_pos[o + (size_t)k * N] = n[k]; }
}

LG_API int lg_mvp_transform_forward(const float* world, const float* view, const float* proj, const int* valid_length,
                                    int V, int N, float* view_pos, float* ndc_pos, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(world, view, proj, view_pos, ndc_pos);
    dim3 grid(lg_cdiv(N, TPB), V);
    hipLaunchKernelGGL(mvp_forward_kernel, grid, dim3(TPB), 0, (hipStream_t)stream, world, view, proj, valid_length, N, view_pos, ndc_pos);
    LG_RETURN_LAST();
}

// a18 mvp_transform_backward (GR/transform.cu:472-598); no gradient to the matrices (reference TODO)
__global__ void __launch_bounds__(TPB) mvp_backward_kernel(const float* __restrict__ g_ndc, const float* __restrict__ g_view,
                                                           const float* __restrict__ view, const float* __restrict__ proj,
                                                           const float* __restrict__ view_pos, const int* __restrict__ valid_length,
                                                           int V, int N, float* __restrict__ g_world)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= lg_valid_len(valid_length, N)) return;
    float acc[4] = { 0.f, 0.f, 0.f, 0.f };
    for (int b = 0; b < V; b++) {
        size_t o = (size_t)b * 4 * N + i;
        float v[4], gn[4], gv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = view_pos[o + (size_t)k * N]; gn[k] = g_ndc[o + (size_t)k * N]; gv[k] = g_view[o + (size_t)k * N]; }
        lg_mvp_bwd(view + b * 16, proj + b * 16, v, gn, gv, acc);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) g_world[(size_t)k * N + i] = acc[k];
}

LG_API int lg_mvp_transform_backward(const float* g_ndc, const float* g_view, const float* view, const float* proj,
                                     const float* view_pos, const int* valid_length, int V, int N, float* g_world, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(g_ndc, view, proj, view_pos, g_world);
    hipLaunchKernelGGL(mvp_backward_kernel, dim3(lg_cdiv(N, TPB)), dim3(TPB), 0, (hipStream_t)stream,
                       g_ndc, g_view, view, proj, view_pos, valid_length, V, N, g_world);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// a4 createTransformMatrix_forward (GR/transform.cu:92-149): T[r][:] = R(q)[r][:] * s_r
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) transform_matrix_forward_kernel(const float* __restrict__ quat, const float* __restrict__ scale,
                                                                       const int* __restrict__ valid_length, int N, float* __restrict__ T)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= lg_valid_len(valid_length, N)) return;
    float q[4] = { quat[i], quat[(size_t)N + i], quat[2 * (size_t)N + i], quat[3 * (size_t)N + i] };
    float sc[3] = { scale[i], scale[(size_t)N + i], scale[2 * (size_t)N + i] };
    float T9[9];
    lg_transform_matrix(q, sc, T9);
#pragma unroll
    for (int k = 0; k < 9; k++) T[(size_t)k * N + i] = T9[k];
}

LG_API int lg_create_transform_matrix_forward(const float* quat, const float* scale, const int* valid_length, int N, float* T, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(quat, scale, T);
    hipLaunchKernelGGL(transform_matrix_forward_kernel, dim3(lg_cdiv(N, TPB)), dim3(TPB), 0, (hipStream_t)stream, quat, scale, valid_length, N, T);
    LG_RETURN_LAST();
}

// a17 createTransformMatrix_backward (GR/transform.cu:151-256)
__global__ void __launch_bounds__(TPB) transform_matrix_backward_kernel(const float* __restrict__ gT, const float* __restrict__ quat,
                                                                        const float* __restrict__ scale, const int* __restrict__ valid_length,
                                                                        int N, float* __restrict__ g_quat, float* __restrict__ g_scale)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= lg_valid_len(valid_length, N)) return;
    float q[4] = { quat[i], quat[(size_t)N + i], quat[2 * (size_t)N + i], quat[3 * (size_t)N + i] };
    float sc[3] = { scale[i], scale[(size_t)N + i], scale[2 * (size_t)N + i] };
    float dt[9], gq[4], gs[3];
#pragma unroll
    for (int k = 0; k < 9; k++) dt[k] = gT[(size_t)k * N + i];
    lg_transform_matrix_bwd(dt, q, sc, gq, gs);
#pragma unroll
    for (int k = 0; k < 3; k++) g_scale[(size_t)k * N + i] = gs[k];
#pragma unroll
    for (int k = 0; k < 4; k++) g_quat[(size_t)k * N + i] = gq[k];
}

LG_API int lg_create_transform_matrix_backward(const float* gT, const float* quat, const float* scale, const int* valid_length,
                                               int N, float* g_quat, float* g_scale, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(gT, quat, scale, g_quat, g_scale);
    hipLaunchKernelGGL(transform_matrix_backward_kernel, dim3(lg_cdiv(N, TPB)), dim3(TPB), 0, (hipStream_t)stream,
                       gT, quat, scale, valid_length, N, g_quat, g_scale);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// a5 jacobianRayspace (GR/transform.cu:23-90).  Writes all 9 entries (5 of them zero) so the caller
// needs no separate memset pass -- the reference does torch::zeros + a 4-entry kernel.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) jacobian_rayspace_kernel(const float* __restrict__ view_pos, const float* __restrict__ proj,
                                                                const int* __restrict__ valid_length, int N, int H, int W,
                                                                float* __restrict__ J)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    int b = blockIdx.y;
    if (i >= N) return;
    size_t jo = (size_t)b * 9 * N + i;
    if (i >= lg_valid_len(valid_length, N)) {
#pragma unroll
        for (int k = 0; k < 9; k++) J[jo + (size_t)k * N] = 0.0f;
        return;
    }
    size_t o = (size_t)b * 4 * N + i;
    float j4[4];
    lg_jacobian(proj + b * 16, H, W, view_pos[o], view_pos[o + (size_t)N], view_pos[o + 2 * (size_t)N], j4);
    J[jo] = j4[0];
    J[jo + (size_t)N] = 0.0f;
    J[jo + 2 * (size_t)N] = 0.0f;
    J[jo + 3 * (size_t)N] = 0.0f;
    J[jo + 4 * (size_t)N] = j4[1];
    J[jo + 5 * (size_t)N] = 0.0f;
    J[jo + 6 * (size_t)N] = j4[2];
    J[jo + 7 * (size_t)N] = j4[3];
    J[jo + 8 * (size_t)N] = 0.0f;
}

LG_API int lg_jacobian_rayspace(const float* view_pos, const float* proj, const int* valid_length, int V, int N, int H, int W,
                                float* J, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(view_pos, proj, J);
    hipLaunchKernelGGL(jacobian_rayspace_kernel, dim3(lg_cdiv(N, TPB), V), dim3(TPB), 0, (hipStream_t)stream,
                       view_pos, proj, valid_length, N, H, W, J);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// a6 createCov2dDirectly_forward (GR/transform.cu:737-821): M=(T.V33).J[:, :2]; cov=M^T M + 0.3 I
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) cov2d_forward_kernel(const float* __restrict__ J, const float* __restrict__ view,
                                                            const float* __restrict__ T, const int* __restrict__ valid_length,
                                                            int N, float* __restrict__ cov)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    int b = blockIdx.y;
    if (i >= lg_valid_len(valid_length, N)) return;
    float T9[9], J6[6], c4[4];
#pragma unroll
    for (int k = 0; k < 9; k++) T9[k] = T[(size_t)k * N + i];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        J6[r * 2] = J[((size_t)b * 9 + r * 3) * N + i];
        J6[r * 2 + 1] = J[((size_t)b * 9 + r * 3 + 1) * N + i];
    }
    lg_cov2d(T9, view + b * 16, J6, c4);
    size_t o = (size_t)b * 4 * N + i;
#pragma unroll
    for (int k = 0; k < 4; k++) cov[o + (size_t)k * N] = c4[k];
}

LG_API int lg_create_cov2d_forward(const float* J, const float* view, const float* T, const int* valid_length, int V, int N,
                                   float* cov, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(J, view, T, cov);
    hipLaunchKernelGGL(cov2d_forward_kernel, dim3(lg_cdiv(N, TPB), V), dim3(TPB), 0, (hipStream_t)stream, J, view, T, valid_length, N, cov);
    LG_RETURN_LAST();
}

// a16 createCov2dDirectly_backward (GR/transform.cu:824-927): dT = 2.M.dcov.(V33.J)^T summed over views;
// entries >= valid_length are written as 0 like the reference (:884-887).
__global__ void __launch_bounds__(TPB) cov2d_backward_kernel(const float* __restrict__ g_cov, const float* __restrict__ J,
                                                             const float* __restrict__ view, const float* __restrict__ T,
                                                             const int* __restrict__ valid_length, int V, int N, float* __restrict__ gT)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= N) return;
    float sum[9];
#pragma unroll
    for (int k = 0; k < 9; k++) sum[k] = 0.0f;
    if (i < lg_valid_len(valid_length, N)) {
        float T9[9];
#pragma unroll
        for (int k = 0; k < 9; k++) T9[k] = T[(size_t)k * N + i];
        for (int b = 0; b < V; b++) {
            float J6[6], g[4];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                J6[r * 2] = J[((size_t)b * 9 + r * 3) * N + i];
                J6[r * 2 + 1] = J[((size_t)b * 9 + r * 3 + 1) * N + i];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) g[k] = g_cov[((size_t)b * 4 + k) * N + i];
            lg_cov2d_bwd(g, J6, view + b * 16, T9, sum);
        }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) gT[(size_t)k * N + i] = sum[k];
}

LG_API int lg_create_cov2d_backward(const float* g_cov, const float* J, const float* view, const float* T, const int* valid_length,
                                    int V, int N, float* gT, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(g_cov, J, view, T, gT);
    hipLaunchKernelGGL(cov2d_backward_kernel, dim3(lg_cdiv(N, TPB)), dim3(TPB), 0, (hipStream_t)stream, g_cov, J, view, T, valid_length, V, N, gT);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// a7 eigh_and_inv_2x2matrix_forward (GR/transform.cu:1365-1487)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) eigh_inv_forward_kernel(const float* __restrict__ in, const int* __restrict__ valid_length, int N,
                                                               float* __restrict__ val, float* __restrict__ vec, float* __restrict__ inv)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    int b = blockIdx.y;
    if (i >= lg_valid_len(valid_length, N)) return;
    size_t o = (size_t)b * 4 * N + i;
    float m00 = in[o], m01 = in[o + (size_t)N], m10 = in[o + 2 * (size_t)N], m11 = in[o + 3 * (size_t)N];
    float det = m00 * m11 - m01 * m10;
    float det1 = (m00 - m01) * (m11 - m01) + m01 * (m00 + m11 - 2 * m01);
    det = (fabsf(det) < fabsf(1e-5f * m01 * m10)) ? det1 : det;
    float t0 = m00 + m11;
    float t1 = sqrtf((m00 - m11) * (m00 - m11) + 4 * m01 * m01);
    t1 = fmaxf(t1, 1e-9f);
    float e0 = 0.5f * (t0 - t1), e1 = 0.5f * (t0 + t1);
    if (val != nullptr) {
        val[((size_t)b * 2) * N + i] = e0;
        val[((size_t)b * 2 + 1) * N + i] = e1;
    }
    if (vec != nullptr) {
        float v00, v01, v10, v11;
        if (fabsf(e0 - m00) > fabsf(e0 - m11)) { v00 = -m01; v01 = m00 - e0; v10 = e1 - m11; v11 = m01; }
        else { v00 = m11 - e0; v01 = -m01; v10 = m01; v11 = e1 - m00; }
        float l0 = 1.0f / sqrtf(v00 * v00 + v01 * v01);
        float l1 = 1.0f / sqrtf(v10 * v10 + v11 * v11);
        vec[o] = v00 * l0; vec[o + (size_t)N] = v10 * l1; vec[o + 2 * (size_t)N] = v01 * l0; vec[o + 3 * (size_t)N] = v11 * l1;
    }
    float i4[4];
    lg_inv2x2(m00, m01, m10, m11, i4);
#pragma unroll
    for (int k = 0; k < 4; k++) inv[o + (size_t)k * N] = i4[k];
}

LG_API int lg_eigh_inv_2x2_forward(const float* in, const int* valid_length, int V, int N, float* val, float* vec, float* inv, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(in, inv);
    hipLaunchKernelGGL(eigh_inv_forward_kernel, dim3(lg_cdiv(N, TPB), V), dim3(TPB), 0, (hipStream_t)stream, in, valid_length, N, val, vec, inv);
    LG_RETURN_LAST();
}

// a15 inv_2x2matrix_backward (GR/transform.cu:1425-1454, 1489-1518) with the caller's nan_to_num_(0)
// (litegs/utils/wrapper.py:591) optionally folded in (zero_nonfinite).
__global__ void __launch_bounds__(TPB) inv2x2_backward_kernel(const float* __restrict__ inv, const float* __restrict__ g_inv,
                                                              const int* __restrict__ valid_length, int N, int zero_nonfinite,
                                                              float* __restrict__ g_in)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    int b = blockIdx.y;
    if (i >= lg_valid_len(valid_length, N)) return;
    size_t o = (size_t)b * 4 * N + i;
    float a[4], g[4], r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { a[k] = inv[o + (size_t)k * N]; g[k] = g_inv[o + (size_t)k * N]; }
    lg_inv2x2_bwd(a, g, zero_nonfinite != 0, r);
#pragma unroll
    for (int k = 0; k < 4; k++) g_in[o + (size_t)k * N] = r[k];
}

LG_API int lg_inv_2x2_backward(const float* inv, const float* g_inv, const int* valid_length, int V, int N, int zero_nonfinite,
                               float* g_in, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(inv, g_inv, g_in);
    hipLaunchKernelGGL(inv2x2_backward_kernel, dim3(lg_cdiv(N, TPB), V), dim3(TPB), 0, (hipStream_t)stream, inv, g_inv, valid_length, N, zero_nonfinite, g_in);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// SH basis: lg_sh.h (shared with the activation kernels)
// ---------------------------------------------------------------------------------------------

// a22 sh2rgb_forward / sh2rgb_backward (GR/transform.cu:952-1363) -- the cluster_size==0 path
template <int DEG>
__global__ void __launch_bounds__(TPB) sh2rgb_forward_kernel(const float* __restrict__ sh0, const float* __restrict__ shr,
                                                             const float* __restrict__ dirs, int N, float* __restrict__ rgb)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    int v = blockIdx.y;
    if (i >= N) return;
    float b[16];
    lg_sh_basis<DEG>(dirs[((size_t)v * 3) * N + i], dirs[((size_t)v * 3 + 1) * N + i], dirs[((size_t)v * 3 + 2) * N + i], b);
    constexpr int NB = (DEG + 1) * (DEG + 1);
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float r = b[0] * sh0[(size_t)ch * N + i];
#pragma unroll
        for (int k = 1; k < NB; k++) r += b[k] * shr[((size_t)(k - 1) * 3 + ch) * N + i];
        rgb[((size_t)v * 3 + ch) * N + i] = r + 0.5f;
    }
}

LG_API int lg_sh2rgb_forward(int degree, const float* sh0, const float* shr, const float* dirs, int V, int N, float* rgb, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(sh0, dirs, rgb);
    dim3 grid(lg_cdiv(N, TPB), V);
    hipStream_t s = (hipStream_t)stream;
    switch (degree) {
    case 0: hipLaunchKernelGGL(sh2rgb_forward_kernel<0>, grid, dim3(TPB), 0, s, sh0, shr, dirs, N, rgb); break;
    case 1: hipLaunchKernelGGL(sh2rgb_forward_kernel<1>, grid, dim3(TPB), 0, s, sh0, shr, dirs, N, rgb); break;
    case 2: hipLaunchKernelGGL(sh2rgb_forward_kernel<2>, grid, dim3(TPB), 0, s, sh0, shr, dirs, N, rgb); break;
    case 3: hipLaunchKernelGGL(sh2rgb_forward_kernel<3>, grid, dim3(TPB), 0, s, sh0, shr, dirs, N, rgb); break;
    default: return (int)hipErrorInvalidValue;
    }
    LG_RETURN_LAST();
}

// rest_dim rows of d_shr beyond the active degree are written as zero; d_dirs is zero (reference drops it)
template <int DEG>
__global__ void __launch_bounds__(TPB) sh2rgb_backward_kernel(const float* __restrict__ g_rgb, const float* __restrict__ dirs,
                                                              int V, int N, int rest_dim, float* __restrict__ d_sh0,
                                                              float* __restrict__ d_shr, float* __restrict__ d_dirs)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= N) return;
    constexpr int NB = (DEG + 1) * (DEG + 1);
    float acc[NB * 3];
#pragma unroll
    for (int k = 0; k < NB * 3; k++) acc[k] = 0.0f;
    for (int v = 0; v < V; v++) {
        float b[16];
        lg_sh_basis<DEG>(dirs[((size_t)v * 3) * N + i], dirs[((size_t)v * 3 + 1) * N + i], dirs[((size_t)v * 3 + 2) * N + i], b);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float g = g_rgb[((size_t)v * 3 + ch) * N + i];
#pragma unroll
            for (int k = 0; k < NB; k++) acc[k * 3 + ch] += b[k] * g;
            d_dirs[((size_t)v * 3 + ch) * N + i] = 0.0f;
        }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ch++) d_sh0[(size_t)ch * N + i] = acc[ch];
#pragma unroll
    for (int k = 1; k < NB; k++)
#pragma unroll
        for (int ch = 0; ch < 3; ch++) d_shr[((size_t)(k - 1) * 3 + ch) * N + i] = acc[k * 3 + ch];
    for (int k = NB; k <= rest_dim; k++)
        for (int ch = 0; ch < 3; ch++) d_shr[((size_t)(k - 1) * 3 + ch) * N + i] = 0.0f;
}

LG_API int lg_sh2rgb_backward(int degree, const float* g_rgb, const float* dirs, int V, int N, int rest_dim,
                              float* d_sh0, float* d_shr, float* d_dirs, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(g_rgb, dirs, d_sh0);
    dim3 grid(lg_cdiv(N, TPB));
    hipStream_t s = (hipStream_t)stream;
    switch (degree) {
    case 0: hipLaunchKernelGGL(sh2rgb_backward_kernel<0>, grid, dim3(TPB), 0, s, g_rgb, dirs, V, N, rest_dim, d_sh0, d_shr, d_dirs); break;
    case 1: hipLaunchKernelGGL(sh2rgb_backward_kernel<1>, grid, dim3(TPB), 0, s, g_rgb, dirs, V, N, rest_dim, d_sh0, d_shr, d_dirs); break;
    case 2: hipLaunchKernelGGL(sh2rgb_backward_kernel<2>, grid, dim3(TPB), 0, s, g_rgb, dirs, V, N, rest_dim, d_sh0, d_shr, d_dirs); break;
    case 3: hipLaunchKernelGGL(sh2rgb_backward_kernel<3>, grid, dim3(TPB), 0, s, g_rgb, dirs, V, N, rest_dim, d_sh0, d_shr, d_dirs); break;
    default: return (int)hipErrorInvalidValue;
    }
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// world2ndc forward/backward (GR/transform.cu:602-731) -- legacy projection, no caller on the render
// path; provided so the 26-symbol surface is complete.  Backward sums over views (the reference
// overwrites per view, i.e. keeps the last one: a bug for V>1, identical for V==1).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) world2ndc_forward_kernel(const float* __restrict__ world, const float* __restrict__ vp, int N,
                                                                float* __restrict__ ndc, float* __restrict__ rw)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    int b = blockIdx.y;
    if (i >= N) return;
    const float* M = vp + b * 16;
    float w0 = world[i], w1 = world[(size_t)N + i], w2 = world[2 * (size_t)N + i], w3 = world[3 * (size_t)N + i];
    float h[4];
#pragma unroll
    for (int k = 0; k < 4; k++) h[k] = w0 * M[k] + w1 * M[4 + k] + w2 * M[8 + k] + w3 * M[12 + k];
    float r = 1.0f / (h[3] + 1e-7f);
    rw[(size_t)b * N + i] = r;
    size_t o = (size_t)b * 4 * N + i;
    ndc[o] = h[0] * r; ndc[o + (size_t)N] = h[1] * r; ndc[o + 2 * (size_t)N] = h[2] * r; ndc[o + 3 * (size_t)N] = 1.0f;
}

LG_API int lg_world2ndc_forward(const float* world, const float* viewproj, int V, int N, float* ndc, float* recp_w, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(world, viewproj, ndc, recp_w);
    hipLaunchKernelGGL(world2ndc_forward_kernel, dim3(lg_cdiv(N, TPB), V), dim3(TPB), 0, (hipStream_t)stream, world, viewproj, N, ndc, recp_w);
    LG_RETURN_LAST();
}

__global__ void __launch_bounds__(TPB) world2ndc_backward_kernel(const float* __restrict__ vp, const float* __restrict__ ndc,
                                                                 const float* __restrict__ rw, const float* __restrict__ g_ndc,
                                                                 int V, int N, float* __restrict__ g_pos)
{
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= N) return;
    float gx = 0, gy = 0, gz = 0;
    for (int b = 0; b < V; b++) {
        const float* M = vp + b * 16;
        size_t o = (size_t)b * 4 * N + i;
        float r = rw[(size_t)b * N + i];
        float m1 = ndc[o] * r, m2 = ndc[o + (size_t)N] * r, m3 = ndc[o + 2 * (size_t)N] * r;
        float g0 = g_ndc[o], g1 = g_ndc[o + (size_t)N], g2 = g_ndc[o + 2 * (size_t)N];
        gx += (M[0] * r - M[3] * m1) * g0 + (M[1] * r - M[3] * m2) * g1 + (M[2] * r - M[3] * m3) * g2;
        gy += (M[4] * r - M[7] * m1) * g0 + (M[5] * r - M[7] * m2) * g1 + (M[6] * r - M[7] * m3) * g2;
        gz += (M[8] * r - M[11] * m1) * g0 + (M[9] * r - M[11] * m2) * g1 + (M[10] * r - M[11] * m3) * g2;
    }
    g_pos[i] = gx; g_pos[(size_t)N + i] = gy; g_pos[2 * (size_t)N + i] = gz; g_pos[3 * (size_t)N + i] = 0.0f;
}

LG_API int lg_world2ndc_backward(const float* viewproj, const float* ndc, const float* recp_w, const float* g_ndc, int V, int N,
                                 float* g_pos, void* stream)
{
    if (N <= 0) return 0;
    LG_REQUIRE(viewproj, ndc, recp_w, g_ndc, g_pos);
    hipLaunchKernelGGL(world2ndc_backward_kernel, dim3(lg_cdiv(N, TPB)), dim3(TPB), 0, (hipStream_t)stream, viewproj, ndc, recp_w, g_ndc, V, N, g_pos);
    LG_RETURN_LAST();
}

// ---------------------------------------------------------------------------------------------
// Learnable cameras: create_viewproj forward/backward (GR/compact.cu:17-316).
// view_params[v] = (qr,qx,qy,qz, tx,ty,tz); row-vector convention: view = [R^T rows | t], p_clip = p.view.proj.
// One lane per view (a few hundred views at most: launch-latency bound, nothing to tile).
// Reference behaviour kept:  d(proj_11)/d(fov) uses the INTEGER ratio img_w/img_h (compact.cu:268), and the
// quaternion gradient is projected on the unit sphere without the 1/|q| factor (compact.cu:271-277).
// Difference: the reference accumulates grad_fov[0] with a plain "+=" from every thread (a race for V>1);
// here the per-view contributions are reduced in a fixed order.
// ---------------------------------------------------------------------------------------------
struct CamMats { float view[16]; float p00, p11, p22, p32; };

__device__ __forceinline__ void lg_camera_matrices(const float* __restrict__ vp7, float fov_recp, int H, int W, float zn, float zf, CamMats& m,
                                                   float q[4])
{
    float r = vp7[0], x = vp7[1], y = vp7[2], z = vp7[3];
    float inv = 1.0f / sqrtf(r * r + x * x + y * y + z * z + 1e-12f);
    r *= inv; x *= inv; y *= inv; z *= inv;
    q[0] = r; q[1] = x; q[2] = y; q[3] = z;
    float* V = m.view;
    V[0] = 1 - 2 * (y * y + z * z); V[1] = 2 * (x * y + r * z);     V[2] = 2 * (x * z - r * y);      V[3] = 0;
    V[4] = 2 * (x * y - r * z);     V[5] = 1 - 2 * (x * x + z * z); V[6] = 2 * (y * z + r * x);      V[7] = 0;
    V[8] = 2 * (x * z + r * y);     V[9] = 2 * (y * z - r * x);     V[10] = 1 - 2 * (x * x + y * y); V[11] = 0;
    V[12] = vp7[4]; V[13] = vp7[5]; V[14] = vp7[6]; V[15] = 1.0f;
    m.p00 = fov_recp;
    m.p11 = fov_recp * W / H;
    m.p22 = zf / (zf - zn);
    m.p32 = -zf * zn / (zf - zn);
}

__global__ void __launch_bounds__(64) create_viewproj_forward_kernel(const float* __restrict__ view_params, const float* __restrict__ fov_recp,
                                                                     int V, int H, int W, float zn, float zf, float* __restrict__ view_m,
                                                                     float* __restrict__ proj_m, float* __restrict__ vp_m,
                                                                     float* __restrict__ planes)
{
    int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= V) return;
    CamMats m; float q[4];
    lg_camera_matrices(view_params + 7 * v, fov_recp[0], H, W, zn, zf, m, q);
    float P[16] = { m.p00, 0, 0, 0,  0, m.p11, 0, 0,  0, 0, m.p22, 1,  0, 0, m.p32, 0 };
    float M[16];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float t = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++) t += m.view[i * 4 + k] * P[k * 4 + j];
            M[i * 4 + j] = t;
        }
#pragma unroll
    for (int e = 0; e < 16; e++) { view_m[16 * v + e] = m.view[e]; proj_m[16 * v + e] = P[e]; vp_m[16 * v + e] = M[e]; }
    // clip-space half spaces: w+x, w-x, w+y, w-y, z, w-z  (column c of viewproj = M[.][c])
    float* pl = planes + 24 * v;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float cx = M[r * 4 + 0], cy = M[r * 4 + 1], cz = M[r * 4 + 2], cw = M[r * 4 + 3];
        pl[0 * 4 + r] = cw + cx; pl[1 * 4 + r] = cw - cx;
        pl[2 * 4 + r] = cw + cy; pl[3 * 4 + r] = cw - cy;
        pl[4 * 4 + r] = cz;      pl[5 * 4 + r] = cw - cz;
    }
}

LG_API int lg_create_viewproj_forward(const float* view_params, const float* recp_tan_half_fov_x, int V, int H, int W, float z_near, float z_far,
                                      float* view_matrix, float* proj_matrix, float* viewproj_matrix, float* frustumplane, void* stream)
{
    if (V <= 0) return 0;
    LG_REQUIRE(view_params, recp_tan_half_fov_x, view_matrix, proj_matrix);
    hipLaunchKernelGGL(create_viewproj_forward_kernel, dim3(lg_cdiv(V, 64)), dim3(64), 0, (hipStream_t)stream, view_params, recp_tan_half_fov_x,
                       V, H, W, z_near, z_far, view_matrix, proj_matrix, viewproj_matrix, frustumplane);
    LG_RETURN_LAST();
}

// single workgroup: lanes stride over views, the fov gradient is reduced in lane order through LDS
__global__ void __launch_bounds__(256) create_viewproj_backward_kernel(const float* __restrict__ g_view, const float* __restrict__ g_proj,
                                                                       const float* __restrict__ g_vp, const float* __restrict__ view_params,
                                                                       const float* __restrict__ fov_recp, int V, int H, int W, float zn,
                                                                       float zf, float* __restrict__ g_params, float* __restrict__ g_fov)
{
    __shared__ float part[256];
    float fov_acc = 0.0f;
    const float int_aspect = (float)(W / H);                    // integer division, as compact.cu:268
    for (int v = threadIdx.x; v < V; v += 256) {
        CamMats m; float q[4];
        lg_camera_matrices(view_params + 7 * v, fov_recp[0], H, W, zn, zf, m, q);
        const float* Gm = g_vp + 16 * v;
        float A[12];                                            // d/d(view[i][0..2]) for i = 0..3
        float gp00 = 0.0f, gp11 = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float g0 = Gm[i * 4 + 0], g1 = Gm[i * 4 + 1], g2 = Gm[i * 4 + 2], g3 = Gm[i * 4 + 3];
            A[i * 3 + 0] = g_view[16 * v + i * 4 + 0] + g0 * m.p00;
            A[i * 3 + 1] = g_view[16 * v + i * 4 + 1] + g1 * m.p11;
            A[i * 3 + 2] = g_view[16 * v + i * 4 + 2] + (g2 * m.p22 + g3);
            gp00 += g0 * m.view[i * 4 + 0];
            gp11 += g1 * m.view[i * 4 + 1];
        }
        float r = q[0], x = q[1], y = q[2], z = q[3];
        float gr = 0, gx = 0, gy = 0, gz = 0, g;
        g = A[0]; gy += g * (-4 * y); gz += g * (-4 * z);
        g = A[1]; gx += g * (2 * y); gy += g * (2 * x); gr += g * (2 * z); gz += g * (2 * r);
        g = A[2]; gx += g * (2 * z); gz += g * (2 * x); gr += g * (-2 * y); gy += g * (-2 * r);
        g = A[3]; gx += g * (2 * y); gy += g * (2 * x); gr += g * (-2 * z); gz += g * (-2 * r);
        g = A[4]; gx += g * (-4 * x); gz += g * (-4 * z);
        g = A[5]; gy += g * (2 * z); gz += g * (2 * y); gr += g * (2 * x); gx += g * (2 * r);
        g = A[6]; gx += g * (2 * z); gz += g * (2 * x); gr += g * (2 * y); gy += g * (2 * r);
        g = A[7]; gy += g * (2 * z); gz += g * (2 * y); gr += g * (-2 * x); gx += g * (-2 * r);
        g = A[8]; gx += g * (-4 * x); gy += g * (-4 * y);
        float norm = sqrtf(r * r + x * x + y * y + z * z);
        float dot = (r * gr + x * gx + y * gy + z * gz) / (norm * norm);
        float* o = g_params + 7 * v;
        o[0] = gr / norm - r * dot; o[1] = gx / norm - x * dot; o[2] = gy / norm - y * dot; o[3] = gz / norm - z * dot;
        o[4] = A[9]; o[5] = A[10]; o[6] = A[11];
        fov_acc += (g_proj[16 * v + 0] + gp00);
        fov_acc += (g_proj[16 * v + 5] + gp11) * int_aspect;
    }
    part[threadIdx.x] = fov_acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        int n = V < 256 ? V : 256;
        for (int i = 0; i < n; i++) s += part[i];
        g_fov[0] = s;
    }
}

LG_API int lg_create_viewproj_backward(const float* view_matrix_grad, const float* proj_matrix_grad, const float* viewproj_matrix_grad,
                                       const float* view_params, const float* recp_tan_half_fov_x, int V, int H, int W, float z_near,
                                       float z_far, float* grad_view_params, float* grad_recp_tan_half_fov_x, void* stream)
{
    if (V <= 0) return 0;
    hipLaunchKernelGGL(create_viewproj_backward_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, view_matrix_grad, proj_matrix_grad,
                       viewproj_matrix_grad, view_params, recp_tan_half_fov_x, V, H, W, z_near, z_far, grad_view_params,
                       grad_recp_tan_half_fov_x);
    LG_RETURN_LAST();
}
