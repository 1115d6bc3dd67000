// Per-Gaussian projection chain as inline device functions, shared by the op-by-op kernels (transform.hip,
// compact.hip: the drop-in `litegs_fused` surface) and by the fused kernels of fused.hip, so both paths execute
// the SAME arithmetic in the same order.  Each function cites the reference kernel it restates.
#pragma once
#include "lg_common.h"
#include "lg_sh.h"

// ---- activation (GR/compact.cu:848-886) -------------------------------------------------------
__device__ __forceinline__ float lg_act_scale(float s) { return __expf(s); }
__device__ __forceinline__ float lg_act_opacity(float o) { return 1.0f / (1.0f + __expf(-o)); }
__device__ __forceinline__ float lg_act_quat(float w, float x, float y, float z, float* q)
{
    float rn = rsqrtf(w * w + x * x + y * y + z * z + 1e-12f);
    q[0] = w * rn; q[1] = x * rn; q[2] = y * rn; q[3] = z * rn;
    return rn;
}
// camera centre = -t . R^T for the row-vector view matrix (compact.cu:875-879)
__device__ __forceinline__ void lg_camera_center(const float* __restrict__ V, float& cx, float& cy, float& cz)
{
    float ix = -V[12], iy = -V[13], iz = -V[14];
    cx = ix * V[0] + iy * V[1] + iz * V[2];
    cy = ix * V[4] + iy * V[5] + iz * V[6];
    cz = ix * V[8] + iy * V[9] + iz * V[10];
}
__device__ __forceinline__ void lg_view_dir(float px, float py, float pz, float cx, float cy, float cz, float& dx, float& dy, float& dz)
{
    dx = px - cx; dy = py - cy; dz = pz - cz;
    float nr = rsqrtf(dx * dx + dy * dy + dz * dz + 1e-12f);
    dx *= nr; dy *= nr; dz *= nr;
}

// ---- a3 mvp (GR/transform.cu:398-436) -------------------------------------------------------------
__device__ __forceinline__ void lg_mvp(const float* __restrict__ V, const float* __restrict__ P, float w0, float w1, float w2, float w3,
                                       float* v, float* ndc)
{
    float h[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = w0 * V[k] + w1 * V[4 + k] + w2 * V[8 + k] + w3 * V[12 + k];
#pragma unroll
    for (int k = 0; k < 4; k++) h[k] = v[0] * P[k] + v[1] * P[4 + k] + v[2] * P[8 + k] + v[3] * P[12 + k];
    float iw = (fabsf(h[3]) > 1e-12f) ? (1.0f / h[3]) : 0.0f;
    ndc[0] = h[0] * iw; ndc[1] = h[1] * iw; ndc[2] = h[2] * iw; ndc[3] = 1.0f;
}

// a18 mvp backward for one view (GR/transform.cu:496-558): returns the contribution to d_world
__device__ __forceinline__ void lg_mvp_bwd(const float* __restrict__ Vm, const float* __restrict__ P, const float* v, const float* gn,
                                           const float* gview, float* acc)
{
    float h[4], dh[4], dv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) h[k] = v[0] * P[k] + v[1] * P[4 + k] + v[2] * P[8 + k] + v[3] * P[12 + k];
    float iw = (fabsf(h[3]) > 1e-12f) ? (1.0f / h[3]) : 0.0f;
    float n0 = h[0] * iw, n1 = h[1] * iw, n2 = h[2] * iw;
    dh[0] = gn[0] * iw; dh[1] = gn[1] * iw; dh[2] = gn[2] * iw;
    dh[3] = -(gn[0] * n0 + gn[1] * n1 + gn[2] * n2) * iw;
#pragma unroll
    for (int k = 0; k < 4; k++) dv[k] = gview[k] + (dh[0] * P[k * 4] + dh[1] * P[k * 4 + 1] + dh[2] * P[k * 4 + 2] + dh[3] * P[k * 4 + 3]);
#pragma unroll
    for (int k = 0; k < 4; k++) acc[k] += dv[0] * Vm[k * 4] + dv[1] * Vm[k * 4 + 1] + dv[2] * Vm[k * 4 + 2] + dv[3] * Vm[k * 4 + 3];
}

// ---- a4 transform matrix (GR/transform.cu:106-125) ---------------------------------------------------
__device__ __forceinline__ void lg_quat_rows(float r, float x, float y, float z, float* R)
{
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y + r * z);     R[2] = 2 * (x * z - r * y);
    R[3] = 2 * (x * y - r * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z + r * x);
    R[6] = 2 * (x * z + r * y);     R[7] = 2 * (y * z - r * x);     R[8] = 1 - 2 * (x * x + y * y);
}
__device__ __forceinline__ void lg_transform_matrix(const float* q, const float* s, float* T)
{
    float R[9];
    lg_quat_rows(q[0], q[1], q[2], q[3], R);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) T[r * 3 + c] = R[r * 3 + c] * s[r];
}
// a17 (GR/transform.cu:185-225); dt is consumed (scaled in place)
__device__ __forceinline__ void lg_transform_matrix_bwd(float* dt, const float* q, const float* s, float* gq, float* gs)
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R[9];
    lg_quat_rows(r, x, y, z, R);
#pragma unroll
    for (int rr = 0; rr < 3; rr++) gs[rr] = R[rr * 3] * dt[rr * 3] + R[rr * 3 + 1] * dt[rr * 3 + 1] + R[rr * 3 + 2] * dt[rr * 3 + 2];
#pragma unroll
    for (int rr = 0; rr < 3; rr++) { dt[rr * 3] *= s[rr]; dt[rr * 3 + 1] *= s[rr]; dt[rr * 3 + 2] *= s[rr]; }
    gq[0] = 2 * z * (dt[1] - dt[3]) + 2 * y * (dt[6] - dt[2]) + 2 * x * (dt[5] - dt[7]);
    gq[1] = 2 * y * (dt[3] + dt[1]) + 2 * z * (dt[6] + dt[2]) + 2 * r * (dt[5] - dt[7]) - 4 * x * (dt[8] + dt[4]);
    gq[2] = 2 * x * (dt[3] + dt[1]) + 2 * r * (dt[6] - dt[2]) + 2 * z * (dt[5] + dt[7]) - 4 * y * (dt[8] + dt[0]);
    gq[3] = 2 * r * (dt[1] - dt[3]) + 2 * x * (dt[6] + dt[2]) + 2 * y * (dt[5] + dt[7]) - 4 * z * (dt[4] + dt[0]);
}

// ---- a5 ray-space Jacobian (GR/transform.cu:36-50): the four non-zero entries J00 J11 J20 J21 ---------
__device__ __forceinline__ void lg_jacobian(const float* __restrict__ P, int H, int W, float tx, float ty, float tz, float* j4)
{
    float fx = P[0] * W * 0.5f, fy = P[5] * H * 0.5f;
    float lx = tz / P[0] * 1.3f, ly = tz / P[5] * 1.3f;
    tx = fmaxf(fminf(tx, lx), -lx);
    ty = fmaxf(fminf(ty, ly), -ly);
    float rz = 1.0f / fmaxf(tz, 1e-2f);
    float rz2 = rz * rz;
    j4[0] = fx * rz; j4[1] = fy * rz; j4[2] = -fx * tx * rz2; j4[3] = -fy * ty * rz2;
}

// ---- a6 cov2d (GR/transform.cu:760-778): J6 = J[:, :2] row-major 3x2 ------------------------------------
__device__ __forceinline__ void lg_cov2d(const float* T9, const float* __restrict__ Vm, const float* J6, float* cov4)
{
    float tv[9], M[6];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += T9[r * 3 + k] * Vm[k * 4 + c];
            tv[r * 3 + c] = s;
        }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            float s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += tv[r * 3 + k] * J6[k * 2 + c];
            M[r * 2 + c] = s;
        }
    float c00 = 0, c01 = 0, c11 = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) { c00 += M[k * 2] * M[k * 2]; c01 += M[k * 2] * M[k * 2 + 1]; c11 += M[k * 2 + 1] * M[k * 2 + 1]; }
    cov4[0] = c00 + 0.3f; cov4[1] = c01; cov4[2] = c01; cov4[3] = c11 + 0.3f;
}
// a16 (GR/transform.cu:849-881): adds this view's 2.M.dcov.(V33.J)^T into sum9
__device__ __forceinline__ void lg_cov2d_bwd(const float* g4, const float* J6, const float* __restrict__ Vm, const float* T9, float* sum9)
{
    float vj[6], M[6], dM[6];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            float s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += Vm[r * 4 + k] * J6[k * 2 + c];
            vj[r * 2 + c] = s;
        }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            float s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += T9[r * 3 + k] * vj[k * 2 + c];
            M[r * 2 + c] = s;
        }
#pragma unroll
    for (int r = 0; r < 3; r++) {
        dM[r * 2] = 2 * (M[r * 2] * g4[0] + M[r * 2 + 1] * g4[2]);
        dM[r * 2 + 1] = 2 * (M[r * 2] * g4[1] + M[r * 2 + 1] * g4[3]);
    }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) sum9[r * 3 + c] += dM[r * 2] * vj[c * 2] + dM[r * 2 + 1] * vj[c * 2 + 1];
}

// ---- a7 inverse of the 2x2 covariance (GR/transform.cu:1379-1383, 1414-1420) -----------------------------
__device__ __forceinline__ void lg_inv2x2(float m00, float m01, float m10, float m11, float* inv4)
{
    float det = m00 * m11 - m01 * m10;
    float det1 = (m00 - m01) * (m11 - m01) + m01 * (m00 + m11 - 2 * m01);
    det = (fabsf(det) < fabsf(1e-5f * m01 * m10)) ? det1 : det;
    det = (fabsf(det) < 1e-9f) ? 1e-9f : det;
    float dr = 1.0f / det;
    inv4[0] = m11 * dr; inv4[1] = -m01 * dr; inv4[2] = -m10 * dr; inv4[3] = m00 * dr;
}
// a15 (GR/transform.cu:1440-1451) with wrapper.py:591's nan_to_num_(0) optionally folded in
__device__ __forceinline__ void lg_inv2x2_bwd(const float* a, const float* g, bool zero_nonfinite, float* out4)
{
    float t[4], r[4];
    t[0] = a[0] * g[0] + a[1] * g[2]; t[1] = a[0] * g[1] + a[1] * g[3];
    t[2] = a[2] * g[0] + a[3] * g[2]; t[3] = a[2] * g[1] + a[3] * g[3];
    r[0] = t[0] * a[0] + t[1] * a[2]; r[1] = t[0] * a[1] + t[1] * a[3];
    r[2] = t[2] * a[0] + t[3] * a[2]; r[3] = t[2] * a[1] + t[3] * a[3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float v = -r[k];
        if (zero_nonfinite && !(fabsf(v) <= 3.402823466e+38f)) v = 0.0f;
        out4[k] = v;
    }
}

// log2 of the activated opacity, folded into the blend exponent by the record packers (raster.hip): one hardware instruction, so the
// operator path's pack kernel and the fused projection produce the same bits
__device__ __forceinline__ float lg_log2_opacity(float o) { return __builtin_amdgcn_logf(o); }

// Blend-backward moments -> gradients of the splat's screen position, conic and opacity (reference: GR/raster.cu:826-841, where the
// same products are formed per (tile, splat) before the atomics; they are linear in the moments, so they are formed once per splat
// here).  Moments are sums over the splat's pixels of m = dL/dalpha * (opacity * G): Mx = sum m dx, ... (layout: raster.hip).
// conic = (a, b, c) of power = -0.5 a dx^2 - b dx dy - 0.5 c dy^2, o = activated opacity (alpha = o G, so dL/dpower = m and
// dL/do = sum m / o).  g[0..4] = d_px, d_py, d_a, d_b (each off-diagonal), d_c; g[8] = d_opacity.
__device__ __forceinline__ void lg_moments_to_grads(float Mx, float My, float Mxx, float Mxy, float Myy, float M0,
                                                    float a, float b, float c, float o, float* g)
{
    g[0] = -(a * Mx + b * My);
    g[1] = -(c * My + b * Mx);
    g[2] = -0.5f * Mxx;
    g[3] = -0.5f * Mxy;
    g[4] = -0.5f * Myy;
    g[8] = o > 0.0f ? M0 / o : 0.0f;
}
