// Per-Gaussian backward of the whole projection chain in registers and the row-batched Adam update, shared by the single-GPU
// executor (fused.hip: project_backward_adam_kernel) and the data-parallel one (dp.hip).  Include only from translation units built
// with -ffp-contract=off (same arithmetic as the forward chain of lg_chain.h).
#pragma once
#include "lg_common.h"
#include "lg_chain.h"

#define LG_GREC 16      // floats per packed gradient record (raster.hip)

struct Camera {
    float V[16];
    float P[16];
    int H, W;
};

// per-Gaussian backward in registers: unpack (GR/raster.cu:866-884) + chain backward + activation backward
struct GaussGrads {
    float pos[3], scale[3], rot[4], opa;
    float gc[3];          // dL/d(colour); SH coefficient k of channel ch gets basis[k] * gc[ch]
    float basis[16];
};

template <int DEG>
// mom: the nine blend-backward moments of the Gaussian (raster.hip: Mx My Mxx Mxy Myy dr dg db M0)
__device__ __forceinline__ void gaussian_backward(const Camera& cam, const float (&mom)[9], float sc,
                                                  float px, float py, float pz, float sr0, float sr1, float sr2,
                                                  float rw, float rx, float ry, float rz, float opa_raw, GaussGrads& G)
{
    // ---- recompute the forward chain from the raw parameters
    float s3[3] = { lg_act_scale(sr0), lg_act_scale(sr1), lg_act_scale(sr2) }, q[4];
    const float rn = lg_act_quat(rw, rx, ry, rz, q);
    float v[4], n[4], T9[9], j4[4], J6[6], c4[4], i4[4];
    lg_mvp(cam.V, cam.P, px, py, pz, 1.0f, v, n);
    lg_transform_matrix(q, s3, T9);
    lg_jacobian(cam.P, cam.H, cam.W, v[0], v[1], v[2], j4);
    J6[0] = j4[0]; J6[1] = 0.0f; J6[2] = 0.0f; J6[3] = j4[1]; J6[4] = j4[2]; J6[5] = j4[3];
    lg_cov2d(T9, cam.V, J6, c4);
    lg_inv2x2(c4[0], c4[1], c4[2], c4[3], i4);
    // ---- unpack: blend-backward moments -> d_pixel, d_conic, d_opacity (raster.hip), then GR/raster.cu:866-884
    float gm[9];
    lg_moments_to_grads(mom[0], mom[1], mom[2], mom[3], mom[4], mom[8], i4[0], i4[1], i4[3], lg_act_opacity(opa_raw), gm);
    float gn[4] = { gm[0] * 0.5f * cam.W * sc, gm[1] * 0.5f * cam.H * sc, 0.0f, 0.0f };
    float ginv[4] = { gm[2] * sc, gm[3] * sc, gm[3] * sc, gm[4] * sc };
    G.gc[0] = mom[5] * sc; G.gc[1] = mom[6] * sc; G.gc[2] = mom[7] * sc;
    const float gop = gm[8] * sc;
    // ---- chain backward
    float gcov[4], gT[9], gq[4], gs[3];
    lg_inv2x2_bwd(i4, ginv, true, gcov);
#pragma unroll
    for (int k = 0; k < 9; k++) gT[k] = 0.0f;
    lg_cov2d_bwd(gcov, J6, cam.V, T9, gT);
    lg_transform_matrix_bwd(gT, q, s3, gq, gs);
    float gw[4] = { 0.f, 0.f, 0.f, 0.f }, gview[4] = { 0.f, 0.f, 0.f, 0.f };
    lg_mvp_bwd(cam.V, cam.P, v, gn, gview, gw);
    // ---- activation backward (GR/compact.cu:925-977)
    G.pos[0] = gw[0]; G.pos[1] = gw[1]; G.pos[2] = gw[2];
#pragma unroll
    for (int k = 0; k < 3; k++) G.scale[k] = s3[k] * gs[k];
    const float dot = gq[0] * q[0] + gq[1] * q[1] + gq[2] * q[2] + gq[3] * q[3];
#pragma unroll
    for (int k = 0; k < 4; k++) G.rot[k] = rn * (gq[k] - dot * q[k]);
    G.opa = gop * (1.0f - 1.0f / (1.0f + __expf(opa_raw)));        // sic: g * sigmoid(x), compact.cu:952
    float cx, cy, cz, dx, dy, dz;
    lg_camera_center(cam.V, cx, cy, cz);
    lg_view_dir(px, py, pz, cx, cy, cz, dx, dy, dz);
    lg_sh_basis<DEG>(dx, dy, dz, G.basis);
}

// the nine moments of compacted Gaussian `od` from the blend backward's 64-byte gradient records
__device__ __forceinline__ void load_moments(const float4* __restrict__ packed_grad, size_t od, float (&mom)[9])
{
    const float4* __restrict__ rec = packed_grad + od * (LG_GREC / 4);
    const float4 a = rec[0], b = rec[1];
    mom[0] = a.x; mom[1] = a.y; mom[2] = a.z; mom[3] = a.w; mom[4] = b.x; mom[5] = b.y; mom[6] = b.z; mom[7] = b.w; mom[8] = rec[2].x;
}

// Gradient replicas (raster.hip "replicas"): the blend backward adds the moments of a splat that covers many tiles into one of R lines
// behind the N regular records instead of hammering a single line; hot = (first replica line << 6) | log2 R, or -1.  The consumer sums
// the regular record and the R replicas.
__device__ __forceinline__ void load_moments_folded(const float4* __restrict__ packed_grad, size_t od, long long N, const int* __restrict__ hot_of,
                                                    float (&mom)[9])
{
    load_moments(packed_grad, od, mom);
    if (hot_of == nullptr) return;
    const int hot = hot_of[od];
    if (hot < 0) return;
    const int R = 1 << (hot & 63);
    const float4* __restrict__ rec = packed_grad + ((size_t)N + (size_t)(hot >> 6)) * (LG_GREC / 4);
    // eight lines (24 loads) in flight at a time: one line after the other, a splat with 64 replicas would stall its workgroup for 64
    // dependent round trips
    for (int r0 = 0; r0 < R; r0 += 8) {
        float4 a[8], b[8];
        float c[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = (r0 + j < R) ? r0 + j : r0;            // R is a power of two: r0 itself is always a valid line (counted once below)
            a[j] = rec[r * (LG_GREC / 4)]; b[j] = rec[r * (LG_GREC / 4) + 1]; c[j] = rec[r * (LG_GREC / 4) + 2].x;
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (r0 + j < R) {
                mom[0] += a[j].x; mom[1] += a[j].y; mom[2] += a[j].z; mom[3] += a[j].w;
                mom[4] += b[j].x; mom[5] += b[j].y; mom[6] += b[j].z; mom[7] += b[j].w; mom[8] += c[j];
            }
    }
}

struct AdamRates { float lr_pos, lr_sh0, lr_shr, lr_opa, lr_scale, lr_rot, b1, b2, eps; };

__device__ __forceinline__ void adam_row(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, size_t o, float g,
                                         float lr, float b1, float b2, float eps)
{
    float mm = b1 * m[o] + (1.0f - b1) * g;
    float vv = b2 * v[o] + (1.0f - b2) * g * g;
    p[o] += -lr * mm / (sqrtf(vv) + eps);
    m[o] = mm;
    v[o] = vv;
}

// NR rows of one parameter tensor at a time: all 3*NR loads are issued before the first store, so a wave keeps 3*NR cache lines in
// flight (one row at a time leaves 3 -- stores to the same tensor cannot be proven disjoint from the next row's loads).
template <int NR>
__device__ __forceinline__ void adam_rows(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const size_t (&o)[NR],
                                          const float (&g)[NR], float lr, float b1, float b2, float eps)
{
    float pp[NR], mm[NR], vv[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) { pp[i] = p[o[i]]; mm[i] = m[o[i]]; vv[i] = v[o[i]]; }
#pragma unroll
    for (int i = 0; i < NR; i++) {
        const float m1 = b1 * mm[i] + (1.0f - b1) * g[i];
        const float v1 = b2 * vv[i] + (1.0f - b2) * g[i] * g[i];
        p[o[i]] = pp[i] + -lr * m1 / (sqrtf(v1) + eps);
        m[o[i]] = m1;
        v[o[i]] = v1;
    }
}

