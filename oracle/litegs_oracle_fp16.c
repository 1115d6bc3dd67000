/*
 * litegs_oracle_fp16.c -- the blend forward / backward of the reference BINARY, emulated on the CPU.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as litegs_oracle.c).
 *
 * The reference's blend kernels do not compute in fp32: GR/raster.cu blends in half2 with range hacks.  This file restates
 * that arithmetic operation by operation so that the distance between "what the reference binary would output" and the fp32
 * oracle / the HIP path can be REPORTED (tests/test_oracle_fp16.py, tools/fp16_distance.py).  It is not a parity
 * target: the HIP path is held to the fp32 oracle (1e-4), which this variant cannot meet by construction.
 *
 * What is emulated (citations: /root/reference/litegs/submodules/gaussian_raster/raster.cu):
 *   - colour and opacity rounded to binary16 when packed (:353-354) -- done by the caller (oracle.py: pack_params_fp16);
 *   - the exponent by forward differences in fp32 along a thread's pixel rows (:237-243, 256-262), rounded to half per pixel;
 *   - G = ex2.approx.f16x2(half(power * log2e)) (:72-78): evaluated as the correctly rounded half of 2^x -- the PTX
 *     approximation's own error (< 1 half ulp by its specification) is NOT modelled;
 *   - alpha = a * G, the 1/256 and 255/256 thresholds, the activity test T > 128/8192, all in half (:264-275);
 *   - transmittance carried as half(128 * T) (:179-180, 213), weight = t * alpha, colour accumulation and t *= (1 - alpha) as half
 *     operations; `x += a * b` is taken as one fused half FMA (nvcc's default contraction);
 *   - outputs = float(half) / 128, colour clamped at 1 (:306-322);
 *   - backward (:651-849): T and the pixel gradients rounded to half (:667-690), t = min(128, t * rcp(1 - alpha)), the colour and
 *     alpha gradients accumulated in half, the three geometric partial sums accumulated in fp32 from half products with the
 *     row index (:785-790), warp reductions of rg / ba in HALF by the shfl_down tree (:81-91, 800-801), the rest in fp32;
 *     per-Gaussian accumulation (atomicAdd, fp32) in list order of the tiles here; unpack multiplies by 1/128 (:866-884).
 * Thread -> pixel map of the reference (32 lanes per tile): x = lane % TW, first row = (lane / TW) * 2 * PPT, rows 2i (".x") and
 * 2i+1 (".y") of the half2 for i < PPT = TH*TW/64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))
#define ORC_REC 16

/* round to the nearest binary16 value (ties to even, gradual underflow, overflow to infinity); returned as double */
static inline double rh(double x)
{
    if (x == 0.0 || x != x || isinf(x)) return x;
    double a = fabs(x);
    if (a >= 65520.0) return copysign(INFINITY, x);
    int e;
    frexp(a, &e);                         /* a = m * 2^e, m in [0.5, 1) */
    int ue = e - 11;                      /* ulp exponent of a normal half with this exponent */
    if (ue < -24) ue = -24;               /* subnormal spacing */
    double ulp = ldexp(1.0, ue);
    double r = nearbyint(a / ulp) * ulp;  /* default rounding mode: ties to even */
    if (r >= 65520.0) return copysign(INFINITY, x);
    return copysign(r, x);
}
static inline double hmul(double a, double b) { return rh(a * b); }
static inline double hadd(double a, double b) { return rh(a + b); }
static inline double hsub(double a, double b) { return rh(a - b); }
static inline double hfma(double a, double b, double c) { return rh(a * b + c); }
static inline double hmin(double a, double b) { return a < b ? a : b; }
/* fast_exp_approx, raster.cu:72-78 */
static inline double hexp(double power_half)
{
    double scaled = hmul(power_half, rh(1.4426950409));
    return rh(exp2(scaled));
}

#define SCALER 128.0

ORC_API void orc_raster_forward_fp16(const int32_t* sorted_points, const int32_t* start_index, const float* packed /*colour, opacity already half-rounded*/,
                                     int V, int64_t L, int N, int H, int W, int TH, int TW,
                                     float* img, float* trans, int16_t* last)
{
    int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    int Wp = gx * TW, Hp = gy * TH, ntiles = gx * gy;
    int PPT = TH * TW / 64;
    size_t plane = (size_t)Hp * Wp;
    for (int b = 0; b < V; b++) {
        const int32_t* sp = sorted_points + (size_t)b * L;
        const int32_t* si = start_index + (size_t)b * (ntiles + 2);
        const float* pk = packed + (size_t)b * N * ORC_REC;
#pragma omp parallel for schedule(dynamic, 8)
        for (int tile = 1; tile <= ntiles; tile++) {
            int start = si[tile], end = si[tile + 1];
            int tx = (tile - 1) % gx, ty = (tile - 1) / gx;
            /* [lane][i][c]: c = 0 row 2i, c = 1 row 2i+1 */
            double R[32][8][2], G_[32][8][2], B[32][8][2], T[32][8][2];
            int lc[32][8][2];
            for (int l = 0; l < 32; l++)
                for (int i = 0; i < PPT; i++)
                    for (int c = 0; c < 2; c++) { R[l][i][c] = G_[l][i][c] = B[l][i][c] = 0.0; T[l][i][c] = SCALER; lc[l][i][c] = 0; }
            if (start != -1) {
                int any_active = 1;
                for (int idx = 0; idx + start < end && any_active; idx++) {
                    const float* r = pk + (size_t)sp[start + idx] * ORC_REC;
                    double cr = r[5], cg = r[6], cb = r[7], ca = r[8];
                    any_active = 0;
                    for (int l = 0; l < 32; l++) {
                        int px = tx * TW + l % TW, py = ty * TH + (l / TW) * PPT * 2;
                        float dx = r[0] - (float)px, dy = r[1] - (float)py;
                        float bxcy = r[4] * dy + r[3] * dx;
                        float axby = r[2] * dx + r[3] * dy;
                        float cur_val = -0.5f * (dx * axby + dy * bxcy);
                        float cur_diff = bxcy - 0.5f * r[4];
                        float second_diff = -r[4];
                        for (int i = 0; i < PPT; i++)
                            for (int c = 0; c < 2; c++) {
                                double power = rh((double)cur_val);
                                cur_val += cur_diff;
                                cur_diff += second_diff;
                                int active = T[l][i][c] > rh(SCALER / 8192);
                                any_active |= active;
                                double alpha = hmul(ca, hexp(power));
                                int valid = active && (alpha >= 1.0 / 256);
                                alpha = hmin(255.0 / 256, alpha);
                                lc[l][i][c] += active;
                                if (!valid) alpha = 0.0;
                                double w = hmul(T[l][i][c], alpha);
                                R[l][i][c] = hfma(cr, w, R[l][i][c]);
                                G_[l][i][c] = hfma(cg, w, G_[l][i][c]);
                                B[l][i][c] = hfma(cb, w, B[l][i][c]);
                                T[l][i][c] = hmul(T[l][i][c], hsub(1.0, alpha));
                            }
                    }
                }
            }
            for (int l = 0; l < 32; l++)
                for (int i = 0; i < PPT; i++)
                    for (int c = 0; c < 2; c++) {
                        int x = tx * TW + l % TW, y = ty * TH + (l / TW) * PPT * 2 + 2 * i + c;
                        size_t o = (size_t)y * Wp + x;
                        float inv = 1.0f / 128;
                        img[((size_t)b * 3 + 0) * plane + o] = fminf((float)R[l][i][c] * inv, 1.0f);
                        img[((size_t)b * 3 + 1) * plane + o] = fminf((float)G_[l][i][c] * inv, 1.0f);
                        img[((size_t)b * 3 + 2) * plane + o] = fminf((float)B[l][i][c] * inv, 1.0f);
                        trans[(size_t)b * plane + o] = (float)T[l][i][c] * inv;
                        last[(size_t)b * plane + o] = (int16_t)lc[l][i][c];
                    }
        }
    }
}

/* shfl_down tree of warp_reduce_sum (raster.cu:81-91) over 32 lanes; half == 1: every add rounds to half */
static double warp_tree(double* v, int half)
{
    for (int s = 16; s >= 1; s >>= 1)
        for (int l = 0; l < 32; l++) {
            double o = (l + s < 32) ? v[l + s] : v[l];          /* shfl_down beyond the warp returns the lane's own value */
            v[l] = half ? hadd(v[l], o) : (double)((float)v[l] + (float)o);
        }
    return v[0];
}

ORC_API void orc_raster_backward_fp16(const int32_t* sorted_points, const int32_t* start_index, const float* packed,
                                      const float* final_T, const int16_t* last, const float* d_img,
                                      float inv_scaler, int V, int64_t L, int N, int H, int W, int TH, int TW,
                                      float* d_ndc /*[V,4,N]*/, float* d_inv_cov /*[V,2,2,N]*/, float* d_color /*[V,3,N]*/, float* d_opacity /*[1,N]*/)
{
    int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    int Wp = gx * TW, Hp = gy * TH, ntiles = gx * gy;
    int PPT = TH * TW / 64;
    size_t plane = (size_t)Hp * Wp;
    double* pg = (double*)calloc((size_t)V * N * 9, sizeof(double));      /* dx dy a00 a01 a11 r g b a */
    for (int b = 0; b < V; b++) {
        const int32_t* sp = sorted_points + (size_t)b * L;
        const int32_t* si = start_index + (size_t)b * (ntiles + 2);
        const float* pk = packed + (size_t)b * N * ORC_REC;
#pragma omp parallel for schedule(dynamic, 8)
        for (int tile = 1; tile <= ntiles; tile++) {
            int start = si[tile], end = si[tile + 1];
            if (start == -1 || start >= end) continue;          /* the reference walks an empty tile's neighbours here: not reproduced */
            int tx = (tile - 1) % gx, ty = (tile - 1) / gx;
            double T[32][8][2], Br[32][8][2], Bg[32][8][2], Bb[32][8][2], gr[32][8][2], gg[32][8][2], gb[32][8][2];
            int lst[32][8][2];
            int top = 0;
            for (int l = 0; l < 32; l++)
                for (int i = 0; i < PPT; i++)
                    for (int c = 0; c < 2; c++) {
                        int x = tx * TW + l % TW, y = ty * TH + (l / TW) * PPT * 2 + 2 * i + c;
                        size_t o = (size_t)y * Wp + x;
                        T[l][i][c] = rh((double)(final_T[(size_t)b * plane + o] * (float)SCALER));
                        gr[l][i][c] = rh(d_img[((size_t)b * 3 + 0) * plane + o]);
                        gg[l][i][c] = rh(d_img[((size_t)b * 3 + 1) * plane + o]);
                        gb[l][i][c] = rh(d_img[((size_t)b * 3 + 2) * plane + o]);
                        Br[l][i][c] = Bg[l][i][c] = Bb[l][i][c] = 0.0;
                        int lc = last[(size_t)b * plane + o];
                        lst[l][i][c] = lc == 0 ? 0 : lc - 1;             /* raster.cu:692-696 */
                        if (lst[l][i][c] > top) top = lst[l][i][c];
                    }
            if (top > end - start - 1) top = end - start - 1;
            for (int idx = top; idx >= 0; idx--) {
                int pid = sp[start + idx];
                const float* r = pk + (size_t)pid * ORC_REC;
                double cr = r[5], cg = r[6], cb = r[7], ca = r[8];
                double v_rg[2][32], v_ba[2][32];                  /* [component][lane] */
                double v_ic0[32], v_ic1[32], v_ic2[32], v_dx[32], v_dy[32];
                int any_nonzero = 0;
                for (int l = 0; l < 32; l++) {
                    int px = tx * TW + l % TW, py = ty * TH + (l / TW) * PPT * 2;
                    float dx = r[0] - (float)px, dy = r[1] - (float)py;
                    float bxcy = r[4] * dy + r[3] * dx;
                    float axby = r[2] * dx + r[3] * dy;
                    float cur_val = -0.5f * (dx * axby + dy * bxcy);
                    float cur_diff = bxcy - 0.5f * r[4];
                    float second_diff = -r[4];
                    double grad_r[2] = { 0, 0 }, grad_g[2] = { 0, 0 }, grad_b[2] = { 0, 0 }, grad_a[2] = { 0, 0 };
                    float grad_bxcy = 0, grad_nhc = 0, grad_basic = 0;
                    for (int i = 0; i < PPT; i++) {
                        double g_bxcy[2], g_nhc[2], d_pow[2];
                        for (int c = 0; c < 2; c++) {
                            double power = rh((double)cur_val);
                            cur_val += cur_diff;
                            cur_diff += second_diff;
                            double Gv = hexp(power);
                            double alpha = hmin(255.0 / 256, hmul(ca, Gv));
                            int valid = (alpha >= 1.0 / 256) && (idx <= lst[l][i][c]);
                            /* raster.cu:752-756: the row pair is gated on a warp-wide any; inside, alpha and G of an invalid pixel are
                             * masked to zero, and with alpha = 0 every update below is the identity -- so no gate is needed here */
                            if (!valid) { alpha = 0.0; Gv = 0.0; }
                            double t = hmin(SCALER, hmul(T[l][i][c], rh(1.0 / hsub(1.0, alpha))));
                            T[l][i][c] = t;
                            double at = hmul(alpha, t);
                            grad_r[c] = hfma(at, gr[l][i][c], grad_r[c]);
                            grad_g[c] = hfma(at, gg[l][i][c], grad_g[c]);
                            grad_b[c] = hfma(at, gb[l][i][c], grad_b[c]);
                            double d_alpha = 0.0;
                            d_alpha = hfma(hmul(hsub(cr, Br[l][i][c]), t), gr[l][i][c], d_alpha);
                            d_alpha = hfma(hmul(hsub(cg, Bg[l][i][c]), t), gg[l][i][c], d_alpha);
                            d_alpha = hfma(hmul(hsub(cb, Bb[l][i][c]), t), gb[l][i][c], d_alpha);
                            Br[l][i][c] = hfma(alpha, hsub(cr, Br[l][i][c]), Br[l][i][c]);
                            Bg[l][i][c] = hfma(alpha, hsub(cg, Bg[l][i][c]), Bg[l][i][c]);
                            Bb[l][i][c] = hfma(alpha, hsub(cb, Bb[l][i][c]), Bb[l][i][c]);
                            grad_a[c] = hfma(d_alpha, Gv, grad_a[c]);
                            double d_power = hmul(Gv, hmul(ca, d_alpha));
                            double row = (double)(2 * i + c);
                            d_pow[c] = d_power;
                            g_bxcy[c] = hmul(d_power, row);
                            g_nhc[c] = hmul(hmul(d_power, row), row);
                        }
                        grad_bxcy += ((float)g_bxcy[0] + (float)g_bxcy[1]);          /* fp32 accumulators, raster.cu:788-790 */
                        grad_nhc += ((float)g_nhc[0] + (float)g_nhc[1]);
                        grad_basic += ((float)d_pow[0] + (float)d_pow[1]);
                    }
                    if (grad_a[0] != 0.0 || grad_a[1] != 0.0) any_nonzero = 1;
                    v_rg[0][l] = hadd(grad_r[0], grad_r[1]); v_rg[1][l] = hadd(grad_g[0], grad_g[1]);
                    v_ba[0][l] = hadd(grad_b[0], grad_b[1]); v_ba[1][l] = hadd(grad_a[0], grad_a[1]);
                    v_ic0[l] = -0.5f * dx * dx * grad_basic;
                    v_ic1[l] = (-dx * dy * grad_basic + dx * grad_bxcy) * 0.5f;
                    v_ic2[l] = -0.5f * dy * dy * grad_basic + dy * grad_bxcy - 0.5f * grad_nhc;
                    v_dx[l] = (-r[2] * dx - r[3] * dy) * grad_basic + r[3] * grad_bxcy;
                    v_dy[l] = (-r[4] * dy - r[3] * dx) * grad_basic + r[4] * grad_bxcy;
                }
                if (!any_nonzero) continue;                              /* raster.cu:795 */
                double g5 = warp_tree(v_rg[0], 1), g6 = warp_tree(v_rg[1], 1), g7 = warp_tree(v_ba[0], 1), g8 = warp_tree(v_ba[1], 1);
                double g2 = warp_tree(v_ic0, 0), g3 = warp_tree(v_ic1, 0), g4 = warp_tree(v_ic2, 0);
                double g0 = warp_tree(v_dx, 0), g1 = warp_tree(v_dy, 0);
                double* g = pg + ((size_t)b * N + pid) * 9;
                double add[9] = { g0, g1, g2, g3, g4, g5, g6, g7, g8 };
                for (int k = 0; k < 9; k++) {
#pragma omp atomic
                    g[k] += add[k];
                }
            }
        }
    }
    float sc = inv_scaler * (1.0f / 128);
    for (int i = 0; i < N; i++) d_opacity[i] = 0.0f;
    for (int b = 0; b < V; b++)
        for (int i = 0; i < N; i++) {
            const double* g = pg + ((size_t)b * N + i) * 9;
            d_ndc[((size_t)b * 4 + 0) * N + i] = (float)g[0] * 0.5f * W * sc;
            d_ndc[((size_t)b * 4 + 1) * N + i] = (float)g[1] * 0.5f * H * sc;
            d_ndc[((size_t)b * 4 + 2) * N + i] = 0.0f;
            d_ndc[((size_t)b * 4 + 3) * N + i] = 0.0f;
            d_inv_cov[((size_t)b * 4 + 0) * N + i] = (float)g[2] * sc;
            d_inv_cov[((size_t)b * 4 + 1) * N + i] = (float)g[3] * sc;
            d_inv_cov[((size_t)b * 4 + 2) * N + i] = (float)g[3] * sc;
            d_inv_cov[((size_t)b * 4 + 3) * N + i] = (float)g[4] * sc;
            d_color[((size_t)b * 3 + 0) * N + i] = (float)g[5] * sc;
            d_color[((size_t)b * 3 + 1) * N + i] = (float)g[6] * sc;
            d_color[((size_t)b * 3 + 2) * N + i] = (float)g[7] * sc;
            if (b == 0) d_opacity[i] = (float)g[8] * sc;                 /* view 0 only, as the reference ("todo fix", :880) */
        }
    free(pg);
}

/* test hook: the binary16 rounding used above (checked against numpy.float16 in tests/test_oracle_fp16.py) */
ORC_API double orc_round_half(double x) { return rh(x); }
