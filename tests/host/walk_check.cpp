// Host fuzz of the two tile walks of the key emission (litegs_amd/csrc/lg_tilewalk.h compiled as sequential C++): the serial AccuTile walk
// that the projection counts with and dup_small_kernel emits with (walk_tiles), against the slice-at-a-time form dup_big_kernel evaluates
// with one lane per slice (slice_bounds).  The table is only fully written if the two agree for EVERY input the visibility test lets
// through -- including degenerate conics (huge / tiny / nearly singular covariances, centres far outside the image, opacity at the
// 1/255 threshold), which is what a trained cloud contains and a synthetic one does not.
// Built and run by tests/test_walk_host.py:  g++ -O2 -std=c++17 -ffp-contract=off -I litegs_amd/csrc tests/host/walk_check.cpp
#define LG_HOST_CHECK
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
typedef int hipError_t;
static inline int hipGetLastError() { return 0; }
enum { hipErrorInvalidValue = 1 };
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p |= v; return o; }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
using std::max;
using std::min;
#include "lg_tilewalk.h"

template <int TH, int TW>
static int check_one(float nx, float ny, float a, float b, float c, float o, int H, int W, long long& walked_tiles)
{
    const int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    int rect[4];
    const int cnt = lg_tile_count<TH, TW>(nx, ny, 1.0f, a, b, c, o, H, W, gx, gy, rect);
    if (cnt <= 0) return 0;
    SplatExtent e;
    splat_extent<TH, TW>(nx, ny, a, b, c, o, H, W, gx, gy, e);
    const WalkFrame f = walk_frame<TH, TW>(e);
    const int nsl = f.rect_max_u - f.rect_min_u;
    int K = 0;
    for (int i = 0; i < nsl; i++) K += ((float)(f.rect_min_u + i) * f.BLOCK_U + f.BLOCK_U <= f.bmax_u) ? 1 : 0;
    int run = 0;
    for (int i = 0; i < nsl; i++) {
        int mn, mx;
        slice_bounds(e, f, i, K, mn, mx);
        run += mx - mn;
    }
    walked_tiles += cnt;
    if (run != cnt) {
        std::printf("MISMATCH serial=%d sliced=%d  ndc=(%.9g, %.9g) conic=(%.9g, %.9g, %.9g) opacity=%.9g  %dx%d tile %dx%d  slices=%d K=%d\n",
                    cnt, run, nx, ny, a, b, c, o, W, H, TW, TH, nsl, K);
        return 1;
    }
    return 0;
}

int main(int argc, char** argv)
{
    const long long n = argc > 1 ? std::atoll(argv[1]) : 4000000;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<float> u01(0.0f, 1.0f);
    long long bad = 0, tiles = 0, visible = 0;
    for (long long it = 0; it < n; it++) {
        // covariance = R diag(s1^2, s2^2) R^T in pixels, log-uniform over ten decades; conic = its inverse
        const float s1 = std::exp((u01(rng) * 2 - 1) * 11.0f), s2 = (it % 3 == 0) ? s1 * std::exp((u01(rng) * 2 - 1) * 0.001f) : std::exp((u01(rng) * 2 - 1) * 11.0f);
        const float th = (it % 5 == 0) ? (float)(rng() % 8) * 0.78539816f + (u01(rng) - 0.5f) * 1e-4f : u01(rng) * 6.2831853f;
        const float cs = std::cos(th), sn = std::sin(th);
        const float c00 = cs * cs * s1 * s1 + sn * sn * s2 * s2, c01 = cs * sn * (s1 * s1 - s2 * s2), c11 = sn * sn * s1 * s1 + cs * cs * s2 * s2;
        const float det = c00 * c11 - c01 * c01;
        float a = c11 / det, b = -c01 / det, c = c00 / det;
        if (it % 11 == 0) b = (rng() & 1) ? std::sqrt(a * c) * (1.0f - 1e-7f * (float)(rng() % 64)) : -std::sqrt(a * c) * (1.0f - 1e-7f * (float)(rng() % 64));
        if (it % 13 == 0) b = 0.0f;
        if (it % 17 == 0) b = (rng() & 1) ? 1e-30f : -1e-30f;
        const float nx = (it % 7 == 0) ? ((rng() & 1) ? 1.3f : -1.3f) * (1.0f - 1e-7f * (float)(rng() % 4)) : (u01(rng) * 2 - 1) * 1.3f;
        const float ny = (it % 7 == 1) ? ((rng() & 1) ? 1.3f : -1.3f) : (u01(rng) * 2 - 1) * 1.3f;
        float o = (it % 4 == 0) ? 1.0f / 255 * (1.0f + 1e-6f * (float)(rng() % 100)) : u01(rng);
        if (it % 19 == 0) o = 1.0f;
        const int before = (int)bad;
        long long t0 = tiles;
        switch (it & 3) {
        case 0: bad += check_one<8, 16>(nx, ny, a, b, c, o, 1080, 1920, tiles); break;
        case 1: bad += check_one<8, 16>(nx, ny, a, b, c, o, 360, 640, tiles); break;
        case 2: bad += check_one<16, 16>(nx, ny, a, b, c, o, 1080, 1920, tiles); break;
        default: bad += check_one<8, 8>(nx, ny, a, b, c, o, 545, 980, tiles); break;
        }
        visible += tiles > t0;
        if (bad > 20 && before != bad) break;
    }
    std::printf("%lld splats, %lld with tiles, %lld tiles, %lld mismatches\n", n, visible, tiles, bad);
    return bad ? 1 : 0;
}
