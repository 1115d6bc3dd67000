// Shared device/host helpers for the litegs_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#ifndef LG_HOST_CHECK          // tests/host/*.cpp compile the exact-arithmetic headers as sequential C++ with their own shims
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#define LG_API extern "C" __attribute__((visibility("default")))
#define LG_WAVE 64

// Every launcher returns the hipError_t of the launch as int (0 == success).
#define LG_RETURN_LAST() return (int)hipGetLastError()

static inline int lg_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Argument validation of the C ABI: a binding that is not this repository's Python / ATen layer gets an error code for a missing
// required buffer instead of a fault inside a kernel (sizes are checked by each launcher; nullable arguments are documented in
// include/litegs_hip.h and not listed).
template <typename... P>
static inline bool lg_nonnull(P... p) { return (... && (p != nullptr)); }
#define LG_REQUIRE(...) do { if (!lg_nonnull(__VA_ARGS__)) return (int)hipErrorInvalidValue; } while (0)

// float -> int with v_cvt_i32_f32 semantics made explicit (NaN -> 0, saturating); the oracle's f2i twin.
__device__ __forceinline__ int lg_f2i(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

// Natural log with a fixed operation order: bit-identical to the CPU checker's twin (orc_logf) when this
// translation unit is compiled with -ffp-contract=off (binning.hip is).  Input: positive normal float.
__device__ __forceinline__ float lg_logf(float x)
{
    uint32_t ux = __float_as_uint(x);
    int e = (int)((ux >> 23) & 0xff) - 127;
    ux = (ux & 0x007fffffu) | 0x3f800000u;
    float m = __uint_as_float(ux);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * (0.40000972152f + w * 0.24279078841f);
    float t2 = z * (0.66666662693f + w * 0.28498786688f);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float dk = (float)e;
    return dk * 0.69313812256f + ((dk * 9.0580006145e-6f + (s * (hfsq + R) - hfsq)) + f);
}

__device__ __forceinline__ int lg_valid_len(const int* valid_length, int n)
{
    if (valid_length == nullptr) return n;
    int v = *valid_length;
    return v < n ? v : n;
}
