// Developer micro-benchmark 3 (not product): scalar-ALU issue cost on gfx950 and how it overlaps with VALU issue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) bench(float* out, long long* cyc, int iters, float seed, int sv)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f, c = 0.999f;
    int s0 = sv, s1 = sv + 1, s2 = sv + 2, s3 = sv + 3, s4 = sv + 4, s5 = sv + 5, s6 = sv + 6, s7 = sv + 7;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#define VS(INS) asm volatile(INS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                                   "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "v"(b), "v"(c) : "scc")
        if constexpr (OP == 0) {          // 8 independent s_add_u32
            VS("s_add_u32 %8, %8, 1\n\ts_add_u32 %9, %9, 1\n\ts_add_u32 %10, %10, 1\n\ts_add_u32 %11, %11, 1\n\t"
               "s_add_u32 %12, %12, 1\n\ts_add_u32 %13, %13, 1\n\ts_add_u32 %14, %14, 1\n\ts_add_u32 %15, %15, 1\n\t");
        } else if constexpr (OP == 1) {   // 8 s_mov_b32
            VS("s_mov_b32 %8, %9\n\ts_mov_b32 %9, %10\n\ts_mov_b32 %10, %11\n\ts_mov_b32 %11, %12\n\t"
               "s_mov_b32 %12, %13\n\ts_mov_b32 %13, %14\n\ts_mov_b32 %14, %15\n\ts_mov_b32 %15, %8\n\t");
        } else if constexpr (OP == 2) {   // 8 fma interleaved with 8 s_add
            VS("v_fmac_f32 %0, %16, %17\n\ts_add_u32 %8, %8, 1\n\tv_fmac_f32 %1, %16, %17\n\ts_add_u32 %9, %9, 1\n\t"
               "v_fmac_f32 %2, %16, %17\n\ts_add_u32 %10, %10, 1\n\tv_fmac_f32 %3, %16, %17\n\ts_add_u32 %11, %11, 1\n\t"
               "v_fmac_f32 %4, %16, %17\n\ts_add_u32 %12, %12, 1\n\tv_fmac_f32 %5, %16, %17\n\ts_add_u32 %13, %13, 1\n\t"
               "v_fmac_f32 %6, %16, %17\n\ts_add_u32 %14, %14, 1\n\tv_fmac_f32 %7, %16, %17\n\ts_add_u32 %15, %15, 1\n\t");
        } else if constexpr (OP == 3) {   // 8 fma then 8 s_add (not interleaved)
            VS("v_fmac_f32 %0, %16, %17\n\tv_fmac_f32 %1, %16, %17\n\tv_fmac_f32 %2, %16, %17\n\tv_fmac_f32 %3, %16, %17\n\t"
               "v_fmac_f32 %4, %16, %17\n\tv_fmac_f32 %5, %16, %17\n\tv_fmac_f32 %6, %16, %17\n\tv_fmac_f32 %7, %16, %17\n\t"
               "s_add_u32 %8, %8, 1\n\ts_add_u32 %9, %9, 1\n\ts_add_u32 %10, %10, 1\n\ts_add_u32 %11, %11, 1\n\t"
               "s_add_u32 %12, %12, 1\n\ts_add_u32 %13, %13, 1\n\ts_add_u32 %14, %14, 1\n\ts_add_u32 %15, %15, 1\n\t");
        } else if constexpr (OP == 4) {   // 8 fma only (reference)
            VS("v_fmac_f32 %0, %16, %17\n\tv_fmac_f32 %1, %16, %17\n\tv_fmac_f32 %2, %16, %17\n\tv_fmac_f32 %3, %16, %17\n\t"
               "v_fmac_f32 %4, %16, %17\n\tv_fmac_f32 %5, %16, %17\n\tv_fmac_f32 %6, %16, %17\n\tv_fmac_f32 %7, %16, %17\n\t");
        } else if constexpr (OP == 5) {   // 8 s_and_b64 on pairs
            asm volatile("s_and_b64 s[20:21], s[22:23], s[24:25]\n\ts_and_b64 s[22:23], s[24:25], s[26:27]\n\ts_and_b64 s[24:25], s[26:27], s[28:29]\n\t"
                         "s_and_b64 s[26:27], s[28:29], s[30:31]\n\ts_and_b64 s[28:29], s[30:31], s[32:33]\n\ts_and_b64 s[30:31], s[32:33], s[34:35]\n\t"
                         "s_and_b64 s[32:33], s[34:35], s[20:21]\n\ts_and_b64 s[34:35], s[20:21], s[22:23]\n\t"
                         ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "scc");
        } else if constexpr (OP == 6) {   // 8 s_nop 0
            asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t");
        } else if constexpr (OP == 7) {   // 8 taken branches
            asm volatile("s_branch 1f\n\t1:\n\ts_branch 2f\n\t2:\n\ts_branch 3f\n\t3:\n\ts_branch 4f\n\t4:\n\ts_branch 5f\n\t5:\n\ts_branch 6f\n\t6:\n\ts_branch 7f\n\t7:\n\ts_branch 8f\n\t8:\n\t");
        } else if constexpr (OP == 8) {   // 8 x (v_cmp -> vcc ; s_and_saveexec ; restore): the `if (lane < 9)` pattern
            VS("s_and_saveexec_b64 s[20:21], vcc\n\ts_or_b64 exec, exec, s[20:21]\n\ts_and_saveexec_b64 s[20:21], vcc\n\ts_or_b64 exec, exec, s[20:21]\n\t"
               "s_and_saveexec_b64 s[20:21], vcc\n\ts_or_b64 exec, exec, s[20:21]\n\ts_and_saveexec_b64 s[20:21], vcc\n\ts_or_b64 exec, exec, s[20:21]\n\t");
        } else if constexpr (OP == 9) {   // v_readlane with sgpr index x8
            asm volatile("v_readlane_b32 %0, %8, %0\n\tv_readlane_b32 %1, %9, %1\n\tv_readlane_b32 %2, %10, %2\n\tv_readlane_b32 %3, %11, %3\n\t"
                         "v_readlane_b32 %4, %12, %4\n\tv_readlane_b32 %5, %13, %5\n\tv_readlane_b32 %6, %14, %6\n\tv_readlane_b32 %7, %15, %7\n\t"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            s0 &= 63; s1 &= 63; s2 &= 63; s3 &= 63; s4 &= 63; s5 &= 63; s6 &= 63; s7 &= 63;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
int run(const char* name, float* out, long long* cyc, int per = 8, int iters = 4000)
{
    printf("%-40s", name);
    for (int w : { 1, 2, 4, 8 }) {
        const int blocks = 256 * w;
        hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f, 3);
        CHECK(hipDeviceSynchronize());
        std::vector<long long> h(blocks * 4);
        CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        double m = (double)h[h.size() / 2] / iters;
        printf("  w=%d: %7.2f cyc/trip/wave (%6.2f/SIMD)", w, m, m / w);
    }
    printf("   [%d ops per trip]\n", per);
    return 0;
}

int main()
{
    float* out; long long* cyc;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 8 * 256));
    CHECK(hipMalloc(&cyc, sizeof(long long) * 256 * 8 * 4));
    run<4>("8 v_fmac", out, cyc);
    run<0>("8 s_add_u32", out, cyc);
    run<1>("8 s_mov_b32", out, cyc);
    run<5>("8 s_and_b64", out, cyc);
    run<2>("8 v_fmac + 8 s_add interleaved", out, cyc, 16);
    run<3>("8 v_fmac then 8 s_add", out, cyc, 16);
    run<6>("8 s_nop 0", out, cyc);
    run<7>("8 s_branch (taken, to next)", out, cyc);
    run<8>("4 x (s_and_saveexec + s_or exec)", out, cyc);
    run<9>("8 v_readlane_b32 (sgpr index) + 8 s_and", out, cyc, 16);
    return 0;
}
