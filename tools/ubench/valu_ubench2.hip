// Developer micro-benchmark 2 (not product): select / compare / integer / cross-lane costs and scalar-load latency on gfx950.
// Each op: ONE asm block of 8 independent instructions per loop trip (no compiler-inserted hazard nops in between).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) bench(float* out, long long* cyc, int iters, float seed, unsigned long long mask, const int* chase)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f, c = 0.999f;
    int sidx = 0;
    long long t0 = __builtin_readcyclecounter();
    long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#define OPS8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(mask) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35")
        if constexpr (OP == 0) {
#define I(k) "v_fmac_f32 %" #k ", %8, %9\n\t"
            OPS8(I);
#undef I
        } else if constexpr (OP == 1) {
#define I(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, %10\n\t"
            OPS8(I);
#undef I
        } else if constexpr (OP == 2) {
#define I(k) "v_cndmask_b32_e64 %" #k ", 0, %" #k ", %10\n\t"
            OPS8(I);
#undef I
        } else if constexpr (OP == 3) {       // compare into distinct SGPR pairs
            asm volatile("v_cmp_le_f32_e64 s[20:21], %0, %8\n\tv_cmp_le_f32_e64 s[22:23], %1, %8\n\tv_cmp_le_f32_e64 s[24:25], %2, %8\n\t"
                         "v_cmp_le_f32_e64 s[26:27], %3, %8\n\tv_cmp_le_f32_e64 s[28:29], %4, %8\n\tv_cmp_le_f32_e64 s[30:31], %5, %8\n\t"
                         "v_cmp_le_f32_e64 s[32:33], %6, %8\n\tv_cmp_le_f32_e64 s[34:35], %7, %8\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
        } else if constexpr (OP == 4) {
#define I(k) "v_and_b32 %" #k ", %" #k ", %8\n\t"
            OPS8(I);
#undef I
        } else if constexpr (OP == 5) {
#define I(k) "v_ashrrev_i32 %" #k ", 31, %" #k "\n\t"
            OPS8(I);
#undef I
        } else if constexpr (OP == 6) {
#define I(k) "v_min_f32 %" #k ", %" #k ", %8\n\t"
            OPS8(I);
#undef I
        } else if constexpr (OP == 7) {
#define I(k) "v_add_f32 %" #k ", %" #k ", %8\n\t"
            OPS8(I);
#undef I
        } else if constexpr (OP == 8) {       // fma each followed by s_nop 0
#define I(k) "v_fmac_f32 %" #k ", %8, %9\n\ts_nop 0\n\t"
            OPS8(I);
#undef I
        } else if constexpr (OP == 9) {       // fma each followed by s_nop 1
#define I(k) "v_fmac_f32 %" #k ", %8, %9\n\ts_nop 1\n\t"
            OPS8(I);
#undef I
        } else if constexpr (OP == 10) {
#define I(k) "v_mov_b32_dpp %" #k ", %" #k " row_mirror row_mask:0xf bank_mask:0xf\n\t"
            OPS8(I);
#undef I
        } else if constexpr (OP == 11) {
            asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                         "s_nop 1\n\tv_permlane16_swap_b32 %1, %2\n\tv_permlane16_swap_b32 %3, %4\n\tv_permlane16_swap_b32 %5, %6\n\tv_permlane16_swap_b32 %7, %0\n\ts_nop 1\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (OP == 12) {
#define I(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n\t"
            OPS8(I);
#undef I
        } else if constexpr (OP == 13) {      // cndmask with VCC (e32)
            asm volatile("s_mov_b64 vcc, %8\n\tv_cndmask_b32 %0, %0, %9, vcc\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_cndmask_b32 %2, %2, %9, vcc\n\tv_cndmask_b32 %3, %3, %9, vcc\n\t"
                         "v_cndmask_b32 %4, %4, %9, vcc\n\tv_cndmask_b32 %5, %5, %9, vcc\n\tv_cndmask_b32 %6, %6, %9, vcc\n\tv_cndmask_b32 %7, %7, %9, vcc\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(mask), "v"(b) : "vcc");
        } else if constexpr (OP == 14) {      // dependent scalar loads (pointer chase through the scalar cache)
            for (int k = 0; k < 8; k++) sidx = __builtin_amdgcn_readfirstlane(chase[sidx]);
        } else if constexpr (OP == 15) {      // dependent vector loads (L2 / L1 hit)
            int v = sidx + (threadIdx.x & 63);
            for (int k = 0; k < 8; k++) v = chase[v & 4095];
            sidx = v & 63;
        } else if constexpr (OP == 16) {      // v_readfirstlane x8
            int s;
            asm volatile("v_readfirstlane_b32 %0, %1\n\tv_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %0, %3\n\tv_readfirstlane_b32 %0, %4\n\t"
                         "v_readfirstlane_b32 %0, %5\n\tv_readfirstlane_b32 %0, %6\n\tv_readfirstlane_b32 %0, %7\n\tv_readfirstlane_b32 %0, %8\n\t"
                         : "=s"(s) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            sidx += s & 1;
        } else if constexpr (OP == 17) {      // global atomic add f32, 9 lanes, distinct lines per wave
            if ((threadIdx.x & 63) < 9) {
                float* p = out + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (threadIdx.x & 63)) ;
                for (int k = 0; k < 8; k++) __hip_atomic_fetch_add(p + ((i * 8 + k) & 1023) * 4096, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + sidx;
    if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }
}

template <int OP>
int run(const char* name, float* out, long long* cyc, const int* chase, int iters = 4000)
{
    printf("%-34s", name);
    for (int w : { 1, 2, 4, 8 }) {
        const int blocks = 256 * w;
        hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f, 0x5555555555555555ull, chase);
        CHECK(hipDeviceSynchronize());
        std::vector<long long> h(blocks * 8);
        CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
        std::vector<double> per, clk;
        for (int k = 0; k < blocks * 4; k++) { per.push_back((double)h[2 * k] / (iters * 8.0)); clk.push_back((double)h[2 * k] / (double)h[2 * k + 1] * 100.0); }
        std::sort(per.begin(), per.end()); std::sort(clk.begin(), clk.end());
        printf("  w=%d: %6.2f cyc/wave (%5.2f/SIMD) clk %4.0f", w, per[per.size() / 2], per[per.size() / 2] / w, clk[clk.size() / 2]);
    }
    printf("\n");
    return 0;
}

int main()
{
    float* out; long long* cyc; int* chase;
    CHECK(hipMalloc(&out, sizeof(float) * 4096 * 1024 + 4096 * 256 * 8));
    CHECK(hipMemset(out, 0, sizeof(float) * 4096 * 1024));
    CHECK(hipMalloc(&cyc, sizeof(long long) * 256 * 8 * 8));
    std::vector<int> hc(4096);
    for (int i = 0; i < 4096; i++) hc[i] = (i * 67 + 13) & 4095;
    CHECK(hipMalloc(&chase, 4096 * 4));
    CHECK(hipMemcpy(chase, hc.data(), 4096 * 4, hipMemcpyHostToDevice));
    run<0>("v_fmac_f32 (VOP2)", out, cyc, chase);
    run<7>("v_add_f32", out, cyc, chase);
    run<6>("v_min_f32", out, cyc, chase);
    run<12>("v_med3_f32", out, cyc, chase);
    run<4>("v_and_b32", out, cyc, chase);
    run<5>("v_ashrrev_i32", out, cyc, chase);
    run<1>("v_cndmask_b32_e64 v,v,sgpr", out, cyc, chase);
    run<2>("v_cndmask_b32_e64 0,v,sgpr", out, cyc, chase);
    run<13>("v_cndmask_b32 vcc", out, cyc, chase);
    run<3>("v_cmp_le_f32_e64 -> sgpr pair", out, cyc, chase);
    run<8>("v_fmac + s_nop 0", out, cyc, chase);
    run<9>("v_fmac + s_nop 1", out, cyc, chase);
    run<10>("v_mov_b32_dpp row_mirror", out, cyc, chase);
    run<11>("v_permlane16_swap_b32", out, cyc, chase);
    run<16>("v_readfirstlane_b32", out, cyc, chase);
    run<14>("dep chain s_load_dword (K$)", out, cyc, chase, 500);
    run<15>("dep chain global_load_dword", out, cyc, chase, 500);
    run<17>("global_atomic_add_f32 9 lanes", out, cyc, chase, 200);
    return 0;
}
