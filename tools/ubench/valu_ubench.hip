// Developer micro-benchmark (not product): issue cost of the VALU / cross-lane instructions the blend kernels are made of, on gfx950.
// For each op: 8 independent chains per lane, `iters` loop trips; shader cycles per wave-instruction seen by one wave (s_memtime)
// at 1, 2, 4, 8 resident waves per SIMD -> per-SIMD throughput = cycles_per_instr / waves.
//   hipcc --offload-arch=gfx950 -O3 -o valu_ubench valu_ubench.hip && ./valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(256) bench(float* out, long long* cyc, int iters, float seed)
{
    float a[8], b = seed, c = 0.999f;
    v2f p[8], pb = { seed, seed }, pc = { 0.999f, 0.999f };
#pragma unroll
    for (int k = 0; k < 8; k++) { a[k] = seed + k + threadIdx.x; p[k] = v2f{ a[k], a[k] + 1.0f }; }
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if constexpr (OP == 0) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "v"(b));
            REP8(X)
#undef X
        } else if constexpr (OP == 1) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pc), "v"(pb));
            REP8(X)
#undef X
        } else if constexpr (OP == 2) {
#define X(k) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
            REP8(X)
#undef X
        } else if constexpr (OP == 3) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : "vcc");
            REP8(X)
#undef X
        } else if constexpr (OP == 4) {
#define X(k) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
            REP8(X)
#undef X
        } else if constexpr (OP == 5) {
#define X(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
            REP8(X)
#undef X
        } else if constexpr (OP == 6) {
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[0]), "+v"(a[1]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[2]), "+v"(a[3]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[4]), "+v"(a[5]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[6]), "+v"(a[7]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[1]), "+v"(a[2]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[3]), "+v"(a[4]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[5]), "+v"(a[6]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[7]), "+v"(a[0]));
        } else if constexpr (OP == 7) {
#define X(k) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,4)" : "+v"(a[k]));
            REP8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (OP == 8) {
#define X(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
            REP8(X)
#undef X
        } else if constexpr (OP == 9) {
#define X(k) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
            REP8(X)
#undef X
        } else if constexpr (OP == 10) {
#define X(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
            REP8(X)
#undef X
        } else if constexpr (OP == 11) {      // dependent chain: one accumulator, 8 back-to-back dependent fmas
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c), "v"(b));
            REP8(X)
#undef X
        } else if constexpr (OP == 12) {      // dependent chain through DPP
#define X(k) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[0]));
            REP8(X)
#undef X
        } else if constexpr (OP == 13) {      // dependent chain through ds_swizzle
#define X(k) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,4)\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[0]));
            REP8(X)
#undef X
        } else if constexpr (OP == 14) {      // dependent chain through permlane32_swap
#define X(k) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a[0]), "+v"(a[1]));
            REP8(X)
#undef X
        } else if constexpr (OP == 15) {      // fma with an SGPR operand
            float s = __builtin_amdgcn_readfirstlane(seed);
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "s"(s), "v"(b));
            REP8(X)
#undef X
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float r = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) r += a[k] + p[k].x + p[k].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name, float* out, long long* cyc)
{
    const int iters = 4000;
    printf("%-28s", name);
    for (int w : { 1, 2, 4, 8 }) {
        const int blocks = 256 * w;
        hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2] / (iters * 8.0);
        // wall-clock view: SIMD-cycles per wave-instruction assuming 2.4 GHz and perfectly even placement
        const double wall = ms * 1e-3 * 2.4e9 / (iters * 8.0 * w);
        printf("  w=%d: %6.2f cyc/inst/wave (%5.2f /SIMD, wall %5.2f)", w, med, med / w, wall);
    }
    printf("\n");
}

int main()
{
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * 256 * 8 * 256);
    hipMalloc(&cyc, sizeof(long long) * 256 * 8 * 4);
    run<0>("v_fma_f32", out, cyc);
    run<15>("v_fma_f32 (sgpr src)", out, cyc);
    run<10>("v_mul_f32", out, cyc);
    run<1>("v_pk_fma_f32", out, cyc);
    run<8>("v_pk_mul_f32", out, cyc);
    run<3>("v_cndmask_b32", out, cyc);
    run<2>("v_add_f32_dpp quad_perm", out, cyc);
    run<9>("v_add_f32_dpp row_mirror", out, cyc);
    run<4>("v_exp_f32", out, cyc);
    run<5>("v_rcp_f32", out, cyc);
    run<6>("v_permlane32_swap_b32", out, cyc);
    run<7>("ds_swizzle_b32 (8 + wait)", out, cyc);
    run<11>("dep chain v_fma_f32", out, cyc);
    run<12>("dep chain dpp add", out, cyc);
    run<13>("dep chain ds_swizzle", out, cyc);
    run<14>("dep chain permlane32_swap", out, cyc);
    return 0;
}
