// Developer micro-benchmark (not product): the shader clock the chip actually sustains under a chip-wide vector load -- the number every
// "fraction of the vector peak" in DESIGN.md divides by.  s_memtime counts shader clocks, s_memrealtime a constant 100 MHz reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) load_kernel(float* out, long long* clk, int iters, float seed)
{
    v2f a0 = { seed + threadIdx.x, seed }, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
    const v2f b = { 0.999f, 1.001f }, c = { 1e-3f, -1e-3f };
    const long long t0 = __builtin_readcyclecounter();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            a0 = a0 * b + c; a1 = a1 * b + c; a2 = a2 * b + c; a3 = a3 * b + c;          // v_pk_fma_f32
            if (MODE == 1) { a0.x = __builtin_amdgcn_exp2f(a0.x * 1e-3f); a1.y = __builtin_amdgcn_rcpf(a1.y + 2.0f); }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0.x + a0.y + a1.x + a1.y + a2.x + a2.y + a3.x + a3.y;
    if ((threadIdx.x & 63) == 0) { clk[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; clk[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }
}

int main()
{
    const int blocks = 256 * 8;                 // 8 waves per SIMD on every CU
    float* out; long long* clk;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
    CHECK(hipMalloc(&clk, sizeof(long long) * blocks * 4 * 2));
    std::vector<long long> h(blocks * 4 * 2);
    for (int mode = 0; mode < 2; mode++)
        for (int iters : { 2000, 20000, 100000 }) {
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(load_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0f);
            else hipLaunchKernelGGL(load_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0f);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), clk, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
            std::vector<double> mhz;
            for (int w = 0; w < blocks * 4; w++) mhz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100.0);
            std::sort(mhz.begin(), mhz.end());
            const double instr = (double)iters * 8 * (mode == 0 ? 4 : 8);         // vector instructions per wave (mode 1: + mul, exp, add, rcp)
            printf("mode %d (%s) iters %6d: kernel %8.3f ms; shader clock per wave: min %.0f  p50 %.0f  max %.0f MHz; %.2f clocks per vector instruction and SIMD (8 waves)\n",
                   mode, mode == 0 ? "v_pk_fma_f32 only" : "v_pk_fma_f32 + exp + rcp", iters, ms, mhz.front(), mhz[mhz.size() / 2], mhz.back(),
                   ms * 1e-3 * mhz[mhz.size() / 2] * 1e6 / (instr * 8));
        }
    return 0;
}
