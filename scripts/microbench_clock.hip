// microbench_clock.hip -- what does s_memtime count?  (round 6: the bench line's `sustained_clock_ghz` is s_memtime ticks per
// 10 ns tick of s_memrealtime inside the GEMM kernel itself; this checks the method on loads whose clock is known to differ.)
//   light : one wave per CU spinning on VALU adds (almost no power) -> expect the boost clock (~2.4 GHz)
//   mfma  : four waves per CU issuing v_mfma_i32_16x16x64_i8 back to back (the matrix pipe's power) -> expect the DVFS-reduced clock
// Each workgroup reports s_memtime / s_memrealtime deltas over its life; median over the workgroups.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    v4i acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4i){0, 0, 0, 0};
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {3, 2, 1, (int)blockIdx.x};
    int x = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x));
        }
    }
    int s = x;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 0x7fffffff) out[0] = 1;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int MODE>
void run(const char* name, int threads, int iters) {
    const int wgs = 256;
    unsigned long long* d;
    hipMalloc(&d, wgs * 16);
    std::vector<unsigned long long> h(2 * wgs);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(threads), 0, 0, d, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, wgs * 16, hipMemcpyDeviceToHost);
        std::vector<double> g;
        for (int i = 0; i < wgs; ++i) g.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0));
        std::sort(g.begin(), g.end());
        printf("%-6s rep %d: kernel %.1f us (events, 4 launches / 4), s_memtime ticks %llu, realtime ticks %llu (x 10 ns = %.1f us): %.3f GHz median (min %.3f max %.3f)\n",
               name, rep, ms * 250.0, h[0], h[1], h[1] * 0.01, g[wgs / 2], g[0], g[wgs - 1]);
    }
    hipFree(d);
}

int main() {
    run<0>("light", 64, 20000);
    run<1>("mfma", 256, 4000);
    run<0>("light", 64, 20000);
    return 0;
}
