// microbench_icache.hip -- does a kernel start with a cold instruction cache on every launch?
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench_icache.hip -o scripts/mb_icache
// One wave per workgroup executes a straight-line block of N independent VALU instructions (N * 8 bytes of code) TWICE inside
// the launch and stamps both passes with s_memtime; the kernel is launched several times back to back, alone (idle chip)
// and beside a streaming kernel.  pass 1 >> pass 2 on every launch = the instruction cache does not survive the launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
template <int KB>
__global__ __launch_bounds__(64) void code_block(float* out, unsigned long long* stamps) {
    float a = threadIdx.x, b = 1.0001f;
    unsigned long long t[3];
    for (int pass = 0; pass < 2; ++pass) {
        t[pass] = __builtin_amdgcn_s_memtime();
        // 256 v_fma (8 bytes each, VOP3) = 2 KiB per R256
#pragma unroll
        for (int i = 0; i < KB / 2; ++i) {
            R256(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));)
        }
    }
    t[2] = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = t[1] - t[0];
        stamps[blockIdx.x * 2 + 1] = t[2] - t[1];
    }
    out[blockIdx.x * 64 + threadIdx.x] = a;
}

__global__ void stream(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

template <int KB>
int run(int grid, bool busy, float* out, unsigned long long* st, float4* a, float4* b, size_t n) {
    std::vector<unsigned long long> h(grid * 2);
    for (int it = 0; it < 4; ++it) {
        if (busy) hipLaunchKernelGGL(stream, dim3(2048), dim3(256), 0, 0, a, b, n);
        hipLaunchKernelGGL(code_block<KB>, dim3(grid), dim3(64), 0, 0, out, st);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), st, grid * 16, hipMemcpyDeviceToHost));
        double p1 = 0, p2 = 0;
        for (int i = 0; i < grid; ++i) p1 += h[2 * i], p2 += h[2 * i + 1];
        printf("  %2d KiB of code, %4d workgroups, %s, launch %d: pass 1 %7.0f ticks, pass 2 %7.0f ticks (%.1f / %.1f per instruction)\n", KB,
               grid, busy ? "behind a streaming kernel" : "idle chip", it, p1 / grid, p2 / grid, p1 / grid / (KB * 128), p2 / grid / (KB * 128));
    }
    return 0;
}

int main() {
    float* out;
    unsigned long long* st;
    float4 *a, *b;
    const size_t n = (size_t)64 << 20;      // 1 GiB each
    CK(hipMalloc(&out, 4096 * 64 * 4));
    CK(hipMalloc(&st, 4096 * 16));
    CK(hipMalloc(&a, n * 16));
    CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 1, n * 16));
    for (int busy = 0; busy < 2; ++busy) {
        if (run<4>(256, busy, out, st, a, b, n)) return 1;
        if (run<16>(256, busy, out, st, a, b, n)) return 1;
        if (run<16>(1024, busy, out, st, a, b, n)) return 1;
    }
    return 0;
}
