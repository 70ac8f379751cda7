// How does the hardware place workgroups when a one-workgroup-per-CU kernel has between 1 and 2 rounds of work?
// Each workgroup (512 threads, ~100 KB of LDS so that only one fits a CU) streams `bytes` of its own slice of a buffer
// and records start / end timestamps and the XCC it ran on.  Prints, per grid size, the kernel time and how the late
// workgroups were distributed over the XCCs.
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench_rounds.hip -o scripts/mb_rounds
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512, 1) void stream_k(const uint4* __restrict__ src, size_t per_block_vec, unsigned long long* t0,
                                                   unsigned long long* t1, unsigned* xcc, unsigned* sink) {
    extern __shared__ unsigned char smem[];
    const unsigned long long a = wall_clock64();
    const uint4* p = src + (size_t)blockIdx.x * per_block_vec;
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < per_block_vec; i += 512) {
        const uint4 v = p[i];
        acc.x ^= v.x;
        acc.y ^= v.y;
        acc.z ^= v.z;
        acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;   // keep the loads
    smem[threadIdx.x] = (unsigned char)acc.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        t0[blockIdx.x] = a;
        t1[blockIdx.x] = wall_clock64();
        xcc[blockIdx.x] = id & 0xF;
    }
}

int main() {
    const size_t per_block = 512 * 1024;               // bytes streamed per workgroup
    const int maxb = 1024;
    uint4* src;
    hipMalloc(&src, per_block * maxb);
    hipMemset(src, 1, per_block * maxb);
    unsigned long long *t0, *t1;
    unsigned *xcc, *sink;
    hipMalloc(&t0, maxb * 8);
    hipMalloc(&t1, maxb * 8);
    hipMalloc(&xcc, maxb * 4);
    hipMalloc(&sink, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream_k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    int clk_khz = 100000;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0);
    for (int blocks : {224, 256, 288, 352, 376, 384, 448, 512, 640, 768}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        stream_k<<<blocks, 512, 100 * 1024>>>(src, per_block / 16, t0, t1, xcc, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        stream_k<<<blocks, 512, 100 * 1024>>>(src, per_block / 16, t0, t1, xcc, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h0(blocks), h1(blocks);
        std::vector<unsigned> hx(blocks);
        hipMemcpy(h0.data(), t0, blocks * 8, hipMemcpyDeviceToHost);
        hipMemcpy(h1.data(), t1, blocks * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hx.data(), xcc, blocks * 4, hipMemcpyDeviceToHost);
        const unsigned long long base = *std::min_element(h0.begin(), h0.end());
        const double us = 1e3 / clk_khz;               // microseconds per tick
        double first_end = 1e30, last_end = 0;
        for (int b = 0; b < blocks; ++b) {
            first_end = std::min(first_end, (h1[b] - base) * us);
            last_end = std::max(last_end, (h1[b] - base) * us);
        }
        int late[16] = {0}, all[16] = {0};
        double late_start_max = 0;
        for (int b = 0; b < blocks; ++b) {
            all[hx[b]]++;
            const double s = (h0[b] - base) * us;
            if (s > 0.5 * first_end) {
                late[hx[b]]++;
                late_start_max = std::max(late_start_max, s);
            }
        }
        printf("blocks %4d: kernel %7.1f us, first block done %6.1f us, last %6.1f us, latest start %6.1f us | per XCC total/late:",
               blocks, ms * 1e3, first_end, last_end, late_start_max);
        for (int x = 0; x < 8; ++x) printf(" %d/%d", all[x], late[x]);
        printf("\n");
    }
    return 0;
}
