// microbench_ldsdma_lat.hip -- what does an ordinary LDS read cost while the CU's waves have LDS-DMA bursts in flight?
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench_ldsdma_lat.hip -o /tmp/mb_ldsdma_lat
// (round 4: the KV4 decode attention trace shows the first ds_read_b128 after the first-round DMA burst taking ~3 000 cycles)
// 512 workgroups x 8 waves, two workgroups per CU (70 KiB LDS each), like the attention launch.
//   waves 0-6 "loaders": BURST 1-KiB LDS-DMA instructions each (nt, own HBM stream), then wait for them; REPS times
//   wave 7 "prober": times ds_read_b128 / ds_bpermute / a scalar-cache-hit s_load with s_memtime, continuously, while the
//                    loaders run; mode 1: the prober itself issues a burst first (its own DMA outstanding while it reads)
// Output per mode: mean and max latency (s_memtime ticks) of each probe, and the loaders' time per burst.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(unsigned voff, const void* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

template <int BURST>
__global__ __launch_bounds__(512, 4) void probe(const unsigned char* __restrict__ src, int reps, int mode, int loaders_on,
                                                unsigned long long* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[8 * 8 * 1024];
    __shared__ __attribute__((aligned(16))) unsigned int s_probe[256];
    __shared__ int s_done;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem + wave * 8 * 1024;
    if (threadIdx.x < 256) s_probe[threadIdx.x] = threadIdx.x;
    if (threadIdx.x == 0) s_done = 0;
    __syncthreads();
    const unsigned char* mine = src + ((size_t)blockIdx.x * 8 + wave) * (size_t)reps * BURST * 1024;
    const unsigned voff = lane * 16;
    if (wave < 7) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (loaders_on) {
            for (int r = 0; r < reps; ++r) {
#pragma unroll
                for (int d = 0; d < BURST; ++d) dma16(voff, mine + ((size_t)r * BURST + d) * 1024, lds0 + (d & 7) * 1024);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            for (int r = 0; r < reps * 40; ++r) __builtin_amdgcn_s_sleep(8);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0 && wave == 0) out[blockIdx.x * 16 + 8] = (t1 - t0) / (unsigned long long)reps;
        if (lane == 0) __hip_atomic_fetch_add(&s_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        unsigned long long sum[3] = {0, 0, 0}, mx[3] = {0, 0, 0}, n = 0;
        const unsigned addr = (unsigned)(size_t)(lptr_t)s_probe + (lane & 15) * 16;
        while (*(volatile __attribute__((address_space(3))) int*)(&s_done) < 7) {
            if (mode == 1) {
#pragma unroll
                for (int d = 0; d < BURST; ++d) dma16(voff, mine + (size_t)((n * BURST + d) % (reps * BURST)) * 1024, lds0 + (d & 7) * 1024);
            }
            // (a) ds_read_b128
            unsigned long long a0 = __builtin_amdgcn_s_memtime();
            v4u x;
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(addr) : "memory");
            unsigned long long a1 = __builtin_amdgcn_s_memtime();
            // (b) ds_bpermute
            int y = __builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, (int)x[0]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y)::"memory");
            unsigned long long a2 = __builtin_amdgcn_s_memtime();
            // (c) scalar load of a kernel-argument-adjacent word (scalar cache hit after the first time)
            int z;
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(z) : "s"(src) : "memory");
            unsigned long long a3 = __builtin_amdgcn_s_memtime();
            const unsigned long long d0 = a1 - a0, d1 = a2 - a1, d2 = a3 - a2;
            sum[0] += d0, sum[1] += d1, sum[2] += d2;
            mx[0] = d0 > mx[0] ? d0 : mx[0], mx[1] = d1 > mx[1] ? d1 : mx[1], mx[2] = d2 > mx[2] ? d2 : mx[2];
            ++n;
            if (mode == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" ::"v"(y), "s"(z));
        }
        if (lane == 0) {
            for (int i = 0; i < 3; ++i) {
                out[blockIdx.x * 16 + 2 * i] = n ? sum[i] / n : 0;
                out[blockIdx.x * 16 + 2 * i + 1] = mx[i];
            }
            out[blockIdx.x * 16 + 6] = n;
        }
    }
}

int main(int argc, char** argv) {
    const int grid = 512, reps = 3;
    constexpr int BURST = 24;
    size_t bytes = (size_t)grid * 8 * reps * BURST * 1024;
    unsigned char* src;
    unsigned long long* out;
    CK(hipMalloc(&src, bytes));
    CK(hipMemset(src, 1, bytes));
    CK(hipMalloc(&out, grid * 16 * 8));
    std::vector<unsigned long long> h(grid * 16);
    for (int loaders = 0; loaders < 2; ++loaders)
        for (int mode = 0; mode < 2; ++mode) {
            for (int it = 0; it < 3; ++it) {
                CK(hipMemset(out, 0, grid * 16 * 8));
                hipLaunchKernelGGL(probe<BURST>, dim3(grid), dim3(512), 0, 0, src, reps, mode, loaders, out);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h.data(), out, grid * 16 * 8, hipMemcpyDeviceToHost));
            double m[7] = {0, 0, 0, 0, 0, 0, 0}, burst = 0;
            for (int b = 0; b < grid; ++b) {
                for (int i = 0; i < 7; ++i) m[i] += (double)h[b * 16 + i] / grid;
                burst += (double)h[b * 16 + 8] / grid;
            }
            printf("loaders %s, prober %s its own burst: ds_read_b128 mean %.0f (mean of max %.0f)  ds_bpermute %.0f (%.0f)  s_load hit %.0f (%.0f)  "
                   "probes/wg %.0f  loader ticks per %d KiB burst %.0f\n",
                   loaders ? "ON " : "off", mode ? "WITH" : "without", m[0], m[1], m[2], m[3], m[4], m[5], m[6], BURST, burst);
        }
    return 0;
}
