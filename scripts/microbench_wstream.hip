// microbench_wstream.hip -- how fast can one MI355X stream the packed W4A8 weights, by access pattern?
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench_wstream.hip -o /tmp/mb_wstream
// Every variant reads the same `bytes` once and folds them into a checksum (so nothing is dead code).
//   P0  lane-linear 16 B loads (lane i reads base + 16 i): the ideal coalesced stream
//   P1  "64-byte row per lane": lane reads 4 x 16 B of its own 64 contiguous bytes, lanes 64 B apart (what the
//        split-K GEMM does for the weight operand)
//   P2  P1 with nontemporal loads
//   P3  LDS-DMA (global_load_lds_dwordx4) lane-linear into a per-wave LDS ring, then ds_read_b128 of the 64-byte rows
// Parameters swept: waves per block, blocks, loads in flight per lane (unroll).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int PAT, int UNROLL>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned char* __restrict__ w, size_t chunk_bytes, int nchunks,
                                                     unsigned* __restrict__ sink) {
    // work unit = 4 KiB chunk (one k-step of one 64-row unit); wave w of block b takes chunks (b*NW + w) + i*total_waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const int gw = blockIdx.x * NW + wave, tw = gridDim.x * NW;
    v4u acc = {0, 0, 0, 0};
    __shared__ __attribute__((aligned(16))) unsigned char lds[4][UNROLL][4096];
    for (int c0 = gw; c0 < nchunks; c0 += tw * UNROLL) {
        v4u r[UNROLL][4];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int c = c0 + u * tw;
            if (c < nchunks) {
                const unsigned char* base = w + (size_t)c * 4096;
                if (PAT == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[u][e] = *reinterpret_cast<const v4u*>(base + e * 1024 + lane * 16);
                } else if (PAT == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[u][e] = *reinterpret_cast<const v4u*>(base + lane * 64 + e * 16);
                } else if (PAT == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        r[u][e] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(base + lane * 64 + e * 16));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void*)(base + e * 1024 + lane * 16),
                            (__attribute__((address_space(3))) void*)(&lds[wave][u][e * 1024]), 16, 0, 0);
                }
            }
        }
        if (PAT == 3) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) r[u][e] = *reinterpret_cast<const v4u*>(&lds[wave][u][lane * 64 + e * 16]);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int c = c0 + u * tw;
            if (c < nchunks) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc ^= r[u][e];
            }
        }
    }
    unsigned x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x12345678u) sink[0] = x;   // practically never: keeps the loads alive
}

template <int PAT, int UNROLL>
float run(const unsigned char* w, size_t bytes, int blocks, int nw, unsigned* sink, int reps) {
    const int nchunks = (int)(bytes / 4096);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<PAT, UNROLL>), dim3(blocks), dim3(nw * 64), 0, 0, w, 4096, nchunks, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<PAT, UNROLL>), dim3(blocks), dim3(nw * 64), 0, 0, w + (size_t)(i % 8) * bytes, 4096, nchunks, sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const size_t bytes = 58720256;   // gate_up weights of one Llama-3-8B layer (28672 x 4096 / 2)
    unsigned char* w; unsigned* sink;
    CK(hipMalloc(&w, bytes * 8)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(w, 0x5a, bytes * 8));
    printf("pattern unroll blocks waves/blk   us      GB/s   (%.1f MB per launch, 8 rotating buffers)\n", bytes / 1e6);
    const int blocks_list[] = {64, 128, 256, 512};
    const int nw_list[] = {4};
#define RUN(P, U) for (int b : blocks_list) for (int nw : nw_list) { float us = run<P, U>(w, bytes, b, nw, sink, 24); printf("P%d      %d      %5d   %d        %7.2f  %7.1f\n", P, U, b, nw, us, bytes / us / 1e3); }
    RUN(0, 1) RUN(0, 4)
    RUN(1, 1) RUN(1, 4)
    RUN(2, 2)
    RUN(3, 4)
    return 0;
}
