// microbench_cufill.hip -- what bounds the rate at which ONE CU can fill its LDS by LDS-DMA?
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench_cufill.hip -o /tmp/mb_cufill
// One 512-thread workgroup per CU (8 waves), every wave keeps D 1-KiB DMA instructions in flight into its own LDS ring
// (nothing consumes the data: this is the fill path alone).  Swept:
//   grid      1 / 8 / 32 / 64 / 128 / 256 workgroups (1 CU alone ... the whole chip; workgroup i sits on XCD i % 8)
//   source    "l2": every workgroup re-reads the SAME 256 KiB (the GEMM's activation matrix: L2 hits after pass 0)
//             "hbm": every workgroup streams its own bytes once (the weights)
//             "mix": alternating 1-KiB pieces of both (the ring GEMM at M = 64: as many activation as weight bytes)
//   pattern   "line": a wave instruction covers 1 KiB contiguous (8 full 128-B lines)
//             "half": 16 rows x 64 B at a row stride of 4096 B (the activation pieces of the ring GEMM: 16 half lines,
//                     the other halves are fetched by the next stage)
//             "row128": 8 rows x 128 B at a row stride of 4096 B (full lines of 8 token rows)
//   D         DMA instructions in flight per wave (x 8 waves x 1 KiB per CU)
// Output: aggregate GB/s (hipEvents) and the per-workgroup mean GB/s from s_memtime (100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lptr_t;

template <int NT>
__device__ __forceinline__ void dma16(unsigned voff, const void* sbase, unsigned lds_addr) {
    if (NT)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

// SRC: 0 l2 (shared region), 1 hbm (own stream), 2 mix.  PAT: 0 line, 1 half, 2 row128 (applies to the shared region;
// the own stream is always contiguous).  region = 256 KiB = [64 rows][4096 B].
template <int SRC, int PAT, int D, int NT>
__global__ __launch_bounds__(512, 1) void fill(const unsigned char* __restrict__ shared, const unsigned char* __restrict__ own,
                                               int passes, unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem + wave * D * 1024;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // per-lane offset inside a piece
    unsigned off_line = lane * 16;
    unsigned off_half = (lane >> 2) * 4096 + (lane & 3) * 16;
    unsigned off_r128 = (lane >> 3) * 4096 + (lane & 7) * 16;
    // full lines whose halves sit 32 lanes apart ([half][row 8][64 B] image) / interleaved inside 16 lanes (rows 2a, 2a+1:
    // A0 B0 A1 B1, odd a swapped: the bank-conflict-free pair image of the GEMMs)
    unsigned off_split = ((lane >> 2) & 7) * 4096 + (lane >> 5) * 64 + (lane & 3) * 16;
    unsigned off_il;
    {
        const int P = lane >> 2, a = P >> 2, q = P & 3, half = (q >> 1) ^ (a & 1), b = q & 1;
        off_il = (2 * a + b) * 4096 + half * 64 + (lane & 3) * 16;
    }
    const unsigned char* mine = own + (size_t)blockIdx.x * passes * 262144;
    int slot = 0;
    for (int ps = 0; ps < passes; ++ps) {
        // 256 pieces of 1 KiB per pass; wave takes pieces wave, wave + 8, ...
        for (int i = 0; i < 32; ++i) {
            const int p = wave + 8 * i;
            const bool use_shared = SRC == 0 || (SRC == 2 && (i & 1) == 0);
            if (use_shared) {
                if (PAT == 0) dma16<0>(off_line, shared + (size_t)p * 1024, lds0 + slot * 1024);
                else if (PAT == 1) {     // piece p = (k-column kc = p >> 2 of 64 B, row block rb = p & 3)
                    dma16<0>(off_half, shared + (size_t)(p & 3) * 16 * 4096 + (p >> 2) * 64, lds0 + slot * 1024);
                } else {                 // piece p = (k-column kc = p >> 3 of 128 B, row block rb = p & 7 of 8 rows)
                    dma16<0>(PAT == 2 ? off_r128 : PAT == 3 ? off_split : off_il,
                             shared + (size_t)(p & 7) * 8 * 4096 + (p >> 3) * 128, lds0 + slot * 1024);
                }
            } else {
                dma16<NT>(off_line, mine + (size_t)ps * 262144 + (size_t)p * 1024, lds0 + slot * 1024);
            }
            slot = slot + 1 == D ? 0 : slot + 1;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        stamps[(blockIdx.x * 8 + wave) * 2 + 0] = t0;
        stamps[(blockIdx.x * 8 + wave) * 2 + 1] = t1;
    }
}

// ---- ds_read_b128 bank conflicts of the activation-operand images -------------------------------------------------
// lane (li = lane & 15, g = lane >> 4) reads 16 B of token row li, k-chunk g.  IMG 0: rows of 64 B, chunk position
// g ^ (li >> 2); IMG 1: the same with position g ^ ((-(li >> 2)) & 3); IMG 2: pair image (1 KiB piece = 8 rows x 2 halves,
// position P = 4a + 2(half ^ (a & 1)) + b of row 2a + b) with the swizzle of IMG 1; IMG 3: linear (lane * 16, conflict-free)
typedef int v4i __attribute__((ext_vector_type(4)));
template <int IMG>
__global__ __launch_bounds__(512, 1) void ldsread(int iters, int* __restrict__ sink, unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4, j = li >> 2;
    int addr;
    if (IMG == 0) addr = li * 64 + ((g ^ j) * 16);
    else if (IMG == 1) addr = li * 64 + ((g ^ ((0 - j) & 3)) * 16);
    else if (IMG == 2) {
        const int piece = li >> 3, r8 = li & 7, a = r8 >> 1, b = r8 & 1, half = 0;
        addr = piece * 1024 + (4 * a + 2 * (half ^ (a & 1)) + b) * 64 + ((g ^ ((0 - j) & 3)) * 16);
    } else addr = lane * 16;
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<int*>(smem)[i] = i;
    __syncthreads();
    const unsigned char* base = smem + wave * 8192;
    v4i acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const v4i v = *reinterpret_cast<const v4i*>(base + addr + m * 2048);
            acc ^= v;
        }
        asm volatile("" : "+v"(acc) :: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc[0] == 0x12345678) sink[threadIdx.x] = acc[1] ^ acc[2] ^ acc[3];
    if (threadIdx.x == 0) { stamps[0] = t0; stamps[1] = t1; }
}
template <int IMG>
void run_lds(const char* name, int* sink, unsigned long long* stamps) {
    auto k = ldsread<IMG>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 20000;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(1), dim3(512), 65536, 0, iters, sink, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double reads = (double)iters * 4 * 8;      // wave-instructions per CU
    printf("%-40s %.2f ns per ds_read_b128 wave-instruction per CU (%.1f B/ns)\n", name, best * 1e6 / reads, 1024.0 * reads / (best * 1e6));
}

typedef void (*kern_t)(const unsigned char*, const unsigned char*, int, unsigned long long*);

template <int SRC, int PAT, int D, int NT>
void run(const char* name, const unsigned char* shared, const unsigned char* own, unsigned long long* stamps, int passes) {
    kern_t k = fill<SRC, PAT, D, NT>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * D * 1024));
    const int grids[] = {1, 8, 32, 64, 128, 256};
    for (int g : grids) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        float best = 1e30f;
        std::vector<unsigned long long> h(g * 16);
        double wg_rate = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(g), dim3(512), 8 * D * 1024, 0, shared, own, passes, stamps);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) {
                best = ms;
                CK(hipMemcpy(h.data(), stamps, g * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                double s = 0;
                for (int b = 0; b < g; ++b) {
                    unsigned long long a = ~0ull, z = 0;
                    for (int w = 0; w < 8; ++w) {
                        a = h[(b * 8 + w) * 2] < a ? h[(b * 8 + w) * 2] : a;
                        z = h[(b * 8 + w) * 2 + 1] > z ? h[(b * 8 + w) * 2 + 1] : z;
                    }
                    s += (double)passes * 262144 / ((double)(z - a) * 10e-9) / 1e9;
                }
                wg_rate = s / g;
            }
        }
        const double bytes = (double)g * passes * 262144;
        printf("%-28s grid %3d  total %8.1f GB/s   per-WG (in-kernel) %6.1f GB/s   %.1f us\n", name, g, bytes / (best * 1e-3) / 1e9,
               wg_rate, best * 1e3);
    }
}

int main() {
    const int passes = 16;
    unsigned char *shared, *own;
    unsigned long long* stamps;
    const size_t own_bytes = (size_t)256 * passes * 262144;
    CK(hipMalloc(&shared, 262144));
    CK(hipMalloc(&own, own_bytes));
    CK(hipMalloc(&stamps, 256 * 16 * sizeof(unsigned long long)));
    CK(hipMemset(shared, 1, 262144));
    CK(hipMemset(own, 2, own_bytes));
    int* sink;
    CK(hipMalloc(&sink, 4096));
    run_lds<3>("lds linear (conflict-free)", sink, stamps);
    run_lds<0>("lds rows64 swizzle g^j (current)", sink, stamps);
    run_lds<1>("lds rows64 swizzle g^(-j)", sink, stamps);
    run_lds<2>("lds pair image", sink, stamps);
    run<0, 3, 8, 0>("l2 row128-split D=8", shared, own, stamps, passes);
    run<0, 4, 8, 0>("l2 row128-interleaved D=8", shared, own, stamps, passes);
    run<2, 3, 8, 1>("mix row128-split D=8 nt", shared, own, stamps, passes);
    run<2, 4, 8, 1>("mix row128-interleaved D=8 nt", shared, own, stamps, passes);
    run<2, 4, 8, 0>("mix row128-interleaved D=8 (no nt)", shared, own, stamps, passes);
    run<2, 1, 8, 0>("mix half D=8 (no nt)", shared, own, stamps, passes);
    return 0;
    run<0, 0, 8, 0>("l2 line D=8", shared, own, stamps, passes);
    run<0, 1, 8, 0>("l2 half D=8", shared, own, stamps, passes);
    run<0, 2, 8, 0>("l2 row128 D=8", shared, own, stamps, passes);
    run<0, 0, 16, 0>("l2 line D=16", shared, own, stamps, passes);
    run<0, 1, 16, 0>("l2 half D=16", shared, own, stamps, passes);
    run<0, 0, 4, 0>("l2 line D=4", shared, own, stamps, passes);
    run<1, 0, 8, 0>("hbm line D=8", shared, own, stamps, passes);
    run<1, 0, 8, 1>("hbm line D=8 nt", shared, own, stamps, passes);
    run<1, 0, 16, 1>("hbm line D=16 nt", shared, own, stamps, passes);
    run<2, 1, 8, 1>("mix half D=8 nt", shared, own, stamps, passes);
    run<2, 2, 8, 1>("mix row128 D=8 nt", shared, own, stamps, passes);
    run<2, 0, 8, 1>("mix line D=8 nt", shared, own, stamps, passes);
    run<2, 1, 16, 1>("mix half D=16 nt", shared, own, stamps, passes);
    run<2, 2, 16, 1>("mix row128 D=16 nt", shared, own, stamps, passes);
    return 0;
}
