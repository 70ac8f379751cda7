// Round 5: is ONE wave per SIMD able to keep the INT8 matrix pipe busy?  scripts/microbench_mfma3.hip (round 2) said no - 2.09 POPS
// for a 128 x 128 wave tile - and the 4-wave / 512-register tile was dropped on that number.  Its ISA (hipcc -save-temps) shows why:
// with 256 accumulator registers the register allocator fails to coalesce the loop-carried accumulator tuples and rotates half of
// them through v_accvgpr_mov copies (212 copies + 108 s_nop per 64 MFMAs).  The measurement was of the compiler, not of the SIMD.
// Here the MFMAs are inline asm on FIXED accumulator registers a[0:255] (nothing for the allocator to decide):
//   0  64 MFMAs per round, nothing else            1..3  + that many independent VALU per MFMA (the issue-slot budget of the gap)
//   4  + 16 ds_read_b128 + 4 ds_read_b64 per round (the wide tile's operand reads) and 24 unpack VALU, one s_barrier per round
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench_mfma5.hip -o scripts/mb_mfma5 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

#define MFMA(ACC, A, B) \
    asm volatile("v_mfma_i32_16x16x64_i8 a[%c0:%c1], %2, %3, a[%c0:%c1]" ::"i"(ACC), "i"((ACC) + 3), "v"(A), "v"(B))

template <int VAR>
__global__ __launch_bounds__(256, 1) void kt(int* out, const v4i* in, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    asm volatile("" ::: "a0", "a255");                    // the descriptor must allocate the whole accumulator file
    v4i A[4], B[16];
    for (int i = 0; i < 4; ++i) A[i] = in[tid + 256 * i];
    for (int i = 0; i < 16; ++i) B[i] = in[tid + 256 * (4 + i)];
    for (int i = tid; i < 65536 / 4; i += 256) reinterpret_cast<int*>(smem)[i] = i * 0x01030507;
    __syncthreads();
    unsigned raw[8];
    for (int i = 0; i < 8; ++i) raw[i] = tid * 0x9E3779B9u + i;
    unsigned f0 = tid, f1 = tid * 3, f2 = tid * 5;
    const unsigned char* lb = smem + (tid & 63) * 16;
    for (int it = 0; it < iters; ++it) {
        if (VAR >= 4) __builtin_amdgcn_s_barrier();
#define STEP(M)                                                                                        \
    do {                                                                                               \
        if (VAR >= 4) {                                                                                \
            B[(M + 4) & 15] = *reinterpret_cast<const v4i*>(lb + (((M) + it) & 15) * 1024);           \
            if ((M) < 4) {                                                                             \
                const v2u r = *reinterpret_cast<const v2u*>(lb + 32768 + (((M) + it) & 3) * 512);     \
                raw[2 * (M)] ^= r.x;                                                                   \
                raw[2 * (M) + 1] ^= r.y;                                                               \
            }                                                                                          \
            if ((M) >= 4 && (M) < 8) {                                                                 \
                const int c = (M) - 4;                                                                 \
                A[c][0] = (int)(raw[2 * c] & 0x0F0F0F0Fu);                                             \
                A[c][1] = (int)((raw[2 * c] >> 4) & 0x0F0F0F0Fu);                                      \
                A[c][2] = (int)(raw[2 * c + 1] & 0x0F0F0F0Fu);                                         \
                A[c][3] = (int)((raw[2 * c + 1] >> 4) & 0x0F0F0F0Fu);                                  \
            }                                                                                          \
        }                                                                                              \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                \
            MFMA(((M) * 4 + c) * 4, A[c], B[M]);                                                       \
            if (VAR >= 1 && VAR <= 3) { f0 = f0 * 3 + 1; asm volatile("" : "+v"(f0)); }               \
            if (VAR >= 2 && VAR <= 3) { f1 = f1 ^ (f1 >> 3); asm volatile("" : "+v"(f1)); }           \
            if (VAR == 3) { f2 = f2 + 0x9E3779B9u; asm volatile("" : "+v"(f2)); }                     \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
        STEP(0); STEP(1); STEP(2); STEP(3); STEP(4); STEP(5); STEP(6); STEP(7);
        STEP(8); STEP(9); STEP(10); STEP(11); STEP(12); STEP(13); STEP(14); STEP(15);
#undef STEP
    }
    int s = (int)(f0 ^ f1 ^ f2);
    asm volatile("s_nop 15\n\ts_nop 15");
#pragma unroll
    for (int r = 0; r < 256; r += 37) {
        int x;
        asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(r));
        s += x;
    }
    out[blockIdx.x * 256 + tid] = s;
}

template <int VAR>
static void go(int* out, const v4i* in, const char* what) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kt<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {
        kt<VAR><<<blocks, 256, 98304>>>(out, in, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) kt<VAR><<<blocks, 256, 98304>>>(out, in, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)blocks * 4 * iters * 64 * (2.0 * 16 * 16 * 64) * 5;
        printf("pass %d %-84s %7.1f TOPS\n", pass, what, ops / (ms * 1e-3) / 1e12);
    }
}

int main() {
    int* out;
    v4i* in;
    hipMalloc(&out, 1024 * 256 * 4);
    hipMalloc(&in, 256 * 20 * 16);
    hipMemset(in, 0x35, 256 * 20 * 16);
    go<0>(out, in, "1 wave/SIMD, 256x64 wave tile, asm MFMA on fixed AGPRs: 64 MFMA per round, nothing else");
    go<1>(out, in, "  + 1 VALU per MFMA");
    go<2>(out, in, "  + 2 VALU per MFMA");
    go<3>(out, in, "  + 3 VALU per MFMA");
    go<4>(out, in, "  64 MFMA + 16 b128 + 4 b64 LDS reads + 16 unpack VALU + s_barrier per round");
    return 0;
}
