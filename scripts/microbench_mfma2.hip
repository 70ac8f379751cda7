// How close can ONE workgroup of 8 waves per CU (2 waves per SIMD, 128 accumulator VGPRs per lane - the tiled W4A8
// GEMM's shape) get to the int8 MFMA peak?  Variants:
//   0  32 independent v_mfma_i32_16x16x64_i8 per round on 4 A x 8 B register operands (no other work)
//   1  + the tiled kernel's VALU work per round (32 and/shift ops)
//   2  + 12 LDS operand reads per round (8 x b128 + 4 x b64) feeding the B operands
//   3  + one s_barrier per round
//   4  variant 0 with AGPR-free but 4 waves per SIMD is impossible (register budget) - instead: 16 accumulators,
//      2 workgroups per CU (4 waves per SIMD)
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench_mfma2.hip -o scripts/mb_mfma2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

template <int VAR>
__global__ __launch_bounds__(512, 1) void kt(int* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    v4i A[4], B[8];
    for (int i = 0; i < 4; ++i) A[i] = (v4i){tid * 0x01010101 + i, 0x11121314 + i, 0x21222324 * (i + 1), 0x31323334 + tid};
    for (int i = 0; i < 8; ++i) B[i] = (v4i){0x0a0b0c0d + i, tid * 0x00010203 + i, 0x2a2b2c2d * (i + 1), 0x3a3b3c3d + tid};
    for (int i = tid; i < 32768 / 4; i += 512) reinterpret_cast<int*>(smem)[i] = i * 0x01030507;
    __syncthreads();
    v4i acc[8][4];
    for (int m = 0; m < 8; ++m)
        for (int c = 0; c < 4; ++c) acc[m][c] = (v4i){0, 0, 0, 0};
    unsigned raw[8];
    for (int i = 0; i < 8; ++i) raw[i] = tid * 0x9E3779B9u + i;
    const unsigned char* lb = smem + (tid & 63) * 16 + (tid >> 6) * 2048;
    for (int it = 0; it < iters; ++it) {
        if (VAR >= 3) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (VAR >= 2) {
                B[m] = *reinterpret_cast<const v4i*>(lb + ((m + it) & 7) * 1024);
                if (m < 4) {
                    const v2u r = *reinterpret_cast<const v2u*>(lb + 16384 + ((m + it) & 3) * 512);
                    raw[2 * m] ^= r.x;
                    raw[2 * m + 1] ^= r.y;
                }
            }
            if (VAR >= 1 && m >= 2 && m < 6) {      // unpack one A operand: 4 x (and, shift+and)
                const int c = m - 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned x = raw[(2 * c + e) & 7];
                    A[c][e] = (int)((e & 1) ? ((x >> 4) & 0x0F0F0F0Fu) : (x & 0x0F0F0F0Fu));
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[m][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[c], B[m], acc[m][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    int s = 0;
    for (int m = 0; m < 8; ++m)
        for (int c = 0; c < 4; ++c) s += acc[m][c][0] + acc[m][c][1] + acc[m][c][2] + acc[m][c][3];
    out[blockIdx.x * 512 + tid] = s;
}

// 16 accumulators, 2 workgroups per CU
__global__ __launch_bounds__(512, 2) void k4(int* out, int iters) {
    const int tid = threadIdx.x;
    v4i A[4], B[4];
    for (int i = 0; i < 4; ++i) A[i] = (v4i){tid * 0x01010101 + i, 0x11121314 + i, 0x21222324 * (i + 1), 0x31323334 + tid};
    for (int i = 0; i < 4; ++i) B[i] = (v4i){0x0a0b0c0d + i, tid * 0x00010203 + i, 0x2a2b2c2d * (i + 1), 0x3a3b3c3d + tid};
    v4i acc[4][4];
    for (int m = 0; m < 4; ++m)
        for (int c = 0; c < 4; ++c) acc[m][c] = (v4i){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[m][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[c], B[m], acc[m][c], 0, 0, 0);
    }
    int s = 0;
    for (int m = 0; m < 4; ++m)
        for (int c = 0; c < 4; ++c) s += acc[m][c][0] + acc[m][c][1] + acc[m][c][2] + acc[m][c][3];
    out[blockIdx.x * 512 + tid] = s;
}

// variant 5: the same wave tile (128 tokens x 64 channels, 128 accumulator registers, 2 waves per SIMD) out of
// v_mfma_i32_32x32x32_i8: 4 x 2 accumulators of 16 registers, 16 MFMAs of 16 passes per 64-k round instead of 32 of 4 -
// does the longer instruction close the issue gap two waves per SIMD leave with the short one?
typedef int v16i __attribute__((ext_vector_type(16)));
template <int VAR>
__global__ __launch_bounds__(512, 1) void k32(int* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    v4i A[4], B[8];
    for (int i = 0; i < 4; ++i) A[i] = (v4i){tid * 0x01010101 + i, 0x11121314 + i, 0x21222324 * (i + 1), 0x31323334 + tid};
    for (int i = 0; i < 8; ++i) B[i] = (v4i){0x0a0b0c0d + i, tid * 0x00010203 + i, 0x2a2b2c2d * (i + 1), 0x3a3b3c3d + tid};
    for (int i = tid; i < 32768 / 4; i += 512) reinterpret_cast<int*>(smem)[i] = i * 0x01030507;
    __syncthreads();
    v16i acc[4][2];
    for (int m = 0; m < 4; ++m)
        for (int c = 0; c < 2; ++c)
            for (int r = 0; r < 16; ++r) acc[m][c][r] = 0;
    unsigned raw[8];
    for (int i = 0; i < 8; ++i) raw[i] = tid * 0x9E3779B9u + i;
    const unsigned char* lb = smem + (tid & 63) * 16 + (tid >> 6) * 2048;
    for (int it = 0; it < iters; ++it) {
        if (VAR >= 3) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < 8; ++m) {              // m = (token tile m >> 1, k half m & 1)
            if (VAR >= 2) {
                B[m] = *reinterpret_cast<const v4i*>(lb + ((m + it) & 7) * 1024);
                if (m < 4) {
                    const v2u r = *reinterpret_cast<const v2u*>(lb + 16384 + ((m + it) & 3) * 512);
                    raw[2 * m] ^= r.x;
                    raw[2 * m + 1] ^= r.y;
                }
            }
            if (VAR >= 1 && m >= 2 && m < 6) {
                const int c = m - 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned x = raw[(2 * c + e) & 7];
                    A[c][e] = (int)((e & 1) ? ((x >> 4) & 0x0F0F0F0Fu) : (x & 0x0F0F0F0Fu));
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
                acc[m >> 1][c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[2 * c + (m & 1)], B[m], acc[m >> 1][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    int s = 0;
    for (int m = 0; m < 4; ++m)
        for (int c = 0; c < 2; ++c)
            for (int r = 0; r < 16; ++r) s += acc[m][c][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <typename F>
static double run(F launch, double ops_per_launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ops_per_launch * 5 / (ms * 1e-3) / 1e12;
}

template <int VAR>
static void go(int* out, const char* what) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kt<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const double ops = (double)blocks * 8 * iters * 32 * (2.0 * 16 * 16 * 64);
    printf("%-72s %7.1f TOPS\n", what, run([&] { kt<VAR><<<blocks, 512, 98304>>>(out, iters); }, ops));
}

template <int VAR>
static void go32(int* out, const char* what) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k32<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const double ops = (double)blocks * 8 * iters * 16 * (2.0 * 32 * 32 * 32);
    printf("%-72s %7.1f TOPS\n", what, run([&] { k32<VAR><<<blocks, 512, 98304>>>(out, iters); }, ops));
}

int main() {
    int* out;
    hipMalloc(&out, 2048 * 512 * 4);
    go32<0>(out, "32x32x32: 2 waves/SIMD, 16 MFMA per round (same tile), nothing else");
    go32<1>(out, "  + 32 VALU unpack ops per round");
    go32<2>(out, "  + 12 LDS operand reads per round");
    go32<3>(out, "  + s_barrier per round");
    go<0>(out, "2 waves/SIMD, 32 MFMA per round (4 A x 8 B operands), nothing else");
    go<1>(out, "  + 32 VALU unpack ops per round");
    go<2>(out, "  + 12 LDS operand reads per round");
    go<3>(out, "  + s_barrier per round");
    const int iters = 4000, blocks = 512;
    const double ops = (double)blocks * 8 * iters * 16 * (2.0 * 16 * 16 * 64);
    printf("%-72s %7.1f TOPS\n", "4 waves/SIMD, 16 MFMA per round, nothing else",
           run([&] { k4<<<blocks, 512>>>(out, iters); }, ops));
    return 0;
}
