// Next-round question: the tiled W4A8 GEMM's marginal rate equals what
// microbench_mfma2.hip gives for its wave tile once the 12 LDS operand reads per 32 MFMAs are added.  Would a
// 128 x 128 wave tile - ONE wave per SIMD, 256 accumulator registers (the compiler places them in AGPRs), every operand
// reused 8 times, 16 LDS reads per 64 MFMAs - do better, although a single wave per SIMD has nobody to hide its stalls?
//   0  64 independent v_mfma_i32_16x16x64_i8 per round on 8 A x 8 B register operands, nothing else
//   1  + the unpack VALU of 8 weight operands per round (64 and / shift ops)
//   2  + 16 LDS operand reads per round (8 x b128 activations, 8 x b64 packed weights)
//   3  + one s_barrier per round
// Result (round 2, profiles/round2_mb_mfma3.txt): 2.09 / 2.12 / 2.65 / 2.68 POPS for variants 0-3 - one wave per SIMD issues the
// MFMA stream at half the pipe's rate; this tile is NOT the way forward.
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench_mfma3.hip -o scripts/mb_mfma3 ; run on the GPU box
// (written at the end of round 2; 256 VGPRs + 256 AGPRs, 12 bytes of scratch outside the loop).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

template <int VAR>
__global__ __launch_bounds__(256, 1) void kt(int* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    v4i A[8], B[8];
    for (int i = 0; i < 8; ++i) A[i] = (v4i){tid * 0x01010101 + i, 0x11121314 + i, 0x21222324 * (i + 1), 0x31323334 + tid};
    for (int i = 0; i < 8; ++i) B[i] = (v4i){0x0a0b0c0d + i, tid * 0x00010203 + i, 0x2a2b2c2d * (i + 1), 0x3a3b3c3d + tid};
    for (int i = tid; i < 65536 / 4; i += 256) reinterpret_cast<int*>(smem)[i] = i * 0x01030507;
    __syncthreads();
    v4i acc[8][8];
    for (int m = 0; m < 8; ++m)
        for (int c = 0; c < 8; ++c) acc[m][c] = (v4i){0, 0, 0, 0};
    unsigned raw[16];
    for (int i = 0; i < 16; ++i) raw[i] = tid * 0x9E3779B9u + i;
    const unsigned char* lb = smem + (tid & 63) * 16 + (tid >> 6) * 8192;
    for (int it = 0; it < iters; ++it) {
        if (VAR >= 3) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (VAR >= 2) {
                B[m] = *reinterpret_cast<const v4i*>(lb + ((m + it) & 7) * 1024);
                const v2u r = *reinterpret_cast<const v2u*>(lb + 32768 + ((m + it) & 7) * 512);
                raw[2 * m] ^= r.x;
                raw[2 * m + 1] ^= r.y;
            }
            if (VAR >= 1) {                          // unpack one weight operand per m: 4 x (and | shift + and)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned x = raw[(2 * m + e) & 15];
                    A[m][e] = (int)((e & 1) ? ((x >> 4) & 0x0F0F0F0Fu) : (x & 0x0F0F0F0Fu));
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[m][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[c], B[m], acc[m][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    int s = 0;
    for (int m = 0; m < 8; ++m)
        for (int c = 0; c < 8; ++c) s += acc[m][c][0] + acc[m][c][1] + acc[m][c][2] + acc[m][c][3];
    out[blockIdx.x * 256 + tid] = s;
}

template <int VAR>
static void go(int* out, const char* what) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kt<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kt<VAR><<<blocks, 256, 98304>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) kt<VAR><<<blocks, 256, 98304>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 4 * iters * 64 * (2.0 * 16 * 16 * 64) * 5;
    printf("%-76s %7.1f TOPS\n", what, ops / (ms * 1e-3) / 1e12);
}

int main() {
    int* out;
    hipMalloc(&out, 1024 * 256 * 4);
    go<0>(out, "1 wave/SIMD, 128x128 wave tile: 64 MFMA per round (8 A x 8 B operands), nothing else");
    go<1>(out, "  + 64 VALU unpack ops per round");
    go<2>(out, "  + 16 LDS operand reads per round");
    go<3>(out, "  + s_barrier per round");
    return 0;
}
