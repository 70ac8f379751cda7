// Round 3: costing the two remaining ideas for the compute-bound W4A8 tile (HISTORY.md, "what comes next" 2) before building
// either.  Same harness as microbench_mfma2.hip (whose variants 0-3 are repeated here as the in-run reference).
//   B  16 waves per workgroup (4 per SIMD), 64 tokens x 64 channels per wave: 16 MFMAs, 4 b128 + 4 b64 LDS reads and the
//      unpack of 4 weight tiles (32 VALU) per 64-k round, 64 accumulators; with / without the per-round barrier
//   D  the shipped wave tile (8 waves, 128 x 64) with the weight operand NOT read from LDS (registers fed some other way):
//      8 b128 reads per round instead of 8 + 4
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench_mfma4.hip -o scripts/mb_mfma4 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

// VAR: 0 MFMA only, 1 + unpack, 2 + LDS reads (WLDS: weight reads too), 3 + barrier
template <int VAR, bool WLDS>
__global__ __launch_bounds__(512, 1) void k8(int* out, int iters) {
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
                    if (WLDS) {
                        const v2u r = *reinterpret_cast<const v2u*>(lb + 16384 + ((m + it) & 3) * 512);
                        raw[2 * m] ^= r.x;
                        raw[2 * m + 1] ^= r.y;
                    } else {
                        raw[2 * m] += 0x01010101u;
                        raw[2 * m + 1] ^= raw[2 * m];
                    }
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
            for (int c = 0; c < 4; ++c) acc[m][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[c], B[m], acc[m][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    int s = 0;
    for (int m = 0; m < 8; ++m)
        for (int c = 0; c < 4; ++c) s += acc[m][c][0] + acc[m][c][1] + acc[m][c][2] + acc[m][c][3];
    out[blockIdx.x * 512 + tid] = s;
}

// 16 waves, 64 x 64 per wave
template <int VAR>
__global__ __launch_bounds__(1024, 1) void k16(int* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    v4i A[4], B[4];
    for (int i = 0; i < 4; ++i) A[i] = (v4i){tid * 0x01010101 + i, 0x11121314 + i, 0x21222324 * (i + 1), 0x31323334 + tid};
    for (int i = 0; i < 4; ++i) B[i] = (v4i){0x0a0b0c0d + i, tid * 0x00010203 + i, 0x2a2b2c2d * (i + 1), 0x3a3b3c3d + tid};
    for (int i = tid; i < 32768 / 4; i += 1024) reinterpret_cast<int*>(smem)[i] = i * 0x01030507;
    __syncthreads();
    v4i acc[4][4];
    for (int m = 0; m < 4; ++m)
        for (int c = 0; c < 4; ++c) acc[m][c] = (v4i){0, 0, 0, 0};
    unsigned raw[8];
    for (int i = 0; i < 8; ++i) raw[i] = tid * 0x9E3779B9u + i;
    // 4 token groups x 4 channel groups of waves: activation image of token group (wave >> 2), weights of (wave & 3)
    const int wave = tid >> 6;
    const unsigned char* lb = smem + (tid & 63) * 16 + (wave >> 2) * 4096;
    const unsigned char* lw = smem + 16384 + (tid & 63) * 8 + (wave & 3) * 2048;
    for (int it = 0; it < iters; ++it) {
        if (VAR >= 3) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (VAR >= 2) {
                B[m] = *reinterpret_cast<const v4i*>(lb + ((m + it) & 3) * 1024);
                const v2u r = *reinterpret_cast<const v2u*>(lw + ((m + it) & 3) * 512);
                raw[2 * m] ^= r.x;
                raw[2 * m + 1] ^= r.y;
            }
            if (VAR >= 1) {                          // unpack one A operand per m: 4 weight tiles per round
                const int c = (m + 2) & 3;
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
    for (int m = 0; m < 4; ++m)
        for (int c = 0; c < 4; ++c) s += acc[m][c][0] + acc[m][c][1] + acc[m][c][2] + acc[m][c][3];
    out[blockIdx.x * 1024 + tid] = s;
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

template <int VAR, bool WLDS>
static void go8(int* out, const char* what) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k8<VAR, WLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const double ops = (double)blocks * 8 * iters * 32 * (2.0 * 16 * 16 * 64);
    printf("%-86s %7.1f TOPS\n", what, run([&] { k8<VAR, WLDS><<<blocks, 512, 98304>>>(out, iters); }, ops));
}
template <int VAR>
static void go16(int* out, const char* what) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k16<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const double ops = (double)blocks * 16 * iters * 16 * (2.0 * 16 * 16 * 64);
    printf("%-86s %7.1f TOPS\n", what, run([&] { k16<VAR><<<blocks, 1024, 98304>>>(out, iters); }, ops));
}


// P: the shipped tile with the operand reads software-pipelined a whole round ahead (two operand sets, no wait in front of
// an MFMA that the previous round did not already cover); Q: the same 12 reads per round issued but never consumed by the
// MFMAs (operands stay constant): separates "a read costs issue / return-path time" from "the MFMAs wait for their read".
template <int MODE>   // 0 = P, 1 = Q
__global__ __launch_bounds__(512, 1) void k8p(int* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    v4i A[4], B0[8], B1[8];
    for (int i = 0; i < 4; ++i) A[i] = (v4i){tid * 0x01010101 + i, 0x11121314 + i, 0x21222324 * (i + 1), 0x31323334 + tid};
    for (int i = 0; i < 8; ++i) B0[i] = B1[i] = (v4i){0x0a0b0c0d + i, tid * 0x00010203 + i, 0x2a2b2c2d * (i + 1), 0x3a3b3c3d + tid};
    for (int i = tid; i < 32768 / 4; i += 512) reinterpret_cast<int*>(smem)[i] = i * 0x01030507;
    __syncthreads();
    v4i acc[8][4];
    for (int m = 0; m < 8; ++m)
        for (int c = 0; c < 4; ++c) acc[m][c] = (v4i){0, 0, 0, 0};
    unsigned raw[8];
    for (int i = 0; i < 8; ++i) raw[i] = tid * 0x9E3779B9u + i;
    v4i sink = {0, 0, 0, 0};
    const unsigned char* lb = smem + (tid & 63) * 16 + (tid >> 6) * 2048;
    auto round = [&](v4i (&Bc)[8], v4i (&Bn)[8], int it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const v4i r = *reinterpret_cast<const v4i*>(lb + ((m + it) & 7) * 1024);
            if (MODE == 0) Bn[m] = r;
            else sink ^= r;
            if (m < 4) {
                const v2u w = *reinterpret_cast<const v2u*>(lb + 16384 + ((m + it) & 3) * 512);
                if (MODE == 0) {
                    raw[2 * m] ^= w.x;
                    raw[2 * m + 1] ^= w.y;
                } else {
                    sink[0] ^= (int)w.x;
                    sink[1] ^= (int)w.y;
                }
            }
            if (m >= 2 && m < 6) {
                const int c = m - 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned x = raw[(2 * c + e) & 7];
                    A[c][e] = (int)((e & 1) ? ((x >> 4) & 0x0F0F0F0Fu) : (x & 0x0F0F0F0Fu));
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[m][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[c], Bc[m], acc[m][c], 0, 0, 0);
        }
    };
    for (int it = 0; it < iters; it += 2) {
        round(B0, B1, it);
        round(B1, B0, it + 1);
    }
    int s = sink[0] + sink[1] + sink[2] + sink[3];
    for (int m = 0; m < 8; ++m)
        for (int c = 0; c < 4; ++c) s += acc[m][c][0] + acc[m][c][1] + acc[m][c][2] + acc[m][c][3];
    out[blockIdx.x * 512 + tid] = s;
}
template <int MODE>
static void go8p(int* out, const char* what) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k8p<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const double ops = (double)blocks * 8 * iters * 32 * (2.0 * 16 * 16 * 64);
    printf("%-86s %7.1f TOPS\n", what, run([&] { k8p<MODE><<<blocks, 512, 98304>>>(out, iters); }, ops));
}

int main() {
    int* out;
    hipMalloc(&out, 2048 * 1024 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        go8<0, true>(out, "8 waves (2/SIMD) 128x64: MFMA only");
        go8<1, true>(out, "  + unpack");
        go8<2, true>(out, "  + 8 b128 + 4 b64 LDS reads (the shipped loop)");
        go8<3, true>(out, "  + barrier");
        go8<2, false>(out, "8 waves 128x64: unpack + 8 b128 reads only (weights not through LDS)");
        go8<3, false>(out, "  + barrier");
        go8p<0>(out, "8 waves 128x64: unpack + 12 reads, operands read one round ahead (compiler-scheduled)");
        go8p<1>(out, "8 waves 128x64: unpack + 12 reads issued but not consumed by the MFMAs");
        go16<0>(out, "16 waves (4/SIMD) 64x64: MFMA only");
        go16<1>(out, "  + unpack (32 VALU per 16 MFMA)");
        go16<2>(out, "  + 4 b128 + 4 b64 LDS reads");
        go16<3>(out, "  + barrier");
    }
    return 0;
}
