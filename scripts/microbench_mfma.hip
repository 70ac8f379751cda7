// Measured INT8 MFMA peak on this GPU ("compute peak from the measured engine clock during the run"):
// 256 CUs x 8 waves, each wave issues back-to-back independent v_mfma_i32_16x16x64_i8 (and 32x32x32) on register operands.
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench_mfma.hip -o scripts/mb_mfma ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int RANDOM>
__global__ __launch_bounds__(512) void k16(int* out, int iters) {
    v4i a = {(int)threadIdx.x * 0x01010101 * RANDOM + 0x01020304, 0x11121314 * RANDOM + 1, 0x21222324 * RANDOM + 2, 0x31323334 * RANDOM + 3};
    v4i b = {0x0a0b0c0d * RANDOM + 4, (int)threadIdx.x * 0x00010203 * RANDOM + 5, 0x2a2b2c2d * RANDOM + 6, 0x3a3b3c3d * RANDOM + 7};
    v4i c[8];
    for (int i = 0; i < 8; ++i) c[i] = (v4i){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int RANDOM>
__global__ __launch_bounds__(512) void k32(int* out, int iters) {
    v4i a = {(int)threadIdx.x * 0x01010101 * RANDOM + 0x01020304, 0x11121314 * RANDOM + 1, 0x21222324 * RANDOM + 2, 0x31323334 * RANDOM + 3};
    v4i b = {0x0a0b0c0d * RANDOM + 4, (int)threadIdx.x * 0x00010203 * RANDOM + 5, 0x2a2b2c2d * RANDOM + 6, 0x3a3b3c3d * RANDOM + 7};
    v16i c[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) c[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += c[i][j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
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

int main() {
    int* out;
    hipMalloc(&out, 2048 * 512 * 4);
    const int iters = 4000, blocks = 512;   // 2 blocks of 8 waves per CU
    const double ops16 = (double)blocks * 8 * iters * 8 * (2.0 * 16 * 16 * 64);
    const double ops32 = (double)blocks * 8 * iters * 4 * (2.0 * 32 * 32 * 32);
    printf("v_mfma_i32_16x16x64_i8  constant operands: %7.1f TOPS   varied operands: %7.1f TOPS\n",
           run([&] { k16<0><<<blocks, 512>>>(out, iters); }, ops16), run([&] { k16<1><<<blocks, 512>>>(out, iters); }, ops16));
    printf("v_mfma_i32_32x32x32_i8  constant operands: %7.1f TOPS   varied operands: %7.1f TOPS\n",
           run([&] { k32<0><<<blocks, 512>>>(out, iters); }, ops32), run([&] { k32<1><<<blocks, 512>>>(out, iters); }, ops32));
    return 0;
}
