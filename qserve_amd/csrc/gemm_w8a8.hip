// gemm_w8a8.hip -- W8A8 int8 GEMM (module `qserve_backend.qgemm_w8a8` must be importable).
// Behaviour of kernels/csrc/qgemm/w8a8/w8a8_gemm_cuda.cu (epilogue :521-524: out = float(acc) * (wscale*ascale)).
// Not a tuned kernel: one wave64 per 16 channels x 64 tokens, v_mfma_i32_16x16x64_i8, operands straight from
// global memory (16 contiguous k per lane for both operands).
#include "common.h"

namespace {

__device__ __forceinline__ float epi(int acc, float ws, float sa) {
#pragma clang fp contract(off)
    const float sc = ws * sa;
    return (float)acc * sc;
}

__global__ __launch_bounds__(256) void w8a8_kernel(const int8_t* __restrict__ A, const int8_t* __restrict__ W,
                                                   const __half* __restrict__ wscales,
                                                   const __half* __restrict__ ascales, _Float16* __restrict__ out,
                                                   int M, int N, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int n0 = (blockIdx.x * 4 + wave) * 16;
    const int m0 = blockIdx.y * 64;
    if (n0 >= N) return;
    const int8_t* wrow = W + (size_t)(n0 + li) * K + 16 * g;
    const int8_t* arow[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        int r = m0 + 16 * mt + li;
        r = r < M ? r : M - 1;
        arow[mt] = A + (size_t)r * K + 16 * g;
    }
    v4i acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = (v4i){0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 64) {
        const v4i a = *reinterpret_cast<const v4i*>(wrow + k0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const v4i b = *reinterpret_cast<const v4i*>(arow[mt] + k0);
            acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[mt], 0, 0, 0);
        }
    }
    // D[i = 4g + r][j = li]: channel n0 + 4g + r, token m0 + 16mt + li
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + 16 * mt + li;
        if (m >= M) continue;
        const float sa = __half2float(ascales[m]);
        h4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (_Float16)epi(acc[mt][r], __half2float(wscales[n0 + 4 * g + r]), sa);
        *reinterpret_cast<h4*>(out + (size_t)m * N + n0 + 4 * g) = o;
    }
}

}  // namespace

extern "C" int qs_w8a8_gemm(const int8_t* in_feats, const int8_t* kernel, const void* wscales, const void* ascales,
                            void* out_feats, int M, int N, int K, qs_stream_t stream) {
    QS_REQUIRE(in_feats && kernel && wscales && ascales && out_feats, "w8a8 gemm: null pointer");
    QS_REQUIRE(N > 0 && N % 16 == 0 && K > 0 && K % 64 == 0, "w8a8 gemm: need N %% 16 == 0 and K %% 64 == 0 (N=%d K=%d)",
               N, K);
    if (M <= 0) return QS_OK;
    dim3 grid((N / 16 + 3) / 4, (M + 63) / 64);
    hipLaunchKernelGGL(w8a8_kernel, grid, dim3(256), 0, (hipStream_t)stream, in_feats, kernel,
                       (const __half*)wscales, (const __half*)ascales, (_Float16*)out_feats, M, N, K);
    return qs_launch_status("w8a8 gemm");
}
