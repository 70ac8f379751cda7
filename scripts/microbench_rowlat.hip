// microbench_rowlat.hip -- what does a one-workgroup-per-token row kernel cost at decode batch sizes, piece by piece?
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench_rowlat.hip -o /tmp/mb_rowlat
// 64 workgroups x 256 threads, row of 4096 fp16 (2 x 16-byte chunks per thread and input), launched back to back from
// a hipGraph (200 dependent launches):
//   NIN  inputs read per element (1 .. 3),  NRED block reductions (wave shuffle + LDS + barrier),  store 16 B per thread
// "cold": inputs rotate over 64 buffers so that they come from HBM / the Infinity Cache like a previous kernel's output.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float block_sum(float v, float* sm) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
    return t;
}

template <int NIN, int NRED>
__global__ __launch_bounds__(256) void rowk(const _Float16* __restrict__ a, const _Float16* __restrict__ b,
                                            const _Float16* __restrict__ c, _Float16* __restrict__ out, int hidden) {
    __shared__ float sm[4][4];
    const size_t base = (size_t)blockIdx.x * hidden;
    h8 x[2], y[2], z[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const int i = (ch * 256 + threadIdx.x) * 8;
        x[ch] = *reinterpret_cast<const h8*>(a + base + i);
        if (NIN > 1) y[ch] = *reinterpret_cast<const h8*>(b + base + i);
        if (NIN > 2) z[ch] = *reinterpret_cast<const h8*>(c + i);
    }
    float s = 0.f;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)x[ch][j] + (NIN > 1 ? (float)y[ch][j] : 0.f) + (NIN > 2 ? (float)z[ch][j] : 0.f);
    float r = s;
    if (NRED > 0) r = block_sum(r, sm[0]);
    if (NRED > 1) r = block_sum(r * 0.5f + s, sm[1]);
    if (NRED > 2) r = block_sum(r * 0.25f + s, sm[2]);
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const int i = (ch * 256 + threadIdx.x) * 8;
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (_Float16)((float)x[ch][j] * r);
        *reinterpret_cast<h8*>(out + base + i) = o;
    }
}
__global__ void emptyk() {}

template <int NIN, int NRED>
void run(const char* name, _Float16* bufs, int nbuf, bool cold) {
    const int hidden = 4096, rows = 64, N = 200;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) {
        const size_t per = (size_t)rows * hidden;
        _Float16* a = bufs + (cold ? (size_t)(i % nbuf) * 4 * per : 0);
        hipLaunchKernelGGL((rowk<NIN, NRED>), dim3(rows), dim3(256), 0, st, a, a + per, a + 2 * per, a + 3 * per, hidden);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %s: %6.2f us per launch\n", name, cold ? "cold" : "warm", ms * 1e3 / (5 * N));
}

int main() {
    const int nbuf = 64;
    _Float16* bufs;
    CK(hipMalloc(&bufs, (size_t)nbuf * 4 * 64 * 4096 * 2));
    CK(hipMemset(bufs, 0, (size_t)nbuf * 4 * 64 * 4096 * 2));
    {   // empty kernel: the launch-to-launch floor of a graph
        hipStream_t st;
        CK(hipStreamCreate(&st));
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(emptyk, dim3(64), dim3(256), 0, st);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s     : %6.2f us per launch\n", "empty kernel (64 x 256)", ms * 1e3 / 1000);
    }
    for (int cold = 0; cold < 2; ++cold) {
        run<1, 0>("1 input, no reduction", bufs, nbuf, cold);
        run<1, 1>("1 input, 1 reduction", bufs, nbuf, cold);
        run<1, 2>("1 input, 2 reductions", bufs, nbuf, cold);
        run<1, 3>("1 input, 3 reductions", bufs, nbuf, cold);
        run<3, 0>("3 inputs, no reduction", bufs, nbuf, cold);
        run<3, 3>("3 inputs, 3 reductions", bufs, nbuf, cold);
    }
    return 0;
}
