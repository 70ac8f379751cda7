// lib.hip -- library identification and error reporting for libqserve_amd.so
#include "common.h"

static thread_local char g_err[512] = "";

void qs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int qs_version(void) { return 1; }
extern "C" const char* qs_arch(void) { return "gfx950"; }
extern "C" const char* qs_last_error(void) { return g_err; }

// Device self-test of the wave reductions (common.h): the DPP / permlane butterfly must round exactly like the
// __shfl_xor loop it replaces.  in: float [n] (n % 64 == 0), out: float [n / 64][4] = {sum, sum via shfl, max, max via shfl}.
namespace {
__global__ void wave_reduce_selftest_kernel(const float* __restrict__ in, float* __restrict__ out) {
    const float v = in[blockIdx.x * 64 + threadIdx.x];
    const float s0 = wave_sum(v), s1 = wave_sum_shfl(v), m0 = wave_max(v), m1 = wave_max_shfl(v);
    // every lane must hold the result: report lane (blockIdx.x % 64)'s copy
    if (threadIdx.x == (blockIdx.x & 63)) {
        out[blockIdx.x * 4 + 0] = s0;
        out[blockIdx.x * 4 + 1] = s1;
        out[blockIdx.x * 4 + 2] = m0;
        out[blockIdx.x * 4 + 3] = m1;
    }
}
}  // namespace
extern "C" int qs_debug_wave_reduce_selftest(const float* in, float* out, int n, qs_stream_t stream) {
    QS_REQUIRE(in && out && n > 0 && n % 64 == 0, "wave_reduce_selftest: bad arguments");
    hipLaunchKernelGGL(wave_reduce_selftest_kernel, dim3(n / 64), dim3(64), 0, (hipStream_t)stream, in, out);
    return qs_launch_status("wave_reduce_selftest");
}
