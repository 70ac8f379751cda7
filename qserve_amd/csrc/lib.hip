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

// ---- bounded in-launch waits: status, reset, fault injection (common.h) ---------------------------------------------------------
qs_flag g_inject_fault = 0;
extern "C" int qs_device_status(int* error_bits) {
    QS_REQUIRE(error_bits, "qs_device_status: null output");
    *error_bits = 0;
    {   // a null-stream copy does not order behind non-blocking streams (torch's pool / side streams are): wait for the device
        const hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            qs_set_error("qs_device_status: %s", hipGetErrorString(e));
            return (int)e;
        }
    }
    for (int i = 0; i < 2 * QS_MAX_STREAM_SLOTS; ++i) {
        unsigned* w = i & 1 ? qs_attn_error_word(i >> 1) : qs_gemm_error_word(i >> 1);
        if (!w) continue;
        unsigned v = 0;
        const hipError_t e = hipMemcpy(&v, w, sizeof(v), hipMemcpyDeviceToHost);   // (blocking: orders behind the launches so far)
        if (e != hipSuccess) {
            qs_set_error("qs_device_status: %s", hipGetErrorString(e));
            return (int)e;
        }
        *error_bits |= (int)v;
    }
    if (*error_bits)
        qs_set_error("a bounded in-launch wait gave up (bits %d: 1 = K-slice seam of a W4A8 GEMM, 2 = attention + quant hand-over): "
                     "results of that launch are invalid; call qs_device_reset() before reusing the library", *error_bits);
    return QS_OK;
}
extern "C" int qs_device_reset(void) {
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        qs_set_error("qs_device_reset: %s", hipGetErrorString(e));
        return (int)e;
    }
    int rc = qs_gemm_reset_handoff();
    if (rc == QS_OK) rc = qs_attn_reset_handoff();
    return rc;
}
// ---- scratch slots (common.h) ----------------------------------------------------------------------------------------------------
#include <mutex>
namespace {
std::mutex g_slot_mutex;
hipStream_t g_slot_stream[QS_MAX_DEVICES][QS_MAX_STREAM_SLOTS];   // [d][0] unused (the shared slot)
bool g_slot_bound[QS_MAX_DEVICES][QS_MAX_STREAM_SLOTS];
int g_slots_bound[QS_MAX_DEVICES];                                // fast path: nothing bound -> slot 0 without taking the lock
}  // namespace
int qs_scratch_slot(hipStream_t stream) {
    const int d = qs_device_slot();
    if (__atomic_load_n(&g_slots_bound[d], __ATOMIC_ACQUIRE) == 0) return 0;
    std::lock_guard<std::mutex> lock(g_slot_mutex);
    for (int i = 1; i < QS_MAX_STREAM_SLOTS; ++i)
        if (g_slot_bound[d][i] && g_slot_stream[d][i] == stream) return i;
    return 0;
}
extern "C" int qs_stream_scratch_bind(qs_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        qs_set_error("qs_stream_scratch_bind: the stream is capturing (bind before the capture: the slot's memory is allocated here)");
        return QS_EINVAL;
    }
    const int d = qs_device_slot();
    int taken = 0;
    {
        std::lock_guard<std::mutex> lock(g_slot_mutex);
        int free_slot = 0;
        for (int i = QS_MAX_STREAM_SLOTS - 1; i >= 1; --i) {
            if (g_slot_bound[d][i] && g_slot_stream[d][i] == st) return QS_OK;       // idempotent
            if (!g_slot_bound[d][i]) free_slot = i;
        }
        if (!free_slot) {
            qs_set_error("qs_stream_scratch_bind: all %d per-stream scratch slots of this device are bound", QS_MAX_STREAM_SLOTS - 1);
            return QS_ENOSUP;
        }
        g_slot_stream[d][free_slot] = st;
        g_slot_bound[d][free_slot] = true;
        __atomic_add_fetch(&g_slots_bound[d], 1, __ATOMIC_RELEASE);
        taken = free_slot;
    }
    const bool ok_g = qs_gemm_scratch_prealloc(st), ok_a = qs_attn_scratch_prealloc(st), ok_m = qs_argmax_scratch_prealloc(st);
    if (!(ok_g && ok_a && ok_m)) {   // (what was allocated stays with the slot; an area whose allocation failed is not retried - `tried`)
        std::lock_guard<std::mutex> lock(g_slot_mutex);
        g_slot_bound[d][taken] = false;
        __atomic_sub_fetch(&g_slots_bound[d], 1, __ATOMIC_RELEASE);
        qs_set_error("qs_stream_scratch_bind: allocating the slot's scratch areas failed (GEMM workspace %s, attention workspace %s, "
                     "argmax workspace %s); the stream stays on the shared set", ok_g ? "ok" : "FAILED", ok_a ? "ok" : "FAILED",
                     ok_m ? "ok" : "FAILED");
        return QS_ENOSUP;
    }
    return QS_OK;
}
extern "C" int qs_stream_scratch_unbind(qs_stream_t stream) {
    const int d = qs_device_slot();
    std::lock_guard<std::mutex> lock(g_slot_mutex);
    for (int i = 1; i < QS_MAX_STREAM_SLOTS; ++i)
        if (g_slot_bound[d][i] && g_slot_stream[d][i] == (hipStream_t)stream) {
            g_slot_bound[d][i] = false;                 // the slot's memory stays (graphs captured on the stream keep working);
            __atomic_sub_fetch(&g_slots_bound[d], 1, __ATOMIC_RELEASE);   // the next bind of this device reuses it
            return QS_OK;
        }
    return QS_OK;
}
extern "C" int qs_debug_inject_fault(int what) {
    QS_REQUIRE(what >= 0 && what <= 3, "qs_debug_inject_fault: what=%d not in 0..3", what);
    g_inject_fault = what;
    return QS_OK;
}

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
