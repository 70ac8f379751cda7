// common.h -- shared helpers for the gfx950 kernels of libqserve_amd.so
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/qserve_amd.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

void qs_set_error(const char* fmt, ...);

#define QS_REQUIRE(cond, ...)             \
    do {                                  \
        if (!(cond)) {                    \
            qs_set_error(__VA_ARGS__);    \
            return QS_EINVAL;             \
        }                                 \
    } while (0)

static inline int qs_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        qs_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return QS_OK;
}

// per-device state (function attributes, scratch areas) is indexed by the CURRENT device of the calling thread
constexpr int QS_MAX_DEVICES = 16;
static inline int qs_device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= QS_MAX_DEVICES) d = 0;
    return d;
}

// silu(x) rounded to fp16 with the hardware exp2 / rcp forms (what the reference's --use_fast_math build computes; see
// fused_small.hip).  Shared by the row kernels and by the GEMM epilogue that applies silu * mul to gate_up.
__device__ __forceinline__ _Float16 qs_silu_h(float xf) {
    const float e = __builtin_amdgcn_exp2f(xf * -1.4426950408889634f);
    return (_Float16)(xf * __builtin_amdgcn_rcpf(1.0f + e));
}

// ---- W4A8 arithmetic shared by every GEMM kernel (gemm_w4a8*.hip) and by the row kernel that finishes K-slice planes -------
// per-byte wrapping add (__vadd4 semantics, w4a8_per_group/gemm_cuda.cu:302)
__device__ __forceinline__ u32 vadd4(u32 a, u32 b) {
    return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
}
// four packed nibbles -> four operand bytes; MODE 1 (per-group): level-2 dequant = 32-bit multiply (byte products carry into
// their neighbours exactly as in the reference, gemm_cuda.cu:300-301) + per-byte wrapping add of the zero byte
template <int MODE>
__device__ __forceinline__ u32 unpack_lo(u32 raw, u32 s, u32 zb) {
    u32 u = raw & 0x0F0F0F0Fu;
    if (MODE == 1) u = vadd4(u * s, zb);
    return u;
}
template <int MODE>
__device__ __forceinline__ u32 unpack_hi(u32 raw, u32 s, u32 zb) {
    u32 u = (raw >> 4) & 0x0F0F0F0Fu;
    if (MODE == 1) u = vadd4(u * s, zb);
    return u;
}
// fp32 epilogues with the reference's evaluation order (oracle/w4a8.py epilogue_*).
// per-channel, w4a8_per_chn/gemm_cuda.cu:586-587: (acc * wscale) * ascale - w_sz * a_ssum.  `fma` selects the CONVENTION
// (qs_set_gemm_epilogue, include/qserve_amd.h): 0 = every operation rounded separately (the library's default; what the
// source statement says without contraction), 1 = fmaf(acc * wscale, ascale, -(w_sz * a_ssum)) - what nvcc's default
// --fmad=true most plausibly makes of that line (oracle epilogue_per_chn(fma=True)).  Wave-uniform.
__device__ __forceinline__ float epi_per_chn(int acc, float ws, float sa, float wz, float ss, int fma = 0) {
#pragma clang fp contract(off)
    float t = (float)acc * ws;
    const float u = wz * ss;
    if (fma) return __builtin_fmaf(t, sa, -u);
    t = t * sa;
    return t - u;
}
// per-group, w4a8_per_group/gemm_cuda.cu:620-621: acc * (wscale * ascale)
__device__ __forceinline__ float epi_per_group(int acc, float ws, float sa) {
#pragma clang fp contract(off)
    const float sc = ws * sa;
    return (float)acc * sc;
}
// Process-wide test / measurement hooks (kernel selection, A/B switches, fault injection) are relaxed atomics: flipping one
// while another host thread launches is not a data race - that launch sees the old or the new value (include/qserve_amd.h says
// which entry sets which).  They are still process-wide: not a per-stream or per-thread configuration.
typedef std::atomic<int> qs_flag;
extern qs_flag g_epi_fma;   // gemm_w4a8.hip: qs_set_gemm_epilogue
extern unsigned long long* g_gemm_clk;   // gemm_w4a8.hip: qs_debug_gemm_clock_probe (device buffer, 2 words per workgroup) or null
extern int g_gemm_clk_cap;               // workgroups the buffer holds

// compute units of the current device (cached per device; 256 on MI355X)
static inline int qs_num_cus() {
    static int cus[QS_MAX_DEVICES] = {};
    int& n = cus[qs_device_slot()];
    if (n == 0) {
        int d = 0, v = 0;
        (void)hipGetDevice(&d);
        n = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && v > 0 ? v : 256;
    }
    return n;
}

// wave64 reductions -------------------------------------------------------------------------------
// The classic xor butterfly (offsets 32, 16, 8, 4, 2, 1; every lane ends with the result) with the SAME pairing and the
// same order of operations as a __shfl_xor loop - so fp32 sums round identically, bit for bit - but without its six
// dependent ds_bpermute round trips through the LDS crossbar (~100+ cycles each; these reductions sit on the latency
// chain of every row kernel):
//   xor 32 / 16 ... v_permlane32_swap / v_permlane16_swap of the value with itself: the two results are (lower | lower)
//                   and (upper | upper) halves (resp. even / odd rows), whose op() is "own op partner" in every lane;
//   xor 8 ......... DPP row_ror:8 (a rotation by 8 inside a 16-lane row IS xor 8);
//   xor 4 ......... DPP row_ror:4: after the xor-8 step the row is 8-periodic, so lane (i +- 4) mod 16 holds exactly the
//                   value of lane i ^ 4;
//   xor 2 / 1 ..... DPP quad_perm [2,3,0,1] / [1,0,3,2].
// All 64 lanes must be active (every call site is wave-uniform).
template <class Op>
__device__ __forceinline__ float wave_butterfly(float v, Op op) {
    {
        const int x = __builtin_bit_cast(int, v);
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        v = op(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
    }
    {
        const int x = __builtin_bit_cast(int, v);
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        v = op(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
    }
#define QS_DPP(ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    v = op(v, QS_DPP(0x128));   // row_ror:8
    v = op(v, QS_DPP(0x124));   // row_ror:4
    v = op(v, QS_DPP(0x4E));    // quad_perm [2,3,0,1]
    v = op(v, QS_DPP(0xB1));    // quad_perm [1,0,3,2]
#undef QS_DPP
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    return wave_butterfly(v, [](float a, float b) { return fmaxf(a, b); });
}
__device__ __forceinline__ float wave_min(float v) {
    return wave_butterfly(v, [](float a, float b) { return fminf(a, b); });
}
__device__ __forceinline__ float wave_sum(float v) {
    return wave_butterfly(v, [](float a, float b) { return a + b; });
}
// the same three through __shfl_xor (ds_bpermute): reference form, kept for the device self-test of the equivalence
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_shfl(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// wave64 all-reduce on the DPP crossbar (no LDS traffic, ~10 short instructions instead of six dependent ds_bpermute
// round trips): xor-1 / xor-2 inside the quads, row_half_mirror and row_mirror inside each 16-lane row, then the four row
// results through v_readlane.  Every lane receives the result.
template <class Op>
__device__ __forceinline__ float wave_allreduce_dpp(float v, Op op) {
#define QS_DPP(ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    v = op(v, QS_DPP(0xB1));    // quad_perm [1,0,3,2]
    v = op(v, QS_DPP(0x4E));    // quad_perm [2,3,0,1]
    v = op(v, QS_DPP(0x141));   // row_half_mirror
    v = op(v, QS_DPP(0x140));   // row_mirror
#undef QS_DPP
    const int x = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
    return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    return wave_allreduce_dpp(v, [](float a, float b) { return fmaxf(a, b); });
}
__device__ __forceinline__ float wave_min_dpp(float v) {
    return wave_allreduce_dpp(v, [](float a, float b) { return fminf(a, b); });
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    return wave_allreduce_dpp(v, [](float a, float b) { return a + b; });
}

// cvt.rni.sat.{s8,u8}.f32 equivalents: round-to-nearest-even then saturate (NaN -> 0)
// Plan-only mode of the GEMM dispatcher (qs_w4a8_gemm_plan): the launchers record which kernel family / geometry they
// were asked for and return without touching the device - the selection heuristics become testable on a CPU-only box.
// K-sliced ring GEMM: the word a slab holds where no partial sum has been delivered (gemm_w4a8_ring.hip, the seam); the workspace
// is filled with it byte-wise (0x80), and no partial sum of a slice of at most 32 768 k can reach it (128 * 255 * 32768 < 2^30)
constexpr int QS_SLAB_SENTINEL = (int)0x80808080u;

// Bounded in-launch waits (round 5).  The two cross-workgroup hand-offs that poll inside a launch - the K-slice seam of the ring
// GEMM and the finisher of the attention + quant fusion - give up after QS_SPIN_CAP polls (each poll is a memory round trip of
// ~1 us plus a sleep: seconds, i.e. never in a healthy launch), OR a bit into the error word of their scratch slot and finish with whatever
// they have; qs_device_status() reads the words of every slot, qs_device_reset() clears them and puts the hand-off areas back into their
// initial state.  qs_debug_inject_fault() arms a one-shot fault (a producer that never delivers) so that the path is testable.
constexpr int QS_SPIN_CAP = 1 << 20;
constexpr unsigned QS_ERR_GEMM_SEAM = 1u;     // K-slice seam: a partial tile never arrived
constexpr unsigned QS_ERR_ATTN_HANDOVER = 2u; // attention + quant: a KV head's result row never arrived
extern qs_flag g_inject_fault;                    // lib.hip: bit 0 next K-sliced ring GEMM launch, bit 1 next attention + quant launch
// Library scratch (split-K slabs, K-slice slabs, split-KV partials, hand-over rows, argmax keys) exists once per device (slot 0,
// shared by every stream: launches that use it must not overlap) plus once per stream that asked for its own with
// qs_stream_scratch_bind() (slots 1 .. QS_MAX_STREAM_SLOTS - 1; lib.hip).  qs_scratch_slot: the slot of a stream on the
// calling thread's current device.
constexpr int QS_MAX_STREAM_SLOTS = 8;
int qs_scratch_slot(hipStream_t stream);
bool qs_gemm_scratch_prealloc(hipStream_t stream);      // gemm_w4a8.hip     } allocate the slot's areas now (qs_stream_scratch_bind:
bool qs_attn_scratch_prealloc(hipStream_t stream);      // attention_mfma.hip } a first use inside a stream capture could not)
bool qs_argmax_scratch_prealloc(hipStream_t stream);    // fused_small.hip   }
unsigned* qs_gemm_error_word(int slot);       // gemm_w4a8.hip: nullptr until that slot's split-K workspace exists
unsigned* qs_attn_error_word(int slot);       // attention_mfma.hip: nullptr until that slot's hand-over workspace exists
int qs_gemm_reset_handoff();                  // gemm_w4a8.hip: sentinel-fill the K-slice slabs, clear the error word
int qs_attn_reset_handoff();                  // attention_mfma.hip: zero generation words / exchange rows, clear the error word

struct QsGemmPlan {
    int active;   // 1 while qs_w4a8_gemm_plan runs the dispatcher
    int family;   // 1 split-K, 2 LDS-pair, 3 ring, 4 tiled
    int p[4];     // ring: m_tiles, units, token blocks, K slices; tiled: m-tiles per wave (8 = 256-token tile, 4 = 128);
                  // split-K: m_tiles, waves, cross-block slices, xcd mapping
};
extern thread_local QsGemmPlan g_qs_plan;
// the same for the decode attention dispatcher (qs_attention_plan): family 1 = matrix-core KV4, 2 = matrix-core KV8,
// 3 = VALU kernel; nsplit = KV splits (workgroups per sequence and KV head), waves = waves per workgroup
struct QsAttnPlan {
    int active, family, nsplit, waves;
};
extern thread_local QsAttnPlan g_qs_attn_plan;
// request of qs_single_query_attention_quant to the attention launchers: fuse invoke_quant(_fuse_sum) of the output
// into the kernel when the chosen kernel can (sets `done`); otherwise the entry point runs the row kernel itself
struct QsAttnQuant {
    int8_t* qout;
    void* qscale;
    void* qsum;     // may be null (invoke_quant without the row sum)
    int done;
};
extern thread_local QsAttnQuant g_qs_attn_quant;

// butterfly exchange with an explicitly supplied lane id: __shfl_xor derives its own (loop-invariant) lane id, which the
// register allocator then keeps alive - or spills - across a long loop
__device__ __forceinline__ float xor_lane(float x, unsigned lid, int mask) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)((lid ^ (unsigned)mask) << 2), __builtin_bit_cast(int, x)));
}
__device__ __forceinline__ unsigned fresh_lane_id() {   // opaque to CSE: not shared with earlier derivations
    unsigned lid;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lid));
    return lid;
}

__device__ __forceinline__ int rni_sat_s8(float x) {
    float r = rintf(x);
    r = fminf(fmaxf(r, -128.f), 127.f);   // fmaxf/fminf drop NaN -> -128 ; handle below
    return (x != x) ? 0 : (int)r;
}
__device__ __forceinline__ unsigned rni_sat_u8(float x) {
    float r = rintf(x);
    r = fminf(fmaxf(r, 0.f), 255.f);
    return (x != x) ? 0u : (unsigned)r;
}
// 8 values x mul -> 8 int8 (cvt.rni.sat.s8.f32), one 8-byte store: the quantising store of the per-token row kernels
// (fused_kernels.cu:78-82), shared by fused_small.hip and the attention + quant fusion
__device__ __forceinline__ void qs_store_q8(int8_t* p, const float (&v)[8], float mul) {
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lo |= ((unsigned)rni_sat_s8(v[j] * mul) & 0xFFu) << (8 * j);
        hi |= ((unsigned)rni_sat_s8(v[4 + j] * mul) & 0xFFu) << (8 * j);
    }
    *reinterpret_cast<uint2*>(p) = make_uint2(lo, hi);
}
