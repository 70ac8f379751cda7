// row_ops.h -- the per-token row operations of the activation side (quantiser, general layer norm + quantiser) as
// device functions behind the stand-alone row kernels (fused_small.hip).  (Round 3 also ran them as tails of the decode
// GEMM launches - the workgroup that finished a token row ran the following row op itself, bit-identical, measured slower
// than the kernel boundary it saved and removed in round 4: HISTORY.md; the virtual-wave layout, the cache-bypassing
// loads `SC` and the `ready` hook below are what made that possible.)
//
// Behaviour follows (not code):
//   invoke_quant(_fuse_sum) ......... kernels/csrc/fused_kernels.cu:52-137
//   rms_norm_general(_fuse_sum) ..... kernels/csrc/layernorm_kernels.cu:20-29,189-326,427-508
//
// Thread layout.  Each op is DEFINED for NT = 64 * NVW "virtual threads" per token row (NVW virtual waves): virtual thread
// t owns the 8-element chunks (c * NT + t) * 8, c = 0 .. NC-1; statistics are reduced per virtual wave with the wave64
// butterfly of common.h and combined over the virtual waves left to right through LDS.  A workgroup of PW physical waves
// executes it with VPW = NVW / PW virtual threads per physical thread (virtual thread tid + j * 64 * PW, i.e. virtual wave
// wave + j * PW).  The arithmetic - and therefore every rounding - depends on (NVW, NC) only, never on PW: the 256-thread
// stand-alone kernel (NVW = PW = 4) and a 512-thread GEMM workgroup running the same (NVW, NC) produce identical bits.
#pragma once
#include "common.h"

namespace qs_row {

// rows wider than this use the 1024-virtual-thread layout (same rule for invoke_quant and silu_and_mul_quant, so the
// two associate their fp32 statistics identically)
constexpr int WIDE_ROW = 4096;

__device__ __forceinline__ h8 load8(const _Float16* p) { return *reinterpret_cast<const h8*>(p); }

// 16-byte loads that bypass the caches (sc0 sc1): rows another workgroup of the SAME launch published with write-through
// stores (the fence-free seam of gemm_w4a8_ring.hip / attention_mfma.hip).  Buffer loads through the BUILTIN (aux 17 =
// sc0 | sc1): the compiler counts them and waits before the first use.  (An inline-asm load would be invisible to it: the
// first version of this file had one, and hipcc copied its destination registers before the data had landed -
// cdna_hip_programming.md 5.7 item 1.)  `row` must be wave-uniform.
struct ScRow {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ ScRow(const _Float16* row, int elems) {
        const uint64_t a = (uint64_t)(uintptr_t)row;
        const uint64_t u = ((uint64_t)(u32)__builtin_amdgcn_readfirstlane((u32)(a >> 32)) << 32) |
                           (uint64_t)(u32)__builtin_amdgcn_readfirstlane((u32)a);
        rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(u), 0, elems * 2, 0x00020000);
    }
    __device__ __forceinline__ v4u load8(int elem) const {
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, elem * 2, 0, 17);
    }
};

__device__ __forceinline__ float ln_val(float x, float mean, float rstd, float g) {
#pragma clang fp contract(off)
    return (x - mean) * rstd * g;                                        // layernorm_kernels.cu:23
}

struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};

// ---- block reductions over the NVW virtual waves (one barrier each; every round has its own LDS slots) ----------------
template <int NVW, int PW>
__device__ __forceinline__ float reduce_sum(float (&v)[NVW / PW], float* sm, int wave, int lane, bool active) {
    constexpr int VPW = NVW / PW;
    if (active) {
#pragma unroll
        for (int j = 0; j < VPW; ++j) {
            const float r = wave_sum(v[j]);
            if (lane == 0) sm[wave + j * PW] = r;
        }
    }
    __syncthreads();
    float r = sm[0];
#pragma unroll
    for (int w = 1; w < NVW; ++w) r = r + sm[w];
    return r;
}
template <int NVW, int PW>
__device__ __forceinline__ void reduce_max_sum(float (&mx)[NVW / PW], float (&sum)[NVW / PW], float* sm, float* sm2,
                                               bool want_sum, int wave, int lane, bool active, float& mx_out,
                                               float& sum_out) {
    constexpr int VPW = NVW / PW;
    if (active) {
#pragma unroll
        for (int j = 0; j < VPW; ++j) {
            const float m = wave_max(mx[j]);
            float s = 0.f;
            if (want_sum) s = wave_sum(sum[j]);
            if (lane == 0) {
                sm[wave + j * PW] = m;
                if (want_sum) sm2[wave + j * PW] = s;
            }
        }
    }
    __syncthreads();
    float r = sm[0], s = want_sum ? sm2[0] : 0.f;
#pragma unroll
    for (int w = 1; w < NVW; ++w) {
        r = fmaxf(r, sm[w]);
        if (want_sum) s = s + sm2[w];
    }
    mx_out = r;
    sum_out = s;
}

// ---- invoke_quant(_fuse_sum) of one row -------------------------------------------------------------------------------
// out int8 [hidden], in fp16 [hidden]; sum_out may be null.  sm: 2 * NVW floats of LDS.  SC: `in` was published by other
// workgroups of this launch (cache-bypassing loads).  `ready` runs before the first load of `in` (the GEMM tail waits there
// for the row to be complete).  Every thread of the workgroup must call (barriers); `active` = tid < 64 * PW.
template <int NC, int NVW, int PW, bool SC, class Hook = NoHook>
__device__ __forceinline__ void quant_row(int8_t* __restrict__ out, const _Float16* __restrict__ in,
                                          __half* __restrict__ sum_out, __half* __restrict__ scale_out, int hidden,
                                          float* sm, int tid, Hook ready = Hook()) {
#pragma clang fp contract(off)
    constexpr int VPW = NVW / PW, NT = 64 * NVW;
    static_assert(NVW % PW == 0, "virtual waves must split evenly over the physical waves");
    const int wave = tid >> 6, lane = tid & 63;
    const bool active = tid < 64 * PW;
    ready();
    v4u raw[VPW][NC];
    if (active) {
        const ScRow sc(in, SC ? hidden : 0);
#pragma unroll
        for (int j = 0; j < VPW; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c) {                 // all requests first: one memory round trip
                const int i = (c * NT + tid + j * 64 * PW) * 8;
                if (i < hidden) {
                    if (SC) raw[j][c] = sc.load8(i);
                    else raw[j][c] = __builtin_bit_cast(v4u, load8(in + i));
                }
            }
    }
    float amax[VPW], sum[VPW];
#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        amax[j] = 0.f;
        sum[j] = 0.f;
        if (active) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int i = (c * NT + tid + j * 64 * PW) * 8;
                if (i < hidden) {
                    const h8 v = __builtin_bit_cast(h8, raw[j][c]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)v[e];
                        sum[j] += f;
                        amax[j] = fmaxf(amax[j], fabsf(f));
                    }
                }
            }
        }
    }
    float mx, s;
    reduce_max_sum<NVW, PW>(amax, sum, sm, sm + NVW, sum_out != nullptr, wave, lane, active, mx, s);
    if (tid == 0) {
        *scale_out = __float2half_rn(mx / 127.0f);                      // fused_kernels.cu:72
        if (sum_out) *sum_out = __float2half_rn(s);                     // :121
    }
    const float mul = 127.0f / mx;                                      // :78 (unrounded fp32 amax)
    if (active) {
#pragma unroll
        for (int j = 0; j < VPW; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int i = (c * NT + tid + j * 64 * PW) * 8;
                if (i < hidden) {
                    const h8 v = __builtin_bit_cast(h8, raw[j][c]);
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
                    qs_store_q8(out + i, f, mul);
                }
            }
    }
}

// ---- hidden += delta (fp16 add, written back) ; rms_norm_general(_fuse_sum)(hidden) of one row --------------------------
// ADD = false: no residual (plain general_norm_quant; `delta` unused, hidden_io read only).  sm: 4 * NVW floats of LDS.
// SC: `delta` was published by other workgroups of this launch.  `ready` runs after the loads of hidden / gamma were
// requested and before the first load of `delta`.
template <int NC, int NVW, int PW, bool ADD, bool SC, class Hook = NoHook>
__device__ __forceinline__ void norm_quant_row(int8_t* __restrict__ out, _Float16* __restrict__ hidden_io,
                                               const _Float16* __restrict__ delta, const _Float16* __restrict__ gamma,
                                               __half* __restrict__ sum_out, __half* __restrict__ scale_out, float eps,
                                               int hidden, float* sm, int tid, Hook ready = Hook()) {
    // No FMA contraction anywhere in the statistics: `vs += d * d` summed over a row may be fused as fma(d1, d1, round(d0 * d0))
    // or as fma(d0, d0, round(d1 * d1)) - both are legal contractions and hipcc picks differently from one instantiation to
    // the next (seen: the 256-thread kernel vs the same code inlined into a GEMM tail differed in one row sum by one fp16
    // ulp).  Separate roundings are also what the numpy oracle computes.
#pragma clang fp contract(off)
    constexpr int VPW = NVW / PW, NT = 64 * NVW;
    static_assert(NVW % PW == 0, "virtual waves must split evenly over the physical waves");
    const int wave = tid >> 6, lane = tid & 63;
    const bool active = tid < 64 * PW;
    h8 v[VPW][NC], g[VPW][NC];
    v4u dl[VPW][NC];
    // every load of the row is requested before the first one is used (with load and use in one loop body the second
    // chunk's requests left only after the first chunk's data had arrived: two memory round trips instead of one)
    if (active) {
#pragma unroll
        for (int j = 0; j < VPW; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int i = (c * NT + tid + j * 64 * PW) * 8;
                if (i < hidden) {
                    v[j][c] = load8(hidden_io + i);
                    g[j][c] = load8(gamma + i);
                }
            }
    }
    ready();
    if (ADD && active) {
        const ScRow sc(delta, SC ? hidden : 0);
#pragma unroll
        for (int j = 0; j < VPW; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int i = (c * NT + tid + j * 64 * PW) * 8;
                if (i < hidden) {
                    if (SC) dl[j][c] = sc.load8(i);
                    else dl[j][c] = __builtin_bit_cast(v4u, load8(delta + i));
                }
            }
    }
    float s[VPW];
#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        s[j] = 0.f;
        if (active) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int i = (c * NT + tid + j * 64 * PW) * 8;
                if (i < hidden) {
                    // (the sum is written back at the END of the kernel, with the other stores: stored here, the compiler's
                    //  vmcnt(0) at the next control-flow join - it cannot count across the exec-masked load blocks above -
                    //  also waited for this store's acknowledgement, a memory round trip in front of the first reduction)
                    if (ADD) v[j][c] = v[j][c] + __builtin_bit_cast(h8, dl[j][c]);         // the residual add's fp16 add
#pragma unroll
                    for (int e = 0; e < 8; ++e) s[j] += (float)v[j][c][e];
                }
            }
        }
    }
    const float mean = reduce_sum<NVW, PW>(s, sm, wave, lane, active) / hidden;              // :248
    float vs[VPW];
#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        vs[j] = 0.f;
        if (active) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if ((c * NT + tid + j * 64 * PW) * 8 < hidden) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = (float)v[j][c][e] - mean;
                        vs[j] += d * d;
                    }
                }
        }
    }
    const float rstd_e = 1.0f / sqrtf(reduce_sum<NVW, PW>(vs, sm + NVW, wave, lane, active) / hidden + eps);   // :271 (rsqrtf there)
    float amax[VPW], sum[VPW];
#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        amax[j] = (float)(_Float16)1e-6f;                                // :285-286 (amax, sum start values)
        sum[j] = 0.f;
        if (active) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if ((c * NT + tid + j * 64 * PW) * 8 < hidden) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const _Float16 hv = (_Float16)ln_val((float)v[j][c][e], mean, rstd_e, (float)g[j][c][e]);   // cast to half, :292
                        amax[j] = fmaxf(amax[j], fabsf((float)hv));
                        sum[j] += (float)hv;
                    }
                }
        }
    }
    float mx, sm_row;
    reduce_max_sum<NVW, PW>(amax, sum, sm + 2 * NVW, sm + 3 * NVW, sum_out != nullptr, wave, lane, active, mx, sm_row);
    const float mul = 127.f / mx;                                        // :308
    if (active) {
#pragma unroll
        for (int j = 0; j < VPW; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int i = (c * NT + tid + j * 64 * PW) * 8;
                if (i < hidden) {
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = ln_val((float)v[j][c][e], mean, rstd_e, (float)g[j][c][e]);   // fp32, :315
                    qs_store_q8(out + i, f, mul);
                    if (ADD) *reinterpret_cast<h8*>(hidden_io + i) = v[j][c];
                }
            }
    }
    if (tid == 0) {
        *scale_out = __float2half_rn(mx / 127.f);                        // :322
        if (sum_out) *sum_out = __float2half_rn(sm_row);                 // :323
    }
}

}  // namespace qs_row
