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
#include <type_traits>

namespace qs_row {

// The fp32 row statistics are DEFINED as sequential sums over a virtual thread's elements (what the numpy oracle computes).  The
// compiler must not see two consecutive additions of a chain at once: in the branch-free (FULL) instantiations of round 4 it
// packed the chain into v_pk_add_f32 - two interleaved partial sums - and a row sum came out one fp16 ulp away from the
// branchy instantiation of the same source.  An empty asm with the accumulator as in-out operand costs no instruction and makes
// every addition opaque to the vectoriser.
#define QS_SEQ(x) asm volatile("" : "+v"(x))

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

// `delta` of norm_quant_row, by where it comes from.  FromRow: an fp16 row in memory (the residual branch's GEMM output).
// FromPlanes (round 4): the K-slice planes a W4A8 GEMM left instead of its output (qs_w4a8_*_gemm_planes: int32 [KS][M][N]) -
// the 8 values of a chunk are the planes' sums pushed through the GEMM's own epilogue, rounded to fp16 exactly as the GEMM would
// have stored them, so everything downstream sees the same bits.
struct FromRow {
    const _Float16* row;
    struct Raw {
        v4u d;
    };
    __device__ __forceinline__ Raw load(int i) const { return Raw{__builtin_bit_cast(v4u, load8(row + i))}; }
    __device__ __forceinline__ v4u finish(const Raw& r) const { return r.d; }
};
template <int KS, int MODE>
struct FromPlanes {
    const int* row0;          // this token's row in plane 0
    size_t pstride;           // elements between planes (M * N)
    const _Float16 *ws, *wz;  // per-channel weight scale, scale * zero (MODE 0 only)
    float sa, ss;             // the token's activation scale / sum as the GEMM saw them
    int fma = 0;              // per-channel epilogue convention (common.h epi_per_chn; the same the GEMM launches use)
    // two phases, so that every request of a row is out before the first value is needed (one memory round trip)
    struct Raw {
        v4i a[KS][2];
        h8 wsv, wzv;
    };
    __device__ __forceinline__ Raw load(int i) const {
        Raw r;
#pragma unroll
        for (int z = 0; z < KS; ++z) {
            r.a[z][0] = *reinterpret_cast<const v4i*>(row0 + z * pstride + i);
            r.a[z][1] = *reinterpret_cast<const v4i*>(row0 + z * pstride + i + 4);
        }
        r.wsv = load8(ws + i);
        if (MODE == 0) r.wzv = load8(wz + i);
        return r;
    }
    __device__ __forceinline__ v4u finish(const Raw& r) const {
        v4i a0 = r.a[0][0], a1 = r.a[0][1];
#pragma unroll
        for (int z = 1; z < KS; ++z) a0 += r.a[z][0], a1 += r.a[z][1];
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int acc = e < 4 ? a0[e & 3] : a1[e & 3];
            o[e] = MODE == 0 ? (_Float16)epi_per_chn(acc, (float)r.wsv[e], sa, (float)r.wzv[e], ss, fma)
                             : (_Float16)epi_per_group(acc, (float)r.wsv[e], sa);
        }
        return __builtin_bit_cast(v4u, o);
    }
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

// invoke_quant(_fuse_sum)'s statistics (round 6).  The row sum is DEFINED over 512-element blocks (64 chunks of 8): lane l of the
// wave that owns a block adds the 8 elements of chunk l left to right (from +0), the 64 lanes go through the wave butterfly, and the
// (at most 64) block sums through one more wave butterfly, lane b = block b, -0.0 beyond the row - oracle.fused.block_order_row_sum.  Independent of the number of threads a kernel runs, and
// reproducible by ANY holder of a whole block: the decode attention's workgroups (G x 128 values of a row each) publish their
// block sums and every one of them ends with the same bits as invoke_quant_fuse_sum over the finished row (attention_mfma.hip).
// mx[j], sum[j][c]: virtual wave j's maximum and its chain over chunk set c (block c * NVW + wave + j * PW); sm: NVW floats,
// sm2: NC * NVW floats; nblk = ceil(hidden / 512).
template <int NC, int NVW, int PW>
__device__ __forceinline__ void reduce_max_blocksum(float (&mx)[NVW / PW], float (&sum)[NVW / PW][NC], float* sm, float* sm2,
                                                    bool want_sum, int nblk, int wave, int lane, float& mx_out, float& sum_out) {
    constexpr int VPW = NVW / PW;
#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        const float m = wave_max(mx[j]);
        if (lane == 0) sm[wave + j * PW] = m;
        if (want_sum) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float sb = wave_sum(sum[j][c]);
                if (lane == 0) sm2[c * NVW + wave + j * PW] = sb;
            }
        }
    }
    __syncthreads();
    float r = sm[0];
#pragma unroll
    for (int w = 1; w < NVW; ++w) r = fmaxf(r, sm[w]);
    // the block sums combined by ONE more wave butterfly - lane b holds block b's sum, lanes beyond the row hold -0.0 (the identity
    // of fp32 addition: x + -0 = x for every x, +0 and -0 included) -, by wave 0 only (only thread 0 stores the row sum).  A fixed tree
    // of depth 6 whatever the row length.  (As a chain of dependent adds, first on every thread, then on wave 0, it cost the 65 536-row
    // quantiser of the prompt phase 28 % / 14 %.)
    float s = 0.f;
    static_assert(NC * NVW <= 64, "one lane per block sum");
    if (want_sum && wave == 0) s = wave_sum(lane < nblk ? sm2[lane] : -0.0f);
    mx_out = r;
    sum_out = s;
}

// ---- the row sum in the REFERENCE's order (qs_set_row_sum_order(1)) ------------------------------------------------------
// generalLayerNorm_fuse_sum (layernorm_kernels.cu:275-306) runs min(hidden, 1024) threads (rounded up to 32) per token; thread t
// adds the normalised fp16 values of elements t, t + nt, ... into a HALF accumulator (`T_scalar sum`; `sum += float` is the
// half + half operator: one fp16 rounding per addition), the partials are widened to fp32 and all-reduced: xor butterfly
// 16, 8, 4, 2, 1 inside each 32-thread warp, warp results through shared memory, the same butterfly over the (at most 32) warp
// slots (reduction_utils.cuh:68-85).  `hv` = the row's normalised fp16 values in LDS (written by every thread, barrier done);
// `sm` = 32 floats of LDS.  256 physical threads play the nt reference threads four at a time; a wave64 half is a reference warp.
// Returns the sum to thread 0 (other threads: unspecified).
__device__ __forceinline__ float ref_order_row_sum(const _Float16* hv, int hidden, float* sm, int tid) {
#pragma clang fp contract(off)
    int nt = hidden < 1024 ? hidden : 1024;
    nt = 32 * ((nt + 31) / 32);                                          // layernorm_kernels.cu:480-481
    if (tid < 32) sm[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rt = tid + 256 * j;                                    // reference thread; 32-aligned groups stay together
        _Float16 acc = (_Float16)0.f;
        if (rt < nt)
            for (int i = rt; i < hidden; i += nt) {
                acc = acc + hv[i];                                       // fp16 add (v_add_f16), one rounding (:286)
            }
        float f = (float)acc;
#pragma unroll
        for (int m = 16; m > 0; m >>= 1) f = f + __shfl_xor(f, m, 32);   // warpReduceSum, reduction_utils.cuh:25-30
        if ((tid & 31) == 0 && rt < nt) sm[rt >> 5] = f;
    }
    __syncthreads();
    float w = tid < 32 ? sm[tid] : 0.f;                                  // slots beyond nt / 32 hold 0 (:82)
#pragma unroll
    for (int m = 16; m > 0; m >>= 1) w = w + __shfl_xor(w, m, 32);
    return w;
}

// ---- invoke_quant(_fuse_sum) of one row -------------------------------------------------------------------------------
// out int8 [hidden], in fp16 [hidden]; sum_out may be null.  sm: (1 + NC) * NVW floats of LDS.  SC: `in` was published by other
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
    constexpr bool active = true;   // (every thread of the workgroup owns elements: the launch has exactly 64 PW threads.  The
                                    //  round-3 GEMM tails ran these functions on a subset of a larger workgroup; as a run-time
                                    //  test the compiler zero-fills and copies every loaded register at the join - with a wait)
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
    float amax[VPW], sum[VPW][NC];
#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        amax[j] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            sum[j][c] = 0.f;                                // (a chunk beyond the row contributes +0 to its block)
            const int i = (c * NT + tid + j * 64 * PW) * 8;
            if (active && i < hidden) {
                const h8 v = __builtin_bit_cast(h8, raw[j][c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[e];
                    sum[j][c] += f;
                    QS_SEQ(sum[j][c]);
                    amax[j] = fmaxf(amax[j], fabsf(f));
                }
            }
        }
    }
    float mx, s;
    reduce_max_blocksum<NC, NVW, PW>(amax, sum, sm, sm + NVW, sum_out != nullptr, (hidden + 511) / 512, wave, lane, mx, s);
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
// REFSUM (round 6): the row sum in the reference's own order (ref_order_row_sum above) instead of this kernel's fp32 chains;
// `hvbuf` = `hidden` halves of LDS, `sm` then needs 4 * NVW + 32 floats.  Everything else is unchanged, bit for bit.
template <int NC, int NVW, int PW, bool ADD, bool SC, class Hook = NoHook, class DeltaFn = FromRow, bool FULL = false,
          bool REFSUM = false>
__device__ __forceinline__ void norm_quant_row(int8_t* __restrict__ out, _Float16* __restrict__ hidden_io,
                                               const _Float16* __restrict__ delta, const _Float16* __restrict__ gamma,
                                               __half* __restrict__ sum_out, __half* __restrict__ scale_out, float eps,
                                               int hidden, float* sm, int tid, Hook ready = Hook(), DeltaFn dfn = DeltaFn(),
                                               _Float16* hvbuf = nullptr) {
    // FULL: the row is exactly NC * NT * 8 values wide (every chunk of every thread exists): no exec-masked blocks, so the compiler
    // can count the outstanding loads across them and issue the final stores back to back (round 4: with the masks it put a
    // vmcnt(0) - a store acknowledgement - between the two chunks' stores).
    // No FMA contraction anywhere in the statistics: `vs += d * d` summed over a row may be fused as fma(d1, d1, round(d0 * d0))
    // or as fma(d0, d0, round(d1 * d1)) - both are legal contractions and hipcc picks differently from one instantiation to
    // the next (seen: the 256-thread kernel vs the same code inlined into a GEMM tail differed in one row sum by one fp16
    // ulp).  Separate roundings are also what the numpy oracle computes.
#pragma clang fp contract(off)
    constexpr int VPW = NVW / PW, NT = 64 * NVW;
    static_assert(NVW % PW == 0, "virtual waves must split evenly over the physical waves");
    static_assert(!REFSUM || (NVW == 4 && PW == 4), "the reference-order row sum is written for the 256-thread row kernels");
    const int wave = tid >> 6, lane = tid & 63;
    constexpr bool active = true;   // (every thread of the workgroup owns elements: the launch has exactly 64 PW threads.  The
                                    //  round-3 GEMM tails ran these functions on a subset of a larger workgroup; as a run-time
                                    //  test the compiler zero-fills and copies every loaded register at the join - with a wait)
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
                if (FULL || i < hidden) {
                    v[j][c] = load8(hidden_io + i);
                    g[j][c] = load8(gamma + i);
                }
            }
    }
    ready();
    if (ADD && active) {
        if constexpr (std::is_same<DeltaFn, FromRow>::value) {
            const ScRow sc(delta, SC ? hidden : 0);
#pragma unroll
            for (int j = 0; j < VPW; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int i = (c * NT + tid + j * 64 * PW) * 8;
                    if (FULL || i < hidden) {
                        if (SC) dl[j][c] = sc.load8(i);
                        else dl[j][c] = __builtin_bit_cast(v4u, load8(delta + i));
                    }
                }
            if (FULL) __builtin_amdgcn_sched_barrier(0);   // every request of the row is out before the first value is used
        } else {
            typename DeltaFn::Raw raw[VPW][NC];
#pragma unroll
            for (int j = 0; j < VPW; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int i = (c * NT + tid + j * 64 * PW) * 8;
                    if (FULL || i < hidden) raw[j][c] = dfn.load(i);
                }
            // (without the exec-masked blocks the scheduler is free to sink a chunk's requests behind the previous chunk's
            //  arithmetic - seen in the ISA: one memory round trip per chunk; nothing may cross this point)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < VPW; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int i = (c * NT + tid + j * 64 * PW) * 8;
                    if (FULL || i < hidden) dl[j][c] = dfn.finish(raw[j][c]);
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
                if (FULL || i < hidden) {
                    // (the sum is written back at the END of the kernel, with the other stores: stored here, the compiler's
                    //  vmcnt(0) at the next control-flow join - it cannot count across the exec-masked load blocks above -
                    //  also waited for this store's acknowledgement, a memory round trip in front of the first reduction)
                    if (ADD) v[j][c] = v[j][c] + __builtin_bit_cast(h8, dl[j][c]);         // the residual add's fp16 add
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s[j] += (float)v[j][c][e];
                        QS_SEQ(s[j]);
                    }
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
                if (FULL || (c * NT + tid + j * 64 * PW) * 8 < hidden) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = (float)v[j][c][e] - mean;
                        vs[j] += d * d;
                        QS_SEQ(vs[j]);
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
                if (FULL || (c * NT + tid + j * 64 * PW) * 8 < hidden) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        // (the fp32 value is made opaque before the conversion: left to itself the compiler turns "multiply,
                        //  convert" into ONE v_fma_mixlo_f16 - a single rounding - in some instantiations and into v_mul + v_cvt
                        //  in others; the row statistics then differ by an fp16 ulp between kernels that share this source)
                        float hvf = ln_val((float)v[j][c][e], mean, rstd_e, (float)g[j][c][e]);
                        QS_SEQ(hvf);
                        const _Float16 hv = (_Float16)hvf;                                                     // cast to half, :292
                        amax[j] = fmaxf(amax[j], fabsf((float)hv));
                        if (REFSUM) {
                            hvbuf[(c * NT + tid + j * 64 * PW) * 8 + e] = hv;
                        } else {
                            sum[j] += (float)hv;
                            QS_SEQ(sum[j]);
                        }
                    }
                }
        }
    }
    float mx, sm_row;
    reduce_max_sum<NVW, PW>(amax, sum, sm + 2 * NVW, sm + 3 * NVW, !REFSUM && sum_out != nullptr, wave, lane, active, mx, sm_row);
    if (REFSUM) sm_row = ref_order_row_sum(hvbuf, hidden, sm + 4 * NVW, tid);   // (reduce_max_sum's barrier published hvbuf)
    const float mul = 127.f / mx;                                        // :308
    if (active) {
#pragma unroll
        for (int j = 0; j < VPW; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int i = (c * NT + tid + j * 64 * PW) * 8;
                if (FULL || i < hidden) {
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
