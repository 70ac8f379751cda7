// DROPPED EXPERIMENT (round 4) - not built, not part of the library.  Result on the MI355X: correct (logits as accurate as the BLAS
// library's, identical argmax), 293 us = 3.6 TB/s against hipBLASLt's 193-205 us = 5.1-5.4 TB/s: the MFMA A layout makes every
// 16-byte lane load a HALF-line request (16 rows x 64 B per instruction), and a CU's vector memory path holds a fixed number of
// outstanding requests - half lines halve its bytes in flight (14 GB/s per CU instead of the 23 the stream needs).  A version that
// reads whole lines has to go through LDS (LDS-DMA + source-side swizzle, as the W4A8 kernels do); hipBLASLt already streams this
// GEMM at 5.4 TB/s, so what is left to win is the 9 us argmax launch and 32 MB of logits traffic: not pursued.  HISTORY.md.
// lm_head.hip -- the un-quantised fp16 output projection of the decode step, fused with the greedy sampler's argmax.
//
// Engine-side helper, not a qserve_backend op: the reference keeps `lm_head` as a torch fp16 Linear and samples in torch
// (qserve/modeling/models/llama_w4a8_unpad.py:392,476; layers/sampler.py).  At decode batch the projection is a pure weight
// stream (Llama-3: 128 256 x 4096 fp16 = 1.05 GB per step, 7 % of the step through the BLAS library at 5.1 TB/s + 16 MB of
// logits written and re-read by the argmax kernel); here one launch streams the weights once and keeps only the per-token
// maximum:
//   * a workgroup = 8 wave64 owns 512 vocabulary rows (64 per wave: 4 x 4 accumulator tiles of v_mfma_f32_16x16x32_f16 over up
//     to 64 tokens) and walks K in stages of 64; the activation tile of a stage (64 tokens x 64 k, 8 KiB) is shared through a
//     two-slot LDS image, so the [M, K] activations are fetched once per workgroup = 12 % on top of the weight bytes;
//   * the weight operand goes global -> VGPR directly (each byte has exactly one consumer lane: lane (i, g) of a wave holds
//     k = 8 g .. 8 g + 7 of vocabulary row i of a tile - the MFMA A layout - so a 16-byte non-temporal load per tile and k-step
//     IS the operand; the two k-steps of a stage touch the same 128-byte lines), double-buffered one stage ahead: 64 KiB in
//     flight per CU, which is what a CU's vector memory path holds anyway;
//   * epilogue: fp32 accumulators rounded to fp16 exactly where the library GEMM rounds its output, then (value, lowest index)
//     maxima per token over lanes -> waves -> workgroups; a 64-thread kernel picks the winner over the workgroups' candidates.
//     The first maximum wins, as torch.argmax does on the fp16 logits.
// The accumulation ORDER differs from the BLAS library's, so individual logits differ in the last fp16 ulp and a token can
// differ where the two largest logits of a row are that close (tests/test_lm_head_gpu.py states the tolerance).
#include "common.h"
#include <type_traits>

namespace {

constexpr int LM_WAVES = 8;
constexpr int LM_ROWS = 64 * LM_WAVES;   // vocabulary rows per workgroup
constexpr int LM_ASTR = 72;              // row stride (halfs) of the activation image: 64 k + 8 pad -> conflict-free 16-byte reads
constexpr int LM_MAXBLK = 8192;          // workgroups the candidate scratch holds (4.2 M vocabulary rows)

template <bool LOGITS, int UNR>
__global__ __launch_bounds__(512, 1) void lm_head_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ W,
                                                         _Float16* __restrict__ logits, int64_t ldl,
                                                         float* __restrict__ cand_val, int* __restrict__ cand_idx, int M, int N,
                                                         int K) {
    __shared__ __attribute__((aligned(16))) _Float16 s_a[2][64 * LM_ASTR];
    __shared__ float s_cv[LM_WAVES][64];
    __shared__ int s_ci[LM_WAVES][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * LM_ROWS + wave * 64;
    // Loads are BUFFER loads: base in a resource, a loop-invariant 32-bit byte offset per lane, the k advance in the scalar
    // offset - no vector address arithmetic in the loop.  (With flat pointers the compiler keeps one bumped 64-bit pointer per
    // load and parks them in the idle operand buffer: the write-after-read on those registers made it drain the whole queue
    // before every stage.)  Rows beyond the vocabulary / tokens beyond M read zeros (out-of-range buffer reads) and are masked.
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(W), 0, (int)((size_t)N * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(x), 0, (int)((size_t)M * K * 2), 0x00020000);
    u32 woff[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const u32 r = (u32)(row0 + rt * 16 + i);
        woff[rt] = r < (u32)N ? (r * (u32)K + g * 8) * 2u : 0xFFFFFF00u;   // (beyond every buffer this entry admits)
    }
    const int ta = tid >> 3, ca = tid & 7;                                // activation loader: token ta, 16-byte chunk ca of the stage
    const u32 aoff = ta < M ? ((u32)ta * (u32)K + ca * 8) * 2u : 0xFFFFFF00u;
    const int a_wr = ta * LM_ASTR + ca * 8;
    v4f acc[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[rt][tt] = (v4f){0.f, 0.f, 0.f, 0.f};
    h8 w0[4][2], w1[4][2];
    auto loadw = [&](h8(&w)[4][2], int k0) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)     // aux 2 = nt: every weight byte is read exactly once
                w[rt][ks] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(wrs, woff[rt] + ks * 64, k0 * 2, 2));
    };
    const int nst = K >> 6;
    loadw(w0, 0);
    *reinterpret_cast<v4u*>(&s_a[0][a_wr]) = __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff, 0, 0);
    // one stage: the next stage's operands are requested first, then this stage's MFMAs run from the registers / the LDS slot.
    // MORE (is there a next stage) is a compile-time fact of the call site, and UNR stages form one straight-line block: inside
    // a block the compiler counts the outstanding loads exactly (vmcnt(9) in front of a stage's first MFMA: the nine requests of
    // the next stage stay in flight); at a loop back-edge it merges states and waits for part of the NEW requests - with the
    // two-stage loop this kernel started as, every stage exposed a memory round trip.
    auto stage = [&](int s, auto more_c, h8(&wc)[4][2], h8(&wn)[4][2]) {
        constexpr bool more = decltype(more_c)::value;
        v4u an = {0, 0, 0, 0};
        if (more) {
            an = __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff, (s + 1) * 128, 0);
            loadw(wn, (s + 1) * 64);
        }
        __syncthreads();                                                  // slot s & 1 is complete; slot (s + 1) & 1 has been read
        const _Float16* const sa = &s_a[s & 1][0];
        h8 b[4][2];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b[tt][ks] = *reinterpret_cast<const h8*>(sa + (tt * 16 + i) * LM_ASTR + ks * 32 + g * 8);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    acc[rt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[rt][ks], b[tt][ks], acc[rt][tt], 0, 0, 0);
        if (more) *reinterpret_cast<v4u*>(&s_a[(s + 1) & 1][a_wr]) = an;
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    int s = 0;
    for (; s + UNR < nst; s += UNR) {                                     // (nst % UNR == 0, checked by the entry)
#pragma unroll
        for (int j = 0; j < UNR; j += 2) {
            stage(s + j, yes{}, w0, w1);
            stage(s + j + 1, yes{}, w1, w0);
        }
    }
#pragma unroll
    for (int j = 0; j < UNR; j += 2) {                                    // the last UNR stages: no request behind the final one
        stage(s + j, yes{}, w0, w1);
        if (j + 2 < UNR) stage(s + j + 1, yes{}, w1, w0);
        else stage(s + j + 1, no{}, w1, w0);
    }
    // ---- epilogue -------------------------------------------------------------------------------------------------------
    // lane (i, g) holds D[row = 16 rt + 4 g + r][token = 16 tt + i], r = 0..3
    if (LOGITS) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int tok = tt * 16 + i;
            if (tok >= M) continue;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const int row = row0 + rt * 16 + g * 4;
                _Float16* const dst = logits + (size_t)tok * ldl + row;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row + r < N) dst[r] = (_Float16)acc[rt][tt][r];
            }
        }
        return;
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        float bv = -3.0e38f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                  // ascending rows: ">" keeps the first maximum
                const int row = row0 + rt * 16 + g * 4 + r;
                const float v = (float)(_Float16)acc[rt][tt][r];          // the fp16 logit the library GEMM would have written
                if (row < N && v > bv) bv = v, bi = row;
            }
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
        }
        if (g == 0) {
            s_cv[wave][tt * 16 + i] = bv;
            s_ci[wave][tt * 16 + i] = bi;
        }
    }
    __syncthreads();
    if (tid < 64) {
        float bv = s_cv[0][tid];
        int bi = s_ci[0][tid];
#pragma unroll
        for (int w = 1; w < LM_WAVES; ++w) {
            const float ov = s_cv[w][tid];
            const int oi = s_ci[w][tid];
            if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
        }
        cand_val[(size_t)blockIdx.x * 64 + tid] = bv;
        cand_idx[(size_t)blockIdx.x * 64 + tid] = bi;
    }
}

// out[t] = the candidate with the largest value, lowest index on ties, over the nblk workgroups (ascending = ascending rows)
__global__ __launch_bounds__(64) void lm_head_pick_kernel(const float* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                                          int64_t* __restrict__ out, int M, int nblk) {
    const int t = threadIdx.x;
    if (t >= M) return;
    float bv = cand_val[t];
    int bi = cand_idx[t];
    for (int b = 1; b < nblk; ++b) {
        const float ov = cand_val[(size_t)b * 64 + t];
        const int oi = cand_idx[(size_t)b * 64 + t];
        if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
    }
    out[t] = bi;
}

// candidate scratch: one fixed allocation per device ([LM_MAXBLK][64] float + int), made on a first EAGER call, never freed
struct LmScratch {
    float* val = nullptr;
    int* idx = nullptr;
};
LmScratch g_lm[QS_MAX_DEVICES];

LmScratch* lm_scratch(hipStream_t st) {
    LmScratch& s = g_lm[qs_device_slot()];
    if (s.val) return &s;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)LM_MAXBLK * 64 * 8) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    s.val = reinterpret_cast<float*>(p);
    s.idx = reinterpret_cast<int*>(s.val + (size_t)LM_MAXBLK * 64);
    return &s;
}

int lm_check(const void* x, const void* w, int M, int N, int K, const char* who) {
    QS_REQUIRE(x && w, "%s: null pointer", who);
    QS_REQUIRE(M >= 1 && M <= 64, "%s: %d tokens (1..64 supported: the decode batch of one launch)", who, M);
    QS_REQUIRE(N >= 1 && K >= 128 && K % 128 == 0, "%s: N=%d K=%d (K must be a positive multiple of 128)", who, N, K);
    QS_REQUIRE((size_t)(N + LM_ROWS - 1) / LM_ROWS <= (size_t)LM_MAXBLK, "%s: N=%d exceeds %d rows", who, N, LM_MAXBLK * LM_ROWS);
    QS_REQUIRE((size_t)N * K * 2 < (1ull << 32) && (size_t)M * K * 2 < (1ull << 32), "%s: N x K x 2 bytes must stay below 4 GiB", who);
    QS_REQUIRE(!((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15), "%s: x and w must be 16-byte aligned", who);
    return QS_OK;
}

}  // namespace

extern "C" int qs_lm_head_argmax(const void* x, const void* w, int64_t* out, int M, int N, int K, qs_stream_t stream) {
    if (const int e = lm_check(x, w, M, N, K, "lm_head_argmax")) return e;
    QS_REQUIRE(out, "lm_head_argmax: null pointer");
    hipStream_t st = (hipStream_t)stream;
    LmScratch* sc = lm_scratch(st);
    if (!sc) {                                        // (first use inside a stream capture: call once eagerly before capturing)
        qs_set_error("lm_head_argmax: the candidate scratch cannot be allocated (first call inside a stream capture?)");
        return QS_ENOSUP;
    }
    const int nblk = (N + LM_ROWS - 1) / LM_ROWS;
    auto kern = K % 512 == 0 ? lm_head_kernel<false, 8> : lm_head_kernel<false, 2>;   // stages per straight-line block
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), 0, st, reinterpret_cast<const _Float16*>(x),
                       reinterpret_cast<const _Float16*>(w), (_Float16*)nullptr, (int64_t)0, sc->val, sc->idx, M, N, K);
    hipLaunchKernelGGL(lm_head_pick_kernel, dim3(1), dim3(64), 0, st, sc->val, sc->idx, out, M, nblk);
    return qs_launch_status("lm_head_argmax");
}

extern "C" int qs_lm_head_logits(const void* x, const void* w, void* logits, int64_t row_stride, int M, int N, int K,
                                 qs_stream_t stream) {
    if (const int e = lm_check(x, w, M, N, K, "lm_head_logits")) return e;
    QS_REQUIRE(logits && row_stride >= N, "lm_head_logits: null pointer or row stride below N");
    const int nblk = (N + LM_ROWS - 1) / LM_ROWS;
    auto kern = K % 512 == 0 ? lm_head_kernel<true, 8> : lm_head_kernel<true, 2>;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16*>(x), reinterpret_cast<const _Float16*>(w),
                       reinterpret_cast<_Float16*>(logits), row_stride, (float*)nullptr, (int*)nullptr, M, N, K);
    return qs_launch_status("lm_head_logits");
}
