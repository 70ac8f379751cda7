/*
 * qserve_amd.h -- C ABI of libqserve_amd.so: MI355X (gfx950) implementation of QServe's W4A8KV4 hot path.
 *
 * This is the drop-in boundary.  Every entry point below is what the reference's pybind11 torch-extension
 * functions (package `qserve_backend`, kernels/setup.py:157-245) reduce to once the torch::Tensor arguments
 * are lowered to device pointers + sizes.  The Python package `qserve_backend/` in this repository is the
 * host-side mirror that performs exactly that lowering (same module names, same callables, same argument
 * order and error behaviour); INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless stated otherwise; `half` data is passed as `const void*`/`void*`
 *     to IEEE binary16 storage (the ABI carries no torch / HIP types);
 *   - `stream` is a hipStream_t passed as void* (NULL = legacy default stream, which is what the reference
 *     GEMMs use: gemm_cuda.cu:53);
 *   - return value: 0 on success, a negative QS_E* code on rejected arguments (nothing was launched),
 *     a positive hipError_t if the launch failed.  qs_last_error() returns a thread-local message;
 *   - callers own every tensor buffer (ownership rules of the reference, SURVEY.md 8(b) "Conventions"); the library
 *     keeps a few device scratch areas of its own (RoPE cos/sin tables, split-KV partials, split-K slabs - 2 x 48 MiB, one of
 *     them sentinel-filled between launches -, the split argmax's keys / tickets, the generation words and exchange rows of
 *     the fused attention quantiser - 64 MiB, batches up to 4096 sequences, larger ones run the un-fused pair).  Each
 *     is a fixed-size allocation made lazily on a first EAGER call (never while the stream is being captured into a
 *     graph: such a call runs the variant that needs no scratch) and is NEVER freed, moved or grown afterwards, so a
 *     hipGraph that captured its address stays valid for the life of the process.  Requests beyond the fixed capacity
 *     fall back to the un-split variants.  The scratch areas exist once per device (the CURRENT device of the calling thread),
 *     shared by every stream: launches that use them (split-KV attention, K-sliced GEMMs, the split argmax, the fused attention
 *     quantiser's hand-over) must not run concurrently on different streams of one device (the reference's engine is
 *     single-stream) - UNLESS the extra streams were given scratch of their own with qs_stream_scratch_bind() (below: up to 7
 *     streams per device; launches issued on, or captured on, a bound stream use that stream's areas).
 *     A launch that is ABORTED mid-way (device reset) may leave the K-slice slabs without their sentinel or an exchange row
 *     half-tagged: call qs_device_reset() before the library is used again.  The in-launch waits on these areas are BOUNDED
 *     (round 5): a violated assumption yields a status bit (qs_device_status) and an invalid result, never a hung GPU.
 */
#ifndef QSERVE_AMD_H
#define QSERVE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QS_OK 0
#define QS_EINVAL (-1)   /* bad shape / alignment / null pointer */
#define QS_ENOSUP (-2)   /* combination the reference never instantiates (e.g. head_dim != 128) */

typedef void* qs_stream_t;

/* Library identification: version = 100*major + minor; arch string is "gfx950". */
int qs_version(void);
const char* qs_arch(void);
const char* qs_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * W4A8 per-channel GEMM.
 * Replaces qserve_backend.qgemm_w4a8_per_chn.gemm_forward_cuda
 *   (kernels/csrc/qgemm/w4a8_per_chn/gemm_cuda.h:11, gemm_cuda.cu:596-652, pybind.cpp:13-16).
 *   in_feats  int8  [M,K] row-major          kernel   int8 [N,K/2]  reference packed layout (w4a8_linear.py:290-322)
 *   wscales   half  [N]                       ascales  half [M]
 *   w_szs     half  [N]  (= zero*scale)       a_ssums  half [M]  (= sum_k x)
 *   out_feats half  [M,N] written in place:   (float(acc)*wscale[n])*ascale[m] - w_sz[n]*a_ssum[m]
 * Requirements: N % 64 == 0, K % 128 == 0 (the reference requires N % CTA_N, K % CTA_K; all model shapes comply).
 * ---------------------------------------------------------------------------------------------------------- */
int qs_w4a8_per_chn_gemm(const int8_t* in_feats, const int8_t* kernel, const void* wscales, const void* ascales,
                         const void* w_szs, const void* a_ssums, void* out_feats, int M, int N, int K,
                         qs_stream_t stream);

/* W4A8 per-group (g128) GEMM.
 * Replaces qserve_backend.qgemm_w4a8_per_group.gemm_forward_cuda
 *   (kernels/csrc/qgemm/w4a8_per_group/gemm_cuda.h:11, gemm_cuda.cu:630-702).
 *   zeros, scales_i8  int8 [K/128, N] in the reference's per-32 channel permutation (w4a8_linear.py:231-277)
 *   out = float(acc) * (wscale[n]*ascale[m]),  acc over the level-2 dequantised int8 weights. */
int qs_w4a8_per_group_gemm(const int8_t* in_feats, const int8_t* kernel, const int8_t* zeros,
                           const int8_t* scales_i8, const void* wscales, const void* ascales, void* out_feats,
                           int M, int N, int K, qs_stream_t stream);

/* gate_up GEMM + silu_and_mul in one launch (engine-side fusion, like the pairs further down; no reference op of its
 * own: LlamaMLP.forward, llama_w4a8_unpad.py:69-93, issues gate_up_proj, then SiluAndMulQuant = silu_and_mul ; invoke_quant).
 *   kernel   the stacked gate_up weight [N, K/2], rows 0 .. N/2-1 = gate, N/2 .. N-1 = up (load_weights' row
 *            concatenation), consumed unchanged;   out_act half [M, N/2];
 *   out_act[m, c] = half(float(silu_h(Y[m, c])) * float(Y[m, N/2 + c]))  with Y = the fp16 result of the plain GEMM entry
 *            above on the same arguments - i.e. bit-identical to  qs_w4a8_*_gemm(..., tmp) ; qs_silu_and_mul(out_act, tmp).
 *   tmp      half [M, N] scratch or NULL: shapes whose kernel family has no activation epilogue (K-sliced geometries,
 *            the round-1 kernels) run as those two launches through it; NULL makes them an error.
 * Requirements: those of the plain entries and N % 128 == 0. */
int qs_w4a8_per_chn_gemm_silu_mul(const int8_t* in_feats, const int8_t* kernel, const void* wscales,
                                  const void* ascales, const void* w_szs, const void* a_ssums, void* out_act, void* tmp,
                                  int M, int N, int K, qs_stream_t stream);
int qs_w4a8_per_group_gemm_silu_mul(const int8_t* in_feats, const int8_t* kernel, const int8_t* zeros,
                                    const int8_t* scales_i8, const void* wscales, const void* ascales, void* out_act,
                                    void* tmp, int M, int N, int K, qs_stream_t stream);

/* Debug/parity entry points: same kernels, but the raw INT32 accumulators are written to acc_out [M,N]
 * instead of the fp16 epilogue (the reference keeps them in registers: gemm_cuda.cu:327). */
int qs_w4a8_per_chn_gemm_acc(const int8_t* in_feats, const int8_t* kernel, int32_t* acc_out, int M, int N, int K,
                             qs_stream_t stream);
int qs_w4a8_per_group_gemm_acc(const int8_t* in_feats, const int8_t* kernel, const int8_t* zeros,
                               const int8_t* scales_i8, int32_t* acc_out, int M, int N, int K, qs_stream_t stream);

/* K-slice PLANES (engine-side pair fusion, round 4): the row-parallel GEMMs of a layer (o_proj, down_proj) are followed by a row
 * kernel that reads their whole output (residual add + norm + quant, llama_w4a8_unpad.py:345-361).  In this form the GEMM leaves
 * the int32 partial sums of its K slices as planes [k_slices][M][N] - no cross-workgroup reduction, no epilogue - and
 * qs_add_residual_rms_norm_general_planes (below) sums them and applies the GEMM's epilogue arithmetic itself, bit for bit.
 *   qs_w4a8_gemm_planes_plan: plan4 = {k_slices, m_tiles, units, token_blocks} the planes launch of this shape will use;
 *                             k_slices == 0: no such launch for this shape (run the ordinary pair).  Deterministic in the shape.
 *   planes: int32 [k_slices][M][N], 16-byte aligned; every element of every plane is written. */
int qs_w4a8_gemm_planes_plan(int per_group, int M, int N, int K, int* plan4);
int qs_w4a8_per_chn_gemm_planes(const int8_t* in_feats, const int8_t* kernel, int32_t* planes, int M, int N, int K,
                                qs_stream_t stream);
int qs_w4a8_per_group_gemm_planes(const int8_t* in_feats, const int8_t* kernel, const int8_t* zeros, const int8_t* scales_i8,
                                  int32_t* planes, int M, int N, int K, qs_stream_t stream);

/* W8A8 GEMM (importable-module requirement only).
 * Replaces qserve_backend.qgemm_w8a8.w8a8_gemm_forward_cuda (kernels/csrc/qgemm/w8a8/w8a8_gemm_cuda.h:11).
 *   kernel int8 [N,K] row-major.  out = float(acc) * (wscale[n]*ascale[m]).  N % 16 == 0, K % 64 == 0. */
int qs_w8a8_gemm(const int8_t* in_feats, const int8_t* kernel, const void* wscales, const void* ascales,
                 void* out_feats, int M, int N, int K, qs_stream_t stream);

/* Kernel-variant selection for benchmarking / A-B tests (process-wide; default -1 = the measured heuristic).  Like every
 * qs_set_* / qs_debug_* switch of this header it is a relaxed atomic inside the library: setting it while another host thread
 * launches is not a data race - that launch sees the old or the new value -, but it is one value for the whole process, not a
 * per-stream or per-thread configuration.  EVERY
 * variant of the shipped library computes the same results; the timing experiments that switch kernel parts off (wrong
 * results by design) exist only in libraries built with -DQS_TIMING (python -m qserve_amd.build --timing ->
 * libqserve_amd_timing.so, loaded by the measurement scripts through QS_AMD_LIBRARY) and are ignored otherwise:
 *   9xx / 1000 + 100*mtile + 10*S + NW ... split-K decode kernel geometries;  2000 / 2001 ... LDS-pair kernel off / forced;
 *   3000 ... tiled (prefill) kernel off, 3001 / 3002 ... forced with the 256- / 128-token tile;
 *   4000 ... ring (decode) kernel off, 4001 ... ring kernel without K slices, 4002 ... cost model without the per-group term,
 *   4004 ... without the 128-token ring geometry <8,2> (round 5),
 *   4600 + 100*(k_slices-1) + 10*m_tiles + units ... forced geometry of the K-slice planes launches,
 *   3003 ... the four-wave compute-bound tile forced (3001: the eight-wave one), 3400 + bits ... [QS_TIMING builds] its ablations,
 *   4100 + 100*(k_slices-1) + 10*m_tiles + units ... forced ring geometry;
 *   5000 + bits ... A/B switches of the ring kernel (1: weight DMA without the non-temporal hint; 256 * d: ring depth d;
 *                   4096: K slices of a channel block on ONE XCD, the mapping of rounds 3-5 - default since round 6: slices across the
 *                   XCDs so that an L2 holds only its K slice of the activations;
 *                   [QS_TIMING builds: 32 / 64 no MFMA / no operand reads]); sticky until reset with 5000;
 *   3200 + 10*p + o ... tiled kernel A/B, sticky until reset with 3200: tile order o (0 super-tiles with the XCD-aware 4 x 8
 *                   placement inside a super-tile, 3 super-tiles without it, 1 / 2 token- / channel-fastest bands); p = 1 one
 *                   workgroup per tile instead of one per CU walking the tiles,
 *                   p = 2 three workgroups walk all tiles (tests of the tile-to-tile hand-over);
 *   3301 / 3300 ... qs_w4a8_*_gemm_silu_mul always as two launches / default (sticky);
 *   3100 + bits ... [QS_TIMING builds only] kernel parts of the tiled kernel switched off. */
void qs_set_gemm_variant(int variant);

/* The codes above by name (round 6; the dispatcher in qserve_amd/csrc/gemm_w4a8.hip uses these, nothing else defines them).
 * The selection state is PROCESS-GLOBAL and NOT THREAD-SAFE: a test / measurement hook, set between launches by one host
 * thread; a serving process never calls it.  Codes that pick a kernel are remembered until the next call; the sticky families
 * (ring flags, tile order, activation split) keep their own word and are reset by their family's base code. */
enum qs_gemm_variant_code {
    QS_GEMM_DEFAULT = -1,              /* the measured heuristic */
    QS_GEMM_SPLITK_BASE = 1000,        /* + 100 * m_tiles + 10 * cross_block_slices + waves: round-1 split-K kernel geometry */
    QS_GEMM_PAIR_OFF = 2000,           /* the round-1 LDS-pair kernel never / ... */
    QS_GEMM_PAIR_FORCED = 2001,        /* ... always (where its preconditions hold) */
    QS_GEMM_TILED_OFF = 3000,          /* compute-bound kernels off */
    QS_GEMM_TILED_256 = 3001,          /* eight-wave 256-token tile forced */
    QS_GEMM_TILED_128 = 3002,          /* 128-token tile forced */
    QS_GEMM_WIDE_256 = 3003,           /* four-wave 256-token tile forced */
    QS_GEMM_TILED_DEBUG_BASE = 3100,   /* + bits: [QS_TIMING builds] parts of the tiled kernel off (sticky) */
    QS_GEMM_TILE_ORDER_BASE = 3200,    /* + 10 * persist_mode + order (sticky until QS_GEMM_TILE_ORDER_BASE) */
    QS_GEMM_ACT_FUSED = 3300,          /* qs_w4a8_*_gemm_silu_mul as one launch where it can [default] (sticky) */
    QS_GEMM_ACT_SPLIT = 3301,          /* ... always as two launches (sticky) */
    QS_GEMM_WIDE_DEBUG_BASE = 3400,    /* + bits: [QS_TIMING builds] parts of the wide kernel off (sticky) */
    QS_GEMM_RING_OFF = 4000,           /* decode ring kernel off (the round-1 kernels serve its shapes) */
    QS_GEMM_RING_NO_KSLICES = 4001,    /* ring kernel without K slices */
    QS_GEMM_RING_NO_GROUP_TERM = 4002, /* cost model without the per-group term */
    QS_GEMM_RING_NO_DOWN_OVERRIDE = 4003, /* without the measured <2,1> x 2 override for Llama-3's down_proj */
    QS_GEMM_RING_NO_MT8 = 4004,        /* without the 128-token geometry <8,2> */
    QS_GEMM_RING_GEOMETRY_BASE = 4100, /* + 100 * (k_slices - 1) + 10 * m_tiles + units: forced ring geometry */
    QS_GEMM_RING_GEOMETRY_END = 4500,
    QS_GEMM_PLANES_GEOMETRY_BASE = 4600, /* the same for the K-slice planes launches */
    QS_GEMM_PLANES_GEOMETRY_END = 5000,
    QS_GEMM_RING_FLAGS_BASE = 5000,    /* + bits (QS_RING_FLAG_*), sticky until QS_GEMM_RING_FLAGS_BASE */
    QS_GEMM_RING_FLAGS_END = 5000 + 16384
};
enum qs_ring_flag {                    /* qs_set_gemm_variant(QS_GEMM_RING_FLAGS_BASE + bits); results never change unless noted */
    QS_RING_FLAG_WEIGHTS_DEFAULT_POLICY = 1,  /* weight DMA without the non-temporal hint everywhere */
    QS_RING_FLAG_WEIGHTS_NT = 2,              /* ... non-temporal everywhere */
    QS_RING_FLAG_T_NO_REDUCTION = 4,          /* [QS_TIMING, wrong results] no cross-group reduction */
    QS_RING_FLAG_T_LEAVE_AFTER_LOOP = 8,      /* [QS_TIMING, wrong results] leave behind the k loop */
    QS_RING_FLAG_T_NO_MFMA = 32,              /* [QS_TIMING, wrong results] */
    QS_RING_FLAG_T_NO_OPERAND_READS = 64,     /* [QS_TIMING, wrong results] */
    QS_RING_FLAG_DEPTH_UNIT = 256,            /* * d: ring depth d = 3 .. 6 */
    QS_RING_FLAG_KSLICES_ONE_XCD = 4096,      /* K slices of a channel block on one XCD (the mapping of rounds 3-5) */
    QS_RING_FLAG_T_NO_LEVEL2 = 8192           /* [QS_TIMING, wrong results] per-group launches without the level-2 arithmetic */
};

/* Per-channel epilogue CONVENTION (process-wide; read at launch time by every per-channel W4A8 GEMM launch and by
 * qs_add_residual_rms_norm_general_planes, which finishes such a GEMM).  The reference statement
 *   (float(acc) * wscale) * ascale - w_sz * a_ssum          (w4a8_per_chn/gemm_cuda.cu:586-587)
 * is compiled there with nvcc's default --fmad=true, which may contract it; the CUDA binary cannot be produced here, so the
 * convention is selectable:  0 [default] = every operation rounded separately (the statement as written),
 *   1 = fmaf(float(acc) * wscale, ascale, -(w_sz * a_ssum)) - the fold of the last multiply into the subtraction that nvcc most
 *   plausibly performs (oracle/w4a8.py epilogue_per_chn(fma=True); differs on ~3.5e-4 of the outputs at Llama-3-8B shapes).
 * Returns QS_EINVAL for any other value.  Per-group GEMMs have no subtraction and are unaffected. */
int qs_set_gemm_epilogue(int convention);
int qs_get_gemm_epilogue(void);

/* Order of the per-token ROW SUM `a_ssums` written by qs_rms_norm_general (input_sum != NULL),
 * qs_add_residual_rms_norm_general and qs_add_residual_rms_norm_general_planes (process-wide; read at launch time).  It is the
 * value the per-channel GEMM epilogue multiplies by w_sz (w4a8_per_chn/gemm_cuda.cu:586).
 *   0 [default] = this library's order: fp32 chains per thread over 8-element chunks, wave butterfly, waves left to right;
 *   1 = the reference's own order (generalLayerNorm_fuse_sum, kernels/csrc/layernorm_kernels.cu:275-306): min(hidden, 1024)
 *       threads, thread t accumulates elements t, t + nt, ... in a HALF variable (one fp16 rounding per addition), partials
 *       all-reduced in fp32 by the 32-lane xor butterfly inside a warp and again over the warp slots
 *       (reduction_utils.cuh:25-30,68-85) - bit-equal to oracle/fused.py rms_norm_general(with_sum=True, sum_order="reference").
 * Int8 rows and scales do not depend on it.  invoke_quant_fuse_sum sums in fp32 in the reference too (fused_kernels.cu:104-122)
 * and has no second form.  Returns QS_EINVAL for any other value. */
int qs_set_row_sum_order(int order);
int qs_get_row_sum_order(void);

/* Measurement hook (no reference counterpart; SURVEY 8(d): "compute peak from the measured engine clock during the run").  While a
 * buffer is set, every launch of the compute-bound GEMM kernels (tiled / wide: M > 1024 or forced) with at most `workgroups`
 * workgroups makes workgroup b write buf[2 b] = its life in shader cycles (s_memtime) and buf[2 b + 1] = the same interval in
 * ticks of the constant 100 MHz counter (s_memrealtime): cycles / (ticks * 10 ns) = the engine clock that launch held under its own
 * load.  buf = device memory of 16 * workgroups bytes owned by the caller; (NULL, 0) stops it.  Results are unaffected. */
int qs_debug_gemm_clock_probe(void* buf, int workgroups);

/* Plan only: runs the W4A8 GEMM dispatcher for an (M, N, K) problem without touching the device and reports its choice
 * in plan5 = {family, p0, p1, p2, p3}: family 1 = split-K kernel (m_tiles, waves, cross-block slices, xcd mapping),
 * 2 = LDS-pair kernel, 3 = ring kernel (m_tiles, units, token blocks, K slices), 4 = tiled kernel (8 = 256-token tile,
 * 4 = 128-token tile).  No reference counterpart: it makes the selection heuristics testable on a CPU-only machine
 * (tests/test_dispatch_plan.py).  per_group: 0 = per-channel, 1 = per-group-128.  Honours qs_set_gemm_variant. */
int qs_w4a8_gemm_plan(int per_group, int M, int N, int K, int* plan5);

/* ------------------------------------------------------------------------------------------------------------
 * Decode attention over the paged, quantised KV cache.
 * Replaces qserve_backend.fused_attention.single_query_attention
 *   (kernels/csrc/fused_attention/fused_attention.h:13-28, fused_attention.cpp:150-240).
 *   q   half [B,H,Dh]   view, element strides (q_stride0, Dh, 1)
 *   k,v half [B,Hkv,Dh] views, element strides (kv_stride0, Dh, 1)          (un-rotated new token)
 *   kv_pointers int64 [B,2,max_blocks]: device ADDRESSES of K pages ([:,0,:]) and V pages ([:,1,:])
 *   length_per_sample int32 [B]: context length INCLUDING the new token, i.e. the new token sits at position
 *                     length-1 and attends to length-1 cached tokens.  May be NULL: then, exactly as in the reference
 *                     (decoderMaskedMultiheadAttentionTemplate.hpp:901, tlength = length_per_sample ?
 *                     length_per_sample[bi]-1 : timestep), every sequence has `timestep` CACHED tokens and the new token
 *                     is written at position `timestep`
 *   out half [B,H,Dh] contiguous (the reference returns torch::empty_like(q))
 * Page layout (kvCacheUtils.h:47-126): [Hkv][tokens_per_block][Dh'] data, then half scale[Hkv][tpb], then
 * half zero[Hkv][tpb]; Dh' = size_per_token / Hkv bytes.
 * Side effect: quantises the new token's rotated K and V into the page of position length-1.
 * Supported (what the reference instantiates): Dh = 128, tokens_per_block = 64, kv_cache_with_zeros.
 * neox_rotary_style is accepted and has no effect: the reference never forwards it (fused_attention.cpp:109 is commented out;
 * update_kv_cache.cu:57 hard-codes kROPE_GPT_NEOX) - RoPE is always the NeoX pairing (d, d + 64).  Likewise the op-level
 * `alibi_slopes` argument is checked and dropped by the reference (fused_attention.cpp:91,193-199) and by both mirrors here.
 * ---------------------------------------------------------------------------------------------------------- */
int qs_single_query_attention(const void* q, const void* k, const void* v, const int64_t* kv_pointers,
                              const int32_t* length_per_sample, void* out, int batch, int num_heads,
                              int num_kv_heads, int head_dim, int64_t q_stride0, int64_t kv_stride0,
                              int max_blocks, int memory_max_seqlen, int tokens_per_block, int size_per_token,
                              int timestep, int rotary_embedding_dim, float rotary_base, int neox_rotary_style,
                              int int4_kv_cache, int kv_cache_with_zeros, qs_stream_t stream);

/* Pair fusion for the decode loop (no reference counterpart): single_query_attention followed by
 * invoke_quant(_fuse_sum) of its output (llama_w4a8_unpad.py:253-282) as ONE call.  `out` receives the fp16 attention
 * output exactly as above; quant_out int8 [B, H*Dh], quant_scale half [B], quant_sum half [B] (may be NULL) receive what
 * qs_invoke_quant(quant_out, out, quant_sum, quant_scale, B, H*Dh) would write - BIT-IDENTICAL.  Where the chosen
 * attention kernel can finish the row itself (matrix-core KV4 kernel, no KV split, H*Dh <= 4096) this is one launch,
 * otherwise the two launches are issued here.  Uses the per-device arrival-counter scratch (same stream rule as above). */
int qs_single_query_attention_quant(const void* q, const void* k, const void* v, const int64_t* kv_pointers,
                                    const int32_t* length_per_sample, void* out, int8_t* quant_out, void* quant_sum,
                                    void* quant_scale, int batch, int num_heads, int num_kv_heads, int head_dim,
                                    int64_t q_stride0, int64_t kv_stride0, int max_blocks, int memory_max_seqlen,
                                    int tokens_per_block, int size_per_token, int timestep, int rotary_embedding_dim,
                                    float rotary_base, int neox_rotary_style, int int4_kv_cache, int kv_cache_with_zeros,
                                    qs_stream_t stream);

/* Kernel selection for A/B tests: 0 = matrix-core kernels (KV4 and KV8) with the split-KV heuristic [default],
 * 1 = VALU kernels, 2 = prefill writer without the RoPE table, 3 = KV4 matrix-core kernel with the service wave always
 * owning pages (A/B of the page-ownership rule), 5 = one raw barrier instead of the polled operand flag, 7 = the attention +
 * quant fusion hands the PAYLOAD to the last KV head's workgroup also where it otherwise gathers the row statistics (group sizes
 * 4 and 8; A/B of the hand-over, same bits), 100 + n = matrix-core kernels with exactly n KV splits.
 * (200 + bits: ablation / trace instantiations of the KV4 kernel, QS_TIMING builds only; ignored by the shipped library.) */
void qs_set_attention_variant(int variant);

/* Plan only (no device access, CPU-testable like qs_w4a8_gemm_plan): which decode attention kernel the dispatcher
 * takes for (batch, heads, kv heads, page-table width, longest context) and how it splits the context:
 * plan3 = {family, kv_splits, waves per workgroup}; family 1 = matrix-core KV4 kernel, 2 = matrix-core KV8 kernel,
 * 3 = VALU kernel (page tables wider than 192 entries, or forced by qs_set_attention_variant(1)). */
int qs_attention_plan(int batch, int num_heads, int num_kv_heads, int max_blocks, int timestep, int int4_kv_cache,
                      int* plan3);

/* Prefill KV writer.  Replaces qserve_backend.fused_attention.apply_bias_rope_update_kv_cache
 *   (kernels/csrc/fused_attention/update_kv_cache.h:11-27, update_kv_cache.cu:20-108).
 *   qkv half [num_tokens, (H+2Hkv)*Dh] modified in place (rotated q and k are written back);
 *   seq_lens int32 [batch]; padding_offset int32 [num_tokens]; kv_pointers as above (NULL: no cache write). */
int qs_apply_bias_rope_update_kv_cache(void* qkv, const int32_t* seq_lens, const int32_t* padding_offset,
                                       const int64_t* kv_pointers, int num_tokens, int batch, int max_blocks,
                                       int head_num, int kv_head_num, int seq_len, int tokens_per_block,
                                       int size_per_token, int rotary_embedding_dim, float rotary_embedding_base,
                                       int rotary_embedding_max_positions, int neox_rotary_style,
                                       int int4_kv_cache, int kv_cache_with_zeros, qs_stream_t stream);

/* Replaces qserve_backend.fused_attention.compute_padding_offsets (input_metadata_helper.h:12-13). */
int qs_compute_padding_offsets(int32_t* padding_offsets, const int32_t* cu_seqlens, int batch, int max_seqlen,
                               qs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Activation-side kernels adjacent to the hot path (they produce the GEMM's A / ascales / a_ssums).
 * Replace qserve_backend.fused_kernels.invoke_quant(_fuse_sum)      (kernels/csrc/fused.cpp:47-71),
 *         qserve_backend.layernorm_ops.rms_norm(_general(_fuse_sum)) (kernels/csrc/layernorm.cpp:47-72),
 *         qserve_backend.activation_ops.silu_and_mul                 (kernels/csrc/activation.cpp:25-39).
 * hidden % 8 == 0 required.  `input_sum` may be NULL (non-_fuse_sum variants).
 * ---------------------------------------------------------------------------------------------------------- */
int qs_invoke_quant(int8_t* out, const void* input, void* input_sum, void* scale, int num_tokens, int hidden,
                    qs_stream_t stream);
int qs_rms_norm_general(int8_t* out, const void* input, const void* weight, void* input_sum, void* scaling,
                        float epsilon, int num_tokens, int hidden, qs_stream_t stream);
int qs_rms_norm(void* out, const void* input, const void* weight, float epsilon, int num_tokens, int hidden,
                qs_stream_t stream);
int qs_silu_and_mul(void* out, const void* input, int num_tokens, int d, qs_stream_t stream);
/* fp16 residual add (the reference does this with a torch add, llama_w4a8_unpad.py:348,360): a += b */
int qs_residual_add(void* a, const void* b, int64_t numel, qs_stream_t stream);
/* greedy sampling helper (the reference's sampler is torch: argmax over the fp16 logits, layers/sampler.py): out[r] = index
 * of the first maximum of row r of x (fp16 [rows, n], row stride in elements, 16-byte aligned rows; NaN-free). */
int qs_argmax_rows(const void* x, int64_t* out, int rows, int n, int64_t row_stride, qs_stream_t stream);
/* A/B and tests: -1 = heuristic (few long rows are split over up to 8 workgroups per row, candidates meeting in a
 * device-scope atomic max; needs a library-owned 96 KiB scratch that is never allocated during stream capture),
 * 1 = one workgroup per row, >= 2 = forced split. */
void qs_debug_argmax_split(int split);

/* Pair fusions for the decode loop (no reference counterpart; each is BIT-IDENTICAL to the two calls it replaces and
 * exists because at decode batch sizes every one of these row kernels is a fixed ~5 us latency chain):
 *   qs_add_residual_rms_norm_general == qs_residual_add(hidden_io, delta) ; qs_rms_norm_general(out, hidden_io, ...)
 *        (llama_w4a8_unpad.py:348-351 / :360 + next layer's :337 - the torch add followed by the layer norm)
 *   qs_silu_and_mul_quant            == qs_silu_and_mul(tmp, input) ; qs_invoke_quant(out, tmp, ...)
 *        (llama_w4a8_unpad.py:84-91: act_fn then invoke_quant(_fuse_sum)); `input_sum` may be NULL as above. */
int qs_add_residual_rms_norm_general(int8_t* out, void* hidden_io, const void* delta, const void* weight,
                                     void* input_sum, void* scaling, float epsilon, int num_tokens, int hidden,
                                     qs_stream_t stream);
/*   == qs_w4a8_*_gemm(in, kernel, ..., delta) ; qs_add_residual_rms_norm_general(out, hidden_io, delta, ...) where the GEMM ran as
 *   qs_w4a8_*_gemm_planes: delta[t][n] = fp16(epilogue(sum over the k_slices planes)) is formed inside the row kernel.
 *   wscales / w_szs fp16 [hidden] and ascales / a_ssums fp16 [num_tokens] are the GEMM's epilogue operands (w_szs and a_ssums
 *   NULL together = per-group epilogue); ascales / a_ssums MAY alias scaling / input_sum (a row reads its own values first).
 *   plane_stride = elements between planes (>= num_tokens * hidden); hidden <= 4096. */
int qs_add_residual_rms_norm_general_planes(int8_t* out, void* hidden_io, const int32_t* planes, int k_slices,
                                            int64_t plane_stride, const void* wscales, const void* w_szs, const void* ascales,
                                            const void* a_ssums, const void* weight, void* input_sum, void* scaling,
                                            float epsilon, int num_tokens, int hidden, qs_stream_t stream);
int qs_silu_and_mul_quant(int8_t* out, const void* input, void* input_sum, void* scale, int num_tokens, int d,
                          qs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Prefill attention: provider for the call the reference makes into the un-vendored flash-attn wheel,
 *   flash_attn.flash_attn_interface.flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q,
 *   max_seqlen_k, dropout_p=0, softmax_scale=None, causal=True)      (llama_w4a8_unpad.py:30,232-242).
 *   q   half [total_q, H, 128]   token stride q_stride0 elements (views into the packed qkv buffer are fine)
 *   k,v half [total_k, Hkv, 128] token strides k_stride0 / v_stride0
 *   out half [total_q, H, 128]   token stride o_stride0
 *   cu_seqlens_q / cu_seqlens_k int32 [batch+1] (device).  causal: bottom-right aligned when lengths differ (v2.1+).
 * head_dim must be 128; dropout, ALiBi, sliding windows and returning probabilities are not provided.
 * ---------------------------------------------------------------------------------------------------------- */
int qs_flash_attn_varlen_fwd(const void* q, const void* k, const void* v, void* out, const int32_t* cu_seqlens_q,
                             const int32_t* cu_seqlens_k, int batch, int num_heads, int num_kv_heads, int head_dim,
                             int64_t q_stride0, int64_t k_stride0, int64_t v_stride0, int64_t o_stride0,
                             int max_seqlen_q, int max_seqlen_k, float softmax_scale, int causal, qs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Direct-access fp16 sum all-reduce for the tensor-parallel step (no reference counterpart: the reference's
 * tensor parallelism is inert).  Every rank owns one uncached buffer pair (input | output) that all peers map through
 * HIP IPC; one kernel per call: flag exchange, each rank reduces its slice straight out of the peers' inputs (fp32, rank
 * order, one rounding) and writes it straight into everybody's output, flag exchange.  A kernel on the caller's stream
 * like any other: capturable in a hipGraph.  All waits are bounded (qs_comm_error reports a timeout).  UNMEASURED on
 * multi-GPU hardware; torch.distributed / RCCL stays the default of qserve_amd.tp.all_reduce_sum_.
 *   qs_comm_create        allocates this rank's region (current device), returns the communicator and a 64-byte IPC handle
 *   qs_comm_connect       handles = world x 64 bytes in rank order (gathered by the caller, e.g. all_gather_object)
 *   qs_comm_connect_local same-process ranks: the other communicators themselves, no IPC (tests)
 *   qs_comm_input/output  device pointers: the rank's addend is written to input (e.g. as the row-parallel GEMM's output
 *                         buffer), the sum over all ranks is in output once qs_comm_all_reduce_f16's kernel has retired
 *   numel % (8 * world) == 0, numel * 2 <= payload_bytes.  After a time-out the ranks' device-side epochs no longer
 *   match: destroy and re-create the communicators. */
int qs_comm_create(int rank, int world, int64_t payload_bytes, void** comm_out, void* ipc_handle64);
int qs_comm_connect(void* comm, const void* handles);
int qs_comm_connect_local(void* comm, void* const* peer_comms);
void* qs_comm_input(void* comm);
void* qs_comm_output(void* comm);
int qs_comm_all_reduce_f16(void* comm, int64_t numel, qs_stream_t stream);
/* tests: the calls of all ranks of a same-process group (qs_comm_connect_local) as ONE dispatch */
int qs_comm_all_reduce_f16_group(void* const* comms, int world, int64_t numel, qs_stream_t stream);
int qs_comm_error(void* comm);
int qs_comm_destroy(void* comm);

/* ------------------------------------------------------------------------------------------------------------
 * Bounded in-launch waits (no reference counterpart).  Two launches of this library contain a cross-workgroup hand-off that
 * polls inside the launch: the K-slice seam of the decode W4A8 GEMMs and the finisher of qs_single_query_attention_quant.  They
 * rely on in-order workgroup dispatch and on one-stream-at-a-time use of a scratch set (see the header comment).  Every such
 * wait is BOUNDED: after ~1e6 polls (seconds) the waiting wave gives up, sets a bit in the error word of its scratch set and finishes
 * the launch with what it has - results of that launch are invalid, the GPU is not hung.
 *   qs_device_status   error_bits = OR of 1 (K-slice seam gave up), 2 (attention + quant hand-over gave up) on the CURRENT
 *                      device since the last reset; blocking (a device-to-host copy behind the work launched so far) - call it
 *                      at checkpoints, not per launch.  Returns QS_OK when the word could be read; a non-zero word also sets
 *                      qs_last_error().
 *   qs_device_reset    synchronises the device, clears the word and puts the hand-off scratch back into its initial state
 *                      (K-slice slabs sentinel-filled, exchange rows / generation words zeroed).  Required after a non-zero
 *                      status and after an aborted launch before the library is used again; replayed hipGraphs stay valid
 *                      (no address changes).
 *   qs_debug_inject_fault  tests only: arms a ONE-SHOT fault - bit 0: the next K-sliced ring GEMM launch, bit 1: the next
 *                      fused attention + quant launch runs with one producer that never delivers (and a short poll bound), so
 *                      that the give-up path, the status word and the recovery can be exercised.  The results of THAT launch
 *                      are wrong by design; 0 disarms. */
int qs_device_status(int* error_bits);
int qs_device_reset(void);
int qs_debug_inject_fault(int what);

/* Per-stream scratch (no reference counterpart; the reference's engine is single-stream).  By default every stream of a device
 * shares ONE set of the library's scratch areas, so scratch-using launches must not overlap across streams.  A caller that runs
 * the library on several streams of one device concurrently binds the extra streams first:
 *   qs_stream_scratch_bind    gives `stream` (on the calling thread's current device) its own split-K / K-slice slabs, split-KV
 *                             partials, hand-over rows and argmax keys (~130 MiB, allocated HERE - call it outside stream capture,
 *                             before the stream's first launch or capture; idempotent).  Launches issued or captured on the stream
 *                             use these areas from then on; qs_device_status / qs_device_reset cover every bound stream.
 *                             QS_ENOSUP when the device's 7 slots are taken, QS_EINVAL while the stream is capturing.
 *   qs_stream_scratch_unbind  forgets the binding (the stream falls back to the shared areas); the slot's memory is kept and
 *                             handed to the next bind on this device, so graphs captured on the stream stay valid - but they
 *                             then share the areas with that next stream. */
int qs_stream_scratch_bind(qs_stream_t stream);
int qs_stream_scratch_unbind(qs_stream_t stream);

/* Device self-test (tests/test_fused_gpu.py): the DPP / permlane wave reductions every row kernel uses round exactly like
 * the shuffle butterfly they replace.  in: float [n] (n % 64 == 0); out: float [n/64][4] = {sum, sum by shuffles, max, max
 * by shuffles} per 64-value block. */
int qs_debug_wave_reduce_selftest(const float* in, float* out, int n, qs_stream_t stream);

/* A/B hook of the prefill attention provider (process-wide, see qs_set_gemm_variant): 0 [default] = the round-6 kernel (Q fragments
 * complete before the key loop, no accumulator copies in it, lazy running maximum, whole-row output through LDS); 1 = the
 * kernel of rounds 2-5.  Both compute the same softmax within the provider's tolerance (tests/test_flash_gpu.py runs both).
 * QS_EINVAL for any other value. */
int qs_debug_flash_variant(int variant);

/* Timing tool (scripts/trace_attn.py): device-to-device copy of the first `bytes` of the split-KV workspace, where the
 * trace instantiation of the KV4 decode attention (qs_set_attention_variant(232)) leaves its s_memtime stamps.
 * (A library built with -DQS_RING_TRACE additionally exports qs_debug_ring_trace(void* buf) for scripts/trace_gemm.py;
 * it is not part of the shipped ABI.) */
int qs_debug_copy_split_workspace(void* dst, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* QSERVE_AMD_H */
