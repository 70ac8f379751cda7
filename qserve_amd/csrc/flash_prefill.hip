// flash_prefill.hip -- prefill attention (causal / full, variable-length, GQA, fp16, head_dim 128) for MI355X (gfx950).
//
// Provider for the call the reference makes into the un-vendored flash-attn package
//   flash_attn.flash_attn_interface.flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q,
//   max_seqlen_k, causal=True)            (qserve/modeling/models/llama_w4a8_unpad.py:30,232-242)
// flash-attn v2 semantics: softmax(scale * Q K^T) V per sequence and head, fp32 softmax, causal mask aligned to the
// bottom-right corner when the query and key lengths differ.  The algorithm is the published FlashAttention-2 forward
// (online softmax over key tiles, no S x S matrix); the mapping below is written for CDNA4:
//   * workgroup = 4 wave64 = 128 query rows of one (sequence, head); wave w owns 32 rows and keeps their Q fragments
//     (8 x 16 dims) in registers for the whole key loop;
//   * key/value tiles of 64 keys are staged through LDS once per workgroup (shared by the 4 waves; the G query heads
//     of a GQA group are separate workgroups that re-read the tiles from L2): K row-major with a 16-byte XOR swizzle,
//     V row-major too (16-byte chunk ^ 4 (key & 3): the 32 lanes of a transpose read then hit 32 distinct 8-byte slots;
//     chunk ^ 2 (key & 3), the round-1 form, left them two-way conflicted - SQ_LDS_BANK_CONFLICT was a third of the LDS cycles) and TRANSPOSED ON READ by ds_read_b64_tr_b16, both tiles
//     double-buffered, one barrier per tile;
//   * "swapped" products on v_mfma_f32_32x32x16_f16:  S^T = K Q^T  (A = K rows, B = Q rows, both plain 16-byte reads)
//     leaves every lane holding 16 of the 32 scores of ITS OWN query row, so the softmax is register-only (one
//     cross-lane max with lane ^ 32) and the probabilities already sit in B-operand order for
//     O^T = V^T P^T  (A = V^T fragments, two transpose reads each; k-slot <-> key mapping chosen to match the S^T layout);
//   * exp2 with the scale folded into one fp32 multiply; rescaling of O only through the running max.
#include "common.h"
#include <type_traits>
#include <utility>

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef u32 v2u __attribute__((ext_vector_type(2)));

#ifndef QS_FLASH_DBG
#define QS_FLASH_DBG 0            // timing experiments (scripts/bench_flash.py; results wrong): 1 no exp2, 2 no P.V, 4 no Q.K^T, 8 no tile loads, 16 no key loop, 32 no output stores, 64 no Q loads
#endif
constexpr int DH = 128;
#ifndef QS_FLASH_NW
#define QS_FLASH_NW 4
#endif
constexpr int NWV = QS_FLASH_NW;  // waves per workgroup (32 query rows each)
constexpr int BM = 32 * NWV;      // query rows per workgroup
constexpr int PPW = 16 / NWV;     // 1 KiB DMA pieces of a K (and of a V) tile per wave
#ifndef QS_FLASH_NKB
#define QS_FLASH_NKB 2
#endif
#ifndef QS_FLASH_OCC
#define QS_FLASH_OCC 2
#endif
constexpr int NKB = QS_FLASH_NKB;  // 32-key blocks per tile
constexpr int BN = 32 * NKB;      // keys per tile
constexpr int KS_BYTES = BN * DH * 2;             // 16 KiB
constexpr int VT_BYTES = BN * DH * 2;             // 16 KiB (row-major like K; transposed on read)

__device__ __forceinline__ u32 pack_h2(float a, float b) {
    const h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(u32, v);
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}
#ifdef QS_FLASH_TRACE
// timing builds only (scripts/trace_flash.py): cycles per phase of the key loop, summed per wave
__device__ unsigned long long* g_flash_trace = nullptr;
#define QS_FT(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ft[i] += (unsigned)(n_ - ft_t); ft_t = n_; } while (0)
#else
#define QS_FT(i) do { } while (0)
#endif

// VAR = 1 (round 6, the default) against VAR = 0 (the kernel of rounds 2-5, kept for the A/B: qs_debug_flash_variant(1)):
//  (1) the Q fragments are complete FOR THE COMPILER before the key loop: it had kept them "load pending" and put counted vmcnt
//      waits in front of the first Q.K^T MFMAs of every tile - waits that also drain the LDS-DMA of tile t + 1, issued a few
//      instructions earlier by asm it does not see.  Every wave sat through the L2 round trip of its own prefetch, every tile;
//  (2) no register copies of the O accumulators in the key loop: 32 v_mov_b64 per tile and wave came from two control-flow merges
//      the 64 accumulators were carried through - the rescale branch (now marked unlikely: its copies live on the cold path) and the
//      skip of fully masked causal tiles (now a loop of its own behind the computing loop).  Pinning O to fixed registers through
//      asm (physical-register constraints) and asm-owned accumulator registers were tried first: the former spills 30 registers in
//      the loop, the latter makes the compiler split the wave's 256 registers 128 / 128 and move the score tiles into AGPRs;
//  (3) LAZY running maximum - a row's reference maximum moves only when a tile exceeds it by more than 2^8 (probabilities stay
//      <= 256 in fp16, sums in fp32: the same softmax), so the rescale, which ran on ~85 % of the tiles of a 1 024-token prompt,
//      becomes rare;
//  (4) the tile loop unrolled by two with the LDS buffer index a compile-time constant (every ds_read address a loop-invariant
//      register + an immediate offset; the loop carried ~47 v_add_u32 of address arithmetic per tile and wave);
//  (5) O leaves through LDS as whole rows (see the epilogue).
// In-run A/B (profiles/round6_flash_ab7.txt): 64 x 1 024 tokens 0.233 -> 0.277 of the dense fp16 MFMA peak, 4 x 8 192: 0.345 -> 0.397.
template <bool CAUSAL, int VAR>
__global__ __launch_bounds__(64 * NWV, QS_FLASH_OCC) void flash_fwd_kernel(const _Float16* __restrict__ q, const _Float16* __restrict__ k,
                                                          const _Float16* __restrict__ v, _Float16* __restrict__ out,
                                                          const int* __restrict__ cu_q, const int* __restrict__ cu_k,
                                                          int num_heads, int num_kv_heads, int64_t q_stride0,
                                                          int64_t k_stride0, int64_t v_stride0, int64_t o_stride0,
                                                          float scale_log2) {
    constexpr bool R6 = VAR != 0;      // VAR: 0 = the kernel of rounds 2-5 (A/B: qs_debug_flash_variant(1)), 1 = round 6 (default)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t (*s_k)[KS_BYTES] = reinterpret_cast<uint8_t (*)[KS_BYTES]>(smem);                    // [2][16 KiB]
    uint8_t (*s_vt)[VT_BYTES] = reinterpret_cast<uint8_t (*)[VT_BYTES]>(smem + 2 * KS_BYTES);    // [2][16 KiB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // grid = (heads, query tiles, sequences).  All workgroups of ONE sequence are dispatched next to each other (its K / V -
    // 0.5 MB per KV head at 1024 tokens - are fetched from HBM once and re-served by L2 / the Infinity Cache to its
    // query tiles and heads), inside a sequence the query tiles run last-to-first for causal launches (the workgroups that
    // see the most keys start first: longest-first keeps the tail short), and the head index is permuted so that
    // workgroups 8 apart - which share an XCD and its L2 - are the G heads of one KV group.
    // (measured at 64 x 1024 / 4 x 8192 tokens: query tile fastest 417 / 740 TFLOP/s, query tile slowest 449 / 822, this
    //  order 508 / 858)
    const int b = blockIdx.z, qt = CAUSAL ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
    const int h = (int)(blockIdx.x % num_kv_heads) * (num_heads / num_kv_heads) + (int)(blockIdx.x / num_kv_heads);
    const int q_start = cu_q[b], len_q = cu_q[b + 1] - q_start;
    const int k_start = cu_k[b], len_k = cu_k[b + 1] - k_start;
    if (qt * BM >= len_q) return;
    const int hkv = h / (num_heads / num_kv_heads);
    const int shift = len_k - len_q;                   // causal diagonal: key <= row + shift
    const int li = lane & 31, hi = lane >> 5;
    const int row = qt * BM + wave * 32 + li;          // this lane's query row (both lane halves share it)
    const int row_ld = row < len_q ? row : len_q - 1;

    // ---- Q fragments: B operand of S^T = K Q^T, lane (row, hi) holds dims 16s + 8hi .. +8 ----------------------------
    h8 qf[8];
    {
        const _Float16* qp = q + (size_t)(q_start + row_ld) * q_stride0 + (size_t)h * DH + 8 * hi;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (QS_FLASH_DBG & 64) qf[s] = (h8){(_Float16)(0.01f * lane), 1, 2, 3, 4, 5, 6, (_Float16)s};
            else qf[s] = *reinterpret_cast<const h8*>(qp + 16 * s);
        }
    }

    // ---- key range ----------------------------------------------------------------------------------------------------
    int kv_end = len_k;
    if (CAUSAL) {
        const int last = qt * BM + BM - 1 + shift;     // largest key any row of this workgroup may see
        kv_end = last + 1 < len_k ? last + 1 : len_k;
    }
    const int ntiles = (QS_FLASH_DBG & 16) ? 0 : kv_end > 0 ? (kv_end + BN - 1) / BN : 0;   // (DBG 16: prologue + epilogue only)

    // ---- tile staging by LDS-DMA: a 1 KiB piece = 4 keys x 256 B; wave w copies K pieces 4w .. 4w+3 and the same V pieces.
    // The DMA writes lane-linear (lane l -> key l >> 4 of the piece, 16-byte position l & 15), so the XOR swizzles of the
    // images are applied to the per-lane SOURCE chunk.  No staging registers, no ds_write pass; keys beyond the sequence
    // are clamped to its last row (finite data; their scores are masked, their probabilities are 0).
    const _Float16* kg = k + (size_t)k_start * k_stride0 + (size_t)hkv * DH;
    const _Float16* vg = v + (size_t)k_start * v_stride0 + (size_t)hkv * DH;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const u32 lds_k = (u32)(size_t)(lptr_t)smem, lds_v = lds_k + 2 * KS_BYTES;
    // scalar base (advances by one tile) + per-lane 32-bit byte offset (constant): no per-lane 64-bit arithmetic per piece
    auto dma16 = [&](u32 voff, const void* sbase, u32 lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    static_assert(NKB == 2, "the DMA staging below is written for 64-key tiles");
    // Per-lane source offsets: ONE register each for K and V (round 6; were PPW each).  Piece i of a wave covers keys
    // 4 (PPW wave + i) + (lane >> 4): the rows of piece i are 4 i keys further on - a wave-uniform distance that goes into the scalar
    // base -, the V swizzle depends on key & 3 = (lane >> 4) & 3 only, and the K swizzle pos ^ (key & 15) differs between the pieces
    // by an XOR with 4 i on the 16-byte position (PPW = 4: key & 15 = 4 i + (lane >> 4)), i.e. by `^ 64 i` on the byte offset.
    static_assert(PPW == 4, "the per-piece offsets below are derived for four pieces per wave");
    const int l4 = lane >> 4, pos = lane & 15;
    const u32 koff0 = (u32)l4 * (u32)k_stride0 * 2u + (u32)((pos ^ l4) * 16);
    const u32 voff0 = (u32)l4 * (u32)v_stride0 * 2u + (u32)((pos ^ (l4 << 2)) * 16);
    auto load_tile = [&](int t, int buf) {
        const _Float16* kb_ = kg + ((size_t)t * BN + 4 * PPW * wave) * k_stride0;   // first key of this wave's pieces
        const _Float16* vb_ = vg + ((size_t)t * BN + 4 * PPW * wave) * v_stride0;
        const bool ragged = t * BN + BN > len_k;        // wave-uniform: only the last tile of a sequence
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (ragged) {                                 // clamp the row to the sequence's last key (offsets from the TILE's base)
                const int key = 4 * (PPW * wave + i) + l4;
                int kc = len_k - 1 - t * BN;
                kc = key < kc ? key : kc;
                const u32 ko = (u32)kc * (u32)k_stride0 * 2u + (u32)((pos ^ (key & 15)) * 16);
                const u32 vo = (u32)kc * (u32)v_stride0 * 2u + (u32)((pos ^ ((key & 3) << 2)) * 16);
                dma16(ko, kg + (size_t)t * BN * k_stride0, lds_k + buf * KS_BYTES + (PPW * wave + i) * 1024);
                dma16(vo, vg + (size_t)t * BN * v_stride0, lds_v + buf * VT_BYTES + (PPW * wave + i) * 1024);
            } else {
                // (the XOR is re-done per tile by an opaque statement: hoisted out of the loop - as the compiler does with the plain
                //  expression - the three extra offsets are exactly what it spills to scratch once O is pinned)
                u32 ko = koff0;
                if (i > 0) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ko) : "n"(64 * i), "v"(koff0));
                dma16(ko, kb_ + (size_t)(4 * i) * k_stride0, lds_k + buf * KS_BYTES + (PPW * wave + i) * 1024);
                dma16(voff0, vb_ + (size_t)(4 * i) * v_stride0, lds_v + buf * VT_BYTES + (PPW * wave + i) * 1024);
            }
        }
    };
    auto tiles_landed = [&]() {                       // every wave's pieces: own queue drained, then the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    v16f oacc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    if (ntiles > 0) load_tile(0, 0);
    // The Q fragments must be COMPLETE FOR THE COMPILER before the key loop (round 6).  They come from ordinary global loads; the
    // only wait in front of the loop was the asm vmcnt(0) of tiles_landed(), which the compiler's wait-count pass does not see - so
    // it kept the fragments "load pending" into the loop and put counted waits in front of the first Q.K^T MFMAs of EVERY tile
    // (s_waitcnt vmcnt(7) ... vmcnt(0), one per fragment).  The hardware counter it waits on also counts the LDS-DMA of tile
    // t + 1, issued a few instructions earlier by asm the compiler does not see either: every wave sat through the L2 round trip
    // of its own prefetch inside Q.K^T of every tile (the "1 430 cycles for DMA issue + Q.K^T" of the round-2 trace, 512 of them
    // MFMA).  An empty asm that rewrites the fragments makes the compiler wait HERE, once (tests/test_kernel_contracts.py pins:
    // no vmcnt wait between the loop's barriers other than the explicit one of tiles_landed()).
    if constexpr (R6) {
#pragma unroll
        for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(qf[s]));
    }
    tiles_landed();
#ifdef QS_FLASH_STAGGER
    // experiment: the two workgroups of a CU run identical code with identical timing and can settle in lockstep (both in
    // their MFMA phase, then both in their VALU phase); delay every other workgroup by about half a tile
    if ((blockIdx.x ^ blockIdx.y ^ blockIdx.z) & 1)
        for (int i = 0; i < QS_FLASH_STAGGER; ++i) __builtin_amdgcn_s_sleep(16);
#endif

#ifdef QS_FLASH_TRACE
    unsigned ft[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ft_t = __builtin_amdgcn_s_memtime();
#endif
    // one tile; BUFC = std::integral_constant<int, 0 / 1> (R6: the buffer index is a compile-time constant) or a run-time int
    auto tile_body = [&](auto bufc, int t) {
        const int buf = bufc;
        QS_FT(0);
        // (round 6, measured and dropped: the pieces of tile t + 1 requested LATER - K behind Q.K^T, V behind the exponentials, where an
        //  LDS-DMA instruction should cost the wave fewer issue cycles than next to 16 ds_read_b128: equal at 64 x 1 024 tokens, -0.8 %
        //  at 4 x 8 192, profiles/round6_flash_ab6.txt)
        if (!(QS_FLASH_DBG & 8) && t + 1 < ntiles) load_tile(t + 1, buf ^ 1);   // lands in the other buffers during this tile
        // causal: the workgroup's key range ends at its LAST row's diagonal; a wave whose 32 rows all lie before this tile
        // has nothing to add (every score masked) - it only takes part in the staging and the barrier
        // (R6: such tiles never reach this body - see the loops below)
        if (!R6 && CAUSAL && t * BN > qt * BM + wave * 32 + 31 + shift) {
            tiles_landed();
            return;
        }

        // ---------------- S^T = K Q^T : two blocks of 32 keys ----------------
        // operand reads run one group of 4 MFMAs ahead of the matrix pipe (two register sets): issued as written, they
        // leave the compiler no choice but counted lgkmcnt waits - with read-then-use in one loop body every MFMA sat behind
        // a full LDS round trip
        v16f sacc[NKB];
        if (!(QS_FLASH_DBG & 4)) {
            h8 ka[2][4];
            auto read_k = [&](int g, h8 (&dst)[4]) {       // group g = (kb = g >> 1, s = 4 (g & 1) .. +3)
                const int key = 32 * (g >> 1) + li;
                const uint8_t* krow = &s_k[buf][key * 256];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sl = 4 * (g & 1) + j;
                    dst[j] = *reinterpret_cast<const h8*>(krow + (((2 * sl + hi) ^ (key & 15)) * 16));
                }
            };
            read_k(0, ka[0]);
            read_k(1, ka[1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // the first MFMA of a 32-key block starts from C = 0 (an inline constant operand: no 16-register zero
                    // fill per block and tile)
                    const v16f zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    const bool first = (g & 1) == 0 && j == 0;
                    sacc[g >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[g & 1][j], qf[4 * (g & 1) + j], first ? zero16 : sacc[g >> 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < 4) read_k(g + 2, ka[g & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
            sacc[0][0] = (float)qf[0][0];
        }
        // A operands of O^T += V^T P^T by the LDS transpose read: a 16-lane group (16 consecutive dims, one lane half) reads
        // the [4 keys][16 dims] block of the row-major tile - lane a supplies the 8-byte piece (key a>>2, dims 4(a&3)..+3) -
        // and lane c receives column c = (dim c, keys 0..3), i.e. exactly its four k-slots of the PV MFMA.  Group d = the
        // four MFMAs of output dims 32d .. 32d+31; group 0 is requested here, under the softmax.
        const int ta = lane & 15, g1 = (lane >> 4) & 1;
        const int tkey = 4 * hi + (ta >> 2);                       // key within a 16-key block; tkey & 3 == ta >> 2
        h8 va[2][4];
        auto read_v = [&](int d, h8 (&dst)[4]) {
            const int chunk = (4 * d + 2 * g1 + ((ta & 3) >> 1)) ^ ((ta >> 2) << 2);   // V image swizzle: chunk ^ 4 (key & 3)
            const uint8_t* vrow = &s_vt[buf][tkey * 256 + chunk * 16 + (ta & 1) * 8];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int kofs = (32 * kb + 16 * m) * 256;
                    typedef short s4 __attribute__((ext_vector_type(4)));
                    typedef __attribute__((address_space(3))) s4* lds_s4;
                    const s4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(vrow + kofs));             // keys +0..3
                    const s4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(vrow + kofs + 8 * 256));   // keys +8..11
                    const v2u lo = __builtin_bit_cast(v2u, t0), hi2 = __builtin_bit_cast(v2u, t1);
                    dst[2 * kb + m] = __builtin_bit_cast(h8, (v4u){lo.x, lo.y, hi2.x, hi2.y});
                }
        };
        QS_FT(1);                                                  // DMA issue + Q.K^T
        if (!(QS_FLASH_DBG & 2)) read_v(0, va[0]);
        __builtin_amdgcn_sched_barrier(0);
        // sacc[kb][r] = score of (this lane's row, key t*64 + 32kb + (r&3) + 8(r>>2) + 4hi)
        // masking only where the tile touches the diagonal or the end of the keys (wave-uniform test); raw scores
        // stay unscaled, the scale is folded into the exponent's fma
        const int tile_last = t * BN + BN - 1;
        const bool need_mask = tile_last >= len_k || (CAUSAL && tile_last > qt * BM + wave * 32 + shift);
        float mx = -INFINITY;
        if (need_mask) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * BN + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool ok = key < len_k && (!CAUSAL || key <= row + shift);
                    const float sv = ok ? sacc[kb][r] : -INFINITY;
                    sacc[kb][r] = sv;
                    mx = fmaxf(mx, sv);
                }
        } else {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2;      // scale > 0: max commutes with it
        // R6: lazy reference maximum (see the kernel's header)
        const float m_new = R6 ? (mx > m_run + 8.0f ? mx : m_run) : fmaxf(m_run, mx);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;     // fully masked so far: keep exp2 arguments finite
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);                  // m_run = -inf -> 0
        m_run = m_new;
        // two scores per instruction where the ISA has a packed form (v_pk_fma_f32, v_pk_add_f32; the exponential has none):
        // round 5, the key loop is as VALU-bound as it is MFMA-bound (HISTORY 5.5)
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f sc2 = {scale_log2, scale_log2}, nm2 = {-m_use, -m_use};
        v2f psum2 = {0.f, 0.f};
        u32 pb[NKB][2][4];                                           // [kb][m]: 8 probabilities in B-operand order
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v2f sv = {sacc[kb][8 * m + 2 * j], sacc[kb][8 * m + 2 * j + 1]};
                    v2f pp = __builtin_elementwise_fma(sv, sc2, nm2);
                    if (!(QS_FLASH_DBG & 1)) {
                        pp[0] = __builtin_amdgcn_exp2f(pp[0]);      // -inf stays -inf
                        pp[1] = __builtin_amdgcn_exp2f(pp[1]);
                    }
                    psum2 += pp;
                    pb[kb][m][j] = pack_h2(pp[0], pp[1]);
                }
        const float psum = psum2[0] + psum2[1];
        l_run = l_run * alpha + psum;
        // (R6: marked unlikely - with the lazy maximum it is the first tile and jumps of more than 2^8.  The not-taken path must not
        //  carry register copies of the 64 accumulators: see the loop structure below)
        if (R6 ? __builtin_expect(__any(alpha != 1.0f), 0) : __any(alpha != 1.0f)) {   // the running max moved for some row of this wave
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }

        QS_FT(2);                                                  // mask + softmax + O rescale
        // ---------------- O^T += V^T P^T ----------------
        if (QS_FLASH_DBG & 2) {
            oacc[0][0] += __builtin_bit_cast(float, pb[0][0][0] ^ pb[NKB - 1][1][3]);
        } else {
            read_v(1, va[1]);
            __builtin_amdgcn_sched_barrier(0);
            static_for<4>([&](auto dc) {
                constexpr int d = decltype(dc)::value;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const h8 pbv = __builtin_bit_cast(h8, (v4u){pb[kb][m][0], pb[kb][m][1], pb[kb][m][2], pb[kb][m][3]});
                        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[d & 1][2 * kb + m], pbv, oacc[d], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (d + 2 < 4) read_v(d + 2, va[d & 1]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        QS_FT(3);                                                  // P.V
        tiles_landed();
        QS_FT(4);                                                  // wait for the next tile + barrier
    };
    if constexpr (R6) {
        // tiles this WAVE computes: a causal tile whose first key lies beyond the wave's last row has nothing to add - for those
        // the wave only takes part in the staging and the barrier (second loop).  Kept out of the first loop on purpose: a skip
        // path that rejoins the computing path inside the loop is a control-flow merge the 64 O accumulators are carried
        // through, and the compiler resolves such merges with register copies.
        int nt_w = ntiles;
        if (CAUSAL) {
            const int last_key = qt * BM + wave * 32 + 31 + shift;
            nt_w = last_key < 0 ? 0 : min(ntiles, last_key / BN + 1);
        }
        int t = 0;
        for (; t + 1 < nt_w; t += 2) {
            tile_body(std::integral_constant<int, 0>(), t);
            tile_body(std::integral_constant<int, 1>(), t + 1);
        }
        if (t < nt_w) {
            tile_body(std::integral_constant<int, 0>(), t);
            ++t;
        }
        for (; t < ntiles; ++t) {
            if (!(QS_FLASH_DBG & 8) && t + 1 < ntiles) load_tile(t + 1, (t + 1) & 1);
            tiles_landed();
        }
    } else {
        for (int t = 0; t < ntiles; ++t) tile_body(t & 1, t);
    }
#ifdef QS_FLASH_TRACE
    if (g_flash_trace && lane == 0) {
        unsigned long long* o = g_flash_trace + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NWV + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = ft[i];
        o[5] = ntiles;
    }
#endif

    // ---- epilogue: normalise, fp16, 8-byte stores (4 consecutive dims per accumulator quad) ---------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    // (the row index is re-derived from a lane id the compiler cannot merge with the one above: kept alive across the key loop
    //  it costs a register the causal instantiation does not have)
    const int row_e = qt * BM + wave * 32 + (int)(fresh_lane_id() & 31u);
    // whole-row stores need 16-byte alignment of every row (wave-uniform)
    const bool rows16 = R6 && (o_stride0 % 8 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (rows16) {
        // Round 6: O goes through LDS and leaves as WHOLE ROWS.  In the accumulator layout a lane owns one query row, so the direct
        // form below is 16 stores of 8 bytes per lane at a row stride (o_stride0, 8 KiB for 32 heads): every store instruction touches
        // 64 different cache lines, and the store tail of a workgroup lasts ~2 key tiles (MI355X_MICROARCH.md: "attention epilogue
        // store tail ... store-ISSUE-bound").  Here a wave writes its 32 x 128 fp16 block into its own 8.5 KiB of the (dead) K / V
        // buffers - row stride 272 B: the rows of a half-wave fall on banks 4 li, two-way conflicts at most - and reads it back 16
        // bytes per lane, 16 lanes per row: 8 stores per lane, each instruction 4 complete 256-byte rows.  No barrier: every wave has
        // passed the last tile's barrier (all reads of the buffers are over) and touches only its own block; the LDS serves a wave's
        // accesses in order.
        constexpr int OST = 272;
        uint8_t* const so = smem + wave * (32 * OST);
        const int li_e = (int)(fresh_lane_id() & 31u), hi_e = (int)(fresh_lane_id() >> 5);
        static_for<4>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            static_for<4>([&](auto rc) {
                constexpr int rq = decltype(rc)::value;
                h4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (_Float16)(oacc[d][4 * rq + j] * inv);
                *reinterpret_cast<h4*>(so + li_e * OST + (32 * d + 8 * rq + 4 * hi_e) * 2) = o;
            });
        });
        const int lid = (int)fresh_lane_id(), rr = lid >> 4, cc = lid & 15;
        const int row0 = qt * BM + wave * 32;
        _Float16* const ob = out + (size_t)(q_start + row0) * o_stride0 + (size_t)h * DH + cc * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 4 * j + rr;
            const v4u x = *reinterpret_cast<const v4u*>(so + r * OST + cc * 16);
            if (row0 + r < len_q && (!(QS_FLASH_DBG & 32) || x.x == 0x12345678u)) *reinterpret_cast<v4u*>(ob + (size_t)r * o_stride0) = x;
        }
    } else if (row_e < len_q) {
        _Float16* op = out + (size_t)(q_start + row_e) * o_stride0 + (size_t)h * DH;
        static_for<4>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            static_for<4>([&](auto rc) {
                constexpr int rq = decltype(rc)::value;
                h4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (_Float16)(oacc[d][4 * rq + j] * inv);
                *reinterpret_cast<h4*>(op + 32 * d + 8 * rq + 4 * hi) = o;
            });
        });
    }
}

}  // namespace

#ifdef QS_FLASH_TRACE
extern "C" int qs_debug_flash_trace(void* buf) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_flash_trace), &p, sizeof(p));
}
#endif

static qs_flag g_flash_variant = 0;
// A/B hook (include/qserve_amd.h): 0 = lazy running maximum + tile loop unrolled over the two LDS buffers (round 6, default),
// 1 = the loop of rounds 2-5.  The same softmax; the reference maximum differs, so low-order bits may.
extern "C" int qs_debug_flash_variant(int variant) {
    QS_REQUIRE(variant == 0 || variant == 1, "qs_debug_flash_variant: %d not in {0, 1}", variant);
    g_flash_variant = variant;
    return QS_OK;
}

extern "C" int qs_flash_attn_varlen_fwd(const void* q, const void* k, const void* v, void* out,
                                        const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, int batch,
                                        int num_heads, int num_kv_heads, int head_dim, int64_t q_stride0,
                                        int64_t k_stride0, int64_t v_stride0, int64_t o_stride0, int max_seqlen_q,
                                        int max_seqlen_k, float softmax_scale, int causal, qs_stream_t stream) {
    QS_REQUIRE(q && k && v && out && cu_seqlens_q && cu_seqlens_k, "flash_attn_varlen: null pointer");
    QS_REQUIRE(batch >= 0 && num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0,
               "flash_attn_varlen: bad head counts H=%d Hkv=%d", num_heads, num_kv_heads);
    if (head_dim != DH) {
        qs_set_error("flash_attn_varlen: head_dim=%d, only 128 is supported (the reference's models)", head_dim);
        return QS_ENOSUP;
    }
    QS_REQUIRE(q_stride0 % 8 == 0 && k_stride0 % 8 == 0 && v_stride0 % 8 == 0 && o_stride0 % 4 == 0,
               "flash_attn_varlen: token strides must keep 16-byte alignment");
    QS_REQUIRE(max_seqlen_q >= 0 && max_seqlen_k >= 0, "flash_attn_varlen: negative max_seqlen");
    QS_REQUIRE(softmax_scale > 0.f, "flash_attn_varlen: softmax_scale must be positive");
    if (batch == 0 || max_seqlen_q == 0) return QS_OK;
    const float scale_log2 = softmax_scale * 1.4426950408889634f;
    dim3 grid(num_heads, (max_seqlen_q + BM - 1) / BM, batch);
#ifndef QS_FLASH_LDSPAD
#define QS_FLASH_LDSPAD 0
#endif
    constexpr int SMEM = 2 * KS_BYTES + 2 * VT_BYTES + QS_FLASH_LDSPAD;   // (pad: occupancy experiments)
    static bool configured_dev[QS_MAX_DEVICES] = {};   // the attribute belongs to the (kernel, device) pair
    bool& configured = configured_dev[qs_device_slot()];
    if (!configured) {
        hipError_t e1 = hipSuccess;
        for (const void* fn : {reinterpret_cast<const void*>(flash_fwd_kernel<true, 0>), reinterpret_cast<const void*>(flash_fwd_kernel<true, 1>),
                               reinterpret_cast<const void*>(flash_fwd_kernel<false, 0>), reinterpret_cast<const void*>(flash_fwd_kernel<false, 1>)}) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
            if (e != hipSuccess) e1 = e;
        }
        if (e1 != hipSuccess) {
            qs_set_error("flash_attn_varlen: cannot reserve %d bytes of LDS", SMEM);
            return (int)e1;
        }
        configured = true;
    }
#define QS_FL(C, P)                                                                                                    \
    hipLaunchKernelGGL((flash_fwd_kernel<C, P>), grid, dim3(64 * NWV), SMEM, (hipStream_t)stream, (const _Float16*)q,   \
                       (const _Float16*)k, (const _Float16*)v, (_Float16*)out, cu_seqlens_q, cu_seqlens_k, num_heads,  \
                       num_kv_heads, q_stride0, k_stride0, v_stride0, o_stride0, scale_log2)
    if (causal) {
        if (g_flash_variant == 0) QS_FL(true, 1);
        else QS_FL(true, 0);
    } else {
        if (g_flash_variant == 0) QS_FL(false, 1);
        else QS_FL(false, 0);
    }
#undef QS_FL
    return qs_launch_status("flash_attn_varlen");
}
