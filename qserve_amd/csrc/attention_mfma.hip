// attention_mfma.hip -- KV4 decode attention on the matrix cores (gfx950), wave-autonomous flash-decoding.
//
// Same contract as decode_attention_kernel in attention.hip (reference: fused_attention.cpp:150-240,
// decoderMaskedMultiheadAttentionTemplate.hpp:717-2222, ZINT4 variant); different mapping:
//
//   * one workgroup (NW = 8 wave64) per (sequence, KV head).  The cache is consumed in PIPELINE UNITS of 32 tokens (half a
//     64-token page): unit j of the workgroup's range belongs to wave j % PS (PS = 7 or 8 page-owning waves); every wave
//     runs QK -> online softmax -> PV on its units WITHOUT any workgroup barrier; the partial (max, sum, out) triples are
//     merged once at the end through LDS (flash-decoding inside the workgroup).  All G = H/Hkv query heads of the group
//     are served from one read of the unit (the reference re-reads it per query head).
//     Round 4: units instead of whole pages.  With pages a wave's first request was 8.5 KiB, the first Q.K^T of the
//     kernel started when 40 % of the launch was over (the first round = half of all bytes had to land first) and 17
//     pages over 7 waves meant 3 page rounds for a mean of 2.43; with units a wave's first request is 4.25 KiB, compute
//     starts with the first quarter of the bytes, and 33 units over 7 waves are 5 unit rounds = 2.5 page rounds.
//   * a unit (2 KiB K + 2 KiB V + 4 x 64 B scales / zeros) travels HBM -> LDS by LDS-DMA (global_load_lds, 16 B per
//     lane, lane-linear = fully coalesced) into wave-private two-slot rings: no staging registers, completion tracked
//     with counted s_waitcnt vmcnt.  Request group A = (K meta, K data x 2), group B = (V meta, V data x 2); A(i+2) is
//     requested when Q.K^T of unit i is done, B(i+2) when P.V is done.
//   * Q.K^T on v_mfma_f32_16x16x32_f16: A = 16 tokens x 32 dims of the K unit: lane (tok, kg) turns the 8 nibbles of a
//     dword into fp16 with the magic-number trick and feeds them IN OFFSET FORM (1024+n, 1024+16n) - the offsets are
//     removed from the 16x16 result with a per-head constant, the per-token scale / zero point likewise:
//         score = ksc[tok] * (c_raw - Qoff[head] - kzr[tok] * qsum[head]) / sqrt(128)
//     so the inner loop spends 5 VALU ops per 8 cache elements instead of 13.
//   * P.V on the same instruction: A = V^T: lane (8-dim group, kg) reads one dword per token for its 8 tokens;
//     v_perm_b32 pairs byte bb of two tokens (0x00BB00AA) and the two nibble masks give the fp16 pairs of dims 2bb and
//     2bb+1 (offset form 1024 + n / 1024 + 16 n), i.e. the 8x8 transposition costs one perm per 4 elements.  B = P'^T with
//     P' = fp16(p * vsc[tok]); the zero-point term sum_t P'_t vzr_t is subtracted from every output dim at the end.
//   * softmax on compacted lanes: only G of the 16 result columns are heads, so the 2 score tiles of a unit are moved
//     (DPP row_shr, bank-masked) onto the idle lanes of their 16-lane row: G <= 4 - lane li = h + 4 (2 t + rp) owns the
//     scores r = 2 rp, 2 rp + 1 of tile t and head h (2 scores per lane, every lane busy for G = 4); G = 5..8 - lane
//     li = h + 8 t owns the 4 scores of tile t.  The scale / zero-point / exp2 / P' work shrinks accordingly and the P'
//     operands return with row_shl moves; scores live in the log2 domain; V operands stay in offset form like K and the
//     offsets leave through sum(P') at the end.
//   * cache integers are exact in fp16 and every accumulation is fp32: the result is the exact attention over the
//     de-quantised cache up to fp16 rounding of q, P' and the output (parity bar 1e-3, tests/test_attention_gpu.py).
#include "common.h"
#include "kv_quant.h"
#include <mutex>

namespace {

constexpr int PAGE_TOK = 64;
constexpr int DH = 128;
constexpr int DHB = 64;        // KV4 bytes per token per head
constexpr int UT = 32;         // tokens per pipeline unit (half a page)
constexpr int USB = UT * DHB;  // bytes of K (or V) data per unit and KV head
constexpr int NW = 8;          // waves per workgroup; all of them may own units
constexpr int NWT = NW;
typedef u32 v2u __attribute__((ext_vector_type(2)));
constexpr int QS_ATTNQ_CAP = 4096;   // sequences the attention + quant fusion can hand over (larger batches run the pair)
constexpr int QS_ATTNQ_ROW = 4096;   // values per row at most (H x 128 <= 4096: quant_kernel's 256-thread mapping)
constexpr int SVC = NW - 1;    // the service wave (RoPE, operand build, new token) - the wave that owns the fewest units
#ifndef QS_SEQ
#define QS_SEQ(x) asm volatile("" : "+v"(x))   // one fp32 addition at a time (no re-association / pairing): row_ops.h
#endif
constexpr int MAXP = 192;      // longest page table this kernel is dispatched for (dispatcher: max_blocks <= MAXP)

// (explicit global address space: a generic pointer would make these FLAT stores, and a flat access may alias LDS, so
// the compiler drains the LDS-DMA queue - vmcnt(0) - in front of it)
typedef __attribute__((address_space(1))) uint8_t* g_u8;
typedef __attribute__((address_space(1))) uint16_t* g_u16;
__device__ __forceinline__ void wave_quant_store4(_Float16 v0, _Float16 v1, uint8_t* dst_, __half* scale_p_,
                                                  __half* zero_p_, int lane) {
    g_u8 dst = (g_u8)dst_;
    g_u16 scale_p = (g_u16)reinterpret_cast<uint16_t*>(scale_p_), zero_p = (g_u16)reinterpret_cast<uint16_t*>(zero_p_);
    const float mx = wave_max_dpp(fmaxf((float)v0, (float)v1));   // (max / min are order-independent: bit-exact)
    const float mn = wave_min_dpp(fminf((float)v0, (float)v1));
    const QParams p = make_qparams<true>(mn, mx);
    const unsigned u0 = quant_u8(v0, p), u1 = quant_u8(v1, p);
    dst[lane] = (uint8_t)((u0 & 0xFu) | (u1 << 4));
    if (lane == 0) {
        *scale_p = __builtin_bit_cast(uint16_t, p.scale);
        *zero_p = __builtin_bit_cast(uint16_t, p.zero);
    }
}

__device__ __forceinline__ u32 pack_h2(float a, float b) {
    const h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(u32, v);
}
// (x & m) | c as ONE v_and_or_b32: m and c live in VGPRs (VOP3 takes no 32-bit literals, which is why the compiler
// otherwise emits v_and + v_or with literal operands)
__device__ __forceinline__ u32 and_or(u32 x, u32 m, u32 c) {
    return (x & m) | c;
}

typedef __attribute__((address_space(3))) const uint8_t* lds_u8;   // 32-bit LDS address (keeps ds_read, not flat_load)

// EXP: timing / ablation switches (libraries built with -DQS_TIMING only: qs_set_attention_variant(200 + EXP), G = 4 only;
// the shipped library instantiates EXP = 0 and ignores the request):
//   1 = unit DMA WITHOUT the non-temporal hint (default: nt - every KV byte is read once per step; measured -4 % at
//       L = 1033 ... -12 % at L = 4096), 2 = no compute (DMA + waits only: results are wrong by design),
//   4 = skip phase A (RoPE / new token: wrong by design), 32 = timeline trace (s_memtime stamps into the split workspace,
//       scripts/trace_attn.py)
template <int G, int EXP = 0>
__global__ __launch_bounds__(NWT * 64, 4) void decode_attention_mfma_kernel(
    const _Float16* __restrict__ q, const _Float16* __restrict__ k, const _Float16* __restrict__ v,
    const int64_t* __restrict__ kv_pointers, const int* __restrict__ lengths, _Float16* __restrict__ out,
    int num_heads, int num_kv_heads, int64_t q_stride0, int64_t kv_stride0, int max_blocks, int timestep,
    float rope_base, const float2* __restrict__ rope_tab, int rope_tab_len, int nsplit, float* __restrict__ ws,
    int8_t* __restrict__ qout, __half* __restrict__ qscale, __half* __restrict__ qrowsum, unsigned* __restrict__ qcounters,
    int kflags) {
    __shared__ __attribute__((aligned(16))) uint8_t s_kv[2 * NW * 2 * USB];     // [K | V][wave][slot 0 / 1][2 KiB]
    __shared__ __attribute__((aligned(16))) _Float16 s_meta[NW][2][4][UT];      // [wave][slot]: k scale, k zero, v scale, v zero
    __shared__ __attribute__((aligned(16))) _Float16 s_q[G][DH];                // rotated q of the G heads
    __shared__ __attribute__((aligned(16))) _Float16 s_qp[16][DH];              // Q.K^T B operand (see below)
    __shared__ __attribute__((aligned(16))) _Float16 s_knew[DH];
    __shared__ __attribute__((aligned(16))) _Float16 s_vnew[DH];                // the new token's raw v
    __shared__ float s_cur[16];
    __shared__ float2 s_qc[16];                                                 // per head: (qsum, Qoff) - see the operand build
    __shared__ float s_m[NW][G], s_l[NW][G];
    __shared__ int s_flag;                                                      // service wave -> unit waves: operands ready

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hkv = blockIdx.x, b = blockIdx.y;
    // EXP & 32: timeline trace (timing tool only): s_memtime stamps of every wave, collected in LDS - a global store per stamp
    // would queue behind the CU's LDS-DMA burst and stall the wave, i.e. perturb exactly what is being measured - and copied
    // to the split workspace when the kernel ends
    __shared__ unsigned long long s_stamps[(EXP & 32) ? NWT * 16 : 1];
    auto stamp = [&](int i) {
        if constexpr (EXP & 32) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) s_stamps[wave * 16 + i] = t;
        }
    };
    if constexpr (EXP & 32) {
        if (lane < 16) s_stamps[wave * 16 + lane] = 0;
    }
    stamp(0);
    // Reset of the hand-over flag behind a workgroup barrier, FIRST THING in the kernel (round 4): every wave has ~1 us of length
    // / page-address latency in front of it anyway, so the barrier is free here - where it used to stand (after the service
    // wave's entry requests) it held the unit waves' first requests back by ~2 000 cycles.
    if (!(kflags & 2)) {
        const u32 fa = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(&s_flag);
        if (tid == SVC * 64) asm volatile("ds_write_b32 %0, %1" ::"v"(fa), "v"(0));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier");
    }
    // the rows of the new token (wave-uniform addresses: scalar bases of the service wave's entry requests below)
    const _Float16* qb = q + (size_t)b * q_stride0 + (size_t)hkv * G * DH;
    const _Float16* kb = k + (size_t)b * kv_stride0 + (size_t)hkv * DH;
    const _Float16* vb = v + (size_t)b * kv_stride0 + (size_t)hkv * DH;
    auto uni_ptr = [](const void* x) -> uint64_t {
        return ((uint64_t)(u32)__builtin_amdgcn_readfirstlane((u32)((uint64_t)(uintptr_t)x >> 32)) << 32) |
               (uint64_t)(u32)__builtin_amdgcn_readfirstlane((u32)(uint64_t)(uintptr_t)x);
    };
    const uint64_t qb_s = uni_ptr(qb), kb_s = uni_ptr(kb), vb_s = uni_ptr(vb);
    const int64_t* ktab = kv_pointers + (size_t)b * 2 * max_blocks;
    const int64_t* vtab = ktab + max_blocks;
    const bool spec = (nsplit == 1 || blockIdx.z == 0);
    // service wave: q / k / v of the new token do not depend on the context length - requested at kernel entry, ahead of
    // every unit DMA of this CU in the memory pipeline (which serves requests in order: loads issued after the burst of
    // the first round would come back ~4 us later); raised issue priority until the operands are published
    // (by LDS-DMA straight into s_q / s_knew / s_vnew: 256 B per instruction, no destination registers - register loads
    // issued under `wave == SVC` here and consumed under the same test further down leave their registers "load pending"
    // on the compiler's infeasible SVC -> non-SVC path, and its waitcnt pass then drains the unit waves' DMA queue - vmcnt(0) -
    // wherever the shared code reuses one of them)
    if (wave == SVC) {
        asm volatile("s_setprio 3");
        if constexpr (!(EXP & 4)) {
            const u32 lane4 = (u32)lane * 4u;
            auto glds4 = [&](uint64_t sb, const _Float16* dst) {
                const u32 ld = __builtin_amdgcn_readfirstlane((u32)(uintptr_t)(__attribute__((address_space(3))) const void*)dst);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dword %0, %1" ::"v"(lane4), "s"(sb), "s"(ld) : "memory");
            };
#pragma unroll
            for (int h = 0; h < G; ++h) glds4(qb_s + (uint64_t)(h * DH * 2), &s_q[h][0]);
            glds4(kb_s, &s_knew[0]);
            glds4(vb_s, &s_vnew[0]);
        }
        stamp(5);      // (trace: service wave - entry requests issued)
    }
    // The length and the page addresses of the wave's first two units, as SCALAR loads in one statement with its own wait
    // (explicit: behind the volatile statements above the compiler would make them vector loads).  The addresses do not depend
    // on the length when this workgroup starts at unit 0 - one memory round trip less on the launch -> first bytes chain: the
    // wave's first unit is unit `wave` (page wave / 2), its second one unit wave + 7 when seven waves share the units (the
    // usual case, see the ownership rule below; with eight the address is fetched when the length is known).  Entries beyond
    // the table and a null `lengths` read a valid dummy (entry 0 / the table itself) and are not used.
    int64_t kpage0, vpage0, kpage7, vpage7;
    int len_raw, gen_raw;                             // gen_raw: the sequence's hand-off generation (attention + quant fusion, below)
    {
        const int p0 = (wave >> 1) < max_blocks ? (wave >> 1) : 0, p7 = ((wave + NW - 1) >> 1) < max_blocks ? ((wave + NW - 1) >> 1) : 0;
        const uint64_t a0 = uni_ptr(ktab + p0), a1 = uni_ptr(vtab + p0), a2 = uni_ptr(ktab + p7), a3 = uni_ptr(vtab + p7);
        const uint64_t a4 = uni_ptr(lengths ? (const void*)(lengths + b) : (const void*)ktab);
        const uint64_t a5 = uni_ptr(qcounters ? (const void*)(qcounters + b) : (const void*)ktab);
        asm volatile("s_load_dwordx2 %0, %6, 0x0\n\ts_load_dwordx2 %1, %7, 0x0\n\ts_load_dwordx2 %2, %8, 0x0\n\t"
                     "s_load_dwordx2 %3, %9, 0x0\n\ts_load_dword %4, %10, 0x0\n\ts_load_dword %5, %11, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(kpage0), "=&s"(vpage0), "=&s"(kpage7), "=&s"(vpage7), "=&s"(len_raw), "=&s"(gen_raw)
                     : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5)
                     : "memory");
    }
    const int tl = lengths ? len_raw - 1 : timestep;   // tlength, Template.hpp:901
    if (tl < 0) return;
    stamp(1);
    const float inv_sqrt = 0.08838834764831845f;
    const float qk_scale = inv_sqrt * 1.4426950408889634f;   // scores live in the log2 domain: exp2 everywhere
    constexpr int GP = G <= 4 ? 4 : 8;   // lane scheme of the compacted softmax: heads padded to 4 or 8 (surplus head lanes idle)
    const int li = lane & 15, tg = lane >> 4;
    uint8_t* const s_kw = s_kv + wave * (2 * USB);                // this wave's K ring (2 slots)
    uint8_t* const s_vw = s_kv + (NW + wave) * (2 * USB);         // this wave's V ring (2 slots)

    // ---- unit fetch by LDS-DMA ------------------------------------------------------------------------------------
    const int nunits = (tl + UT - 1) >> 5;
    // split-KV (flash-decoding across workgroups, gridDim.z = nsplit > 1 when batch x kv-heads cannot fill the chip):
    // this workgroup handles units [u_begin, u_end); split 0 also owns the new token (cache write + its own term)
    const int z = blockIdx.z;
    const int ups = (nunits + nsplit - 1) / nsplit;
    const int u_begin = z * ups, u_end = min(nunits, u_begin + ups);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // Page addresses of the later units: SCALAR loads (scalar cache / lgkmcnt - they never enter the in-order vmcnt queue
    // of the unit DMA; the table rows were touched by the first-round lookups).  From inline asm, because behind the DMA
    // statements' "memory" clobber the compiler itself would fall back to vector loads for kv_pointers.
    auto next_pages = [&](int p, int64_t& kn, int64_t& vn) {
        const uint64_t ka = (uint64_t)(uintptr_t)(ktab + p), va = (uint64_t)(uintptr_t)(vtab + p);
        // (readfirstlane returns a signed int: without the u32 casts the low half would be sign-extended into the high one)
        const uint64_t ks = ((uint64_t)(u32)__builtin_amdgcn_readfirstlane((u32)(ka >> 32)) << 32) |
                            (uint64_t)(u32)__builtin_amdgcn_readfirstlane((u32)ka);
        const uint64_t vs = ((uint64_t)(u32)__builtin_amdgcn_readfirstlane((u32)(va >> 32)) << 32) |
                            (uint64_t)(u32)__builtin_amdgcn_readfirstlane((u32)va);
        asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(kn), "=&s"(vn) : "s"(ks), "s"(vs) : "memory");
    };
    // LDS-DMA issued from inline asm (recipe: cdna_hip_programming.md 5.7 - M0 carries the wave-uniform LDS base and is
    // written in the statement that reads it).  Deliberately NOT the builtin: the compiler's waitcnt pass models every
    // builtin LDS-DMA as a pending LDS write and puts a vmcnt(0) in front of the next ds_read of the unit loop as soon
    // as units are in flight at loop entry - which serialises fetch and compute and makes the counted waits below
    // meaningless.  With asm the compiler sees no DMA at all; every wait on this queue is explicit (vmcnt(9) / (6) / (3) / (0)).
    // Scalar-base form (global_load_lds voffset, s[base:base+1]): the source address of a piece is a wave-uniform base
    // (page + head + half + piece offset: SALU adds) plus a per-lane byte offset that never changes (16 * lane) - no 64-bit
    // vector address arithmetic per request.  (s_nop 4: five wait states between the write of an SGPR by a VALU
    // instruction - v_readfirstlane - and its use as a VMEM base; covers the M0 write -> LDS-DMA state as well.)
    auto uni64 = [&](int64_t x) -> uint64_t {        // provably wave-uniform 64-bit value for an "s" operand
        return ((uint64_t)(u32)__builtin_amdgcn_readfirstlane((u32)((uint64_t)x >> 32)) << 32) |
               (uint64_t)(u32)__builtin_amdgcn_readfirstlane((u32)(uint64_t)x);
    };
    auto glds16 = [&](uint64_t sbase, u32 voff, u32 ldst) {
        if constexpr (!(EXP & 1))
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(ldst) : "memory");
        else
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(ldst) : "memory");
    };
    // 4 B per lane from sbase + voff on HALF of the lanes (EXEC narrowed inside the statement - the surrounding code is
    // wave-uniform, all 64 lanes active); LDS destination = ldst + 4 * lane (the lane's own number, also for the upper half)
    auto glds4_lower = [&](uint64_t sbase, u32 voff, u32 ldst) {
        uint64_t ex;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_hi, 0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b64 exec, %0"
                     : "=&s"(ex) : "v"(voff), "s"(sbase), "s"(ldst) : "memory");
    };
    auto glds4_upper = [&](uint64_t sbase, u32 voff, u32 ldst) {
        uint64_t ex;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, 0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b64 exec, %0"
                     : "=&s"(ex) : "v"(voff), "s"(sbase), "s"(ldst) : "memory");
    };
    // Unit u = tokens [32 u, 32 u + 32) = half (u & 1) of page u >> 1.  Request group A: the K scales | zeros of the unit's
    // tokens (lanes 0-15 | 16-31: 128 B) and 2 x 1 KiB of K data; group B: the same of V (meta by lanes 32-63, so that the
    // lane-linear LDS destination of both groups is ONE 256-byte record [k scale | k zero | v scale | v zero] per slot).
    // 3 VMEM instructions each - what the counted waits below rely on.
    const u32 meta_lane = (u32)(((lane >> 4) & 1) * num_kv_heads * PAGE_TOK * 2 + (lane & 15) * 4);
    const u32 lane16 = (u32)lane * 16u;
    const u32 lds_k = __builtin_amdgcn_readfirstlane((u32)(uintptr_t)(lptr_t)s_kw);
    const u32 lds_v = __builtin_amdgcn_readfirstlane((u32)(uintptr_t)(lptr_t)s_vw);
    const u32 lds_m = __builtin_amdgcn_readfirstlane((u32)(uintptr_t)(lptr_t)(&s_meta[wave][0][0][0]));
    const u32 meta_off = (u32)(num_kv_heads * PAGE_TOK * DHB + hkv * PAGE_TOK * 2);
    const u32 data_off = (u32)(hkv * PAGE_TOK * DHB);
    auto dma_data = [&](uint64_t base, u32 ldst, int valid_tok) {   // 2 x 1 KiB of one unit
        if (valid_tok >= UT) {                    // wave-uniform: every unit but the sequence's last one
            glds16(base, lane16, ldst);
            glds16(base + 1024, lane16, ldst + 1024);
        } else {
            // lanes of tokens >= valid stay idle (their bytes are never read: masked scores / zero probabilities);
            // lane 0 always loads so that the instruction issues and the vmcnt bookkeeping holds
            if ((lane >> 2) < valid_tok || lane == 0) glds16(base, lane16, ldst);
            if (16 + (lane >> 2) < valid_tok || lane == 0) glds16(base + 1024, lane16, ldst + 1024);
        }
    };
    auto dma_a = [&](int64_t page, int half, int slot, int valid_tok) {
        const uint64_t pg = uni64(page);
        glds4_lower(pg + (meta_off + (u32)(half * UT * 2)), meta_lane, lds_m + (u32)(slot * (4 * UT * 2)));
        dma_data(pg + (data_off + (u32)(half * USB)), lds_k + (u32)(slot * USB), valid_tok);
    };
    auto dma_b = [&](int64_t page, int half, int slot, int valid_tok) {
        const uint64_t pg = uni64(page);
        glds4_upper(pg + (meta_off + (u32)(half * UT * 2)), meta_lane, lds_m + (u32)(slot * (4 * UT * 2)));
        dma_data(pg + (data_off + (u32)(half * USB)), lds_v + (u32)(slot * USB), valid_tok);
    };
    // ---- phase A on the SERVICE wave, concurrent with the first unit fetch ---------------------------------------------
    // What the timeline trace (EXP & 32) showed for the one-barrier-after-DMA form: issuing the first-round DMA takes a
    // wave 2-4 us (the memory pipeline back-pressures the burst), so any phase-A work that sits behind that issue in
    // program order, and any barrier all waves must reach after it, completes at ~10 us.  Now: seven waves do nothing
    // but issue their DMA; the last wave (it owns the fewest units) first loads q / k / v / the RoPE coefficients with
    // ordinary loads - its vmcnt queue is still empty, the compiler's own counted waits are right - rotates, builds the
    // Q.K^T operand image, raises an LDS flag (~2 us after launch) and only then issues its own unit DMA; the other waves
    // poll that flag after their issue (no barrier couples the waves to each other) and start on whichever unit has
    // landed.  The new token's cache write and its own score follow on the service wave after its units, off everybody's
    // critical path.  (A ninth, unit-less wave was tried first: 576-thread workgroups no longer fit twice on a CU.)
    const int blk = tl >> 6, slot_new = tl & 63;
    // Unit ownership.  The service wave starts its units ~2 us after the others (operand build first) and then still has
    // the new token's cache write and score to do: with its round-robin share it is the last wave to finish.  Whenever
    // seven waves need no more rounds than eight would - ceil(n / 7) == ceil(n / 8) - it therefore owns NO units, finishes
    // the new token right after the operand build and waits at the merge; the other seven take the units round-robin.
    // (round 4 also tried a graded form - the service wave takes the last 0-7 units of the range, chosen by a time model read
    // off the timeline trace so that all eight waves finish together: 0.1-0.5 us SLOWER at 1 030-2 000 tokens, 1.2 us at 640:
    // units requested that late arrive behind everything else.  Dropped; HISTORY.md.)
    const int nu = max(u_end - u_begin, 0);
    const bool svc_free = (nu + NW - 2) / (NW - 1) == (nu + NW - 1) / NW;
    const int PS = svc_free ? NW - 1 : NW;                            // unit stride of a wave
    const int u_first = u_begin + wave;
    // (round 4, measured and dropped - HISTORY.md: two-phase ownership (the second-dispatched waves 4-6 one round fewer, the rest
    //  over waves 0-3): -0.4-0.6 us at 1 030-1 100 tokens, +0.1-0.5 at 640, equal elsewhere; ownership by SIMD load (wave 3, whose
    //  SIMD partner is the service wave, also works the service wave's slot): +1 us - a wave's time per unit does not depend on
    //  what shares its SIMD, it is the back-pressured issue of its own six requests)
    auto unit_of = [&](int i) -> int { return u_first + i * PS; };
    const int cnt = ((!svc_free || wave != SVC) && wave < nu) ? (nu - wave + PS - 1) / PS : 0;   // units this wave owns
    // speculative page addresses fit: the first unit always, the second one for the seven-wave round-robin
    const bool spec2 = spec && PS == NW - 1;
    // Hand-over of the operands from the service wave to the unit waves: an LDS flag the unit waves poll after issuing their
    // first requests (reset behind the barrier at the top of the kernel).  Round 4 measured the alternative - ONE raw s_barrier that the
    // unit waves enter after their first requests and the service wave when the operands are in LDS, no reset barrier, no
    // polling - at +-0.2 us of this form over 300 ... 2 000 tokens (kflags & 2, qs_set_attention_variant(5)): not adopted.
    const bool use_flag = !(kflags & 2);
    // first request of a wave: its unit 0 (4.25 KiB), then - see the call sites for WHEN - its unit 1
    auto first_round = [&](int which) {       // which: 1 = unit 0 only, 2 = unit 1 only, 3 = both
        if (cnt > 0) {
            // page addresses: requested together with the length for split 0 (scalar loads, before any asm statement:
            // behind an asm "memory" clobber the compiler falls back to vector loads for kv_pointers)
            const int u1 = unit_of(1);
            int64_t kfirst = kpage0, vfirst = vpage0, ksecond = kpage7, vsecond = vpage7;
            // (everything else by the asm scalar loads - compiler-visible loads here would be vector loads whose pending state
            //  reaches the shared code on some path and is drained there with vmcnt(0))
            if ((which & 1) && !spec) next_pages(u_first >> 1, kfirst, vfirst);
            if ((which & 2) && cnt > 1 && !spec2) next_pages(u1 >> 1, ksecond, vsecond);
            if (which & 1) {
                const int vt0 = min(UT, tl - u_first * UT);
                dma_a(kfirst, u_first & 1, 0, vt0);
                stamp(3);
                dma_b(vfirst, u_first & 1, 0, vt0);
            }
            if ((which & 2) && cnt > 1) {
                const int vt1 = min(UT, tl - u1 * UT);
                dma_a(ksecond, u1 & 1, 1, vt1);
                dma_b(vsecond, u1 & 1, 1, vt1);
            }
        }
        if (which & 2) stamp(2);
    };
    // the new token's cache write (split 0) and its own score: service wave, after its units - or, when it owns none,
    // straight after the operand build
    auto new_token_work = [&]() {
        if (z == 0) {
            uint8_t* pgk = reinterpret_cast<uint8_t*>(ktab[blk]);
            __half* sck = reinterpret_cast<__half*>(pgk + (size_t)num_kv_heads * PAGE_TOK * DHB);
            wave_quant_store4(s_knew[2 * lane], s_knew[2 * lane + 1], pgk + ((size_t)hkv * PAGE_TOK + slot_new) * DHB,
                              sck + hkv * PAGE_TOK + slot_new, sck + num_kv_heads * PAGE_TOK + hkv * PAGE_TOK + slot_new, lane);
            uint8_t* pgv = reinterpret_cast<uint8_t*>(vtab[blk]);
            __half* scv = reinterpret_cast<__half*>(pgv + (size_t)num_kv_heads * PAGE_TOK * DHB);
            wave_quant_store4(s_vnew[2 * lane], s_vnew[2 * lane + 1], pgv + ((size_t)hkv * PAGE_TOK + slot_new) * DHB,
                              scv + hkv * PAGE_TOK + slot_new, scv + num_kv_heads * PAGE_TOK + hkv * PAGE_TOK + slot_new, lane);
        }
#pragma unroll
        for (int h = 0; h < G; ++h) {
            float d = (float)s_q[h][lane] * (float)s_knew[lane] + (float)s_q[h][64 + lane] * (float)s_knew[64 + lane];
            d = wave_sum_dpp(d);
            if (lane == 0) s_cur[h] = d * qk_scale;
        }
    };
    if (wave != SVC) {
        // (round 4, measured and dropped - HISTORY.md: the wave's second unit requested later (after the hand-over, when group A
        //  of the first unit has landed, after fixed delays), the first unit's requests at raised issue priority, raised priority
        //  for the waves that own one unit more than the others, the service wave taking a graded share of the units: each within
        //  +-0.3 us of this form at 640 ... 2 000 tokens, none better everywhere)
        first_round(3);
        if (use_flag) {
            while (*(volatile __attribute__((address_space(3))) int*)(&s_flag) == 0) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        stamp(4);
    } else {
        if constexpr (!(EXP & 4)) {
            // ordinary loads: this wave's queue carries nothing else, the compiler's own counted waits are right here
            stamp(6);  // (trace: service wave - phase A entered)
            RopeCS cs;
            if (rope_tab && tl < rope_tab_len) {
                const float2 t = rope_tab[(size_t)tl * 64 + lane];   // same double-evaluated, float-rounded values
                cs.c = t.x;
                cs.s = t.y;
            } else {
                cs = rope_coef(lane, tl, rope_base, DH);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the raw q / k / v rows have landed (this wave's queue holds nothing else)
            stamp(7);  // (trace: service wave - q / k / v and the RoPE coefficients are here)
#pragma unroll
            for (int h = 0; h < G; ++h) {                      // rotated in place: a lane touches only its own two elements
                _Float16 a, bb;
                rope_pair((float)s_q[h][lane], (float)s_q[h][64 + lane], cs, a, bb);
                s_q[h][lane] = a;
                s_q[h][64 + lane] = bb;
            }
            _Float16 ka, kbb;
            rope_pair((float)s_knew[lane], (float)s_knew[64 + lane], cs, ka, kbb);
            s_knew[lane] = ka;
            s_knew[64 + lane] = kbb;
            // B operand of Q.K^T for lane (head li, kg = tg), MFMA w: dims 32tg + 8w + {0,4,1,5,2,6,3,7}; the positions that
            // meet hi-nibble operands (1024 + 16 n) carry q/16 (same wave wrote s_q: LDS keeps a wave's accesses in order)
            // The per-head constants of the offset form are summed HERE, from the values being written (round 4; rounds 2-3: by
            // every unit wave after the hand-over, from s_qp - 4 LDS reads and 4 cross-lane shuffles on the critical path of
            // every wave's first Q.K^T, ~3 000 cycles in the trace): qsum = sum_d q_eff_d, Qoff = sum over the operand of
            // 1024 * q' (what the 1024+n / 1024+16n operand form adds to the raw dot product); q_eff = what the MFMA effectively
            // multiplies n by.  Same summation order as before (per lane over its 32 dims, then lanes xor 16, xor 32).
            float se = 0.f, so = 0.f;   // sums over lo-form / hi-form operand positions of this lane's 32 dims
            h8 xr[4];                   // (all four reads first: one LDS round trip instead of four)
#pragma unroll
            for (int w = 0; w < 4; ++w) xr[w] = *reinterpret_cast<const h8*>(&s_q[li < G ? li : 0][32 * tg + 8 * w]);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                h8 x = xr[w];
                if (li >= G) x = (h8){0, 0, 0, 0, 0, 0, 0, 0};         // heads >= G: zero rows of the operand
                const _Float16 s16 = (_Float16)0.0625f;
                const h8 op = {x[0], x[4], x[1] * s16, x[5] * s16, x[2], x[6], x[3] * s16, x[7] * s16};
                *reinterpret_cast<h8*>(&s_qp[li][32 * tg + 8 * w]) = op;
                se += (float)op[0] + (float)op[1] + (float)op[4] + (float)op[5];
                so += (float)op[2] + (float)op[3] + (float)op[6] + (float)op[7];
            }
            {
                auto xor16 = [](float v) {
                    const int xi = __builtin_bit_cast(int, v);
                    const auto r = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
                    return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
                };
                auto xor32 = [](float v) {
                    const int xi = __builtin_bit_cast(int, v);
                    const auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
                    return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
                };
                se = xor32(xor16(se));
                so = xor32(xor16(so));
                if (tg == 0) s_qc[li] = make_float2(se + 16.f * so, 1024.f * (se + so));
            }
            if (use_flag) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) *(volatile __attribute__((address_space(3))) int*)(&s_flag) = 1;
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            asm volatile("s_setprio 0");
            stamp(4);
            if (svc_free) new_token_work();
        } else {
            if (use_flag) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) *(volatile __attribute__((address_space(3))) int*)(&s_flag) = 1;
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        first_round(3);                    // the service wave's own units: requested only now (its queue was kept clean)
    }

    // per-lane constants of head li (summed by the service wave with the operand image)
    const float2 qc = s_qc[li & (GP - 1)];
    const float qsum = qc.x, nqoff = -qc.y;
    if constexpr (EXP & 32) {
        float probe = nqoff;
        asm volatile("" : "+v"(probe));
        stamp(11);
    }

    // mask / magic constants parked in VGPRs so that (x & m) | c is one v_and_or_b32
    u32 c_lo = 0x000F000Fu, c_hi = 0x00F000F0u, c_magic = 0x64006400u;
    asm volatile("" : "+v"(c_lo), "+v"(c_hi), "+v"(c_magic));

    v4f acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = (v4f){0.f, 0.f, 0.f, 0.f};
    float m_run = -3.0e38f, l_part = 0.f, corr = 0.f, psum = 0.f;   // psum = sum of P' (removes the V operand offsets)

    // Lane-constant LDS addresses and the Q.K^T B operand stay in registers across the unit loop (the unit form of the
    // loop needs ~30 registers fewer than the page form did, so nothing spills: tests/test_kernel_contracts.py)
    constexpr int NS = GP == 4 ? 2 : 4;     // scores per lane in the compacted softmax
    // this lane's tokens there: 16 t + 4 tg + r0 + {0 .. NS-1}
    //   GP = 4: lane li = h + 4 (2 t + rp): r0 = 2 rp;   GP = 8: lane li = h + 8 t: r0 = 0
    const int tok0 = 16 * (li >> 3) + 4 * tg + (GP == 4 ? 2 * ((li >> 2) & 1) : 0);
    u32 kl0 = (u32)(uintptr_t)(lptr_t)s_kw + (u32)(li * DHB + 16 * tg);
    u32 vl0 = (u32)(uintptr_t)(lptr_t)s_vw + (u32)((4 * tg) * DHB + 4 * li);
    u32 ml0 = (u32)(uintptr_t)(lptr_t)(&s_meta[wave][0][0][0]) + (u32)(2 * tok0);
    asm volatile("" : "+v"(kl0), "+v"(vl0), "+v"(ml0));   // opaque: one register each, every access below is base + immediate
    h8 qB[4];
    {
        const lds_u8 ql = (lds_u8)(&s_qp[0][0]) + (li * (DH * 2) + 64 * tg);
#pragma unroll
        for (int w = 0; w < 4; ++w) qB[w] = *(const __attribute__((address_space(3))) h8*)(ql + 16 * w);
    }

    int it = 0;          // units this wave has consumed: ring slot = it & 1
    int rem = cnt - 1;   // units this wave still owns after the current one
    for (int u = u_first; it < cnt; ++it, --rem, u = unit_of(it)) {
        const bool has2 = rem >= 2;   // (has1 = rem >= 1)
        // A(u) landed?  Younger VMEM operations of this wave at this point: B(u) (3) and, if it exists, unit u + PS (6).
        // (the wave-uniform choice between the two counted waits is a scalar branch inside the statement)
        asm volatile("; QS_LOOP_BEGIN\n\ts_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(9)\n\ts_branch 2f\n1:\n\ts_waitcnt vmcnt(3)\n2:"
                     ::"s"(rem) : "memory", "scc");
        if (it < 2) stamp(5 + 2 * it);
        const int slot = it & 1;
        const int valid = min(UT, tl - u * UT);
        const bool full = valid == UT;   // wave-uniform: only the last unit needs masking
        const int u2 = unit_of(it + 2), valid2 = min(UT, tl - u2 * UT);
        int64_t kpage_next = 0, vpage_next = 0;
        if (has2) next_pages(u2 >> 1, kpage_next, vpage_next);
        if constexpr (EXP & 2) {               // timing experiment: memory side only
            if (has2) dma_a(kpage_next, u2 & 1, slot, valid2);
            asm volatile("s_cmp_lt_u32 %0, 2\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(9)\n\ts_branch 3f\n1:\n\ts_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 2f\n\t"
                         "s_waitcnt vmcnt(6)\n\ts_branch 3f\n2:\n\ts_waitcnt vmcnt(0)\n3:" ::"s"(rem) : "memory", "scc");
            if (has2) dma_b(vpage_next, u2 & 1, slot, valid2);
            continue;
        }
        const lds_u8 kl = (lds_u8)(kl0 + (u32)(slot * USB));
        const lds_u8 vl = (lds_u8)(vl0 + (u32)(slot * USB));
        const lds_u8 ml = (lds_u8)(ml0 + (u32)(slot * (4 * UT * 2)));
        // ---------------- Q.K^T : 2 tiles of 16 tokens ----------------
        v4f craw[2];   // craw[t][r] = raw dot (offsets already cancelled) of token 16t + 4tg + r with head li
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const v4u raw = *(const __attribute__((address_space(3))) v4u*)(kl + 16 * t * DHB);
            v4f c = {nqoff, nqoff, nqoff, nqoff};   // start at -Qoff: the operand offsets cancel inside the MFMA chain
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const u32 x = raw[w], xs = x >> 8;
                const v4u a4 = {and_or(x, c_lo, c_magic), and_or(x, c_hi, c_magic), and_or(xs, c_lo, c_magic),
                                and_or(xs, c_hi, c_magic)};
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a4), qB[w], c, 0, 0, 0);
            }
            craw[t] = c;
        }
        // Only the columns li < G of the 16x16 results are real heads.  Instead of running the softmax on 8 values per
        // lane with 3/4 (1/2) of the lanes idle, the values move to the idle lanes of their 16-lane row (DPP row_shr
        // with bank masks): every lane then owns NS scores of head h = li % GP.
        float sc[NS];
        float m_new;
        {
            float ksf[NS], kzf[NS];
            if constexpr (GP == 4) {
                const h2 ks = *(const __attribute__((address_space(3))) h2*)(ml);
                const h2 kz = *(const __attribute__((address_space(3))) h2*)(ml + UT * 2);
                ksf[0] = (float)ks[0], ksf[1] = (float)ks[1], kzf[0] = (float)kz[0], kzf[1] = (float)kz[1];
            } else {
                const h4 ks = *(const __attribute__((address_space(3))) h4*)(ml);
                const h4 kz = *(const __attribute__((address_space(3))) h4*)(ml + UT * 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) ksf[j] = (float)ks[j], kzf[j] = (float)kz[j];
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                // (scalar copies first: __builtin_bit_cast of a vector-element lvalue reads element 0)
                int x;
                if constexpr (GP == 4) {   // whole 4-lane banks move: bank-masked DPP writes
                    const float c00 = craw[0][j], c02 = craw[0][2 + j], c10 = craw[1][j], c12 = craw[1][2 + j];
                    x = __builtin_bit_cast(int, c00);
                    x = __builtin_amdgcn_update_dpp(x, __builtin_bit_cast(int, c02), 0x114, 0xF, 0x2, false);   // row_shr:4 -> bank 1
                    x = __builtin_amdgcn_update_dpp(x, __builtin_bit_cast(int, c10), 0x118, 0xF, 0x4, false);   // row_shr:8 -> bank 2
                    x = __builtin_amdgcn_update_dpp(x, __builtin_bit_cast(int, c12), 0x11C, 0xF, 0x8, false);   // row_shr:12 -> bank 3
                } else {
                    const float c0 = craw[0][j], c1 = craw[1][j];
                    x = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, c0), __builtin_bit_cast(int, c1), 0x118, 0xF, 0xC, false);
                }
                sc[j] = (ksf[j] * qk_scale) * (__builtin_bit_cast(float, x) - kzf[j] * qsum);
            }
            if (!full) {
#pragma unroll
                for (int j = 0; j < NS; ++j)
                    if (tok0 + j >= valid) sc[j] = -3.0e38f;   // also discards NaN from garbage scales
            }
            // K slot consumed -> request A(u + 2 PS) into it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (has2) dma_a(kpage_next, u2 & 1, slot, valid2);
            // maximum over the unit's 32 tokens of head h: own scores, the lanes li +- 4 / 8 of the row (GP = 4; li +- 8 for
            // GP = 8) as DPP source operands of the max itself, the other three rows by lane-permute swaps - no LDS round trip
            // (s_nop 1: two wait states between a VALU write and its use as a DPP source)
            float mx = sc[0];
#pragma unroll
            for (int j = 1; j < NS; ++j) mx = fmaxf(mx, sc[j]);
            if constexpr (GP == 4) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(mx));
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(mx));
            {
                const int xi = __builtin_bit_cast(int, mx);
                const auto r = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
                mx = fmaxf(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
            }
            {
                const int xi = __builtin_bit_cast(int, mx);
                const auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
                mx = fmaxf(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
            }
            m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            if (__any(alpha != 1.0f)) {
                l_part *= alpha;
                corr *= alpha;
                psum *= alpha;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] *= alpha;
            }
        }
        // B(u) landed?  Younger: unit u + PS (6, if it exists) and A(u + 2 PS) (3, if it exists)
        asm volatile("s_cmp_lt_u32 %0, 2\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(9)\n\ts_branch 3f\n1:\n\ts_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 2f\n\t"
                     "s_waitcnt vmcnt(6)\n\ts_branch 3f\n2:\n\ts_waitcnt vmcnt(0)\n3:" ::"s"(rem) : "memory", "scc");
        u32 pbv[4];   // P'^T operand (B of P.V) of the unit
        {
            float vsf[NS], vzf[NS];
            if constexpr (GP == 4) {
                const h2 vs = *(const __attribute__((address_space(3))) h2*)(ml + 2 * UT * 2);
                const h2 vz = *(const __attribute__((address_space(3))) h2*)(ml + 3 * UT * 2);
                vsf[0] = (float)vs[0], vsf[1] = (float)vs[1], vzf[0] = (float)vz[0], vzf[1] = (float)vz[1];
            } else {
                const h4 vs = *(const __attribute__((address_space(3))) h4*)(ml + 2 * UT * 2);
                const h4 vz = *(const __attribute__((address_space(3))) h4*)(ml + 3 * UT * 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) vsf[j] = (float)vs[j], vzf[j] = (float)vz[j];
            }
            float pp[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const float pe = __builtin_amdgcn_exp2f(sc[j] - m_new);   // 0 for masked tokens
                l_part += pe;
                // P' = p * v-scale rounded to fp16 for the MFMA; the zero-point term uses the SAME rounded value
                float ps = (float)(_Float16)(pe * vsf[j]);
                float pz = ps * vzf[j];
                if (!full && tok0 + j >= valid) {   // garbage (possibly NaN) scales of unused slots
                    ps = 0.f;
                    pz = 0.f;
                }
                corr += pz;
                psum += ps;
                pp[j] = ps;
            }
            // B operand of P.V for lane (head li < G, kg = tg): k index 8 tg + e <-> token 16 (e >> 2) + 4 tg + (e & 3); the
            // values sit in the lanes li + 4 d (GP = 4: dword d = e >> 1) resp. li, li + 8 (GP = 8) of the same row
            // (row_shl; lanes >= GP receive other heads' values or zeros: their output columns are never read)
            if constexpr (GP == 4) {
                const int pk = (int)pack_h2(pp[0], pp[1]);
                pbv[0] = (u32)pk;
                pbv[1] = (u32)__builtin_amdgcn_update_dpp(0, pk, 0x104, 0xF, 0xF, true);
                pbv[2] = (u32)__builtin_amdgcn_update_dpp(0, pk, 0x108, 0xF, 0xF, true);
                pbv[3] = (u32)__builtin_amdgcn_update_dpp(0, pk, 0x10C, 0xF, 0xF, true);
            } else {
                const int pk0 = (int)pack_h2(pp[0], pp[1]), pk1 = (int)pack_h2(pp[2], pp[3]);
                pbv[0] = (u32)pk0;
                pbv[1] = (u32)pk1;
                pbv[2] = (u32)__builtin_amdgcn_update_dpp(0, pk0, 0x108, 0xF, 0xF, true);
                pbv[3] = (u32)__builtin_amdgcn_update_dpp(0, pk1, 0x108, 0xF, 0xF, true);
            }
        }
        // ---------------- P.V : the unit's 32 tokens ----------------
        {
            const h8 pB = __builtin_bit_cast(h8, (v4u){pbv[0], pbv[1], pbv[2], pbv[3]});
            u32 raw[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                raw[jj] = *(const __attribute__((address_space(3))) u32*)(vl + (16 * (jj >> 2) + (jj & 3)) * DHB);
            }
            // V operands stay in offset form (1024 + n for the low nibbles, 1024 + 16 n for the high ones): the offsets add
            // 1024 * sum(P') to every accumulator and the high-nibble dims come out 16 x too large - both removed once at
            // the end (psum; exact: powers of two)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                u32 lo[4], hi[4];
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    const u32 W = __builtin_amdgcn_perm(raw[2 * pq + 1], raw[2 * pq], 0x0c000c00u | bb | ((4u + bb) << 16));
                    lo[pq] = and_or(W, c_lo, c_magic);
                    hi[pq] = and_or(W, c_hi, c_magic);
                }
                const h8 a_lo = __builtin_bit_cast(h8, (v4u){lo[0], lo[1], lo[2], lo[3]});
                const h8 a_hi = __builtin_bit_cast(h8, (v4u){hi[0], hi[1], hi[2], hi[3]});
                acc[2 * bb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, pB, acc[2 * bb], 0, 0, 0);
                acc[2 * bb + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, pB, acc[2 * bb + 1], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0) ; QS_LOOP_END" ::: "memory");
        if (it < 2) stamp(6 + 2 * it);
        if (has2) dma_b(vpage_next, u2 & 1, slot, valid2);
    }

    // ---- per-wave partials -> LDS.  Lane (head li, tg) holds out dims 8*(4tg + r) + e in acc[e][r] ---------------
    // (lane-derived values are re-derived here so that none of them has to survive the page loop in a register)
    const u32 lid2 = fresh_lane_id();
    const int li2 = lid2 & 15, tg2 = lid2 >> 4, tid2 = wave * 64 + (int)lid2;
    // the head's tokens were spread over the lanes li = h + GP j of the row as well
    if constexpr (GP == 4) {
        l_part += xor_lane(l_part, lid2, 4);
        corr += xor_lane(corr, lid2, 4);
        psum += xor_lane(psum, lid2, 4);
    }
    l_part += xor_lane(l_part, lid2, 8);
    corr += xor_lane(corr, lid2, 8);
    psum += xor_lane(psum, lid2, 8);
    l_part += xor_lane(l_part, lid2, 16);
    l_part += xor_lane(l_part, lid2, 32);
    corr += xor_lane(corr, lid2, 16);
    corr += xor_lane(corr, lid2, 32);
    psum += xor_lane(psum, lid2, 16);
    psum += xor_lane(psum, lid2, 32);
    // ---- service wave, off every critical path: the new token's cache write (split 0) and its own score -------------------
    // (placed here, after the code the page waves share with it: pending stores of the service wave at a control-flow
    // merge in front of the page loop would make the compiler put a vmcnt(0) there - which drains the page waves' DMA)
    if constexpr (!(EXP & 4)) {
        if (wave == SVC && !svc_free) new_token_work();
    }
    // every LDS-DMA of this wave has landed (a wave without pages never waited for its first-round fetch, and the merge
    // area below aliases the page buffers)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(12);
    __syncthreads();   // every wave is done with its page buffers: reuse s_k as the [NW][G][DH+4] fp32 merge area
    stamp(13);
    constexpr int OS = DH + 4;
    float (*s_o)[G][OS] = reinterpret_cast<float (*)[G][OS]>(&s_kv[0]);
    static_assert(sizeof(float) * NW * G * OS <= sizeof(s_kv), "merge area must fit the page buffers");
    if (li2 < G) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s_o[wave][li2][8 * (4 * tg2 + r) + e] = ((e & 1) ? acc[e][r] * 0.0625f : acc[e][r]) - (corr + ((e & 1) ? 64.f : 1024.f) * psum);
        if (tg2 == 0) {
            s_m[wave][li2] = m_run;
            s_l[wave][li2] = l_part;
        }
    }
    __syncthreads();
    stamp(14);
    for (int o = tid2; o < G * DH; o += NWT * 64) {
        const int h = o / DH, d = o % DH;
        float M = z == 0 ? s_cur[h] : -3.0e38f;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, s_m[w][h]);
        const float pc = z == 0 ? __builtin_amdgcn_exp2f(s_cur[h] - M) : 0.f;      // the new token's own term (split 0 only)
        // the new token's raw v from the LDS copy the service wave fetched at kernel entry (round 4: read from memory here it
        // was a memory round trip between the merge barrier and the result - every thread's first instruction of the tail)
        float num = pc * (float)((EXP & 4) ? vb[d] : s_vnew[d]), den = pc;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float f = __builtin_amdgcn_exp2f(s_m[w][h] - M);
            num += f * s_o[w][h][d];
            den += f * s_l[w][h];
        }
        if (nsplit == 1) {
            const _Float16 res = (_Float16)(num / (den + 1.e-6f));   // Template.hpp:1819
            if (qout) reinterpret_cast<_Float16*>(&s_meta[0][0][0])[o] = res;   // (page metadata is dead: staging area)
            else out[((size_t)b * num_heads + (size_t)hkv * G + h) * DH + d] = res;
        } else {                                                    // un-normalised partial: [O(128) | M | L]
            float* pw = ws + ((((size_t)b * num_kv_heads + hkv) * nsplit + z) * G + h) * (DH + 2);
            pw[d] = num;
            if (d == 0) {
                pw[DH] = M * 0.6931471805599453f;   // the merge kernel works in natural-log units
                pw[DH + 1] = den;
            }
        }
    }
    // ---- fused invoke_quant(_fuse_sum) of the attention output (qs_single_query_attention_quant) -----------------------
    // The per-token statistics span all KV heads = all workgroups of the sequence.  Round 4: the workgroup of the LAST KV head
    // (dispatched last: every workgroup it waits for already holds a CU or is done) does the row.  The others hand their
    // G x 128 fp16 results over as 8-byte granules {two values, tag} - ONE write-through store each, no acknowledgement, no
    // ticket; tag = the sequence's generation word + 1, read with the kernel's first scalar loads; the finisher polls the
    // granules with cache-missing loads until every tag is this launch's, runs quant_kernel's exact arithmetic on its first
    // 256 threads - same thread -> element mapping, same shuffle trees, same combination order over the four waves, so qout /
    // qscale / qsum are BIT-IDENTICAL to invoke_quant(_fuse_sum)(out) - and advances the generation word (ordered against the
    // next launch by the kernel boundary).  A granule is written by one store, so data and tag arrive together
    // (MI355X_MICROARCH.md, hand-off forms: "R2's granule needs no ordering at all"); nothing is reset, nothing can be mistaken
    // for data (NaN-proof), one hand-off round trip on the critical path.  Before (rounds 2-3): acknowledged stores -> device-
    // scope ticket -> row re-read by the last arriver, three dependent round trips (+1.0-2.2 us on the launch).
    // Workspace (qs_attn_quant_counters): [QS_ATTNQ_CAP generation words | QS_ATTNQ_CAP rows of 2048 granules], zeroed once.
    if (nsplit == 1 && qout) {
        __syncthreads();
        const int hidden = num_heads * DH;
        const unsigned tag = (unsigned)gen_raw + 1u;
        if (tid2 < G * 16) {                           // the fp16 result rows, as the plain launch writes them
            const v4u x = reinterpret_cast<const v4u*>(&s_meta[0][0][0])[tid2];
            *reinterpret_cast<v4u*>(out + ((size_t)b * num_heads + (size_t)hkv * G) * DH + tid2 * 8) = x;
        }
        v2u* const xrow = reinterpret_cast<v2u*>(qcounters + QS_ATTNQ_CAP) + (size_t)b * (QS_ATTNQ_ROW / 2);
        const int own_lo = (num_kv_heads - 1) * G * DH;                           // first row element of the finishing workgroup
        if ((G == 4 || G == 8) && !(kflags & 1024)) {
            // ---- Round 6: ALL-GATHER of the row statistics (G = 4, 8: a workgroup's G x 128 values are whole 512-element blocks of the
            // row).  Every workgroup publishes ONE 16-byte granule {amax, block sum(s), tag} of its own values - one write-through
            // store -, polls the Hkv granules of its sequence (one cache-missing 16-byte load per lane, lanes < Hkv; its own values come
            // from registers) and quantises its OWN values: 16 bytes per workgroup cross the chip instead of 1 KiB, and the row's
            // work is spread over the Hkv workgroups instead of waiting for the last one.  invoke_quant_fuse_sum's row sum is defined
            // block by block (row_ops.h reduce_max_blocksum, oracle.fused.block_order_row_sum): the block sums travel, every workgroup
            // puts them through the same final butterfly - qout / qscale / qsum are BIT-IDENTICAL to invoke_quant(_fuse_sum)(out), as before.
            // In-run against the payload form (kflags & 1024 = qs_set_attention_variant(7), kept for the other group sizes):
            // bs = 64: 20.4 -> 19.9 us at 1 033 tokens, equal at 1 535; bs = 128 (BASELINE config 3): 37.0 -> 34.5 us
            // (profiles/round6_attn_allgather.txt).  Waiting for LATER-dispatched workgroups is safe under in-order dispatch: the
            // workgroups of a sequence are consecutive block ids, so at most one sequence straddles the resident set and its waiting
            // members never hold back the slots its missing members need (everybody else finishes); every wait is bounded anyway.
            // The generation word moves on only after a successful gather - i.e. after every member has read it.
            float* const sm = reinterpret_cast<float*>(&s_kv[0]);                 // (the rings are dead)
            v4u* const gran = reinterpret_cast<v4u*>(xrow);
            if (wave == 0) {
                const int ln = (int)lid2;
                float amax = 0.f, s0 = 0.f, s1 = 0.f;
                {
                    const h8 vv = __builtin_bit_cast(h8, reinterpret_cast<const v4u*>(&s_meta[0][0][0])[ln]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float f = (float)vv[j];
                        s0 += f;
                        QS_SEQ(s0);
                        amax = fmaxf(amax, fabsf(f));
                    }
                }
                if (G == 8) {
                    const h8 vv = __builtin_bit_cast(h8, reinterpret_cast<const v4u*>(&s_meta[0][0][0])[64 + ln]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float f = (float)vv[j];
                        s1 += f;
                        QS_SEQ(s1);
                        amax = fmaxf(amax, fabsf(f));
                    }
                }
                amax = wave_max(amax);
                s0 = wave_sum(s0);
                if (G == 8) s1 = wave_sum(s1);
                v4u mine = {__builtin_bit_cast(u32, amax), __builtin_bit_cast(u32, s0), __builtin_bit_cast(u32, s1), tag};
                if (ln == 0 && !((kflags & 64) && b == 0 && hkv == 0) && !(kflags & 512)) {
                    v4u* const dst = gran + hkv;
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(mine) : "memory");
                }
                const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(gran, 0, num_kv_heads * 16, 0x00020000);
                const bool peer = ln < num_kv_heads && ln != hkv;
                const u32 goff = ln < num_kv_heads ? (u32)ln * 16u : 0xFFFFFF00u;
                const int spin_cap = (kflags & 256) ? 1 : (kflags & 64) ? 4096 : QS_SPIN_CAP;
                int polls = 0;
                v4u g;
                for (;;) {
                    g = __builtin_amdgcn_raw_buffer_load_b128(grs, goff, 0, 17);
                    if (!__builtin_amdgcn_ballot_w64(peer && g.w != tag)) break;
                    if (++polls >= spin_cap) {
                        if (ln == 0 && !(kflags & 256))
                            atomicOr(qcounters + QS_ATTNQ_CAP + (size_t)QS_ATTNQ_CAP * QS_ATTNQ_ROW, QS_ERR_ATTN_HANDOVER);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (ln == hkv) g = mine;                                           // own values from registers, not from memory
                // (scalar copies first: __builtin_bit_cast of a vector-element lvalue reads element 0)
                const u32 gx = g.x, gy = g.y, gz = g.z;
                float am = ln < num_kv_heads ? __builtin_bit_cast(float, gx) : 0.f;
                am = wave_max(am);
                // the row sum: one more wave butterfly over the block sums, lane b = block b of the row, -0.0 beyond it (row_ops.h
                // reduce_max_blocksum).  G = 4: block j is workgroup j's; G = 8: blocks 2 j, 2 j + 1
                float bs;
                if (G == 4) {
                    bs = ln < num_kv_heads ? __builtin_bit_cast(float, gy) : -0.0f;
                } else {
                    const int src = (ln >> 1) < num_kv_heads ? (ln >> 1) : 0;
                    const int yv = __builtin_amdgcn_ds_bpermute(4 * src, (int)gy), zv = __builtin_amdgcn_ds_bpermute(4 * src, (int)gz);
                    bs = (ln >> 1) < num_kv_heads ? __builtin_bit_cast(float, (ln & 1) ? zv : yv) : -0.0f;
                }
                const float tot = wave_sum(bs);
                if (ln == 0) {
                    sm[0] = am;
                    sm[1] = tot;
                }
            }
            __syncthreads();
            const float r = sm[0];
            if (tid2 < G * 16) {
                const h8 vv = __builtin_bit_cast(h8, reinterpret_cast<const v4u*>(&s_meta[0][0][0])[tid2]);
                float f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (float)vv[j];
                qs_store_q8(qout + (size_t)b * hidden + (size_t)hkv * G * DH + tid2 * 8, f, 127.0f / r);
            }
            if (hkv == num_kv_heads - 1 && tid2 == 0) {
                qscale[b] = __float2half_rn(r / 127.0f);
                if (qrowsum) qrowsum[b] = __float2half_rn(sm[1]);
                qcounters[b] = tag;
            }
        } else if (hkv != num_kv_heads - 1) {
            // (kflags & 64: the armed one-shot fault of qs_debug_inject_fault - sequence 0's KV head 0 never delivers)
            // (kflags & 512: timing libraries only - nobody publishes: with 256 below, what the hand-over costs apart from the wait)
            for (int e = tid2; e < G * DH / 2 && !((kflags & 64) && b == 0 && hkv == 0) && !(kflags & 512); e += NWT * 64) {
                v2u gr;
                gr.x = reinterpret_cast<const u32*>(&s_meta[0][0][0])[e];
                gr.y = tag;
                v2u* const dst = xrow + (size_t)hkv * (G * DH / 2) + e;
                asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(gr) : "memory");
            }
        } else {
            // cache-missing loads through the buffer builtin (the compiler counts them; an asm load inside a retry loop is not
            // safe - gemm_w4a8_ring.hip, the K-slice seam)
            const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(xrow, 0, QS_ATTNQ_ROW * 4, 0x00020000);
            v4u raw[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            if (tid2 < 256) {
                // quant_kernel: thread t owns row elements (c 256 + t) 8 .. + 7, c = 0, 1 = granules i/2 .. i/2 + 3 = 32 bytes of the
                // exchange row.  All four requests of a poll go out together, whatever the thread needs of them: a chunk that
                // belongs to this workgroup's own heads (or lies beyond the row) is requested at an offset beyond the buffer - no
                // memory access, zeros - instead of being skipped under an exec mask (with one masked block per chunk the compiler
                // waited for chunk 0 before it requested chunk 1: two dependent round trips per poll)
                const int i0 = tid2 * 8, i1 = (256 + tid2) * 8;
                const bool p0 = i0 < own_lo, p1 = i1 < own_lo;
                const u32 o0 = p0 ? (u32)i0 * 4u : 0xFFFFFF00u, o1 = p1 ? (u32)i1 * 4u : 0xFFFFFF00u;
                // BOUNDED (round 5): after spin_cap polls the wave stops waiting, sets QS_ERR_ATTN_HANDOVER in the device error word
                // (the word behind the exchange rows) and quantises what it has - a wrong row and a status bit instead of a hung
                // GPU.  The generation still advances, so the NEXT launch is clean by itself (a late granule carries a stale tag).
                // (kflags & 256: timing libraries only, WRONG RESULTS - the finisher takes whatever its first poll returns: the launch
                //  without the wait for the other KV heads' workgroups, profiles/round6_attn_handover.txt)
                const int spin_cap = (kflags & 256) ? 1 : (kflags & 64) ? 4096 : QS_SPIN_CAP;
                int polls = 0;
                for (;;) {
                    const v4u g00 = __builtin_amdgcn_raw_buffer_load_b128(xrs, o0, 0, 17);
                    const v4u g01 = __builtin_amdgcn_raw_buffer_load_b128(xrs, o0, 16, 17);
                    const v4u g10 = __builtin_amdgcn_raw_buffer_load_b128(xrs, o1, 0, 17);
                    const v4u g11 = __builtin_amdgcn_raw_buffer_load_b128(xrs, o1, 16, 17);
                    const int m0 = (g00.y != tag) | (g00.w != tag) | (g01.y != tag) | (g01.w != tag);
                    const int m1 = (g10.y != tag) | (g10.w != tag) | (g11.y != tag) | (g11.w != tag);
                    raw[0] = (v4u){g00.x, g00.z, g01.x, g01.z};
                    raw[1] = (v4u){g10.x, g10.z, g11.x, g11.z};
                    if (!__builtin_amdgcn_ballot_w64((p0 && m0) || (p1 && m1))) break;
                    if (++polls >= spin_cap) {
                        if ((tid2 & 63) == 0 && !(kflags & 256))
                            atomicOr(qcounters + QS_ATTNQ_CAP + (size_t)QS_ATTNQ_CAP * QS_ATTNQ_ROW, QS_ERR_ATTN_HANDOVER);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!p0 && i0 < hidden) raw[0] = reinterpret_cast<const v4u*>(&s_meta[0][0][0])[(i0 - own_lo) >> 3];
                if (!p1 && i1 < hidden) raw[1] = reinterpret_cast<const v4u*>(&s_meta[0][0][0])[(i1 - own_lo) >> 3];
            }
            float* const sm = reinterpret_cast<float*>(&s_kv[0]);                // 2 x 4 floats (the rings are dead)
            float amax = 0.f, sum = 0.f;
            if (tid2 < 256) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int i = (c * 256 + tid2) * 8;
                    if (i < hidden) {
                        const h8 vv = __builtin_bit_cast(h8, raw[c]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float f = (float)vv[j];
                            sum += f;
                            amax = fmaxf(amax, fabsf(f));
                        }
                    }
                }
                amax = wave_max(amax);
                if (qrowsum) sum = wave_sum(sum);
                if ((tid2 & 63) == 0) {
                    sm[tid2 >> 6] = amax;
                    sm[4 + (tid2 >> 6)] = sum;
                }
            }
            __syncthreads();
            if (tid2 < 256) {
                float r = sm[0], s2 = qrowsum ? sm[4] : 0.f;
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    r = fmaxf(r, sm[w]);
                    if (qrowsum) s2 = s2 + sm[4 + w];
                }
                if (tid2 == 0) {
                    qscale[b] = __float2half_rn(r / 127.0f);                     // fused_kernels.cu:72
                    if (qrowsum) qrowsum[b] = __float2half_rn(s2);                     // :121
                    qcounters[b] = tag;                                          // the next launch hands over under tag + 1
                }
                const float mul = 127.0f / r;                                     // :78 (unrounded fp32 amax)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int i = (c * 256 + tid2) * 8;
                    if (i < hidden) {
                        const h8 vv = __builtin_bit_cast(h8, raw[c]);
                        float f[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) f[j] = (float)vv[j];
                        qs_store_q8(qout + (size_t)b * hidden + i, f, mul);
                    }
                }
            }
        }
    }
    stamp(15);
    if constexpr (EXP & 32) {
        if (lane < 16)
            reinterpret_cast<unsigned long long*>(ws)[(((size_t)b * gridDim.x + hkv) * NWT + wave) * 16 + lane] = s_stamps[wave * 16 + lane];
    }
}

// second phase of split-KV: one workgroup per (sequence, query head) combines the nsplit partials
__global__ __launch_bounds__(DH) void decode_attention_merge_kernel(const float* __restrict__ ws, _Float16* __restrict__ out,
                                                                    int num_heads, int num_kv_heads, int G, int nsplit) {
    const int b = blockIdx.y, hq = blockIdx.x, d = threadIdx.x;
    const int hkv = hq / G, h = hq % G;
    const float* base = ws + (((size_t)b * num_kv_heads + hkv) * nsplit * G + h) * (DH + 2);
    const size_t zs = (size_t)G * (DH + 2);
    float M = -3.0e38f;
    for (int zz = 0; zz < nsplit; ++zz) M = fmaxf(M, base[zz * zs + DH]);
    float num = 0.f, den = 0.f;
    for (int zz = 0; zz < nsplit; ++zz) {
        const float f = __expf(base[zz * zs + DH] - M);
        num += f * base[zz * zs + d];
        den += f * base[zz * zs + DH + 1];
    }
    out[((size_t)b * num_heads + hq) * DH + d] = (_Float16)(num / (den + 1.e-6f));
}

// cos/sin table [max_pos][64] (float2), every entry double-evaluated and rounded once to float32: bit-identical to the
// in-kernel path and to the oracle (oracle/kvattn.py rope_coef)
__global__ void rope_table_kernel(float2* __restrict__ tab, int max_pos, float base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= max_pos * 64) return;
    const RopeCS cs = rope_coef(i & 63, i >> 6, base, DH);
    tab[i] = make_float2(cs.c, cs.s);
}

struct RopeTable {
    float2* tab = nullptr;
    int len = 0;
    float base = 0.f;
};
// Tables are keyed by (device, base) and NEVER freed or replaced once handed out: a captured hipGraph may hold their
// address.  A longer table for the same base is a new entry (the old one stays alive for the graphs that captured it).
constexpr int ROPE_SLOTS = 8;
RopeTable g_rope[16][ROPE_SLOTS];

}  // namespace

// Library-managed RoPE table (per device, per base).  Returns nullptr when it cannot be (re)built right now, e.g.
// while the stream is being captured into a graph before the first eager call; callers then compute in-kernel.
const float2* qs_rope_table(float base, int max_pos, hipStream_t st, int* len_out) {
    static std::mutex g_rope_mutex;   // host state of the slots (several host threads may drive their own bound streams)
    std::lock_guard<std::mutex> lock(g_rope_mutex);
    int dev = 0;
    *len_out = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (max_pos > 32768) max_pos = 32768;
    if (max_pos <= 0) return nullptr;
    RopeTable* slot = nullptr;
    for (int i = 0; i < ROPE_SLOTS; ++i) {
        RopeTable& r = g_rope[dev][i];
        if (r.tab && r.base == base && r.len >= max_pos) {
            *len_out = r.len;
            return r.tab;
        }
        if (!r.tab && !slot) slot = &r;
    }
    if (!slot) return nullptr;   // every slot taken by other (base, length) pairs: callers compute in-kernel
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)max_pos * 64 * sizeof(float2)) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    // The table is shared by every stream of the device: fill it and WAIT before publishing it (a launch on another stream
    // issued right after this call must not read it half-built; the eager first build happens once per (device, base)).
    hipLaunchKernelGGL(rope_table_kernel, dim3((max_pos * 64 + 255) / 256), dim3(256), 0, st, reinterpret_cast<float2*>(p),
                       max_pos, base);
    if (hipStreamSynchronize(st) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return nullptr;
    }
    slot->len = max_pos;
    slot->base = base;
    slot->tab = reinterpret_cast<float2*>(p);
    *len_out = slot->len;
    return slot->tab;
}

// Library-managed split-KV workspace: ONE fixed-size allocation per device, made on the first eager call that needs
// it and never freed, moved or grown (captured hipGraphs keep its address).  Requests beyond its capacity, or a first
// request that arrives while the stream is being captured, return nullptr: callers then run with fewer / no splits.
namespace {
constexpr size_t SPLIT_WS_BYTES = (size_t)32 << 20;   // heuristic needs < 1024 (workgroup, split) pairs x 8 heads x 520 B
struct SplitWs {
    float* p = nullptr;
};
SplitWs g_ws[16][QS_MAX_STREAM_SLOTS];   // [device][scratch slot] (common.h)
}  // namespace
// second phase of split-KV, shared with the KV8 kernel (attention_mfma8.hip)
void qs_launch_attention_merge(const float* ws, _Float16* out, int H, int Hkv, int G, int nsplit, int batch,
                               hipStream_t st) {
    hipLaunchKernelGGL(decode_attention_merge_kernel, dim3(H, batch), dim3(DH), 0, st, ws, out, H, Hkv, G, nsplit);
}
float* qs_split_workspace(size_t bytes, hipStream_t st) {
    int dev = 0;
    if (bytes > SPLIT_WS_BYTES || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SplitWs& w = g_ws[dev][qs_scratch_slot(st)];
    if (w.p) return w.p;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    void* p = nullptr;
    if (hipMalloc(&p, SPLIT_WS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    w.p = reinterpret_cast<float*>(p);
    return w.p;
}
size_t qs_split_workspace_capacity() { return SPLIT_WS_BYTES; }

// Split-KV factor of the matrix-core decode kernels (KV4: this file, KV8: attention_mfma8.hip): the number of workgroups a
// (sequence, KV head) pair's page range is cut into.  Cost model fitted to a sweep on the MI355X (profiles/round5_split_sweep*.txt:
// 1 .. 48 sequences x 8 KV heads, 1 030 .. 7 700 tokens, 1 .. 32 splits; mean regret 0.7 %, worst 10 %, against the forced best;
// the round-2 rule "aim at >= 512 workgroups" it replaces: 8 % / 39 %):
//   time(n) = rounds * F + max(rounds * pages / n * t_cu,  blocks * pages * t_hbm) + merge(n)
// rounds = ceil(blocks * n / 256): workgroups land on the 256 CUs round-robin and a CU streams at its own request budget whatever
// the number of resident workgroups, so 320 workgroups take as long as 512; F = a workgroup's head + tail; t_cu = one CU's time
// per page (64 tokens of K and V), t_hbm = the chip's; merge = the second launch (boundary + n partials per head).
// ns / ps units, integer arithmetic (the plan is part of the ABI: tests/test_dispatch_plan.py pins it).
int qs_attn_choose_splits(int blocks, int pages, int kv8, int fused_quant) {
    if (blocks >= 512 || pages < 4) return 1;
    const long F = kv8 ? 5000 : 4000, t_cu = kv8 ? 450 : 340, t_hbm_ps = kv8 ? 2540 : 1330;
    long best = -1;
    int best_n = 1;
    // (n <= 8: within this kernel's range - page tables of <= 192 entries, 12 288 tokens; longer tables go to the VALU kernels -
    //  more splits do not pay: 1 / 2 / 4 sequences at 8 191 and 12 200 tokens, forced 4 / 8 / 12 / 16 / 24 splits against this
    //  choice, KV4 and KV8: the choice is the best or within 1 % of it everywhere except one sequence of 12 200 tokens over an INT8
    //  cache, where 12 splits measure 20.4 against 21.7 us - profiles/round6_split_long.txt.  ADVICE r05 extrapolated the cost
    //  formula to 512 pages, which this function is never asked about.)
    for (int n = 1; n <= 8; ++n) {
        if (n > 1 && pages / n < 2) break;
        const long rounds = ((long)blocks * n + 255) / 256;
        const long per_cu = rounds * pages * t_cu / n, chip = (long)blocks * pages * t_hbm_ps / 1000;
        // (fused_quant: qs_single_query_attention_quant - only the un-split launch can finish the row itself; a split one is
        // followed by the quantiser as a launch of its own, 3.2 us + boundary)
        const long cost = rounds * F + (per_cu > chip ? per_cu : chip) + (n > 1 ? 2500 + 300 * n + (fused_quant ? 3500 : 0) : 0);
        if (best < 0 || cost < best) best = cost, best_n = n;
    }
    return best_n;
}
// timing tool (scripts/trace_attn.py): copy the first `bytes` of the split workspace (the EXP & 32 timeline stamps) to dst
extern "C" int qs_debug_copy_split_workspace(void* dst, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || !g_ws[dev][0].p || bytes > SPLIT_WS_BYTES) return QS_EINVAL;
    return (int)hipMemcpy(dst, g_ws[dev][0].p, bytes, hipMemcpyDeviceToDevice);      // (the shared slot: the trace tools bind nothing)
}

// workspace of the attention + quant fusion: one fixed allocation per device - QS_ATTNQ_CAP generation words followed by
// QS_ATTNQ_CAP exchange rows of QS_ATTNQ_ROW / 2 granules (8 bytes: two fp16 values + the generation tag) -, zeroed once
// (generation 0, every tag stale), never freed; nullptr for larger batches or while it cannot be allocated (first use inside
// a stream capture): the caller then runs the un-fused pair
namespace {
unsigned* g_qcounters[16][QS_MAX_STREAM_SLOTS];
}
unsigned* qs_attn_quant_counters(hipStream_t st, int batch) {
    int dev = 0;
    if (batch > QS_ATTNQ_CAP || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    unsigned*& slot_p = g_qcounters[dev][qs_scratch_slot(st)];
    if (slot_p) return slot_p;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    void* p = nullptr;
    const size_t bytes = (size_t)QS_ATTNQ_CAP * 4 + (size_t)QS_ATTNQ_CAP * QS_ATTNQ_ROW * 4 + 64;   // + the device error word
    // (the memset runs on the NULL stream: synchronise, or a launch on a non-blocking stream could overtake it)
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    slot_p = reinterpret_cast<unsigned*>(p);
    return slot_p;
}

unsigned* qs_attn_error_word(int slot) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || slot < 0 || slot >= QS_MAX_STREAM_SLOTS || !g_qcounters[dev][slot])
        return nullptr;
    return g_qcounters[dev][slot] + QS_ATTNQ_CAP + (size_t)QS_ATTNQ_CAP * QS_ATTNQ_ROW;
}
bool qs_attn_scratch_prealloc(hipStream_t st) {
    const bool a = qs_split_workspace(1, st) != nullptr;
    const bool b = qs_attn_quant_counters(st, 1) != nullptr;
    return a && b;
}
int qs_attn_reset_handoff() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return QS_OK;
    const size_t bytes = (size_t)QS_ATTNQ_CAP * 4 + (size_t)QS_ATTNQ_CAP * QS_ATTNQ_ROW * 4 + 64;
    for (int slot = 0; slot < QS_MAX_STREAM_SLOTS; ++slot) {
        if (!g_qcounters[dev][slot]) continue;
        hipError_t e = hipMemset(g_qcounters[dev][slot], 0, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            qs_set_error("qs_device_reset (attention): %s", hipGetErrorString(e));
            return (int)e;
        }
    }
    return QS_OK;
}

// called from attention.hip's dispatcher for KV4.  force_split: 0 = heuristic, n > 0 = exactly n splits (tests)
int qs_launch_decode_mfma(int G, dim3 grid, hipStream_t st, const _Float16* q, const _Float16* k, const _Float16* v,
                          const int64_t* kvp, const int* len, _Float16* out, int H, int Hkv, int64_t qs, int64_t kvs,
                          int mb, int timestep, float base, int max_pos, int force_split, int kflags) {
    if (G < 1 || G > 8) {
        qs_set_error("single_query_attention: num_heads/num_kv_heads = %d not in 1..8", G);
        return QS_ENOSUP;
    }
    int exp_flags = 0;                      // qs_set_attention_variant(200 + EXP): ablation builds of the G = 4 kernel
    if (force_split >= 100) {
        exp_flags = force_split - 100;
        force_split = 0;
    }
    int tab_len = 0;
    const float2* tab = g_qs_attn_plan.active ? nullptr : qs_rope_table(base, max_pos, st, &tab_len);
    // split-KV (qs_attn_choose_splits above)
    const int blocks = (int)(grid.x * grid.y);
    const int pages_max = (timestep + PAGE_TOK - 1) / PAGE_TOK;
    int nsplit = force_split > 0 ? force_split : qs_attn_choose_splits(blocks, pages_max, 0, g_qs_attn_quant.qout != nullptr && H * DH <= 4096);
    if (g_qs_attn_plan.active) {
        g_qs_attn_plan.family = 1, g_qs_attn_plan.nsplit = nsplit, g_qs_attn_plan.waves = NW;
        return QS_OK;
    }
    float* ws = nullptr;
    if (nsplit > 1) {
        const size_t per_split = (size_t)blocks * G * (DH + 2) * sizeof(float);
        if (per_split * nsplit > qs_split_workspace_capacity()) nsplit = (int)(qs_split_workspace_capacity() / per_split);
        ws = nsplit > 1 ? qs_split_workspace(per_split * nsplit, st) : nullptr;
        if (!ws) nsplit = 1;
    }
    grid.z = nsplit;
    // attention + invoke_quant fusion (qs_single_query_attention_quant): un-split launches over rows of <= 4096 values
    // (quant_kernel's 256-thread mapping is what the finishing workgroup reproduces bit for bit)
    int8_t* qout = nullptr;
    __half *qscale = nullptr, *qsum = nullptr;
    unsigned* qcnt = nullptr;
    // (qs_set_row_sum_order(1) with a row sum asked for: the finisher reproduces THIS library's order of invoke_quant_fuse_sum, not the
    //  reference's - the pair is issued instead, so that the fused entry stays bit-identical to the two calls in both modes)
    const bool sum_order_ok = !(qs_get_row_sum_order() && g_qs_attn_quant.qsum);
    if (g_qs_attn_quant.qout && sum_order_ok && nsplit == 1 && H * DH <= 4096 && (qcnt = qs_attn_quant_counters(st, (int)grid.y))) {
        qout = g_qs_attn_quant.qout;
        qscale = reinterpret_cast<__half*>(g_qs_attn_quant.qscale);
        qsum = reinterpret_cast<__half*>(g_qs_attn_quant.qsum);
        g_qs_attn_quant.done = 1;
        if (g_inject_fault & 2) kflags |= 64, g_inject_fault &= ~2;   // one-shot (qs_debug_inject_fault)
    }
#define QS_LAUNCH_G(GG)                                                                                             \
    hipLaunchKernelGGL((decode_attention_mfma_kernel<GG>), grid, dim3(NWT * 64), 0, st, q, k, v, kvp, len, out, H, Hkv, \
                       qs, kvs, mb, timestep, base, tab, tab_len, nsplit, ws, qout, qscale, qsum, qcnt, kflags)
#define QS_LAUNCH_EXP(E)                                                                                              \
    case E:                                                                                                            \
        hipLaunchKernelGGL((decode_attention_mfma_kernel<4, E>), grid, dim3(NWT * 64), 0, st, q, k, v, kvp, len, out, H,  \
                           Hkv, qs, kvs, mb, timestep, base, tab, tab_len, nsplit, ws, qout, qscale, qsum, qcnt, kflags); \
        return qs_launch_status("single_query_attention")
#ifdef QS_TIMING   // ablation / trace instantiations (some are wrong by design): not in the shipped library
    if (const char* e = getenv("QS_ATTN_KFLAGS")) kflags = atoi(e);   // experiment switches under a trace variant
    if (exp_flags & 32) {                     // timeline trace: stamps go to the (otherwise unused) split workspace
        ws = qs_split_workspace((size_t)blocks * NWT * 16 * 8, st);
        if (!ws) exp_flags = 0;
    }
    if (G == 4 && exp_flags > 0 && nsplit == 1) {
        switch (exp_flags) {
            QS_LAUNCH_EXP(1);
            QS_LAUNCH_EXP(2);
            QS_LAUNCH_EXP(4);
            QS_LAUNCH_EXP(6);
            QS_LAUNCH_EXP(32);
            default: break;
        }
    }
#endif
    switch (G) {
        case 1: QS_LAUNCH_G(1); break;
        case 2: QS_LAUNCH_G(2); break;
        case 3: QS_LAUNCH_G(3); break;
        case 4: QS_LAUNCH_G(4); break;
        case 5: QS_LAUNCH_G(5); break;
        case 6: QS_LAUNCH_G(6); break;
        case 7: QS_LAUNCH_G(7); break;
        case 8: QS_LAUNCH_G(8); break;
        default:
            qs_set_error("single_query_attention: num_heads/num_kv_heads = %d not in 1..8", G);
            return QS_ENOSUP;
    }
#undef QS_LAUNCH_G
    if (nsplit > 1)
        hipLaunchKernelGGL(decode_attention_merge_kernel, dim3(H, grid.y), dim3(DH), 0, st, ws, out, H, Hkv, G, nsplit);
    return qs_launch_status("single_query_attention");
}
