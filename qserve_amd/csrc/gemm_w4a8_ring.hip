// gemm_w4a8_ring.hip -- decode-shape W4A8 GEMM (M <= 64 tokens per workgroup): weight streaming through LDS-DMA rings
// with operand reads software-pipelined one stage ahead of the matrix cores.
//
// Same arithmetic, operand mapping and epilogue as the other W4A8 kernels (gemm_w4a8.hip header; reference kernels
// kernels/csrc/qgemm/w4a8_per_chn/gemm_cuda.cu:303-594, w4a8_per_group/gemm_cuda.cu:328-628).  Measured on MI355X
// (qs_set_gemm_variant 3100+ experiments): the earlier decode kernels spend most of their time NOT streaming - one
// wave per SIMD, LDS-read latency exposed after every barrier, bank-conflicted weight reads.  Here:
//   * a workgroup = 8 wave64 = KG K-groups x WN channel units (64 channels each); a group's WN waves share the
//     activation tile of their stage; groups take the 64-wide k-stages round-robin (stage u belongs to group u % KG),
//     so together they read every weight row contiguously;
//   * the activation tiles of groups 2h and 2h+1 (stages u, u+1 = one 128-byte line per token row) are fetched
//     TOGETHER, 8 rows x 128 B per DMA instruction, into one image the two groups share: fetched per stage, every line
//     would be requested twice from L2 (half a line each time) by every workgroup, and with all CUs doing so the L2s
//     deliver 12.5 TB/s instead of 25-29 TB/s (scripts/microbench_cufill.hip: +15-20 % on the decode mix);
//   * every byte goes HBM/L2 -> LDS by LDS-DMA (asm global_load_lds, 16 B per lane) into an NS-deep ring per group;
//     one raw s_barrier per round; counted s_waitcnt vmcnt keeps NS-2 stages per group in flight across it;
//   * while the MFMAs of stage i run, the wave already reads stage i+1's activation operands and weight nibbles from
//     LDS and unpacks them (registers double-buffered by unrolling the round loop by two), and issues its share of
//     the DMA for stage i+NS-1;
//   * LDS images are bank-conflict free (permutation applied on the DMA source side, as in gemm_w4a8_tiled.hip);
//   * the KG partial int32 tiles are summed exactly through LDS, the fp32 epilogue is fused, its scale operands are
//     requested before the reduction, and the fp16 tile leaves through LDS as whole 128-byte rows.
#include "common.h"
#include <type_traits>

// A/B switches (qs_set_gemm_variant(5000 + bits); every setting computes the same results):
//   1 = weight DMA with the default cache policy everywhere, 2 = non-temporal everywhere (default: non-temporal unless several
//       token blocks share the weight bytes of a channel block and N <= 8192 - see launch_ring and `issue`)
//   4 / 8 = (QS_TIMING libraries only) no cross-group reduction / leave behind the k loop: the price of the tail (round 5)
//   32 / 64 = (libraries built with -DQS_TIMING only; ignored by the shipped library) timing only, WRONG RESULTS: no MFMA /
//        no LDS operand reads (what is left is the DMA + barrier pipeline:
//        gate_up at M = 64 17.5 us -> 16.3 / 16.8, both off 15.9 us = 4.2 us of head and tail + 512 KB per CU at 44 GB/s,
//        the rate scripts/microbench_cufill.hip measures for this L2 + HBM mix with nothing else going on)
//   16 = (set by the launcher, not by the variant) per-channel epilogue convention qs_set_gemm_epilogue(1): fmaf form
//   128 = (set by the launcher) the armed one-shot fault of qs_debug_inject_fault: tile 0's K-slice producers do not deliver
//   256 * d = ring depth d (3..6, if 160 KiB allow): sensitivity to the bytes in flight
//   2048 = (set by the launcher) K slices across the XCDs (ring_coords); 4096 = keep every slice of a channel block on one XCD
//        (the mapping of rounds 3-5; A/B of the activation traffic)
//   8192 = (QS_TIMING libraries only, WRONG RESULTS) per-group launches without the level-2 dequant arithmetic
qs_flag g_ring_flags = 0;

namespace {

// epilogue operands staged in LDS (see ring_body): the top 2 KiB of the 160 KiB window
constexpr int SC_BYTES = 2048;
constexpr int SC_OFF = 160 * 1024 - SC_BYTES;


typedef __attribute__((address_space(3))) void* lptr_t;
typedef u32 v2u __attribute__((ext_vector_type(2)));

// QS_RING_TRACE builds (scripts/trace_gemm.py; never the shipped library): per-wave timeline into a caller buffer
#ifdef QS_RING_TRACE
__device__ unsigned long long* g_ring_trace = nullptr;
#define QS_STAMP(i)                                                                          \
    do {                                                                                     \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                          \
        if (g_ring_trace && lane == 0) g_ring_trace[((size_t)blockIdx.x * 8 + wave) * 16 + (i)] = t_; \
    } while (0)
#define QS_ACC(var, t0_) var += (unsigned)(__builtin_amdgcn_s_memtime() - (t0_))
#else
#define QS_STAMP(i) \
    do {            \
    } while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void raw_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// bytes of one K-group's ring slot: activations | weights | per-group meta (kernel and launcher must agree)
template <int MT, int WN, int MODE>
constexpr int ring_gstage() {
    return 16 * MT * 64 + WN * 2048 + (MODE == 1 ? 2 * (WN > 2 ? 256 : 128) : 0);
}

// workgroup -> (channel block, token block, K slice); token blocks of one channel block sit on ONE XCD (b % 8) so that the
// weights are fetched from HBM once and re-served by that XCD's L2; K slices (ksplit > 1) likewise
struct RingCoords {
    int nblk, mblk, kq;
};
// wave-uniform unsigned division by a small run-time divisor that is almost always a power of two (token blocks, K slices): a
// shift then; the general form costs ~35 dependent scalar / vector instructions each - three of them stood in front of a
// workgroup's first memory request (round 4, timeline trace: 1 200-1 400 cycles from kernel entry to the first request)
__device__ __forceinline__ unsigned udivmod(unsigned x, unsigned d, unsigned& rem) {
    if ((d & (d - 1)) == 0) {
        const unsigned sh = (unsigned)__builtin_ctz(d);
        rem = x & (d - 1);
        return x >> sh;
    }
    rem = x % d;
    return x / d;
}
__device__ __forceinline__ RingCoords ring_coords(int b, int N, int wn, int mblocks, int ksplit, bool kxcd) {
    RingCoords c = {b, 0, 0};
    const int per = mblocks * ksplit;                 // workgroups per channel block
    if (per > 1) {
        if (kxcd) {
            // K slices ACROSS the XCDs (round 6; launch_ring sets it when ksplit is 2 / 4 / 8 and the channel blocks divide): XCD x
            // = b % 8 serves K slice x / xs (xs = 8 / ksplit XCDs per slice) of the channel blocks j = x % xs (mod xs), every
            // token block of a channel block on the same XCD as before (its weights come from that L2) - so an L2 holds only ITS
            // K slice of the activation matrix instead of all of it: down_proj's [64, 14336] int8 matrix was fetched once per XCD
            // (8 x 0.9 MB = 24 % on top of the 30.8 MB the GEMM needs, profiles/round5_b_pmc_traffic.json).  Inside a group of 8
            // consecutive workgroups the slice index grows with b: the seam's finisher (the last slice) is still dispatched last.
            const unsigned xs = 8u / (unsigned)ksplit, x = (unsigned)b & 7u;
            unsigned mb;
            const unsigned gi = udivmod((unsigned)b >> 3, (unsigned)mblocks, mb);
            c.kq = (int)(x / xs);
            c.nblk = (int)(gi * xs + x % xs);
            c.mblk = (int)mb;
            return c;
        }
        const int n8 = (N / (64 * wn)) & ~7;          // channel blocks covered by whole groups of 8 (one per XCD)
        unsigned sub, mb;
        if (b < n8 * per) {
            const int slot = b >> 3;
            c.nblk = (int)udivmod((unsigned)slot, (unsigned)per, sub) * 8 + (b & 7);
        } else {                                      // remainder (< 8 channel blocks): plain order
            const int r = b - n8 * per;
            c.nblk = n8 + (int)udivmod((unsigned)r, (unsigned)per, sub);
        }
        c.kq = (int)udivmod(sub, (unsigned)mblocks, mb);
        c.mblk = (int)mb;
    }
    return c;
}

// The GEMM proper.
template <int MT, int WN, int MODE, int OUTK, bool KSPLIT>
__device__ __forceinline__ void ring_body(const int8_t* __restrict__ A, const uint8_t* __restrict__ W,
                                         const int8_t* __restrict__ zeros, const int8_t* __restrict__ scales8,
                                         const __half* __restrict__ wscales, const __half* __restrict__ ascales,
                                         const __half* __restrict__ wszs, const __half* __restrict__ assums,
                                         void* __restrict__ out, int M, int N, int K, int mblocks, int ns, int ksplit_arg,
                                         int* __restrict__ slabs, unsigned* __restrict__ counters, int flags,
                                         uint8_t* smem) {
    const int ksplit = KSPLIT ? ksplit_arg : 1;
    constexpr int KG = 8 / WN;
    constexpr int ASTAGE = 16 * MT * 64;              // activation bytes per stage (64 k)
    constexpr int WSTAGE = WN * 2048;                 // packed weight bytes per stage
    constexpr int MZ = WN > 2 ? 256 : 128;            // per-group meta of a stage: scales of the WN units | zeros at + MZ
    constexpr int MSTAGE = MODE == 1 ? 2 * MZ : 0;
    constexpr int GSTAGE = ring_gstage<MT, WN, MODE>();  // ring slot of one group
    static_assert(GSTAGE == ASTAGE + WSTAGE + MSTAGE, "slot layout");
    constexpr int NPIECE = MT + 2 * WN;               // 1 KiB DMA pieces per group and stage
    static_assert(NPIECE % WN == 0, "pieces must split evenly over the waves of a group");
    constexpr int NDMA = NPIECE / WN + (MODE == 1 ? 1 : 0);   // VMEM instructions per wave and stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / WN, wn = wave % WN;
    QS_STAMP(0);
    const int li = lane & 15, g = lane >> 4;
    const int tsel = li >> 3, c = li & 7;
    // workgroup -> (channel block, token block); token blocks of one channel block sit on ONE XCD (b % 8) so that the
    // weights are fetched from HBM once and re-served by that XCD's L2
    // K split (ksplit > 1): the K range is cut into ksplit slices handled by different workgroups (same XCD as well);
    // the int32 partial tiles meet in a workspace, the last-dispatched slice finishes (see the seam below).
    const RingCoords rc = ring_coords(blockIdx.x, N, WN, mblocks, ksplit, KSPLIT && (flags & 2048));
    const int nblk = rc.nblk, mblk = rc.mblk, kq = rc.kq;
    const int unit0 = nblk * WN;                      // first 64-channel unit of the workgroup
    // OUTK == 2 (gate_up + silu * mul): N stacks [gate | up] (N/2 channels each); "unit" j then means gate channels
    // 32 j .. + 31 as its 32-channel tile row t = 0 and up channels N/2 + 32 j .. as t = 1 - both values of an output
    // element end up in one wave (lanes l and l + 32), and the workgroup writes 32 WN channels of the [M, N/2] result
    constexpr bool ACT = OUTK == 2;
    static_assert(!(ACT && KSPLIT), "the activation epilogue exists in the un-split form only");
    auto trow = [&](int unit, int t) { return ACT ? (t ? N / 64 + unit : unit) : unit * 2 + t; };          // 32-channel tile row
    auto chan32 = [&](int unit, int t) { return ACT ? (t ? N / 2 + 32 * unit : 32 * unit) : unit * 64 + 32 * t; };
    const int m0 = mblk * (16 * MT);
    const int KT = K >> 5;
    unsigned nst_rem;
    const int nst = (int)udivmod((unsigned)(K >> 6), (unsigned)ksplit, nst_rem);   // 64-k stages of this workgroup's K slice
    const int u0 = kq * nst;                          // first global stage
    const int nloc = nst / KG;                        // stages per group (dispatcher: (K/64/ksplit) % KG == 0)

    uint8_t* const ring = smem + kg * ns * GSTAGE;    // this group's ring: slot s = [A | W | meta]
    const u32 ring_lds = (u32)(size_t)(lptr_t)ring;

    // ---- activation pair image (shared by groups 2h, 2h+1 = stages u even, u+1) ---------------------------------------
    // 2 MT pieces of 1 KiB = 8 token rows x 128 B; piece P lives in the activation area of group 2h (P < MT) or 2h+1
    // (P >= MT), same ring slot.  Inside a piece, half hf of row 2a + b sits at 64-byte position 4a + 2(hf ^ (a & 1)) + b
    // and chunk c at 16-byte position c ^ ((-(row >> 2)) & 3): every 16 consecutive lanes of the DMA fetch two whole
    // lines, and the ds_read_b128 of one stage (lane (li, g): row li, chunk g of its half) is bank-conflict free for
    // the instruction's lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS; measured
    // 2.3 ns per wave read against 3.4 ns for the round-1 image, scripts/microbench_cufill.hip).
    auto aswz = [&](int j) { return (0 - j) & 3; };
    static_assert(KG % 2 == 0, "groups pair up for the activation image");
    const int hf = kg & 1;                            // this group's half of the pair image
    // ---- DMA sources: per-lane 32-bit offsets, stage advance in the scalar base ------------------------------------
    // pieces of a group-stage: MT activation pieces (this wave's share of the PAIR's 2 MT: P = w2 + j * 2WN with
    // w2 = hf * WN + wn), then 2 WN weight pieces (tile t of unit q: [e 4][k32 ^ t 2][c 8][16 B]); this wave takes
    // piece slots wn, wn+WN, ...
    u32 p_off[NPIECE / WN];
    bool p_isw[NPIECE / WN];
    u32 p_lds[NPIECE / WN];                           // relative to ring_lds + slot * GSTAGE
#pragma unroll
    for (int j = 0; j < NPIECE / WN; ++j) {
        const int p = wn + j * WN;
        if (p < MT) {
            const int P = hf * WN + wn + j * 2 * WN;  // pair piece 0 .. 2MT-1
            const int P4 = lane >> 2, pa = P4 >> 2, q = P4 & 3;
            const int half = (q >> 1) ^ (pa & 1);
            const int r = 8 * P + 2 * pa + (q & 1);   // row of the workgroup's token tile
            int row = m0 + r;
            row = row < M ? row : M - 1;
            p_off[j] = (u32)row * (u32)K + half * 64 + (((lane & 3) ^ aswz((r >> 2) & 3)) * 16);
            p_isw[j] = false;
            p_lds[j] = (u32)((P >= MT ? ns * GSTAGE : 0) - hf * ns * GSTAGE) + (P % MT) * 1024;
        } else {
            const int q = p - MT, unit = q >> 1, t = q & 1;
            const int e = lane >> 4, kk = ((lane >> 3) & 1) ^ t, cc = lane & 7;
            p_off[j] = ((u32)trow(unit0 + unit, t) * (u32)KT + kk) * 512u + cc * 64 + e * 16;
            p_isw[j] = true;
            p_lds[j] = ASTAGE + q * 1024;
        }
    }
    // per-group meta: 16*WN dwords of scales (lanes 0-31) | zeros (lanes 32-63); surplus lanes repeat valid addresses
    // (WN <= 2: one instruction carries both, lanes 0-31 scales | 32-63 zeros; WN = 4: 64 dwords each - the even waves
    //  of a group fetch the scales, the odd ones the zeros)
    const int m_dw = WN > 2 ? lane : ((lane & 31) & (16 * WN - 1));    // dword of the workgroup's 16 WN (unit-major)
    const int8_t* const m_base = (WN > 2 ? ((wn & 1) ? zeros : scales8) : ((lane & 32) ? zeros : scales8)) +
                                 chan32(unit0 + (m_dw >> 4), (m_dw >> 3) & 1) + (m_dw & 7) * 4;

    auto dma16 = [&](u32 voff, const void* sbase, u32 lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr)
                     : "memory");
    };
    // weights that exactly ONE workgroup reads (one token block per channel block) are non-temporal (MI355X_MICROARCH.md
    // "nt-weights": issued -> landed -18 %); with several token blocks the same bytes are wanted again by the neighbours on the
    // XCD a moment later, and a streamed line may be gone by then: default policy there (launch_ring sets flags bit 0; round 4,
    // in-run A/B at M = 64: qkv (2 token blocks) 8.45 -> 7.84 us, o (4) 6.66 -> 6.48, down (2 x 4 K slices) 14.05 -> 13.98,
    // gate_up (1 block) 16.97 -> 17.05 the other way; M = 128: qkv 11.70 -> 11.28, o 7.94 -> 7.70).  Measured exception: the
    // 28 672-channel gate_up stream stays 2-3 % faster non-temporal at every M, shared or not (M = 128: 24.56 vs 25.06, g128 35.24
    // vs 36.23; M = 256: 32.55 vs 33.29) - the rule is "several token blocks AND N <= 8192".  The activation tile is re-read by
    // every workgroup from L2 and always keeps the default policy
    auto dma16_nt = [&](u32 voff, const void* sbase, u32 lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_addr)
                     : "memory");
    };
    auto issue = [&](int i, int slot) {               // group-local stage i -> global stage u = i*KG + kg
        const int u = u0 + i * KG + kg;
        const u32 dst = ring_lds + slot * GSTAGE;
#pragma unroll
        for (int j = 0; j < NPIECE / WN; ++j) {
            const void* sb = p_isw[j] ? static_cast<const void*>(W + (size_t)u * 1024)
                                      : static_cast<const void*>(A + (size_t)(u - hf) * 64);   // the pair's line
            static_assert(MT % WN == 0, "piece kind must be a compile-time function of j");
            if (j >= MT / WN && !(flags & 1)) dma16_nt(p_off[j], sb, dst + p_lds[j]);   // (p = wn + j*WN >= MT: a weight piece)
            else dma16(p_off[j], sb, dst + p_lds[j]);
        }
        if (MODE == 1) {
            const int8_t* src = m_base + (size_t)(u >> 1) * N;
            const u32 ml = dst + ASTAGE + WSTAGE + (WN > 2 ? (wn & 1) * MZ : 0);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(ml) : "memory");
        }
    };

    // ---- epilogue operands (round 4): the per-channel / per-token scale vectors of the workgroup's tile are requested HERE,
    // by LDS-DMA into the top 2 KiB of the LDS window (above every ring and reduction area), by wave 0, ahead of the first
    // ring stage in its in-order queue (the counted ring waits are unaffected: the extra requests are OLDER than any stage).
    // Before, every lane loaded its own values from memory after the k loop - a dependent memory round trip between the last
    // MFMA and the epilogue while the other workgroups still stream: 2 000 - 5 600 cycles in the timeline trace (gate_up:
    // 14 % of the launch).  Layout: [w scale: 64 WN halfs | w scale*zero: 64 WN halfs | token scale: 64 x 4-byte slots | token
    // sum: 64 slots]; local channel = 64 wn + 32 t + ... (ACT: the 32 WN gate channels, then the 32 WN up channels).
    uint8_t* const s_sc = smem + SC_OFF;
    constexpr int TSL = MT > 4 ? 512 : 256;           // bytes of one staged token vector (4-byte slots)
    static_assert(4 * 64 * WN + 2 * TSL <= SC_BYTES, "epilogue operand staging area");
    if ((OUTK == 0 || OUTK == 2) && wave == 0) {
        const u32 sc_lds = (u32)(size_t)(lptr_t)s_sc;
        auto dma4p = [&](const void* src, u32 dst) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(dst) : "memory");
        };
        constexpr int NCH = 64 * WN;                  // channels of the workgroup's tile
#pragma unroll
        for (int i = 0; i < (NCH + 127) / 128; ++i) { // 128 halfs per instruction
            const int lc = 128 * i + 2 * lane;
            // lanes beyond the tile's channels (one-unit workgroups: lanes 32-63) are masked OUT of the DMA: an LDS-DMA lane writes
            // M0 + 4 * lane whatever its source, and a surplus lane would land in the NEXT region of this staging area (correct
            // only as long as the later DMA that fills that region also lands later - nothing guarantees write order)
            if (lc < NCH) {
                const int gc = ACT ? (lc < 32 * WN ? 32 * unit0 + lc : N / 2 + 32 * unit0 + (lc - 32 * WN)) : unit0 * 64 + lc;
                dma4p(reinterpret_cast<const _Float16*>(wscales) + gc, sc_lds + 256 * i);
                if (MODE == 0) dma4p(reinterpret_cast<const _Float16*>(wszs) + gc, sc_lds + 2 * NCH + 256 * i);
            }
        }
        // token vectors: one half per lane in a 4-byte slot (2-byte requests: no alignment assumption on a [M] vector); 64 tokens
        // per instruction, TSL bytes per vector
#pragma unroll
        for (int i = 0; i < (16 * MT + 63) / 64; ++i) {
            int m = m0 + 64 * i + lane;
            m = m < M ? m : M - 1;
            const _Float16* sa_p = reinterpret_cast<const _Float16*>(ascales) + m;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_ushort %0, off" ::"v"(sa_p), "s"(sc_lds + 4 * NCH + 256 * i)
                         : "memory");
            if (MODE == 0) {
                const _Float16* ss_p = reinterpret_cast<const _Float16*>(assums) + m;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_ushort %0, off" ::"v"(ss_p),
                             "s"(sc_lds + 4 * NCH + TSL + 256 * i)
                             : "memory");
            }
        }
    }

    // ---- prologue: stages 0..ns-2 in flight, operands of stage 0 in registers ---------------------------------------
    QS_STAMP(15);
    for (int j = 0; j < ns - 1; ++j)
        if (j < nloc) issue(j, j);
    QS_STAMP(1);
    // Everything the k loop needs but the first requests do not - operand reader addresses, accumulators - is computed BEHIND
    // the prologue (round 4, timeline trace: the first request left 1 200-1 400 cycles after kernel entry; with two waves per
    // SIMD every instruction in front of it costs 4-5 cycles of a launch whose data is ~2.5 us away).  Nothing may be hoisted
    // back above the requests:
    __builtin_amdgcn_sched_barrier(0);
    // ---- LDS operand readers -------------------------------------------------------------------------------------
    // activation operand of m-tile mt = rows 16 mt + li: pieces 2 mt + (li >> 3) of the pair image
    int a_rd[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r8 = li & 7, pa = r8 >> 1;
        const int in_piece = (4 * pa + 2 * (hf ^ (pa & 1)) + (r8 & 1)) * 64 + ((g ^ aswz(li >> 2)) * 16);
        if (MT == 1) a_rd[mt] = (li >> 3) * (ns * GSTAGE) + in_piece;
        else a_rd[mt] = (2 * mt >= MT ? ns * GSTAGE : 0) + ((2 * mt) % MT + (li >> 3)) * 1024 + in_piece;
    }
    const uint8_t* const pair_ring = smem + (kg - hf) * ns * GSTAGE;
    const int w_rd = ASTAGE + wn * 2048 + tsel * 1024 + (((g >> 1) ^ tsel)) * 128 + c * 16 + (g & 1) * 8;   // + e*256
    const int m_rd = ASTAGE + WSTAGE + wn * 64 + (tsel * 8 + c) * 4;                               // zeros at + MZ
    struct Raw {
        v2u r[4];
        u32 sdw, zdw;
    };
    struct Ops {
        v4i a[4];
        v4i b[MT];
    };
    // per-group: separate address registers keep these four 8-byte reads from being merged into ds_read2_b64 (half rate
    // and, on this image, two-way conflicts; gemm_w4a8_tiled.hip has the measurement)
    int w_rd_e[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        w_rd_e[e] = w_rd + e * 256;
        if (MODE == 1) asm volatile("" : "+v"(w_rd_e[e]));
    }
    auto read_raw = [&](int slot) -> Raw {
        const uint8_t* s = ring + slot * GSTAGE;
        Raw q;
#pragma unroll
        for (int e = 0; e < 4; ++e) q.r[e] = *reinterpret_cast<const v2u*>(s + w_rd_e[e]);
        q.sdw = 0;
        q.zdw = 0;
        if (MODE == 1) {
            q.sdw = *reinterpret_cast<const u32*>(s + m_rd);
            q.zdw = *reinterpret_cast<const u32*>(s + m_rd + MZ);
        }
        return q;
    };
    auto read_b = [&](int slot, int mt) -> v4i {
        return *reinterpret_cast<const v4i*>(pair_ring + slot * GSTAGE + a_rd[mt]);
    };
    auto build = [&](const Raw& q, int cl) -> v4i {
        u32 s = 0, zb = 0;
#ifdef QS_TIMING
        // timing builds only, WRONG RESULTS (flags & 8192): per-group launches keep their meta DMA and LDS reads but skip the level-2
        // arithmetic - what the dequant VALU costs on top of the per-channel stream (round 6, config 3 decomposition)
        if (MODE == 1 && (flags & 8192)) {
            v4i a;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const u32 raw = ((cl & 1) ? q.r[e].y : q.r[e].x) ^ (q.sdw & q.zdw & 1u);
                a[e] = (int)((cl & 2) ? unpack_hi<0>(raw, 0, 0) : unpack_lo<0>(raw, 0, 0));
            }
            return a;
        }
#endif
        if (MODE == 1) {
            s = (q.sdw >> (8 * cl)) & 0xFFu;
            zb = ((q.zdw >> (8 * cl)) & 0xFFu) * 0x01010101u;
        }
        v4i a;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const u32 raw = (cl & 1) ? q.r[e].y : q.r[e].x;
            a[e] = (int)((cl & 2) ? unpack_hi<MODE>(raw, s, zb) : unpack_lo<MODE>(raw, s, zb));
        }
        return a;
    };

    v4i acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) acc[mt][cl] = (v4i){0, 0, 0, 0};

    // wait for stage 0: younger stages outstanding = min(ns-1, nloc) - 1
    {
        const int young = (nloc < ns - 1 ? nloc : ns - 1) - 1;
        if (young >= 3) wait_vm<3 * NDMA>();
        else if (young == 2) wait_vm<2 * NDMA>();
        else if (young == 1) wait_vm<1 * NDMA>();
        else wait_vm<0>();
    }
    raw_barrier();
    QS_STAMP(2);
#ifdef QS_RING_TRACE
    unsigned t_wait = 0, t_round = 0;
#endif
    Ops o0, o1;
    {
        const Raw q0 = read_raw(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) o0.b[mt] = read_b(0, mt);
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) o0.a[cl] = build(q0, cl);
    }

    // One round: MFMAs of stage i from `cur`; meanwhile operands of stage i+1 -> `nxt`, DMA of stage i+ns-1.
    auto round = [&](int i, int slot, const Ops& cur, Ops& nxt) {
        const int slot_n = slot + 1 == ns ? 0 : slot + 1;
        const int slot_d = slot == 0 ? ns - 1 : slot - 1;
        Raw qn;
#ifdef QS_TIMING
        const bool t_reads = !(flags & 64), t_mfma = !(flags & 32);   // timing builds only: results wrong by design
#else
        constexpr bool t_reads = true, t_mfma = true;
#endif
        if (t_reads) {
            qn = read_raw(slot_n);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) nxt.b[mt] = read_b(slot_n, mt);
        }
        if (i + ns - 1 < nloc) issue(i + ns - 1, slot_d);
        if (t_mfma) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int cl = 0; cl < 4; ++cl)
                    acc[mt][cl] = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur.a[cl], cur.b[mt], acc[mt][cl], 0, 0, 0);
                if (mt == 0) {
#pragma unroll
                    for (int cl = 0; cl < 4; ++cl) nxt.a[cl] = build(qn, cl);
                }
            }
            // Per-group (round 6): the operands built for the NEXT round must be finished INSIDE this round, under its MFMAs.  Left to
            // itself the compiler sinks the second round's level-2 dequant (20 multiplies, ~120 VALU instructions) across the loop's
            // back edge to its use - in front of the first round's MFMAs: one round then carries 240 VALU instructions and its
            // MFMAs wait for them, the other round's 16 MFMAs run with nothing beside them (seen in the ISA: blocks of 16 MFMAs +
            // 40 v_mul_lo_u32 and of 16 MFMAs + 0).  An empty asm that reads the registers pins the computation here.
            if (MODE == 1) {
#pragma unroll
                for (int cl = 0; cl < 4; ++cl) asm volatile("" ::"v"(nxt.a[cl]));
            }
        }
    };
    // barrier(i): stage i+1 landed for every wave of the group (own pieces by the counted wait), everybody is done
    // reading stage i-1 (its slot is refilled by the DMA issued in round i)
    auto wait_next = [&](int i) {
        const int rem = nloc - 1 - i;                 // stages after i
        const int young = (rem < ns - 2 ? rem : ns - 2) - 1;   // stages allowed to stay in flight beyond i+1
        if (young >= 3) wait_vm<3 * NDMA>();
        else if (young == 2) wait_vm<2 * NDMA>();
        else if (young == 1) wait_vm<1 * NDMA>();
        else wait_vm<0>();
    };
    int slot = 0;
    for (int i = 0; i < nloc; i += 2) {               // two rounds per trip (operand registers ping-pong); nloc may be odd
#ifdef QS_RING_TRACE
        unsigned long long ta = __builtin_amdgcn_s_memtime();
#endif
        wait_next(i);
        raw_barrier();
#ifdef QS_RING_TRACE
        QS_ACC(t_wait, ta);
        ta = __builtin_amdgcn_s_memtime();
#endif
        round(i, slot, o0, o1);
#ifdef QS_RING_TRACE
        QS_ACC(t_round, ta);
        ta = __builtin_amdgcn_s_memtime();
#endif
        slot = slot + 1 == ns ? 0 : slot + 1;
        if (i + 1 < nloc) {
            wait_next(i + 1);
            raw_barrier();
#ifdef QS_RING_TRACE
            QS_ACC(t_wait, ta);
            ta = __builtin_amdgcn_s_memtime();
#endif
            round(i + 1, slot, o1, o0);
#ifdef QS_RING_TRACE
            QS_ACC(t_round, ta);
#endif
            slot = slot + 1 == ns ? 0 : slot + 1;
        }
    }
    QS_STAMP(3);
#ifdef QS_RING_TRACE
    if (g_ring_trace && lane == 0) {
        g_ring_trace[((size_t)blockIdx.x * 8 + wave) * 16 + 8] = t_wait;
        g_ring_trace[((size_t)blockIdx.x * 8 + wave) * 16 + 9] = t_round;
    }
#endif

    // ---- reduce the KG partial tiles through LDS, fused epilogue -----------------------------------------------------
    const int ncol0 = chan32(unit0 + wn, g >> 1) + 4 * (g & 1);
    const int lcol0 = (ACT ? (g >> 1) * 32 * WN + 32 * wn : 64 * wn + 32 * (g >> 1)) + 4 * (g & 1);   // the same, local to the tile
    constexpr int NP = MT * 4;                         // 16 x 16 result pieces per wave: piece pc = mt*4 + cl
    // DISTRIBUTED: piece pc is finished by K-group pc % KG - every group sums, scales and converts a 1/KG share of the tile
    // instead of groups 1..KG-1 handing everything to group 0 and leaving (timeline trace, scripts/trace_gemm.py: reduction +
    // epilogue on two of eight waves = 3 us of a 19 us launch at gate_up size; in-run A/B against the group-0 form in the r2
    // builds: qkv 9.8 -> 8.7, o 7.5 -> 7.2, gate_up 19.5 -> 18.1 us).  Since round 4 the K-sliced launches take this form as
    // well: all eight waves carry the seam's traffic (below).
    {
        constexpr int PPG = (NP + KG - 1) / KG;        // pieces per owning group
        // 128-token workgroups (MT = 8): the partial tiles of all groups are 192 KB (128 KB for four-unit workgroups) - they go
        // through LDS in NPASS passes of PPP pieces per owner
        constexpr int NPASS = MT > 4 ? 2 : 1;
        constexpr int PPP = PPG / NPASS;
        static_assert(PPG % NPASS == 0, "pieces per owner must split evenly over the passes");
        constexpr int RS = ACT ? 64 * WN + 16 : 144;   // staged fp16 row: 128 B + 16 (keeps 16-byte alignment); ACT: the
                                                       // workgroup's 32 WN result channels in one row
        uint8_t* const st = smem + (size_t)KG * WN * (KG - 1) * PPP * 1024 + (ACT ? wn * 64 : wn * (16 * MT * RS));
        // The pieces this wave finishes are pc = q KG + kg, q < PPG.  Their accumulators are picked by wave-uniform SELECTS, not by
        // branches (round 4: with one basic block per (q, kk) the four pieces of a gate_up wave ran one after the other, each
        // a chain LDS read -> sum -> scale -> exp / rcp -> LDS write with nothing to overlap it: 4 400 cycles between the two
        // barriers in the timeline trace; as straight-line code the reads of all pieces are in flight together).
        v4i sum[PPG];
        v4i* const red4 = reinterpret_cast<v4i*>(smem);
#ifdef QS_TIMING
        // timing builds only, WRONG RESULTS (what would a geometry without the cross-group reduction save?): 8 = leave behind the k
        // loop (one store per wave keeps the accumulators alive), 4 = no exchange of partial tiles - every wave finishes its pieces
        // from its own accumulators, epilogue / staging / row stores as usual
        if (flags & 8) {
            v4i x = acc[0][0];
#pragma unroll
            for (int pc = 1; pc < NP; ++pc) x += acc[pc >> 2][pc & 3];
            if (x[0] == 0x12345678) *reinterpret_cast<v4i*>(out) = x;
            return;
        }
        const bool t_red = !(flags & 4);
#else
        constexpr bool t_red = true;
#endif
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (t_red) __syncthreads();                // rings are dead (every wave drained its DMA queue) / previous pass consumed
            if (pass == 0) QS_STAMP(4);
            // partial of piece pc from group kg -> slot [owner][wn][source index among the other groups][pc / KG - pass PPP][lane]
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                const int own = pc % KG, qq = pc / KG;
                if (qq < pass * PPP || qq >= (pass + 1) * PPP) continue;
                if (own != kg && t_red) {
                    const int src = kg < own ? kg : kg - 1;
                    red4[((((own * WN + wn) * (KG - 1) + src) * PPP + (qq - pass * PPP)) << 6) + lane] = acc[pc >> 2][pc & 3];
                }
            }
            if (pass == 0) QS_STAMP(7);
            if (t_red) __syncthreads();
            if (pass == 0) QS_STAMP(10);
#pragma unroll
            for (int q = pass * PPP; q < (pass + 1) * PPP; ++q) {
                v4i own = acc[(q * KG) >> 2][(q * KG) & 3];
#pragma unroll
                for (int kk = 1; kk < KG; ++kk)
                    if (q * KG + kk < NP) own = kg == kk ? acc[(q * KG + kk) >> 2][(q * KG + kk) & 3] : own;
                sum[q] = own;
            }
#pragma unroll
            for (int q = pass * PPP; q < (pass + 1) * PPP; ++q)
#pragma unroll
                for (int sidx = 0; sidx < KG - 1; ++sidx)
                    if (t_red) sum[q] += red4[((((kg * WN + wn) * (KG - 1) + sidx) * PPP + (q - pass * PPP)) << 6) + lane];
        }
#ifdef QS_RING_TRACE
#pragma unroll
        for (int q = 0; q < PPG; ++q) asm volatile("" : "+v"(sum[q]));
        QS_STAMP(11);
#endif
        // ---- K-split seam (per wave: its own pieces of unit wn's tile) ------------------------------------------------------
        // Round 4: no ticket, and every wave takes part.  The slabs hold a SENTINEL word (QS_SLAB_SENTINEL, a value no partial
        // sum of an admitted K slice can take) wherever no partial has been delivered.  Slices 0 .. ksplit-2 store their
        // pieces (system-scope write-through, fire and forget) and leave; the LAST slice - dispatched last, so every slice it
        // waits for already holds a CU - loads the other slices' pieces with cache-missing loads, repeats the load while any
        // word still shows the sentinel, adds, and puts the sentinel back for the next launch (ordered by the kernel
        // boundary).  Before: acknowledged stores -> device-scope ticket -> loads by the last arriver, three dependent round
        // trips, the last one 48 KB through ONE wave per unit (MI355X_MICROARCH.md handoff-payload: "one wave reading a fresh
        // 64 KB slot = 4.9 us") - 9 000 cycles in the timeline trace at down's shape.
        // No fences (an agent-scope release / acquire writes back and invalidates the XCD's whole L2 under every other
        // workgroup's feet, measured +10 us): the slab traffic bypasses the caches (sc0 sc1), and the data is its own flag - a
        // word is either the sentinel or final, so torn 16-byte accesses are harmless.
        if (KSPLIT && OUTK != 3 && ksplit > 1) {
            const size_t tile = (size_t)(unit0 + wn) * mblocks + mblk;
            v4i* const slab = reinterpret_cast<v4i*>(slabs) + tile * ksplit * (size_t)(NP * 64) + lane;
            if (kq != ksplit - 1) {
                // (flags & 128: the armed one-shot fault of qs_debug_inject_fault - tile 0's producers never deliver)
                if (!((flags & 128) && tile == 0)) {
#pragma unroll
                    for (int q = 0; q < PPG; ++q) {
                        const int pc = q * KG + kg;
                        if (NP % KG != 0 && pc >= NP) continue;
                        v4i* const dst = slab + ((size_t)kq * NP + pc) * 64;
                        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(sum[q]) : "memory");
                    }
                }
                return;
            }
            // BOUNDED (round 5): after `spin_cap` polls the wave stops waiting, reports through the device error word (`counters`
            // = qs_gemm_error_word()) and finishes with what it has - a wrong tile and a status bit instead of a hung GPU, if the
            // dispatch-order assumption above ever fails or the slabs were left without their sentinel (include/qserve_amd.h)
            const int spin_cap = (flags & 128) ? 4096 : QS_SPIN_CAP;
            int polls = 0;
            bool gave_up = false;
            constexpr int ZB = PPG <= 4 ? 3 : 1;       // slices requested together (up to 12 x 16 B per lane in flight)
            const v4i sent = {QS_SLAB_SENTINEL, QS_SLAB_SENTINEL, QS_SLAB_SENTINEL, QS_SLAB_SENTINEL};
            // cache-missing loads through the buffer BUILTIN (aux 17 = sc0 | sc1): the compiler counts them and waits before
            // the first use.  An inline-asm load is invisible to it - inside this retry loop the loaded registers are loop-
            // carried, and hipcc copied them BEFORE the asm's own s_waitcnt (seen in the ISA; the MT = 1 geometry returned
            // garbage; cdna_hip_programming.md 5.7 item 1)
            const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, 0x7FFFFFFF, 0x00020000);
            const u32 sl_off = (u32)((tile * ksplit * (size_t)(NP * 64) + lane) * 16);   // (the workspace is 48 MiB)
            for (int j0 = 0; j0 < ksplit - 1; j0 += ZB) {
                v4i t[ZB][PPG];
                for (;;) {
#pragma unroll
                    for (int jb = 0; jb < ZB; ++jb) {
                        const int z = j0 + jb;
                        if (z < ksplit - 1) {
#pragma unroll
                            for (int q = 0; q < PPG; ++q) {
                                const int pc = q * KG + kg;
                                if (NP % KG != 0 && pc >= NP) continue;
                                t[jb][q] = __builtin_bit_cast(
                                    v4i, __builtin_amdgcn_raw_buffer_load_b128(srs, sl_off, (z * NP + pc) * 1024, 17));
                            }
                        }
                    }
                    int missing = 0;
#pragma unroll
                    for (int jb = 0; jb < ZB; ++jb)
                        if (j0 + jb < ksplit - 1) {
#pragma unroll
                            for (int q = 0; q < PPG; ++q) {
                                if (NP % KG != 0 && q * KG + kg >= NP) continue;
#pragma unroll
                                for (int r = 0; r < 4; ++r) missing |= t[jb][q][r] == QS_SLAB_SENTINEL;
                            }
                        }
#ifdef QS_RING_TRACE
                    if (g_ring_trace && lane == 0) g_ring_trace[((size_t)blockIdx.x * 8 + wave) * 16 + 13] += 1;
#endif
                    if (!__builtin_amdgcn_ballot_w64(missing != 0)) break;
                    if (++polls >= spin_cap) {
                        gave_up = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
#pragma unroll
                for (int jb = 0; jb < ZB; ++jb) {
                    const int z = j0 + jb;
                    if (z < ksplit - 1) {
#pragma unroll
                        for (int q = 0; q < PPG; ++q) {
                            const int pc = q * KG + kg;
                            if (NP % KG != 0 && pc >= NP) continue;
                            sum[q] += t[jb][q];
                            v4i* const dst = slab + ((size_t)z * NP + pc) * 64;
                            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(sent) : "memory");
                        }
                    }
                }
            }
            if (gave_up && counters && lane == 0) atomicOr(counters, QS_ERR_GEMM_SEAM);
            QS_STAMP(14);
        }
#pragma unroll
        for (int q = 0; q < PPG; ++q) {
            const int pc = q * KG + kg;                // wave-uniform
            if (NP % KG != 0 && pc >= NP) continue;    // (fewer pieces than groups: MT = 1 with eight K-groups)
            const int mt = pc >> 2, cl = pc & 3;
            if (OUTK == 1 || OUTK == 3) {
                // OUTK == 3 (round 4): K-slice PLANES - every slice leaves its own int32 partial tile in plane kq of
                // out [ksplit][M][N]; no seam at all: the consumer (the row kernel that follows, qs_add_residual_rms_norm_general_planes)
                // sums the planes and applies this kernel's epilogue itself, and the kernel boundary is the hand-off
                const int m = m0 + 16 * mt + li;
                int* const plane = reinterpret_cast<int*>(out) + (OUTK == 3 ? (size_t)kq * M * N : (size_t)0);
                if (m < M) *reinterpret_cast<v4i*>(plane + (size_t)m * N + ncol0 + 8 * cl) = sum[q];
            } else {
                // scale operands from the LDS staging area (requested at kernel start, see above)
                const int lcol = lcol0 + 8 * cl;
                const h4 ws4 = *reinterpret_cast<const h4*>(s_sc + 2 * lcol);
                const float sa = (float)*reinterpret_cast<const _Float16*>(s_sc + 4 * (64 * WN) + 4 * (16 * mt + li));
                h4 o;
                if (MODE == 0) {
                    const h4 wz4 = *reinterpret_cast<const h4*>(s_sc + 2 * (64 * WN) + 2 * lcol);
                    const float ss = (float)*reinterpret_cast<const _Float16*>(s_sc + 4 * (64 * WN) + TSL + 4 * (16 * mt + li));
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (_Float16)epi_per_chn(sum[q][r], (float)ws4[r], sa, (float)wz4[r], ss, flags & 16);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (_Float16)epi_per_group(sum[q][r], (float)ws4[r], sa);
                }
                if (ACT) {
                    // lanes 0-31 hold the gate values, lanes 32-63 the up values of the same (token, channel):
                    // silu_and_mul's arithmetic on the fp16-rounded GEMM outputs (activation_kernels.cu:11,21-32)
                    const v2u ob = __builtin_bit_cast(v2u, o);
                    // one swap hands every lane the pair it finishes: r[0] = (x of lanes 0-31 | y of lanes 0-31) = gate elements
                    // 0, 1 for the lower half, 2, 3 for the upper; r[1] = (x | y of lanes 32-63) = the matching up elements
                    const auto sw = __builtin_amdgcn_permlane32_swap(ob.x, ob.y, false, false);
                    const h2 gt = __builtin_bit_cast(h2, (u32)sw[0]), up = __builtin_bit_cast(h2, (u32)sw[1]);
                    const int hh = g >> 1;
                    h2 a;
                    a[0] = (_Float16)((float)qs_silu_h((float)gt[0]) * (float)up[0]);
                    a[1] = (_Float16)((float)qs_silu_h((float)gt[1]) * (float)up[1]);
                    *reinterpret_cast<h2*>(st + (16 * mt + li) * RS + (8 * cl + 4 * (g & 1) + 2 * hh) * 2) = a;
                } else {
                    *reinterpret_cast<h4*>(st + (16 * mt + li) * RS + (32 * (g >> 1) + 8 * cl + 4 * (g & 1)) * 2) = o;
                }
            }
        }
        if (OUTK == 1 || OUTK == 3) return;
        QS_STAMP(12);
        __syncthreads();                               // the fp16 tile of every unit is staged
        QS_STAMP(5);
        if (ACT) {                                     // rows of 64 WN bytes, shared by all eight waves
            constexpr int LPR = 4 * WN, RPI = 64 / LPR;                // lanes per row, rows per instruction
            const uint8_t* const sa0 = smem + (size_t)KG * WN * (KG - 1) * PPP * 1024;
            _Float16* const arow = reinterpret_cast<_Float16*>(out) + unit0 * 32 + (lane % LPR) * 8;
#pragma unroll
            for (int i = 0; i < 16 * MT / RPI; ++i) {
                if (i % 8 != wave) continue;
                const int r = i * RPI + lane / LPR;
                const int m = m0 + r;
                const v4u v = *reinterpret_cast<const v4u*>(sa0 + r * RS + (lane % LPR) * 16);
                if (m < M) *reinterpret_cast<v4u*>(arow + (size_t)m * (N / 2)) = v;
            }
            return;
        }
        _Float16* const orow = reinterpret_cast<_Float16*>(out) + (unit0 + wn) * 64 + (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            if (i % KG != kg) continue;                    // the rows of unit wn are shared by its KG waves
            const int r = i * 8 + (lane >> 3);
            const int m = m0 + r;
            const v4u v = *reinterpret_cast<const v4u*>(st + r * RS + (lane & 7) * 16);
            if (m < M) *reinterpret_cast<v4u*>(orow + (size_t)m * N) = v;
        }
        QS_STAMP(6);
    }
}

// MT m-tiles (16 tokens each) per wave = tokens per workgroup / 16 (1, 2, 4; 8 = 128-token workgroups, round 5: one level-2
// dequant of a weight byte serves 128 tokens, ring depth 3-4, reduction through LDS in two passes); WN units per workgroup;
// KG = 8 / WN K-groups.
// KSPLIT = false folds every K-slice path away (the un-split launches keep exactly their earlier code).
template <int MT, int WN, int MODE, int OUTK, bool KSPLIT>
__global__ __launch_bounds__(512, 1) void w4a8_gemm_ring(const int8_t* __restrict__ A, const uint8_t* __restrict__ W,
                                                         const int8_t* __restrict__ zeros,
                                                         const int8_t* __restrict__ scales8,
                                                         const __half* __restrict__ wscales,
                                                         const __half* __restrict__ ascales,
                                                         const __half* __restrict__ wszs,
                                                         const __half* __restrict__ assums, void* __restrict__ out,
                                                         int M, int N, int K, int mblocks, int ns, int ksplit_arg,
                                                         int* __restrict__ slabs, unsigned* __restrict__ counters,
                                                         int flags) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    ring_body<MT, WN, MODE, OUTK, KSPLIT>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, mblocks, ns,
                                          ksplit_arg, slabs, counters, flags, smem);
}

template <int MT, int WN, int MODE, int OUTK, bool KSPLIT>
int launch_ring(const int8_t* A, const uint8_t* W, const int8_t* zeros, const int8_t* scales8, const void* wscales,
                const void* ascales, const void* wszs, const void* assums, void* out, int M, int N, int K,
                int mblocks, int ksplit, int* slabs, unsigned* counters, hipStream_t stream) {
    auto kern = w4a8_gemm_ring<MT, WN, MODE, OUTK, KSPLIT>;
    constexpr int KG = 8 / WN;
    constexpr int GSTAGE = ring_gstage<MT, WN, MODE>();
    // ring depth: as deep as 144 KiB of LDS allows, never deeper than a group's stage count + 1.  At most 5: every
    // stage layer carries 16 KiB of weights per CU, and beyond 4 layers (64 KiB per CU, 16 MB over the chip) in flight
    // the stream no longer gains while the pipeline fill gets longer (depth sweep 3 / 4 / 5 / 6 at M = 16: o 8.9 / 6.7 /
    // 6.45 / 6.66 us, gate_up 18.7 / 13.8 / 13.6 / 13.9 us; M = 64: o 9.2 / 7.1 / 6.74 / 6.96 us; depth 3 costs 25-40 %)
    int ns = (144 * 1024) / (KG * GSTAGE);
    if (ns > 5) ns = 5;
    const int nloc = (K / 64) / ksplit / KG;
    if (ns > nloc + 1) ns = nloc + 1;
    if (const int f = (g_ring_flags >> 8) & 7) {       // A/B: forced depth (sensitivity to the bytes in flight)
        if (f >= 3 && (size_t)KG * f * GSTAGE <= SC_OFF && f <= nloc + 1) ns = f;
    }
    if (ns < 3) ns = 3;                                // the slot read ahead and the slot refilled must differ
    size_t smem = (size_t)KG * ns * GSTAGE;
    size_t tail;
    {   // reduction area [owner KG][WN][KG-1 sources][pieces per group and pass] x 1 KiB (two passes for 128-token workgroups) +
        // the staged fp16 tile (activation epilogue: rows of 64 WN + 16 bytes)
        const size_t ppp = (MT * 4 + KG - 1) / KG / (MT > 4 ? 2 : 1);
        tail = (size_t)KG * WN * (KG - 1) * ppp * 1024 + (OUTK == 2 ? (size_t)16 * MT * (64 * WN + 16) : (size_t)WN * 16 * MT * 144);
    }
    if (smem < tail) smem = tail;
    if (smem > (size_t)SC_OFF) {
        qs_set_error("w4a8 gemm (ring): LDS layout overflow (%zu bytes below the epilogue operands)", smem);
        return QS_EINVAL;
    }
    smem = 160 * 1024;                                 // the epilogue operands sit in the top SC_BYTES of the window
    static size_t configured_dev[QS_MAX_DEVICES] = {};   // per instantiation and device
    size_t& configured = configured_dev[qs_device_slot()];
    if (configured < smem) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) {
            qs_set_error("w4a8 gemm (ring): cannot reserve LDS: %s", hipGetErrorString(e));
            return (int)e;
        }
        configured = 160 * 1024;
    }
    dim3 grid((N / (64 * WN)) * mblocks * ksplit);
    // K slices across the XCDs (ring_coords): always for the planes form (no finisher); for the seam form with TWO slices only -
    // with four, every finishing workgroup (the last slice: polls, adds, epilogue, row stores) sits on XCDs 6 and 7 and the launch
    // waits for a quarter of the chip (measured: g128 down_proj at 128 tokens, <4,2> x 2 token blocks x 4 slices, 25.4 -> 26.4 us;
    // with two slices the finishers are half the chip: per-channel down at 64 tokens 14.4 -> 13.4 us, g128 18.9 -> 17.7)
    const bool kxcd = KSPLIT && (ksplit == 2 || (OUTK == 3 && (ksplit == 4 || ksplit == 8))) &&
                      (N / (64 * WN)) % (8 / ksplit) == 0 && !(g_ring_flags & 4096);
    int inject = 0;
    if (KSPLIT && OUTK != 3 && ksplit > 1) {
        counters = qs_gemm_error_word(qs_scratch_slot(stream));   // the kernel's `counters` is the error word of the seam's bounded wait
        if (g_inject_fault & 1) inject = 128, g_inject_fault &= ~1;
    }
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, stream, A, W, zeros, scales8,
                       reinterpret_cast<const __half*>(wscales), reinterpret_cast<const __half*>(ascales),
                       reinterpret_cast<const __half*>(wszs), reinterpret_cast<const __half*>(assums), out, M, N, K,
                       mblocks, ns, ksplit, slabs, counters,
                       (g_ring_flags & ~(3 | 16 | 128 | 2048 | 4096)) | inject | (g_epi_fma ? 16 : 0) | (kxcd ? 2048 : 0) |
                           ((g_ring_flags & 1) || (mblocks > 1 && N <= 8192 && !(g_ring_flags & 2)) ? 1 : 0));
    return qs_launch_status("w4a8 gemm (ring)");
}

}  // namespace

#ifdef QS_RING_TRACE
extern "C" int qs_debug_ring_trace(void* buf) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ring_trace), &p, sizeof(p));
}
#endif

namespace {
template <int MT, int WN, int MODE, int OUTK>
int ring_go(bool ks, const int8_t* A, const uint8_t* W, const int8_t* zeros, const int8_t* scales8, const void* wscales,
            const void* ascales, const void* wszs, const void* assums, void* out, int M, int N, int K, int mblocks,
            int ksplit, int* slabs, unsigned* counters, hipStream_t stream) {
    if constexpr (OUTK == 3) {                        // planes exist in the K-sliced instantiation only (one slice = OUTK 1)
        return launch_ring<MT, WN, MODE, OUTK, true>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, mblocks,
                                                     ksplit, nullptr, nullptr, stream);
    } else {
        if constexpr (OUTK != 2) {
            if (ks)
                return launch_ring<MT, WN, MODE, OUTK, true>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K,
                                                             mblocks, ksplit, slabs, counters, stream);
        }
        return launch_ring<MT, WN, MODE, OUTK, false>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K,
                                                      mblocks, 1, nullptr, nullptr, stream);
    }
}
}  // namespace

// Entry used by the dispatcher in gemm_w4a8.hip.  mt = m-tiles per workgroup (1, 2, 4), wn = units per workgroup
// (1, 2, 4); ksplit = K slices (1 = none; > 1 needs the slab / counter workspace: (N/64) * mblocks * ksplit * mt KiB * 4 and
// (N/64) * mblocks counters); preconditions (checked there): N % (64*wn) == 0, (K/64/ksplit) % (8/wn) == 0,
// M*K and N*K/2 below 4 GiB.
int qs_launch_gemm_ring(int mode, int outk, int mt, int wn, const int8_t* A, const uint8_t* W, const int8_t* zeros,
                        const int8_t* scales8, const void* wscales, const void* ascales, const void* wszs,
                        const void* assums, void* out, int M, int N, int K, int mblocks, int ksplit, int* slabs,
                        unsigned* counters, hipStream_t stream) {
    if (g_qs_plan.active) {
        g_qs_plan.family = 3;
        g_qs_plan.p[0] = mt, g_qs_plan.p[1] = wn, g_qs_plan.p[2] = mblocks, g_qs_plan.p[3] = ksplit;
        return QS_OK;
    }
    if (outk == 2 && ksplit > 1) {
        qs_set_error("w4a8 gemm (ring): the activation epilogue has no K-sliced form");
        return QS_ENOSUP;
    }
    if ((outk == 0 || outk == 2) && ((reinterpret_cast<uintptr_t>(wscales) & 3) || (mode == 0 && (reinterpret_cast<uintptr_t>(wszs) & 3)))) {
        // (the reference reads them as half2, gemm_cuda.cu:581-582: the same requirement)
        qs_set_error("w4a8 gemm: wscales / w_szs must be 4-byte aligned");
        return QS_EINVAL;
    }
    const bool ks = ksplit > 1;
#define QS_R(MTV, WNV, MODEV, OUTV)                                                                                  \
    return ring_go<MTV, WNV, MODEV, OUTV>(ks, A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, \
                                          mblocks, ksplit, slabs, counters, stream)
#define QS_RM(MODEV, OUTV)                              \
    do {                                                \
        if (wn == 4) {                                  \
            if (mt == 4) QS_R(4, 4, MODEV, OUTV);       \
        } else if (wn == 2) {                           \
            if (mt == 8) QS_R(8, 2, MODEV, OUTV);       \
            if (mt == 4) QS_R(4, 2, MODEV, OUTV);       \
            if (mt == 2) QS_R(2, 2, MODEV, OUTV);       \
        } else {                                        \
            if (mt == 4) QS_R(4, 1, MODEV, OUTV);       \
            if (mt == 2) QS_R(2, 1, MODEV, OUTV);       \
            if (mt == 1) QS_R(1, 1, MODEV, OUTV);       \
        }                                               \
    } while (0)
    if (mode == 0 && outk == 0) QS_RM(0, 0);
    if (mode == 0 && outk == 1) QS_RM(0, 1);
    if (mode == 0 && outk == 2) QS_RM(0, 2);
    if (mode == 1 && outk == 0) QS_RM(1, 0);
    if (mode == 1 && outk == 1) QS_RM(1, 1);
    if (mode == 1 && outk == 2) QS_RM(1, 2);
    if (mode == 0 && outk == 3) QS_RM(0, 3);
    if (mode == 1 && outk == 3) QS_RM(1, 3);
#undef QS_RM
#undef QS_R
    qs_set_error("w4a8 gemm (ring): unsupported geometry mt=%d wn=%d", mt, wn);
    return QS_ENOSUP;
}
