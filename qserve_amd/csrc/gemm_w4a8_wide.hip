// gemm_w4a8_wide.hip -- compute-bound W4A8 GEMM, round-5 tile: FOUR waves per workgroup, one per SIMD, 512 registers each.
//
// Same arithmetic, operand mapping, LDS images and DMA pipeline as gemm_w4a8_tiled.hip (reference kernels
// kernels/csrc/qgemm/w4a8_per_chn/gemm_cuda.cu:303-594 - main loop :427-513 -, w4a8_per_group/gemm_cuda.cu:328-628 - level-2
// dequant :271-326).  What changes is who computes what:
//   * the 256-token x 256-channel workgroup tile is cut into four 256 x 64 WAVE tiles (wave = one 64-channel unit x all 16
//     m-tiles): 256 int32 accumulators per lane = the whole accumulator file a[0:255], addressed by INLINE-ASM MFMAs with fixed
//     register numbers.  (As C++ values the allocator cannot coalesce 64 loop-carried 4-register tuples: it rotates half of them
//     through v_accvgpr_mov copies - 212 copies + 108 s_nop per 64 MFMAs in scripts/microbench_mfma3.hip, which is what round 2
//     read as "one wave per SIMD issues at half rate" and why this tile was not built then; scripts/microbench_mfma5.hip has
//     the clean stream.)
//   * per 64-k stage a wave issues 64 MFMAs against 16 + 4 LDS operand reads and ONE unpack (level-2 dequant) of its unit's
//     weights, where the 8-wave tile issues 2 x 32 MFMAs against 2 x 12 reads and the unpack twice per unit (both token halves
//     of a unit each build the operands): 0.31 reads per MFMA instead of 0.375, half the unpack / dequant VALU per MFMA - the
//     per-group loop was VALU-bound on exactly that.
//   * one wave per SIMD has nobody to hide its stalls, so the stage is written as a fixed interleave: after every MFMA at most
//     a few independent instructions (the next operand read, a DMA issue, a slice of the unpack), in source order; memory
//     operations and the asm MFMAs keep that order through the compiler (both are ordered side effects for its scheduler).
// Everything else - 6-deep weight ring / 3-deep activation pair ring filled by LDS-DMA in whole 128-byte lines, counted vmcnt,
// one raw barrier per stage, 16 x 16 super-tile walk, next tile's fill issued before the epilogue, LDS-staged fp16 rows - is
// the tiled kernel's, re-derived for 256 threads (4 KiB per all-thread DMA instruction instead of 8).
#include "common.h"
#include <type_traits>
#include <utility>

qs_flag g_wide_order = 0;  // tile order A/B (same meaning as g_tiled_order % 10)
qs_flag g_wide_dbg = 0;    // QS_TIMING builds only (qs_set_gemm_variant(3400 + bits), results WRONG by design): 1 no MFMA, 2 no DMA,
                       // 4 no activation operand reads, 8 no barrier, 16 no weight reads / unpack

namespace {

constexpr int NS = 6;                      // weight ring depth (stages of 64 k); the activation ring holds NS/2 stage pairs
constexpr int PD = 4;                      // operand LDS reads run this many m-tiles ahead of the MFMAs
constexpr int BN = 256;                    // channels per workgroup (4 units, one per wave)
constexpr int BM = 256;                    // tokens per workgroup
constexpr int MT = BM / 16;                // m-tiles per wave
constexpr int WSTAGE = BN * 32;            // packed weight bytes per stage = 8 KiB
constexpr int APAIR = BM * 128;            // activation bytes per stage pair (128 k) = 32 KiB
constexpr int NA2 = APAIR / 4096;          // 4 KiB all-thread DMA instructions per activation pair

typedef __attribute__((address_space(3))) void* lptr_t;
typedef u32 v2u __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vm_dyn(int n) {   // pipeline fill and drain only; the steady state uses constants
    switch (n) {
#define QS_W(N) case N: wait_vm<N>(); break
        QS_W(1); QS_W(2); QS_W(3); QS_W(4); QS_W(5); QS_W(6); QS_W(7); QS_W(8); QS_W(9); QS_W(10); QS_W(11); QS_W(12);
        QS_W(13); QS_W(14); QS_W(15); QS_W(16); QS_W(17); QS_W(18); QS_W(19); QS_W(20); QS_W(21); QS_W(22); QS_W(23); QS_W(24);
#undef QS_W
    default: wait_vm<0>(); break;
    }
}
__device__ __forceinline__ void raw_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#define QS_PIN() __builtin_amdgcn_sched_barrier(0)

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>) - the accumulator register numbers of the asm MFMAs
// must be immediates
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// acc(mt, cl) = a[(4 mt + cl) 4 .. + 3].  FIRST: the tile's first stage writes A x B + 0 (no zero-fill of 256 registers).
// No hazard padding is needed inside: the operands are LDS read results or were built a whole stage earlier (the disassembly
// contract in tests/test_kernel_contracts.py checks that no VALU write of a source precedes an MFMA by fewer than two
// instructions), and an accumulate chain on the same registers needs no wait states.
template <int ACC, bool FIRST>
__device__ __forceinline__ void mfma_acc(const v4i& a, const v4i& b) {
    if constexpr (FIRST)
        asm volatile("v_mfma_i32_16x16x64_i8 a[%c0:%c1], %2, %3, 0" ::"n"(ACC), "n"(ACC + 3), "v"(a), "v"(b));
    else
        asm volatile("v_mfma_i32_16x16x64_i8 a[%c0:%c1], %2, %3, a[%c0:%c1]" ::"n"(ACC), "n"(ACC + 3), "v"(a), "v"(b));
}
template <int ACC>
__device__ __forceinline__ v4i acc_read() {
    int x0, x1, x2, x3;
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\t"
                 "v_accvgpr_read_b32 %3, a[%c7]"
                 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3)
                 : "n"(ACC), "n"(ACC + 1), "n"(ACC + 2), "n"(ACC + 3));
    return (v4i){x0, x1, x2, x3};
}

template <int MODE, int OUTK, int DBG = 0>
__global__ __launch_bounds__(256, 1) void w4a8_gemm_wide(const int8_t* __restrict__ A, const uint8_t* __restrict__ W,
                                                         const int8_t* __restrict__ zeros,
                                                         const int8_t* __restrict__ scales8,
                                                         const __half* __restrict__ wscales,
                                                         const __half* __restrict__ ascales,
                                                         const __half* __restrict__ wszs,
                                                         const __half* __restrict__ assums, void* __restrict__ out,
                                                         int M, int N, int K, int nbm, int order, int epi_fma,
                                                         unsigned long long* __restrict__ clk) {
    // clk != nullptr (qs_debug_gemm_clock_probe, bench.py): this workgroup's life in shader cycles (s_memtime) and in ticks of
    // the constant 100 MHz counter (s_memrealtime) -> the engine clock the launch actually held.  Four scalar instructions.
    unsigned long long ck0 = 0, rt0 = 0;
    if (clk) {
        ck0 = __builtin_amdgcn_s_memtime();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    constexpr int NW = 2 + (MODE == 1 ? 1 : 0);       // DMA instructions per wave and stage for the weights (+ per-group meta)
    static_assert(NA2 + NW <= MT && MT % PD == 0 && NS % 2 == 0, "pipeline slots");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    asm volatile("" ::: "a0", "a255");                // the kernel owns the whole accumulator file (descriptor: 256 AGPRs)
    uint8_t* const a_ring = smem;                     // [NS/2][APAIR]  pair image, see gemm_w4a8_tiled.hip
    uint8_t* const w_ring = smem + (NS / 2) * APAIR;  // [NS][unit 4][tile 2][e 4][k32^tile 2][c 8][16 B]
    uint8_t* const m_ring = w_ring + NS * WSTAGE;     // [NS][512]: 256 scales | 256 zeros (storage order)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = the wave's 64-channel unit of the tile
    const int li = lane & 15, g = lane >> 4;
    const int tsel = li >> 3, c = li & 7;
    constexpr bool PERSIST = OUTK != 1;
    const int ntiles = nbm * (N / BN);
    auto tile_coords = [&](int id, int& bm, int& bn) {
        const int nbn = N / BN;
        if (order == 1) {
            bm = id % nbm, bn = id / nbm;
        } else if (order == 2) {
            bn = id % nbn, bm = id / nbn;
        } else {                                              // 16 x 16 super-tiles (gemm_w4a8_tiled.hip)
            const int sm = 16, sn = 16;
            const int full_m = nbm / sm, rem_m = nbm % sm;
            const int per_row = sm * nbn;
            int srow = id / per_row, in_row = id % per_row, hgt = sm;
            if (srow >= full_m) {
                srow = full_m;
                in_row = id - full_m * per_row;
                hgt = rem_m;
            }
            const int per_st = hgt * sn;
            int scol = in_row / per_st, in_st = in_row % per_st;
            const int full_n = nbn / sn;
            if (scol >= full_n) {
                scol = full_n;
                in_st = in_row - full_n * per_st;
            }
            bm = srow * sm + in_st % hgt;
            bn = scol * sn + in_st / hgt;
            if (order != 3 && hgt == sm && scol < full_n) {
                // XCD-aware placement inside a full 16 x 16 super-tile (round 5; order 3 = without it, A/B): workgroup b runs on
                // XCD b % 8 (observed; speed only), so the 32 tiles of one XCD form a 4 x 8 block - 4 activation + 8 weight tiles
                // through that L2 (8 MB at K = 4096) instead of 2 + 16 (10 MB).  In-run: per-channel +0.6 ... +1.6 % on the prompt
                // shapes, 4096^3 56.4 -> 54.2 us; per-group unchanged (profiles/round5_tile_order.txt)
                const int x = in_st & 7, j = in_st >> 3;
                bm = srow * sm + 4 * (x & 3) + (j & 3);
                bn = scol * sn + 8 * (x >> 2) + (j >> 2);
            }
        }
    };
    constexpr bool ACT = OUTK == 2;                   // gate_up + silu * mul: see gemm_w4a8_ring.hip
    auto trow = [&](int unit, int t) { return ACT ? (t ? N / 64 + unit : unit) : unit * 2 + t; };
    auto chan32 = [&](int unit, int t) { return ACT ? (t ? N / 2 + 32 * unit : 32 * unit) : unit * 64 + 32 * t; };
    int m0, n0;
    const int KT = K >> 5;
    const int nh = K >> 6;                            // stages (even: K % 128 == 0)

    auto aswz = [&](int j) { return (0 - j) & 3; };
    u32 a_off[NA2];
    const uint8_t* w_base;
    const int8_t* m_base;
    u32 w_off[2], m_off;
    {
        // weights: this wave copies both tile rows (t = 0, 1: one 1 KiB piece each) of ITS unit
        const int e = lane >> 4, cc = lane & 7;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int kk = ((lane >> 3) & 1) ^ t;
            w_off[t] = ((u32)trow(wave, t) * (u32)KT + kk) * 512u + cc * 64 + e * 16;
        }
        m_off = chan32(lane >> 4, (lane >> 3) & 1) + (lane & 7) * 4;   // dword `lane` of the tile's 64
    }
    auto fresh = [](int v) {
        asm volatile("" : "+v"(v));
        return v;
    };
    // activation instruction i of a wave copies piece 4 i + wave = tile rows 32 i + 8 wave .. + 7
    auto setup = [&](int id) {
        int bm, bn;
        tile_coords(id, bm, bn);
        m0 = bm * BM, n0 = bn * BN;
        const int ln = fresh(lane);
        const int a_r0 = wave * 8 + 2 * (ln >> 4) + ((ln >> 2) & 1);
        const u32 a_c = ((((ln >> 3) & 1) ^ ((ln >> 4) & 1)) * 64) + (((ln & 3) ^ aswz((a_r0 >> 2) & 3)) * 16);
#pragma unroll
        for (int i = 0; i < NA2; ++i) {
            int row = m0 + a_r0 + 32 * i;
            row = row < M ? row : M - 1;
            a_off[i] = __umul24((u32)row, (u32)K) + a_c;           // M, K < 2^24 and M * K < 2^32 (checked by the dispatcher)
        }
        w_base = W + (size_t)trow(n0 / 64, 0) * KT * 512;
        m_base = ((wave & 1) ? zeros : scales8) + chan32(n0 / 64, 0);
    };
    const u32 lds0 = (u32)(size_t)(lptr_t)smem;

    // ---- DMA issue -----------------------------------------------------------------------------------------------------
    // LDS destinations = a per-wave scalar base + a COMPILE-TIME offset wherever the ring slot is a compile-time constant (the
    // steady state is unrolled over the six slots): `s_add_u32 m0, base, imm` replaces the address arithmetic + s_mov, and the
    // wait state M0 needs before the LDS-DMA is an MFMA of the stream instead of an s_nop (mfma_dma: both live in ONE asm
    // statement - M0 is not preserved between statements).  Round-5 ablation (profiles/round5_wide_ablation.txt): the first
    // version of this kernel spent 126 non-MFMA instructions per 64 MFMAs, 40 of them on ten DMA issues, and ran at the
    // eight-wave tile's speed; one wave per SIMD hides about one instruction per MFMA, every further one costs 4-5 cycles.
    const u32 lds_a = lds0 + wave * 1024;                                          // + pslot*APAIR + i*4096
    const u32 lds_w = lds0 + (NS / 2) * APAIR + wave * 2048;                       // + slot*WSTAGE + i*1024
    const u32 lds_m = lds0 + (NS / 2) * APAIR + NS * WSTAGE + (wave & 1) * 256;    // + slot*512
    auto dma16 = [&](u32 voff, const void* sbase, u32 lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr)
                     : "memory");
    };
    auto dma4 = [&](u32 voff, const void* sbase, u32 lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr)
                     : "memory");
    };
    auto a_src = [&](int pr) { return static_cast<const void*>(A + (size_t)pr * 128); };
    auto w_src = [&](int u) { return static_cast<const void*>(w_base + (size_t)u * 1024); };
    auto m_src = [&](int u) { return static_cast<const void*>(m_base + (size_t)(u >> 1) * N); };
    auto issue_a = [&](int pr, int pslot, int i) {                 // instruction i of activation pair pr (stages 2pr, 2pr+1)
        dma16(a_off[i], a_src(pr), lds_a + pslot * APAIR + i * 4096);
    };
    auto issue_w = [&](int u, int slot, int i) {                   // i = 0, 1: weight pieces of stage u, 2: its per-group meta
        if (i < 2) dma16(w_off[i], w_src(u), lds_w + slot * WSTAGE + i * 1024);
        else dma4(m_off, m_src(u), lds_m + slot * 512);
    };
    // issue order of a wave and what a stage needs: gemm_w4a8_tiled.hip (`allowed`), with NW / NA2 of this geometry
    auto allowed = [&](int v) {
        const int nw = v + 5 < nh ? v + 5 : nh, na = 2 + ((v + 1) >> 1) < (nh >> 1) ? 2 + ((v + 1) >> 1) : (nh >> 1);
        const int q = v >> 1;
        const int need = (v & 1) ? NW + (q + 1) * (NA2 + 2 * NW) + NA2 : 2 * NW + q * (NA2 + 2 * NW) + NA2;
        return NW * nw + NA2 * na - need;
    };

    // ---- LDS operand readers ----------------------------------------------------------------------------------------
    const int w_rd = wave * 2048 + tsel * 1024 + (((g >> 1) ^ tsel)) * 128 + c * 16 + (g & 1) * 8;   // + e*256
    const int m_rd = wave * 64 + (tsel * 8 + c) * 4;
    const int a_rd0 = (li >> 3) * 1024 + (4 * ((li & 7) >> 1) + 2 * ((li >> 1) & 1) + (li & 1)) * 64 +
                      ((g ^ aswz(li >> 2)) * 16);                                                    // half 0; + mt*2048
    // two base registers (one per k half of a pair); pair slot and m-tile are immediates of the read wherever they are
    // compile-time constants
    const uint8_t* const a_rd_p[2] = {a_ring + a_rd0, a_ring + (a_rd0 ^ 128)};
    auto read_b = [&](int pslot, int half, int mt) -> v4i {
        return *reinterpret_cast<const v4i*>(a_rd_p[half] + pslot * APAIR + mt * 2048);
    };
    struct Raw {
        v2u r[4];
        u32 sdw, zdw;
    };
    int w_rd_e[4];                                   // separate address registers: no ds_read2_b64 merging (tiled kernel)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        w_rd_e[e] = w_rd + e * 256;
        asm volatile("" : "+v"(w_rd_e[e]));
    }
    auto read_w = [&](int slot) -> Raw {
        Raw q;
#pragma unroll
        for (int e = 0; e < 4; ++e) q.r[e] = *reinterpret_cast<const v2u*>(w_ring + slot * WSTAGE + w_rd_e[e]);
        q.sdw = 0;
        q.zdw = 0;
        if (MODE == 1) {
            q.sdw = *reinterpret_cast<const u32*>(m_ring + slot * 512 + m_rd);
            q.zdw = *reinterpret_cast<const u32*>(m_ring + slot * 512 + 256 + m_rd);
        }
        return q;
    };
    auto build1 = [&](const Raw& q, int cl, int e) -> int {      // register e (4 k) of row class cl
        u32 s = 0, zb = 0;
        if (MODE == 1) {
            s = (q.sdw >> (8 * cl)) & 0xFFu;
            zb = ((q.zdw >> (8 * cl)) & 0xFFu) * 0x01010101u;
        }
        const u32 raw = (cl & 1) ? q.r[e].y : q.r[e].x;
        return (int)((cl & 2) ? unpack_hi<MODE>(raw, s, zb) : unpack_lo<MODE>(raw, s, zb));
    };
    auto build = [&](const Raw& q, int cl) -> v4i {
        v4i a;
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = build1(q, cl, e);
        return a;
    };

    constexpr int NB = 8;                             // activation operand buffers (tile t lives in bq[t % NB]; NB divides MT)
    v4i a0[4], a1[4], bq[NB];

    auto issue_fill = [&]() {
#pragma unroll
        for (int i = 0; i < NW; ++i) issue_w(0, 0, i);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int i = 0; i < NA2; ++i) issue_a(q, q, i);
#pragma unroll
            for (int t = 1; t <= 2; ++t)
                if (2 * q + t < nh) {
#pragma unroll
                    for (int i = 0; i < NW; ++i) issue_w(2 * q + t, 2 * q + t, i);
                }
        }
    };

    // One stage = 64 MFMAs of this wave (16 m-tiles x 4 row classes) with, in source order (memory operations and asm statements
    // keep it through the compiler), per m-tile mt:
    //     MFMA cl 0 | activation operand reads: at EVEN mt the tiles mt + 5, mt + 4 (in that order: the wait for the second covers
    //     the first, so the compiler waits once per pair) | MFMA cl 1 - with this m-tile's DMA issue wrapped around it (PAR 0:
    //     activation pieces at mt 0 .. 7, weights / meta of stage u + 5 at mt 8 .. 10; PAR 1: weights at mt 0 .. 2) | raw weight
    //     words of stage u + 1 (mt 4) or half a row class of its unpack / level-2 dequant (mt 8 .. 15: class (mt - 8) / 2,
    //     registers 2 (mt & 1), + 1) | MFMA cl 2 | MFMA cl 3.
    // slot_c: the ring slot as a compile-time constant (the unrolled steady state) or a run-time int (first pair, drain).
    auto stage = [&](auto par_c, auto first_c, auto pref_static, auto slot_c, bool pref_a, bool pref_w, int u, v4i(&ac)[4],
                     v4i(&an)[4], Raw& qn) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr bool st = decltype(pref_static)::value;
        constexpr bool SLOT_CT = !std::is_same<decltype(slot_c), int>::value;
        const int slot = slot_c;
        const int slot_n = slot + 1 == NS ? 0 : slot + 1;
        const int slot_d = slot == 0 ? NS - 1 : slot - 1;          // (u + NS - 1) % NS
        const int ps = slot >> 1, ps_n = slot_n >> 1;              // activation pair slots of stage u / u+1
        const int ps_d = ps == 0 ? NS / 2 - 1 : ps - 1;            // (u/2 + 2) % (NS/2)
        static_for<MT>([&](auto mt_c) {
            constexpr int mt = decltype(mt_c)::value;
            const v4i b_use = bq[mt % NB];
            if constexpr (!(DBG & 1)) mfma_acc<(4 * mt + 0) * 4, FIRST>(ac[0], b_use);
            if constexpr (!(DBG & 4) && mt % 2 == 0) {
#pragma unroll
                for (int t = mt + 5; t >= mt + 4; --t) {
                    if (t < MT) bq[t % NB] = read_b(ps, PAR, t);
                    else bq[t % NB] = read_b(ps_n, PAR ^ 1, t - MT);
                }
            }
            // this m-tile's DMA, wrapped around MFMA cl 1
            constexpr int di = mt - (PAR == 0 ? NA2 : 0);          // weight / meta instruction index of this m-tile
            constexpr bool is_a = PAR == 0 && mt < NA2;
            constexpr bool is_w = !is_a && di >= 0 && di < NW;
            bool fused = false;
            if constexpr (!(DBG & 2) && !(DBG & 1) && (is_a || is_w)) {
                if (st || (is_a ? pref_a : pref_w)) {
                    fused = true;
                    const u32 voff = is_a ? a_off[is_a ? mt : 0] : (di < 2 ? w_off[di < 2 ? (di < 0 ? 0 : di) : 0] : m_off);
                    const void* const src = is_a ? a_src((u >> 1) + 2) : (di < 2 ? w_src(u + NS - 1) : m_src(u + NS - 1));
                    const u32 lbase = is_a ? lds_a : (di < 2 ? lds_w : lds_m);
                    const int loff = is_a ? ps_d * APAIR + mt * 4096 : (di < 2 ? slot_d * WSTAGE + di * 1024 : slot_d * 512);
                    // m0 = LDS destination (scalar base + compile-time offset where the slot is a compile-time constant) | the MFMA
                    // (= the wait state M0 needs) | the LDS-DMA
#define QS_MFMA_DMA(LOADOP, CARG, LB, LO)                                                                                         \
    asm volatile("s_add_u32 m0, %4, %5\n\tv_mfma_i32_16x16x64_i8 a[%c0:%c1], %2, %3, " CARG "\n\t" LOADOP " %6, %7" ::"n"((4 * mt + 1) * 4), \
                 "n"((4 * mt + 1) * 4 + 3), "v"(ac[1]), "v"(b_use), "s"(LB), "n"(LO), "v"(voff), "s"(src)                          \
                 : "memory", "scc")
#define QS_MFMA_DMA_F(LOADOP, LB, LO)                                          \
    do {                                                                       \
        if constexpr (FIRST) QS_MFMA_DMA(LOADOP, "0", LB, LO);                 \
        else QS_MFMA_DMA(LOADOP, "a[%c0:%c1]", LB, LO);                        \
    } while (0)
                    if constexpr (SLOT_CT) {
                        constexpr int S = decltype(slot_c)::value, SD = S == 0 ? NS - 1 : S - 1, PSD = (S >> 1) == 0 ? NS / 2 - 1 : (S >> 1) - 1;
                        constexpr int LOFF = is_a ? PSD * APAIR + mt * 4096 : (di < 2 ? SD * WSTAGE + di * 1024 : SD * 512);
                        if constexpr (is_a || di < 2) QS_MFMA_DMA_F("global_load_lds_dwordx4", lbase, LOFF);
                        else QS_MFMA_DMA_F("global_load_lds_dword", lbase, LOFF);
                    } else {
                        const u32 laddr = lbase + loff;
                        if constexpr (is_a || di < 2) QS_MFMA_DMA_F("global_load_lds_dwordx4", laddr, 0);
                        else QS_MFMA_DMA_F("global_load_lds_dword", laddr, 0);
                    }
#undef QS_MFMA_DMA_F
#undef QS_MFMA_DMA
                }
            }
            if constexpr (DBG & 1) {                                   // (timing build without MFMAs: plain DMA issue)
                if constexpr (!(DBG & 2) && (is_a || is_w)) {
                    if (st || (is_a ? pref_a : pref_w)) {
                        if constexpr (is_a) issue_a((u >> 1) + 2, ps_d, mt);
                        else issue_w(u + NS - 1, slot_d, di < 0 ? 0 : di);
                    }
                }
            } else {
                if (!fused) mfma_acc<(4 * mt + 1) * 4, FIRST>(ac[1], b_use);
            }
            if constexpr (mt == 4 && !(DBG & 16)) qn = read_w(slot_n);
            if constexpr (mt >= 8 && !(DBG & 16)) {
                constexpr int cl = (mt - 8) / 2, e0 = 2 * (mt & 1);
                // pinned HERE (asm with the value as in-out operand): left alone, the compiler sinks the unpack into the block of
                // its first use - straight in front of the asm MFMA that reads it, where nothing pads the VALU -> MFMA hazard
                an[cl][e0] = build1(qn, cl, e0);
                asm volatile("" : "+v"(an[cl][e0]));
            }
            if constexpr (!(DBG & 1)) mfma_acc<(4 * mt + 2) * 4, FIRST>(ac[2], b_use);
            if constexpr (mt >= 8 && !(DBG & 16)) {
                constexpr int cl = (mt - 8) / 2, e1 = 2 * (mt & 1) + 1;
                an[cl][e1] = build1(qn, cl, e1);
                asm volatile("" : "+v"(an[cl][e1]));
            }
            if constexpr (!(DBG & 1)) mfma_acc<(4 * mt + 3) * 4, FIRST>(ac[3], b_use);
            QS_PIN();
        });
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    int tile = blockIdx.x;
    setup(tile);
    if (!(DBG & 2)) issue_fill();
    bool first = true;
    auto stage_sync = [&](int n_static, int v) {       // counted wait + barrier in front of stage v (n_static < 0: by formula)
        if (DBG & 2) wait_vm<0>();
        else if (n_static >= 0) wait_vm<3 * NW + NA2>();
        else wait_vm_dyn(allowed(v));
        if (!(DBG & 8)) raw_barrier();
    };
    Raw qraw;                                          // raw weight words of the next stage (read at mt 4, unpacked at mt 8 .. 15)
    while (true) {
        // first tile: W(0), A(0), W(1) have landed, the rest of the fill stays in flight.  Later tiles: the fill was issued
        // before the previous tile's epilogue - everything (its stores included) is complete
        if (first && !(DBG & 2)) wait_vm_dyn(allowed(0));
        else wait_vm<0>();
        first = false;
        raw_barrier();
        {
            const Raw q0 = read_w(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) bq[t] = read_b(0, 0, t);
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) a0[cl] = build(q0, cl);
            // a VALU result needs two wait states before an MFMA reads it as a source operand; nothing pads that for an asm MFMA
            asm volatile("s_nop 1" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]));
        }
        // the tile's first pair: stage 0 writes the accumulators (A x B + 0); prefetch conditions evaluated (nh may be small)
        stage(c0{}, std::true_type{}, std::false_type{}, 0, 4 < nh, NS - 1 < nh, 0, a0, a1, qraw);
        stage_sync(-1, 1);
        stage(c1{}, std::false_type{}, std::false_type{}, 1, false, NS < nh, 1, a1, a0, qraw);
        int u = 2;
        // steady state, unrolled over the six ring slots (u = 2 (mod 6) here: slots 2, 3, 4, 5, 0, 1): every LDS address of
        // the stage is an immediate; every stage of the block prefetches (u + 5 + 5 < nh)
        for (; u + 11 <= nh; u += 6) {
            static_for<6>([&](auto k_c) {
                constexpr int k = decltype(k_c)::value;
                constexpr int S = (2 + k) % NS;
                stage_sync(0, u + k);
                if constexpr (k % 2 == 0) stage(c0{}, std::false_type{}, std::true_type{}, std::integral_constant<int, S>{}, true, true, u + k, a0, a1, qraw);
                else stage(c1{}, std::false_type{}, std::true_type{}, std::integral_constant<int, S>{}, true, true, u + k, a1, a0, qraw);
            });
        }
        int slot = 2;                                  // (u = 2 (mod 6) again)
        for (; u + NS < nh; u += 2) {                  // the last full-prefetch pairs, run-time slot
            stage_sync(0, u);
            stage(c0{}, std::false_type{}, std::true_type{}, slot, true, true, u, a0, a1, qraw);
            slot = slot + 1 == NS ? 0 : slot + 1;
            stage_sync(0, u + 1);
            stage(c1{}, std::false_type{}, std::true_type{}, slot, true, true, u + 1, a1, a0, qraw);
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
        for (; u < nh; u += 2) {                       // drain
            stage_sync(-1, u);
            stage(c0{}, std::false_type{}, std::false_type{}, slot, u + 4 < nh, u + NS - 1 < nh, u, a0, a1, qraw);
            slot = slot + 1 == NS ? 0 : slot + 1;
            stage_sync(-1, u + 1);
            stage(c1{}, std::false_type{}, std::false_type{}, slot, false, u + NS < nh, u + 1, a1, a0, qraw);
            slot = slot + 1 == NS ? 0 : slot + 1;
        }

        // ---- fused epilogue -------------------------------------------------------------------------------------
        const int em0 = m0, en0 = n0;                  // (m0 / n0 move on to the next tile below)
        const int next = tile + (int)gridDim.x;
        const int lane_e = fresh(lane);
        const int li = lane_e & 15, g = lane_e >> 4;   // (shadow the loop's copies, see `fresh`)
        const int ncol0 = chan32(en0 / 64 + wave, g >> 1) + 4 * (g & 1);
        const int mrow0 = em0 + li;
        // the last MFMA's result registers: 4 passes + read of the accumulator file by a VALU instruction
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        if (OUTK == 1) {
            static_for<MT>([&](auto mt_c) {
                constexpr int mt = decltype(mt_c)::value;
                const int m = mrow0 + 16 * mt;
                static_for<4>([&](auto cl_c) {
                    constexpr int cl = decltype(cl_c)::value;
                    const v4i s = acc_read<(4 * mt + cl) * 4>();
                    if (m < M) *reinterpret_cast<v4i*>(reinterpret_cast<int*>(out) + (size_t)m * N + ncol0 + 8 * cl) = s;
                });
            });
            if (next >= ntiles) break;
            raw_barrier();                             // every wave left the k loop: the rings may be refilled
            tile = next;
            setup(tile);
            issue_fill();
            continue;
        }
        // all scale loads first (one latency), pinned as complete BEFORE the next tile's fill is issued: a wait the
        // compiler places after the fill would also wait for the fill (vmcnt retires in order)
        h4 ws4[4], wz4[4];
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            ws4[cl] = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(wscales) + ncol0 + 8 * cl);
            if (MODE == 0) wz4[cl] = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(wszs) + ncol0 + 8 * cl);
        }
        _Float16 sa_h[MT], ss_h[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int m = mrow0 + 16 * mt;
            m = m < M ? m : M - 1;
            sa_h[mt] = reinterpret_cast<const _Float16*>(ascales)[m];
            if (MODE == 0) ss_h[mt] = reinterpret_cast<const _Float16*>(assums)[m];
        }
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            asm volatile("" : "+v"(ws4[cl]));
            if (MODE == 0) asm volatile("" : "+v"(wz4[cl]));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            asm volatile("" : "+v"(sa_h[mt]));
            if (MODE == 0) asm volatile("" : "+v"(ss_h[mt]));
        }
        raw_barrier();                                 // the rings are dead: every wave left the k loop
        if (next < ntiles) {                           // next tile's fill: pair slots 0, 1 and weight slots 0..4
            setup(next);
            if (!(DBG & 2)) issue_fill();
        }
        // The fp16 tile of this wave goes through LDS, 16 tokens x 64 channels at a time (whole 128-byte rows per store
        // instruction).  The staging rows live in activation pair slot 2, which the fill does not touch; written and read by
        // the same wave: an LDS wait, no barrier.
        constexpr int RS = 144;                        // staged row stride (bytes): 128 + 16 keeps 16-byte alignment
        uint8_t* const st = a_ring + 2 * APAIR + wave * (16 * RS);
        _Float16* const orow = reinterpret_cast<_Float16*>(out) + en0 + wave * 64 + (lane_e & 7) * 8;
        static_for<MT>([&](auto mt_c) {
            constexpr int mt = decltype(mt_c)::value;
            const float sa = (float)sa_h[mt];
            const float ss = MODE == 0 ? (float)ss_h[mt] : 0.f;
            static_for<4>([&](auto cl_c) {
                constexpr int cl = decltype(cl_c)::value;
                const v4i s = acc_read<(4 * mt + cl) * 4>();
                h4 o;
                if (MODE == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (_Float16)epi_per_chn(s[r], (float)ws4[cl][r], sa, (float)wz4[cl][r], ss, epi_fma);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (_Float16)epi_per_group(s[r], (float)ws4[cl][r], sa);
                }
                if (ACT) {     // lanes 0-31: gate, lanes 32-63: up of the same (token, channel) -> silu_and_mul's arithmetic
                    const v2u ob = __builtin_bit_cast(v2u, o);
                    const auto sw = __builtin_amdgcn_permlane32_swap(ob.x, ob.y, false, false);
                    const h2 gt = __builtin_bit_cast(h2, (u32)sw[0]), up = __builtin_bit_cast(h2, (u32)sw[1]);
                    const int hh = g >> 1;
                    h2 a;
                    a[0] = (_Float16)((float)qs_silu_h((float)gt[0]) * (float)up[0]);
                    a[1] = (_Float16)((float)qs_silu_h((float)gt[1]) * (float)up[1]);
                    *reinterpret_cast<h2*>(st + li * RS + (8 * cl + 4 * (g & 1) + 2 * hh) * 2) = a;
                } else {
                    *reinterpret_cast<h4*>(st + li * RS + (32 * (g >> 1) + 8 * cl + 4 * (g & 1)) * 2) = o;
                }
            });
            if (ACT) {         // 16 tokens x 32 channels of this wave: 64-byte row pieces (four waves complete a 256-byte row)
                const int r = lane_e >> 2;
                const int m = em0 + 16 * mt + r;
                const v4u v = *reinterpret_cast<const v4u*>(st + r * RS + (lane_e & 3) * 16);
                if (m < M)
                    *reinterpret_cast<v4u*>(reinterpret_cast<_Float16*>(out) + (size_t)m * (N / 2) + (en0 / 64 + wave) * 32 +
                                            (lane_e & 3) * 8) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = i * 8 + (lane_e >> 3);
                    const int m = em0 + 16 * mt + r;
                    const v4u v = *reinterpret_cast<const v4u*>(st + r * RS + (lane_e & 7) * 16);
                    if (m < M) *reinterpret_cast<v4u*>(orow + (size_t)m * N) = v;
                }
            }
        });
        if (!PERSIST || next >= ntiles) break;
        tile = next;
    }
    if (clk && threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - ck0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

template <int MODE, int OUTK, int DBG = 0>
int launch_wide(const int8_t* A, const uint8_t* W, const int8_t* zeros, const int8_t* scales8, const void* wscales,
                const void* ascales, const void* wszs, const void* assums, void* out, int M, int N, int K, int persist_mode,
                hipStream_t stream) {
    auto kern = w4a8_gemm_wide<MODE, OUTK, DBG>;
    const size_t smem = (size_t)NS * (BM * 64 + WSTAGE + 512);   // (the epilogue's staging rows alias activation pair slot 2)
    static bool configured_dev[QS_MAX_DEVICES] = {};   // the attribute belongs to the (kernel, device) pair
    bool& configured = configured_dev[qs_device_slot()];
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            qs_set_error("w4a8 gemm (wide): cannot reserve %zu bytes of LDS: %s", smem, hipGetErrorString(e));
            return (int)e;
        }
        configured = true;
    }
    const int nbm = (M + BM - 1) / BM;
    const int ntiles = nbm * (N / BN);
    const int cus = qs_num_cus();
    // persist_mode 0: one workgroup per CU walks the tile list, 1: one per tile, 2: three workgroups walk all tiles (tests of
    // the tile-to-tile hand-over); the int32-output form has no hand-over (one workgroup per tile)
    const int walkers = persist_mode == 2 ? 3 : cus;
    dim3 grid(OUTK != 1 && persist_mode != 1 && ntiles > walkers ? walkers : ntiles);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, A, W, zeros, scales8,
                       reinterpret_cast<const __half*>(wscales), reinterpret_cast<const __half*>(ascales),
                       reinterpret_cast<const __half*>(wszs), reinterpret_cast<const __half*>(assums), out, M, N, K,
                       nbm, g_wide_order, g_epi_fma, grid.x <= (unsigned)g_gemm_clk_cap ? g_gemm_clk : nullptr);
    return qs_launch_status("w4a8 gemm (wide)");
}

}  // namespace

// Entry used by the dispatcher in gemm_w4a8.hip.  Preconditions (checked there): N % 256 == 0, K % 128 == 0, K >= 256,
// M * K and N * K / 2 below 4 GiB.
int qs_launch_gemm_wide(int mode, int outk, const int8_t* A, const uint8_t* W, const int8_t* zeros,
                        const int8_t* scales8, const void* wscales, const void* ascales, const void* wszs,
                        const void* assums, void* out, int M, int N, int K, int persist_mode, hipStream_t stream) {
    if (g_qs_plan.active) {
        g_qs_plan.family = 5;
        g_qs_plan.p[0] = 8, g_qs_plan.p[1] = g_qs_plan.p[2] = g_qs_plan.p[3] = 0;
        return QS_OK;
    }
#define QS_T(MODEV, OUTV) \
    return launch_wide<MODEV, OUTV>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, persist_mode, stream)
#ifdef QS_TIMING   // timing experiments (results are wrong by design): not in the shipped library
    if (mode == 0 && outk == 0 && g_wide_dbg) {
#define QS_D(D) case D: return launch_wide<0, 0, D>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, persist_mode, stream)
        switch (g_wide_dbg) {
            QS_D(1); QS_D(2); QS_D(4); QS_D(8); QS_D(16); QS_D(20); QS_D(6); QS_D(22); QS_D(30);
        default: break;
        }
#undef QS_D
    }
#endif
    if (mode == 0 && outk == 2) QS_T(0, 2);
    if (mode == 1 && outk == 2) QS_T(1, 2);
    if (mode == 0 && outk == 0) QS_T(0, 0);
    if (mode == 0 && outk == 1) QS_T(0, 1);
    if (mode == 1 && outk == 0) QS_T(1, 0);
    QS_T(1, 1);
#undef QS_T
}
