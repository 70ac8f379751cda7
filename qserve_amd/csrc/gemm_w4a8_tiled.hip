// gemm_w4a8_tiled.hip -- compute-bound W4A8 GEMM (prefill shapes, BASELINE config 1 = 4096^3): INT8 MFMA, both operands
// staged through LDS by LDS-DMA, software-pipelined so that LDS reads, DMA issue and operand unpacking all hide under
// the MFMA stream.
//
// Same arithmetic and operand mapping as the decode kernels (gemm_w4a8.hip header; reference kernels
// kernels/csrc/qgemm/w4a8_per_chn/gemm_cuda.cu:303-594 with its 128x128x64 tile for M > 256, and
// w4a8_per_group/gemm_cuda.cu:328-628).  Tile geometry chosen for CDNA4, not translated from the reference:
//   * workgroup = 8 wave64 (512 threads) computes 256 tokens x 256 channels; wave (wm, wn) owns 128 tokens x one
//     64-channel unit: 8 m-tiles x 4 row classes of v_mfma_i32_16x16x64_i8 accumulators (128 VGPRs).
//   * pipeline stage = 64 k ("half-step", one MFMA k-slice): 16 KiB of activations + 8 KiB of packed weights (+512 B of
//     per-group scales) in LDS rings filled by global_load_lds: the weights in a 6-deep ring of stages, the activations
//     in a 3-deep ring of stage PAIRS - one DMA instruction copies 8 token rows x 128 B, i.e. whole 128-byte lines
//     (a 64-k stage on its own is half a line per row: every line would be requested twice from L2, and with all 256
//     CUs doing so the L2s deliver 12.5 TB/s instead of 25-29 TB/s - scripts/microbench_cufill.hip).  One raw
//     s_barrier per stage; counted s_waitcnt vmcnt keeps four stages (~97 KiB per CU) in flight across it.
//   * inside a stage every wave runs 32 MFMAs; between them it issues the LDS reads of operands needed 3 m-tiles ahead
//     (rolling into the next stage), the next stage's weight nibbles + their unpacking, and its share of the DMA for
//     the stage five ahead - pinned in that order with sched_barrier so the matrix pipe never waits at a stage edge.
//   * LDS images are bank-conflict free by construction: the DMA writes lane-linear, so the permutation is applied to
//     the per-lane SOURCE address (activations: see "pair image" below; weights: [tile][chunk e][k32 ^ tile][c]).
//   * per-group: level-2 dequant in registers exactly as in the decode kernels (bit-faithful to the reference).
#include "common.h"
#include <type_traits>

qs_flag g_tiled_dbg = 0;   // qs_set_gemm_variant(3100 + bits): 1 no MFMA, 2 no DMA, 4 no operand reads, 8 no barrier
qs_flag g_tiled_order = 0; // qs_set_gemm_variant(3200 + 10 * p + mode): tile order A/B (mode 0 default, 1 M-fastest bands, 2 N-fastest
                       // bands); p = 1: one workgroup per tile instead of per CU, p = 2: three workgroups walk all tiles (tests)
namespace {

constexpr int NS = 6;                      // weight ring depth (stages of 64 k); the activation ring holds NS/2 stage pairs
constexpr int PD = 4;                      // operand LDS reads run this many m-tiles ahead of the MFMAs
constexpr int BN = 256;                    // channels per workgroup (4 units)
constexpr int WSTAGE = BN * 32;            // packed weight bytes per stage = 8 KiB


typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef u32 v2u __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// counted wait with a run-time count (pipeline fill and drain only; the steady state uses constants)
__device__ __forceinline__ void wait_vm_dyn(int n) {
    switch (n) {
#define QS_W(N) case N: wait_vm<N>(); break
        QS_W(1); QS_W(2); QS_W(3); QS_W(4); QS_W(5); QS_W(6); QS_W(7); QS_W(8); QS_W(9); QS_W(10); QS_W(11); QS_W(12);
        QS_W(13); QS_W(14); QS_W(15); QS_W(16);
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

// MT = m-tiles per wave (8 -> 256-token workgroup tile, 4 -> 128)
template <int MT, int MODE, int OUTK, int DBG = 0>
__global__ __launch_bounds__(512, 1) void w4a8_gemm_tiled(const int8_t* __restrict__ A, const uint8_t* __restrict__ W,
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
    constexpr int BM = 32 * MT;                       // tokens per workgroup
    constexpr int APAIR = BM * 128;                   // activation bytes per stage pair (128 k)
    constexpr int NA2 = APAIR / 8192;                 // 8 KiB all-thread DMA instructions per activation pair
    constexpr int NW = 1 + (MODE == 1 ? 1 : 0);       // DMA instructions per stage for the weights (+ per-group meta)
    static_assert(NA2 + NW <= MT && MT % PD == 0 && NS % 2 == 0, "pipeline slots");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const a_ring = smem;                     // [NS/2][APAIR]  pair image, see below
    uint8_t* const w_ring = smem + (NS / 2) * APAIR;  // [NS][unit 4][tile 2][e 4][k32^tile 2][c 8][16 B]
    uint8_t* const m_ring = w_ring + NS * WSTAGE;     // [NS][512]: 256 scales | 256 zeros (storage order)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, g = lane >> 4;
    const int tsel = li >> 3, c = li & 7;
    // A workgroup walks the tiles id = blockIdx.x, + gridDim.x, ... (PERSIST: one workgroup per CU, the next tile's pipeline
    // fill is issued before this tile's epilogue; otherwise gridDim.x = number of tiles and the loop runs once)
    constexpr bool PERSIST = MT == 8 && OUTK != 1 && !(DBG & 2);
    const int ntiles = nbm * (N / BN);
    auto tile_coords = [&](int id, int& bm, int& bn) {
        const int nbn = N / BN;
        if (order == 1) {                                 // round-1 order: M fastest inside a band of channels
            bm = id % nbm, bn = id / nbm;
        } else if (order == 2) {                          // N fastest inside a band of tokens
            bn = id % nbn, bm = id / nbn;
        } else {
            // 16 x 16 super-tiles (256 workgroups = one round of the chip): 16 activation tiles + 16 weight tiles = 24 MB at
            // K = 4096 are fetched once per super-tile and re-served by L2 / the Infinity Cache, instead of every
            // activation tile once per channel band (M = 65 536 x N = 28 672: 30 GB of activation reads -> ~3 GB)
            const int sm = 16, sn = 16;
            const int full_m = nbm / sm, rem_m = nbm % sm;        // super-rows; the last one may be narrower
            const int per_row = sm * nbn;                         // workgroups per full super-row
            int srow = id / per_row, in_row = id % per_row, hgt = sm;
            if (srow >= full_m) {                                 // remainder rows: height rem_m
                srow = full_m;
                in_row = id - full_m * per_row;
                hgt = rem_m;
            }
            const int per_st = hgt * sn;                          // workgroups per (full-width) super-tile of this row
            int scol = in_row / per_st, in_st = in_row % per_st, wid = sn;
            const int full_n = nbn / sn;
            if (scol >= full_n) {                                 // last, narrower super-tile of the row
                scol = full_n;
                in_st = in_row - full_n * per_st;
                wid = nbn % sn;
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
            (void)wid;
        }
    };
    // OUTK == 2 (gate_up + silu * mul, see gemm_w4a8_ring.hip): N stacks [gate | up]; unit j = gate channels 32 j .. as its
    // tile row t = 0 and up channels N/2 + 32 j .. as t = 1; the workgroup writes 128 channels of the [M, N/2] result
    constexpr bool ACT = OUTK == 2;
    auto trow = [&](int unit, int t) { return ACT ? (t ? N / 64 + unit : unit) : unit * 2 + t; };
    auto chan32 = [&](int unit, int t) { return ACT ? (t ? N / 2 + 32 * unit : 32 * unit) : unit * 64 + 32 * t; };
    int m0, n0;                                       // current tile (of the DMA sources: runs one tile ahead at the end)
    const int KT = K >> 5;
    const int nh = K >> 6;                            // stages (even: K % 128 == 0)

    // ---- activation pair image --------------------------------------------------------------------------------------
    // One DMA instruction fills a 1 KiB piece = 8 token rows x 128 B (two stages).  Inside the piece the 16 half rows of
    // 64 B are placed so that (i) every 16 consecutive lanes fetch two whole 128-byte lines, and (ii) the readers of ONE
    // stage (half h) - lane (li, g) wants 16 B of token row li, k-chunk g - never meet on a bank: half h of row 2a + b
    // sits at 64-byte position 4a + 2(h ^ (a & 1)) + b, so for a fixed h the position mod 4 (= the bank quarter) is a
    // function of li & 3 that takes all four values; the four rows li, li+4, li+8, li+12 that share a quarter hold
    // chunk c at 16-byte position c ^ ((-(li >> 2)) & 3), which is distinct inside each of ds_read_b128's lane groups
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS; measured conflict-free).
    auto aswz = [&](int j) { return (0 - j) & 3; };
    // ---- DMA sources: per-lane 32-bit byte offsets; the k advance goes into the scalar base --------------------------
    // (inline asm rather than __builtin_amdgcn_global_load_lds: the compiler books the builtin as a FLAT access and
    // from then on degrades every counted LDS wait in the loop to lgkmcnt(0))
    u32 a_off[NA2];
    // weights and per-group meta: per-lane offsets are tile-independent, the tile moves the scalar bases
    const uint8_t* w_base;
    const int8_t* m_base;
    u32 w_off, m_off;
    {
        const int unit = wave >> 1, t = wave & 1;                  // this wave copies tile row t of the tile's unit `unit`
        const int e = lane >> 4, kk = ((lane >> 3) & 1) ^ t, cc = lane & 7;
        w_off = ((u32)trow(unit, t) * (u32)KT + kk) * 512u + cc * 64 + e * 16;
        m_off = chan32(lane >> 4, (lane >> 3) & 1) + (lane & 7) * 4;   // dword `lane` of the tile's 64
    }
    // instruction i of a wave copies tile rows (i*8 + wave)*8 .. +7: lane -> row a_r0 + 64 i, byte a_c inside the row's
    // 128 k (the swizzle term (r >> 2) & 3 does not depend on i)
    // (everything derived from the lane id outside the k loop is recomputed from a "laundered" copy: kept live across
    //  the loop - 128 accumulators + operand buffers, 256 registers - these constants were what the compiler spilled)
    auto fresh = [](int v) {
        asm volatile("" : "+v"(v));
        return v;
    };
    auto setup = [&](int id) {
        int bm, bn;
        tile_coords(id, bm, bn);
        m0 = bm * BM, n0 = bn * BN;
        const int ln = fresh(lane);
        const int a_r0 = wave * 8 + 2 * (ln >> 4) + ((ln >> 2) & 1);
        const u32 a_c = ((((ln >> 3) & 1) ^ ((ln >> 4) & 1)) * 64) + (((ln & 3) ^ aswz((a_r0 >> 2) & 3)) * 16);
#pragma unroll
        for (int i = 0; i < NA2; ++i) {
            int row = m0 + a_r0 + 64 * i;
            row = row < M ? row : M - 1;
            a_off[i] = __umul24((u32)row, (u32)K) + a_c;           // M, K < 2^24 and M * K < 2^32 (checked by the dispatcher)
        }
        // first tile row / channel of the tile: trow(u0 + unit, t) = trow(unit, t) + trow(u0, 0), same for chan32
        w_base = W + (size_t)trow(n0 / 64, 0) * KT * 512;
        m_base = ((wave & 1) ? zeros : scales8) + chan32(n0 / 64, 0);
    };
    const u32 lds0 = (u32)(size_t)(lptr_t)smem;

    auto dma16 = [&](u32 voff, const void* sbase, u32 lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr)
                     : "memory");
    };
    auto dma4 = [&](u32 voff, const void* sbase, u32 lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr)
                     : "memory");
    };
    // Epilogue operands of a tile (PERSIST kernels, round 6): the 256 channel scales (+ scale * zero) and the 256 token scales (+ token
    // sums) are requested by LDS-DMA into 3 KiB above the rings when the PREVIOUS tile's epilogue has read its own (kernel start for
    // the first tile) and read from LDS after the k loop.  As register loads behind the k loop they were a dependent memory round
    // trip per tile with the matrix pipe idle, and the compiler's vmcnt(0) for them also drained the next tile's prefetched stages.
    // Layout: [w scale 256 halfs | w scale*zero 256 halfs | token scale 256 x 4-byte slots | token sum 256 slots].
    constexpr int SC_OFF = NS * (BM * 64 + WSTAGE + 512);
    uint8_t* const s_sc = smem + SC_OFF;
    auto issue_scales = [&](int tm0, int tn0) {
        const u32 sc_lds = lds0 + SC_OFF;
        auto dma4p = [&](const void* src, u32 dst) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(dst) : "memory");
        };
        auto dma2p = [&](const void* src, u32 dst) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_ushort %0, off" ::"v"(src), "s"(dst) : "memory");
        };
        const int ln = fresh(lane);
        if (wave < 4) {                                // waves 0, 1: w scale; 2, 3: w scale * zero - 128 halfs per instruction
            const int lc = 128 * (wave & 1) + 2 * ln;  // local channel: ACT = [128 gate | 128 up]
            const int gc = ACT ? (lc < 128 ? 32 * (tn0 / 64) + lc : N / 2 + 32 * (tn0 / 64) + (lc - 128)) : tn0 + lc;
            if (wave < 2) dma4p(reinterpret_cast<const _Float16*>(wscales) + gc, sc_lds + 256 * wave);
            else if (MODE == 0) dma4p(reinterpret_cast<const _Float16*>(wszs) + gc, sc_lds + 512 + 256 * (wave - 2));
        }
        {                                              // waves 0-3: token scale of 64 tokens each; waves 4-7: token sums
            int m = tm0 + 64 * (wave & 3) + ln;
            m = m < M ? m : M - 1;
            if (wave < 4) dma2p(reinterpret_cast<const _Float16*>(ascales) + m, sc_lds + 1024 + 256 * wave);
            else if (MODE == 0) dma2p(reinterpret_cast<const _Float16*>(assums) + m, sc_lds + 2048 + 256 * (wave - 4));
        }
    };
    // Round 6: the stream of stages runs ACROSS tile boundaries.  While the last NS - 1 stages of a tile compute, stage indices
    // beyond the tile (u >= nh, pair >= nh / 2) are the FIRST stages of the workgroup's next tile: same per-lane offsets (the next
    // tile's token rows are a wave-uniform distance away when neither tile is clipped by M), the next tile's scalar bases.
    // Before, the DMA engine idled from stage nh - 6 to the fill behind the epilogue's barrier - with the k loop within 10 % of a
    // CU's LDS-DMA fill rate that pause is throughput lost: 12.5 us per tile whatever K (65 536 x 4 096 x K: 22.4 / 33.8 / 52.2 /
    // 93.4 us per tile at K = 1024 / 2048 / 4096 / 8192 = 0.62 us per stage + 12.5).
    const uint8_t* w_base_n = nullptr;                // (valid while `stream`)
    const int8_t* m_base_n = nullptr;
    long long a_delta_n = 0;
    auto issue_a = [&](int pr, int pslot, int i) {                 // instruction i of activation pair pr (stages 2pr, 2pr+1)
        const bool nx = pr >= (nh >> 1);                           // wave-uniform
        const int8_t* base = nx ? A + a_delta_n + (size_t)(pr - (nh >> 1)) * 128 : A + (size_t)pr * 128;
        dma16(a_off[i], base, lds0 + pslot * APAIR + (i * 8 + wave) * 1024);
    };
    auto issue_w = [&](int u, int slot, int i) {                   // i = 0: weights of stage u, 1: its per-group meta
        const bool nx = u >= nh;
        const int uu = nx ? u - nh : u;
        if (i == 0) dma16(w_off, (nx ? w_base_n : w_base) + (size_t)uu * 1024, lds0 + (NS / 2) * APAIR + slot * WSTAGE + wave * 1024);
        else dma4(m_off, (nx ? m_base_n : m_base) + (size_t)(uu >> 1) * N, lds0 + (NS / 2) * APAIR + NS * WSTAGE + slot * 512 + (wave & 1) * 256);
    };
    // Issue order of a wave (vmcnt retires in order): W(0), then for q = 0, 1, ...: A(q), W(2q+1), W(2q+2).  The
    // prologue issues W(0..4), A(0), A(1); stage u issues W(u+5) and, when u is even, A(u/2 + 2) before it.  Stage v
    // reads its own and ALREADY stage v+1's operands (rolling prefetch), so before stage v the wave needs everything up
    // to W(v+1) (v even; A(v/2) precedes it) or up to A((v+1)/2) (v odd; W(v+1) precedes it) complete; whatever was
    // issued later may stay in flight.  Steady state: 3 NW + NA2 instructions.
    auto allowed = [&](int v) {
        const int nw = v + 5 < nh ? v + 5 : nh, na = 2 + ((v + 1) >> 1) < (nh >> 1) ? 2 + ((v + 1) >> 1) : (nh >> 1);
        const int q = v >> 1;
        const int need = (v & 1) ? NW + (q + 1) * (NA2 + 2 * NW) + NA2 : 2 * NW + q * (NA2 + 2 * NW) + NA2;
        return NW * nw + NA2 * na - need;          // <= 0 at the end of K: everything has to be there
    };

    // ---- LDS operand readers ----------------------------------------------------------------------------------------
    const int w_rd = wn * 2048 + tsel * 1024 + (((g >> 1) ^ tsel)) * 128 + c * 16 + (g & 1) * 8;   // + e*256
    const int m_rd = wn * 64 + (tsel * 8 + c) * 4;
    const int a_rd0 = (2 * wm * MT + (li >> 3)) * 1024 + (4 * ((li & 7) >> 1) + 2 * ((li >> 1) & 1) + (li & 1)) * 64 +
                      ((g ^ aswz(li >> 2)) * 16);                                                  // half 0; + mt*2048
    auto read_b = [&](int pslot, int half, int mt) -> v4i {
        return *reinterpret_cast<const v4i*>(a_ring + pslot * APAIR + (a_rd0 ^ (half * 128)) + mt * 2048);
    };
    struct Raw {
        v2u r[4];
        u32 sdw, zdw;
    };
    // Per-group only: four separate address registers.  With one base the compiler merges pairs of these 8-byte reads into
    // ds_read2_b64, which is serviced in 4 x 16-lane groups on 32 banks (MI355X_MICROARCH.md, LDS): half the rate of
    // ds_read_b64 and a two-way conflict on this image (laid out for ds_read_b64's 2 x 32 lanes on 64 banks).  In-run
    // A/B: per-group +2-3 % with the separate reads (the VALU-bound loop hides their issue slots), per-channel -3.5 %
    // (two more LDS instructions to issue per stage cost more than the LDS cycles saved) - so per-channel keeps the merge.
    int w_rd_e[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        w_rd_e[e] = w_rd + e * 256;
        if (MODE == 1) asm volatile("" : "+v"(w_rd_e[e]));
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
    auto build = [&](const Raw& q, int cl) -> v4i {
        u32 s = 0, zb = 0;
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
    v4i a0[4], a1[4], bq[PD];

    // ---- pipeline fill: weights of stages 0..NS-2 and two activation pairs in flight --------------------------------
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

    // One stage = 4*MT MFMAs of this wave.  `ac` holds the unpacked weight operands of stage u, `an` receives those of
    // stage u+1; bq is the rolling activation-operand buffer (tile t lives in bq[t % PD], PD divides MT).  PAR = u & 1.
    auto stage = [&](auto par_c, auto pref_static, bool pref_a, bool pref_w, int u, int slot, v4i(&ac)[4], v4i(&an)[4]) {
        constexpr int PAR = decltype(par_c)::value;
        const int slot_n = slot + 1 == NS ? 0 : slot + 1;
        const int slot_d = slot == 0 ? NS - 1 : slot - 1;          // (u + NS - 1) % NS
        const int ps = slot >> 1, ps_n = slot_n >> 1;              // activation pair slots of stage u / u+1
        const int ps_d = ps == 0 ? NS / 2 - 1 : ps - 1;            // (u/2 + 2) % (NS/2)
        Raw qn;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const v4i b_use = bq[mt % PD];
            if (!(DBG & 4)) {
                if (mt + PD < MT) bq[mt % PD] = read_b(ps, PAR, mt + PD);
                else bq[mt % PD] = read_b(ps_n, PAR ^ 1, mt + PD - MT);
            }
            if (mt == 0) qn = read_w(slot_n);
            if (!(DBG & 2)) {
                constexpr bool st = decltype(pref_static)::value;
                if (PAR == 0 && mt < NA2) {
                    if (st || pref_a) issue_a((u >> 1) + 2, ps_d, mt);
                } else if (mt - (PAR == 0 ? NA2 : 0) < NW) {
                    if (st || pref_w) issue_w(u + NS - 1, slot_d, mt - (PAR == 0 ? NA2 : 0));
                }
            }
            if (MT == 8 && mt >= 2 && mt < 6) an[mt - 2] = build(qn, mt - 2);
            if (MT == 4 && mt >= 2) {
                an[2 * (mt - 2)] = build(qn, 2 * (mt - 2));
                an[2 * (mt - 2) + 1] = build(qn, 2 * (mt - 2) + 1);
            }
#pragma unroll
            for (int cl = 0; cl < 4; ++cl)
                if (!(DBG & 1)) acc[mt][cl] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ac[cl], b_use, acc[mt][cl], 0, 0, 0);
                else acc[mt][cl][0] += ac[cl][0] ^ b_use[cl];
            // measured: pinning the read / DMA / unpack order between the MFMA groups gains 4-8 % per-channel, but with
            // the VALU-heavy per-group dequant the compiler's own interleaving is 5-6 % faster
            if (MODE == 0) QS_PIN();
        }
    };
    // barrier(u): stage u+1 has landed (this wave's pieces by the counted wait; the barrier extends that to every
    // wave's) while later stages stay in flight, and every wave is done reading stage u-1, whose slot is refilled next
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    int tile = blockIdx.x;
    setup(tile);
    if (PERSIST) issue_scales(m0, n0);
    if (!(DBG & 2)) issue_fill();
    bool first = true;
    bool streamed = false;            // this tile's first NS - 1 stages were requested during the previous tile's last stages
    int slot = 0;                     // ring position of the stage about to run (continues across streamed tile boundaries)
    while (true) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) acc[mt][cl] = (v4i){0, 0, 0, 0};
        // first tile: W(0), A(0), W(1) have landed, the rest of the fill stays in flight.  Later tiles: the fill was issued
        // before the previous tile's epilogue - everything (its stores included) is complete
        // (streamed: the steady-state count - the requests of the last stages are its fill; the epilogue's stores are younger, so
        //  "at most 3 NW + NA2 outstanding" can only over-wait)
        if (first) wait_vm_dyn((DBG & 2) ? 0 : allowed(0));
        else if (streamed) wait_vm<3 * NW + NA2>();
        else wait_vm<0>();
        first = false;
        raw_barrier();
        if (!streamed) slot = 0;
        {
            const Raw q0 = read_w(slot);
#pragma unroll
            for (int t = 0; t < PD; ++t) bq[t] = read_b(slot >> 1, 0, t);
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) a0[cl] = build(q0, cl);
        }
        int u = 0;
        for (; u + NS < nh; u += 2) {                  // steady state: both stages of the pair prefetch, no branches
            if (u) wait_vm<(DBG & 2) ? 0 : 3 * NW + NA2>();        // (u = 0: the fill's wait and barrier)
            if (u && !(DBG & 8)) raw_barrier();
            stage(c0{}, std::true_type{}, true, true, u, slot, a0, a1);
            slot = slot + 1 == NS ? 0 : slot + 1;
            wait_vm<(DBG & 2) ? 0 : 3 * NW + NA2>();
            if (!(DBG & 8)) raw_barrier();
            stage(c1{}, std::true_type{}, true, true, u + 1, slot, a1, a0);
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
        // the next tile of this workgroup, if its first stages can ride on this tile's last ones: both tiles inside M (their
        // token rows then differ by a wave-uniform distance) and K long enough for the fill pattern W(0..4), A(0), A(1)
        const int next = tile + (int)gridDim.x;
        bool stream = false;
        if (PERSIST && !(DBG & 2) && next < ntiles && nh >= NS + 2 && m0 + BM <= M) {
            int bmn, bnn;
            tile_coords(next, bmn, bnn);
            if (bmn * BM + BM <= M) {
                stream = true;
                a_delta_n = (long long)(bmn * BM - m0) * K;
                w_base_n = W + (size_t)trow(bnn * BN / 64, 0) * KT * 512;
                m_base_n = ((wave & 1) ? zeros : scales8) + chan32(bnn * BN / 64, 0);
            }
        }
        if (stream) {
            for (; u < nh; u += 2) {                   // the last stages: steady-state issue pattern, indices beyond the tile = the next one's
                wait_vm<3 * NW + NA2>();
                if (!(DBG & 8)) raw_barrier();
                stage(c0{}, std::true_type{}, true, true, u, slot, a0, a1);
                slot = slot + 1 == NS ? 0 : slot + 1;
                wait_vm<3 * NW + NA2>();
                if (!(DBG & 8)) raw_barrier();
                stage(c1{}, std::true_type{}, true, true, u + 1, slot, a1, a0);
                slot = slot + 1 == NS ? 0 : slot + 1;
            }
        }
        for (; u < nh; u += 2) {                       // drain
            if (u) {
                wait_vm_dyn((DBG & 2) ? 0 : allowed(u));
                raw_barrier();
            }
            stage(c0{}, std::false_type{}, u + 4 < nh, u + NS - 1 < nh, u, slot, a0, a1);
            slot = slot + 1 == NS ? 0 : slot + 1;
            wait_vm_dyn((DBG & 2) ? 0 : allowed(u + 1));
            raw_barrier();
            stage(c1{}, std::false_type{}, false, u + NS < nh, u + 1, slot, a1, a0);
            slot = slot + 1 == NS ? 0 : slot + 1;
        }

        // ---- fused epilogue -------------------------------------------------------------------------------------
        const int em0 = m0, en0 = n0;                  // (m0 / n0 move on to the next tile below)
        const int lane_e = fresh(lane);
        const int li = lane_e & 15, g = lane_e >> 4;   // (shadow the loop's copies, see `fresh`)
        const int ncol0 = chan32(en0 / 64 + wn, g >> 1) + 4 * (g & 1);
        const int mrow0 = em0 + wm * (16 * MT) + li;
        if (OUTK == 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mrow0 + 16 * mt;
                if (m >= M) continue;
#pragma unroll
                for (int cl = 0; cl < 4; ++cl)
                    *reinterpret_cast<v4i*>(reinterpret_cast<int*>(out) + (size_t)m * N + ncol0 + 8 * cl) = acc[mt][cl];
            }
            if (next >= ntiles) break;
            raw_barrier();                             // every wave left the k loop: the rings may be refilled
            tile = next;
            setup(tile);
            if (!(DBG & 2)) issue_fill();
            continue;
        }
        // all scale loads first (one latency), pinned as complete BEFORE the next tile's fill is issued: a wait the
        // compiler places after the fill would also wait for the fill (vmcnt retires in order)
        h4 ws4[4], wz4[4];
        _Float16 sa_h[MT], ss_h[MT];
        if (PERSIST) {                                 // staged by LDS-DMA one tile ago (issue_scales); every wave's DMA of it is long complete
            const int lcol = ACT ? ((g >> 1) * 128 + 32 * wn + 4 * (g & 1)) : (64 * wn + 32 * (g >> 1) + 4 * (g & 1));
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
                ws4[cl] = *reinterpret_cast<const h4*>(s_sc + 2 * (lcol + 8 * cl));
                if (MODE == 0) wz4[cl] = *reinterpret_cast<const h4*>(s_sc + 512 + 2 * (lcol + 8 * cl));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int tl = wm * (16 * MT) + 16 * mt + li;
                sa_h[mt] = *reinterpret_cast<const _Float16*>(s_sc + 1024 + 4 * tl);
                if (MODE == 0) ss_h[mt] = *reinterpret_cast<const _Float16*>(s_sc + 2048 + 4 * tl);
            }
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {           // read BEFORE the barrier behind which the next tile's operands are requested
                asm volatile("" : "+v"(ws4[cl]));
                if (MODE == 0) asm volatile("" : "+v"(wz4[cl]));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                asm volatile("" : "+v"(sa_h[mt]));
                if (MODE == 0) asm volatile("" : "+v"(ss_h[mt]));
            }
        } else {
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
                ws4[cl] = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(wscales) + ncol0 + 8 * cl);
                if (MODE == 0) wz4[cl] = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(wszs) + ncol0 + 8 * cl);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int m = mrow0 + 16 * mt;
                m = m < M ? m : M - 1;
                sa_h[mt] = reinterpret_cast<const _Float16*>(ascales)[m];
                if (MODE == 0) ss_h[mt] = reinterpret_cast<const _Float16*>(assums)[m];
            }
        }
        raw_barrier();                                 // the rings are dead: every wave left the k loop
        // the pair slot of this tile's LAST stages: dead now, refilled only by the next tile's first stage (behind the barrier at
        // the top of the tile) - the staging rows below live there (not streamed: the ring restarts, that slot is pair slot 2)
        const int ps_last = stream ? ((slot == 0 ? NS - 1 : slot - 1) >> 1) : 2;
        if (PERSIST && next < ntiles) {                // next tile: its fill is already in flight (streamed) or issued here
            setup(next);
            issue_scales(m0, n0);
            if (!stream) issue_fill();
        }
        streamed = stream;
        // The fp16 tile of this wave goes through LDS, 16 tokens x 64 channels at a time, so that every store
        // instruction writes whole 128-byte rows (the accumulator layout would scatter 8-byte pieces over 32 lines per
        // instruction).  The staging rows live in activation pair slot 2, which the fill does not touch; they are
        // written and read by the same wave: an LDS wait, no barrier.
        constexpr int RS = 144;                        // staged row stride (bytes): 128 + 16 keeps 16-byte alignment
        uint8_t* const st = a_ring + (PERSIST ? ps_last * APAIR : 0) + wave * (16 * RS);
        _Float16* const orow = reinterpret_cast<_Float16*>(out) + en0 + wn * 64 + (lane_e & 7) * 8;
        // The per-channel convention (qs_set_gemm_epilogue) is a wave-uniform BRANCH around two copies of the loop, not a per-element
        // select (round 6): with the run-time flag inside epi_per_chn the compiler evaluated both forms of every output and picked one -
        // 64 v_pk_fma_f32 + 128 v_cndmask on top of 450 VALU per tile and wave, in the one part of the kernel the matrix pipe idles in.
        auto store_tile = [&](auto fma_c) {
            constexpr int FMA = decltype(fma_c)::value;
    #pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float sa = (float)sa_h[mt];
                const float ss = MODE == 0 ? (float)ss_h[mt] : 0.f;
    #pragma unroll
                for (int cl = 0; cl < 4; ++cl) {
                    const v4i s = acc[mt][cl];
                    h4 o;
                    if (MODE == 0) {
    #pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (_Float16)epi_per_chn(s[r], (float)ws4[cl][r], sa, (float)wz4[cl][r], ss, FMA);
                    } else {
    #pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (_Float16)epi_per_group(s[r], (float)ws4[cl][r], sa);
                    }
                    if (ACT) {     // lanes 0-31: gate, lanes 32-63: up of the same (token, channel) -> silu_and_mul's arithmetic
                        const v2u ob = __builtin_bit_cast(v2u, o);
                        // one swap hands every lane the pair it finishes: r[0] = (x of lanes 0-31 | y of lanes 0-31) = gate elements
                        // 0, 1 for the lower half, 2, 3 for the upper; r[1] = (x | y of lanes 32-63) = the matching up elements
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
                }
                if (ACT) {         // 16 tokens x 32 channels of this wave: 64-byte row pieces (four waves complete a 256-byte row)
                    const int r = lane_e >> 2;
                    const int m = em0 + wm * (16 * MT) + 16 * mt + r;
                    const v4u v = *reinterpret_cast<const v4u*>(st + r * RS + (lane_e & 3) * 16);
                    if (m < M)
                        *reinterpret_cast<v4u*>(reinterpret_cast<_Float16*>(out) + (size_t)m * (N / 2) + (en0 / 64 + wn) * 32 +
                                                (lane_e & 3) * 8) = v;
                    continue;
                }
    #pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = i * 8 + (lane_e >> 3);
                    const int m = em0 + wm * (16 * MT) + 16 * mt + r;
                    const v4u v = *reinterpret_cast<const v4u*>(st + r * RS + (lane_e & 7) * 16);
                    if (m < M) *reinterpret_cast<v4u*>(orow + (size_t)m * N) = v;
                }
            }
        };
        if (MODE == 0 && epi_fma) store_tile(std::integral_constant<int, 1>());
        else store_tile(std::integral_constant<int, 0>());
        if (!PERSIST || next >= ntiles) break;
        tile = next;
    }
    if (clk && threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - ck0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

template <int MT, int MODE, int OUTK, int DBG = 0>
int launch_tiled(const int8_t* A, const uint8_t* W, const int8_t* zeros, const int8_t* scales8, const void* wscales,
                 const void* ascales, const void* wszs, const void* assums, void* out, int M, int N, int K,
                 hipStream_t stream) {
    auto kern = w4a8_gemm_tiled<MT, MODE, OUTK, DBG>;
    constexpr int BM = 32 * MT;
    const size_t smem = (size_t)NS * (BM * 64 + WSTAGE + 512) + 3072;   // rings + the staged epilogue operands (the epilogue's 18 KiB of staging rows alias the rings)
    static bool configured_dev[QS_MAX_DEVICES] = {};   // the attribute belongs to the (kernel, device) pair
    bool& configured = configured_dev[qs_device_slot()];
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            qs_set_error("w4a8 gemm (tiled): cannot reserve %zu bytes of LDS: %s", smem, hipGetErrorString(e));
            return (int)e;
        }
        configured = true;
    }
    const int nbm = (M + BM - 1) / BM;
    const int ntiles = nbm * (N / BN);
    constexpr bool persist = MT == 8 && OUTK != 1 && !(DBG & 2);
    const int cus = qs_num_cus();
    const int pmode = g_tiled_order / 10;             // 0: one workgroup per CU, 1: one per tile, 2: three (tests)
    dim3 grid(persist && pmode != 1 && ntiles > (pmode == 2 ? 3 : cus) ? (pmode == 2 ? 3 : cus) : ntiles);
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, stream, A, W, zeros, scales8,
                       reinterpret_cast<const __half*>(wscales), reinterpret_cast<const __half*>(ascales),
                       reinterpret_cast<const __half*>(wszs), reinterpret_cast<const __half*>(assums), out, M, N, K,
                       nbm, g_tiled_order % 10, g_epi_fma, grid.x <= (unsigned)g_gemm_clk_cap ? g_gemm_clk : nullptr);
    return qs_launch_status("w4a8 gemm (tiled)");
}

}  // namespace

// Entry used by the dispatcher in gemm_w4a8.hip.  Preconditions (checked there): N % 256 == 0, K % 128 == 0, K >= 256.
int qs_launch_gemm_tiled(int mode, int outk, const int8_t* A, const uint8_t* W, const int8_t* zeros,
                         const int8_t* scales8, const void* wscales, const void* ascales, const void* wszs,
                         const void* assums, void* out, int M, int N, int K, int mtile, hipStream_t stream) {
    if (g_qs_plan.active) {
        g_qs_plan.family = 4;
        g_qs_plan.p[0] = mtile == 0 ? (M > 128 ? 8 : 4) : mtile, g_qs_plan.p[1] = g_qs_plan.p[2] = g_qs_plan.p[3] = 0;
        return QS_OK;
    }
#define QS_T(MTV, MODEV, OUTV) \
    return launch_tiled<MTV, MODEV, OUTV>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, stream)
    const bool big = mtile == 0 ? M > 128 : mtile == 8;
#ifdef QS_TIMING   // timing experiments (results are wrong by design): not in the shipped library
    if (mode == 0 && outk == 0 && big && g_tiled_dbg) {
#define QS_D(D) case D: return launch_tiled<8, 0, 0, D>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, stream)
        switch (g_tiled_dbg) {
            QS_D(1); QS_D(2); QS_D(3); QS_D(4); QS_D(6); QS_D(8); QS_D(10); QS_D(14);
        default: break;
        }
#undef QS_D
    }
#endif
    if (mode == 0 && outk == 2) { if (big) QS_T(8, 0, 2); QS_T(4, 0, 2); }
    if (mode == 1 && outk == 2) { if (big) QS_T(8, 1, 2); QS_T(4, 1, 2); }
    if (mode == 0 && outk == 0) { if (big) QS_T(8, 0, 0); QS_T(4, 0, 0); }
    if (mode == 0 && outk == 1) { if (big) QS_T(8, 0, 1); QS_T(4, 0, 1); }
    if (mode == 1 && outk == 0) { if (big) QS_T(8, 1, 0); QS_T(4, 1, 0); }
    if (big) QS_T(8, 1, 1);
    QS_T(4, 1, 1);
#undef QS_T
}
