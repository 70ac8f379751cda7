// gemm_w4a8_lds.hip -- W4A8 GEMM for decode shapes with MANY output channels (N/64 >= 256 units, e.g. gate_up_proj).
//
// Same arithmetic, operand mapping and epilogue as w4a8_gemm_splitk (gemm_w4a8.hip; reference kernels
// kernels/csrc/qgemm/w4a8_per_chn/gemm_cuda.cu:303-594, w4a8_per_group/gemm_cuda.cu:328-628).  What changes is how the
// operands reach the matrix cores.  Measured on MI355X (scripts/bench_gemm*.py, microbench_wstream.hip): one CU's
// vector-memory path sustains ~48 GB/s, the weight stream alone needs ~24 GB/s per CU at the HBM roofline, and in the
// split-K kernel every 64-channel workgroup re-reads the whole int8 activation matrix (2 bytes of activations per
// byte of weights).  Here:
//   * a workgroup = 4 wave64 = 2 channel units (128 channels) x 2 K-halves; the two waves of a K-half walk the same
//     k-steps in lock-step and SHARE the activation tile of each step through LDS (activation : weight bytes = 1 : 1);
//   * every byte - activations and weights - travels HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B per lane) into
//     NS-deep rings: no staging registers, NS-1 k-steps (up to ~100 KiB per CU) in flight, completion tracked with
//     counted s_waitcnt vmcnt + one raw s_barrier per k-step;
//   * the activation image is XOR-swizzled through the DMA SOURCE address (LDS destination stays lane-linear) so that
//     the 16-row x 16-byte operand reads are conflict-free;
//   * the two K-halves are summed exactly (int32) through LDS; fused fp32 epilogue as before.
#include "common.h"

extern qs_flag g_tiled_dbg;   // gemm_w4a8_tiled.hip: qs_set_gemm_variant(3100 + bits) timing experiments
namespace {

constexpr int NS = 4;                 // ring depth (k-steps)
constexpr int WBYTES = 4096;          // weight bytes of one unit per k-step
constexpr int MAXM = 64;              // tokens per workgroup


typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() {
    // counted wait for this wave's LDS-DMA queue; the "memory" clobber keeps the compiler from hoisting LDS reads
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void raw_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();          // NOT __syncthreads(): that would drain the DMA queue (vmcnt(0))
    asm volatile("" ::: "memory");
}

template <int MT, int MODE, int OUTK, int DBG = 0>
__global__ __launch_bounds__(256, 1) void w4a8_gemm_pair(const int8_t* __restrict__ A, const uint8_t* __restrict__ W,
                                                         const int8_t* __restrict__ zeros,
                                                         const int8_t* __restrict__ scales8,
                                                         const __half* __restrict__ wscales,
                                                         const __half* __restrict__ ascales,
                                                         const __half* __restrict__ wszs,
                                                         const __half* __restrict__ assums, void* __restrict__ out,
                                                         int M, int N, int K, int epi_fma) {
    constexpr int ATILE = 16 * MT * 128;          // activation bytes per k-step
    constexpr int APART = MT;                     // 1 KiB DMA instructions per wave for its half of the tile
    constexpr int NDMA = APART + 4 + (MODE == 1 ? 1 : 0);   // VMEM instructions per wave per k-step
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const a_ring = smem;                                   // [2 kh][NS][ATILE]
    uint8_t* const w_ring = smem + 2 * NS * ATILE;                  // [4 waves][NS][WBYTES]
    uint8_t* const m_ring = w_ring + 4 * NS * WBYTES;               // [4 waves][NS][256]  (per-group scales | zeros)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = wave & 1, kh = wave >> 1;
    const int li = lane & 15, g = lane >> 4;
    const int tsel = li >> 3, c = li & 7;
    const int T0 = (blockIdx.x * 2 + u) * 2;       // first n32 tile of this wave's unit
    const int m0 = blockIdx.y * (16 * MT);
    const int KT = K >> 5;
    const int nsteps = K >> 7;
    const int ks_begin = (nsteps * kh) / 2, ks_end = (nsteps * (kh + 1)) / 2;
    const int nloc = ks_end - ks_begin;            // both K-halves differ by at most one step
    const int rot = (int)((blockIdx.x * 5u + blockIdx.y * 3u) % (unsigned)nloc);

    uint8_t* const a_my = a_ring + kh * NS * ATILE;
    uint8_t* const w_my = w_ring + wave * NS * WBYTES;
    uint8_t* const m_my = m_ring + wave * NS * 256;

    // ---- DMA sources: per-lane 32-bit byte offsets; the k-step advance goes into the scalar base ----------------------
    // (inline asm rather than __builtin_amdgcn_global_load_lds: the compiler books the builtin as a FLAT access and
    // from then on degrades every counted LDS wait in the loop to lgkmcnt(0))
    // activations: instruction i of this wave covers rows 8*(u*MT + i) .. +8 of the tile; lane -> (row, chunk);
    // LDS slot (row, pos) receives global chunk pos ^ (row & 7)   [swizzle on the source side]
    u32 a_off[APART];
#pragma unroll
    for (int i = 0; i < APART; ++i) {
        const int r = 8 * (u * MT + i) + (lane >> 3);
        int row = m0 + r;
        row = row < M ? row : M - 1;
        a_off[i] = (u32)row * (u32)K + (((lane & 7) ^ (r & 7)) * 16);
    }
    // weights: instruction e covers 1 KiB = n32 tile (T0 + (e >> 1)), k32 tiles 4*ks + 2*(e & 1) + {0, 1}
    u32 w_off[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w_off[e] = ((u32)(T0 + (e >> 1)) * (u32)KT + 2 * (e & 1)) * 512u + lane * 16;
    // per-group: 16 dwords of scales (lanes 0-15) | zeros (16-31); the zeros table is addressed relative to scales8
    const u32 m_off = (u32)(T0 * 32 + (lane & 15) * 4);
    const int8_t* const m_base = (lane & 16) ? zeros : scales8;
    const u32 lds0 = (u32)(size_t)(lptr_t)smem;
    const u32 a_lds = lds0 + kh * NS * ATILE, w_lds = lds0 + 2 * NS * ATILE + wave * NS * WBYTES,
              m_lds = lds0 + 2 * NS * ATILE + 4 * NS * WBYTES + wave * NS * 256;

    auto dma16 = [&](u32 voff, const void* sbase, u32 lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr)
                     : "memory");
    };
    auto issue = [&](int j) {   // DMA everything of local step j into ring slot j % NS
        // workgroups walk their k-range from different starting points (integer accumulation is order-free): without
        // the rotation all ~224 workgroups request the same activation lines from L2 at the same moment
        const int ks = ks_begin + (j + rot) % nloc, slot = j % NS;
#pragma unroll
        for (int i = 0; i < APART; ++i)
            dma16(a_off[i], A + (size_t)ks * 128, a_lds + slot * ATILE + (u * MT + i) * 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e) dma16(w_off[e], W + (size_t)ks * 2048, w_lds + slot * WBYTES + e * 1024);
        if (MODE == 1) {
            // scales / zeros live in different allocations: the (divergent) base stays a per-lane 64-bit address
            const int8_t* src = m_base + (size_t)ks * N + m_off;
            const u32 ml = m_lds + slot * 256;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(ml) : "memory");
        }
    };

    v4i acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) acc[mt][cl] = (v4i){0, 0, 0, 0};

#pragma unroll
    for (int j = 0; j < NS - 1; ++j)
        if (j < nloc) issue(j);

    // both K-halves run the same number of barrier rounds (max of the two local step counts)
    const int rounds = (nsteps + 1) / 2;
    for (int j = 0; j < rounds; ++j) {
        // own DMAs of step j complete?  younger steps in flight: min(NS-2, nloc-1-j)
        const int younger = nloc - 1 - j;
        if (DBG & 2) wait_vm<0>();
        else if (younger >= NS - 2) wait_vm<(NS - 2) * NDMA>();
        else if (younger == 1) wait_vm<NDMA>();
        else wait_vm<0>();
        raw_barrier();                      // partner's half of the tile landed too; everyone is done with step j-1
        if (j + NS - 1 < nloc && !(DBG & 2)) issue(j + NS - 1);
        if (j < nloc) {
            const int slot = j % NS;
            const uint8_t* wb = w_my + slot * WBYTES + (tsel * 4 + g) * 512 + c * 64;
            v4u ch[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) ch[e] = *reinterpret_cast<const v4u*>(wb + e * 16);
            u32 sdw = 0, zdw = 0;
            if (MODE == 1) {
                sdw = *reinterpret_cast<const u32*>(m_my + slot * 256 + (tsel * 8 + c) * 4);
                zdw = *reinterpret_cast<const u32*>(m_my + slot * 256 + 64 + (tsel * 8 + c) * 4);
            }
            const uint8_t* ab = a_my + slot * ATILE;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v4i b[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int r = 16 * mt + li;
                    b[mt] = *reinterpret_cast<const v4i*>(ab + r * 128 + (((2 * g + h) ^ (r & 7)) * 16));
                }
                u32 rx[4], ry[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rx[e] = h ? ch[e].z : ch[e].x;
                    ry[e] = h ? ch[e].w : ch[e].y;
                }
#pragma unroll
                for (int cl = 0; cl < 4; ++cl) {
                    u32 s = 0, zb = 0;
                    if (MODE == 1) {
                        s = (sdw >> (8 * cl)) & 0xFFu;
                        zb = ((zdw >> (8 * cl)) & 0xFFu) * 0x01010101u;
                    }
                    v4i a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const u32 raw = (cl & 1) ? ry[e] : rx[e];
                        a[e] = (int)((cl & 2) ? unpack_hi<MODE>(raw, s, zb) : unpack_lo<MODE>(raw, s, zb));
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        if (!(DBG & 1)) acc[mt][cl] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[mt], acc[mt][cl], 0, 0, 0);
                        else acc[mt][cl][0] += a[0] ^ b[mt][cl];
                }
            }
        }
    }

    // ---- sum the two K-halves through LDS (rings are dead: every wave drained its DMAs and passed the last round) -----
    const int ncol0 = 32 * (T0 + (g >> 1)) + 4 * (g & 1);
    // epilogue operands are requested now so that their latency overlaps the reduction (and nothing but stores is left
    // after it: a load between stores would wait for the stores, vmcnt being one in-order counter)
    h4 ws4[4], wz4[4];
    _Float16 sa_h[MT], ss_h[MT];
    if (OUTK == 0 && kh == 0) {
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            ws4[cl] = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(wscales) + ncol0 + 8 * cl);
            if (MODE == 0) wz4[cl] = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(wszs) + ncol0 + 8 * cl);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int m = m0 + 16 * mt + li;
            m = m < M ? m : M - 1;
            sa_h[mt] = reinterpret_cast<const _Float16*>(ascales)[m];
            if (MODE == 0) ss_h[mt] = reinterpret_cast<const _Float16*>(assums)[m];
        }
    }
    __syncthreads();
    int* red = reinterpret_cast<int*>(smem);       // [2 units][MT*16][64]
    constexpr int NP = MT * 4;
    if (kh == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int cl = 0; cl < 4; ++cl)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((u * NP + mt * 4 + cl) * 4 + r) * 64 + lane] = acc[mt][cl][r];
    }
    __syncthreads();
    if (kh == 1) return;

#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int cl = 0; cl < 4; ++cl)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][cl][r] += red[((u * NP + mt * 4 + cl) * 4 + r) * 64 + lane];
    if (OUTK == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + 16 * mt + li;
            if (m >= M) continue;
#pragma unroll
            for (int cl = 0; cl < 4; ++cl)
                *reinterpret_cast<v4i*>(reinterpret_cast<int*>(out) + (size_t)m * N + ncol0 + 8 * cl) = acc[mt][cl];
        }
        return;
    }
    // fp16 tile of this wave (16*MT tokens x 64 channels) through LDS: every store instruction then writes whole
    // 128-byte rows instead of 8-byte pieces scattered over 32 lines.  The staging area sits behind the reduction slab.
    constexpr int RS = 144;
    uint8_t* const st = smem + 2 * NP * 4 * 64 * 4 + u * (16 * MT * RS);
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
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)epi_per_chn(s[r], (float)ws4[cl][r], sa, (float)wz4[cl][r], ss, epi_fma);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)epi_per_group(s[r], (float)ws4[cl][r], sa);
            }
            *reinterpret_cast<h4*>(st + (16 * mt + li) * RS + (32 * (g >> 1) + 8 * cl + 4 * (g & 1)) * 2) = o;
        }
    }
    _Float16* const orow = reinterpret_cast<_Float16*>(out) + 64 * (T0 / 2) + (lane & 7) * 8;
#pragma unroll
    for (int i = 0; i < 2 * MT; ++i) {
        const int r = i * 8 + (lane >> 3);
        const int m = m0 + r;
        const v4u v = *reinterpret_cast<const v4u*>(st + r * RS + (lane & 7) * 16);
        if (m < M) *reinterpret_cast<v4u*>(orow + (size_t)m * N) = v;
    }
}

template <int MT, int MODE, int OUTK, int DBG = 0>
int launch_pair(const int8_t* A, const uint8_t* W, const int8_t* zeros, const int8_t* scales8, const void* wscales,
                const void* ascales, const void* wszs, const void* assums, void* out, int M, int N, int K,
                hipStream_t stream) {
    auto kern = w4a8_gemm_pair<MT, MODE, OUTK, DBG>;
    const size_t smem = (size_t)2 * NS * (16 * MT * 128) + 4 * NS * WBYTES + 4 * NS * 256;
    static bool configured_dev[QS_MAX_DEVICES] = {};   // the attribute belongs to the (kernel, device) pair
    bool& configured = configured_dev[qs_device_slot()];
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            qs_set_error("w4a8 gemm (lds): cannot reserve %zu bytes of LDS: %s", smem, hipGetErrorString(e));
            return (int)e;
        }
        configured = true;
    }
    dim3 grid(N / 128, (M + 16 * MT - 1) / (16 * MT));
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, A, W, zeros, scales8,
                       reinterpret_cast<const __half*>(wscales), reinterpret_cast<const __half*>(ascales),
                       reinterpret_cast<const __half*>(wszs), reinterpret_cast<const __half*>(assums), out, M, N, K, g_epi_fma);
    return qs_launch_status("w4a8 gemm (lds)");
}

}  // namespace

// Entry used by the dispatcher in gemm_w4a8.hip.  Preconditions (checked there): N % 128 == 0, K % 128 == 0, K >= 256.
int qs_launch_gemm_pair(int mode, int outk, const int8_t* A, const uint8_t* W, const int8_t* zeros,
                        const int8_t* scales8, const void* wscales, const void* ascales, const void* wszs,
                        const void* assums, void* out, int M, int N, int K, hipStream_t stream) {
    if (g_qs_plan.active) {
        g_qs_plan.family = 2;
        g_qs_plan.p[0] = g_qs_plan.p[1] = g_qs_plan.p[2] = g_qs_plan.p[3] = 0;
        return QS_OK;
    }
    const int mtile = M <= 16 ? 1 : M <= 32 ? 2 : M <= 48 ? 3 : 4;
    if (mode == 0 && outk == 0 && mtile == 4 && g_tiled_dbg) {   // timing experiments only (wrong results by design)
        if (g_tiled_dbg == 1) return launch_pair<4, 0, 0, 1>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, stream);
        if (g_tiled_dbg == 2) return launch_pair<4, 0, 0, 2>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, stream);
        if (g_tiled_dbg == 3) return launch_pair<4, 0, 0, 3>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, stream);
    }
#define QS_P(MTV, MODEV, OUTV) \
    return launch_pair<MTV, MODEV, OUTV>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, stream)
#define QS_PM(MODEV, OUTV)        \
    do {                          \
        if (mtile == 1) QS_P(1, MODEV, OUTV); \
        if (mtile == 2) QS_P(2, MODEV, OUTV); \
        if (mtile == 3) QS_P(3, MODEV, OUTV); \
        QS_P(4, MODEV, OUTV);     \
    } while (0)
    if (mode == 0 && outk == 0) QS_PM(0, 0);
    if (mode == 0 && outk == 1) QS_PM(0, 1);
    if (mode == 1 && outk == 0) QS_PM(1, 0);
    QS_PM(1, 1);
#undef QS_PM
#undef QS_P
}
