// gemm_w4a8_tiled.hip -- compute-bound W4A8 GEMM (prefill shapes, BASELINE config 1 = 4096^3): INT8 MFMA, both operands
// staged through LDS by LDS-DMA.
//
// Same arithmetic and operand mapping as the decode kernels (gemm_w4a8.hip header; reference kernels
// kernels/csrc/qgemm/w4a8_per_chn/gemm_cuda.cu:303-594 with its 128x128x64 tile for M > 256, and
// w4a8_per_group/gemm_cuda.cu:328-628).  Tile geometry chosen for CDNA4, not translated from the reference:
//   * workgroup = 8 wave64 (512 threads) computes 256 tokens x 256 channels; wave (wm, wn) owns 128 tokens x one
//     64-channel unit: 8 m-tiles x 4 row classes of v_mfma_i32_16x16x64_i8 accumulators (128 VGPRs), 64 MFMAs per
//     128-wide k-step against 20 ds_read_b128.
//   * a k-step moves 32 KiB of activations + 16 KiB of packed weights (+512 B of per-group scales) into a 3-deep LDS
//     ring with global_load_lds (no staging registers); one raw s_barrier per k-step, counted s_waitcnt vmcnt so that two
//     k-steps stay in flight across the barrier.  Per CU this is ~24 B/clk of fill at 100 % MFMA rate - right at what
//     one CU's memory path sustains, which is why the tile is not smaller.
//   * activation image XOR-swizzled via the DMA source address (conflict-free 16-row operand reads); weight image is
//     the checkpoint's own tile order (the 64-byte c-rows the operand unpacking wants).
//   * per-group: level-2 dequant in registers exactly as in the decode kernels (bit-faithful to the reference).
#include "common.h"

int g_tiled_dbg = 0;
namespace {

constexpr int NS = 3;                      // LDS ring depth (k-steps)
constexpr int BN = 256;                    // channels per workgroup (4 units)
constexpr int WSTEP = BN * 64;             // packed weight bytes per k-step = 16 KiB

__device__ __forceinline__ u32 vadd4(u32 a, u32 b) {
    return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
}
template <int MODE>
__device__ __forceinline__ u32 unpack_lo(u32 raw, u32 s, u32 zb) {
    u32 u = raw & 0x0F0F0F0Fu;
    if (MODE == 1) u = vadd4(u * s, zb);
    return u;
}
template <int MODE>
__device__ __forceinline__ u32 unpack_hi(u32 raw, u32 s, u32 zb) {
    u32 u = (raw >> 4) & 0x0F0F0F0Fu;
    if (MODE == 1) u = vadd4(u * s, zb);
    return u;
}
__device__ __forceinline__ float epi_per_chn(int acc, float ws, float sa, float wz, float ss) {
#pragma clang fp contract(off)
    float t = (float)acc * ws;
    t = t * sa;
    const float u = wz * ss;
    return t - u;
}
__device__ __forceinline__ float epi_per_group(int acc, float ws, float sa) {
#pragma clang fp contract(off)
    const float sc = ws * sa;
    return (float)acc * sc;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void raw_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// MT = m-tiles per wave (8 -> 256-token workgroup tile, 4 -> 128)
template <int MT, int MODE, int OUTK, int DBG = 0>
__global__ __launch_bounds__(512, 1) void w4a8_gemm_tiled(const int8_t* __restrict__ A, const uint8_t* __restrict__ W,
                                                          const int8_t* __restrict__ zeros,
                                                          const int8_t* __restrict__ scales8,
                                                          const __half* __restrict__ wscales,
                                                          const __half* __restrict__ ascales,
                                                          const __half* __restrict__ wszs,
                                                          const __half* __restrict__ assums, void* __restrict__ out,
                                                          int M, int N, int K, int nbm) {
    constexpr int dbg = DBG;
    constexpr int BM = 32 * MT;                       // tokens per workgroup
    constexpr int ASTEP = BM * 128;                   // activation bytes per k-step
    constexpr int NA = ASTEP / 8192;                  // 8 KiB all-thread DMA instructions for the activation tile
    constexpr int NDMA = NA + 2 + (MODE == 1 ? 1 : 0);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const a_ring = smem;                     // [NS][ASTEP]
    uint8_t* const w_ring = smem + NS * ASTEP;        // [NS][WSTEP]
    uint8_t* const m_ring = w_ring + NS * WSTEP;      // [NS][512]: 256 scales | 256 zeros (storage order)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, g = lane >> 4;
    const int tsel = li >> 3, c = li & 7;
    // tile coordinates: consecutive workgroups walk M first inside a band of channels (weights of the band stay in L2)
    const int bm = blockIdx.x % nbm, bn = blockIdx.x / nbm;
    const int m0 = bm * BM, n0 = bn * BN;
    const int m0l = (dbg & 4) ? 0 : m0, n0l = (dbg & 4) ? 0 : n0;
    const int KT = K >> 5;
    const int nsteps = K >> 7;

    // ---- DMA sources --------------------------------------------------------------------------------------------
    const int8_t* a_src[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = (i * 8 + wave) * 8 + (lane >> 3);          // row of the tile this lane copies in instruction i
        int row = m0l + r;
        row = row < M ? row : M - 1;
        a_src[i] = A + (size_t)row * K + (((lane & 7) ^ (r & 7)) * 16);
    }
    const uint8_t* w_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = j * 8 + wave;                               // 1 KiB piece: unit q>>2, e = q&3
        const int unit = q >> 2, e = q & 3;
        w_src[j] = W + ((size_t)(n0l / 32 + unit * 2 + (e >> 1)) * KT + 2 * (e & 1)) * 512 + lane * 16;
    }
    const int8_t* m_src = ((wave & 1) ? zeros : scales8) + n0 + lane * 4;

    // one DMA instruction of k-step ks (pieces 0..NDMA-1); spread over the MFMA stream of the previous step so the
    // memory system sees a steady trickle instead of a burst after every barrier
    auto issue_piece = [&](int ks, int p) {
        const int slot = ks % NS;
        if (p < NA)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[p] + (size_t)ks * 128),
                                             (lptr_t)(a_ring + slot * ASTEP + (p * 8 + wave) * 1024), 16, 0, 0);
        else if (p < NA + 2)
            __builtin_amdgcn_global_load_lds((gptr_t)(w_src[p - NA] + (size_t)ks * 2048),
                                             (lptr_t)(w_ring + slot * WSTEP + ((p - NA) * 8 + wave) * 1024), 16, 0, 0);
        else if (MODE == 1)
            __builtin_amdgcn_global_load_lds((gptr_t)(m_src + (size_t)ks * N),
                                             (lptr_t)(m_ring + slot * 512 + (wave & 1) * 256), 4, 0, 0);
    };
    auto issue = [&](int ks) {
#pragma unroll
        for (int p = 0; p < NDMA; ++p) issue_piece(ks, p);
    };

    v4i acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) acc[mt][cl] = (v4i){0, 0, 0, 0};

#pragma unroll
    for (int j = 0; j < NS - 1; ++j)
        if (j < nsteps) issue(j);

    for (int ks = 0; ks < nsteps; ++ks) {
        if (ks + 1 < nsteps) wait_vm<(NS - 2) * NDMA>();
        else wait_vm<0>();
        raw_barrier();                                   // every wave's pieces of step ks landed; step ks-1 fully consumed
        const bool pref = ks + NS - 1 < nsteps && !(dbg & 2);
        const int slot = ks % NS;
        const uint8_t* wb = w_ring + slot * WSTEP + wn * 4096 + (tsel * 4 + g) * 512 + c * 64;
        v4u ch[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ch[e] = *reinterpret_cast<const v4u*>(wb + e * 16);
        u32 sdw = 0, zdw = 0;
        if (MODE == 1) {
            sdw = *reinterpret_cast<const u32*>(m_ring + slot * 512 + wn * 64 + (tsel * 8 + c) * 4);
            zdw = *reinterpret_cast<const u32*>(m_ring + slot * 512 + 256 + wn * 64 + (tsel * 8 + c) * 4);
        }
        const uint8_t* ab = a_ring + slot * ASTEP + wm * (16 * MT * 128);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v4i b[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r = 16 * mt + li;               // (r & 7) == (li & 7): the swizzle key survives the wm offset
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
                    if (!(dbg & 1)) acc[mt][cl] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[mt], acc[mt][cl], 0, 0, 0);
                    else acc[mt][cl][0] += a[0] + b[mt][0];
                if (pref && h * 4 + cl < NDMA) issue_piece(ks + NS - 1, h * 4 + cl);
            }
        }
    }

    // ---- fused epilogue -----------------------------------------------------------------------------------------
    // all scale loads first (one latency), then a store-only tail the memory pipe can stream
    const int ncol0 = n0 + wn * 64 + 32 * (g >> 1) + 4 * (g & 1);
    const int mrow0 = m0 + wm * (16 * MT) + li;
    h4 ws4[4], wz4[4];
    _Float16 sa_h[MT], ss_h[MT];
    if (OUTK == 0) {
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
    if (OUTK == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mrow0 + 16 * mt;
            if (m >= M) continue;
#pragma unroll
            for (int cl = 0; cl < 4; ++cl)
                *reinterpret_cast<v4i*>(reinterpret_cast<int*>(out) + (size_t)m * N + ncol0 + 8 * cl) = acc[mt][cl];
        }
        return;
    }
    // fp16 tile of this wave (16*MT tokens x 64 channels) goes through LDS so that every store instruction writes
    // whole 128-byte rows (the accumulator layout would scatter 8-byte pieces over 32 lines per instruction)
    constexpr int RS = 144;                            // staged row stride (bytes): 128 + 16 keeps 16-byte alignment
    raw_barrier();                                     // the rings are dead: every wave left the k loop
    uint8_t* const st = smem + wave * (16 * MT * RS);
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
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)epi_per_chn(s[r], (float)ws4[cl][r], sa, (float)wz4[cl][r], ss);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)epi_per_group(s[r], (float)ws4[cl][r], sa);
            }
            *reinterpret_cast<h4*>(st + (16 * mt + li) * RS + (32 * (g >> 1) + 8 * cl + 4 * (g & 1)) * 2) = o;
        }
    }
    _Float16* const orow = reinterpret_cast<_Float16*>(out) + n0 + wn * 64 + (lane & 7) * 8;
#pragma unroll
    for (int i = 0; i < 2 * MT; ++i) {
        const int r = i * 8 + (lane >> 3);
        const int m = m0 + wm * (16 * MT) + r;
        const v4u v = *reinterpret_cast<const v4u*>(st + r * RS + (lane & 7) * 16);
        if (m < M) *reinterpret_cast<v4u*>(orow + (size_t)m * N) = v;
    }
}

template <int MT, int MODE, int OUTK, int DBG = 0>
int launch_tiled(const int8_t* A, const uint8_t* W, const int8_t* zeros, const int8_t* scales8, const void* wscales,
                 const void* ascales, const void* wszs, const void* assums, void* out, int M, int N, int K,
                 hipStream_t stream) {
    auto kern = w4a8_gemm_tiled<MT, MODE, OUTK, DBG>;
    constexpr int BM = 32 * MT;
    const size_t smem = (size_t)NS * (BM * 128 + WSTEP + 512);
    static bool configured = false;
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
    dim3 grid(nbm * (N / BN));
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, stream, A, W, zeros, scales8,
                       reinterpret_cast<const __half*>(wscales), reinterpret_cast<const __half*>(ascales),
                       reinterpret_cast<const __half*>(wszs), reinterpret_cast<const __half*>(assums), out, M, N, K,
                       nbm);
    return qs_launch_status("w4a8 gemm (tiled)");
}

}  // namespace

// Entry used by the dispatcher in gemm_w4a8.hip.  Preconditions (checked there): N % 256 == 0, K % 128 == 0.
int qs_launch_gemm_tiled(int mode, int outk, const int8_t* A, const uint8_t* W, const int8_t* zeros,
                         const int8_t* scales8, const void* wscales, const void* ascales, const void* wszs,
                         const void* assums, void* out, int M, int N, int K, hipStream_t stream) {
#define QS_T(MTV, MODEV, OUTV) \
    return launch_tiled<MTV, MODEV, OUTV>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, stream)
    const bool big = M > 128;
    if (mode == 0 && outk == 0 && big) {
        switch (g_tiled_dbg) {
        case 1: return launch_tiled<8, 0, 0, 1>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, stream);
        case 2: return launch_tiled<8, 0, 0, 2>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, stream);
        case 3: return launch_tiled<8, 0, 0, 3>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, stream);
        case 4: return launch_tiled<8, 0, 0, 4>(A, W, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, stream);
        default: break;
        }
    }
    if (mode == 0 && outk == 0) { if (big) QS_T(8, 0, 0); QS_T(4, 0, 0); }
    if (mode == 0 && outk == 1) { if (big) QS_T(8, 0, 1); QS_T(4, 0, 1); }
    if (mode == 1 && outk == 0) { if (big) QS_T(8, 1, 0); QS_T(4, 1, 0); }
    if (big) QS_T(8, 1, 1);
    QS_T(4, 1, 1);
#undef QS_T
}
