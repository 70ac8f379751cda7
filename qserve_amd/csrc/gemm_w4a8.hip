// gemm_w4a8.hip -- W4A8 GEMMs for MI355X (gfx950): int8 activations x uint4 weights -> int32 (MFMA) -> fp16.
//
// Replaces (behaviour, not code) the reference kernels
//   kernels/csrc/qgemm/w4a8_per_chn/gemm_cuda.cu:303-594   (per-channel; zero point folded into the epilogue)
//   kernels/csrc/qgemm/w4a8_per_group/gemm_cuda.cu:328-628 (per-group-128; level-2 dequant u4 -> s8 in registers)
// and consumes the reference's packed `qweight` layout as is (w4a8_linear.py:196-226):
//   bytes [N/32][K/32][lane 32][16];  lane = c*4+e;  byte t = d*8+b*4+f;
//   low nibble  = W[32*n32 + 8b + c      ][32*k32 + 16d + 4e + f]
//   high nibble = W[32*n32 + 8b + c + 16 ][ same ]
// i.e. the 64 contiguous bytes at tile*512 + c*64 hold, for the four rows {c, 8+c, 16+c, 24+c} of the tile, all
// 32 k of the tile; dword e of {bytes 0-3 | 4-7 | 8-11 | 12-15} of chunk e = k {4e..4e+3} (+16 for bytes 8-15) of
// rows {c | 8+c} (low nibble) and {16+c | 24+c} (high nibble).
//
// CDNA4 mapping (this is NOT the reference's mma.m16n8k32 lane scheme):
//   v_mfma_i32_16x16x64_i8:  D[i][j] += sum_k A[i][k] B[k][j];  lane l supplies A[i=l&15][16 k of group g=l>>4] and
//   B[same 16 k][j=l&15] and receives D[i=4g+r][j=l&15], r=0..3.
//   * A operand = WEIGHTS.  MFMA row i <-> (tile select tsel=i>>3, c=i&7) of a 64-row "unit" (two n32 tiles);
//     lane (i,g) owns the 64-byte c-row of tile (T0+tsel, k32 = 4*kstep+g).  From those 64 bytes it builds, with
//     AND / shift only, eight operands: 4 row classes (x-lo, y-lo, x-hi, y-hi = rows c, 8+c, 16+c, 24+c) x 2 k-halves
//     (bytes 0-7 / 8-15 of every chunk).  Operand byte p=4e+f <-> k = 32*k32 + 16h + p: 16 CONTIGUOUS k, so
//   * B operand = ACTIVATIONS: lane (m=l&15, g) needs act[m][128*kstep + 32g + 16h .. +16]: plain 16-byte loads,
//     16 rows x 128 contiguous bytes per instruction.
//   * one k-step = 128 k = exactly one quantisation group: per-group scales/zeros are one dword per lane per step.
//   * accumulator (mt, cls)[r] of lane (m,g) is out[m0+16mt+m][32*(T0+(g>>1)) + 8cls + 4(g&1) + r]: four
//     consecutive channels -> one 8-byte fp16x4 store.
//   Integer accumulation is exact, so any consistent k-permutation is legal; the one above needs no cross-lane
//   movement at all.
//
// Kernel `w4a8_gemm_splitk`: one workgroup = NW waves that all own the same 64 output channels and (16*MT) tokens and
// split K between them (k-steps interleaved); partial int32 tiles are reduced through LDS and the fp32 epilogue is
// fused.  This is the decode-shape kernel (M <= 128: weight-streaming, HBM-bound; every weight byte is read once).
#include "common.h"

namespace {


// MODE 0 = per-channel, 1 = per-group(128).  OUTK 0 = fp16 epilogue, 1 = raw int32 accumulators.
//
// Work decomposition: grid = (N/64, ceil(M/(16 MT)), S).  The S blocks of one output tile and the NW waves of each
// block split the K/128 k-steps: block z owns a contiguous range, its waves take the steps of that range
// round-robin.  Each wave keeps NSTAGE k-steps of operands in flight in registers (weights are streamed from HBM
// exactly once; with ~2 us of loaded-HBM latency the bytes in flight per CU decide the bandwidth).
// Reduction: waves -> LDS (int32, exact) ; blocks -> per-tile int32 slabs in a workspace + arrival counter, the last
// arriving block sums the slabs and runs the fused fp32 epilogue (agent-scope release / acquire, placement
// independent; counters are reset by the last arriver so the workspace is reusable without host work).
template <int MT>
struct Stage {
    v4u w[4];
    v4i b[MT][2];
    u32 sdw, zdw;
};

template <int MT, int MODE, int OUTK, int NSTAGE>
__global__ __launch_bounds__(MT <= 2 ? 512 : 256, 2) void w4a8_gemm_splitk(const int8_t* __restrict__ A, const uint8_t* __restrict__ W,
                                                        const int8_t* __restrict__ zeros,
                                                        const int8_t* __restrict__ scales8,
                                                        const __half* __restrict__ wscales,
                                                        const __half* __restrict__ ascales,
                                                        const __half* __restrict__ wszs,
                                                        const __half* __restrict__ assums, void* __restrict__ out,
                                                        int* __restrict__ slabs, unsigned* __restrict__ counters,
                                                        int M, int N, int K, int mblocks, int epi_fma) {
    extern __shared__ __attribute__((aligned(16))) int red[];   // [NW][MT*16][64] ; red[0] doubles as the "last" flag
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = blockDim.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int tsel = li >> 3, c = li & 7;
    // (unit, token block) of this workgroup.  mblocks > 0 selects the XCD-aware 1-D mapping used when M is split over
    // workgroups: the observed dispatch puts workgroup b on XCD b % 8, so the `mblocks` workgroups that stream the SAME
    // 64 channels are made adjacent on ONE XCD (b, b+8, b+16, ...) and the weights are fetched from HBM once and
    // served to the others by that XCD's L2.  Placement affects speed only, never results.
    int unit = blockIdx.x, mblk = blockIdx.y;
    if (mblocks > 0) {
        const int b = blockIdx.x, slot = b >> 3;
        mblk = slot % mblocks;
        unit = (slot / mblocks) * 8 + (b & 7);
    }
    const int T0 = unit * 2;
    const int m0 = mblk * (16 * MT);
    const int KT = K >> 5;
    const int nsteps_all = K >> 7;
    const int S = gridDim.z, z = blockIdx.z;
    const int ks_begin = (int)(((long)nsteps_all * z) / S), ks_end = (int)(((long)nsteps_all * (z + 1)) / S);

    const uint8_t* wrow = W + ((size_t)(T0 + tsel) * KT) * 512 + c * 64 + (size_t)g * 512;
    const int8_t* arow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int row = m0 + 16 * mt + li;
        row = row < M ? row : M - 1;
        arow[mt] = A + (size_t)row * K + 32 * g;
    }
    const int meta_off = (T0 + tsel) * 32 + c * 4;   // per-group scale / zero dword of this lane's 4 row classes

    v4i acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) acc[mt][cl] = (v4i){0, 0, 0, 0};

    auto load_stage = [&](Stage<MT>& st, int ks) {
        const v4u* wp = reinterpret_cast<const v4u*>(wrow + (size_t)ks * 2048);
#pragma unroll
        for (int e = 0; e < 4; ++e) st.w[e] = wp[e];
        // (nontemporal loads measured 1.6x SLOWER for this 64-byte-row pattern: scripts/microbench_wstream.hip P2)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const v4i* ap = reinterpret_cast<const v4i*>(arow[mt] + (size_t)ks * 128);
            st.b[mt][0] = ap[0];
            st.b[mt][1] = ap[1];
        }
        if (MODE == 1) {
            st.sdw = *reinterpret_cast<const u32*>(scales8 + (size_t)ks * N + meta_off);
            st.zdw = *reinterpret_cast<const u32*>(zeros + (size_t)ks * N + meta_off);
        }
    };
    auto compute_stage = [&](const Stage<MT>& st) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32 rx[4], ry[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rx[e] = h ? st.w[e].z : st.w[e].x;
                ry[e] = h ? st.w[e].w : st.w[e].y;
            }
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
                u32 s = 0, zb = 0;
                if (MODE == 1) {
                    s = (st.sdw >> (8 * cl)) & 0xFFu;
                    zb = ((st.zdw >> (8 * cl)) & 0xFFu) * 0x01010101u;
                }
                v4i a;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const u32 raw = (cl & 1) ? ry[e] : rx[e];
                    a[e] = (int)((cl & 2) ? unpack_hi<MODE>(raw, s, zb) : unpack_lo<MODE>(raw, s, zb));
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][cl] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, st.b[mt][h], acc[mt][cl], 0, 0, 0);
            }
        }
    };

    Stage<MT> st[NSTAGE];
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) {
        const int kk = ks_begin + wave + s * NW;
        if (kk < ks_end) load_stage(st[s], kk);
    }
    for (int ks = ks_begin + wave; ks < ks_end; ks += NW * NSTAGE) {
#pragma unroll
        for (int s = 0; s < NSTAGE; ++s) {
            const int kk = ks + s * NW;
            if (kk < ks_end) {
                compute_stage(st[s]);
                const int nk = kk + NSTAGE * NW;
                if (nk < ks_end) load_stage(st[s], nk);
            }
        }
    }

    // ---- cross-wave (split-K) reduction through LDS; wave w finalises pairs p = w, w+NW, ...
    constexpr int NP = MT * 4;
    if (NW > 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int cl = 0; cl < 4; ++cl)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(wave * NP * 4 + (mt * 4 + cl) * 4 + r) * 64 + lane] = acc[mt][cl][r];
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
                const int p = mt * 4 + cl;
                if ((p % NW) != wave) continue;
                v4i s = (v4i){0, 0, 0, 0};
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[r] += red[(w * NP * 4 + p * 4 + r) * 64 + lane];
                acc[mt][cl] = s;
            }
    }

    // ---- cross-block reduction: slabs + arrival ticket, last arriver finalises ---------------------------------
    if (S > 1) {
        const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        v4i* slab = reinterpret_cast<v4i*>(slabs) + (tile * S) * (size_t)(NP * 64);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
                const int p = mt * 4 + cl;
                if (NW > 1 && (p % NW) != wave) continue;
                slab[((size_t)z * NP + p) * 64 + lane] = acc[mt][cl];
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // also orders the LDS reads above before red[0] is reused as a flag
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            red[0] = (t == (unsigned)(S - 1)) ? 1 : 0;
        }
        __syncthreads();
        if (red[0] == 0) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-cleaning
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
                const int p = mt * 4 + cl;
                if (NW > 1 && (p % NW) != wave) continue;
                v4i s = (v4i){0, 0, 0, 0};
                for (int zz = 0; zz < S; ++zz) {
                    const v4i t = slab[((size_t)zz * NP + p) * 64 + lane];
                    s += t;
                }
                acc[mt][cl] = s;
            }
    }

    // ---- fused epilogue -----------------------------------------------------------------------------------------
    const int ncol0 = 32 * (T0 + (g >> 1)) + 4 * (g & 1);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            const int p = mt * 4 + cl;
            if (NW > 1 && (p % NW) != wave) continue;
            const v4i s = acc[mt][cl];
            const int m = m0 + 16 * mt + li;
            const int n = ncol0 + 8 * cl;
            if (m < M) {
                if (OUTK == 1) {
                    *reinterpret_cast<v4i*>(reinterpret_cast<int*>(out) + (size_t)m * N + n) = s;
                } else {
                    h4 o;
                    const float sa = __half2float(ascales[m]);
                    const h4 ws4 = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(wscales) + n);
                    if (MODE == 0) {
                        const float ss = __half2float(assums[m]);
                        const h4 wz4 = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(wszs) + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            o[r] = (_Float16)epi_per_chn(s[r], (float)ws4[r], sa, (float)wz4[r], ss, epi_fma);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (_Float16)epi_per_group(s[r], (float)ws4[r], sa);
                    }
                    *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(out) + (size_t)m * N + n) = o;
                }
            }
        }
    }
}

qs_flag g_variant = QS_GEMM_DEFAULT;   // process-global test / measurement hook (include/qserve_amd.h qs_gemm_variant_code): not thread-safe
}  // namespace
thread_local QsGemmPlan g_qs_plan = {0, 0, {0, 0, 0, 0}};
namespace {

// Split-K workspace (per device): int32 slabs + arrival counters, allocated lazily on first use (never while a
// stream is being captured: a failed allocation simply disables cross-block split-K).
struct Workspace {
    int* slabs = nullptr;
    int* ring_slabs = nullptr;                    // K-sliced ring kernel: sentinel-filled between launches (QS_SLAB_SENTINEL)
    unsigned* counters = nullptr;
    size_t slab_bytes = 0;
    int ncounters = 0;
    bool tried = false;
};
Workspace g_ws[16][QS_MAX_STREAM_SLOTS];   // [device][scratch slot] (common.h)

Workspace* get_workspace(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    Workspace& w = g_ws[dev][qs_scratch_slot(stream)];
    if (!w.tried) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return nullptr;                       // first use inside a capture: run un-split, allocate on a later eager call
        }
        w.tried = true;
        const size_t slab_bytes = 48u << 20;
        const int ncnt = 1 << 16;
        void *a = nullptr, *b = nullptr, *c = nullptr;
        if (hipMalloc(&a, slab_bytes) == hipSuccess && hipMalloc(&b, ncnt * sizeof(unsigned)) == hipSuccess &&
            hipMalloc(&c, slab_bytes) == hipSuccess && hipMemset(c, 0x80, slab_bytes) == hipSuccess &&
            hipMemset(b, 0, ncnt * sizeof(unsigned)) == hipSuccess && hipDeviceSynchronize() == hipSuccess) {
            w.slabs = reinterpret_cast<int*>(a);
            w.ring_slabs = reinterpret_cast<int*>(c);
            w.counters = reinterpret_cast<unsigned*>(b);
            w.slab_bytes = slab_bytes;
            w.ncounters = ncnt;
        } else {
            (void)hipGetLastError();
        }
    }
    return w.slabs ? &w : nullptr;
}
}  // namespace
// the error word of the GEMM hand-offs of a scratch slot: the LAST ticket counter (never used as a ticket: see ring_ws / launch_splitk)
unsigned* qs_gemm_error_word(int slot) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || slot < 0 || slot >= QS_MAX_STREAM_SLOTS) return nullptr;
    Workspace& w = g_ws[dev][slot];
    return w.slabs ? w.counters + (w.ncounters - 1) : nullptr;
}
bool qs_gemm_scratch_prealloc(hipStream_t stream) { return get_workspace(stream) != nullptr; }
int qs_gemm_reset_handoff() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return QS_OK;
    for (int slot = 0; slot < QS_MAX_STREAM_SLOTS; ++slot) {
        Workspace& w = g_ws[dev][slot];
        if (!w.slabs) continue;
        hipError_t e = hipMemset(w.ring_slabs, 0x80, w.slab_bytes);
        if (e == hipSuccess) e = hipMemset(w.counters, 0, (size_t)w.ncounters * sizeof(unsigned));
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            qs_set_error("qs_device_reset (gemm): %s", hipGetErrorString(e));
            return (int)e;
        }
    }
    return QS_OK;
}
namespace {

template <int MT, int MODE, int OUTK, int NSTAGE>
int launch_splitk(const int8_t* A, const uint8_t* W, const int8_t* zeros, const int8_t* scales8, const void* wscales,
                  const void* ascales, const void* wszs, const void* assums, void* out, int M, int N, int K, int NW,
                  int S, bool xcd_map, hipStream_t stream) {
    if (g_qs_plan.active) {
        g_qs_plan.family = 1;
        g_qs_plan.p[0] = MT, g_qs_plan.p[1] = NW, g_qs_plan.p[2] = S, g_qs_plan.p[3] = xcd_map ? 1 : 0;
        return QS_OK;
    }
    auto kern = w4a8_gemm_splitk<MT, MODE, OUTK, NSTAGE>;
    size_t smem = NW > 1 ? (size_t)NW * MT * 16 * 64 * sizeof(int) : 16;
    static size_t configured_dev[QS_MAX_DEVICES] = {};   // per instantiation and device
    size_t& configured = configured_dev[qs_device_slot()];
    if (smem > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            qs_set_error("w4a8 gemm: cannot reserve %zu bytes of LDS: %s", smem, hipGetErrorString(e));
            return (int)e;
        }
        configured = smem;
    }
    dim3 grid(N / 64, (M + 16 * MT - 1) / (16 * MT), 1);
    int mblocks = 0;
    if (xcd_map && grid.y > 1 && grid.x % 8 == 0 && S == 1) {   // 1-D XCD-aware mapping (see kernel)
        mblocks = grid.y;
        grid.x *= grid.y;
        grid.y = 1;
    }
    int* slabs = nullptr;
    unsigned* counters = nullptr;
    if (S > 1) {
        Workspace* ws = get_workspace(stream);
        const size_t tiles = (size_t)grid.x * grid.y;
        const size_t need = tiles * S * (size_t)(MT * 4 * 64) * 16;
        if (ws && need <= ws->slab_bytes && tiles < (size_t)ws->ncounters) {     // (the last counter is the error word)
            slabs = ws->slabs;
            counters = ws->counters;
            grid.z = S;
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, stream, A, W, zeros, scales8,
                       reinterpret_cast<const __half*>(wscales), reinterpret_cast<const __half*>(ascales),
                       reinterpret_cast<const __half*>(wszs), reinterpret_cast<const __half*>(assums), out, slabs,
                       counters, M, N, K, mblocks, g_epi_fma);
    return qs_launch_status("w4a8 gemm");
}

}  // namespace
// many-channel decode kernel (gemm_w4a8_lds.hip)
int qs_launch_gemm_pair(int mode, int outk, const int8_t* A, const uint8_t* W, const int8_t* zeros,
                        const int8_t* scales8, const void* wscales, const void* ascales, const void* wszs,
                        const void* assums, void* out, int M, int N, int K, hipStream_t stream);
// decode ring kernel (gemm_w4a8_ring.hip)
int qs_launch_gemm_ring(int mode, int outk, int mt, int wn, const int8_t* A, const uint8_t* W, const int8_t* zeros,
                        const int8_t* scales8, const void* wscales, const void* ascales, const void* wszs,
                        const void* assums, void* out, int M, int N, int K, int mblocks, int ksplit, int* slabs,
                        unsigned* counters, hipStream_t stream);
// compute-bound tiled kernel (gemm_w4a8_tiled.hip)
int qs_launch_gemm_tiled(int mode, int outk, const int8_t* A, const uint8_t* W, const int8_t* zeros,
                         const int8_t* scales8, const void* wscales, const void* ascales, const void* wszs,
                         const void* assums, void* out, int M, int N, int K, int mtile, hipStream_t stream);
// compute-bound kernel, four-wave tile (gemm_w4a8_wide.hip)
int qs_launch_gemm_wide(int mode, int outk, const int8_t* A, const uint8_t* W, const int8_t* zeros,
                        const int8_t* scales8, const void* wscales, const void* ascales, const void* wszs,
                        const void* assums, void* out, int M, int N, int K, int persist_mode, hipStream_t stream);
extern qs_flag g_tiled_order;   // gemm_w4a8_tiled.hip
namespace {

constexpr int QS_UNFUSED = 1 << 20;   // internal: the chosen kernel has no activation epilogue

template <int MODE, int OUTK>
int dispatch(const int8_t* A, const int8_t* W, const int8_t* zeros, const int8_t* scales8, const void* wscales,
             const void* ascales, const void* wszs, const void* assums, void* out, int M, int N, int K,
             qs_stream_t stream_, bool act = false) {
    // act: `out` is [M, N/2] = silu(gate) * up of the stacked gate_up result (epilogue of the ring / tiled kernels,
    // OUTK = 2 there); QS_UNFUSED when the shape is served by a kernel without that epilogue (the caller runs the two ops)
    const int outk = act ? 2 : OUTK;
    QS_REQUIRE(M >= 0 && N > 0 && K > 0, "w4a8 gemm: bad shape M=%d N=%d K=%d", M, N, K);
    QS_REQUIRE(N % 64 == 0, "w4a8 gemm: N=%d must be a multiple of 64", N);
    QS_REQUIRE(K % 128 == 0, "w4a8 gemm: K=%d must be a multiple of 128", K);
    if (M == 0) return QS_OK;   // empty batch: nothing to do (zero-size tensors carry null pointers)
    QS_REQUIRE(A && W && out, "w4a8 gemm: null pointer");
    if (OUTK == 0) QS_REQUIRE(wscales && ascales, "w4a8 gemm: null scale pointer");
    if (MODE == 0 && OUTK == 0) QS_REQUIRE(wszs && assums, "w4a8 per-channel gemm: null w_szs / a_ssums");
    if (MODE == 1) QS_REQUIRE(zeros && scales8, "w4a8 per-group gemm: null zeros / scales_i8");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const uint8_t* Wu = reinterpret_cast<const uint8_t*>(W);
    const int nsteps = K / 128;
    // Heuristic (measured, scripts/bench_gemm*.py):
    //  * every 64-channel unit is one workgroup whose waves split K (exact int32 reduction in LDS);
    //  * few units (N/64 < 256 = CUs): the token dimension is split over workgroups as well (16 tokens each, XCD-aware
    //    mapping so that the co-streaming workgroups share an L2) - more CUs pull the same weight bytes, no reduction
    //    traffic, 1/4 of the accumulators per wave;
    //  * cross-block split-K (S > 1) stays off: the release/acquire fences cost more than they save at these sizes.
    const int units = N / 64;
    // compute-bound shapes (prefill): LDS-tiled kernel (gemm_w4a8_tiled.hip), 256- or 128-token tiles, taken once the
    // tiles fill the chip; variant 3000 disables it, 3001 / 3002 force the 256- / 128-token tile
    if (N % 256 == 0 && K >= 256 && K < (1 << 24) && (size_t)M * K < (1ull << 32) && (size_t)N * K / 2 < (1ull << 32)) {
        int tmt = 0;
        if (g_variant == QS_GEMM_TILED_256 || g_variant == QS_GEMM_WIDE_256) tmt = 8;
        else if (g_variant == QS_GEMM_TILED_128) tmt = 4;
        else if (g_variant < QS_GEMM_SPLITK_BASE || g_variant > QS_GEMM_WIDE_256) {
            const long nb = N / 256;
            // measured crossovers (scripts/bench_gemm_big.py, N=4096..28672): the tiles must (nearly) fill 256 CUs
            // (M >= 192: a 256-token tile must be mostly real tokens - without this bound every N >= 49 152 took the tiled
            // kernel even at M = 64 and ran at 2 TB/s)
            if (M >= 192 && ((M + 255) / 256) * nb >= 192) tmt = 8;
            else if (M >= 256 && ((M + 127) / 128) * nb >= (MODE == 0 ? 96 : 192)) tmt = 4;
        }
        // 256-token tiles, PER-GROUP: the four-wave kernel (gemm_w4a8_wide.hip; round 5) - one level-2 dequant per weight byte for
        // 256 tokens instead of two: +10 ... 18 % in-run (profiles/round5_wide_ab.txt: 4096^3 70.2 -> 64.0 us, 8192 x 4096 x 14336
        // 436 -> 369 us).  Per-channel the two tiles measure the same within +-3 % (both ~3.2 POPS marginal): the eight-wave one
        // stays.  Variant 3003 forces the four-wave tile for any problem, 3001 the eight-wave one (A/B, tests).
        if (tmt == 8 && (g_variant == QS_GEMM_WIDE_256 || (MODE == 1 && g_variant != QS_GEMM_TILED_256)))
            return qs_launch_gemm_wide(MODE, outk, A, Wu, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K,
                                       g_tiled_order / 10, stream);
        if (tmt)
            return qs_launch_gemm_tiled(MODE, outk, A, Wu, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K,
                                        tmt, stream);
    }
    // decode shapes: LDS-DMA ring kernel with operands read one stage ahead (gemm_w4a8_ring.hip); variant 4000
    // disables it (A/B tests against the two older decode kernels below)
    // K slices need the slab / counter workspace: tiles * ksplit * (mt KiB * 4) bytes, one counter per tile
    auto ring_ws = [&](int mt, int mb, int ks, int** slabs, unsigned** counters) -> bool {
        *slabs = nullptr;
        *counters = nullptr;
        if (ks <= 1) return true;
        if (g_qs_plan.active) return true;      // plan-only: assume the workspace exists
        Workspace* ws = get_workspace(stream);
        const size_t tiles = (size_t)units * mb;
        if (!ws || tiles >= (size_t)ws->ncounters || tiles * ks * mt * 4096 > ws->slab_bytes) return false;
        *slabs = ws->ring_slabs;
        *counters = ws->counters;
        return true;
    };
    if (g_variant >= QS_GEMM_RING_GEOMETRY_BASE && g_variant < QS_GEMM_RING_GEOMETRY_END) {       // tests: forced geometry
        const int v = g_variant - QS_GEMM_RING_GEOMETRY_BASE, ks = v / 100 + 1, mt = (v % 100) / 10, wn = v % 10;
        const int mb = ((M + 15) / 16 + mt - 1) / mt;
        if (act && ks > 1) return QS_UNFUSED;
        QS_REQUIRE((mt == 1 || mt == 2 || mt == 4 || mt == 8) && (wn == 1 || wn == 2 || (wn == 4 && mt == 4)) &&
                       !(mt == 1 && wn == 2) && !(mt == 8 && wn != 2) &&
                       N % (64 * wn) == 0 && (K / 64) % ks == 0 && (K / 64 / ks) % (8 / wn) == 0 && (ks == 1 || K / ks <= 32768),
                   "w4a8 gemm: forced ring geometry mt=%d wn=%d ksplit=%d does not fit M=%d N=%d K=%d", mt, wn, ks, M, N,
                   K);
        int* slabs = nullptr;
        unsigned* counters = nullptr;
        QS_REQUIRE(ring_ws(mt, mb, ks, &slabs, &counters), "w4a8 gemm: no split-K workspace for the forced geometry");
        return qs_launch_gemm_ring(MODE, outk, mt, wn, A, Wu, zeros, scales8, wscales, ascales, wszs, assums, out, M, N,
                                   K, mb, ks, slabs, counters, stream);
    }
    // Geometry choice (measured: scripts/bench_gemm.py for the Llama-3-8B shapes, scripts/bench_gemm_shard.py for the
    // tensor-parallel shard shapes): a workgroup of (16 mt tokens) x (64 wn channels) streams K (16 mt + 32 wn) bytes
    // through its CU, one workgroup per CU at a time, and the per-CU fill rate (~47 GB/s) is what bounds these shapes -
    // so take the geometry with the fewest bytes per CU over all its rounds; ties go to the two-unit workgroups (the
    // activation tile is shared by two waves).  Short K (< 1024) at M <= 64 stays on the split-K kernel (fixed costs).
    if (M <= 1024 && !(K < 1024 && M <= 64) && g_variant != QS_GEMM_RING_OFF && (g_variant < QS_GEMM_SPLITK_BASE || g_variant >= QS_GEMM_RING_OFF) &&
        (size_t)M * K < (1ull << 32) && (size_t)N * K / 2 < (1ull << 32)) {
        const int mt_all = (M + 15) / 16;
        // <8,2> = 128-token workgroups (round 5): PER-GROUP only, un-split, from 65 tokens on - one level-2 dequant of a weight byte
        // serves 128 tokens instead of being repeated per 64-token block.  Measured (scripts/gpu_mt8_ab.sh, weights from HBM, g128):
        // gate_up 28 672 x 4096 at M = 128: 30.6 us against 35.6 for <4,4> x 2 token blocks (M = 96: 29.6 / 35.2); per-channel
        // the same geometry LOSES (28.5 vs 25.4 us: ring depth 3 instead of 5, nothing to share), K-sliced or four-unit forms of it
        // lose everywhere (<8,4>: 60 us), and for N <= 6144 the 64-token geometries fill the chip better (qkv 28.7 vs 15.9 us).
        // Variant 4004 keeps it out (A/B).
        static const int geo[7][2] = {{4, 2}, {2, 2}, {4, 1}, {2, 1}, {1, 1}, {4, 4}, {8, 2}};   // 4 units (2 K-groups): fewer bytes per CU where two-unit
        // workgroups need a second round - M = 128 x N = 28 672: 26.8 vs 32.4 us (per-group 41.0 vs 47.0), M = 64 x 49 152: 48.9 vs 58.4
        // K slices (ksplit 2 / 4, int32 partial tiles meeting in a workspace, the last-dispatched slice finishes): fewer bytes per
        // CU when neither tokens nor channels can be cut further, against the seam's cost; variant 4001
        // keeps ksplit = 1 (A/B)
        // seam cost in bytes of streaming, calibrated on scripts/bench_gemm_shard.py (VARIANTS=4001,-1): slab stores ->
        // ticket -> slab loads (one batch for mt <= 2, one per slice for mt = 4); K-sliced streams are charged 10 % extra
        auto seam = [](int ks, int mt) -> long {   // measured 3-7 us: three dependent system-scope round trips
            if (ks <= 1) return 0;
            return ((ks == 2 ? 150L : 250L) + (mt > 2 ? 40L * (ks - 2) : 0)) * 1024;
        };
        long best = -1;
        int bmt = 0, bwn = 0, bks = 1;
        for (int ks = 1; ks <= (g_variant == QS_GEMM_RING_NO_KSLICES || act ? 1 : 4); ks *= 2)
            for (int i = 0; i < 7; ++i) {
                const int mt = geo[i][0], wn = geo[i][1];
                if (mt == 8 && (MODE != 1 || ks > 1 || mt_all <= 4 || mt_all > 8 || g_variant == QS_GEMM_RING_NO_MT8)) continue;   // (65 .. 128 tokens)
                if (N % (64 * wn) != 0 || (K / 64) % ks != 0 || (K / 64 / ks) % (8 / wn) != 0) continue;
                if (ks > 1 && K / ks > 32768) continue;        // the seam's sentinel must stay out of reach of a partial sum
                const int mb = (mt_all + mt - 1) / mt;
                const long blocks = (long)mb * (N / (64 * wn)) * ks;
                if (ks > 1 && (long)mb * (N / (64 * wn)) > 256) continue;   // K slices are for under-filled grids only
                // per-group: the level-2 dequant is VALU work per weight byte a workgroup streams (measured at M = 128, g128:
                // qkv 16.0 us with (4,1) against 18.2 with the equal-bytes (2,2)) - charged as a quarter of the weight bytes
                const long pg = MODE == 1 && g_variant != QS_GEMM_RING_NO_GROUP_TERM ? 8 * wn : 0;
                const long cost = ((blocks + 255) / 256) * (16 * mt + 32 * wn + pg) * (long)(K / ks) * (ks > 1 ? 11 : 10) / 10 +
                                  seam(ks, mt);
                if (best < 0 || cost < best) best = cost, bmt = mt, bwn = wn, bks = ks;
            }
        // Measured override of the byte model (round 4, scripts/gpu_plan_check.sh, weights from HBM): where the model takes
        // <2,2> x 4 K slices over two token blocks and <2,1> x 2 slices fills the chip with the same 256 workgroups, the latter
        // is 3-7 % faster - a third of the slab traffic (stored, read, restored) on a launch that is HBM-bound, and since round 4
        // the two token blocks share their weight stream in L2 (default cache policy, gemm_w4a8_ring.hip).  Llama-3-8B down_proj
        // (N = 4096, K = 14336) at 33-64 tokens: 14.4-14.7 vs 15.3-15.8 us per-channel, 19.6-19.9 vs 20.6-20.9 g128.  The byte
        // model puts the two 1 KB apart and cannot be tuned to separate them without flipping M = 32 (measured the other way).
        // (Inside the decode step the two are equal, 2.809 vs 2.813 ms: kept for the traffic - one slab per tile instead of three.)
        if (best >= 0 && bmt == 2 && bwn == 2 && bks == 4 && (mt_all + 1) / 2 == 2 && (long)2 * (N / 64) * 2 == 256 &&
            (K / 64) % 2 == 0 && (K / 64 / 2) % 8 == 0 && g_variant != QS_GEMM_RING_NO_DOWN_OVERRIDE)
            bwn = 1, bks = 2;
        // the older register-staged split-K kernel takes any K and cuts the tokens down to 16 per workgroup: same byte
        // model, ~20 % slower at equal bytes (measured) - it wins where K leaves the ring kernel only coarse geometries
        // (Llama-2-7B down_proj: K = 11 008 = 172 stages, two-unit workgroups only)
        if (best >= 0 && !act && units < 256 && units % 8 == 0 && M > 16 && M <= 128) {
            int mto = 1;
            for (int cand = 4; cand >= 1; cand >>= 1)
                if (cand <= mt_all && (long)units * ((mt_all + cand - 1) / cand) >= 192) {
                    mto = cand;
                    break;
                }
            const long blocks_o = (long)units * ((mt_all + mto - 1) / mto);
            const long cost_o = ((blocks_o + 255) / 256) * (16 * mto + 32) * (long)K * 12 / 10;
            if (cost_o < best) best = -1;
        }
        if (best >= 0) {
            int mb = (mt_all + bmt - 1) / bmt;
            int* slabs = nullptr;
            unsigned* counters = nullptr;
            if (!ring_ws(bmt, mb, bks, &slabs, &counters)) {   // no workspace (e.g. first call inside a capture): best un-split
                best = -1;
                bks = 1;
                for (int i = 0; i < 6; ++i) {
                    const int mt = geo[i][0], wn = geo[i][1];
                    if (N % (64 * wn) != 0 || (K / 64) % (8 / wn) != 0) continue;
                    const long blocks = (long)((mt_all + mt - 1) / mt) * (N / (64 * wn));
                    const long cost = ((blocks + 255) / 256) * (16 * mt + 32 * wn);
                    if (best < 0 || cost < best) best = cost, bmt = mt, bwn = wn;
                }
                mb = (mt_all + bmt - 1) / bmt;
            }
            if (best >= 0)
                return qs_launch_gemm_ring(MODE, outk, bmt, bwn, A, Wu, zeros, scales8, wscales, ascales, wszs, assums, out,
                                           M, N, K, mb, bks, slabs, counters, stream);
        }
    }
    if (act) return QS_UNFUSED;
    // many channels: LDS-shared activation tiles + LDS-DMA rings (gemm_w4a8_lds.hip); variant 2000 forces the
    // split-K kernel, 2001 forces the LDS kernel (A/B tests)
    if ((((units >= 256 && M > 16) || M >= 384) && g_variant != QS_GEMM_PAIR_OFF || g_variant == QS_GEMM_PAIR_FORCED) && N % 128 == 0 && K >= 256)
        return qs_launch_gemm_pair(MODE, OUTK, A, Wu, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K,
                                   stream);
    int mtile = M <= 16 ? 1 : M <= 32 ? 2 : M <= 48 ? 3 : 4;
    bool xcd_map = false;
    if (units < 256 && units % 8 == 0 && M > 16) {
        // fewest token blocks that still give >= 192 workgroups (measured: N=6144 -> 2 blocks of 32, N=4096 -> 4 of 16)
        const int mt_all = (M + 15) / 16;
        mtile = 1;
        for (int cand = 4; cand >= 1; cand >>= 1)
            if (cand <= mt_all && (long)units * ((mt_all + cand - 1) / cand) >= 192) {
                mtile = cand;
                break;
            }
        xcd_map = mtile < mt_all;
    }
    int NW = nsteps >= 16 && mtile <= 2 ? 8 : nsteps >= 4 ? 4 : (nsteps >= 2 ? 2 : 1);
    int S = 1;
    if (g_variant >= QS_GEMM_SPLITK_BASE && g_variant < QS_GEMM_PAIR_OFF) {   // A/B: QS_GEMM_SPLITK_BASE + 100*mtile_override + 10*S + NW
        const int v = g_variant - QS_GEMM_SPLITK_BASE;
        NW = v % 10;
        S = (v / 10) % 10;
        const int mo = v / 100;
        if (mo >= 1 && mo <= 4) {
            mtile = mo;
            xcd_map = mo < (M + 15) / 16;
        } else if (mo == 9) {        // 9 = classic mapping, one workgroup per unit
            mtile = M <= 16 ? 1 : M <= 32 ? 2 : M <= 48 ? 3 : 4;
            xcd_map = false;
        }
        if (S < 1) S = 1;
        if (NW < 1) NW = 1;
        if (NW > (mtile <= 2 ? 8 : 4)) NW = mtile <= 2 ? 8 : 4;
        if (NW == 3 || NW == 5 || NW == 6 || NW == 7) NW = 4;
    }
    if (NW > nsteps) NW = 1;
#define QS_GO(MTV) \
    return launch_splitk<MTV, MODE, OUTK, 2>(A, Wu, zeros, scales8, wscales, ascales, wszs, assums, out, M, N, K, NW, S, xcd_map, stream)
    if (mtile == 1) QS_GO(1);
    if (mtile == 2) QS_GO(2);
    if (mtile == 3) QS_GO(3);
    QS_GO(4);
#undef QS_GO
}

}  // namespace

extern qs_flag g_tiled_dbg;   // gemm_w4a8_tiled.hip: timing experiments (3100 + bits)
extern qs_flag g_wide_order;  // gemm_w4a8_wide.hip: the same switch for the four-wave kernel
extern qs_flag g_wide_dbg;    // gemm_w4a8_wide.hip: timing experiments (3400 + bits; QS_TIMING builds only)
extern qs_flag g_ring_flags;  // gemm_w4a8_ring.hip: A/B switches of the decode kernel (5000 + bits), results unchanged
extern qs_flag g_act_off;
extern "C" void qs_set_gemm_variant(int variant) {
    // the sticky families keep their own word (include/qserve_amd.h qs_gemm_variant_code)
    if (variant >= QS_GEMM_TILED_DEBUG_BASE && variant < QS_GEMM_TILE_ORDER_BASE) {
        g_tiled_dbg = variant - QS_GEMM_TILED_DEBUG_BASE;
        return;
    }
    if (variant >= QS_GEMM_TILE_ORDER_BASE && variant < QS_GEMM_ACT_FUSED) {
        g_tiled_order = variant - QS_GEMM_TILE_ORDER_BASE;
        g_wide_order = (variant - QS_GEMM_TILE_ORDER_BASE) % 10;
        return;
    }
    if (variant >= QS_GEMM_WIDE_DEBUG_BASE && variant < QS_GEMM_WIDE_DEBUG_BASE + 100) {
        g_wide_dbg = variant - QS_GEMM_WIDE_DEBUG_BASE;
        return;
    }
    if (variant == QS_GEMM_ACT_FUSED || variant == QS_GEMM_ACT_SPLIT) {
        g_act_off = variant - QS_GEMM_ACT_FUSED;
        return;
    }
    if (variant >= QS_GEMM_RING_FLAGS_BASE && variant < QS_GEMM_RING_FLAGS_END) {
        g_ring_flags = variant - QS_GEMM_RING_FLAGS_BASE;
        return;
    }
    g_variant = variant;
}

// Per-channel epilogue convention (include/qserve_amd.h): 0 = (acc*ws)*sa - wz*ss with every operation rounded separately
// (default), 1 = fmaf(acc*ws, sa, -(wz*ss)).  Process-wide, read at launch time by every W4A8 per-channel GEMM launch and by
// qs_add_residual_rms_norm_general_planes (which finishes such a GEMM).
qs_flag g_epi_fma = 0;
extern "C" int qs_set_gemm_epilogue(int convention) {
    QS_REQUIRE(convention == 0 || convention == 1, "qs_set_gemm_epilogue: convention %d not in {0, 1}", convention);
    g_epi_fma = convention;
    return QS_OK;
}
extern "C" int qs_get_gemm_epilogue(void) { return g_epi_fma; }

unsigned long long* g_gemm_clk = nullptr;
int g_gemm_clk_cap = 0;
extern "C" int qs_debug_gemm_clock_probe(void* buf, int workgroups) {
    QS_REQUIRE((buf == nullptr) == (workgroups == 0) && workgroups >= 0, "qs_debug_gemm_clock_probe: buffer and capacity come together");
    g_gemm_clk = reinterpret_cast<unsigned long long*>(buf);
    g_gemm_clk_cap = workgroups;
    return QS_OK;
}

extern "C" int qs_w4a8_gemm_plan(int per_group, int M, int N, int K, int* plan5) {
    QS_REQUIRE(plan5, "w4a8 gemm plan: null output");
    const int8_t* d8 = reinterpret_cast<const int8_t*>(uintptr_t(256));   // never dereferenced in plan-only mode
    void* dv = reinterpret_cast<void*>(uintptr_t(256));
    g_qs_plan = {1, 0, {0, 0, 0, 0}};
    const int rc = per_group ? dispatch<1, 0>(d8, d8, d8, d8, dv, dv, nullptr, nullptr, dv, M, N, K, nullptr)
                             : dispatch<0, 0>(d8, d8, nullptr, nullptr, dv, dv, dv, dv, dv, M, N, K, nullptr);
    plan5[0] = g_qs_plan.family;
    for (int i = 0; i < 4; ++i) plan5[1 + i] = g_qs_plan.p[i];
    g_qs_plan.active = 0;
    return rc;
}

extern "C" int qs_w4a8_per_chn_gemm(const int8_t* in_feats, const int8_t* kernel, const void* wscales,
                                    const void* ascales, const void* w_szs, const void* a_ssums, void* out_feats,
                                    int M, int N, int K, qs_stream_t stream) {
    return dispatch<0, 0>(in_feats, kernel, nullptr, nullptr, wscales, ascales, w_szs, a_ssums, out_feats, M, N, K,
                          stream);
}

extern "C" int qs_w4a8_per_group_gemm(const int8_t* in_feats, const int8_t* kernel, const int8_t* zeros,
                                      const int8_t* scales_i8, const void* wscales, const void* ascales,
                                      void* out_feats, int M, int N, int K, qs_stream_t stream) {
    return dispatch<1, 0>(in_feats, kernel, zeros, scales_i8, wscales, ascales, nullptr, nullptr, out_feats, M, N, K,
                          stream);
}

// gate_up GEMM + silu_and_mul in one launch where the kernel family has the epilogue, as two launches through `tmp`
// ([M, N] fp16) otherwise - bit-identical either way (the epilogue applies silu_and_mul's arithmetic to the fp16-rounded
// GEMM outputs)
qs_flag g_act_off = 0;   // qs_set_gemm_variant(3301 / 3300): always two launches / default (A/B, tests)
namespace {
template <int MODE>
int gate_up_silu(const int8_t* in_feats, const int8_t* kernel, const int8_t* zeros, const int8_t* scales_i8,
                 const void* wscales, const void* ascales, const void* w_szs, const void* a_ssums, void* out_act,
                 void* tmp, int M, int N, int K, qs_stream_t stream) {
    QS_REQUIRE(N > 0 && N % 128 == 0, "w4a8 gate_up + silu: N=%d must stack two multiples of 64 channels", N);
    if (M == 0) return QS_OK;
    QS_REQUIRE(out_act, "w4a8 gate_up + silu: null output");
    int rc = g_act_off ? QS_UNFUSED
                       : dispatch<MODE, 0>(in_feats, kernel, zeros, scales_i8, wscales, ascales, w_szs, a_ssums, out_act, M,
                                           N, K, stream, true);
    if (rc != QS_UNFUSED) return rc;
    QS_REQUIRE(tmp, "w4a8 gate_up + silu: this shape needs the [M, N] fp16 scratch `tmp` (two launches)");
    rc = dispatch<MODE, 0>(in_feats, kernel, zeros, scales_i8, wscales, ascales, w_szs, a_ssums, tmp, M, N, K, stream);
    if (rc != QS_OK) return rc;
    return qs_silu_and_mul(out_act, tmp, M, N / 2, stream);
}
}  // namespace

extern "C" int qs_w4a8_per_chn_gemm_silu_mul(const int8_t* in_feats, const int8_t* kernel, const void* wscales,
                                             const void* ascales, const void* w_szs, const void* a_ssums,
                                             void* out_act, void* tmp, int M, int N, int K, qs_stream_t stream) {
    return gate_up_silu<0>(in_feats, kernel, nullptr, nullptr, wscales, ascales, w_szs, a_ssums, out_act, tmp, M, N, K,
                           stream);
}

extern "C" int qs_w4a8_per_group_gemm_silu_mul(const int8_t* in_feats, const int8_t* kernel, const int8_t* zeros,
                                               const int8_t* scales_i8, const void* wscales, const void* ascales,
                                               void* out_act, void* tmp, int M, int N, int K, qs_stream_t stream) {
    return gate_up_silu<1>(in_feats, kernel, zeros, scales_i8, wscales, ascales, nullptr, nullptr, out_act, tmp, M, N, K,
                           stream);
}

// ---- K-slice planes (round 4) ---------------------------------------------------------------------------------------------
// The row-parallel GEMMs of a layer (o, down) are followed by a ROW kernel that reads their whole output anyway (residual add
// + norm + quant).  In this form the GEMM leaves its K slices as int32 planes [k_slices][M][N] - no cross-workgroup seam, no
// epilogue - and qs_add_residual_rms_norm_general_planes sums the planes and applies the GEMM's epilogue arithmetic itself:
// the kernel boundary that is there anyway is the hand-off (in-launch it costs a 2.4-3 us round trip under load, see
// gemm_w4a8_ring.hip).  Geometry by the ring kernel's byte model with the seam replaced by the planes' traffic (written once,
// read once: 8 bytes per element and slice, weighted by the CU : HBM rate ratio).
namespace {
struct PlanesGeo {
    int mt, wn, ks, mb;
};
bool planes_geometry(int mode, int M, int N, int K, PlanesGeo& g) {
    if (M < 1 || M > 1024 || N < 64 || N % 64 || K < 1024 || K % 128 || (size_t)M * K >= (1ull << 32) ||
        (size_t)N * K / 2 >= (1ull << 32))
        return false;
    const int mt_all = (M + 15) / 16;
    static const int geo[7][2] = {{4, 2}, {2, 2}, {4, 1}, {2, 1}, {1, 1}, {4, 4}, {8, 2}};   // <8,2>: forced only (tests)
    const int force = g_variant >= QS_GEMM_PLANES_GEOMETRY_BASE && g_variant < QS_GEMM_PLANES_GEOMETRY_END ? g_variant - QS_GEMM_PLANES_GEOMETRY_BASE : -1;   // tests / A-B: 4600 + 100*(ks-1) + 10*mt + wn
    long best = -1;
    for (int ks = 1; ks <= 4; ks *= 2)
        for (int i = 0; i < 7; ++i) {
            const int mt = geo[i][0], wn = geo[i][1];
            if (mt == 8 && force < 0) continue;         // (measured: the 128-token geometry never wins as planes)
            if (N % (64 * wn) != 0 || (K / 64) % ks != 0 || (K / 64 / ks) % (8 / wn) != 0) continue;
            if (force >= 0 && force != 100 * (ks - 1) + 10 * mt + wn) continue;
            const int mb = (mt_all + mt - 1) / mt;
            const long blocks = (long)mb * (N / (64 * wn)) * ks;
            const long pg = mode == 1 ? 8 * wn : 0;
            const long planes = (long)ks * M * N * 8 / 256 * 23 / 10;
            const long cost = ((blocks + 255) / 256) * (16 * mt + 32 * wn + pg) * (long)(K / ks) * (ks > 1 ? 11 : 10) / 10 + planes;
            if (best < 0 || cost < best) best = cost, g = {mt, wn, ks, mb};
        }
    return best >= 0;
}
template <int MODE>
int gemm_planes(const int8_t* A, const int8_t* W, const int8_t* zeros, const int8_t* scales8, int32_t* planes, int M, int N,
                int K, qs_stream_t stream) {
    QS_REQUIRE(A && W && planes && (MODE == 0 || (zeros && scales8)), "w4a8 gemm (planes): null pointer");
    PlanesGeo g;
    if (!planes_geometry(MODE, M, N, K, g)) {
        qs_set_error("w4a8 gemm (planes): no ring geometry for M=%d N=%d K=%d (ask qs_w4a8_gemm_planes_plan first)", M, N, K);
        return QS_ENOSUP;
    }
    return qs_launch_gemm_ring(MODE, 3, g.mt, g.wn, A, reinterpret_cast<const uint8_t*>(W), zeros, scales8, nullptr, nullptr,
                               nullptr, nullptr, planes, M, N, K, g.mb, g.ks, nullptr, nullptr, (hipStream_t)stream);
}
}  // namespace

extern "C" int qs_w4a8_gemm_planes_plan(int per_group, int M, int N, int K, int* plan4) {
    QS_REQUIRE(plan4, "w4a8 gemm planes plan: null output");
    PlanesGeo g = {0, 0, 0, 0};
    if (!planes_geometry(per_group ? 1 : 0, M, N, K, g)) g = {0, 0, 0, 0};      // k_slices == 0: not available, run the pair
    plan4[0] = g.ks, plan4[1] = g.mt, plan4[2] = g.wn, plan4[3] = g.mb;
    return QS_OK;
}
extern "C" int qs_w4a8_per_chn_gemm_planes(const int8_t* in_feats, const int8_t* kernel, int32_t* planes, int M, int N, int K,
                                           qs_stream_t stream) {
    return gemm_planes<0>(in_feats, kernel, nullptr, nullptr, planes, M, N, K, stream);
}
extern "C" int qs_w4a8_per_group_gemm_planes(const int8_t* in_feats, const int8_t* kernel, const int8_t* zeros,
                                             const int8_t* scales_i8, int32_t* planes, int M, int N, int K, qs_stream_t stream) {
    return gemm_planes<1>(in_feats, kernel, zeros, scales_i8, planes, M, N, K, stream);
}

extern "C" int qs_w4a8_per_chn_gemm_acc(const int8_t* in_feats, const int8_t* kernel, int32_t* acc_out, int M, int N,
                                        int K, qs_stream_t stream) {
    return dispatch<0, 1>(in_feats, kernel, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, acc_out, M, N, K,
                          stream);
}

extern "C" int qs_w4a8_per_group_gemm_acc(const int8_t* in_feats, const int8_t* kernel, const int8_t* zeros,
                                          const int8_t* scales_i8, int32_t* acc_out, int M, int N, int K,
                                          qs_stream_t stream) {
    return dispatch<1, 1>(in_feats, kernel, zeros, scales_i8, nullptr, nullptr, nullptr, nullptr, acc_out, M, N, K,
                          stream);
}
