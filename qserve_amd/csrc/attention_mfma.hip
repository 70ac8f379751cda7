// attention_mfma.hip -- KV4 decode attention on the matrix cores (gfx950), wave-autonomous flash-decoding.
//
// Same contract as decode_attention_kernel in attention.hip (reference: fused_attention.cpp:150-240,
// decoderMaskedMultiheadAttentionTemplate.hpp:717-2222, ZINT4 variant); different mapping:
//
//   * one workgroup per (sequence, KV head); its NW waves each own whole 64-token pages (page p -> wave p % NW) and
//     run QK -> online softmax -> PV on them WITHOUT any workgroup barrier; the partial (max, sum, out) triples are
//     merged once at the end through LDS (flash-decoding inside the workgroup).
//   * a page slice (4 KiB K + 4 KiB V + 512 B scales/zeros) is fetched lane-linearly (16 B per lane per load, fully
//     coalesced), one page ahead, and parked in a wave-private LDS buffer.
//   * Q.K^T on v_mfma_f32_16x16x32_f16: A = 16 tokens x 32 dims of the K page (lane (tok, kg) holds the 8 nibbles of
//     one dword, exactly converted to fp16 integers 0..15), B = the G query heads (padded to 16 columns).
//     score = ksc[tok] * (dot_raw - kzr[tok] * sum_d q_d) / sqrt(128): the per-token scale / zero point is applied to
//     the 16x16 result instead of to 128 x 64 elements.
//   * P.V on the same instruction: A = V^T (lane (8-dim group, kg) gathers one dword per token for 8 tokens and
//     transposes the 8x8 fp16 block in registers with v_perm_b32), B = P'^T with P' = p * vsc[tok] (scale folded into
//     the probabilities) and the zero-point term sum_t p_t vsc_t vzr_t subtracted from every output dim at the end.
//   * integer nibbles are exact in fp16 and all accumulation is fp32, so the result is the mathematically exact
//     attention over the de-quantised cache up to fp16 rounding of q, P' and the output (tighter than the reference's
//     own fp16 arithmetic; parity bar: 1e-3 on the output, tests/test_attention_gpu.py).
#include "common.h"

namespace {

constexpr int PAGE_TOK = 64;
constexpr int DH = 128;
constexpr int DHB = 64;        // KV4 bytes per token per head
constexpr int NW = 8;          // waves per workgroup

struct RopeCS {
    float c, s;
};
__device__ __forceinline__ RopeCS rope_coef(int pair, int pos, float base, int dim) {
    const float expo = (float)(2 * pair) / (float)dim;
    const float denom = (float)pow((double)base, (double)expo);
    const float ang = (float)pos / denom;
    RopeCS r;
    r.c = (float)cos((double)ang);
    r.s = (float)sin((double)ang);
    return r;
}
__device__ __forceinline__ void rope_pair(float a, float b, RopeCS cs, _Float16& oa, _Float16& ob) {
#pragma clang fp contract(off)
    const float ra = cs.c * a - cs.s * b;
    const float rb = cs.c * b + cs.s * a;
    oa = (_Float16)ra;
    ob = (_Float16)rb;
}

// 8 nibbles of x -> fp16 integers, order (e0,e4),(e1,e5),(e2,e6),(e3,e7)   (exact)
__device__ __forceinline__ void nib8_to_h2(u32 x, h2 (&o)[4]) {
    const u32 t = x >> 8;
    const u32 w0 = (x & 0x000F000Fu) | 0x64006400u;
    const u32 w1 = (x & 0x00F000F0u) | 0x64006400u;
    const u32 w2 = (t & 0x000F000Fu) | 0x64006400u;
    const u32 w3 = (t & 0x00F000F0u) | 0x64006400u;
    const h2 k1024 = {(_Float16)1024.f, (_Float16)1024.f};
    const h2 k16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    const h2 km64 = {(_Float16)-64.f, (_Float16)-64.f};
    o[0] = __builtin_bit_cast(h2, w0) - k1024;
    o[1] = __builtin_elementwise_fma(__builtin_bit_cast(h2, w1), k16, km64);
    o[2] = __builtin_bit_cast(h2, w2) - k1024;
    o[3] = __builtin_elementwise_fma(__builtin_bit_cast(h2, w3), k16, km64);
}

struct QParams {
    _Float16 scale, zero;
    float inv;
};
__device__ __forceinline__ QParams make_qparams(float mn, float mx) {
    QParams p;
    const float rng = mx - mn;
    p.scale = (_Float16)(rng / 15.f);
    p.zero = (_Float16)((-15.f * mn) / rng);
    p.inv = 1.0f / (float)p.scale;
    return p;
}
__device__ __forceinline__ void wave_quant_store4(_Float16 v0, _Float16 v1, uint8_t* dst, __half* scale_p,
                                                  __half* zero_p, int lane) {
    const float mx = wave_max(fmaxf((float)v0, (float)v1));
    const float mn = wave_min(fminf((float)v0, (float)v1));
    const QParams p = make_qparams(mn, mx);
    const unsigned u0 = rni_sat_u8(fmaf((float)v0, p.inv, (float)p.zero));
    const unsigned u1 = rni_sat_u8(fmaf((float)v1, p.inv, (float)p.zero));
    dst[lane] = (uint8_t)((u0 & 0xFu) | (u1 << 4));
    if (lane == 0) {
        *scale_p = __builtin_bit_cast(__half, p.scale);
        *zero_p = __builtin_bit_cast(__half, p.zero);
    }
}

__device__ __forceinline__ u32 pack_h2(float a, float b) {
    const h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(u32, v);
}

template <int G>
__global__ __launch_bounds__(NW * 64, 4) void decode_attention_mfma_kernel(
    const _Float16* __restrict__ q, const _Float16* __restrict__ k, const _Float16* __restrict__ v,
    const int64_t* __restrict__ kv_pointers, const int* __restrict__ lengths, _Float16* __restrict__ out,
    int num_heads, int num_kv_heads, int64_t q_stride0, int64_t kv_stride0, int max_blocks, int timestep,
    float rope_base) {
    __shared__ __attribute__((aligned(16))) uint8_t s_k[NW][PAGE_TOK * DHB];
    __shared__ __attribute__((aligned(16))) uint8_t s_v[NW][PAGE_TOK * DHB];
    __shared__ __attribute__((aligned(16))) _Float16 s_meta[NW][4][PAGE_TOK];   // k scale, k zero, v scale, v zero
    __shared__ __attribute__((aligned(16))) _Float16 s_q[16][DH];               // rotated q, rows >= G are zero
    __shared__ __attribute__((aligned(16))) _Float16 s_qp[16][DH];              // s_q in MFMA operand (nibble) order
    __shared__ __attribute__((aligned(16))) _Float16 s_knew[DH];
    __shared__ float s_cur[16];
    __shared__ float s_m[NW][G], s_l[NW][G];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hkv = blockIdx.x, b = blockIdx.y;
    const int tl = (lengths ? lengths[b] : timestep) - 1;
    if (tl < 0) return;
    const int64_t* ktab = kv_pointers + (size_t)b * 2 * max_blocks;
    const int64_t* vtab = ktab + max_blocks;
    const float inv_sqrt = 0.08838834764831845f;
    const int li = lane & 15, tg = lane >> 4;

    // ---- page fetch: LDS-DMA (global_load_lds) straight into the wave-private buffers, no staging registers.
    // K(p+1) is requested as soon as Q.K^T of page p has consumed the K buffer, V(p+1) after P.V of page p; each
    // transfer has about half an iteration of cover.  Completion is tracked with counted s_waitcnt vmcnt.
    const int npages = (tl + PAGE_TOK - 1) >> 6;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto dma_k = [&](int p) {     // 4 x 1 KiB data + 256 B (scales | zeros): 5 VMEM instructions
        const uint8_t* kbase = reinterpret_cast<const uint8_t*>(ktab[p]);
        const uint8_t* kd = kbase + (size_t)hkv * PAGE_TOK * DHB + lane * 16;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            __builtin_amdgcn_global_load_lds((gptr_t)(kd + e * 1024), (lptr_t)(&s_k[wave][e * 1024]), 16, 0, 0);
        const uint8_t* mb = kbase + (size_t)num_kv_heads * PAGE_TOK * DHB +
                            (size_t)((lane >> 5) * num_kv_heads + hkv) * PAGE_TOK * 2 + (lane & 31) * 4;
        __builtin_amdgcn_global_load_lds((gptr_t)mb, (lptr_t)(&s_meta[wave][0][0]), 4, 0, 0);
    };
    auto dma_v = [&](int p) {
        const uint8_t* vbase = reinterpret_cast<const uint8_t*>(vtab[p]);
        const uint8_t* vd = vbase + (size_t)hkv * PAGE_TOK * DHB + lane * 16;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            __builtin_amdgcn_global_load_lds((gptr_t)(vd + e * 1024), (lptr_t)(&s_v[wave][e * 1024]), 16, 0, 0);
        const uint8_t* mb = vbase + (size_t)num_kv_heads * PAGE_TOK * DHB +
                            (size_t)((lane >> 5) * num_kv_heads + hkv) * PAGE_TOK * 2 + (lane & 31) * 4;
        __builtin_amdgcn_global_load_lds((gptr_t)mb, (lptr_t)(&s_meta[wave][2][0]), 4, 0, 0);
    };
    if (wave < npages) {
        dma_k(wave);
        dma_v(wave);
    }

    // ---- phase A: RoPE of the G query heads and of k; quantise + store the new token's K and V ------------------
    const _Float16* qb = q + (size_t)b * q_stride0 + (size_t)hkv * G * DH;
    const _Float16* kb = k + (size_t)b * kv_stride0 + (size_t)hkv * DH;
    const _Float16* vb = v + (size_t)b * kv_stride0 + (size_t)hkv * DH;
    if (tid < 64) {
        const RopeCS cs = rope_coef(tid, tl, rope_base, DH);
#pragma unroll
        for (int h = 0; h < G; ++h) {
            _Float16 a, bb;
            rope_pair((float)qb[h * DH + tid], (float)qb[h * DH + 64 + tid], cs, a, bb);
            s_q[h][tid] = a;
            s_q[h][64 + tid] = bb;
        }
        _Float16 a, bb;
        rope_pair((float)kb[tid], (float)kb[64 + tid], cs, a, bb);
        s_knew[tid] = a;
        s_knew[64 + tid] = bb;
    } else {
        for (int i = tid - 64; i < (16 - G) * DH; i += NW * 64 - 64) s_q[G + i / DH][i % DH] = (_Float16)0.f;
    }
    __syncthreads();
    {
        const int blk = tl >> 6, slot = tl & 63;
        if (wave == 0) {
            uint8_t* pg = reinterpret_cast<uint8_t*>(ktab[blk]);
            __half* sc = reinterpret_cast<__half*>(pg + (size_t)num_kv_heads * PAGE_TOK * DHB);
            wave_quant_store4(s_knew[2 * lane], s_knew[2 * lane + 1], pg + ((size_t)hkv * PAGE_TOK + slot) * DHB,
                              sc + hkv * PAGE_TOK + slot, sc + num_kv_heads * PAGE_TOK + hkv * PAGE_TOK + slot, lane);
        } else if (wave == 1) {
            uint8_t* pg = reinterpret_cast<uint8_t*>(vtab[blk]);
            __half* sc = reinterpret_cast<__half*>(pg + (size_t)num_kv_heads * PAGE_TOK * DHB);
            wave_quant_store4(vb[2 * lane], vb[2 * lane + 1], pg + ((size_t)hkv * PAGE_TOK + slot) * DHB,
                              sc + hkv * PAGE_TOK + slot, sc + num_kv_heads * PAGE_TOK + hkv * PAGE_TOK + slot, lane);
        } else {
            for (int h = wave - 2; h < G; h += NW - 2) {
                float d = (float)s_q[h][lane] * (float)s_knew[lane] + (float)s_q[h][64 + lane] * (float)s_knew[64 + lane];
                d = wave_sum(d);
                if (lane == 0) s_cur[h] = d * inv_sqrt;
            }
        }
    }

    // ---- per-lane constants: B operand of Q.K^T (head li, dims 32 tg + 8 w + {0,4,1,5,2,6,3,7}) and sum_d q_d ----
    // The operand is parked in LDS (s_qp) and re-read per page: 16 fewer live registers in the page loop.
    float qsum = 0.f;
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const h8 x = *reinterpret_cast<const h8*>(&s_q[li][32 * tg + 8 * w]);
            *reinterpret_cast<h8*>(&s_qp[li][32 * tg + 8 * w]) = (h8){x[0], x[4], x[1], x[5], x[2], x[6], x[3], x[7]};
        }
    }
    for (int d = 0; d < DH; ++d) qsum += (float)s_q[li][d];   // fp32 sum of the fp16 rotated q of head li
    __syncthreads();

    v4f acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = (v4f){0.f, 0.f, 0.f, 0.f};
    float m_run = -3.0e38f, l_part = 0.f, corr = 0.f;

    for (int p = wave; p < npages; p += NW) {
        // K(p) landed?  Outstanding younger VMEM ops at this point: the 5 of V(p).
        asm volatile("s_waitcnt vmcnt(5) ; QS_LOOP_BEGIN" ::: "memory");
        const int valid = min(PAGE_TOK, tl - p * PAGE_TOK);

        // ---------------- Q.K^T : 4 tiles of 16 tokens ----------------
        v4f sc[4];
        h8 qB[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) qB[w] = *reinterpret_cast<const h8*>(&s_qp[li][32 * tg + 8 * w]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const v4u raw = *reinterpret_cast<const v4u*>(&s_k[wave][(16 * t + li) * DHB + 16 * tg]);
            v4f c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                h2 kk[4];
                nib8_to_h2(raw[w], kk);
                const h8 a = {kk[0][0], kk[0][1], kk[1][0], kk[1][1], kk[2][0], kk[2][1], kk[3][0], kk[3][1]};
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qB[w], c, 0, 0, 0);
            }
            // c[r] = dot_raw(token 16t + 4tg + r, head li); apply the token's scale / zero point
            const h4 ks = *reinterpret_cast<const h4*>(&s_meta[wave][0][16 * t + 4 * tg]);
            const h4 kz = *reinterpret_cast<const h4*>(&s_meta[wave][1][16 * t + 4 * tg]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = 16 * t + 4 * tg + r;
                const float s = (float)ks[r] * (c[r] - (float)kz[r] * qsum) * inv_sqrt;
                sc[t][r] = tok < valid ? s : -3.0e38f;
            }
        }
        // K buffer consumed (all ds_reads returned: their results fed the MFMAs above) -> request K(p+NW)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const bool more = p + NW < npages;
        if (more) dma_k(p + NW);
        // ---------------- online softmax (per head = per li; the 4 tg lanes of a head hold 16 tokens each) ---------
        float mx = sc[0][0];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        m_run = m_new;
        l_part *= alpha;
        corr *= alpha;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] *= alpha;
        }
        // V(p) landed?  Younger VMEM ops: the 5 of K(p+NW) if it was requested.
        if (more) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---------------- P.V : two half pages of 32 tokens ----------------
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            u32 pb[4];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = 2 * hp + tt;
                const h4 vs = *reinterpret_cast<const h4*>(&s_meta[wave][2][16 * t + 4 * tg]);
                const h4 vz = *reinterpret_cast<const h4*>(&s_meta[wave][3][16 * t + 4 * tg]);
                float pp[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tok = 16 * t + 4 * tg + r;
                    const float pe = __expf(sc[t][r] - m_new);          // 0 for masked tokens
                    l_part += pe;
                    // P' = p * v-scale, rounded to fp16 for the MFMA; the zero-point term uses the SAME rounded value
                    const float ps = tok < valid ? (float)(_Float16)(pe * (float)vs[r]) : 0.f;
                    corr += tok < valid ? ps * (float)vz[r] : 0.f;
                    pp[r] = ps;
                }
                pb[2 * tt] = pack_h2(pp[0], pp[1]);
                pb[2 * tt + 1] = pack_h2(pp[2], pp[3]);
            }
            const h8 pB = __builtin_bit_cast(h8, (v4u){pb[0], pb[1], pb[2], pb[3]});
            // V^T operand: lane (dim group li, kg = tg) gathers the dword of its 8 tokens; v_perm_b32 pairs byte bb of two
            // tokens into one word (0x00BB00AA), whose low / high nibbles become the fp16 pairs of dims 2bb / 2bb+1
            // through the exact magic-number conversion (no per-element shifts, only the raw dwords stay live).
            u32 raw[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int tok = 16 * (2 * hp + (jj >> 2)) + 4 * tg + (jj & 3);
                raw[jj] = *reinterpret_cast<const u32*>(&s_v[wave][tok * DHB + 4 * li]);
            }
            const h2 k1024 = {(_Float16)1024.f, (_Float16)1024.f};
            const h2 k16 = {(_Float16)0.0625f, (_Float16)0.0625f};
            const h2 km64 = {(_Float16)-64.f, (_Float16)-64.f};
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                u32 lo[4], hi[4];
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    const u32 W = __builtin_amdgcn_perm(raw[2 * pq + 1], raw[2 * pq], 0x0c000c00u | bb | ((4u + bb) << 16));
                    const u32 wl = (W & 0x000F000Fu) | 0x64006400u;
                    const u32 wh = (W & 0x00F000F0u) | 0x64006400u;
                    lo[pq] = __builtin_bit_cast(u32, __builtin_bit_cast(h2, wl) - k1024);
                    hi[pq] = __builtin_bit_cast(u32, __builtin_elementwise_fma(__builtin_bit_cast(h2, wh), k16, km64));
                }
                const h8 a_lo = __builtin_bit_cast(h8, (v4u){lo[0], lo[1], lo[2], lo[3]});
                const h8 a_hi = __builtin_bit_cast(h8, (v4u){hi[0], hi[1], hi[2], hi[3]});
                acc[2 * bb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, pB, acc[2 * bb], 0, 0, 0);
                acc[2 * bb + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, pB, acc[2 * bb + 1], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0) ; QS_LOOP_END" ::: "memory");
        if (more) dma_v(p + NW);
    }

    // ---- per-wave partials -> LDS.  Lane (head li, tg) holds out dims 8*(4tg + r) + e in acc[e][r] ---------------
    l_part += __shfl_xor(l_part, 16, 64);
    l_part += __shfl_xor(l_part, 32, 64);
    corr += __shfl_xor(corr, 16, 64);
    corr += __shfl_xor(corr, 32, 64);
    __syncthreads();   // every wave is done with its page buffers: reuse s_k/s_v as the [NW][G][DH] fp32 merge area
    float (*s_o)[G][DH] = reinterpret_cast<float (*)[G][DH]>(&s_k[0][0]);
    static_assert(sizeof(float) * G * DH <= PAGE_TOK * DHB, "merge area [NW][G][DH] fp32 must fit s_k");
    if (li < G) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_o[wave][li][8 * (4 * tg + r) + e] = acc[e][r] - corr;
        if (tg == 0) {
            s_m[wave][li] = m_run;
            s_l[wave][li] = l_part;
        }
    }
    __syncthreads();
    for (int o = tid; o < G * DH; o += NW * 64) {
        const int h = o / DH, d = o % DH;
        float M = s_cur[h];
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, s_m[w][h]);
        const float pc = __expf(s_cur[h] - M);
        float num = pc * (float)vb[d], den = pc + 1.e-6f;          // Template.hpp:1819 (sum + 1e-6)
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float f = __expf(s_m[w][h] - M);
            num += f * s_o[w][h][d];
            den += f * s_l[w][h];
        }
        out[((size_t)b * num_heads + (size_t)hkv * G + h) * DH + d] = (_Float16)(num / den);
    }
}

}  // namespace

// called from attention.hip's dispatcher for KV4
int qs_launch_decode_mfma(int G, dim3 grid, hipStream_t st, const _Float16* q, const _Float16* k, const _Float16* v,
                          const int64_t* kvp, const int* len, _Float16* out, int H, int Hkv, int64_t qs, int64_t kvs,
                          int mb, int timestep, float base) {
#define QS_LAUNCH_G(GG)                                                                                             \
    hipLaunchKernelGGL((decode_attention_mfma_kernel<GG>), grid, dim3(NW * 64), 0, st, q, k, v, kvp, len, out, H, Hkv, \
                       qs, kvs, mb, timestep, base)
    switch (G) {
        case 1: QS_LAUNCH_G(1); break;
        case 2: QS_LAUNCH_G(2); break;
        case 4: QS_LAUNCH_G(4); break;
        case 8: QS_LAUNCH_G(8); break;
        default:
            qs_set_error("single_query_attention: num_heads/num_kv_heads = %d not in {1,2,4,8}", G);
            return QS_ENOSUP;
    }
#undef QS_LAUNCH_G
    return qs_launch_status("single_query_attention");
}
