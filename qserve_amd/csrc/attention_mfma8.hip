// attention_mfma8.hip -- KV8 decode attention on the matrix cores (gfx950), wave-autonomous flash-decoding.
//
// The INT8-cache twin of attention_mfma.hip (reference: fused_attention.cpp:150-240,
// decoderMaskedMultiheadAttentionTemplate.hpp:717-2222, ZINT8 variant; de-quantisation
// decoderMaskedMultiheadAttentionUtils.h:2095-2107, store :2045-2053).  Same structure - one workgroup per
// (sequence, KV head[, KV split]), every wave owns whole 64-token pages, pages travel HBM -> LDS by LDS-DMA into
// wave-private buffers, Q.K^T and P.V on v_mfma_f32_16x16x32_f16, partials merged through LDS - with these differences:
//   * a page slice is 8 KiB K + 8 KiB V, so a workgroup has NW = 4 waves (two workgroups per CU);
//   * cache bytes become fp16 1024 + b with ONE v_perm_b32 per two elements (byte next to the constant 0x64); the
//     offsets 1024 * sum(q) and 1024 * sum(P') are removed from the 16x16 results afterwards, the per-token scale and
//     zero point are applied to the score (K) or folded into the probabilities (V), exactly as in the KV4 kernel;
//   * LDS images are bank-conflict free by permuting on the DMA SOURCE side: K rows keep their place and the 16-byte
//     chunk at position p of row r holds chunk p ^ ((r >> 1) & 7); V rows are stored in slot s = tok ^ ((tok >> 2) & 1)
//     (the probabilities are packed in the same order, so the contraction is unchanged).
#include "common.h"
#include "kv_quant.h"

namespace {

constexpr int PAGE_TOK = 64;
constexpr int DH = 128;
constexpr int DHB = 128;       // KV8 bytes per token per head
constexpr int NW = 4;          // waves per workgroup
constexpr int MAXP = 192;      // page-table entries cached in LDS per sequence (dispatcher: max_blocks <= MAXP)
constexpr int NVM = 9;         // VMEM instructions of one page-slice fetch (8 x 1 KiB + scales|zeros)

// 8-bit quantiser of one (token, head) vector held two elements per lane (Template.hpp:1045-1082, Utils.h:2045-2053)
__device__ __forceinline__ void wave_quant_store8(_Float16 v0, _Float16 v1, uint8_t* dst, __half* scale_p,
                                                  __half* zero_p, int lane) {
    const float mx = wave_max(fmaxf((float)v0, (float)v1));
    const float mn = wave_min(fminf((float)v0, (float)v1));
    const QParams p = make_qparams<false>(mn, mx);
    const unsigned u0 = quant_u8(v0, p), u1 = quant_u8(v1, p);
    *reinterpret_cast<uint16_t*>(dst + 2 * lane) = (uint16_t)(u0 | (u1 << 8));
    if (lane == 0) {
        *scale_p = __builtin_bit_cast(__half, p.scale);
        *zero_p = __builtin_bit_cast(__half, p.zero);
    }
}

__device__ __forceinline__ u32 pack_h2(float a, float b) {
    const h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(u32, v);
}

// cache policy of the page DMA: non-temporal (aux bit 1) - every KV byte is read exactly once per step (measured on the
// KV4 twin: -4 % at L = 1033 ... -12 % at L = 4096); QS_KV8_NT=0 at build time restores the default policy for A/B
#ifndef QS_KV8_NT
#define QS_KV8_NT 1
#endif
constexpr int NT_AUX = QS_KV8_NT ? 2 : 0;
typedef __attribute__((address_space(3))) const uint8_t* lds_u8;   // 32-bit LDS address (keeps ds_read, not flat_load)
typedef u32 v2u __attribute__((ext_vector_type(2)));
#define LDS_AT(T, p) (*(const __attribute__((address_space(3))) T*)(p))

template <int G>
__global__ __launch_bounds__(NW * 64, 2) void decode_attention_mfma8_kernel(
    const _Float16* __restrict__ q, const _Float16* __restrict__ k, const _Float16* __restrict__ v,
    const int64_t* __restrict__ kv_pointers, const int* __restrict__ lengths, _Float16* __restrict__ out,
    int num_heads, int num_kv_heads, int64_t q_stride0, int64_t kv_stride0, int max_blocks, int timestep,
    float rope_base, const float2* __restrict__ rope_tab, int rope_tab_len, int nsplit, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) uint8_t s_kv[2 * NW * PAGE_TOK * DHB];   // [K | V][wave][8 KiB]
    __shared__ __attribute__((aligned(16))) _Float16 s_meta[NW][4][PAGE_TOK];   // k scale, k zero, v scale, v zero
    __shared__ __attribute__((aligned(16))) _Float16 s_qp[16][DH];              // rotated q, rows >= G zero (B operand)
    __shared__ int64_t s_ptab[2][MAXP];                                         // page addresses of this sequence
    __shared__ __attribute__((aligned(16))) _Float16 s_knew[DH];
    __shared__ float s_cur[16];
    __shared__ float s_m[NW][G], s_l[NW][G];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hkv = blockIdx.x, b = blockIdx.y, z = blockIdx.z;
    const int tl = lengths ? lengths[b] - 1 : timestep;   // tlength, Template.hpp:901
    if (tl < 0) return;
    const int64_t* ktab = kv_pointers + (size_t)b * 2 * max_blocks;
    const int64_t* vtab = ktab + max_blocks;
    const float inv_sqrt = 0.08838834764831845f;
    const float qk_scale = inv_sqrt * 1.4426950408889634f;   // scores live in the log2 domain: exp2 everywhere
    constexpr int GP = G <= 1 ? 1 : G <= 2 ? 2 : G <= 4 ? 4 : 8;   // group size padded to a power of two (lane mapping)
    constexpr bool COMPACT = (GP <= 4);                            // softmax on compacted lanes (see the page loop)
    const int li = lane & 15, tg = lane >> 4;
    uint8_t* const s_kw = s_kv + wave * (PAGE_TOK * DHB);                // this wave's K page buffer
    uint8_t* const s_vw = s_kv + (NW + wave) * (PAGE_TOK * DHB);         // this wave's V page buffer

    // ---- page range of this workgroup (split-KV) and page fetch by LDS-DMA ---------------------------------------
    const int npages = (tl + PAGE_TOK - 1) >> 6;
    const int pps = (npages + nsplit - 1) / nsplit;
    const int p_begin = z * pps, p_end = min(npages, p_begin + pps);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // instruction e copies LDS rows 8e .. 8e+7 (lane -> row 8e + (lane >> 3), 16-byte position lane & 7)
    const int r8 = lane >> 3, p8 = lane & 7;
    auto dma_k = [&](int64_t page) {
        const uint8_t* kbase = reinterpret_cast<const uint8_t*>(page);
        const uint8_t* kd = kbase + (u32)(hkv * PAGE_TOK * DHB);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = 8 * e + r8;
            const u32 off = row * DHB + ((p8 ^ ((row >> 1) & 7)) * 16);              // chunk swizzle on the source side
            __builtin_amdgcn_global_load_lds((gptr_t)(kd + off), (lptr_t)(s_kw + e * 1024), 16, 0, NT_AUX);
        }
        const uint8_t* mb = kbase + (u32)(num_kv_heads * PAGE_TOK * DHB +
                                          ((lane >> 5) * num_kv_heads + hkv) * PAGE_TOK * 2 + (lane & 31) * 4);
        __builtin_amdgcn_global_load_lds((gptr_t)mb, (lptr_t)(&s_meta[wave][0][0]), 4, 0, 0);
    };
    auto dma_v = [&](int64_t page) {
        const uint8_t* vbase = reinterpret_cast<const uint8_t*>(page);
        const uint8_t* vd = vbase + (u32)(hkv * PAGE_TOK * DHB);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int slot = 8 * e + r8;
            const int tok = slot ^ ((slot >> 2) & 1);                                 // row permutation on the source side
            const u32 off = tok * DHB + p8 * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)(vd + off), (lptr_t)(s_vw + e * 1024), 16, 0, NT_AUX);
        }
        const uint8_t* mb = vbase + (u32)(num_kv_heads * PAGE_TOK * DHB +
                                          ((lane >> 5) * num_kv_heads + hkv) * PAGE_TOK * 2 + (lane & 31) * 4);
        __builtin_amdgcn_global_load_lds((gptr_t)mb, (lptr_t)(&s_meta[wave][2][0]), 4, 0, 0);
    };
    if (p_begin + wave < p_end) {
        dma_k(ktab[p_begin + wave]);
        dma_v(vtab[p_begin + wave]);
    }
    for (int i = tid; i < 2 * MAXP; i += NW * 64) {
        const int pi = i >> 1;
        if (pi < npages) s_ptab[i & 1][pi] = (i & 1) ? vtab[pi] : ktab[pi];
    }

    // ---- phase A: RoPE of the G query heads and of k; quantise + store the new token's K and V (split 0) ------------
    const _Float16* qb = q + (size_t)b * q_stride0 + (size_t)hkv * G * DH;
    const _Float16* kb = k + (size_t)b * kv_stride0 + (size_t)hkv * DH;
    const _Float16* vb = v + (size_t)b * kv_stride0 + (size_t)hkv * DH;
    if (tid < 64) {
        RopeCS cs;
        if (rope_tab && tl < rope_tab_len) {
            const float2 t = rope_tab[(size_t)tl * 64 + tid];
            cs.c = t.x;
            cs.s = t.y;
        } else {
            cs = rope_coef(tid, tl, rope_base, DH);
        }
#pragma unroll
        for (int h = 0; h < G; ++h) {
            _Float16 a, bb;
            rope_pair((float)qb[h * DH + tid], (float)qb[h * DH + 64 + tid], cs, a, bb);
            s_qp[h][tid] = a;
            s_qp[h][64 + tid] = bb;
        }
        _Float16 a, bb;
        rope_pair((float)kb[tid], (float)kb[64 + tid], cs, a, bb);
        s_knew[tid] = a;
        s_knew[64 + tid] = bb;
    } else {
        for (int i = tid - 64; i < (16 - G) * DH; i += NW * 64 - 64) s_qp[G + i / DH][i % DH] = (_Float16)0.f;
    }
    __syncthreads();
    {
        const int blk = tl >> 6, slot = tl & 63;
        if (wave == 0 && z == 0) {
            uint8_t* pg = reinterpret_cast<uint8_t*>(ktab[blk]);
            __half* sc = reinterpret_cast<__half*>(pg + (size_t)num_kv_heads * PAGE_TOK * DHB);
            wave_quant_store8(s_knew[2 * lane], s_knew[2 * lane + 1], pg + ((size_t)hkv * PAGE_TOK + slot) * DHB,
                              sc + hkv * PAGE_TOK + slot, sc + num_kv_heads * PAGE_TOK + hkv * PAGE_TOK + slot, lane);
        } else if (wave == 1 && z == 0) {
            uint8_t* pg = reinterpret_cast<uint8_t*>(vtab[blk]);
            __half* sc = reinterpret_cast<__half*>(pg + (size_t)num_kv_heads * PAGE_TOK * DHB);
            wave_quant_store8(vb[2 * lane], vb[2 * lane + 1], pg + ((size_t)hkv * PAGE_TOK + slot) * DHB,
                              sc + hkv * PAGE_TOK + slot, sc + num_kv_heads * PAGE_TOK + hkv * PAGE_TOK + slot, lane);
        } else if (wave >= 2) {
            for (int h = wave - 2; h < G; h += NW - 2) {
                float d = (float)s_qp[h][lane] * (float)s_knew[lane] + (float)s_qp[h][64 + lane] * (float)s_knew[64 + lane];
                d = wave_sum(d);
                if (lane == 0) s_cur[h] = d * qk_scale;
            }
        }
    }
    __syncthreads();

    // per-lane constants of head li: qsum = sum of the head's 128 q values; the 1024+b operand form adds 1024*qsum
    float qsum;
    {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const h8 x = *reinterpret_cast<const h8*>(&s_qp[li & (GP - 1)][32 * tg + 8 * w]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)x[j];
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        qsum = s;
    }
    const float nqoff = -1024.f * qsum;

    u32 c_magic = 0x64646464u, c_magic2 = 0x64006400u;
    asm volatile("" : "+v"(c_magic), "+v"(c_magic2));

    v4f acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = (v4f){0.f, 0.f, 0.f, 0.f};
    float m_run = -3.0e38f, l_part = 0.f, corr = 0.f, psum = 0.f;

    for (int p = p_begin + wave; p < p_end; p += NW) {
        // K(p) landed?  Outstanding younger VMEM ops at this point: the NVM of V(p).
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NVM) : "memory");
        const bool more = p + NW < p_end;
        const int valid = min(PAGE_TOK, tl - p * PAGE_TOK);
        const bool full = valid == PAGE_TOK;   // wave-uniform: only the last page needs masking

        // per-lane LDS addresses re-derived every page (see attention_mfma.hip: nothing per-lane may live across pages)
        u32 lid;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lid));
        const int li_ = lid & 15, tg_ = lid >> 4;
        const int kp0 = (2 * tg_) ^ ((li_ >> 1) & 7);                  // position of this lane's first 16-byte chunk
        const lds_u8 kl0 = (lds_u8)s_kw + (li_ * DHB + kp0 * 16);
        const lds_u8 kl1 = (lds_u8)s_kw + (li_ * DHB + (kp0 ^ 1) * 16);
        const lds_u8 vl = (lds_u8)s_vw + ((4 * tg_) * DHB + 8 * li_);
        const lds_u8 ml = (lds_u8)(&s_meta[wave][0][0]) + 8 * tg_;
        const lds_u8 ql = (lds_u8)(&s_qp[0][0]) + (li_ * (DH * 2) + 64 * tg_);
        const bool odd = tg_ & 1;
        // ---------------- Q.K^T : 4 tiles of 16 tokens ----------------
        v4f craw[4];   // craw[t][r] = raw dot (offsets already cancelled) of token 16t + 4tg + r with head li
        h8 qB[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) qB[w] = LDS_AT(h8, ql + 16 * w);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const v4u ra = LDS_AT(v4u, kl0 + 16 * t * DHB);       // dims 32tg + 0..15 of token 16t + li
            const v4u rb = LDS_AT(v4u, kl1 + 16 * t * DHB);       // dims 32tg + 16..31
            v4f c = {nqoff, nqoff, nqoff, nqoff};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const u32 xa = w < 2 ? ra[2 * w] : rb[2 * w - 4], xb = w < 2 ? ra[2 * w + 1] : rb[2 * w - 3];
                const v4u a4 = {__builtin_amdgcn_perm(xa, c_magic, 0x00050004u), __builtin_amdgcn_perm(xa, c_magic, 0x00070006u),
                                __builtin_amdgcn_perm(xb, c_magic, 0x00050004u), __builtin_amdgcn_perm(xb, c_magic, 0x00070006u)};
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a4), qB[w], c, 0, 0, 0);
            }
            craw[t] = c;
        }
        // softmax on compacted lanes for G <= 4 (tile t' -> lanes li = G t' + h, see attention_mfma.hip), scores in the
        // log2 domain
        u32 pbv[2][4];
        float m_new;
        float scc[4];
        float sc8[8];                                   // GP == 8: two lane groups, 8 scores per lane
        const int tq2 = li_ >> 3;
        const int tq_raw = li_ / GP;
        const bool lane_ok = GP == 4 || !COMPACT || tq_raw < 4;
        const int tq = GP == 4 ? tq_raw : min(tq_raw, 3);
        if constexpr (COMPACT) {
            float (&sc)[4] = scc;
            const h4 ks = LDS_AT(h4, ml + 32 * tq);
            const h4 kz = LDS_AT(h4, ml + 2 * PAGE_TOK + 32 * tq);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // (scalar copies first: __builtin_bit_cast of a vector-element lvalue reads element 0)
                const float c0 = craw[0][r], c1 = craw[1][r], c2 = craw[2][r], c3 = craw[3][r];
                int x = __builtin_bit_cast(int, c0);
                if constexpr (GP == 4) {
                    x = __builtin_amdgcn_update_dpp(x, __builtin_bit_cast(int, c1), 0x114, 0xF, 0x2, false);
                    x = __builtin_amdgcn_update_dpp(x, __builtin_bit_cast(int, c2), 0x118, 0xF, 0x4, false);
                    x = __builtin_amdgcn_update_dpp(x, __builtin_bit_cast(int, c3), 0x11C, 0xF, 0x8, false);
                } else {
                    const int s1 = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c1), 0x110 + GP, 0xF, 0xF, true);
                    const int s2 = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c2), 0x110 + 2 * GP, 0xF, 0xF, true);
                    const int s3 = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c3), 0x110 + 3 * GP, 0xF, 0xF, true);
                    x = tq_raw == 1 ? s1 : x;
                    x = tq_raw == 2 ? s2 : x;
                    x = tq_raw == 3 ? s3 : x;
                }
                sc[r] = ((float)ks[r] * qk_scale) * (__builtin_bit_cast(float, x) - (float)kz[r] * qsum);
                if (!lane_ok) sc[r] = -3.0e38f;
            }
            if (!full) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * tq + 4 * tg_ + r >= valid) sc[r] = -3.0e38f;   // also discards NaN from garbage scales
            }
        } else {   // GP == 8
            // two lane groups (see attention_mfma.hip): li < 8 keeps tiles 0 / 2 of head li, li >= 8 takes tiles 1 / 3
            const h4 ksa = LDS_AT(h4, ml + 32 * tq2);
            const h4 kza = LDS_AT(h4, ml + 2 * PAGE_TOK + 32 * tq2);
            const h4 ksb = LDS_AT(h4, ml + 32 * (2 + tq2));
            const h4 kzb = LDS_AT(h4, ml + 2 * PAGE_TOK + 32 * (2 + tq2));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float c0 = craw[0][r], c1 = craw[1][r], c2 = craw[2][r], c3 = craw[3][r];
                const int xa = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, c0), __builtin_bit_cast(int, c1), 0x118,
                                                           0xF, 0xC, false);
                const int xb = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, c2), __builtin_bit_cast(int, c3), 0x118,
                                                           0xF, 0xC, false);
                sc8[r] = ((float)ksa[r] * qk_scale) * (__builtin_bit_cast(float, xa) - (float)kza[r] * qsum);
                sc8[4 + r] = ((float)ksb[r] * qk_scale) * (__builtin_bit_cast(float, xb) - (float)kzb[r] * qsum);
            }
            if (!full) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (16 * tq2 + 4 * tg_ + r >= valid) sc8[r] = -3.0e38f;
                    if (16 * (2 + tq2) + 4 * tg_ + r >= valid) sc8[4 + r] = -3.0e38f;
                }
            }
        }
        // K buffer consumed -> request K(p+NW)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (more) dma_k(s_ptab[0][p + NW]);
        // ---------------- online softmax ----------------
        {
            float mx;
            if constexpr (COMPACT) {
                mx = fmaxf(fmaxf(scc[0], scc[1]), fmaxf(scc[2], scc[3]));
                mx = fmaxf(mx, xor_lane(mx, lid, GP));
                mx = fmaxf(mx, xor_lane(mx, lid, 2 * GP));
            } else {   // GP == 8
                mx = sc8[0];
#pragma unroll
                for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sc8[j]);
                mx = fmaxf(mx, xor_lane(mx, lid, 8));
            }
            mx = fmaxf(mx, xor_lane(mx, lid, 16));
            mx = fmaxf(mx, xor_lane(mx, lid, 32));
            m_new = fmaxf(m_run, mx);
        }
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        if (__any(alpha != 1.0f)) {
            l_part *= alpha;
            corr *= alpha;
            psum *= alpha;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] *= alpha;
        }
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NVM) : "memory");   // V(p) landed (K(p+NW) may be in flight)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // P' = fp16(p * v-scale); V rows sit in slot tok ^ ((tok >> 2) & 1): lanes of odd tg see each token pair swapped
        if constexpr (COMPACT) {
            const h4 vs = LDS_AT(h4, ml + 4 * PAGE_TOK + 32 * tq);
            const h4 vz = LDS_AT(h4, ml + 6 * PAGE_TOK + 32 * tq);
            float pp[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pe = __builtin_amdgcn_exp2f(scc[r] - m_new);   // 0 for masked tokens
                l_part += pe;
                float ps = (float)(_Float16)(pe * (float)vs[r]);           // P' rounded to fp16 (what the MFMA sees)
                float pz = ps * (float)vz[r];
                if ((!full && 16 * tq + 4 * tg_ + r >= valid) || !lane_ok) {   // garbage (possibly NaN) scales of unused slots
                    ps = 0.f;
                    pz = 0.f;
                }
                corr += pz;
                psum += ps;
                pp[r] = ps;
            }
            const int pk0 = (int)(odd ? pack_h2(pp[1], pp[0]) : pack_h2(pp[0], pp[1]));
            const int pk1 = (int)(odd ? pack_h2(pp[3], pp[2]) : pack_h2(pp[2], pp[3]));
            pbv[0][0] = (u32)pk0;
            pbv[0][1] = (u32)pk1;
            pbv[0][2] = (u32)__builtin_amdgcn_update_dpp(0, pk0, 0x100 + GP, 0xF, 0xF, true);
            pbv[0][3] = (u32)__builtin_amdgcn_update_dpp(0, pk1, 0x100 + GP, 0xF, 0xF, true);
            pbv[1][0] = (u32)__builtin_amdgcn_update_dpp(0, pk0, 0x100 + 2 * GP, 0xF, 0xF, true);
            pbv[1][1] = (u32)__builtin_amdgcn_update_dpp(0, pk1, 0x100 + 2 * GP, 0xF, 0xF, true);
            pbv[1][2] = (u32)__builtin_amdgcn_update_dpp(0, pk0, 0x100 + 3 * GP, 0xF, 0xF, true);
            pbv[1][3] = (u32)__builtin_amdgcn_update_dpp(0, pk1, 0x100 + 3 * GP, 0xF, 0xF, true);
        } else {   // GP == 8
            const h4 vsa = LDS_AT(h4, ml + 4 * PAGE_TOK + 32 * tq2);
            const h4 vza = LDS_AT(h4, ml + 6 * PAGE_TOK + 32 * tq2);
            const h4 vsb = LDS_AT(h4, ml + 4 * PAGE_TOK + 32 * (2 + tq2));
            const h4 vzb = LDS_AT(h4, ml + 6 * PAGE_TOK + 32 * (2 + tq2));
            float pp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = j & 3, t = j < 4 ? tq2 : 2 + tq2;
                const float pe = __builtin_amdgcn_exp2f(sc8[j] - m_new);   // 0 for masked tokens
                l_part += pe;
                float ps = (float)(_Float16)(pe * (float)(j < 4 ? vsa[r] : vsb[r]));
                float pz = ps * (float)(j < 4 ? vza[r] : vzb[r]);
                if (!full && 16 * t + 4 * tg_ + r >= valid) {
                    ps = 0.f;
                    pz = 0.f;
                }
                corr += pz;
                psum += ps;
                pp[j] = ps;
            }
            const int pka0 = (int)(odd ? pack_h2(pp[1], pp[0]) : pack_h2(pp[0], pp[1]));
            const int pka1 = (int)(odd ? pack_h2(pp[3], pp[2]) : pack_h2(pp[2], pp[3]));
            const int pkb0 = (int)(odd ? pack_h2(pp[5], pp[4]) : pack_h2(pp[4], pp[5]));
            const int pkb1 = (int)(odd ? pack_h2(pp[7], pp[6]) : pack_h2(pp[6], pp[7]));
            pbv[0][0] = (u32)pka0;
            pbv[0][1] = (u32)pka1;
            pbv[0][2] = (u32)__builtin_amdgcn_update_dpp(0, pka0, 0x108, 0xF, 0xF, true);
            pbv[0][3] = (u32)__builtin_amdgcn_update_dpp(0, pka1, 0x108, 0xF, 0xF, true);
            pbv[1][0] = (u32)pkb0;
            pbv[1][1] = (u32)pkb1;
            pbv[1][2] = (u32)__builtin_amdgcn_update_dpp(0, pkb0, 0x108, 0xF, 0xF, true);
            pbv[1][3] = (u32)__builtin_amdgcn_update_dpp(0, pkb1, 0x108, 0xF, 0xF, true);
        }
        // ---------------- P.V : two half pages of 32 tokens ----------------
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            const h8 pB = __builtin_bit_cast(h8, (v4u){pbv[hp][0], pbv[hp][1], pbv[hp][2], pbv[hp][3]});
            v2u raw[8];                                                  // 8 dims (bytes) of each of the lane's 8 slots
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) raw[jj] = LDS_AT(v2u, vl + (16 * (2 * hp + (jj >> 2)) + (jj & 3)) * DHB);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                u32 w4[4];
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    const u32 hi = e < 4 ? raw[2 * pq + 1].x : raw[2 * pq + 1].y, lo = e < 4 ? raw[2 * pq].x : raw[2 * pq].y;
                    w4[pq] = __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (e & 3) | ((4u + (e & 3)) << 16)) | c_magic2;
                }
                const h8 a_e = __builtin_bit_cast(h8, (v4u){w4[0], w4[1], w4[2], w4[3]});
                acc[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_e, pB, acc[e], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (more) dma_v(s_ptab[1][p + NW]);
    }

    // ---- per-wave partials -> LDS.  Lane (head li, tg) holds out dims 8*(4tg + r) + e in acc[e][r] ---------------
    const u32 lid2 = fresh_lane_id();   // nothing lane-derived lives across the page loop
    if constexpr (GP == 8) {
        l_part += xor_lane(l_part, lid2, 8);
        corr += xor_lane(corr, lid2, 8);
        psum += xor_lane(psum, lid2, 8);
    }
    if constexpr (COMPACT) {
        l_part += xor_lane(l_part, lid2, GP);
        l_part += xor_lane(l_part, lid2, 2 * GP);
        corr += xor_lane(corr, lid2, GP);
        corr += xor_lane(corr, lid2, 2 * GP);
        psum += xor_lane(psum, lid2, GP);
        psum += xor_lane(psum, lid2, 2 * GP);
    }
    l_part += xor_lane(l_part, lid2, 16);
    l_part += xor_lane(l_part, lid2, 32);
    corr += xor_lane(corr, lid2, 16);
    corr += xor_lane(corr, lid2, 32);
    psum += xor_lane(psum, lid2, 16);
    psum += xor_lane(psum, lid2, 32);
    corr += 1024.f * psum;                              // zero-point term + the 1024 offsets of the V operand
    __syncthreads();   // every wave is done with its page buffers: reuse them as the [NW][G][DH+4] fp32 merge area
    constexpr int OS = DH + 4;
    float (*s_o)[G][OS] = reinterpret_cast<float (*)[G][OS]>(&s_kv[0]);
    static_assert(sizeof(float) * NW * G * OS <= sizeof(s_kv), "merge area must fit the page buffers");
    const int li2 = lid2 & 15, tg2 = lid2 >> 4, tid2 = wave * 64 + (int)lid2;
    if (li2 < G) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_o[wave][li2][8 * (4 * tg2 + r) + e] = acc[e][r] - corr;
        if (tg2 == 0) {
            s_m[wave][li2] = m_run;
            s_l[wave][li2] = l_part;
        }
    }
    __syncthreads();
    for (int o = tid2; o < G * DH; o += NW * 64) {
        const int h = o / DH, d = o % DH;
        float M = z == 0 ? s_cur[h] : -3.0e38f;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, s_m[w][h]);
        const float pc = z == 0 ? __builtin_amdgcn_exp2f(s_cur[h] - M) : 0.f;      // the new token's own term (split 0 only)
        float num = pc * (float)vb[d], den = pc;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float f = __builtin_amdgcn_exp2f(s_m[w][h] - M);
            num += f * s_o[w][h][d];
            den += f * s_l[w][h];
        }
        if (nsplit == 1) {
            out[((size_t)b * num_heads + (size_t)hkv * G + h) * DH + d] = (_Float16)(num / (den + 1.e-6f));   // Template.hpp:1819
        } else {                                                    // un-normalised partial: [O(128) | M | L]
            float* pw = ws + ((((size_t)b * num_kv_heads + hkv) * nsplit + z) * G + h) * (DH + 2);
            pw[d] = num;
            if (d == 0) {
                pw[DH] = M * 0.6931471805599453f;   // the merge kernel works in natural-log units
                pw[DH + 1] = den;
            }
        }
    }
}

}  // namespace

// shared with attention_mfma.hip
const float2* qs_rope_table(float base, int max_pos, hipStream_t st, int* len_out);
float* qs_split_workspace(size_t bytes, hipStream_t st);
size_t qs_split_workspace_capacity();
int qs_attn_choose_splits(int blocks, int pages, int kv8, int fused_quant);
void qs_launch_attention_merge(const float* ws, _Float16* out, int H, int Hkv, int G, int nsplit, int batch, hipStream_t st);

// called from attention.hip's dispatcher for KV8.  force_split: 0 = heuristic, n > 0 = exactly n splits (tests)
int qs_launch_decode_mfma8(int G, dim3 grid, hipStream_t st, const _Float16* q, const _Float16* k, const _Float16* v,
                           const int64_t* kvp, const int* len, _Float16* out, int H, int Hkv, int64_t qs, int64_t kvs,
                           int mb, int timestep, float base, int max_pos, int force_split) {
    if (G < 1 || G > 8) {
        qs_set_error("single_query_attention: num_heads/num_kv_heads = %d not in 1..8", G);
        return QS_ENOSUP;
    }
    int tab_len = 0;
    const float2* tab = g_qs_attn_plan.active ? nullptr : qs_rope_table(base, max_pos, st, &tab_len);
    const int blocks = (int)(grid.x * grid.y);
    const int pages_max = (timestep + PAGE_TOK - 1) / PAGE_TOK;
    int nsplit = force_split > 0 ? force_split : qs_attn_choose_splits(blocks, pages_max, 1, 0);   // attention_mfma.hip
    if (g_qs_attn_plan.active) {
        g_qs_attn_plan.family = 2, g_qs_attn_plan.nsplit = nsplit, g_qs_attn_plan.waves = NW;
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
#define QS_LAUNCH_G(GG)                                                                                              \
    hipLaunchKernelGGL((decode_attention_mfma8_kernel<GG>), grid, dim3(NW * 64), 0, st, q, k, v, kvp, len, out, H, Hkv, \
                       qs, kvs, mb, timestep, base, tab, tab_len, nsplit, ws)
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
    if (nsplit > 1) qs_launch_attention_merge(ws, out, H, Hkv, G, nsplit, (int)grid.y, st);
    return qs_launch_status("single_query_attention");
}
