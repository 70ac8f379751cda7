// attention.hip -- paged, quantised-KV decode attention + prefill KV writer for MI355X (gfx950).
//
// Behaviour follows (not code):
//   single_query_attention ........... kernels/csrc/fused_attention/fused_attention.cpp:150-240 and
//                                      decoderMaskedMultiheadAttentionTemplate.hpp:717-2222 (ZINT4 / ZINT8 variants)
//   apply_bias_rope_update_kv_cache .. update_kv_cache.cu:20-108, applyBiasRopeUpdateKVCache.h:94-455
//   compute_padding_offsets .......... input_metadata_helper.cu:11-45
//   page layout ...................... kvCacheUtils.h:47-126:  [Hkv][64 tok][Dh' bytes] | half scale[Hkv][64] |
//                                      half zero[Hkv][64];  Dh' = 64 (KV4) / 128 (KV8)
//
// MI355X design of the decode kernel (differs from the reference on purpose):
//   * ONE workgroup (256 threads = 4 wave64) per (sequence, KV head) serves all G = H/Hkv query heads of the group,
//     so every KV byte is read from HBM once (the reference launches one block per QUERY head and re-reads the
//     group's pages G times).
//   * a page's slice for one KV head is contiguous (4 KiB K + 4 KiB V for KV4): it is fetched with one 16-byte
//     load per thread (256 x 16 B = 4 KiB, fully coalesced), double-buffered through LDS (register-staged
//     prefetch of page p+1 while page p is consumed), then read back in the two shapes the math wants:
//       QK:  lane = (token, 32-dim quarter), ds_read_b128, v_dot2_f32_f16 against q held in registers,
//            4-lane butterfly;   PV: lane = (8-dim group, token slot), ds_read_b32, fp32 accumulators.
//   * flash-style online softmax per 64-token page (running max / sum per head), so shared memory does not grow
//     with the context length (the reference needs O(L) smem and a separate kernel variant above 2048 tokens).
//   * loads of tokens >= length are masked by construction (whole pages only below the last valid page, and the
//     tail of the last page gets p = 0); the reference's out-of-range reads do not exist.
//   * numerics: KV4 dequantisation reproduces the reference bit for bit (exact nibble -> fp16, then one
//     hfma2(h, half(scale), half(-scale*zero)), Utils.h:2190-2213); dot products and P.V accumulate in fp32.
#include "common.h"
#include "kv_quant.h"

namespace {

constexpr int TPB = 256;
constexpr int PAGE_TOK = 64;
constexpr int DH = 128;

// exact uint4 -> fp16 for the 8 nibbles of x, in the order (e0,e4),(e1,e5),(e2,e6),(e3,e7) (Utils.h:2125-2188)
__device__ __forceinline__ void nib8_to_h2(u32 x, h2 (&o)[4]) {
    const u32 t = x >> 8;
    u32 w0 = (x & 0x000F000Fu) | 0x64006400u;
    u32 w1 = (x & 0x00F000F0u) | 0x64006400u;
    u32 w2 = (t & 0x000F000Fu) | 0x64006400u;
    u32 w3 = (t & 0x00F000F0u) | 0x64006400u;
    const h2 k1024 = {(_Float16)1024.f, (_Float16)1024.f};
    const h2 k16 = {(_Float16)0.0625f, (_Float16)0.0625f};
    const h2 km64 = {(_Float16)-64.f, (_Float16)-64.f};
    o[0] = __builtin_bit_cast(h2, w0) - k1024;
    o[1] = __builtin_elementwise_fma(__builtin_bit_cast(h2, w1), k16, km64);
    o[2] = __builtin_bit_cast(h2, w2) - k1024;
    o[3] = __builtin_elementwise_fma(__builtin_bit_cast(h2, w3), k16, km64);
}

// Quantise 128 fp16 values held as `vals[lane*2], vals[lane*2+1]` by the 64 lanes of ONE wave into dst (bytes of one
// token/head) and write scale / zero.  All 64 lanes must call.
template <bool INT4>
__device__ __forceinline__ void wave_quant_store(_Float16 v0, _Float16 v1, uint8_t* dst, __half* scale_p,
                                                 __half* zero_p, int lane) {
    const float mx = wave_max(fmaxf((float)v0, (float)v1));
    const float mn = wave_min(fminf((float)v0, (float)v1));
    const QParams p = make_qparams<INT4>(mn, mx);
    const unsigned u0 = quant_u8(v0, p), u1 = quant_u8(v1, p);
    if (INT4) {
        dst[lane] = (uint8_t)((u0 & 0xFu) | (u1 << 4));               // Utils.h:1838-1852
    } else {
        *reinterpret_cast<uint16_t*>(dst + 2 * lane) = (uint16_t)(u0 | (u1 << 8));
    }
    if (lane == 0) {
        *scale_p = __builtin_bit_cast(__half, p.scale);
        *zero_p = __builtin_bit_cast(__half, p.zero);
    }
}

struct PageAddr {
    const uint8_t* data;     // this head's [64][DHB] bytes
    const __half* scale;     // this head's 64 scales
    const __half* zero;
};
template <int DHB>
__device__ __forceinline__ PageAddr page_addr(int64_t page, int hkv, int num_kv_heads) {
    const uint8_t* base = reinterpret_cast<const uint8_t*>(page);
    PageAddr a;
    a.data = base + (size_t)hkv * PAGE_TOK * DHB;
    const __half* sc = reinterpret_cast<const __half*>(base + (size_t)num_kv_heads * PAGE_TOK * DHB);
    a.scale = sc + hkv * PAGE_TOK;
    a.zero = sc + num_kv_heads * PAGE_TOK + hkv * PAGE_TOK;
    return a;
}

// ---------------------------------------------------------------------------------------------------------
// decode attention
// ---------------------------------------------------------------------------------------------------------
template <int G, bool INT4>
__global__ __launch_bounds__(TPB) void decode_attention_kernel(
    const _Float16* __restrict__ q, const _Float16* __restrict__ k, const _Float16* __restrict__ v,
    const int64_t* __restrict__ kv_pointers, const int* __restrict__ lengths, _Float16* __restrict__ out,
    int num_heads, int num_kv_heads, int64_t q_stride0, int64_t kv_stride0, int max_blocks, int timestep,
    float rope_base) {
    constexpr int DHB = INT4 ? DH / 2 : DH;            // bytes per token per head
    constexpr int PAGE_BYTES = PAGE_TOK * DHB;         // 4 KiB / 8 KiB per head per page
    constexpr int NLD = PAGE_BYTES / (TPB * 16);       // 16-byte loads per thread per page (1 or 2)

    __shared__ __attribute__((aligned(16))) uint8_t s_k[2][PAGE_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t s_v[2][PAGE_BYTES];
    __shared__ __attribute__((aligned(16))) _Float16 s_meta[2][4][PAGE_TOK];   // k scale, k zero, v scale, v zero
    __shared__ __attribute__((aligned(16))) _Float16 s_q[G][DH];               // rotated q
    __shared__ __attribute__((aligned(16))) _Float16 s_knew[DH];
    __shared__ float s_sc[G][PAGE_TOK];                                        // scores, then probabilities
    __shared__ float s_alpha[G];
    __shared__ float s_cur[G];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hkv = blockIdx.x, b = blockIdx.y;
    const int tl = lengths ? lengths[b] - 1 : timestep;          // tlength, Template.hpp:901
    if (tl < 0) return;
    const int64_t* ktab = kv_pointers + (size_t)b * 2 * max_blocks;
    const int64_t* vtab = ktab + max_blocks;
    const float inv_sqrt = 0.08838834764831845f;               // 1/sqrt(128)

    // ---- phase A: RoPE of the G query heads and of k; quantise + store the new token's K and V --------------
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
    }
    __syncthreads();
    {
        const int blk = tl >> 6, slot = tl & 63;
        if (wave == 0) {
            uint8_t* pg = reinterpret_cast<uint8_t*>(ktab[blk]);
            __half* sc = reinterpret_cast<__half*>(pg + (size_t)num_kv_heads * PAGE_TOK * DHB);
            wave_quant_store<INT4>(s_knew[2 * lane], s_knew[2 * lane + 1],
                                   pg + ((size_t)hkv * PAGE_TOK + slot) * DHB, sc + hkv * PAGE_TOK + slot,
                                   sc + num_kv_heads * PAGE_TOK + hkv * PAGE_TOK + slot, lane);
        } else if (wave == 1) {
            uint8_t* pg = reinterpret_cast<uint8_t*>(vtab[blk]);
            __half* sc = reinterpret_cast<__half*>(pg + (size_t)num_kv_heads * PAGE_TOK * DHB);
            wave_quant_store<INT4>(vb[2 * lane], vb[2 * lane + 1], pg + ((size_t)hkv * PAGE_TOK + slot) * DHB,
                                   sc + hkv * PAGE_TOK + slot, sc + num_kv_heads * PAGE_TOK + hkv * PAGE_TOK + slot,
                                   lane);
        } else {
            // waves 2,3: q.k of the new token in fp32 (Template.hpp:1356-1364); wave w-2 takes heads w-2, w, ...
            for (int h = wave - 2; h < G; h += 2) {
                float d = (float)s_q[h][lane] * (float)s_knew[lane] + (float)s_q[h][64 + lane] * (float)s_knew[64 + lane];
                d = wave_sum(d);
                if (lane == 0) s_cur[h] = d * inv_sqrt;
            }
        }
    }

    // ---- per-thread q fragments for the QK phase: thread (tok = tid>>2, qd = tid&3) covers dims 32qd..32qd+31 ----
    const int qd = tid & 3, qtok = tid >> 2;
    h2 qf[G][16];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d0 = 32 * qd + 8 * w;
                if (INT4) qf[h][w * 4 + r] = (h2){s_q[h][d0 + r], s_q[h][d0 + r + 4]};   // nibble order e_r, e_{r+4}
                else qf[h][w * 4 + r] = (h2){s_q[h][d0 + 2 * r], s_q[h][d0 + 2 * r + 1]};
            }

    // PV mapping: thread (dg = tid&15 -> dims 8dg..8dg+7, ts = tid>>4 -> tokens ts, ts+16, ts+32, ts+48)
    const int dg = tid & 15, ts = tid >> 4;
    float acc[G][8];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[h][j] = 0.f;
    constexpr int HPW = (G + 3) / 4;       // heads per wave in the softmax phase (wave w owns heads w, w+4, ...)
    float m_run[HPW], l_run[HPW];          // maintained redundantly by every lane of the owning wave
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) {
        m_run[hh] = -3.0e38f;
        l_run[hh] = 0.f;
    }

    const int npages = (tl + PAGE_TOK - 1) >> 6;   // pages holding tokens < tl
    uint4 rk[NLD], rv[NLD];
    _Float16 rmeta = (_Float16)0.f;
    auto issue = [&](int p) {
        const PageAddr ka = page_addr<DHB>(ktab[p], hkv, num_kv_heads);
        const PageAddr va = page_addr<DHB>(vtab[p], hkv, num_kv_heads);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            rk[i] = reinterpret_cast<const uint4*>(ka.data)[tid + i * TPB];
            rv[i] = reinterpret_cast<const uint4*>(va.data)[tid + i * TPB];
        }
        const __half* src = wave == 0 ? ka.scale : wave == 1 ? ka.zero : wave == 2 ? va.scale : va.zero;
        rmeta = __builtin_bit_cast(_Float16, src[lane]);
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            reinterpret_cast<uint4*>(s_k[buf])[tid + i * TPB] = rk[i];
            reinterpret_cast<uint4*>(s_v[buf])[tid + i * TPB] = rv[i];
        }
        s_meta[buf][wave][lane] = rmeta;
    };
    if (npages > 0) {
        issue(0);
        commit(0);
    }
    __syncthreads();

    for (int p = 0; p < npages; ++p) {
        const int buf = p & 1;
        if (p + 1 < npages) issue(p + 1);
        const int valid = min(PAGE_TOK, tl - p * PAGE_TOK);

        // ---------------- QK ----------------
        {
            float sc[G];
#pragma unroll
            for (int h = 0; h < G; ++h) sc[h] = 0.f;
            const float ksc = (float)s_meta[buf][0][qtok], kzr = (float)s_meta[buf][1][qtok];
            if (INT4) {
                const uint4 raw = *reinterpret_cast<const uint4*>(&s_k[buf][qtok * DHB + qd * 16]);
                const _Float16 hs = (_Float16)ksc, hz = (_Float16)(-ksc * kzr);
                const h2 vs = {hs, hs}, vz = {hz, hz};
                const u32 rw[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    h2 kk[4];
                    nib8_to_h2(rw[w], kk);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const h2 kd = __builtin_elementwise_fma(kk[r], vs, vz);
#pragma unroll
                        for (int h = 0; h < G; ++h) sc[h] = __builtin_amdgcn_fdot2(kd, qf[h][w * 4 + r], sc[h], false);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint4 raw = *reinterpret_cast<const uint4*>(&s_k[buf][qtok * DHB + qd * 32 + i * 16]);
                    const u32 rw[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                    for (int w = 0; w < 4; ++w)
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const float f0 = ksc * ((float)((rw[w] >> (16 * r)) & 0xFFu) - kzr);      // Utils.h:2104
                            const float f1 = ksc * ((float)((rw[w] >> (16 * r + 8)) & 0xFFu) - kzr);
                            const h2 kd = {(_Float16)f0, (_Float16)f1};
#pragma unroll
                            for (int h = 0; h < G; ++h)
                                sc[h] = __builtin_amdgcn_fdot2(kd, qf[h][(i * 2 + (w >> 1)) * 4 + (w & 1) * 2 + r], sc[h], false);
                        }
                }
            }
#pragma unroll
            for (int h = 0; h < G; ++h) {
                float t = sc[h];
                t += __shfl_xor(t, 1, 64);
                t += __shfl_xor(t, 2, 64);
                if (qd == 0) s_sc[h][qtok] = qtok < valid ? t * inv_sqrt : -3.0e38f;
            }
        }
        __syncthreads();

        // ---------------- online softmax: wave w owns heads w, w+4, ... ----------------
#pragma unroll
        for (int hh = 0; hh < HPW; ++hh) {
            const int h = wave + 4 * hh;
            if (h < G) {
                const float s = s_sc[h][lane];
                const float m_new = fmaxf(m_run[hh], wave_max(s));
                const float pexp = lane < valid ? __expf(s - m_new) : 0.f;
                const float alpha = __expf(m_run[hh] - m_new);
                l_run[hh] = l_run[hh] * alpha + wave_sum(pexp);
                m_run[hh] = m_new;
                s_sc[h][lane] = pexp;
                if (lane == 0) s_alpha[h] = alpha;
            }
        }
        __syncthreads();

        // ---------------- PV ----------------
        {
            float al[G];
#pragma unroll
            for (int h = 0; h < G; ++h) {
                al[h] = s_alpha[h];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[h][j] *= al[h];
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int tok = ts + 16 * jj;
                // tokens >= valid may hold uninitialised bytes (NaN scales): force their dequantised value to 0
                const float vsc = tok < valid ? (float)s_meta[buf][2][tok] : 0.f;
                const float vzr = tok < valid ? (float)s_meta[buf][3][tok] : 0.f;
                float vf[8];
                if (INT4) {
                    const u32 raw = *reinterpret_cast<const u32*>(&s_v[buf][tok * DHB + dg * 4]);
                    const _Float16 hs = (_Float16)vsc, hz = (_Float16)(-vsc * vzr);
                    const h2 vs2 = {hs, hs}, vz2 = {hz, hz};
                    h2 vv[4];
                    nib8_to_h2(raw, vv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const h2 d = __builtin_elementwise_fma(vv[r], vs2, vz2);
                        vf[r] = (float)d[0];
                        vf[r + 4] = (float)d[1];
                    }
                } else {
                    const uint2 raw = *reinterpret_cast<const uint2*>(&s_v[buf][tok * DHB + dg * 8]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        vf[j] = (float)(_Float16)(vsc * ((float)((raw.x >> (8 * j)) & 0xFFu) - vzr));
                        vf[4 + j] = (float)(_Float16)(vsc * ((float)((raw.y >> (8 * j)) & 0xFFu) - vzr));
                    }
                }
#pragma unroll
                for (int h = 0; h < G; ++h) {
                    const float pr = s_sc[h][tok];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[h][j] = fmaf(pr, vf[j], acc[h][j]);
                }
            }
        }
        __syncthreads();
        if (p + 1 < npages) commit(buf ^ 1);
        __syncthreads();
    }

    // ---- finish: merge the new token, reduce the 16 token slots, normalise, store ---------------------------
    // wave w broadcasts (m_run, l_run) of its heads through LDS
    __shared__ float s_m[G], s_l[G];
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) {
        const int h = wave + 4 * hh;
        if (h < G && lane == 0) {
            s_m[h] = m_run[hh];
            s_l[h] = l_run[hh];
        }
    }
    __syncthreads();
    float fac_old[G], p_cur[G], inv_l[G];
#pragma unroll
    for (int h = 0; h < G; ++h) {
        const float mf = fmaxf(s_m[h], s_cur[h]);
        fac_old[h] = __expf(s_m[h] - mf);
        p_cur[h] = __expf(s_cur[h] - mf);
        inv_l[h] = 1.0f / (s_l[h] * fac_old[h] + p_cur[h] + 1.e-6f);   // Template.hpp:1819
    }
    // 16 token slots -> 1 through LDS: partials stored [h][j][tid] (conflict-free), summed by one thread per output
    __shared__ float s_part[G][8][TPB];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) s_part[h][j][tid] = acc[h][j];
    __syncthreads();
    // G heads x 8 x 16 dim groups outputs; thread o = (h*8 + j)*16 + odg reads 16 consecutive-bank partials
    for (int o = tid; o < G * DH; o += TPB) {
        const int odg = o & 15, j = (o >> 4) & 7, h = o >> 7;
        const int d = 8 * odg + j;
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) sum += s_part[h][j][s * 16 + odg];
        const float vn = (float)vb[d];
        const float r = (sum * fac_old[h] + p_cur[h] * vn) * inv_l[h];
        out[((size_t)b * num_heads + (size_t)hkv * G + h) * DH + d] = (_Float16)r;
    }
}

// ---------------------------------------------------------------------------------------------------------
// prefill KV writer: one wave per (token, q-head group); RoPE in place + quantised page write
// ---------------------------------------------------------------------------------------------------------
template <bool INT4>
__global__ __launch_bounds__(TPB) void prefill_kv_kernel(_Float16* __restrict__ qkv, const int* __restrict__ seq_lens,
                                                         const int* __restrict__ padding_offset,
                                                         const int64_t* __restrict__ kv_pointers, int num_tokens,
                                                         int max_blocks, int head_num, int kv_head_num, int seq_len,
                                                         float rope_base) {
    constexpr int DHB = INT4 ? DH / 2 : DH;
    // grid.x = token, block = 4 waves; wave w loops over heads w, w+4, ... of the H q heads followed by Hkv "kv jobs"
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = t + (padding_offset ? padding_offset[t] : 0);
    const int b = g / seq_len, pos = g % seq_len;          // applyBias...h:186-194
    if (pos >= seq_lens[b]) return;
    const int n = (head_num + 2 * kv_head_num) * DH;
    _Float16* row = qkv + (size_t)t * n;
    const RopeCS cs = rope_coef(lane, pos, rope_base, DH);
    for (int job = wave; job < head_num + kv_head_num; job += 4) {
        if (job < head_num) {
            _Float16* qh = row + job * DH;
            _Float16 a, bb;
            rope_pair((float)qh[lane], (float)qh[64 + lane], cs, a, bb);
            qh[lane] = a;
            qh[64 + lane] = bb;
        } else {
            const int hk = job - head_num;
            _Float16* kh = row + head_num * DH + hk * DH;
            const _Float16* vh = row + (head_num + kv_head_num) * DH + hk * DH;
            _Float16 a, bb;
            rope_pair((float)kh[lane], (float)kh[64 + lane], cs, a, bb);
            kh[lane] = a;                                   // STORE_QKV, :386-388
            kh[64 + lane] = bb;
            if (kv_pointers) {
                // lane holds rotated (k[lane], k[64+lane]); the quantiser wants (k[2*lane], k[2*lane+1])
                const float fa = (float)a, fb = (float)bb;
                const int s0 = (2 * lane) & 63, s1 = (2 * lane + 1) & 63;
                const float a0 = __shfl(fa, s0, 64), b0 = __shfl(fb, s0, 64);
                const float a1 = __shfl(fa, s1, 64), b1 = __shfl(fb, s1, 64);
                const _Float16 k0 = (_Float16)(lane < 32 ? a0 : b0), k1 = (_Float16)(lane < 32 ? a1 : b1);
                const int blk = pos >> 6, slot = pos & 63;
                const int64_t* tab = kv_pointers + (size_t)b * 2 * max_blocks;
                uint8_t* kp = reinterpret_cast<uint8_t*>(tab[blk]);
                uint8_t* vp = reinterpret_cast<uint8_t*>(tab[max_blocks + blk]);
                __half* ksc = reinterpret_cast<__half*>(kp + (size_t)kv_head_num * PAGE_TOK * DHB);
                __half* vsc = reinterpret_cast<__half*>(vp + (size_t)kv_head_num * PAGE_TOK * DHB);
                wave_quant_store<INT4>(k0, k1, kp + ((size_t)hk * PAGE_TOK + slot) * DHB, ksc + hk * PAGE_TOK + slot,
                                       ksc + kv_head_num * PAGE_TOK + hk * PAGE_TOK + slot, lane);
                wave_quant_store<INT4>(vh[2 * lane], vh[2 * lane + 1], vp + ((size_t)hk * PAGE_TOK + slot) * DHB,
                                       vsc + hk * PAGE_TOK + slot,
                                       vsc + kv_head_num * PAGE_TOK + hk * PAGE_TOK + slot, lane);
            }
        }
    }
}

// Vectorised form (used whenever the library's RoPE table covers the sequence): 8 lanes per head, every lane owns dims
// 8 dg .. 8 dg + 7 of the low half and the same dims of the high half (a NeoX pair is (d, 64 + d)), i.e. two 16-byte
// accesses per head and direction instead of 2-byte ones, cos / sin from the table (the same double-evaluated,
// float-rounded values as rope_coef), min / max over the 8 lanes of a head group.  Identical arithmetic per element.
template <bool INT4>
__device__ __forceinline__ void group_quant_store(const h8& lo, const h8& hi, uint8_t* dst, __half* scale_p, __half* zero_p,
                                                  int dg) {
    float mx = -INFINITY, mn = INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mx = fmaxf(mx, fmaxf((float)lo[j], (float)hi[j]));
        mn = fminf(mn, fminf((float)lo[j], (float)hi[j]));
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        mn = fminf(mn, __shfl_xor(mn, m, 64));
    }
    const QParams p = make_qparams<INT4>(mn, mx);
    unsigned ul[8], uh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ul[j] = quant_u8(lo[j], p);
        uh[j] = quant_u8(hi[j], p);
    }
    if (INT4) {                                            // byte i = dims (2i, 2i+1): Utils.h:1838-1852
        u32 wl = 0, wh = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            wl |= (((ul[2 * m] & 0xFu) | ((ul[2 * m + 1] & 0xFu) << 4)) << (8 * m));
            wh |= (((uh[2 * m] & 0xFu) | ((uh[2 * m + 1] & 0xFu) << 4)) << (8 * m));
        }
        *reinterpret_cast<u32*>(dst + 4 * dg) = wl;
        *reinterpret_cast<u32*>(dst + 32 + 4 * dg) = wh;
    } else {
        u32 a0 = 0, a1 = 0, b0 = 0, b1 = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            a0 |= ul[m] << (8 * m);
            a1 |= ul[4 + m] << (8 * m);
            b0 |= uh[m] << (8 * m);
            b1 |= uh[4 + m] << (8 * m);
        }
        *reinterpret_cast<uint2*>(dst + 8 * dg) = make_uint2(a0, a1);
        *reinterpret_cast<uint2*>(dst + 64 + 8 * dg) = make_uint2(b0, b1);
    }
    if (dg == 0) {
        *scale_p = __builtin_bit_cast(__half, p.scale);
        *zero_p = __builtin_bit_cast(__half, p.zero);
    }
}

template <bool INT4>
__global__ __launch_bounds__(TPB) void prefill_kv_vec_kernel(_Float16* __restrict__ qkv, const int* __restrict__ seq_lens,
                                                             const int* __restrict__ padding_offset,
                                                             const int64_t* __restrict__ kv_pointers, int num_tokens,
                                                             int max_blocks, int head_num, int kv_head_num, int seq_len,
                                                             const float2* __restrict__ rope_tab) {
    constexpr int DHB = INT4 ? DH / 2 : DH;
    const int t = blockIdx.x;
    const int g = t + (padding_offset ? padding_offset[t] : 0);
    const int b = g / seq_len, pos = g % seq_len;          // applyBias...h:186-194
    if (pos >= seq_lens[b]) return;
    const int dg = threadIdx.x & 7, slot0 = threadIdx.x >> 3;   // TPB / 8 head slots per pass
    const int n = (head_num + 2 * kv_head_num) * DH;
    _Float16* row = qkv + (size_t)t * n;
    RopeCS cs[8];
    {
        const float4* tp = reinterpret_cast<const float4*>(rope_tab + (size_t)pos * 64 + 8 * dg);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 v = tp[j];
            cs[2 * j].c = v.x, cs[2 * j].s = v.y, cs[2 * j + 1].c = v.z, cs[2 * j + 1].s = v.w;
        }
    }
    for (int job = slot0; job < head_num + kv_head_num; job += TPB / 8) {
        _Float16* hp = row + job * DH;                      // q heads, then k heads: contiguous in the row
        const h8 lo = *reinterpret_cast<const h8*>(hp + 8 * dg), hi = *reinterpret_cast<const h8*>(hp + 64 + 8 * dg);
        h8 rl, rh;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 a, bb;
            rope_pair((float)lo[j], (float)hi[j], cs[j], a, bb);
            rl[j] = a;
            rh[j] = bb;
        }
        *reinterpret_cast<h8*>(hp + 8 * dg) = rl;           // STORE_QKV, :386-388
        *reinterpret_cast<h8*>(hp + 64 + 8 * dg) = rh;
        if (job >= head_num && kv_pointers) {
            const int hk = job - head_num;
            const _Float16* vh = row + (head_num + kv_head_num) * DH + hk * DH;
            const h8 vl = *reinterpret_cast<const h8*>(vh + 8 * dg), vhh = *reinterpret_cast<const h8*>(vh + 64 + 8 * dg);
            const int blk = pos >> 6, slot = pos & 63;
            const int64_t* tab = kv_pointers + (size_t)b * 2 * max_blocks;
            uint8_t* kp = reinterpret_cast<uint8_t*>(tab[blk]);
            uint8_t* vp = reinterpret_cast<uint8_t*>(tab[max_blocks + blk]);
            __half* ksc = reinterpret_cast<__half*>(kp + (size_t)kv_head_num * PAGE_TOK * DHB);
            __half* vsc = reinterpret_cast<__half*>(vp + (size_t)kv_head_num * PAGE_TOK * DHB);
            group_quant_store<INT4>(rl, rh, kp + ((size_t)hk * PAGE_TOK + slot) * DHB, ksc + hk * PAGE_TOK + slot,
                                    ksc + kv_head_num * PAGE_TOK + hk * PAGE_TOK + slot, dg);
            group_quant_store<INT4>(vl, vhh, vp + ((size_t)hk * PAGE_TOK + slot) * DHB, vsc + hk * PAGE_TOK + slot,
                                    vsc + kv_head_num * PAGE_TOK + hk * PAGE_TOK + slot, dg);
        }
    }
}

__global__ void padding_offsets_kernel(int* __restrict__ out, const int* __restrict__ cu, int max_seqlen) {
    const int b = blockIdx.x;
    const int beg = cu[b], end = cu[b + 1];
    const int off = b * max_seqlen - beg;                  // input_metadata_helper.cu:24
    for (int i = threadIdx.x; i < end - beg; i += blockDim.x) out[beg + i] = off;
}

template <bool INT4>
int launch_decode(int G, dim3 grid, hipStream_t st, const _Float16* q, const _Float16* k, const _Float16* v,
                  const int64_t* kvp, const int* len, _Float16* out, int H, int Hkv, int64_t qs, int64_t kvs, int mb,
                  int timestep, float base) {
    if (G < 1 || G > 8) {
        qs_set_error("single_query_attention: num_heads/num_kv_heads = %d not in 1..8", G);
        return QS_ENOSUP;
    }
    if (g_qs_attn_plan.active) {
        g_qs_attn_plan.family = 3, g_qs_attn_plan.nsplit = 1, g_qs_attn_plan.waves = TPB / 64;
        return QS_OK;
    }
#define QS_LAUNCH_G(GG)                                                                                            \
    hipLaunchKernelGGL((decode_attention_kernel<GG, INT4>), grid, dim3(TPB), 0, st, q, k, v, kvp, len, out, H, Hkv, \
                       qs, kvs, mb, timestep, base)
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
    return qs_launch_status("single_query_attention");
}

}  // namespace

// library-managed RoPE table (attention_mfma.hip)
const float2* qs_rope_table(float base, int max_pos, hipStream_t st, int* len_out);
// KV4 fast path on the matrix cores (attention_mfma.hip)
int qs_launch_decode_mfma(int G, dim3 grid, hipStream_t st, const _Float16* q, const _Float16* k, const _Float16* v,
                          const int64_t* kvp, const int* len, _Float16* out, int H, int Hkv, int64_t qs, int64_t kvs,
                          int mb, int timestep, float base, int max_pos, int force_split, int kflags);
// KV8 twin (attention_mfma8.hip)
int qs_launch_decode_mfma8(int G, dim3 grid, hipStream_t st, const _Float16* q, const _Float16* k, const _Float16* v,
                           const int64_t* kvp, const int* len, _Float16* out, int H, int Hkv, int64_t qs, int64_t kvs,
                           int mb, int timestep, float base, int max_pos, int force_split);
// 0 = MFMA kernel for KV4 with the split-KV heuristic (default), 1 = VALU kernel everywhere, 2 = prefill writer in its
// per-lane form (no RoPE table), 3 = KV4 MFMA kernel whose service wave always owns pages (the round-2 form: A/B),
// 100 + n = MFMA kernel with exactly n KV splits (A/B tests)
static qs_flag g_attn_variant = 0;
extern "C" void qs_set_attention_variant(int variant) { g_attn_variant = variant; }

extern "C" int qs_single_query_attention(const void* q, const void* k, const void* v, const int64_t* kv_pointers,
                                         const int32_t* length_per_sample, void* out, int batch, int num_heads,
                                         int num_kv_heads, int head_dim, int64_t q_stride0, int64_t kv_stride0,
                                         int max_blocks, int memory_max_seqlen, int tokens_per_block,
                                         int size_per_token, int timestep, int rotary_embedding_dim, float rotary_base,
                                         int neox_rotary_style, int int4_kv_cache, int kv_cache_with_zeros,
                                         qs_stream_t stream) {
    QS_REQUIRE(q && k && v && kv_pointers && out, "single_query_attention: null pointer");
    QS_REQUIRE(batch >= 0 && num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0,
               "single_query_attention: bad head counts H=%d Hkv=%d", num_heads, num_kv_heads);
    // neox_rotary_style is accepted and has NO effect, as in the reference: set_params leaves the field unset
    // (fused_attention.cpp:109, commented out) and the kernel's only rotary branch is the NeoX one
    // (decoderMaskedMultiheadAttentionTemplate.hpp:1136-1215: the GPT-J case is commented out).
    (void)neox_rotary_style;
    if (head_dim != 128 || rotary_embedding_dim != 128 || tokens_per_block != 64 || !kv_cache_with_zeros) {
        qs_set_error("single_query_attention: only head_dim=128, rotary_dim=128, tokens_per_block=64 and "
                     "zero-point KV caches are supported (the variants the reference instantiates)");
        return QS_ENOSUP;
    }
    const int dhb = int4_kv_cache ? 64 : 128;
    QS_REQUIRE(size_per_token == num_kv_heads * dhb, "single_query_attention: size_per_token=%d, expected %d",
               size_per_token, num_kv_heads * dhb);
    QS_REQUIRE(max_blocks > 0 && memory_max_seqlen > 0, "single_query_attention: bad max_blocks / memory_max_seqlen");
    if (batch == 0) return QS_OK;
    dim3 grid(num_kv_heads, batch);
    const int G = num_heads / num_kv_heads;
    hipStream_t st = (hipStream_t)stream;
    // the matrix-core kernel caches a sequence's page addresses in LDS (192 pages = 12288 tokens per sequence); longer
    // page tables take the VALU kernel
    if (int4_kv_cache && g_attn_variant != 1 && max_blocks <= 192)
        return qs_launch_decode_mfma(G, grid, st, (const _Float16*)q, (const _Float16*)k, (const _Float16*)v,
                                     kv_pointers, length_per_sample, (_Float16*)out, num_heads, num_kv_heads,
                                     q_stride0, kv_stride0, max_blocks, timestep, rotary_base, memory_max_seqlen,
                                     g_attn_variant >= 100 ? g_attn_variant - 100 : 0, g_attn_variant == 5 ? 2 : g_attn_variant == 7 ? 1024 : 0);
    if (!int4_kv_cache && g_attn_variant != 1 && max_blocks <= 192)
        return qs_launch_decode_mfma8(G, grid, st, (const _Float16*)q, (const _Float16*)k, (const _Float16*)v,
                                      kv_pointers, length_per_sample, (_Float16*)out, num_heads, num_kv_heads,
                                      q_stride0, kv_stride0, max_blocks, timestep, rotary_base, memory_max_seqlen,
                                      g_attn_variant >= 100 ? g_attn_variant - 100 : 0);
    if (int4_kv_cache)
        return launch_decode<true>(G, grid, st, (const _Float16*)q, (const _Float16*)k, (const _Float16*)v,
                                   kv_pointers, length_per_sample, (_Float16*)out, num_heads, num_kv_heads, q_stride0,
                                   kv_stride0, max_blocks, timestep, rotary_base);
    return launch_decode<false>(G, grid, st, (const _Float16*)q, (const _Float16*)k, (const _Float16*)v, kv_pointers,
                                length_per_sample, (_Float16*)out, num_heads, num_kv_heads, q_stride0, kv_stride0,
                                max_blocks, timestep, rotary_base);
}

thread_local QsAttnPlan g_qs_attn_plan = {0, 0, 0, 0};
thread_local QsAttnQuant g_qs_attn_quant = {nullptr, nullptr, nullptr, 0};

// single_query_attention followed by invoke_quant(_fuse_sum) of its output, as ONE call (decode loop of
// llama_w4a8_unpad.py:253-282: attention, reshape, self.invoke_quant).  Results are bit-identical to the two calls; where
// the chosen attention kernel can finish the row itself (matrix-core KV4 kernel, no KV split, rows <= 4096) the row
// kernel and a kernel boundary disappear, otherwise the two launches are issued here.
extern "C" int qs_single_query_attention_quant(const void* q, const void* k, const void* v, const int64_t* kv_pointers,
                                               const int32_t* length_per_sample, void* out, int8_t* quant_out,
                                               void* quant_sum, void* quant_scale, int batch, int num_heads,
                                               int num_kv_heads, int head_dim, int64_t q_stride0, int64_t kv_stride0,
                                               int max_blocks, int memory_max_seqlen, int tokens_per_block,
                                               int size_per_token, int timestep, int rotary_embedding_dim,
                                               float rotary_base, int neox_rotary_style, int int4_kv_cache,
                                               int kv_cache_with_zeros, qs_stream_t stream) {
    QS_REQUIRE(quant_out && quant_scale, "single_query_attention_quant: null pointer");
    g_qs_attn_quant = {quant_out, quant_scale, quant_sum, 0};
    const int rc = qs_single_query_attention(q, k, v, kv_pointers, length_per_sample, out, batch, num_heads, num_kv_heads,
                                             head_dim, q_stride0, kv_stride0, max_blocks, memory_max_seqlen,
                                             tokens_per_block, size_per_token, timestep, rotary_embedding_dim, rotary_base,
                                             neox_rotary_style, int4_kv_cache, kv_cache_with_zeros, stream);
    const int done = g_qs_attn_quant.done;
    g_qs_attn_quant = {nullptr, nullptr, nullptr, 0};
    if (rc != QS_OK || done || batch == 0) return rc;
    return qs_invoke_quant(quant_out, out, quant_sum, quant_scale, batch, num_heads * head_dim, stream);
}
extern "C" int qs_attention_plan(int batch, int num_heads, int num_kv_heads, int max_blocks, int timestep,
                                 int int4_kv_cache, int* plan3) {
    QS_REQUIRE(plan3, "attention plan: null output");
    void* d = reinterpret_cast<void*>(uintptr_t(256));          // never dereferenced in plan-only mode
    g_qs_attn_plan = {1, 0, 0, 0};
    const int rc = qs_single_query_attention(d, d, d, reinterpret_cast<const int64_t*>(d), nullptr, d, batch, num_heads,
                                             num_kv_heads, 128, (int64_t)(num_heads + 2 * num_kv_heads) * 128,
                                             (int64_t)(num_heads + 2 * num_kv_heads) * 128, max_blocks,
                                             max_blocks * 64, 64, num_kv_heads * (int4_kv_cache ? 64 : 128), timestep,
                                             128, 10000.f, 1, int4_kv_cache, 1, nullptr);
    plan3[0] = g_qs_attn_plan.family;
    plan3[1] = g_qs_attn_plan.nsplit;
    plan3[2] = g_qs_attn_plan.waves;
    g_qs_attn_plan.active = 0;
    return rc;
}

extern "C" int qs_apply_bias_rope_update_kv_cache(void* qkv, const int32_t* seq_lens, const int32_t* padding_offset,
                                                  const int64_t* kv_pointers, int num_tokens, int batch,
                                                  int max_blocks, int head_num, int kv_head_num, int seq_len,
                                                  int tokens_per_block, int size_per_token, int rotary_embedding_dim,
                                                  float rotary_embedding_base, int rotary_embedding_max_positions,
                                                  int neox_rotary_style, int int4_kv_cache, int kv_cache_with_zeros,
                                                  qs_stream_t stream) {
    QS_REQUIRE(qkv && seq_lens, "apply_bias_rope_update_kv_cache: null pointer");
    QS_REQUIRE(head_num > 0 && kv_head_num > 0 && seq_len > 0 && batch >= 0,
               "apply_bias_rope_update_kv_cache: bad sizes");
    // neox_rotary_style: accepted, no effect - the reference hard-codes PositionEmbeddingType::kROPE_GPT_NEOX whatever the
    // flag says (update_kv_cache.cu:57)
    (void)neox_rotary_style;
    if (rotary_embedding_dim != 128 || tokens_per_block != 64 || !kv_cache_with_zeros) {
        qs_set_error("apply_bias_rope_update_kv_cache: only head_dim=128, tokens_per_block=64 and "
                     "zero-point KV caches are supported");
        return QS_ENOSUP;
    }
    if (kv_pointers) {
        const int dhb = int4_kv_cache ? 64 : 128;
        QS_REQUIRE(size_per_token == kv_head_num * dhb && max_blocks > 0,
                   "apply_bias_rope_update_kv_cache: size_per_token=%d, expected %d", size_per_token,
                   kv_head_num * dhb);
    }
    (void)rotary_embedding_max_positions;
    if (num_tokens <= 0) return QS_OK;
    hipStream_t st = (hipStream_t)stream;
    // the vectorised writer reads cos / sin from the library's table (positions < seq_len); without one (first call inside
    // a stream capture, all table slots taken) the per-lane form computes them in the kernel
    int tab_len = 0;
    const float2* tab = g_attn_variant == 2 ? nullptr : qs_rope_table(rotary_embedding_base, seq_len, st, &tab_len);
    if (tab && tab_len >= seq_len) {
        if (int4_kv_cache)
            hipLaunchKernelGGL(prefill_kv_vec_kernel<true>, dim3(num_tokens), dim3(TPB), 0, st, (_Float16*)qkv, seq_lens,
                               padding_offset, kv_pointers, num_tokens, max_blocks, head_num, kv_head_num, seq_len, tab);
        else
            hipLaunchKernelGGL(prefill_kv_vec_kernel<false>, dim3(num_tokens), dim3(TPB), 0, st, (_Float16*)qkv, seq_lens,
                               padding_offset, kv_pointers, num_tokens, max_blocks, head_num, kv_head_num, seq_len, tab);
        return qs_launch_status("apply_bias_rope_update_kv_cache");
    }
    if (int4_kv_cache)
        hipLaunchKernelGGL(prefill_kv_kernel<true>, dim3(num_tokens), dim3(TPB), 0, st, (_Float16*)qkv, seq_lens,
                           padding_offset, kv_pointers, num_tokens, max_blocks, head_num, kv_head_num, seq_len,
                           rotary_embedding_base);
    else
        hipLaunchKernelGGL(prefill_kv_kernel<false>, dim3(num_tokens), dim3(TPB), 0, st, (_Float16*)qkv, seq_lens,
                           padding_offset, kv_pointers, num_tokens, max_blocks, head_num, kv_head_num, seq_len,
                           rotary_embedding_base);
    return qs_launch_status("apply_bias_rope_update_kv_cache");
}

extern "C" int qs_compute_padding_offsets(int32_t* padding_offsets, const int32_t* cu_seqlens, int batch,
                                          int max_seqlen, qs_stream_t stream) {
    QS_REQUIRE(padding_offsets && cu_seqlens, "compute_padding_offsets: null pointer");
    if (batch <= 0) return QS_OK;
    hipLaunchKernelGGL(padding_offsets_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, padding_offsets,
                       cu_seqlens, max_seqlen);
    return qs_launch_status("compute_padding_offsets");
}
