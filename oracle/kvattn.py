"""Oracle: paged quantised KV cache (KV4 / KV8, asymmetric), prefill KV writer, decode attention.

TEST INFRASTRUCTURE (see oracle/__init__.py).  numpy only.  PARITY UNPINNED by reference tests
(none exist); restated from the kernel sources:

  * page layout / addressing ............ kernels/csrc/fused_attention/kvCacheUtils.h:47-126,
                                          qserve/worker/cache_engine.py:59-66
  * RoPE (NeoX) coefficients ............ decoderMaskedMultiheadAttentionUtils.h:1147-1160
  * decode: new-token handling .......... decoderMaskedMultiheadAttentionTemplate.hpp:1045-1082 (V scale/zero),
                                          :1158-1217 (RoPE), :1221-1258 (K scale/zero), :1310-1348 (K store),
                                          :1356-1400 (q.k of the new token)
  * decode: key loop / softmax / values . ...Template.hpp:1474-1624, :1764-1845, :1901-1977, :2123-2221
  * quantise (store) .................... ...Utils.h:1687-1697, 1838-1884, 2045-2077
  * dequantise .......................... ...Utils.h:2095-2107 (KV8, fp32), 2125-2213 (KV4, fp16 hfma2)
  * prefill writer ...................... applyBiasRopeUpdateKVCache.h:94-455, ...Utils.h:2536-2613
  * padding offsets ..................... input_metadata_helper.cu:11-31

Arithmetic conventions chosen where the reference is ambiguous (nvcc FMA contraction, fast-math
libm): RoPE coefficients use float32 `pow`, `cos`, `sin` of numpy; `a*b+c` written in one C
expression is evaluated as a fused multiply-add ONLY for the quantiser `x*inv_scale + zero`
(nvcc --fmad default) and as separate roundings everywhere else.  The HIP kernels follow the same
choices; they are stated in DESIGN.md.
"""
import numpy as np

TOKENS_PER_BLOCK = 64  # hard-coded at the call sites, llama_w4a8_unpad.py:211,256


# --------------------------------------------------------------------------------------------------
# page pool (host-side model of CacheEngine, cache_engine.py:59-115)
# --------------------------------------------------------------------------------------------------
def page_bytes(num_kv_heads, head_dim, int4):
    """cache_engine.py:60-66: data bytes + fp16 scale + fp16 zero per (head, token)."""
    data = num_kv_heads * TOKENS_PER_BLOCK * head_dim // (2 if int4 else 1)
    return data + TOKENS_PER_BLOCK * num_kv_heads * 4


class PagePool:
    """One layer's K pool and V pool: uint8 [num_blocks, page_bytes] each (cache_engine.py:100-114)."""

    def __init__(self, num_blocks, num_kv_heads, head_dim, int4, fill=0):
        self.hkv, self.dh, self.int4 = num_kv_heads, head_dim, int4
        self.dhb = head_dim // 2 if int4 else head_dim  # bytes per head per token
        self.pb = page_bytes(num_kv_heads, head_dim, int4)
        self.k = np.full((num_blocks, self.pb), fill, np.uint8)
        self.v = np.full((num_blocks, self.pb), fill, np.uint8)
        self.scale_off = num_kv_heads * TOKENS_PER_BLOCK * self.dhb  # kvCacheUtils.h:75 mBytesPerSeq
        self.zero_off = self.scale_off + num_kv_heads * TOKENS_PER_BLOCK * 2

    # kvCacheUtils.h:117-125 getKVLocalIdx; scales [Hkv][64] then zeros [Hkv][64]
    def _views(self, pool, block):
        page = pool[block]
        data = page[: self.scale_off].reshape(self.hkv, TOKENS_PER_BLOCK, self.dhb)
        sc = page[self.scale_off: self.zero_off].view(np.float16).reshape(self.hkv, TOKENS_PER_BLOCK)
        zr = page[self.zero_off:].view(np.float16).reshape(self.hkv, TOKENS_PER_BLOCK)
        return data, sc, zr

    def write_token(self, which, block, tok_in_block, head, qbytes, scale, zero):
        data, sc, zr = self._views(self.k if which == "k" else self.v, block)
        data[head, tok_in_block] = qbytes
        sc[head, tok_in_block] = scale
        zr[head, tok_in_block] = zero

    def read_tokens(self, which, block_table_row, head, length):
        """Gather tokens 0..length-1 of one sequence/head -> (bytes [L, dhb], scale f16 [L], zero f16 [L])."""
        pool = self.k if which == "k" else self.v
        nb = (length + TOKENS_PER_BLOCK - 1) // TOKENS_PER_BLOCK
        ds, ss, zs = [], [], []
        for b in range(nb):
            data, sc, zr = self._views(pool, block_table_row[b])
            ds.append(data[head]); ss.append(sc[head]); zs.append(zr[head])
        if nb == 0:
            return (np.zeros((0, self.dhb), np.uint8), np.zeros(0, np.float16), np.zeros(0, np.float16))
        return (np.concatenate(ds)[:length], np.concatenate(ss)[:length], np.concatenate(zs)[:length])


# --------------------------------------------------------------------------------------------------
# scalar helpers
# --------------------------------------------------------------------------------------------------
def _f32(x):
    return np.asarray(x, np.float32)


def rni_sat_u8(x):
    """cvt.rni.sat.u8.f32 (Utils.h:1687-1697): round to nearest even, saturate to 0..255; NaN -> 0."""
    x = _f32(x)
    r = np.rint(np.nan_to_num(x, nan=0.0, posinf=255.0, neginf=0.0))
    return np.clip(r, 0, 255).astype(np.uint8)


def rope_coef(pos, dim, base):
    """(cos, sin) float32 [dim/2] for rotary pair i, Utils.h:1147-1152:
    inv_freq = pos / pow(base, (2i) / (float)dim) all in float32."""
    i2 = (np.arange(dim // 2, dtype=np.float32) * np.float32(2.0))
    expo = (i2 / np.float32(dim)).astype(np.float32)
    # pow / cos / sin: evaluated in float64 on the float32 inputs and rounded once to float32 (= correctly rounded
    # float32 results; the reference's --use_fast_math __powf/__cosf/__sinf are not reproducible off-device)
    denom = np.power(np.float64(np.float32(base)), expo.astype(np.float64)).astype(np.float32)
    ang = (np.float32(pos) / denom).astype(np.float32)
    return np.cos(ang.astype(np.float64)).astype(np.float32), np.sin(ang.astype(np.float64)).astype(np.float32)


def rope_neox(x, pos, base):
    """NeoX rotary on fp16 x[..., dim]: pair (i, i+dim/2); fp32 math without contraction, round to fp16.
    Utils.h:1154-1167 (decode) and :2536-2557 (prefill) compute the same two expressions."""
    x = np.asarray(x, np.float16)
    dim = x.shape[-1]
    c, s = rope_coef(pos, dim, base)
    a = x[..., : dim // 2].astype(np.float32)
    b = x[..., dim // 2:].astype(np.float32)
    ra = ((c * a).astype(np.float32) - (s * b).astype(np.float32)).astype(np.float32)
    rb = ((c * b).astype(np.float32) + (s * a).astype(np.float32)).astype(np.float32)
    return np.concatenate([ra, rb], axis=-1).astype(np.float16)


def kv_scale_zero(x, int4):
    """Per-(token, head) asymmetric parameters, Template.hpp:1051-1082 / applyBias...h:288-331.
    x fp16 [..., dim] -> (scale f16, zero f16, inv_scale f32)."""
    xf = np.asarray(x, np.float16).astype(np.float32)
    mx = xf.max(axis=-1)
    mn = xf.min(axis=-1)
    levels = np.float32(15.0 if int4 else 255.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        rng = (mx - mn).astype(np.float32)
        scale = (rng / levels).astype(np.float32).astype(np.float16)
        zero = ((-levels * mn).astype(np.float32) / rng).astype(np.float32).astype(np.float16)
        inv = (np.float32(1.0) / scale.astype(np.float32)).astype(np.float32)
    return scale, zero, inv


def kv_quantize(x, int4):
    """fp16 x[..., dim] -> (bytes uint8 [..., dim/2 or dim], scale f16, zero f16).
    u = rni_sat_u8(fma(x, inv_scale, zero)); KV4 packs (u_even & 0xF) | (u_odd << 4) (Utils.h:1838-1852:
    saturation is to 255, then the low nibble is kept -> 16 wraps to 0)."""
    scale, zero, inv = kv_scale_zero(x, int4)
    xf = np.asarray(x, np.float16).astype(np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        t = (xf * inv[..., None].astype(np.float64) + zero[..., None].astype(np.float64)).astype(np.float32)
    u = rni_sat_u8(t)
    if int4:
        lo = u[..., 0::2] & 0xF
        hi = (u[..., 1::2].astype(np.uint16) << 4).astype(np.uint8)  # int8[2] << 4 truncated to 8 bits
        u = lo | hi
    return u, scale, zero


def _hfma(a, b, c):
    """fma.rn.f16: single rounding of a*b+c to fp16."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float16)


def kv_dequantize(qbytes, scale, zero, int4, mode="kernel"):
    """bytes [..., L, dhb] + scale/zero f16 [..., L] -> fp16 [..., L, dim] in NATURAL element order.

    KV4 (Utils.h:2125-2213): nibble -> exact fp16 integer (magic-number trick, exact), then
        h = hfma2(h, half_rn(scale), half_rn(-scale*zero))   [scale, zero promoted to fp32 first]
    KV8 (Utils.h:2095-2107): half_rn( scale * (float(u8) - zero) ) in fp32.
    mode="fp32": plain fp32 `scale*(q-zero)` for both, rounded to fp16 (strict reference-free variant);
    mode="exact": the same value NOT rounded to fp16 (float32 result): the mathematical de-quantisation."""
    q = np.asarray(qbytes, np.uint8)
    sc = np.asarray(scale, np.float16)
    zr = np.asarray(zero, np.float16)
    if int4:
        vals = np.empty(q.shape[:-1] + (q.shape[-1] * 2,), np.uint8)
        vals[..., 0::2] = q & 0xF
        vals[..., 1::2] = q >> 4
    else:
        vals = q
    scf = sc.astype(np.float32)[..., None]
    zrf = zr.astype(np.float32)[..., None]
    if mode == "exact":
        return (scf * (vals.astype(np.float32) - zrf).astype(np.float32)).astype(np.float32)
    if mode == "fp32" or not int4:
        out = (scf * (vals.astype(np.float32) - zrf).astype(np.float32)).astype(np.float32)
        return out.astype(np.float16)
    hs = sc[..., None]
    hz = ((-scf) * zrf).astype(np.float32).astype(np.float16)
    return _hfma(vals.astype(np.float16), hs, hz)


# --------------------------------------------------------------------------------------------------
# prefill: apply_bias_rope_update_kv_cache + compute_padding_offsets
# --------------------------------------------------------------------------------------------------
def compute_padding_offsets(cu_seqlens, max_seqlen, tot):
    """input_metadata_helper.cu:11-31."""
    cu = np.asarray(cu_seqlens, np.int64)
    out = np.zeros(tot, np.int32)
    for b in range(len(cu) - 1):
        out[cu[b]:cu[b + 1]] = b * max_seqlen - cu[b]
    return out


def prefill_update_kv_cache(qkv, seq_lens, padding_offset, block_tables, pool, num_heads, num_kv_heads,
                            max_seq_len, rope_base, head_dim=128):
    """In-place on `qkv` (fp16 [T, (H+2Hkv)*Dh]) and `pool`.  applyBiasRopeUpdateKVCache.h:178-454 with
    the arguments `update_kv_cache.cu:41-90` passes (no bias, NeoX, STORE_QKV, kv_seq_lens = seq_lens)."""
    T = qkv.shape[0]
    H, Hkv, Dh = num_heads, num_kv_heads, head_dim
    q_sz, kv_sz = H * Dh, Hkv * Dh
    for t in range(T):
        g = t + int(padding_offset[t])
        b, pos = g // max_seq_len, g % max_seq_len      # token_idx_in_seq, :186-194
        if pos >= int(seq_lens[b]):
            continue
        row = qkv[t]
        q = row[:q_sz].reshape(H, Dh)
        k = row[q_sz:q_sz + kv_sz].reshape(Hkv, Dh)
        v = row[q_sz + kv_sz:].reshape(Hkv, Dh)
        q[:] = rope_neox(q, pos, rope_base)
        k[:] = rope_neox(k, pos, rope_base)             # written back (STORE_QKV :386-388)
        kb, ks, kz = kv_quantize(k, pool.int4)
        vb, vs, vz = kv_quantize(v, pool.int4)
        blk_k = int(block_tables[b, 0, pos // TOKENS_PER_BLOCK])
        blk_v = int(block_tables[b, 1, pos // TOKENS_PER_BLOCK])
        for h in range(Hkv):
            pool.write_token("k", blk_k, pos % TOKENS_PER_BLOCK, h, kb[h], ks[h], kz[h])
            pool.write_token("v", blk_v, pos % TOKENS_PER_BLOCK, h, vb[h], vs[h], vz[h])


# --------------------------------------------------------------------------------------------------
# decode: single_query_attention
# --------------------------------------------------------------------------------------------------
def decode_attention(q, k, v, block_tables, lengths, pool, rope_base, mode="kernel"):
    """q fp16 [B,H,Dh], k/v fp16 [B,Hkv,Dh] (new token, un-rotated), block_tables int [B,2,maxb] of BLOCK
    INDICES into `pool`, lengths int [B] = context length INCLUDING the new token.
    Mutates `pool` (new token's quantised K/V) and returns out fp16 [B,H,Dh].

    mode="kernel": the reference's precisions - fp16 dequant (KV4) / fp32 dequant (KV8), fp16x2 partial
        products `mul + 3 fma` over the 8 dims one thread holds, fp16 add of the two lanes, fp32 sum over
        the 16 threads of a key (Template.hpp:450-467), fp32 softmax with probabilities rounded to fp16
        (:1794-1832), fp32 accumulation of p(fp16)*v(fp16) per 16-token stripe and the final fp16-rounded
        tree reduction over the 16 stripes (:1901-1977, :2163-2187).
    mode="fp32": everything after dequantisation (to fp16 values) in fp64/fp32;
    mode="exact": as "fp32" but the de-quantised cache is kept in float32 (no fp16 rounding of K/V values): the
        mathematical definition of attention over the quantised cache."""
    q = np.asarray(q, np.float16); k = np.asarray(k, np.float16); v = np.asarray(v, np.float16)
    B, H, Dh = q.shape
    Hkv = k.shape[1]
    G = H // Hkv
    out = np.zeros((B, H, Dh), np.float16)
    inv_sqrt = np.float32(1.0 / np.sqrt(np.float32(Dh)))
    for b in range(B):
        tl = int(lengths[b]) - 1                                 # tlength, :901
        qr = rope_neox(q[b], tl, rope_base)                      # [H,Dh]
        kr = rope_neox(k[b], tl, rope_base)                      # [Hkv,Dh]
        kb, ks, kz = kv_quantize(kr, pool.int4)                  # K measured AFTER RoPE, :1221-1258
        vb, vs, vz = kv_quantize(v[b], pool.int4)
        blk_k = int(block_tables[b, 0, tl // TOKENS_PER_BLOCK])
        blk_v = int(block_tables[b, 1, tl // TOKENS_PER_BLOCK])
        for h in range(Hkv):
            pool.write_token("k", blk_k, tl % TOKENS_PER_BLOCK, h, kb[h], ks[h], kz[h])
            pool.write_token("v", blk_v, tl % TOKENS_PER_BLOCK, h, vb[h], vs[h], vz[h])
        for hk in range(Hkv):
            kq, ksc, kzr = pool.read_tokens("k", block_tables[b, 0], hk, tl)
            vq, vsc, vzr = pool.read_tokens("v", block_tables[b, 1], hk, tl)
            Kd = kv_dequantize(kq, ksc, kzr, pool.int4, mode)    # [tl, Dh] fp16
            Vd = kv_dequantize(vq, vsc, vzr, pool.int4, mode)
            for g in range(G):
                h = hk * G + g
                out[b, h] = _attend_one(qr[h], kr[hk], v[b, hk], Kd, Vd, inv_sqrt, mode, pool.int4)
    return out


def _qk_kernel_order(qv, Kd, int4=True):
    """Template.hpp:450-467: thread j (0..15) holds dims 8j..8j+7 as 4 half2 (pairs permuted, irrelevant
    for the sum structure: lane x accumulates 4 products, lane y 4 products).  With the 0,4,1,5,2,6,3,7
    register order (Utils.h:1938-1951) half2 #r of a thread is (d[r], d[r+4]) of its 8 dims."""
    L = Kd.shape[0]
    if int4:
        qd = qv.reshape(16, 2, 4)       # [thread, half2 lane (d0-3 / d4-7), r]
        kd = Kd.reshape(L, 16, 2, 4)
    else:                               # KV8: natural order, half2 #r = (d[2r], d[2r+1])
        qd = qv.reshape(16, 4, 2).transpose(0, 2, 1)
        kd = Kd.reshape(L, 16, 4, 2).transpose(0, 1, 3, 2)
    acc = (qd[None, :, :, 0].astype(np.float32) * kd[..., 0].astype(np.float32)).astype(np.float16)
    for r in range(1, 4):
        acc = _hfma(qd[None, :, :, r], kd[..., r], acc)
    lane = (acc[..., 0].astype(np.float32) + acc[..., 1].astype(np.float32)).astype(np.float16)  # __hadd
    return lane.astype(np.float32).sum(axis=1, dtype=np.float32)   # 16-thread butterfly, fp32


def _attend_one(qv, k_new, v_new, Kd, Vd, inv_sqrt, mode, int4=True):
    L = Kd.shape[0]
    Dh = qv.shape[0]
    # new token: fp32 dot of the fp16 rotated q and k (MMHA_USE_FP32_ACUM_FOR_FMA, :1356-1364), then *1/sqrt(Dh)
    qk_cur = np.float32(np.dot(qv.astype(np.float64), k_new.astype(np.float64))) * inv_sqrt
    if mode == "kernel":
        s = (_qk_kernel_order(qv, Kd, int4) * inv_sqrt).astype(np.float32) if L else np.zeros(0, np.float32)
    else:
        s = (Kd.astype(np.float64) @ qv.astype(np.float64)).astype(np.float32) * inv_sqrt
    s = np.concatenate([s, np.array([qk_cur], np.float32)]).astype(np.float32)
    m = s.max()
    e = np.exp((s - m).astype(np.float32)).astype(np.float32)
    inv_sum = np.float32(1.0) / (e.sum(dtype=np.float32) + np.float32(1e-6))   # :1819
    p = (e * inv_sum).astype(np.float32)
    Vall = np.concatenate([Vd, v_new[None, :]], axis=0)
    if mode == "kernel":
        p16 = p.astype(np.float16)                               # logits stored as fp16, :1831
        # 16 stripes: token ti belongs to stripe ti % 16 (V_PER_ITER = 256/16), fp32 accumulate per stripe
        parts = np.zeros((16, Dh), np.float32)
        for st in range(16):
            idx = np.arange(st, L + 1, 16)
            if len(idx):
                parts[st] = (p16[idx].astype(np.float32)[:, None] * Vall[idx].astype(np.float32)).sum(axis=0, dtype=np.float32)
        # tree reduction: upper half rounded to fp16 in smem, added in fp32 (:2163-2187)
        n = 16
        while n >= 2:
            mid = n // 2
            parts[:mid] = parts[:mid] + parts[mid:n].astype(np.float16).astype(np.float32)
            n = mid
        return parts[0].astype(np.float16)
    return (p.astype(np.float64) @ Vall.astype(np.float64)).astype(np.float16)
