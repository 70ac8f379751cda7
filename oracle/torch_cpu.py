"""CPU baseline: "the reference's PyTorch-CPU linear / attention path" (BASELINE.json north_star, BASELINE.md section 3,
SURVEY.md 8d) - plain torch on the host cores, no custom kernels.  TEST / BENCH INFRASTRUCTURE (see oracle/__init__.py):
only `bench.py`'s `cpu_baseline` leg and the tests import this.

  * W4A8 linear: unpack the reference-layout `qweight` (inverse of w4a8_linear.py:196-226), de-quantise
    ((q - z) * s1 per channel; level-2 (q - z) * s2 then * s1 per group), fp32 `torch.matmul` against the de-quantised
    int8 activations (A * ascale).
  * decode attention: gather the pages through the block table, de-quantise K / V with the stored fp16 scale / zero,
    fp32 softmax(q K^T / sqrt(128)) V per (sequence, head), GQA-expanded.
Checked against the exact oracle (oracle/w4a8.py, oracle/kvattn.py) in tests/test_oracle_golden.py.
"""
import torch


def unpack_qweight(qweight):
    """int8 [N, K/2] (reference layout) -> uint8 [N, K] nibbles."""
    N, K2 = qweight.shape
    K = K2 * 2
    p = qweight.view(torch.uint8).reshape(N // 32, K // 32, 8, 4, 2, 2, 4)   # n32 k32 c e d b f
    lo = (p & 0xF).permute(0, 5, 2, 1, 4, 3, 6)                             # n32 b c k32 d e f
    hi = (p >> 4).permute(0, 5, 2, 1, 4, 3, 6)
    return torch.stack([lo, hi], dim=1).reshape(N, K)


def _unpermute_meta(x):
    ng, N = x.shape
    return x.reshape(ng, N // 32, 8, 4).permute(0, 1, 3, 2).reshape(ng, N)


def dequant_per_channel(qweight, s1_scales, s1_szeros):
    """fp32 [N, K] = q * s1 - z*s1."""
    return unpack_qweight(qweight).float() * s1_scales.float()[:, None] - s1_szeros.float()[:, None]


def dequant_per_group(qweight, s2_zeros, s2_scales, s1_scales):
    """fp32 [N, K] = ((q * s2 + z') as int8) * s1 inside the protective range: (q * s2 + z') computed in int32."""
    q = unpack_qweight(qweight).to(torch.int32)
    N, K = q.shape
    s2 = _unpermute_meta(s2_scales.view(torch.uint8).to(torch.int32)).t()     # [N, K/128]
    z2 = _unpermute_meta(s2_zeros.to(torch.int32)).t()                        # signed: -z*s2
    w8 = q.reshape(N, K // 128, 128) * s2[..., None] + z2[..., None]
    return w8.reshape(N, K).float() * s1_scales.float()[:, None]


def linear(A, ascales, Wdeq):
    """fp32 [M, N] = (A * ascale) @ Wdeq^T."""
    return (A.float() * ascales.float()[:, None]) @ Wdeq.t()


def linear_per_channel(A, qweight, s1_scales, ascales, s1_szeros):
    return linear(A, ascales, dequant_per_channel(qweight, s1_scales, s1_szeros))


def linear_per_group(A, qweight, s2_zeros, s2_scales, s1_scales, ascales):
    return linear(A, ascales, dequant_per_group(qweight, s2_zeros, s2_scales, s1_scales))


def gather_dequant(pool, table, L, num_kv_heads, int4):
    """pool uint8 [nblocks, page_bytes], table int64 [B, mb] block indices -> fp32 [B, Hkv, L, 128]."""
    B, mb = table.shape
    dhb = 64 if int4 else 128
    nd = num_kv_heads * 64 * dhb
    pg = pool[table]                                                          # [B, mb, page_bytes]
    data = pg[..., :nd].reshape(B, mb, num_kv_heads, 64, dhb)
    sc = pg[..., nd:nd + num_kv_heads * 128].contiguous().view(torch.float16).reshape(B, mb, num_kv_heads, 64)
    zr = pg[..., nd + num_kv_heads * 128:].contiguous().view(torch.float16).reshape(B, mb, num_kv_heads, 64)
    if int4:
        vals = torch.stack([(data & 0xF).float(), (data >> 4).float()], dim=-1).reshape(B, mb, num_kv_heads, 64, 128)
    else:
        vals = data.float()
    x = sc.float()[..., None] * (vals - zr.float()[..., None])
    return x.permute(0, 2, 1, 3, 4).reshape(B, num_kv_heads, mb * 64, 128)[:, :, :L]


def decode_attention(q_rot, kpool, vpool, tables, L, num_kv_heads, int4):
    """q_rot fp32/fp16 [B, H, 128] (already rotated), tables int64 [B, 2, mb] block INDICES, all sequences L cached
    tokens -> fp32 [B, H, 128].  (The new token's own term is left out: it is 1 of L+1 keys and does not change the
    cost.)"""
    B, H, _ = q_rot.shape
    G = H // num_kv_heads
    K = gather_dequant(kpool, tables[:, 0], L, num_kv_heads, int4)
    V = gather_dequant(vpool, tables[:, 1], L, num_kv_heads, int4)
    qr = q_rot.float().reshape(B, num_kv_heads, G, 128)
    s = torch.einsum("bkgd,bktd->bkgt", qr, K) / (128 ** 0.5)
    p = torch.softmax(s, dim=-1)
    return torch.einsum("bkgt,bktd->bkgd", p, V).reshape(B, H, 128)
