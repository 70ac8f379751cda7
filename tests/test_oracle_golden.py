"""Pin the oracle: packer against golden vectors produced by the reference's own Python (tests/golden/),
and internal consistency of every oracle piece (CPU only)."""
import glob
import os

import numpy as np
import pytest

from oracle import fused, kvattn, synth, w4a8

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "w4a8_pack_per_chn_*.npz"))))
def test_pack_per_channel_matches_reference(path):
    d = np.load(path)
    qw, s1, sz = w4a8.pack_per_channel(d["q"], d["z"], d["s1"])
    assert np.array_equal(qw, d["qweight"])
    assert np.array_equal(s1.view(np.uint16), d["s1_scales"].view(np.uint16))
    assert np.array_equal(sz.view(np.uint16), d["s1_szeros"].view(np.uint16))
    assert np.array_equal(w4a8.unpack_qweight(d["qweight"]), d["q"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "w4a8_pack_per_group_*.npz"))))
def test_pack_per_group_matches_reference(path):
    d = np.load(path)
    qw, s1, s2s, s2z = w4a8.pack_per_group(d["q"], d["z"], d["s2"], d["s1"])
    assert np.array_equal(qw, d["qweight"])
    assert np.array_equal(s2s, d["s2_scales"])
    assert np.array_equal(s2z, d["s2_zeros"])
    # level-2 dequant of the REFERENCE's tensors gives (q - z) * s2
    w8 = w4a8.dequant_per_group_w8(d["qweight"], d["s2_zeros"], d["s2_scales"]).astype(int)
    ref = (d["q"].astype(int) - np.repeat(d["z"].astype(int), 128, 1)) * np.repeat(d["s2"].astype(int), 128, 1)
    assert np.array_equal(w8, ref)


def _full_size_inputs(kind, N, K, seed):
    """Same seeded recipe as tests/golden/make_golden.py::full_size_inputs (kept in sync by the digest test below)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    if kind == "per_chn":
        q = torch.randint(0, 16, (N, K), generator=g)
        z = torch.randint(0, 16, (N,), generator=g)
        s1 = (torch.rand((N,), generator=g) * 0.018 + 0.002).to(torch.float16)
        return dict(q=q.numpy(), z=z.numpy(), s1=s1.numpy())
    ng = K // 128
    s2 = torch.randint(1, 9, (N, ng), generator=g)
    z = torch.randint(0, 16, (N, ng), generator=g)
    q = torch.randint(0, 16, (N, ng, 128), generator=g)
    lo = torch.ceil((-128.0 / s2.float()) + z.float()).clamp(0, 15).long()
    hi = torch.floor((127.0 / s2.float()) + z.float()).clamp(0, 15).long()
    q = torch.maximum(torch.minimum(q, hi[..., None]), lo[..., None])
    s1 = (torch.rand((N,), generator=g) * 0.018 + 0.002).to(torch.float16)
    return dict(q=q.reshape(N, K).numpy(), z=z.numpy(), s2=s2.numpy(), s1=s1.numpy())


def full_size_reference_pinned(kind):
    """Oracle-packed Llama-3-8B o_proj-sized checkpoint tensors, verified byte-for-byte (SHA-256) against what the
    reference's own from_linear produced for the same seeded inputs (tests/golden/make_golden.py).  Returns the raw
    inputs and the packed tensors; used by the CPU pin below and by the GPU parity tests."""
    import hashlib
    import json
    meta = json.load(open(os.path.join(GOLD, "w4a8_pack_llama3_8b_o_proj_digests.json")))
    N, K = meta["shape"]
    ent = meta["entries"][kind]
    i = _full_size_inputs(kind, N, K, ent["seed"])
    if kind == "per_chn":
        qw, s1, sz = w4a8.pack_per_channel(i["q"], i["z"], i["s1"])
        packed = dict(qweight=qw, s1_scales=s1, s1_szeros=sz)
    else:
        qw, s1, s2s, s2z = w4a8.pack_per_group(i["q"], i["z"], i["s2"], i["s1"])
        packed = dict(qweight=qw, s1_scales=s1, s2_scales=s2s, s2_zeros=s2z)
    for name, t in packed.items():
        got = hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()
        assert got == ent[name], f"{kind}.{name}: oracle packer output differs from the reference's from_linear"
    return i, packed


@pytest.mark.parametrize("kind", ["per_chn", "per_group"])
def test_pack_full_size_matches_reference_digest(kind):
    """4096 x 4096 (Llama-3-8B o_proj): the oracle packer reproduces the reference's from_linear byte for byte."""
    full_size_reference_pinned(kind)


def test_golden_files_present():
    assert len(glob.glob(os.path.join(GOLD, "*.npz"))) >= 4


@pytest.mark.parametrize("N,K", [(32, 32), (64, 128), (160, 96)])
def test_pack_unpack_roundtrip(N, K):
    q = np.random.default_rng(N + K).integers(0, 16, (N, K), dtype=np.uint8)
    assert np.array_equal(w4a8.unpack_qweight(w4a8.pack_qweight(q)), q)


def test_pack_layout_hand_checked():
    # byte t=d*8+b*4+f of lane l=c*4+e in tile (n32,k32): lo = W[8b+c][16d+4e+f], hi = row + 16
    q = np.random.default_rng(5).integers(0, 16, (64, 64), dtype=np.uint8)
    p = w4a8.pack_qweight(q).view(np.uint8).reshape(2, 2, 32, 16)
    for n32, k32, c, e, d, b, f in [(0, 0, 0, 0, 0, 0, 0), (1, 1, 7, 3, 1, 1, 3), (0, 1, 3, 2, 1, 0, 1), (1, 0, 5, 1, 0, 1, 2)]:
        byte = p[n32, k32, c * 4 + e, d * 8 + b * 4 + f]
        r, col = 32 * n32 + 8 * b + c, 32 * k32 + 16 * d + 4 * e + f
        assert byte & 0xF == q[r, col] and byte >> 4 == q[r + 16, col]


def test_group_meta_permutation():
    x = np.arange(64)[None, :]
    p = w4a8.permute_group_meta(x)
    # storage index c*4+j holds channel j*8+c
    for c in range(8):
        for j in range(4):
            assert p[0, c * 4 + j] == j * 8 + c and p[0, 32 + c * 4 + j] == 32 + j * 8 + c
    assert np.array_equal(w4a8.unpermute_group_meta(p), x)


def test_per_channel_gemm_definition():
    pr = synth.per_channel_problem(13, 64, 256, seed=1)
    acc, out = w4a8.gemm_per_chn(pr["A"], pr["qweight"], pr["wscales"], pr["ascales"], pr["w_szs"], pr["a_ssums"])
    ref = pr["A"].astype(np.int64) @ pr["q"].astype(np.int64).T
    assert np.array_equal(acc, ref)
    # the epilogue equals the real-valued linear layer x_fp @ W_fp^T up to fp16 rounding of scales/sums
    x = pr["A"].astype(np.float64) * pr["ascales"].astype(np.float64)[:, None]
    w = (pr["q"].astype(np.float64) - pr["z"].astype(np.float64)[:, None]) * pr["wscales"].astype(np.float64)[:, None]
    assert np.allclose(out.astype(np.float64), x @ w.T, rtol=5e-3, atol=0.15)


def test_per_group_gemm_valid_and_wrapping():
    pr = synth.per_group_problem(9, 64, 256, seed=2)
    acc, out = w4a8.gemm_per_group(pr["A"], pr["qweight"], pr["s2_zeros"], pr["s2_scales"], pr["wscales"], pr["ascales"])
    w8 = (pr["q"].astype(np.int64) - np.repeat(pr["z"], 128, 1)) * np.repeat(pr["s2"], 128, 1)
    assert np.array_equal(acc, pr["A"].astype(np.int64) @ w8.T)
    # outside the protective range the packed-byte arithmetic wraps / carries; per-element model of the first byte
    bad = synth.per_group_problem(4, 32, 128, seed=3, valid=False)
    w8b = w4a8.dequant_per_group_w8(bad["qweight"], bad["s2_zeros"], bad["s2_scales"])
    q, z, s2 = bad["q"].astype(np.int64), np.repeat(bad["z"], 128, 1), np.repeat(bad["s2"], 128, 1)
    k0 = np.arange(0, 128, 4)   # byte 0 of every 32-bit word never receives a carry
    exp0 = ((q[:, k0] * s2[:, k0]) + ((-z[:, k0] * s2[:, k0]) & 0xFF)) & 0xFF
    assert np.array_equal(w8b[:, k0].view(np.uint8), exp0.astype(np.uint8))


def test_int_matmul_exact_on_large_k():
    r = np.random.default_rng(0)
    A = r.integers(-128, 128, (3, 20000), dtype=np.int8)
    W = r.integers(-128, 128, (5, 20000), dtype=np.int8)
    assert np.array_equal(w4a8.int_matmul(A, W), A.astype(np.int64) @ W.astype(np.int64).T)


# ---------------------------------------------------------------------------------------------- KV cache
@pytest.mark.parametrize("int4", [True, False])
def test_kv_quant_roundtrip_error_bound(int4):
    x = np.random.default_rng(1).standard_normal((50, 128)).astype(np.float16)
    b, s, z = kvattn.kv_quantize(x, int4)
    assert b.shape == (50, 64 if int4 else 128)
    for mode in ("kernel", "fp32"):
        d = kvattn.kv_dequantize(b, s, z, int4, mode).astype(np.float32)
        step = s.astype(np.float32)[:, None]
        assert np.all(np.abs(d - x.astype(np.float32)) <= 0.75 * step + 2e-2)


def test_kv4_nibble_wrap_quirk():
    # value that rounds to 16 wraps to 0: max element hits 15.x only through fp16 zero rounding,
    # so emulate directly on the helper
    assert kvattn.rni_sat_u8(np.float32(15.5)) == 16 and (16 & 0xF) == 0
    assert kvattn.rni_sat_u8(np.float32(-3.0)) == 0 and kvattn.rni_sat_u8(np.float32(300.0)) == 255
    assert kvattn.rni_sat_u8(np.float32(2.5)) == 2 and kvattn.rni_sat_u8(np.float32(3.5)) == 4   # ties to even


def test_page_layout_matches_cache_engine_formula():
    # cache_engine.py:60-66 with Llama-3-8B: 8 heads * 64 tok * 128 dims / 2 + 64*8*4 = 34816 ; KV8 = 67584
    assert kvattn.page_bytes(8, 128, True) == 34816
    assert kvattn.page_bytes(8, 128, False) == 67584
    pool = kvattn.PagePool(3, 8, 128, True)
    assert pool.scale_off == 8 * 64 * 64 and pool.zero_off == pool.scale_off + 8 * 64 * 2


def test_rope_is_rotation_and_position_zero_identity():
    x = np.random.default_rng(2).standard_normal((4, 128)).astype(np.float16)
    assert np.array_equal(kvattn.rope_neox(x, 0, 5e5), x)
    y = kvattn.rope_neox(x, 1234, 5e5).astype(np.float32)
    n0 = np.linalg.norm(x.astype(np.float32), axis=1)
    assert np.allclose(np.linalg.norm(y, axis=1), n0, rtol=2e-3)


def test_padding_offsets():
    cu = np.array([0, 3, 3, 10], np.int32)
    off = kvattn.compute_padding_offsets(cu, 8, 10)
    assert off.tolist() == [0, 0, 0] + [13] * 7   # seq 2 starts at token 3 -> 2*8 - 3


@pytest.mark.parametrize("int4", [True, False])
def test_prefill_then_decode_modes_agree(int4):
    H, Hkv, B = 4, 2, 3
    pr = synth.attention_problem(B, H, Hkv, [1, 70, 130], seed=4)
    pool = kvattn.PagePool(pr["nblocks"], Hkv, 128, int4)
    hist = np.concatenate(pr["hist"])
    seq = (pr["lengths"] - 1).astype(np.int32)
    cu = np.concatenate([[0], np.cumsum(seq)]).astype(np.int32)
    pad = kvattn.compute_padding_offsets(cu, int(seq.max()), hist.shape[0])
    qkv = hist.copy()
    kvattn.prefill_update_kv_cache(qkv, seq, pad, pr["tables"], pool, H, Hkv, int(seq.max()), 5e5)
    # rotated q/k written back, v untouched
    assert not np.array_equal(qkv[:, :H * 128], hist[:, :H * 128]) or hist.shape[0] == 0
    assert np.array_equal(qkv[:, (H + Hkv) * 128:], hist[:, (H + Hkv) * 128:])
    import copy
    p1, p2 = copy.deepcopy(pool), copy.deepcopy(pool)
    o1 = kvattn.decode_attention(pr["q"], pr["k"], pr["v"], pr["tables"], pr["lengths"], p1, 5e5, "kernel")
    o2 = kvattn.decode_attention(pr["q"], pr["k"], pr["v"], pr["tables"], pr["lengths"], p2, 5e5, "fp32")
    assert np.array_equal(p1.k, p2.k) and np.array_equal(p1.v, p2.v)
    assert np.max(np.abs(o1.astype(np.float32) - o2.astype(np.float32))) < 1e-3
    # first sequence has only the new token: output == v (softmax of one element, /(1+1e-6))
    g = H // Hkv
    for h in range(H):
        assert np.allclose(o2[0, h].astype(np.float32), pr["v"][0, h // g].astype(np.float32), atol=2e-3)


# ---------------------------------------------------------------------------------------------- fused
def test_quant_per_token():
    x = (np.random.default_rng(3).standard_normal((7, 256)) * 3).astype(np.float16)
    q, s, sm, pre = fused.quant_per_token(x, with_sum=True)
    assert np.abs(q.astype(np.int32)).max() == 127
    assert np.allclose(q.astype(np.float32) * s.astype(np.float32)[:, None], x.astype(np.float32), atol=0.05)
    assert np.allclose(sm.astype(np.float32), x.astype(np.float32).sum(1), rtol=2e-3, atol=2e-2)


def test_general_norm_is_mean_subtracting():
    x = (np.random.default_rng(4).standard_normal((5, 512)) + 2.0).astype(np.float16)
    g = np.ones(512, np.float16)
    q, s, sm, pre = fused.rms_norm_general(x, g, 1e-6, with_sum=True)
    deq = q.astype(np.float32) * s.astype(np.float32)[:, None]
    assert abs(deq.mean()) < 0.05 and abs(deq.std() - 1.0) < 0.05      # LayerNorm, not RMSNorm (appendix B.1)
    assert np.all(np.abs(sm.astype(np.float32)) < 1.0)


def test_silu_and_rms_norm_shapes():
    x = np.random.default_rng(5).standard_normal((3, 64)).astype(np.float16)
    assert fused.silu_and_mul(x).shape == (3, 32)
    assert fused.rms_norm(x, np.ones(64, np.float16), 1e-5).shape == (3, 64)


def test_torch_cpu_baseline_matches_exact_oracle():
    """bench.py's cpu_baseline (oracle/torch_cpu.py, the PyTorch-CPU dequant + matmul / attention path of BASELINE.md 3)
    computes the same GEMM and the same attention as the exact oracle, up to fp32 rounding."""
    import torch
    from oracle import torch_cpu as T
    t = torch.from_numpy
    pc = synth.per_channel_problem(5, 64, 256, seed=1)
    out = T.linear_per_channel(t(pc["A"]), t(pc["qweight"]), t(pc["wscales"]), t(pc["ascales"]), t(pc["w_szs"])).numpy()
    Wd = pc["q"].astype(np.float64) * pc["wscales"].astype(np.float64)[:, None] - pc["w_szs"].astype(np.float64)[:, None]
    assert np.abs(out - (pc["A"].astype(np.float64) * pc["ascales"].astype(np.float64)[:, None]) @ Wd.T).max() < 1e-4
    pg = synth.per_group_problem(5, 64, 256, seed=2)
    out = T.linear_per_group(t(pg["A"]), t(pg["qweight"]), t(pg["s2_zeros"]), t(pg["s2_scales"]), t(pg["wscales"]),
                             t(pg["ascales"])).numpy()
    w8 = w4a8.dequant_per_group_w8(pg["qweight"], pg["s2_zeros"], pg["s2_scales"]).astype(np.float64)
    ref = (pg["A"].astype(np.float64) * pg["ascales"].astype(np.float64)[:, None]) @ (w8 * pg["wscales"].astype(np.float64)[:, None]).T
    assert np.abs(out - ref).max() < 1e-4
    for int4 in (True, False):
        r = np.random.default_rng(3)
        B, H, Hkv, L = 2, 4, 2, 100
        pool = kvattn.PagePool(6, Hkv, 128, int4)
        tables = np.stack([r.permutation(6)[:4].reshape(2, 2), r.permutation(6)[:4].reshape(2, 2)], 1)
        for b in range(B):
            for pos in range(L):
                for which in ("k", "v"):
                    x = r.standard_normal((Hkv, 128)).astype(np.float16)
                    qb, sc, zr = kvattn.kv_quantize(x, int4)
                    for h in range(Hkv):
                        pool.write_token(which, int(tables[b, 0 if which == "k" else 1, pos // 64]), pos % 64, h, qb[h], sc[h], zr[h])
        q = r.standard_normal((B, H, 128)).astype(np.float32)
        got = T.decode_attention(t(q), t(pool.k), t(pool.v), t(tables.astype(np.int64)), L, Hkv, int4).numpy()
        for b in range(B):
            for hk in range(Hkv):
                kq, ks, kz = pool.read_tokens("k", tables[b, 0], hk, L)
                vq, vs, vz = pool.read_tokens("v", tables[b, 1], hk, L)
                Kd = kvattn.kv_dequantize(kq, ks, kz, int4, "exact").astype(np.float64)
                Vd = kvattn.kv_dequantize(vq, vs, vz, int4, "exact").astype(np.float64)
                for g in range(H // Hkv):
                    s = Kd @ q[b, hk * 2 + g].astype(np.float64) / np.sqrt(128.0)
                    p = np.exp(s - s.max())
                    ref = (p / p.sum()) @ Vd
                    assert np.abs(got[b, hk * 2 + g] - ref).max() < 1e-4


def test_oracle_fmaf_is_correctly_rounded():
    """oracle.w4a8._fmaf (the fma = True / "sub" epilogue conventions) against exact rational arithmetic, including results
    that sit exactly on a float32 rounding boundary before the residual is taken into account."""
    from fractions import Fraction

    import numpy as np

    from oracle import w4a8
    r = np.random.default_rng(3)
    a = (r.standard_normal(3000) * 1000).astype(np.float32)
    b = r.standard_normal(3000).astype(np.float32)
    c = (-(a.astype(np.float64) * b) + r.standard_normal(3000) * 1e-3).astype(np.float32)     # cancellation: many low bits matter
    # constructed ties: a * b = 1 + 2^-24 exactly (midpoint of two float32), c = +-2^-60 decides
    a[:2] = np.float32(1 + 2.0 ** -12)
    b[:2] = np.float32((1 + 2.0 ** -24) / (1 + 2.0 ** -12))
    c[0], c[1] = np.float32(2.0 ** -60), np.float32(-2.0 ** -60)
    got = w4a8._fmaf(a, b, c)
    for i in range(a.size):
        ex = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
        g = np.float32(got[i])
        dg = abs(ex - Fraction(float(g)))
        for nb in (np.nextafter(g, np.float32(-np.inf)), np.nextafter(g, np.float32(np.inf))):
            assert abs(ex - Fraction(float(nb))) >= dg, (i, float(a[i]), float(b[i]), float(c[i]), float(g))
