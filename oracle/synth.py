"""Seeded synthetic inputs for the hot path.  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import numpy as np

from . import w4a8


def per_channel_problem(M, N, K, seed=0):
    """A ~ U{-127..127}; Q ~ U{0..15}; z ~ U{0..15}; s1 ~ U(0.002, 0.02); ascales ~ U(0.005, 0.05);
    a_ssums = half(ascale * sum_k A) (consistent with a real quantiser)."""
    r = np.random.default_rng(seed)
    A = r.integers(-127, 128, (M, K), dtype=np.int8)
    q = r.integers(0, 16, (N, K), dtype=np.uint8)
    z = r.integers(0, 16, (N,), dtype=np.uint8)
    s1 = r.uniform(0.002, 0.02, N).astype(np.float16)
    qweight, wscales, w_szs = w4a8.pack_per_channel(q, z, s1)
    ascales = r.uniform(0.005, 0.05, M).astype(np.float16)
    a_ssums = (ascales.astype(np.float32) * A.astype(np.int64).sum(axis=1).astype(np.float32)).astype(np.float16)
    return dict(A=A, q=q, z=z, qweight=qweight, wscales=wscales, w_szs=w_szs, ascales=ascales, a_ssums=a_ssums)


def per_group_problem(M, N, K, seed=0, G=128, valid=True):
    """QoQ-style two-level weights inside the protective range (q*s2 <= 255, (q-z)*s2 in [-128,127]).
    valid=False draws s2 up to 40 so that byte products overflow (exercises the wrap/carry semantics)."""
    r = np.random.default_rng(seed)
    A = r.integers(-127, 128, (M, K), dtype=np.int8)
    ng = K // G
    if valid:
        w8 = r.integers(-119, 120, (N, ng, G)).astype(np.int32)
        mx, mn = w8.max(axis=2), w8.min(axis=2)
        s2 = np.maximum(1, np.ceil((mx - mn) / 15.0)).astype(np.int32)
        z = np.clip(np.rint(-mn / s2), 0, 15).astype(np.int32)
        q = np.clip(np.rint(w8 / s2[..., None]) + z[..., None], 0, 15).astype(np.int32)
        # enforce the protective range exactly
        lo = np.clip(np.ceil(-128.0 / s2 + z), 0, 15).astype(np.int32)
        hi = np.minimum(np.clip(np.floor(127.0 / s2 + z), 0, 15), 255 // s2).astype(np.int32)
        q = np.minimum(np.maximum(q, lo[..., None]), hi[..., None])
        assert (q * s2[..., None]).max() <= 255
        assert ((q - z[..., None]) * s2[..., None]).min() >= -128 and ((q - z[..., None]) * s2[..., None]).max() <= 127
    else:
        s2 = r.integers(1, 41, (N, ng)).astype(np.int32)
        z = r.integers(0, 16, (N, ng)).astype(np.int32)
        q = r.integers(0, 16, (N, ng, G)).astype(np.int32)
    s1 = r.uniform(0.002, 0.02, N).astype(np.float16)
    qweight, wscales, s2_scales, s2_zeros = w4a8.pack_per_group(q.reshape(N, K), z, s2, s1)
    ascales = r.uniform(0.005, 0.05, M).astype(np.float16)
    return dict(A=A, q=q.reshape(N, K), z=z, s2=s2, qweight=qweight, wscales=wscales, s2_scales=s2_scales,
                s2_zeros=s2_zeros, ascales=ascales)


def attention_problem(B, H, Hkv, lengths, seed=0, Dh=128, extra_blocks=3, shuffle=True):
    """Random block tables (a permutation of the pool), q/k/v ~ N(0,1) fp16 for the NEW token, and the fp16
    history (K/V sources ~ N(0,1)) that the prefill writer will quantise.  `lengths` includes the new token."""
    r = np.random.default_rng(seed)
    lengths = np.asarray(lengths, np.int32)
    max_len = int(lengths.max())
    mb = (max_len + 63) // 64
    nblocks = B * mb + extra_blocks
    perm = r.permutation(nblocks) if shuffle else np.arange(nblocks)
    tables = np.zeros((B, 2, mb), np.int64)
    # K and V pools are separate tensors in the reference; use independent permutations
    permv = r.permutation(nblocks) if shuffle else np.arange(nblocks)
    for b in range(B):
        tables[b, 0] = perm[b * mb:(b + 1) * mb]
        tables[b, 1] = permv[b * mb:(b + 1) * mb]
    q = r.standard_normal((B, H, Dh)).astype(np.float16)
    k = r.standard_normal((B, Hkv, Dh)).astype(np.float16)
    v = r.standard_normal((B, Hkv, Dh)).astype(np.float16)
    hist = [r.standard_normal((int(l) - 1, (H + 2 * Hkv) * Dh)).astype(np.float16) for l in lengths]
    return dict(q=q, k=k, v=v, tables=tables, lengths=lengths, nblocks=nblocks, max_blocks=mb, hist=hist)
