"""Oracle: W4A8 weight packing and the two W4A8 GEMMs (per-channel, per-group-128).

TEST INFRASTRUCTURE (see oracle/__init__.py).  numpy only.

Reference sources restated here:
  * packing / compute-aware reorder ...... qserve/modeling/layers/quantized_linear/w4a8_linear.py:166-330
  * per-channel kernel, nibble unpack ..... kernels/csrc/qgemm/w4a8_per_chn/gemm_cuda.cu:276-301
  * per-channel epilogue .................. kernels/csrc/qgemm/w4a8_per_chn/gemm_cuda.cu:564-593
  * per-group level-2 dequant ............. kernels/csrc/qgemm/w4a8_per_group/gemm_cuda.cu:271-326
  * per-group epilogue .................... kernels/csrc/qgemm/w4a8_per_group/gemm_cuda.cu:600-627
"""
import numpy as np

GROUP = 128  # constexpr G, w4a8_per_group/gemm_cuda.cu:652


# --------------------------------------------------------------------------------------------------
# packing (w4a8_linear.py:193-226 per-group, :290-322 per-channel; identical byte layout)
# --------------------------------------------------------------------------------------------------
def pack_qweight(q):
    """uint4 values q[N, K] (0..15) -> reference `qweight` int8 [N, K/2].

    Follows w4a8_linear.py:196-226: view rows as (n32, a, b, c) = N/32 x 2 x 2 x 8 and columns as
    (k32, d, e, f) = K/32 x 2 x 4 x 4; the byte stream is ordered (n32, k32, c, e, d, b, f) and each
    byte holds the a=0 value in its low nibble and the a=1 value (row + 16) in its high nibble.
    """
    q = np.asarray(q)
    N, K = q.shape
    assert N % 32 == 0 and K % 32 == 0
    assert q.min() >= 0 and q.max() <= 15
    w = q.astype(np.uint8).reshape(N // 32, 2, 2, 8, K // 32, 2, 4, 4)
    #           axes:          n32 a  b  c  k32    d  e  f
    w = w.transpose(0, 4, 3, 6, 5, 2, 7, 1)  # n32 k32 c e d b f a
    packed = (w[..., 1] << 4) | w[..., 0]
    return np.ascontiguousarray(packed).reshape(N, K // 2).view(np.int8)


def unpack_qweight(qweight, N=None, K=None):
    """Inverse of pack_qweight: int8 [N, K/2] -> uint8 [N, K] with values 0..15."""
    qw = np.asarray(qweight).view(np.uint8)
    if N is None:
        N, K = qw.shape[0], qw.shape[1] * 2
    p = qw.reshape(N // 32, K // 32, 8, 4, 2, 2, 4)  # n32 k32 c e d b f
    out = np.empty((N // 32, 2, 2, 8, K // 32, 2, 4, 4), np.uint8)  # n32 a b c k32 d e f
    lo = (p & 0xF).transpose(0, 5, 2, 1, 4, 3, 6)  # -> n32 b c k32 d e f
    hi = (p >> 4).transpose(0, 5, 2, 1, 4, 3, 6)
    out[:, 0] = lo
    out[:, 1] = hi
    return out.reshape(N, K)


def permute_group_meta(x):
    """[K/G, N] per-(group, channel) values in natural channel order -> reference storage order.

    w4a8_linear.py:231-248 / :255-274: within every 32 channels, storage index c*4+j holds channel j*8+c
    (the order in which one lane of the reference kernel consumes 4 scale bytes, gemm_cuda.cu:294-296).
    """
    x = np.asarray(x)
    ng, N = x.shape
    return np.ascontiguousarray(x.reshape(ng, N // 32, 4, 8).transpose(0, 1, 3, 2)).reshape(ng, N)


def unpermute_group_meta(x):
    x = np.asarray(x)
    ng, N = x.shape
    return np.ascontiguousarray(x.reshape(ng, N // 32, 8, 4).transpose(0, 1, 3, 2)).reshape(ng, N)


def pack_per_channel(q, z, s1):
    """Per-channel checkpoint tensors from q[N,K] in 0..15, zero point z[N], fp16 scale s1[N].

    w4a8_linear.py:322-330: qweight, s1_scales, s1_szeros = z * s1 (fp16 multiply).
    """
    s1 = np.asarray(s1, np.float16)
    szeros = (np.asarray(z).astype(np.float16) * s1).astype(np.float16)  # fp16 * fp16, rn
    return pack_qweight(q), s1.copy(), szeros


def pack_per_group(q, z, s2, s1):
    """Per-group checkpoint tensors from q[N,K] (0..15), z[N,K/G], s2[N,K/G] (int), s1[N] fp16.

    w4a8_linear.py:226-277: qweight; s2_scales = permuted s2 as int8 [K/G, N];
    s2_zeros = (-z) * s2 as int8 two's complement, same permutation.
    """
    z = np.asarray(z).astype(np.int32)
    s2 = np.asarray(s2).astype(np.int32)
    s2_scales = permute_group_meta(s2.T).astype(np.int8)
    s2_zeros = permute_group_meta((-z * s2).T).astype(np.int8)  # values in [-255, 0] wrap mod 256
    return pack_qweight(q), np.asarray(s1, np.float16).copy(), s2_scales, s2_zeros


# --------------------------------------------------------------------------------------------------
# exact integer matmul helper (BLAS sgemm on slices whose sums stay below 2**24)
# --------------------------------------------------------------------------------------------------
def int_matmul(A, W, amax=128, wmax=128):
    """A[M,K] (int) @ W[N,K].T (int) -> int64 [M,N], exact.

    Uses float32 BLAS on K-slices short enough that every partial sum is < 2**24 in magnitude.
    """
    A = np.asarray(A)
    W = np.asarray(W)
    K = A.shape[1]
    step = max(1, (1 << 24) // (amax * wmax) - 1)
    acc = np.zeros((A.shape[0], W.shape[0]), np.int64)
    for k0 in range(0, K, step):
        a = A[:, k0:k0 + step].astype(np.float32)
        w = W[:, k0:k0 + step].astype(np.float32)
        acc += (a @ w.T).astype(np.int64)
    return acc


# --------------------------------------------------------------------------------------------------
# per-channel GEMM
# --------------------------------------------------------------------------------------------------
def gemm_per_chn_acc(A, qweight):
    """int32 accumulator of the per-channel kernel: sum_k A[m,k] * Q[n,k], Q unsigned 0..15.

    w4a8_per_chn/gemm_cuda.cu:291-298 feeds the raw nibbles (no zero-point subtraction) to the s8 MMA.
    """
    Q = unpack_qweight(qweight)
    return int_matmul(np.asarray(A, np.int8), Q, 128, 15).astype(np.int32)


def _fmaf(a, b, c):
    """Correctly rounded float32 fma(a, b, c) for float32 arrays: the product of two float32 is exact in float64, the sum
    with c is made exact by an error-free TwoSum, and the pair (s, e) is rounded to float32 ONCE - the rare cases where s sits
    exactly on a float32 rounding boundary are decided by the sign of the residual e (no double rounding)."""
    p = a.astype(np.float64) * b.astype(np.float64)
    c = np.broadcast_to(c.astype(np.float64), p.shape)
    s = p + c
    bb = s - p
    e = (p - (s - bb)) + (c - bb)                     # s + e == p + c exactly
    r = s.astype(np.float32)
    d = s - r.astype(np.float64)                      # exact: what the float64 -> float32 rounding dropped
    tie = (np.abs(d) * 2 == np.spacing(np.abs(r)).astype(np.float64)) & (e != 0) & np.isfinite(r)
    if tie.any():
        # s is the midpoint of r and its neighbour on the side of d: the exact value s + e lies on the side of e
        toward = np.where(d > 0, np.float32(np.inf), np.float32(-np.inf))
        other = np.nextafter(r, toward.astype(np.float32))
        pick_other = tie & (np.sign(e) == np.sign(d))
        pick_r = tie & (np.sign(e) != np.sign(d))
        r = np.where(pick_other, other, r)
        del pick_r                                    # (r already is the near side)
    # also the non-tie case where e pushes s + e across the midpoint cannot occur: |e| <= ulp64(s) / 2 << ulp32 / 2 - |d|
    return r.astype(np.float32)


def epilogue_per_chn(acc, wscales, ascales, w_szs, a_ssums, fma=False):
    """fp32 epilogue, evaluation order of gemm_cuda.cu:585-588:
        out = half_rn( (float(acc) * wscale[n]) * ascale[m] - w_sz[n] * a_ssum[m] )
    fma = False: every operation rounded separately (no FMA contraction) - the convention the HIP kernels are built to
        (`#pragma clang fp contract(off)`), bit-exact in tests/test_gemm_gpu.py;
    fma = True:  t = acc * wscale ; u = w_sz * a_ssum ; out = fmaf(t, ascale, -u) - what nvcc's default --fmad=true most
        likely makes of line 586 (the last multiply fused with the subtraction);
    fma = "sub": the other legal contraction, fmaf(-w_sz, a_ssum, (acc * wscale) * ascale).
    The reference cannot be compiled here (inline PTX, no nvcc), so which of the three its binary computes is unpinned; they
    differ by at most one fp16 ulp on a small fraction of the outputs (tests/test_gemm_gpu.py::test_epilogue_fma_envelope
    records it)."""
    p = acc.astype(np.float32)
    ws = np.asarray(wscales, np.float16).astype(np.float32)[None, :]
    sa = np.asarray(ascales, np.float16).astype(np.float32)[:, None]
    wz = np.asarray(w_szs, np.float16).astype(np.float32)[None, :]
    ss = np.asarray(a_ssums, np.float16).astype(np.float32)[:, None]
    t = (p * ws).astype(np.float32)
    if fma is True:
        u = (wz * ss).astype(np.float32)
        return _fmaf(t, np.broadcast_to(sa, t.shape), -u).astype(np.float16)
    t = (t * sa).astype(np.float32)
    if fma == "sub":
        return _fmaf(np.broadcast_to(-wz, t.shape), np.broadcast_to(ss, t.shape), t).astype(np.float16)
    u = (wz * ss).astype(np.float32)
    return (t - u).astype(np.float32).astype(np.float16)


def gemm_per_chn(A, qweight, wscales, ascales, w_szs, a_ssums):
    """Returns (acc int32 [M,N], out fp16 [M,N]) of `qgemm_w4a8_per_chn.gemm_forward_cuda`."""
    acc = gemm_per_chn_acc(A, qweight)
    return acc, epilogue_per_chn(acc, wscales, ascales, w_szs, a_ssums)


# --------------------------------------------------------------------------------------------------
# per-group GEMM
# --------------------------------------------------------------------------------------------------
def dequant_per_group_w8(qweight, s2_zeros, s2_scales):
    """Level-2 ("progressive") dequant uint4 -> int8, exactly as the packed-byte arithmetic of
    w4a8_per_group/gemm_cuda.cu:298-324 does it:

        word = four u4 bytes;  word * scale_byte  (32-bit multiply: bytes carry into their neighbours
        when a byte product exceeds 255);  then __vadd4(word, broadcast(zero_byte)) = per-byte wrapping add;
        the result bytes are reinterpreted as int8.

    For checkpoints inside QoQ's protective range (q*s2 <= 255) this equals (q - z) * s2.  The carry
    behaviour for invalid inputs is reproduced too (the four bytes of one 32-bit word are the four
    consecutive k values 4e+f, f=0..3 of one row, gemm_cuda.cu:284-291 + packing order).
    Returns int8 [N, K].
    """
    Q = unpack_qweight(qweight).astype(np.uint64)  # [N,K]
    N, K = Q.shape
    s = unpermute_group_meta(np.asarray(s2_scales).view(np.uint8)).T.astype(np.uint64)  # [N, K/G]
    z = unpermute_group_meta(np.asarray(s2_zeros).view(np.uint8)).T.astype(np.uint64)
    # 32-bit words of four consecutive k (byte 0 = lowest k)
    Qw = Q.reshape(N, K // 4, 4)
    word = Qw[..., 0] | (Qw[..., 1] << 8) | (Qw[..., 2] << 16) | (Qw[..., 3] << 24)
    sg = np.repeat(s, GROUP // 4, axis=1)  # per word
    zg = np.repeat(z, GROUP // 4, axis=1)
    prod = (word * sg) & 0xFFFFFFFF
    out = np.empty((N, K // 4, 4), np.uint8)
    for b in range(4):
        out[..., b] = (((prod >> (8 * b)) & 0xFF) + zg) & 0xFF
    return out.reshape(N, K).view(np.int8)


def gemm_per_group_acc(A, qweight, s2_zeros, s2_scales):
    W8 = dequant_per_group_w8(qweight, s2_zeros, s2_scales)
    return int_matmul(np.asarray(A, np.int8), W8, 128, 128).astype(np.int32)


def epilogue_per_group(acc, wscales, ascales):
    """gemm_cuda.cu:617-622: out = half_rn( float(acc) * (wscale[n] * ascale[m]) )."""
    p = acc.astype(np.float32)
    ws = np.asarray(wscales, np.float16).astype(np.float32)[None, :]
    sa = np.asarray(ascales, np.float16).astype(np.float32)[:, None]
    s = (ws * sa).astype(np.float32)
    return (p * s).astype(np.float32).astype(np.float16)


def gemm_per_group(A, qweight, s2_zeros, s2_scales, wscales, ascales):
    """Returns (acc int32, out fp16) of `qgemm_w4a8_per_group.gemm_forward_cuda`."""
    acc = gemm_per_group_acc(A, qweight, s2_zeros, s2_scales)
    return acc, epilogue_per_group(acc, wscales, ascales)


# --------------------------------------------------------------------------------------------------
# W8A8 (kernels/csrc/qgemm/w8a8/w8a8_gemm_cuda.cu:521-524 epilogue: acc * (wscale * ascale))
# --------------------------------------------------------------------------------------------------
def gemm_w8a8(A, W, wscales, ascales):
    acc = int_matmul(np.asarray(A, np.int8), np.asarray(W, np.int8), 128, 128).astype(np.int32)
    return acc, epilogue_per_group(acc, wscales, ascales)
