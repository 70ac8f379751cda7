"""Oracle: the activation-side kernels adjacent to the hot path (they produce the GEMM's A, ascales, a_ssums).

TEST INFRASTRUCTURE (see oracle/__init__.py).  numpy only.  PARITY UNPINNED by reference tests.

  * invoke_quant / invoke_quant_fuse_sum ... kernels/csrc/fused_kernels.cu:52-137
  * rms_norm_general(_fuse_sum) ............ kernels/csrc/layernorm_kernels.cu:20-29, 189-326, 427-508
        NOTE: despite the name this is TRT-LLM's mean-subtracting generalLayerNorm
        with beta = nullptr; reproduced as is.
  * rms_norm ............................... kernels/csrc/layernorm_kernels.cu:330-362
  * silu_and_mul ........................... kernels/csrc/activation_kernels.cu:7-30

Each function returns the exact integer outputs plus `pre`, the fp32 value that was rounded to int8,
so that tests can exclude exact-tie cases whose rounding depends on fp32 summation order.
"""
import numpy as np


def rni_sat_s8(x):
    """cvt.rni.sat.s8.f32 (utils.cuh:79-84)."""
    r = np.rint(np.nan_to_num(np.asarray(x, np.float32), nan=0.0, posinf=127.0, neginf=-128.0))
    return np.clip(r, -128, 127).astype(np.int8)


def _wave64_butterfly_sum(f):
    """qserve_amd/csrc/common.h wave_sum: xor butterfly 32, 16, 8, 4, 2, 1 over the last axis (64 lanes), fp32."""
    f = np.asarray(f, np.float32)
    lanes = np.arange(64)
    for m in (32, 16, 8, 4, 2, 1):
        f = (f + f[..., lanes ^ m]).astype(np.float32)
    return f[..., 0]


def hip_order_row_sum(contrib, nt=256):
    """The row sums exactly as qserve_amd/csrc/row_ops.h orders them (the LIBRARY's own order, stated here so that the GPU tests can
    demand bit-equality instead of "equal up to the association of an fp32 sum") for the NORM kernels' statistics (mean, variance,
    the sum of the normalised row; invoke_quant_fuse_sum has its own order since round 6: block_order_row_sum): `nt` = 256
    virtual threads per row; thread t owns the 8-element chunks (c * nt + t) * 8 .. + 7 and adds
    its fp32 contributions sequentially (c outer, element inner, starting from +0); wave64 butterfly; waves left to right.
    contrib: float32 [T, H] (H % 8 == 0)."""
    contrib = np.asarray(contrib, np.float32)
    T, H = contrib.shape
    nc = (H + nt * 8 - 1) // (nt * 8)
    pad = np.zeros((T, nc * nt * 8), np.float32)
    pad[:, :H] = contrib
    a = pad.reshape(T, nc, nt, 8)
    acc = np.zeros((T, nt), np.float32)
    for c in range(nc):
        for e in range(8):
            acc = (acc + a[:, c, :, e]).astype(np.float32)
    w = _wave64_butterfly_sum(acc.reshape(T, nt // 64, 64))          # [T, waves]
    tot = w[:, 0]
    for k in range(1, nt // 64):
        tot = (tot + w[:, k]).astype(np.float32)
    return tot


def block_order_row_sum(contrib):
    """The row sum of invoke_quant_fuse_sum (and of the fusions that end in it: silu_and_mul + quant, decode attention + quant)
    exactly as this library orders it since round 6 (qserve_amd/csrc/row_ops.h reduce_max_blocksum): the row is cut into
    512-element blocks = 64 chunks of 8; lane l of a block adds the 8 elements of chunk l left to right (from +0; a chunk beyond
    the row is +0), the 64 lanes go through the wave butterfly (xor 32, 16, 8, 4, 2, 1), and the block sums through ONE MORE wave
    butterfly with lane b = block b and -0.0 in the lanes beyond the row (a fixed tree of depth 6 whatever the row length).  Independent of how many threads a kernel runs - which is what lets the workgroups of the decode
    attention (each holds G x 128 values of the row = whole blocks for G = 4, 8) reproduce invoke_quant_fuse_sum's bits from
    one published partial each.  contrib: float32 [T, H] (H % 8 == 0)."""
    contrib = np.asarray(contrib, np.float32)
    T, H = contrib.shape
    nblk = (H + 511) // 512
    pad = np.zeros((T, nblk * 512), np.float32)
    pad[:, :H] = contrib
    a = pad.reshape(T, nblk, 64, 8)
    acc = np.zeros((T, nblk, 64), np.float32)
    for e in range(8):
        acc = (acc + a[..., e]).astype(np.float32)
    blk = _wave64_butterfly_sum(acc)                                  # [T, nblk]
    assert nblk <= 64, "one lane per block (rows up to 32 768 elements)"
    lanes = np.full((T, 64), -0.0, np.float32)                        # -0.0 = the identity of fp32 addition (x + -0 = x, also for +-0)
    lanes[:, :nblk] = blk
    return _wave64_butterfly_sum(lanes)


def quant_per_token(x, with_sum=False, sum_order="exact"):
    """fused_kernels.cu:58-82 / :106-130: amax over the row (fp32), scale = half_rn(amax/127),
    q = rni_sat_s8(x * (127/amax)) with the UNROUNDED fp32 amax, sum = half_rn(fp32 row sum)."""
    xf = np.asarray(x, np.float16).astype(np.float32)
    amax = np.abs(xf).max(axis=-1)
    scale = (amax / np.float32(127.0)).astype(np.float32).astype(np.float16)
    with np.errstate(divide="ignore", invalid="ignore"):
        tmp = (np.float32(127.0) / amax).astype(np.float32)
        pre = (xf * tmp[..., None]).astype(np.float32)
    q = rni_sat_s8(pre)
    if with_sum:
        # the reference sums in fp32 (fused_kernels.cu:104-122; its own association: 1024 strided threads + warp butterflies);
        # "exact" = the order-free sum rounded once, "hip" = this library's association, bit for bit (block_order_row_sum)
        if sum_order == "hip":
            s = block_order_row_sum(xf).astype(np.float16)
        else:
            s = xf.sum(axis=-1, dtype=np.float64).astype(np.float32).astype(np.float16)
        return q, scale, s, pre
    return q, scale, pre


def _threads_for(hidden):
    b = min(hidden, 1024)
    return 32 * ((b + 31) // 32)   # layernorm_kernels.cu:436-437


def _butterfly_sum32(f):
    """warpReduceSum (reduction_utils.cuh:25-30): xor butterfly 16, 8, 4, 2, 1 over the last axis (32 lanes), fp32 adds;
    every lane ends with the same value (fp32 addition commutes), returned from lane 0."""
    f = np.asarray(f, np.float32)
    lanes = np.arange(32)
    for m in (16, 8, 4, 2, 1):
        f = (f + f[..., lanes ^ m]).astype(np.float32)
    return f[..., 0]


def reference_order_row_sum(val16):
    """`input_sum` of generalLayerNorm_fuse_sum exactly as the reference orders it (layernorm_kernels.cu:275,286,306,323):
    nt = min(H, 1024) threads rounded up to 32 (:480-481); thread t adds elements t, t + nt, ... into a HALF accumulator
    (`T_scalar sum`; `sum += float` resolves to __half::operator+=: one fp16 rounding per addition - computed here in float64,
    which holds any sum of two halves exactly, then rounded once); blockAllReduceSum (reduction_utils.cuh:68-85): butterfly
    inside each warp, warp results to shared[wid], the butterfly again over the 32 slots (unused ones = 0); half_rn."""
    val16 = np.asarray(val16, np.float16)
    T, H = val16.shape
    nt = _threads_for(H)
    part = np.zeros((T, nt), np.float16)
    for i0 in range(0, H, nt):
        seg = val16[:, i0:i0 + nt].astype(np.float64)
        w = seg.shape[1]
        part[:, :w] = (part[:, :w].astype(np.float64) + seg).astype(np.float16)
    warp = _butterfly_sum32(part.astype(np.float32).reshape(T, nt // 32, 32))      # [T, nwarps]
    slots = np.zeros((T, 32), np.float32)
    slots[:, : nt // 32] = warp
    return _butterfly_sum32(slots).astype(np.float16)


def rms_norm_general(x, gamma, eps, with_sum=False, sum_order="reference", stats_order="exact"):
    """generalLayerNorm(_fuse_sum), per-token dynamic scaling branch (layernorm_kernels.cu:207-326).

    mean = sum(x)/H; var = sum((x-mean)^2)/H; rstd = rsqrt(var+eps);
    val  = half_rn((x-mean)*rstd*gamma)         (compute_layernorm in fp32, cast to T=half)
    amax = max(|val|, 1e-6) kept in half; sum accumulated PER THREAD in a half variable over that thread's
    strided elements, then reduced over the block in fp32 (:275,:286) and stored as half;
    q = rni_sat_s8( ((x-mean)*rstd*gamma) * (127/amax) )  - the un-rounded fp32 value is re-computed (:312);
    scale = half_rn(amax/127).
    sum_order: "reference" (default) = reference_order_row_sum above - what the HIP kernels compute under
    qs_set_row_sum_order(1), bit for bit; "hip" = the library's default order, bit for bit; "fp32" = the order-free exact sum.
    stats_order: "exact" (default) = mean / variance order-free (float64, rounded once); "hip" = the library's fp32 association
    (then int8 rows, scales and sums equal the HIP kernels' bit for bit; with "exact" they agree except where an fp32 statistic
    one ulp apart flips a rounding).
    """
    xf = np.asarray(x, np.float16).astype(np.float32)
    g = np.asarray(gamma, np.float16).astype(np.float32)
    T, H = xf.shape
    if stats_order == "hip":
        # mean / variance with the library's own fp32 association (hip_order_row_sum): the normalised values - and with them
        # every int8 byte, the scale and both forms of the row sum - are then BIT-EQUAL to the HIP kernels', ties included
        mean = (hip_order_row_sum(xf) / np.float32(H)).astype(np.float32)
        diff = (xf - mean[:, None]).astype(np.float32)
        var = (hip_order_row_sum((diff * diff).astype(np.float32)) / np.float32(H)).astype(np.float32)
    else:
        mean = (xf.sum(axis=-1, dtype=np.float64) / H).astype(np.float32)
        diff = (xf - mean[:, None]).astype(np.float32)
        var = ((diff.astype(np.float64) ** 2).sum(axis=-1) / H).astype(np.float32)
    rstd = (np.float32(1.0) / np.sqrt((var + np.float32(eps)).astype(np.float32))).astype(np.float32)
    valf = ((diff * rstd[:, None]).astype(np.float32) * g[None, :]).astype(np.float32)
    val16 = valf.astype(np.float16)
    amax = np.maximum(np.abs(val16).max(axis=-1), np.float16(1e-6)).astype(np.float32)
    scale = (amax / np.float32(127.0)).astype(np.float32).astype(np.float16)
    dyn = (np.float32(127.0) / amax).astype(np.float32)
    pre = (valf * dyn[:, None]).astype(np.float32)
    q = rni_sat_s8(pre)
    if not with_sum:
        return q, scale, pre
    if sum_order == "reference":
        s = reference_order_row_sum(val16)
    elif sum_order == "hip":                           # the library's default order (qs_set_row_sum_order(0)), bit for bit
        s = hip_order_row_sum(val16.astype(np.float32)).astype(np.float16)
    else:                                              # "fp32": the exact sum of the fp16 values, rounded once (order-free)
        s = val16.astype(np.float64).sum(axis=-1).astype(np.float32).astype(np.float16)
    return q, scale, s, pre


def rms_norm(x, weight, eps):
    """rms_norm_kernel<use_quant=false> (layernorm_kernels.cu:330-362): out = half(x*rstd) * weight (half mul)."""
    xf = np.asarray(x, np.float16).astype(np.float32)
    H = xf.shape[-1]
    var = ((xf.astype(np.float64) ** 2).sum(axis=-1) / H).astype(np.float32)
    rstd = (np.float32(1.0) / np.sqrt((var + np.float32(eps)).astype(np.float32))).astype(np.float32)
    t = (xf * rstd[..., None]).astype(np.float32).astype(np.float16)
    return (t.astype(np.float32) * np.asarray(weight, np.float16).astype(np.float32)).astype(np.float16)


def silu_and_mul(x):
    """activation_kernels.cu:9-30: out = half( half(x/(1+exp(-x))) * y ), x = in[:, :d], y = in[:, d:]."""
    xin = np.asarray(x, np.float16)
    d = xin.shape[-1] // 2
    a = xin[..., :d].astype(np.float32)
    b = xin[..., d:].astype(np.float32)
    s = (a / (np.float32(1.0) + np.exp(-a).astype(np.float32))).astype(np.float32).astype(np.float16)
    return (s.astype(np.float32) * b).astype(np.float32).astype(np.float16)
