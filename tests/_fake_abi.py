"""Host-memory simulator of libqserve_amd.so's C ABI, built on the oracle -- TEST INFRASTRUCTURE ONLY.

Why: the reference's Python (qserve/modeling/...) can only be imported in the authoring container (CPU, /root/reference
present) while the HIP kernels only run on the GPU box (no /root/reference).  To prove that the UNCHANGED reference
modules drive the `qserve_backend` mirror correctly, the tests here replace the ctypes entry points of
`qserve_amd._lib.lib` with the functions below.  Each one receives exactly what the real C function receives - raw
addresses, sizes and flags in the order of include/qserve_amd.h - re-materialises the tensors from those addresses
(host memory in this case) and computes the result with the oracle.  Everything above the C ABI (argument checks,
shape / stride lowering, in-place vs returned outputs, op order) is the product code under test.

Never imported by the product; `install(monkeypatch)` is the only entry point.
"""
import ctypes

import numpy as np

from oracle import flash as oflash
from oracle import fused as ofused
from oracle import kvattn, w4a8


def _arr(addr, shape, dtype):
    """numpy view of host memory at `addr` (what a device pointer is on the real library)."""
    addr = getattr(addr, "value", addr)            # (a ctypes.c_void_p the caller cast itself)
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if n == 0:
        return np.zeros(shape, dtype)
    buf = (ctypes.c_uint8 * n).from_address(int(addr))
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def _strided_rows(addr, rows, row_stride_elems, width, dtype):
    """[rows, width] view with an element stride between rows (views into the packed qkv buffer)."""
    if rows == 0:
        return np.zeros((0, width), dtype)
    item = np.dtype(dtype).itemsize
    n = ((rows - 1) * row_stride_elems + width) * item
    buf = (ctypes.c_uint8 * n).from_address(int(addr))
    base = np.frombuffer(buf, dtype=dtype)
    return np.lib.stride_tricks.as_strided(base, (rows, width), (row_stride_elems * item, item))


class _AddrSpace:
    """pool[block] for the oracle's PagePool when 'block' is a raw page ADDRESS (kv_pointers carry addresses)."""

    def __init__(self, page_bytes):
        self.pb = page_bytes

    def __getitem__(self, addr):
        return _arr(addr, (self.pb,), np.uint8)


class AddrPool(kvattn.PagePool):
    def __init__(self, num_kv_heads, int4):
        self.hkv, self.dh, self.int4 = num_kv_heads, 128, int4
        self.dhb = 64 if int4 else 128
        self.pb = kvattn.page_bytes(num_kv_heads, 128, int4)
        self.k = _AddrSpace(self.pb)
        self.v = _AddrSpace(self.pb)
        self.scale_off = num_kv_heads * 64 * self.dhb
        self.zero_off = self.scale_off + num_kv_heads * 64 * 2


class _NullPool(AddrPool):
    def write_token(self, *a, **k):
        pass


CALLS = []   # (symbol, selected scalar args) in call order: lets tests assert the op sequence


def qs_w4a8_per_chn_gemm(in_feats, kernel, wscales, ascales, w_szs, a_ssums, out, M, N, K, stream):
    CALLS.append(("qs_w4a8_per_chn_gemm", M, N, K))
    assert N % 64 == 0 and K % 128 == 0
    A = _arr(in_feats, (M, K), np.int8)
    W = _arr(kernel, (N, K // 2), np.int8)
    _, o = w4a8.gemm_per_chn(A, W, _arr(wscales, (N,), np.float16), _arr(ascales, (M,), np.float16),
                             _arr(w_szs, (N,), np.float16), _arr(a_ssums, (M,), np.float16))
    _arr(out, (M, N), np.float16)[:] = o
    return 0


def qs_w4a8_per_group_gemm(in_feats, kernel, zeros, scales_i8, wscales, ascales, out, M, N, K, stream):
    CALLS.append(("qs_w4a8_per_group_gemm", M, N, K))
    assert N % 64 == 0 and K % 128 == 0
    A = _arr(in_feats, (M, K), np.int8)
    W = _arr(kernel, (N, K // 2), np.int8)
    _, o = w4a8.gemm_per_group(A, W, _arr(zeros, (K // 128, N), np.int8), _arr(scales_i8, (K // 128, N), np.int8),
                               _arr(wscales, (N,), np.float16), _arr(ascales, (M,), np.float16))
    _arr(out, (M, N), np.float16)[:] = o
    return 0


def qs_w4a8_per_chn_gemm_silu_mul(in_feats, kernel, wscales, ascales, w_szs, a_ssums, out_act, tmp, M, N, K, stream):
    y = np.empty((M, N), np.float16)
    qs_w4a8_per_chn_gemm(in_feats, kernel, wscales, ascales, w_szs, a_ssums, y.ctypes.data, M, N, K, stream)
    _arr(out_act, (M, N // 2), np.float16)[:] = ofused.silu_and_mul(y)
    return 0


def qs_w4a8_per_group_gemm_silu_mul(in_feats, kernel, zeros, scales_i8, wscales, ascales, out_act, tmp, M, N, K, stream):
    y = np.empty((M, N), np.float16)
    qs_w4a8_per_group_gemm(in_feats, kernel, zeros, scales_i8, wscales, ascales, y.ctypes.data, M, N, K, stream)
    _arr(out_act, (M, N // 2), np.float16)[:] = ofused.silu_and_mul(y)
    return 0


def qs_invoke_quant(out, inp, input_sum, scale, T, hidden, stream):
    CALLS.append(("qs_invoke_quant", T, hidden, bool(input_sum)))
    x = _arr(inp, (T, hidden), np.float16)
    if input_sum:
        q, sc, sm, _ = ofused.quant_per_token(x, with_sum=True)
        _arr(input_sum, (T,), np.float16)[:] = sm
    else:
        q, sc, _ = ofused.quant_per_token(x)
    _arr(out, (T, hidden), np.int8)[:] = q
    _arr(scale, (T,), np.float16)[:] = sc
    return 0


def qs_rms_norm_general(out, inp, weight, input_sum, scaling, eps, T, hidden, stream):
    CALLS.append(("qs_rms_norm_general", T, hidden, bool(input_sum)))
    x = _arr(inp, (T, hidden), np.float16)
    g = _arr(weight, (hidden,), np.float16)
    if input_sum:
        q, sc, sm, _ = ofused.rms_norm_general(x, g, eps, with_sum=True)
        _arr(input_sum, (T,), np.float16)[:] = sm
    else:
        q, sc, _ = ofused.rms_norm_general(x, g, eps)
    _arr(out, (T, hidden), np.int8)[:] = q
    _arr(scaling, (T,), np.float16)[:] = sc
    return 0


def qs_rms_norm(out, inp, weight, eps, T, hidden, stream):
    CALLS.append(("qs_rms_norm", T, hidden))
    _arr(out, (T, hidden), np.float16)[:] = ofused.rms_norm(_arr(inp, (T, hidden), np.float16),
                                                             _arr(weight, (hidden,), np.float16), eps)
    return 0


def qs_silu_and_mul(out, inp, T, d, stream):
    CALLS.append(("qs_silu_and_mul", T, d))
    _arr(out, (T, d), np.float16)[:] = ofused.silu_and_mul(_arr(inp, (T, 2 * d), np.float16))
    return 0


def qs_residual_add(a, b, numel, stream):
    CALLS.append(("qs_residual_add", numel))
    x = _arr(a, (numel,), np.float16)
    x[:] = (x.astype(np.float32) + _arr(b, (numel,), np.float16).astype(np.float32)).astype(np.float16)
    return 0


def qs_argmax_rows(x, out, rows, n, row_stride, stream):
    CALLS.append(("qs_argmax_rows", rows, n))
    logits = _strided_rows(x, rows, row_stride, n, np.float16)
    _arr(out, (rows,), np.int64)[:] = np.argmax(logits.astype(np.float32), axis=1)   # first maximum, like the kernel
    return 0


def qs_add_residual_rms_norm_general(out, hidden_io, delta, weight, input_sum, scaling, eps, T, hidden, stream):
    qs_residual_add(hidden_io, delta, T * hidden, stream)
    return qs_rms_norm_general(out, hidden_io, weight, input_sum, scaling, eps, T, hidden, stream)


# ---- K-slice planes: the GEMM leaves int32 partial sums per slice, the row kernel finishes it (include/qserve_amd.h) --------------
FAKE_PLANE_SLICES = 2


def qs_w4a8_gemm_planes_plan(per_group, M, N, K, plan4):
    p = _arr(plan4, (4,), np.int32)
    p[:] = (FAKE_PLANE_SLICES if K >= 1024 and N % 64 == 0 and K % 128 == 0 else 0, 2, 1, (M + 31) // 32)
    return 0


def _split_planes(acc, planes, M, N):
    """the exact sum in `FAKE_PLANE_SLICES` unequal integer parts (the product must sum them, whatever the split)"""
    pl = _arr(planes, (FAKE_PLANE_SLICES, M, N), np.int32)
    pl[1:] = 7
    pl[0] = acc - 7 * (FAKE_PLANE_SLICES - 1)


def qs_w4a8_per_chn_gemm_planes(in_feats, kernel, planes, M, N, K, stream):
    CALLS.append(("qs_w4a8_per_chn_gemm_planes", M, N, K))
    _split_planes(w4a8.gemm_per_chn_acc(_arr(in_feats, (M, K), np.int8), _arr(kernel, (N, K // 2), np.int8)), planes, M, N)
    return 0


def qs_w4a8_per_group_gemm_planes(in_feats, kernel, zeros, scales_i8, planes, M, N, K, stream):
    CALLS.append(("qs_w4a8_per_group_gemm_planes", M, N, K))
    acc = w4a8.gemm_per_group_acc(_arr(in_feats, (M, K), np.int8), _arr(kernel, (N, K // 2), np.int8),
                                  _arr(zeros, (K // 128, N), np.int8), _arr(scales_i8, (K // 128, N), np.int8))
    _split_planes(acc, planes, M, N)
    return 0


def qs_add_residual_rms_norm_general_planes(out, hidden_io, planes, k_slices, plane_stride, wscales, w_szs, ascales, a_ssums,
                                            weight, input_sum, scaling, eps, T, hidden, stream):
    CALLS.append(("qs_add_residual_rms_norm_general_planes", k_slices, T, hidden))
    assert plane_stride == T * hidden
    acc = _arr(planes, (k_slices, T, hidden), np.int32).astype(np.int64).sum(axis=0).astype(np.int32)
    ws, sa = _arr(wscales, (hidden,), np.float16), _arr(ascales, (T,), np.float16).copy()
    if w_szs:
        delta = w4a8.epilogue_per_chn(acc, ws, sa, _arr(w_szs, (hidden,), np.float16), _arr(a_ssums, (T,), np.float16).copy())
    else:
        delta = w4a8.epilogue_per_group(acc, ws, sa)
    delta = np.ascontiguousarray(delta)
    return qs_add_residual_rms_norm_general(out, hidden_io, delta.ctypes.data, weight, input_sum, scaling, eps, T, hidden, stream)


def qs_silu_and_mul_quant(out, inp, input_sum, scale, T, d, stream):
    tmp = np.empty((T, d), np.float16)
    tmp[:] = ofused.silu_and_mul(_arr(inp, (T, 2 * d), np.float16))
    return qs_invoke_quant(out, tmp.ctypes.data, input_sum, scale, T, d, stream)


def qs_compute_padding_offsets(out, cu_seqlens, batch, max_seqlen, stream):
    CALLS.append(("qs_compute_padding_offsets", batch, max_seqlen))
    cu = _arr(cu_seqlens, (batch + 1,), np.int32)
    tot = int(cu[batch])
    _arr(out, (tot,), np.int32)[:] = kvattn.compute_padding_offsets(cu, max_seqlen, tot)
    return 0


def qs_apply_bias_rope_update_kv_cache(qkv, seq_lens, padding_offset, kv_pointers, num_tokens, batch, max_blocks, H, Hkv,
                                       seq_len, tokens_per_block, size_per_token, rot_dim, base, max_pos, neox, int4,
                                       with_zeros, stream):
    CALLS.append(("qs_apply_bias_rope_update_kv_cache", num_tokens, batch, H, Hkv, seq_len, int4))
    assert tokens_per_block == 64 and rot_dim == 128 and with_zeros      # neox: no effect (update_kv_cache.cu:57)
    x = _arr(qkv, (num_tokens, (H + 2 * Hkv) * 128), np.float16)
    sl = _arr(seq_lens, (batch,), np.int32)
    pad = _arr(padding_offset, (num_tokens,), np.int32)
    if kv_pointers:
        assert size_per_token == Hkv * (64 if int4 else 128)
        tables = _arr(kv_pointers, (batch, 2, max_blocks), np.int64)
        pool = AddrPool(Hkv, bool(int4))
    else:
        tables = np.zeros((batch, 2, (seq_len + 63) // 64), np.int64)
        pool = _NullPool(Hkv, bool(int4))
    kvattn.prefill_update_kv_cache(x, sl, pad, tables, pool, H, Hkv, seq_len, np.float32(base))
    return 0


def qs_single_query_attention(q, k, v, kv_pointers, length_per_sample, out, batch, H, Hkv, head_dim, q_stride0, kv_stride0,
                              max_blocks, memory_max_seqlen, tokens_per_block, size_per_token, timestep, rot_dim, base,
                              neox, int4, with_zeros, stream):
    CALLS.append(("qs_single_query_attention", batch, H, Hkv, max_blocks, timestep, int4))
    assert head_dim == 128 and tokens_per_block == 64 and rot_dim == 128 and with_zeros   # neox: no effect
    assert size_per_token == Hkv * (64 if int4 else 128)
    qa = _strided_rows(q, batch, q_stride0, H * 128, np.float16).reshape(batch, H, 128)
    ka = _strided_rows(k, batch, kv_stride0, Hkv * 128, np.float16).reshape(batch, Hkv, 128)
    va = _strided_rows(v, batch, kv_stride0, Hkv * 128, np.float16).reshape(batch, Hkv, 128)
    tables = _arr(kv_pointers, (batch, 2, max_blocks), np.int64)
    if length_per_sample:
        lengths = _arr(length_per_sample, (batch,), np.int32)
    else:                                            # Template.hpp:901: tlength = timestep
        lengths = np.full((batch,), timestep + 1, np.int32)
    o = kvattn.decode_attention(np.array(qa), np.array(ka), np.array(va), tables, lengths, AddrPool(Hkv, bool(int4)),
                                np.float32(base), "kernel")
    _arr(out, (batch, H, 128), np.float16)[:] = o
    return 0


def qs_single_query_attention_quant(q, k, v, kv_pointers, length_per_sample, out, quant_out, quant_sum, quant_scale, batch, H,
                                    Hkv, head_dim, q_stride0, kv_stride0, max_blocks, memory_max_seqlen, tokens_per_block,
                                    size_per_token, timestep, rot_dim, base, neox, int4, with_zeros, stream):
    """The pair it is defined as (include/qserve_amd.h): attention, then invoke_quant(_fuse_sum) of its output."""
    rc = qs_single_query_attention(q, k, v, kv_pointers, length_per_sample, out, batch, H, Hkv, head_dim, q_stride0,
                                   kv_stride0, max_blocks, memory_max_seqlen, tokens_per_block, size_per_token, timestep,
                                   rot_dim, base, neox, int4, with_zeros, stream)
    return rc or qs_invoke_quant(quant_out, out, quant_sum, quant_scale, batch, H * head_dim, stream)


def qs_flash_attn_varlen_fwd(q, k, v, out, cu_q, cu_k, batch, H, Hkv, head_dim, qs0, ks0, vs0, os0, max_q, max_k, scale,
                             causal, stream):
    CALLS.append(("qs_flash_attn_varlen_fwd", batch, H, Hkv, max_q, max_k, causal))
    cq = _arr(cu_q, (batch + 1,), np.int32)
    ck = _arr(cu_k, (batch + 1,), np.int32)
    tq, tk = int(cq[batch]), int(ck[batch])
    qa = _strided_rows(q, tq, qs0, H * 128, np.float16).reshape(tq, H, 128)
    ka = _strided_rows(k, tk, ks0, Hkv * 128, np.float16).reshape(tk, Hkv, 128)
    va = _strided_rows(v, tk, vs0, Hkv * 128, np.float16).reshape(tk, Hkv, 128)
    o = oflash.attention_varlen(qa, ka, va, cq, ck, scale, bool(causal))
    _strided_rows(out, tq, os0, H * 128, np.float16)[:] = o.reshape(tq, H * 128).astype(np.float16)
    return 0


SYMBOLS = {f.__name__: f for f in (
    qs_w4a8_per_chn_gemm, qs_w4a8_per_group_gemm, qs_w4a8_per_chn_gemm_silu_mul, qs_w4a8_per_group_gemm_silu_mul,
    qs_invoke_quant, qs_rms_norm_general, qs_rms_norm, qs_silu_and_mul,
    qs_residual_add, qs_argmax_rows, qs_add_residual_rms_norm_general, qs_silu_and_mul_quant, qs_compute_padding_offsets,
    qs_w4a8_gemm_planes_plan, qs_w4a8_per_chn_gemm_planes, qs_w4a8_per_group_gemm_planes, qs_add_residual_rms_norm_general_planes,
    qs_apply_bias_rope_update_kv_cache, qs_single_query_attention, qs_single_query_attention_quant,
    qs_flash_attn_varlen_fwd)}


class _NoGuard:
    def __init__(self, t):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def install(monkeypatch):
    """Route the mirror's C calls to this simulator and let it accept host tensors: only the device check of `expect`,
    the stream lookup and the device guard are neutralised - dtype / shape / stride / contiguity checks stay live."""
    import torch

    import qserve_amd.backend._util as U
    import qserve_amd.flash as flashmod
    import qserve_amd.fused as fusedmod
    from qserve_amd._lib import lib
    for name, fn in SYMBOLS.items():
        monkeypatch.setattr(lib, name, fn, raising=True)

    def expect_host(t, dtype, name, contiguous=True):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name} must be a torch.Tensor")
        if t.dtype != dtype:
            raise RuntimeError(f"expected scalar type {dtype} for {name} but found {t.dtype}")
        if contiguous and not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")

    import importlib
    mods = [importlib.import_module("qserve_amd.backend." + m) for m in
            ("qgemm_w4a8_per_chn", "qgemm_w4a8_per_group", "qgemm_w8a8", "fused_attention", "fused_kernels",
             "layernorm_ops", "activation_ops")] + [fusedmod, flashmod, U, importlib.import_module("qserve_amd.decode")]
    for m in mods:
        for attr, repl in (("stream", lambda: 0), ("expect", expect_host), ("guard", _NoGuard),
                           ("on_device", lambda t: True)):
            if hasattr(m, attr):
                monkeypatch.setattr(m, attr, repl)
    del CALLS[:]
    return CALLS
