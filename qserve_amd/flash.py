"""Prefill attention provider: `flash_attn_varlen_func` with flash-attn v2's signature and semantics for the subset the
reference uses (llama_w4a8_unpad.py:232-242 and its w8a8 / w16a16 / mixtral twins): fp16, head_dim 128, GQA, causal or
full, variable-length batches described by cu_seqlens.  Backed by qserve_amd/csrc/flash_prefill.hip."""
import math

import torch

from .backend._util import check, guard, lib, on_device, ptr, stream


def _expect_f16(t, name):
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float16 or not on_device(t):
        raise RuntimeError(f"flash_attn_varlen_func: {name} must be a CUDA float16 tensor")
    if t.dim() != 3 or t.stride(2) != 1 or t.stride(1) != t.size(2):
        raise RuntimeError(f"flash_attn_varlen_func: {name} must be [tokens, heads, head_dim] with contiguous heads "
                           "(a view into a packed qkv buffer is fine)")


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                           softmax_scale=None, causal=False, window_size=(-1, -1), alibi_slopes=None,
                           deterministic=False, return_attn_probs=False, block_table=None):
    if dropout_p != 0.0 or tuple(window_size) != (-1, -1) or alibi_slopes is not None or return_attn_probs or \
            block_table is not None:
        raise NotImplementedError("flash_attn_varlen_func: dropout / sliding window / ALiBi / returned probabilities / "
                                  "paged KV are not provided (the reference's prefill path does not use them)")
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _expect_f16(t, n)
    if q.size(2) != 128 or k.size(2) != 128 or v.size(2) != 128:
        raise NotImplementedError("flash_attn_varlen_func: only head_dim 128 is provided")
    if k.size(1) != v.size(1) or q.size(1) % k.size(1) != 0 or k.size(0) != v.size(0):
        raise RuntimeError("flash_attn_varlen_func: inconsistent head / token counts")
    for c, n in ((cu_seqlens_q, "cu_seqlens_q"), (cu_seqlens_k, "cu_seqlens_k")):
        if c.dtype != torch.int32 or not on_device(c) or not c.is_contiguous():
            raise RuntimeError(f"flash_attn_varlen_func: {n} must be a contiguous CUDA int32 tensor")
    batch = cu_seqlens_q.numel() - 1
    if cu_seqlens_k.numel() - 1 != batch:
        raise RuntimeError("flash_attn_varlen_func: cu_seqlens_q / cu_seqlens_k describe different batch sizes")
    out = torch.empty((q.size(0), q.size(1), 128), dtype=torch.float16, device=q.device)
    scale = 1.0 / math.sqrt(128.0) if softmax_scale is None else float(softmax_scale)
    with guard(q):
        check(lib.qs_flash_attn_varlen_fwd(ptr(q), ptr(k), ptr(v), ptr(out), ptr(cu_seqlens_q), ptr(cu_seqlens_k), batch,
                                           q.size(1), k.size(1), 128, q.stride(0), k.stride(0), v.stride(0), out.stride(0),
                                           int(max_seqlen_q), int(max_seqlen_k), scale, 1 if causal else 0, stream()),
              "flash_attn_varlen_func")
    return out
