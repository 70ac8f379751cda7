"""Mirror of `qserve_backend.fused_attention` (kernels/csrc/fused_attention/fused_attention.cpp:243-256)."""
import torch

from ._util import check, expect, guard, lib, ptr, stream


def single_query_attention(q, k, v, kv_pointers, length_per_sample, alibi_slopes, memory_max_seqlen,
                           tokens_per_block, size_per_token, timestep, rotary_embedding_dim, rotary_base,
                           neox_rotary_style, int4_kv_cache, kv_cache_with_zeros):
    """fused_attention.h:13-28.  q [B,H,Dh] / k,v [B,Hkv,Dh] fp16 views of the qkv buffer; kv_pointers int64
    [B,2,max_blocks] of page addresses; returns a fresh contiguous fp16 [B,H,Dh] (torch::empty_like(q)).
    `alibi_slopes` and `neox_rotary_style` are accepted and have no effect - as in the reference (see below)."""
    for n, t in (("q", q), ("k", k), ("v", v)):
        expect(t, torch.float16, n, contiguous=False)
    expect(kv_pointers, torch.int64, "kv_pointers")
    batch = kv_pointers.size(0)
    nheads, nheads_kv, headdim = q.size(1), k.size(1), k.size(-1)
    # fused_attention.cpp:179-180
    if not (k.stride(2) == 1 and k.stride(1) == headdim and v.stride(2) == 1 and v.stride(1) == headdim):
        raise RuntimeError("k and v must have stride(2) == 1 and stride(1) == head_dim")
    if not (q.stride(2) == 1 and q.stride(1) == headdim):
        raise RuntimeError("q must have stride(2) == 1 and stride(1) == head_dim")
    if length_per_sample is not None:
        expect(length_per_sample, torch.int32, "length_per_sample")
        if tuple(length_per_sample.shape) != (batch,):
            raise RuntimeError("length_per_sample must have shape (batch_size)")
    if alibi_slopes is not None:
        # validated and then IGNORED, exactly as the reference does: fused_attention.cpp:193-199 checks device / shape /
        # dtype, set_params never stores the pointer (`// params.linear_bias_slopes = alibi_slopes_ptr;`, :91) and the
        # kernel's linear-bias lines are commented out (decoderMaskedMultiheadAttentionTemplate.hpp:1604-1615)
        expect(alibi_slopes, torch.float32, "alibi_slopes")
        if tuple(alibi_slopes.shape) != (nheads,):
            raise RuntimeError("alibi_slopes must have shape (nheads)")
    out = torch.empty((q.size(0), nheads, headdim), dtype=q.dtype, device=q.device)
    with guard(q):
        check(lib.qs_single_query_attention(ptr(q), ptr(k), ptr(v), ptr(kv_pointers), ptr(length_per_sample), ptr(out),
                                            batch, nheads, nheads_kv, headdim, q.stride(0), k.stride(0),
                                            kv_pointers.size(-1), int(memory_max_seqlen), int(tokens_per_block),
                                            int(size_per_token), int(timestep), int(rotary_embedding_dim),
                                            float(rotary_base), int(bool(neox_rotary_style)), int(bool(int4_kv_cache)),
                                            int(bool(kv_cache_with_zeros)), stream()),
              "fused_attention.single_query_attention")
    return out


def apply_bias_rope_update_kv_cache(qkv, seq_lens, padding_offset, kv_pointers, head_num, kv_head_num, seq_len,
                                    tokens_per_block, size_per_token, rotary_embedding_dim, rotary_embedding_base,
                                    rotary_embedding_max_positions, neox_rotary_style, int4_kv_cache,
                                    kv_cache_with_zeros):
    """update_kv_cache.h:11-27.  In place on qkv (fp16 [T, (H+2Hkv)*Dh]) and on the pages behind kv_pointers."""
    expect(qkv, torch.float16, "qkv")
    expect(seq_lens, torch.int32, "seq_lens")
    expect(padding_offset, torch.int32, "padding_offset")
    mb = 0
    if kv_pointers is not None:
        expect(kv_pointers, torch.int64, "kv_pointers")
        mb = kv_pointers.size(-1)
    with guard(qkv):
        check(lib.qs_apply_bias_rope_update_kv_cache(ptr(qkv), ptr(seq_lens), ptr(padding_offset), ptr(kv_pointers),
                                                     qkv.size(0), seq_lens.size(0), mb, int(head_num), int(kv_head_num),
                                                     int(seq_len), int(tokens_per_block), int(size_per_token),
                                                     int(rotary_embedding_dim), float(rotary_embedding_base),
                                                     int(rotary_embedding_max_positions), int(bool(neox_rotary_style)),
                                                     int(bool(int4_kv_cache)), int(bool(kv_cache_with_zeros)), stream()),
              "fused_attention.apply_bias_rope_update_kv_cache")


def compute_padding_offsets(cu_seqlens, max_seqlen, tot_num_tokens):
    """input_metadata_helper.h:12-13 -> int32 [tot_num_tokens]."""
    expect(cu_seqlens, torch.int32, "cu_seqlens")
    out = torch.empty((int(tot_num_tokens),), dtype=torch.int32, device=cu_seqlens.device)
    with guard(out):
        check(lib.qs_compute_padding_offsets(ptr(out), ptr(cu_seqlens), cu_seqlens.size(0) - 1, int(max_seqlen), stream()),
              "fused_attention.compute_padding_offsets")
    return out
