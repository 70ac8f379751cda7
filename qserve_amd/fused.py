"""Pair fusions for the decode loop (extensions; NOT part of the reference's `qserve_backend` surface).

Each function is bit-identical to the two reference ops it replaces (tests/test_fused_gpu.py checks that on the GPU);
they exist because at decode batch sizes each row kernel is a fixed ~5 us latency chain, so a pair costs one.

    add_residual_rms_norm_general(_fuse_sum)  ==  hidden += delta ; layernorm_ops.rms_norm_general(_fuse_sum)(hidden)
    silu_and_mul_quant(_fuse_sum)             ==  activation_ops.silu_and_mul ; fused_kernels.invoke_quant(_fuse_sum)
    single_query_attention_quant(_fuse_sum)   ==  fused_attention.single_query_attention ; fused_kernels.invoke_quant(_fuse_sum)
    gemm_silu_and_mul_per_chn / _per_group    ==  qgemm_w4a8_per_*.gemm_forward_cuda(gate_up) ; activation_ops.silu_and_mul

"""
import torch

from .backend._util import check, expect, guard, lib, ptr, stream


def add_residual_rms_norm_general(out, hidden, delta, weight, scaling, epsilon, input_sum=None):
    """hidden (fp16 [T, hid], updated in place) += delta; then out/scaling(/input_sum) = rms_norm_general(hidden)."""
    expect(out, torch.int8, "out")
    expect(hidden, torch.float16, "hidden")
    expect(delta, torch.float16, "delta")
    expect(weight, torch.float16, "weight")
    expect(scaling, torch.float16, "scaling")
    if input_sum is not None:
        expect(input_sum, torch.float16, "input_sum")
    if hidden.shape != delta.shape:
        raise RuntimeError(f"add_residual_rms_norm_general: hidden {tuple(hidden.shape)} vs delta {tuple(delta.shape)}")
    hid = hidden.size(-1)
    with guard(out):
        check(lib.qs_add_residual_rms_norm_general(ptr(out), ptr(hidden), ptr(delta), ptr(weight),
                                                   ptr(input_sum) if input_sum is not None else 0, ptr(scaling),
                                                   float(epsilon), hidden.numel() // hid, hid, stream()),
              "fused.add_residual_rms_norm_general")


def silu_and_mul_quant(out, input, scale, input_sum=None):
    """out int8 [T, d], scale(/input_sum) fp16 [T] = invoke_quant(silu_and_mul(input fp16 [T, 2d]))."""
    expect(out, torch.int8, "out")
    expect(input, torch.float16, "input")
    expect(scale, torch.float16, "scale")
    if input_sum is not None:
        expect(input_sum, torch.float16, "input_sum")
    d = input.size(-1) // 2
    if out.size(-1) != d:
        raise RuntimeError(f"silu_and_mul_quant: out width {out.size(-1)} != {d}")
    with guard(out):
        check(lib.qs_silu_and_mul_quant(ptr(out), ptr(input), ptr(input_sum) if input_sum is not None else 0, ptr(scale),
                                        input.numel() // (2 * d), d, stream()), "fused.silu_and_mul_quant")


def gemm_silu_and_mul_per_chn(in_feats, kernel, wscales, ascales, w_szs, a_ssums, out_act, tmp=None):
    """out_act fp16 [M, N/2] = silu_and_mul(per-channel W4A8 GEMM of the stacked gate_up weight `kernel` [N, K/2]) - one
    launch where the GEMM kernel has the activation epilogue, two through `tmp` (fp16 [M, N]) otherwise; bit-identical."""
    _gemm_silu(False, in_feats, kernel, (wscales, ascales, w_szs, a_ssums), out_act, tmp)


def gemm_silu_and_mul_per_group(in_feats, kernel, zeros, scales_i8, wscales, ascales, out_act, tmp=None):
    """Per-group (g128) form of `gemm_silu_and_mul_per_chn`."""
    _gemm_silu(True, in_feats, kernel, (zeros, scales_i8, wscales, ascales), out_act, tmp)


def _gemm_silu(per_group, in_feats, kernel, rest, out_act, tmp):
    expect(in_feats, torch.int8, "in_feats")
    expect(kernel, torch.int8, "kernel")
    expect(out_act, torch.float16, "out_act")
    for i, t in enumerate(rest):
        expect(t, torch.int8 if per_group and i < 2 else torch.float16, f"operand {i}")
    K = in_feats.size(-1)
    M, N = in_feats.numel() // K, kernel.size(0)
    if kernel.size(1) * 2 != K:
        raise RuntimeError(f"gemm_silu_and_mul: kernel {tuple(kernel.shape)} does not match K={K}")
    if out_act.numel() != M * (N // 2):
        raise RuntimeError(f"gemm_silu_and_mul: out_act has {out_act.numel()} elements, expected {M} x {N // 2}")
    if tmp is not None:
        expect(tmp, torch.float16, "tmp")
        if tmp.numel() < M * N:
            raise RuntimeError(f"gemm_silu_and_mul: tmp has {tmp.numel()} elements, needs {M} x {N}")
    fn = lib.qs_w4a8_per_group_gemm_silu_mul if per_group else lib.qs_w4a8_per_chn_gemm_silu_mul
    with guard(out_act):
        check(fn(ptr(in_feats), ptr(kernel), *[ptr(t) for t in rest], ptr(out_act), ptr(tmp) if tmp is not None else 0,
                 M, N, K, stream()), "fused.gemm_silu_and_mul")


def single_query_attention_quant(q, k, v, kv_pointers, length_per_sample, quant_out, quant_scale, memory_max_seqlen,
                                 tokens_per_block, size_per_token, timestep, rotary_embedding_dim, rotary_base,
                                 neox_rotary_style, int4_kv_cache, kv_cache_with_zeros, quant_sum=None):
    """attn = single_query_attention(q, k, v, ...) (fp16 [B, H, Dh], returned) and, in the same call,
    quant_out / quant_scale (/ quant_sum) = invoke_quant(_fuse_sum)(attn.reshape(B, -1)) - bit-identical to the pair
    (llama_w4a8_unpad.py:253-282); one launch where the attention kernel can finish the row itself."""
    for n, t in (("q", q), ("k", k), ("v", v)):
        expect(t, torch.float16, n, contiguous=False)
    expect(kv_pointers, torch.int64, "kv_pointers")
    expect(quant_out, torch.int8, "quant_out")
    expect(quant_scale, torch.float16, "quant_scale")
    if quant_sum is not None:
        expect(quant_sum, torch.float16, "quant_sum")
    batch = kv_pointers.size(0)
    nheads, nheads_kv, headdim = q.size(1), k.size(1), k.size(-1)
    if not (k.stride(2) == 1 and k.stride(1) == headdim and v.stride(2) == 1 and v.stride(1) == headdim):
        raise RuntimeError("k and v must have stride(2) == 1 and stride(1) == head_dim")
    if not (q.stride(2) == 1 and q.stride(1) == headdim):
        raise RuntimeError("q must have stride(2) == 1 and stride(1) == head_dim")
    if length_per_sample is not None:
        expect(length_per_sample, torch.int32, "length_per_sample")
    if quant_out.numel() != q.size(0) * nheads * headdim:
        raise RuntimeError("quant_out must hold batch x heads x head_dim int8 values")
    out = torch.empty((q.size(0), nheads, headdim), dtype=q.dtype, device=q.device)
    with guard(q):
        check(lib.qs_single_query_attention_quant(
            ptr(q), ptr(k), ptr(v), ptr(kv_pointers), ptr(length_per_sample), ptr(out), ptr(quant_out),
            ptr(quant_sum) if quant_sum is not None else 0, ptr(quant_scale), batch, nheads, nheads_kv, headdim,
            q.stride(0), k.stride(0), kv_pointers.size(-1), int(memory_max_seqlen), int(tokens_per_block),
            int(size_per_token), int(timestep), int(rotary_embedding_dim), float(rotary_base),
            int(bool(neox_rotary_style)), int(bool(int4_kv_cache)), int(bool(kv_cache_with_zeros)), stream()),
            "fused.single_query_attention_quant")
    return out


def _gemm_common(per_group, in_feats, kernel, rest):
    expect(in_feats, torch.int8, "in_feats")
    expect(kernel, torch.int8, "kernel")
    for i, t in enumerate(rest):
        expect(t, torch.int8 if per_group and i < 2 else torch.float16, f"operand {i}")
    K = in_feats.size(-1)
    M, N = in_feats.numel() // K, kernel.size(0)
    if kernel.size(1) * 2 != K:
        raise RuntimeError(f"w4a8 gemm: kernel {tuple(kernel.shape)} does not match K={K}")
    return M, N, K


# ---- K-slice planes: (row-parallel GEMM, add + norm + quant) with the GEMM's cross-workgroup reduction and epilogue moved into
# the row kernel that follows it anyway (include/qserve_amd.h, "K-slice PLANES") ------------------------------------------------
_PLANES_PLAN = {}


def gemm_planes_plan(M, N, K, per_group=False):
    """-> number of K slices the planes launch of this shape uses (0: no such launch, run the ordinary pair).  Deterministic
    in the shape (and in the library's A/B switch, which tests change): cached per shape while the switch is at its default."""
    import ctypes as C
    key = (int(M), int(N), int(K), bool(per_group))
    ks = _PLANES_PLAN.get(key)
    if ks is None or _PLANES_PLAN.get("nocache"):
        buf = (C.c_int * 4)()
        check(lib.qs_w4a8_gemm_planes_plan(int(key[3]), key[0], key[1], key[2], C.cast(buf, C.c_void_p)), "fused.gemm_planes_plan")
        ks = int(buf[0])
        _PLANES_PLAN[key] = ks
    return ks


def gemm_planes(in_feats, kernel, planes, zeros=None, scales_i8=None):
    """planes int32 [k_slices, M, N] (k_slices = gemm_planes_plan(M, N, K)): the W4A8 GEMM's partial sums per K slice,
    per-channel (zeros / scales_i8 None) or per-group."""
    per_group = zeros is not None
    M, N, K = _gemm_common(per_group, in_feats, kernel, (zeros, scales_i8) if per_group else ())
    expect(planes, torch.int32, "planes")
    if planes.dim() != 3 or planes.size(1) != M or planes.size(2) != N or not planes.is_contiguous():
        raise RuntimeError(f"gemm_planes: planes {tuple(planes.shape)} for a [{M}, {N}] product")
    ks = gemm_planes_plan(M, N, K, per_group)
    if ks == 0 or planes.size(0) != ks:
        raise RuntimeError(f"gemm_planes: planes {tuple(planes.shape)} but this shape runs as {ks} x [{M}, {N}]")
    with guard(in_feats):
        if per_group:
            check(lib.qs_w4a8_per_group_gemm_planes(ptr(in_feats), ptr(kernel), ptr(zeros), ptr(scales_i8), ptr(planes), M, N, K,
                                                    stream()), "fused.gemm_planes")
        else:
            check(lib.qs_w4a8_per_chn_gemm_planes(ptr(in_feats), ptr(kernel), ptr(planes), M, N, K, stream()), "fused.gemm_planes")


def add_residual_rms_norm_general_planes(out, hidden, planes, wscales, ascales, weight, scaling, epsilon, w_szs=None,
                                         a_ssums=None, input_sum=None):
    """== gemm(..., delta) ; add_residual_rms_norm_general(out, hidden, delta, weight, scaling, epsilon, input_sum) where the GEMM
    ran as gemm_planes: wscales / w_szs [hidden] and ascales / a_ssums [T] are that GEMM's epilogue operands (w_szs and a_ssums
    None together = per-group).  ascales / a_ssums may be the very tensors scaling / input_sum are written to."""
    expect(out, torch.int8, "out")
    expect(hidden, torch.float16, "hidden")
    expect(planes, torch.int32, "planes")
    for n, t in (("wscales", wscales), ("ascales", ascales), ("weight", weight), ("scaling", scaling)):
        expect(t, torch.float16, n)
    if (w_szs is None) != (a_ssums is None):
        raise RuntimeError("add_residual_rms_norm_general_planes: w_szs and a_ssums come together or not at all")
    for n, t in (("w_szs", w_szs), ("a_ssums", a_ssums), ("input_sum", input_sum)):
        if t is not None:
            expect(t, torch.float16, n)
    hid = hidden.size(-1)
    T = hidden.numel() // hid
    if planes.dim() != 3 or planes.size(1) != T or planes.size(2) != hid:
        raise RuntimeError(f"add_residual_rms_norm_general_planes: planes {tuple(planes.shape)} vs hidden [{T}, {hid}]")
    # the C entry takes pointers + a plane stride: everything it will read or write must be there (a wrong-sized tensor would be
    # read or written out of bounds on the device)
    if not planes[0].is_contiguous() or not hidden.is_contiguous() or not out.is_contiguous():
        raise RuntimeError("add_residual_rms_norm_general_planes: planes[i], hidden and out must be contiguous [T, hidden] blocks")
    if out.numel() != T * hid:
        raise RuntimeError(f"add_residual_rms_norm_general_planes: out has {out.numel()} elements, expected {T * hid}")
    for n, t, need in (("wscales", wscales, hid), ("w_szs", w_szs, hid), ("weight", weight, hid), ("ascales", ascales, T),
                       ("a_ssums", a_ssums, T), ("scaling", scaling, T), ("input_sum", input_sum, T)):
        if t is not None and (t.numel() < need or not t.is_contiguous()):
            raise RuntimeError(f"add_residual_rms_norm_general_planes: {n} needs {need} contiguous values, has {t.numel()}")
    with guard(out):
        check(lib.qs_add_residual_rms_norm_general_planes(
            ptr(out), ptr(hidden), ptr(planes), planes.size(0), planes.stride(0), ptr(wscales),
            ptr(w_szs) if w_szs is not None else 0, ptr(ascales), ptr(a_ssums) if a_ssums is not None else 0, ptr(weight),
            ptr(input_sum) if input_sum is not None else 0, ptr(scaling), float(epsilon), T, hid, stream()),
            "fused.add_residual_rms_norm_general_planes")
