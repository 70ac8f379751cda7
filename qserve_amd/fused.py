"""Pair fusions for the decode loop (extensions; NOT part of the reference's `qserve_backend` surface).

Each function is bit-identical to the two reference ops it replaces (tests/test_fused_gpu.py checks that on the GPU);
they exist because at decode batch sizes each row kernel is a fixed ~5 us latency chain, so a pair costs one.

    add_residual_rms_norm_general(_fuse_sum)  ==  hidden += delta ; layernorm_ops.rms_norm_general(_fuse_sum)(hidden)
    silu_and_mul_quant(_fuse_sum)             ==  activation_ops.silu_and_mul ; fused_kernels.invoke_quant(_fuse_sum)
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
