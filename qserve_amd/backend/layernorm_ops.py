"""Mirror of `qserve_backend.layernorm_ops` (kernels/csrc/layernorm.cpp:47-72)."""
import torch

from ._util import check, expect, guard, lib, ptr, stream


def rms_norm(out, input, weight, epsilon, use_quant=False):
    if use_quant:
        raise NotImplementedError("rms_norm(use_quant=True) belongs to the W8A8 path (out of scope)")
    expect(out, torch.float16, "out")
    expect(input, torch.float16, "input")
    expect(weight, torch.float16, "weight")
    hidden = input.size(-1)
    with guard(out):
        check(lib.qs_rms_norm(ptr(out), ptr(input), ptr(weight), float(epsilon), input.numel() // hidden, hidden,
                              stream()), "layernorm_ops.rms_norm")


def rms_norm_general(out, input, weight, scaling, epsilon, use_per_token_quant=False):
    """layernorm_kernels.cu:427-462; only the per-token branch (the one the W4A8 models use, layers/layernorm.py)."""
    if not use_per_token_quant:
        raise NotImplementedError("rms_norm_general per-tensor scaling belongs to the W8A8 path (out of scope)")
    expect(out, torch.int8, "out")
    expect(input, torch.float16, "input")
    expect(weight, torch.float16, "weight")
    expect(scaling, torch.float16, "scaling")
    hidden = input.size(-1)
    with guard(out):
        check(lib.qs_rms_norm_general(ptr(out), ptr(input), ptr(weight), 0, ptr(scaling), float(epsilon),
                                      input.numel() // hidden, hidden, stream()), "layernorm_ops.rms_norm_general")


def rms_norm_general_fuse_sum(out, input, weight, input_sum, scaling, epsilon, use_per_token_quant=False):
    """layernorm_kernels.cu:464-508 (per-tensor branch asserts in the reference)."""
    if not use_per_token_quant:
        raise NotImplementedError("rms_norm_general_fuse_sum has no per-tensor variant (the reference asserts)")
    expect(out, torch.int8, "out")
    expect(input, torch.float16, "input")
    expect(weight, torch.float16, "weight")
    expect(input_sum, torch.float16, "input_sum")
    expect(scaling, torch.float16, "scaling")
    hidden = input.size(-1)
    with guard(out):
        check(lib.qs_rms_norm_general(ptr(out), ptr(input), ptr(weight), ptr(input_sum), ptr(scaling), float(epsilon),
                                      input.numel() // hidden, hidden, stream()),
              "layernorm_ops.rms_norm_general_fuse_sum")


def invoke_dequant_add_residual_rms_norm_quant(*args, **kwargs):
    raise NotImplementedError("invoke_dequant_add_residual_rms_norm_quant belongs to the W8A8 path (out of scope)")
