"""Mirror of `qserve_backend.fused_kernels` (kernels/csrc/fused.cpp:47-71) -- per-token quantisation ops.

Only the per-token overloads used by the W4A8 models are implemented (llama_w4a8_unpad.py:170-183,
layers/activation.py:57,70); the per-tensor overloads and the dequant ops belong to the W8A8 path."""
import torch

from ._util import check, expect, guard, lib, ptr, stream


def invoke_quant(out, input, scale):
    """fused.cpp: invoke_quant(out int8 [T,H], input f16 [T,H], scale f16 [T]) -- per-token dynamic int8."""
    if not isinstance(scale, torch.Tensor):
        raise NotImplementedError("invoke_quant with a per-tensor scalar scale (W8A8 path) is out of scope")
    expect(out, torch.int8, "out")
    expect(input, torch.float16, "input")
    expect(scale, torch.float16, "scale")
    hidden = input.size(-1)
    with guard(out):
        check(lib.qs_invoke_quant(ptr(out), ptr(input), 0, ptr(scale), input.numel() // hidden, hidden, stream()),
              "fused_kernels.invoke_quant")


def invoke_quant_fuse_sum(out, input, input_sum, scale):
    """fused.cpp: invoke_quant_fuse_sum(out, input, input_sum f16 [T], scale f16 [T])."""
    if not isinstance(scale, torch.Tensor):
        raise NotImplementedError("invoke_quant_fuse_sum with a per-tensor scalar scale is out of scope")
    expect(out, torch.int8, "out")
    expect(input, torch.float16, "input")
    expect(input_sum, torch.float16, "input_sum")
    expect(scale, torch.float16, "scale")
    hidden = input.size(-1)
    with guard(out):
        check(lib.qs_invoke_quant(ptr(out), ptr(input), ptr(input_sum), ptr(scale), input.numel() // hidden, hidden,
                                  stream()), "fused_kernels.invoke_quant_fuse_sum")


def invoke_dequant(*args, **kwargs):
    raise NotImplementedError("invoke_dequant belongs to the W8A8 path (out of scope, SURVEY.md 2 row 6)")


def invoke_dequant_add_residual(*args, **kwargs):
    raise NotImplementedError("invoke_dequant_add_residual belongs to the W8A8 path (out of scope)")
