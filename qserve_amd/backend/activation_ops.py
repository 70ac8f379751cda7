"""Mirror of `qserve_backend.activation_ops` (kernels/csrc/activation.cpp:25-39)."""
import torch

from ._util import check, expect, guard, lib, ptr, stream


def silu_and_mul(out, input):
    """out f16 [..., d] = silu(input[..., :d]) * input[..., d:]."""
    expect(out, torch.float16, "out")
    expect(input, torch.float16, "input")
    d = input.size(-1) // 2
    with guard(out):
        check(lib.qs_silu_and_mul(ptr(out), ptr(input), input.numel() // input.size(-1), d, stream()),
              "activation_ops.silu_and_mul")


def gelu_new(out, input):
    raise NotImplementedError("gelu_new is not used by the W4A8KV4 models (out of scope)")


def gelu_fast(out, input):
    raise NotImplementedError("gelu_fast is not used by the W4A8KV4 models (out of scope)")


def invoke_dequant_silu_and_mul_quant(*args, **kwargs):
    raise NotImplementedError("invoke_dequant_silu_and_mul_quant belongs to the W8A8 path (out of scope)")
