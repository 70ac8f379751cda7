"""Mirror of `qserve_backend.qgemm_w4a8_per_group` (kernels/csrc/qgemm/w4a8_per_group/pybind.cpp:13-16)."""
import torch

from ._util import check, expect, guard, lib, ptr, stream


def gemm_forward_cuda(in_feats, kernel, zeros, scales_i8, wscales, ascales, out_feats):
    """gemm_cuda.h:11 -- zeros / scales_i8 int8 [K/128, N] (reference permutation), wscales f16 [N], ascales f16 [M]."""
    expect(in_feats, torch.int8, "in_feats")
    expect(kernel, torch.int8, "kernel")
    expect(zeros, torch.int8, "zeros")
    expect(scales_i8, torch.int8, "scales_i8")
    expect(wscales, torch.float16, "wscales")
    expect(ascales, torch.float16, "ascales")
    expect(out_feats, torch.float16, "out_feats")
    M, N, K = out_feats.size(-2), out_feats.size(-1), in_feats.size(1)
    with guard(in_feats):
        check(lib.qs_w4a8_per_group_gemm(ptr(in_feats), ptr(kernel), ptr(zeros), ptr(scales_i8), ptr(wscales),
                                         ptr(ascales), ptr(out_feats), M, N, K, stream()),
              "qgemm_w4a8_per_group.gemm_forward_cuda")


def gemm_forward_acc(in_feats, kernel, zeros, scales_i8, acc_out):
    """Parity/debug entry point (not in the reference): raw int32 accumulators."""
    expect(in_feats, torch.int8, "in_feats")
    expect(kernel, torch.int8, "kernel")
    expect(zeros, torch.int8, "zeros")
    expect(scales_i8, torch.int8, "scales_i8")
    expect(acc_out, torch.int32, "acc_out")
    M, N, K = acc_out.size(-2), acc_out.size(-1), in_feats.size(1)
    with guard(in_feats):
        check(lib.qs_w4a8_per_group_gemm_acc(ptr(in_feats), ptr(kernel), ptr(zeros), ptr(scales_i8), ptr(acc_out), M, N,
                                             K, stream()), "qgemm_w4a8_per_group.gemm_forward_acc")
