"""Mirror of `qserve_backend.qgemm_w8a8` (kernels/csrc/qgemm/w8a8/pybind.cpp)."""
import torch

from ._util import check, expect, guard, lib, ptr, stream


def w8a8_gemm_forward_cuda(in_feats, kernel, wscales, ascales, out_feats):
    """w8a8_gemm_cuda.h:11 -- kernel int8 [N,K]."""
    expect(in_feats, torch.int8, "in_feats")
    expect(kernel, torch.int8, "kernel")
    expect(wscales, torch.float16, "wscales")
    expect(ascales, torch.float16, "ascales")
    expect(out_feats, torch.float16, "out_feats")
    M, N, K = out_feats.size(-2), out_feats.size(-1), in_feats.size(1)
    with guard(in_feats):
        check(lib.qs_w8a8_gemm(ptr(in_feats), ptr(kernel), ptr(wscales), ptr(ascales), ptr(out_feats), M, N, K, stream()),
              "qgemm_w8a8.w8a8_gemm_forward_cuda")
