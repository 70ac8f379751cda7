"""Mirror of `qserve_backend.qgemm_w4a8_per_chn` (kernels/csrc/qgemm/w4a8_per_chn/pybind.cpp:13-16)."""
import torch

from ._util import check, expect, guard, lib, ptr, stream


def gemm_forward_cuda(in_feats, kernel, wscales, ascales, w_szs, a_ssums, out_feats):
    """gemm_cuda.h:11 -- in_feats int8 [M,K], kernel int8 [N,K/2], wscales/w_szs f16 [N], ascales/a_ssums f16 [M],
    out_feats f16 [..., M, N] written in place.  Shapes are taken as the reference takes them
    (gemm_cuda.cu:604-613): M = out.size(-2), N = out.size(-1), K = in_feats.size(1)."""
    expect(in_feats, torch.int8, "in_feats")
    expect(kernel, torch.int8, "kernel")
    expect(out_feats, torch.float16, "out_feats")
    for n, t in (("wscales", wscales), ("ascales", ascales), ("w_szs", w_szs), ("a_ssums", a_ssums)):
        expect(t, torch.float16, n)
    M, N, K = out_feats.size(-2), out_feats.size(-1), in_feats.size(1)
    with guard(in_feats):
        check(lib.qs_w4a8_per_chn_gemm(ptr(in_feats), ptr(kernel), ptr(wscales), ptr(ascales), ptr(w_szs), ptr(a_ssums),
                                       ptr(out_feats), M, N, K, stream()), "qgemm_w4a8_per_chn.gemm_forward_cuda")


def gemm_forward_acc(in_feats, kernel, acc_out):
    """Parity/debug entry point (not in the reference): raw int32 accumulators -> acc_out int32 [M,N]."""
    expect(in_feats, torch.int8, "in_feats")
    expect(kernel, torch.int8, "kernel")
    expect(acc_out, torch.int32, "acc_out")
    M, N, K = acc_out.size(-2), acc_out.size(-1), in_feats.size(1)
    with guard(in_feats):
        check(lib.qs_w4a8_per_chn_gemm_acc(ptr(in_feats), ptr(kernel), ptr(acc_out), M, N, K, stream()),
              "qgemm_w4a8_per_chn.gemm_forward_acc")
