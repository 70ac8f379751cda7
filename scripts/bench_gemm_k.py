#!/usr/bin/env python3
"""GPU time of the W4A8 per-channel GEMM vs K (fixed N, M) -> fixed cost and per-k-step cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qserve_backend.qgemm_w4a8_per_chn as op
from qserve_amd._lib import lib
from bench_gemm import timeit  # noqa
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
var = int(sys.argv[3]) if len(sys.argv) > 3 else 1014
lib.qs_set_gemm_variant(var)
for K in (128, 512, 1024, 2048, 4096, 8192, 14336):
    Ws = [torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=dev) for _ in range(8)]
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    ws = torch.rand((N,), device=dev).half() * 0.01
    sa = torch.rand((M,), device=dev).half() * 0.01
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    us = timeit(lambda i: op.gemm_forward_cuda(A, Ws[i % 8], ws, sa, ws, sa, out))
    print(f"M={M} N={N} K={K:6d} variant={var}: {us:7.2f} us   ({K // 128} k-steps, {us / (K // 128):.2f} us/step)")
