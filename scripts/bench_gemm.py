#!/usr/bin/env python3
"""A/B micro-benchmark of the W4A8 GEMM kernel variants (within one process, interleaved).
usage: python scripts/bench_gemm.py [M] [variants...]   variant codes: see qs_set_gemm_variant / gemm_w4a8.hip"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import qserve_backend.qgemm_w4a8_per_chn as op
import qserve_backend.qgemm_w4a8_per_group as opg
from qserve_amd._lib import lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
variants = [int(v) for v in sys.argv[2:]] or [-1]
dev = torch.device("cuda:0")
shapes = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]
NBUF = 8
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, reps=32, replays=6):
    """GPU-side time per launch: the launches are captured in a hipGraph (no host overhead between them)."""
    for i in range(2):
        fn(i)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=s):
            for i in range(reps):
                fn(i)
    torch.cuda.synchronize()
    gph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        gph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)


if __name__ == "__main__":
    print(f"M={M}   columns: variant -> us (GB/s of weight bytes)")
    for name, N, K in shapes:
        Ws = [torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=dev, generator=g) for _ in range(NBUF)]
        A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
        ws = torch.rand((N,), device=dev).half() * 0.01
        sa = torch.rand((M,), device=dev).half() * 0.01
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        row = []
        for v in variants:
            lib.qs_set_gemm_variant(v)
            us = timeit(lambda i: op.gemm_forward_cuda(A, Ws[i % NBUF], ws, sa, ws, sa, out))
            row.append(f"{v}: {us:6.2f} ({N * K / 2 / us / 1e3:6.0f})")
        lib.qs_set_gemm_variant(-1)
        print(f"{name:8s} N={N:5d} K={K:5d}  " + "   ".join(row))
