#!/usr/bin/env python3
"""Compute-bound W4A8 GEMM shapes (prefill / BASELINE config 1): TOPS and fraction of the int8 MFMA peak.
usage: bench_gemm_big.py [MxNxK ...]   env QS_GEMM_VARIANT=<int> forces a kernel variant (qs_set_gemm_variant)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qserve_backend.qgemm_w4a8_per_chn as op
import qserve_backend.qgemm_w4a8_per_group as opg
from qserve_amd._lib import lib

PEAK_TOPS = 5000.0   # MI355X dense int8 MFMA (256 CU x 4 SIMD x 2048 ops/clk x 2.4 GHz ~ 5.0 POPS)
dev = torch.device("cuda:0")


def timeit(fn, reps=4, replays=3):
    """GPU-side time per launch: launches captured in a hipGraph, timed with events around the replays."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=s):
            for _ in range(reps):
                fn()
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


shapes = [(4096, 4096, 4096), (1024, 4096, 4096), (256, 4096, 4096), (8192, 6144, 4096), (8192, 28672, 4096), (8192, 4096, 14336)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in s.split("x")) for s in sys.argv[1:]]
if "QS_GEMM_VARIANT" in os.environ:
    lib.qs_set_gemm_variant(int(os.environ["QS_GEMM_VARIANT"]))
for M, N, K in shapes:
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=dev)
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    ws = torch.rand((N,), device=dev).half() * 0.01
    sa = torch.rand((M,), device=dev).half() * 0.01
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    us = timeit(lambda: op.gemm_forward_cuda(A, W, ws, sa, ws, sa, out))
    tops = 2.0 * M * N * K / us / 1e6
    z = torch.randint(-8, 8, (K // 128, N), dtype=torch.int8, device=dev)
    s8 = torch.randint(1, 9, (K // 128, N), dtype=torch.int8, device=dev)
    usg = timeit(lambda: opg.gemm_forward_cuda(A, W, z, s8, ws, sa, out))
    print(f"M={M:6d} N={N:6d} K={K:6d}: per-channel {us:9.1f} us {tops:7.1f} TOPS ({tops / PEAK_TOPS * 100:4.1f}% of {PEAK_TOPS:.0f})"
          f"   per-group {usg:9.1f} us {2.0 * M * N * K / usg / 1e6:7.1f} TOPS", flush=True)
