#!/usr/bin/env python3
"""bench_wide_variant.py V1,V2,... MxNxK ...: per-channel compute-bound GEMM under the four-wave tile (3003) with a timing switch
(3400 + bits, QS_TIMING library) set on top; prints us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qserve_backend.qgemm_w4a8_per_chn as op
from qserve_amd._lib import lib

dev = torch.device("cuda:0")


def timeit(fn, reps=4, replays=3):
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


shapes = [tuple(int(x) for x in sh.split("x")) for sh in sys.argv[2:]]
data = {}
for (M, N, K) in shapes:
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=dev)
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    ws = torch.rand((N,), device=dev).half() * 0.01
    sa = torch.rand((M,), device=dev).half() * 0.01
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    data[(M, N, K)] = (A, W, ws, sa, out)
for v in (int(x) for x in sys.argv[1].split(",")):
    lib.qs_set_gemm_variant(3003)
    lib.qs_set_gemm_variant(3400)
    if v != 3003:
        lib.qs_set_gemm_variant(v)
    for (M, N, K), (A, W, ws, sa, out) in data.items():
        ts = [timeit(lambda: op.gemm_forward_cuda(A, W, ws, sa, ws, sa, out)) for _ in range(3)]
        print(f"variant {v} M={M} N={N} K={K}: " + " ".join(f"{t:8.1f}" for t in ts) + " us", flush=True)
lib.qs_set_gemm_variant(3400)
lib.qs_set_gemm_variant(-1)
