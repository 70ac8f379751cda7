#!/usr/bin/env python3
"""Tile order A/B of the compute-bound kernels inside one process: qs_set_gemm_variant(3200 + o), o = 3 plain super-tiles, 0 super-tiles with
the XCD-aware 4 x 8 placement (the default since round 5); per-channel (eight-wave tile) and per-group (four-wave tile); medians of ROUNDS alternating rounds."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import qserve_backend.qgemm_w4a8_per_chn as op
import qserve_backend.qgemm_w4a8_per_group as opg
from qserve_amd._lib import lib

dev = torch.device("cuda:0")
shapes = [(4096, 4096, 4096), (8192, 28672, 4096), (8192, 4096, 14336), (65536, 6144, 4096)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in s.split("x")) for s in sys.argv[1:]]
for M, N, K in shapes:
    nset = 4 if M * K + N * K // 2 < (64 << 20) else 1
    A = [torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev) for _ in range(nset)]
    W = [torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=dev) for _ in range(nset)]
    ws = torch.rand((N,), device=dev).half() * 0.01
    sa = torch.rand((M,), device=dev).half() * 0.01
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    z = torch.randint(-8, 8, (K // 128, N), dtype=torch.int8, device=dev)
    s8 = torch.randint(1, 9, (K // 128, N), dtype=torch.int8, device=dev)
    for mode, fn in (("chn", lambda i: op.gemm_forward_cuda(A[i % nset], W[i % nset], ws, sa, ws, sa, out)),
                     ("grp", lambda i: opg.gemm_forward_cuda(A[i % nset], W[i % nset], z, s8, ws, sa, out))):
        t = {3: [], 0: []}
        for _ in range(int(os.environ.get("ROUNDS", "5"))):
            for o in (3, 0):
                lib.qs_set_gemm_variant(3200 + o)
                t[o].append(bench.time_kernel(fn, 8, torch))
        lib.qs_set_gemm_variant(3200)
        a, b = statistics.median(t[3]), statistics.median(t[0])
        print(f"M={M:6d} N={N:6d} K={K:6d} {mode}: super-tiles {a:9.1f} us | XCD-aware {b:9.1f} us | ratio {a / b:5.3f}  "
              f"[{' '.join(f'{x:.1f}' for x in t[3])}] [{' '.join(f'{x:.1f}' for x in t[0])}]", flush=True)
