#!/usr/bin/env python3
"""Compute-bound W4A8 GEMM: the eight-wave tile (qs_set_gemm_variant 3001) against the four-wave tile of round 5 (3003 / the
dispatcher's default), alternating inside ONE process (box-to-box spread of this kernel reaches 15 %), medians of ROUNDS rounds.
usage: bench_wide_ab.py [MxNxK ...]   env ROUNDS (default 5), MODES=chn,grp, ACT=1 adds gate_up + silu*mul for N % 512 == 0,
VARIANTS=3001,3002,3003 (round 6): any list of forced variants instead of the pair (3002 = the 128-token tile); the two-variant
summary line is printed for the first and the last of the list"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qserve_backend.qgemm_w4a8_per_chn as op
import qserve_backend.qgemm_w4a8_per_group as opg
from qserve_amd import fused as fz
from qserve_amd._lib import lib

dev = torch.device("cuda:0")
ROUNDS = int(os.environ.get("ROUNDS", "5"))
MODES = os.environ.get("MODES", "chn,grp").split(",")


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


shapes = [(4096, 4096, 4096), (8192, 28672, 4096), (8192, 4096, 14336), (8192, 6144, 4096), (65536, 6144, 4096)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in s.split("x")) for s in sys.argv[1:]]
for M, N, K in shapes:
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=dev)
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    ws = torch.rand((N,), device=dev).half() * 0.01
    sa = torch.rand((M,), device=dev).half() * 0.01
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    z = torch.randint(-8, 8, (K // 128, N), dtype=torch.int8, device=dev)
    s8 = torch.randint(1, 9, (K // 128, N), dtype=torch.int8, device=dev)
    fns = {"chn": lambda: op.gemm_forward_cuda(A, W, ws, sa, ws, sa, out),
           "grp": lambda: opg.gemm_forward_cuda(A, W, z, s8, ws, sa, out)}
    if os.environ.get("ACT") == "1" and N % 512 == 0:
        act = torch.empty((M, N // 2), dtype=torch.float16, device=dev)
        fns["chn+act"] = lambda: fz.gemm_silu_and_mul_per_chn(A, W, ws, sa, ws, sa, act, None)
    for mode, fn in fns.items():
        if mode.split("+")[0] not in MODES:
            continue
        VARS = [int(x) for x in os.environ.get("VARIANTS", "3001,3003").split(",")]
        t = {v: [] for v in VARS}
        for _ in range(ROUNDS):
            for v in VARS:
                lib.qs_set_gemm_variant(v)
                t[v].append(timeit(fn))
        lib.qs_set_gemm_variant(-1)
        if len(VARS) != 2 or VARS != [3001, 3003]:
            print(f"M={M:6d} N={N:6d} K={K:6d} {mode:8s}: " + "  ".join(
                f"{v}: {statistics.median(t[v]):8.1f} us ({2.0 * M * N * K / statistics.median(t[v]) / 1e6:6.0f} TOPS)" for v in VARS), flush=True)
            continue
        a, b = statistics.median(t[3001]), statistics.median(t[3003])
        tops = lambda us: 2.0 * M * N * K / us / 1e6
        print(f"M={M:6d} N={N:6d} K={K:6d} {mode:8s}: eight-wave {a:9.1f} us {tops(a):7.1f} TOPS | four-wave {b:9.1f} us {tops(b):7.1f} TOPS "
              f"({tops(b) / 5000 * 100:4.1f}% of 5 POPS) | ratio {a / b:5.3f}   [{' '.join(f'{x:.1f}' for x in t[3001])}] "
              f"[{' '.join(f'{x:.1f}' for x in t[3003])}]", flush=True)
