#!/usr/bin/env python3
"""Decode-shape W4A8 GEMM timing for library A/Bs (the library is chosen by QS_AMD_LIBRARY): the four GEMMs of a Llama-3-8B
layer at M tokens (env M, default 64; MODE=group for g128), weights rotating over 8 sets, launches captured in a hipGraph,
median of 5 measurements."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qserve_backend.qgemm_w4a8_per_chn as gc
import qserve_backend.qgemm_w4a8_per_group as gg
from qserve_amd import fused as fz
dev = torch.device("cuda:0")
M = int(os.environ.get("M", "64"))
group = os.environ.get("MODE", "") == "group"
g = torch.Generator(device=dev).manual_seed(0)
NL = 8


def timeit(fn, reps=32, replays=8):
    for i in range(3):
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
    res = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            gph.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / (reps * replays))
    return sorted(res)[2]


row = []
for name, N, K, act in (("qkv", 6144, 4096, False), ("o", 4096, 4096, False), ("gate_up+silu*mul", 28672, 4096, True), ("down", 4096, 14336, False)):
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
    W = [torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=dev, generator=g) for _ in range(NL)]
    ws = (torch.rand((N,), device=dev, generator=g) * 0.01).half()
    sa = (torch.rand((M,), device=dev, generator=g) * 0.01).half()
    out = torch.empty((M, N // 2 if act else N), dtype=torch.float16, device=dev)
    tmp = torch.empty((M, N), dtype=torch.float16, device=dev)
    if group:
        Z = torch.randint(-8, 8, (K // 128, N), dtype=torch.int8, device=dev, generator=g)
        S = torch.randint(1, 8, (K // 128, N), dtype=torch.int8, device=dev, generator=g)
        if act:
            fn = lambda i: fz.gemm_silu_and_mul_per_group(A, W[i % NL], Z, S, ws, sa, out, tmp)
        else:
            fn = lambda i: gg.gemm_forward_cuda(A, W[i % NL], Z, S, ws, sa, out)
    else:
        if act:
            fn = lambda i: fz.gemm_silu_and_mul_per_chn(A, W[i % NL], ws, sa, ws, sa, out, tmp)
        else:
            fn = lambda i: gc.gemm_forward_cuda(A, W[i % NL], ws, sa, ws, sa, out)
    vs = os.environ.get("VARIANT_" + name.split("+")[0])   # e.g. VARIANT_down=4221,4422: forced geometries (qs_set_gemm_variant)
    if vs:
        from qserve_amd._lib import lib
        res = []
        for v in vs.split(","):
            lib.qs_set_gemm_variant(int(v))
            res.append(f"{v}: {timeit(fn):.2f}")
            lib.qs_set_gemm_variant(-1)
        row.append(f"{name} [" + " ".join(res) + "]")
        continue
    row.append(f"{name} {timeit(fn):6.2f}")
print(f"M={M}{' g128' if group else ''}: " + "   ".join(row) + " us")
