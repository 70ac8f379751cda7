#!/usr/bin/env python3
"""Row-kernel timing for library A/Bs (library chosen by QS_AMD_LIBRARY): add_residual + general norm + quant over [B, 4096] and
invoke_quant over [B, 14336] (the three row launches of a Llama-3-8B decode layer), each alone in a hipGraph of 64 launches over
rotating buffers, and alternating with a 9 MB streaming GEMM (what precedes / follows them in the step); median of 5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qserve_amd import fused as fz
import qserve_backend.fused_kernels as fk
import qserve_backend.qgemm_w4a8_per_chn as gc
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "64"))
g = torch.Generator(device=dev).manual_seed(0)
NB = 8


def timeit(fn, reps=64, replays=8):
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


hid = [torch.randn((B, 4096), device=dev, generator=g).half() for _ in range(NB)]
dl = [torch.randn((B, 4096), device=dev, generator=g).half() for _ in range(NB)]
gam = torch.randn((4096,), device=dev, generator=g).half()
q8 = torch.empty((B, 4096), dtype=torch.int8, device=dev)
sc, sm = torch.empty((B,), dtype=torch.float16, device=dev), torch.empty((B,), dtype=torch.float16, device=dev)
act = [torch.randn((B, 14336), device=dev, generator=g).half() for _ in range(NB)]
q14 = torch.empty((B, 14336), dtype=torch.int8, device=dev)
A = torch.randint(-127, 128, (B, 4096), dtype=torch.int8, device=dev, generator=g)
W = [torch.randint(-128, 128, (4096, 2048), dtype=torch.int8, device=dev, generator=g) for _ in range(NB)]
ws = (torch.rand((4096,), device=dev, generator=g) * 0.01).half()
sa = (torch.rand((B,), device=dev, generator=g) * 0.01).half()
og = torch.empty((B, 4096), dtype=torch.float16, device=dev)
norm = lambda i: fz.add_residual_rms_norm_general(q8, hid[i % NB], dl[i % NB], gam, sc, 1e-5, sm)
quant = lambda i: fk.invoke_quant_fuse_sum(q14, act[i % NB], sm, sc)
gemm = lambda i: gc.gemm_forward_cuda(A, W[i % NB], ws, sa, ws, sa, og)
t_g = timeit(gemm)
t_n = timeit(norm)
t_q = timeit(quant)


def pair(row):
    def f(i):
        gemm(i)
        row(i)
    return f


print(f"B={B}: add+norm+quant {t_n:5.2f} us   quant[14336] {t_q:5.2f} us   o-GEMM {t_g:5.2f}   GEMM+norm pair {timeit(pair(norm), 32):5.2f}   "
      f"GEMM+quant pair {timeit(pair(quant), 32):5.2f} us")
