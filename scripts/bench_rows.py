#!/usr/bin/env python3
"""Launch-to-launch period of the row kernels at decode batch size (64 tokens), 200 dependent launches per hipGraph,
inputs rotating over 32 buffers (like a previous kernel's output: not in this CU's L2).  Compare with
scripts/microbench_rowlat.hip (empty kernel 1.6 us; 3 inputs + 3 block reductions + one store: 3.5 us)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qserve_amd import fused as fz
import qserve_backend.fused_kernels as fk
import qserve_backend.layernorm_ops as ln
dev = torch.device("cuda:0")
B, H, D, NB = 64, 4096, 14336, 32
g = torch.Generator(device=dev).manual_seed(0)
hs = [torch.randn((B, H), device=dev, generator=g).half() for _ in range(NB)]
ds = [torch.randn((B, H), device=dev, generator=g).half() for _ in range(NB)]
gu = [torch.randn((B, 2 * D), device=dev, generator=g).half() for _ in range(NB)]
w = torch.rand((H,), device=dev, generator=g).half() + 0.5
q = torch.empty((B, H), dtype=torch.int8, device=dev)
qm = torch.empty((B, D), dtype=torch.int8, device=dev)
sc = torch.empty((B,), dtype=torch.float16, device=dev)
sm = torch.empty((B,), dtype=torch.float16, device=dev)


def period(fn, n=200):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=s):
            for i in range(n):
                fn(i)
    torch.cuda.synchronize()
    gph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


print(f"add_residual + rms_norm_general_fuse_sum (fused)  {period(lambda i: fz.add_residual_rms_norm_general(q, hs[i % NB], ds[i % NB], w, sc, 1e-5, sm)):6.2f} us")
print(f"rms_norm_general_fuse_sum                         {period(lambda i: ln.rms_norm_general_fuse_sum(q, hs[i % NB], w, sm, sc, 1e-5, True)):6.2f} us")
print(f"invoke_quant_fuse_sum                             {period(lambda i: fk.invoke_quant_fuse_sum(q, hs[i % NB], sm, sc)):6.2f} us")
print(f"silu_and_mul + quant (fused, d = {D})           {period(lambda i: fz.silu_and_mul_quant(qm, gu[i % NB], sc, sm)):6.2f} us")


# instruction-cache check: different kernels alternating vs the same kernel repeated
def mix2(i):
    if i & 1:
        fk.invoke_quant_fuse_sum(q, hs[i % NB], sm, sc)
    else:
        fz.add_residual_rms_norm_general(q, hs[i % NB], ds[i % NB], w, sc, 1e-5, sm)


def mix4(i):
    k = i & 3
    if k == 0:
        fz.add_residual_rms_norm_general(q, hs[i % NB], ds[i % NB], w, sc, 1e-5, sm)
    elif k == 1:
        fk.invoke_quant_fuse_sum(q, hs[i % NB], sm, sc)
    elif k == 2:
        ln.rms_norm_general_fuse_sum(q, hs[i % NB], w, sm, sc, 1e-5, True)
    else:
        fz.silu_and_mul_quant(qm, gu[i % NB], sc, sm)


print(f"alternating add+norm+quant / quant                {period(mix2):6.2f} us (mean of the two alone: {(3.72 + 2.25) / 2:.2f})")
print(f"cycling 4 different row kernels                   {period(mix4):6.2f} us (mean of the four alone: {(3.72 + 2.25 + 3.36 + 6.22) / 4:.2f})")
