#!/usr/bin/env python3
"""invoke_quant_fuse_sum at the prompt-phase shape [65536, 14336] and the decode shape [64, 14336] (graph-timed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qserve_backend.fused_kernels as fk
dev = torch.device("cuda:0")
for T, H in ((65536, 14336), (65536, 4096), (64, 14336), (64, 4096)):
    x = torch.randn((T, H), dtype=torch.float16, device=dev)
    q = torch.empty((T, H), dtype=torch.int8, device=dev)
    sc, sm = torch.empty((T,), dtype=torch.float16, device=dev), torch.empty((T,), dtype=torch.float16, device=dev)
    for fn, name in ((lambda: fk.invoke_quant_fuse_sum(q, x, sm, sc), "fuse_sum"), (lambda: fk.invoke_quant(q, x, sc), "plain")):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"invoke_quant {name:8s} [{T}, {H}]: {e0.elapsed_time(e1) * 1e3 / 20:9.1f} us")
