#!/usr/bin/env python3
"""The fp16 lm_head GEMM of the decode step ([64, 4096] x [128256, 4096]^T) as torch issues it; PYTORCH_TUNABLEOP_ENABLED=1
lets torch time the rocBLAS / hipBLASLt solutions for the shape and keep the best.  env: B."""
import os, sys, time
import torch
dev = torch.device("cuda:0")
B, H, V = int(os.environ.get("B", "64")), 4096, 128256
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((B, H), device=dev, dtype=torch.float16, generator=g)
ws = [(torch.randn((V, H), device=dev, generator=g) * 0.02).half() for _ in range(4)]     # rotate: 4 x 1 GB (no cache reuse)
for i in range(8):
    torch.matmul(x, ws[i % 4].t())
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph, stream=s):
        for i in range(8):
            y = torch.matmul(x, ws[i % 4].t())
torch.cuda.synchronize()
gph.replay(); torch.cuda.synchronize()
res = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        gph.replay()
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) * 1e3 / 32)
us = sorted(res)[2]
print(f"lm_head B={B}: {us:.1f} us  {V * H * 2 / us / 1e6:.2f} TB/s  tunableop={os.environ.get('PYTORCH_TUNABLEOP_ENABLED', '0')}")
