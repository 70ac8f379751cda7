#!/usr/bin/env python3
"""fp16 lm_head GEMM [B, 4096] x [128256, 4096]^T: default hipBLASLt choice vs PyTorch TunableOp (env TUNE=1), hipGraph-timed."""
import os, sys
if os.environ.get("TUNE"):
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.environ.get("TUNE_FILE", "/tmp/tunableop_lm_head.csv")
    os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "200")
import torch
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "64"))
x = torch.randn((B, 4096), device=dev, dtype=torch.float16)
W = [(torch.randn((128256, 4096), device=dev, dtype=torch.float16) * 0.02) for _ in range(2)]
for i in range(3):
    y = torch.matmul(x, W[i % 2].t())
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for i in range(8):
            y = torch.matmul(x, W[i % 2].t())
torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
res = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): g.replay()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) * 1e3 / 32)
us = sorted(res)[2]
print(f"lm_head B={B} {'tuned' if os.environ.get('TUNE') else 'default'}: {us:.1f} us = {128256*4096*2/us/1e6:.2f} TB/s")

