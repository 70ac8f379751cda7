#!/usr/bin/env python3
"""lm_head (fp16, 64 x 128256 x 4096) through torch's BLAS in its possible call forms: us per launch and TB/s of the 1.05 GB
weight stream (hipGraph-captured, weights rotated over 3 copies so the Infinity Cache cannot serve them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = "cuda:0"
V, K, B = 128256, 4096, int(os.environ.get("B", "64"))
Ws = [(torch.randn((V, K), device=dev) * 0.02).half() for _ in range(3)]
x = torch.randn((B, K), device=dev).half()
xt = x.t().contiguous()


def timeit(fn, reps=6, replays=3):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(reps):
                fn(i)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)


out = torch.empty((B, V), dtype=torch.float16, device=dev)
outT = torch.empty((V, B), dtype=torch.float16, device=dev)
forms = {
    "x @ W.t()": lambda i: torch.matmul(x, Ws[i % 3].t()),
    "x @ W.t() (out=)": lambda i: torch.matmul(x, Ws[i % 3].t(), out=out),
    "F.linear(x, W)": lambda i: torch.nn.functional.linear(x, Ws[i % 3]),
    "W @ x.t()  -> [V, B]": lambda i: torch.matmul(Ws[i % 3], x.t()),
    "W @ xt (contiguous [K, B])": lambda i: torch.matmul(Ws[i % 3], xt, out=outT),
    "addmm-free mm(x, W.t())": lambda i: torch.mm(x, Ws[i % 3].t()),
}
for name, fn in forms.items():
    try:
        us = timeit(fn)
        print(f"{name:34s} {us:8.1f} us  {V * K * 2 / us / 1e6:6.2f} TB/s", flush=True)
    except Exception as e:
        print(f"{name:34s} failed: {type(e).__name__}: {str(e).splitlines()[0]}")
