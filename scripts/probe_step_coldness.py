#!/usr/bin/env python3
"""Round 5 probe (timing only, results wrong by design in the hacked runs): what keeps the row kernels at ~4.6 us inside the decode
step when they take 3.1-3.5 us in their own / pair chains?  The same engine, graph-captured, with (a) nothing changed, (b) every
layer sharing layer 0's norm weights (no cold 8 KB vector per row launch), (c) every layer sharing layer 0's GEMM weights too
(everything Infinity-Cache-resident: an upper bound on 'cold lines')."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qserve_amd import decode as D


def timed(eng, steps=32):
    eng.lengths.fill_(1025)                   # every run over the same contexts (1025 .. 1025 + 2 + 8 + 3 * 32 < 1024 + max_new)
    eng.capture()
    for _ in range(8):
        eng.graph.replay()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            eng.graph.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / steps)
    return sorted(res)[1]


cfg = D.LLAMA3_8B if hasattr(D, "LLAMA3_8B") else D.CONFIGS["llama3-8b"]
eng = D.DecodeEngine(cfg, batch=64, prompt_len=1024, max_new=256, device="cuda:0", seed=0, fuse_pairs=True)
eng.prefill_cache(1024)
print(f"as shipped:                      {timed(eng):.4f} ms per step")
print(f"as shipped, again:               {timed(eng):.4f} ms per step")
for L in eng.layers:
    L["ln1"], L["ln2"] = eng.layers[0]["ln1"], eng.layers[0]["ln2"]
print(f"one norm-weight vector for all:  {timed(eng):.4f} ms per step")
for L in eng.layers[1:]:
    for k in ("qkv", "o", "gate_up", "down"):
        L[k] = eng.layers[0][k]
print(f"+ one set of GEMM weights:       {timed(eng):.4f} ms per step")
