#!/usr/bin/env python3
"""Round 5 probe (timing only, results wrong by design in the hacked runs): what would Infinity-Cache-resident weights be worth to the
decode step, GEMM by GEMM?  The same engine, graph-captured, with the layers sharing layer 0's tensors of one kind (every launch of
that kind then streams bytes that were read 78 us earlier and are still in the 256 MB cache) - the upper bound of any scheme that
touches a GEMM's weights ahead of its launch."""
import os, sys
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


eng = D.DecodeEngine(D.LLAMA3_8B, batch=64, prompt_len=1024, max_new=256, device="cuda:0", seed=0, fuse_pairs=True)
eng.prefill_cache(1024)
own = [dict(L) for L in eng.layers]


def share(kinds):
    for L, o in zip(eng.layers, own):
        for k in ("qkv", "o", "gate_up", "down", "ln1", "ln2"):
            L[k] = eng.layers[0][k] if k in kinds else o[k]
    eng.layers[0].update(own[0])


base = timed(eng)
print(f"as shipped:                          {base:.4f} ms per step")
for kinds in (("ln1", "ln2"), ("qkv",), ("o",), ("qkv", "o"), ("gate_up",), ("down",), ("qkv", "o", "gate_up", "down"), ()):
    share(kinds)
    t = timed(eng)
    print(f"one tensor for all layers: {'+'.join(kinds) or '(none)':22s} {t:.4f} ms per step  ({(t - base) * 1e3:+6.1f} us)")
