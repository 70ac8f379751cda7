#!/usr/bin/env python3
"""Prefill attention provider (flash_prefill.hip): TFLOP/s on Llama-3-8B shapes (causal, GQA 32/8, head_dim 128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_attn.flash_attn_interface import flash_attn_varlen_func
from qserve_amd._lib import lib

dev = torch.device("cuda:0")
H, Hkv = 32, 8
for B, L in [(16, 1024), (64, 1024), (8, 4096), (4, 8192)]:
    T = B * L
    qkv = torch.randn((T, (H + 2 * Hkv) * 128), dtype=torch.float16, device=dev)
    q, k, v = qkv.split([H * 128, Hkv * 128, Hkv * 128], dim=-1)
    q, k, v = q.reshape(T, H, 128), k.reshape(T, Hkv, 128), v.reshape(T, Hkv, 128)
    cu = torch.arange(0, B + 1, dtype=torch.int32, device=dev) * L
    flops = 4.0 * B * H * L * L * 128 / 2          # causal: half of the score matrix
    line = f"B={B:3d} L={L:5d}:"
    # VARIANTS (round 6): qs_debug_flash_variant codes, alternating in one process (0 = block-pipelined loop, 1 = the loop of rounds 2-5)
    vs = [int(x) for x in os.environ.get("VARIANTS", "0").split(",")]
    res = {v_: [] for v_ in vs}
    for rnd in range(int(os.environ.get("ROUNDS", "3"))):
        for v_ in vs:
            if hasattr(lib, "qs_debug_flash_variant"):
                lib.qs_debug_flash_variant(v_)
            for _ in range(2):
                flash_attn_varlen_func(q, k, v, cu, cu, L, L, causal=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                flash_attn_varlen_func(q, k, v, cu, cu, L, L, causal=True)
            e1.record()
            torch.cuda.synchronize()
            res[v_].append(e0.elapsed_time(e1) / reps)
    if hasattr(lib, "qs_debug_flash_variant"):
        lib.qs_debug_flash_variant(0)
    for v_ in vs:
        ms = sorted(res[v_])[len(res[v_]) // 2]
        line += f"  variant {v_}: {ms:7.3f} ms {flops / ms / 1e9:6.1f} TFLOP/s ({flops / ms / 1e9 / 2500:.3f} of the dense fp16 MFMA peak)"
    print(line, flush=True)
