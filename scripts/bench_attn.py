#!/usr/bin/env python3
"""A/B micro-benchmark of decode attention (graph-timed).  env: B, LS (comma list), VARS (attention variants),
SAMEPAGE; --kv8 for the INT8 cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import qserve_backend.fused_attention as fa
from qserve_amd._lib import lib


def timeit(fn, reps=32, replays=6):
    """GPU-side time per launch: the launches are captured in a hipGraph (no host overhead between them)."""
    for i in range(2):
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        gph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)


dev = torch.device("cuda:0")
B, H, Hkv = int(os.environ.get("B", "64")), int(os.environ.get("H", "32")), int(os.environ.get("HKV", "8"))
VARS = [int(x) for x in os.environ.get("VARS", "0,1").split(",")]   # 0 auto, 1 VALU, 100+n = n KV splits
int4 = "--kv8" not in sys.argv
NL = int(os.environ.get("NL", "8"))   # KV pools rotated per launch (1 = Infinity-Cache-resident)
LS = [int(x) for x in os.environ.get("LS", "1024,1280,1535,4096").split(",")]
for L in LS:
    mb = (L + 63) // 64 + 1
    dhb = 64 if int4 else 128
    pb = Hkv * 64 * dhb + 64 * Hkv * 4
    nblocks = B * mb
    pools, tables = [], []
    for _ in range(NL):
        kp = torch.randint(0, 255, (nblocks, pb), dtype=torch.uint8, device=dev)
        vp = torch.randint(0, 255, (nblocks, pb), dtype=torch.uint8, device=dev)
        # sane fp16 scale/zero tails
        kp[:, Hkv * 64 * dhb:] = torch.tensor([0x00, 0x34], dtype=torch.uint8, device=dev).repeat((pb - Hkv * 64 * dhb) // 2)
        vp[:, Hkv * 64 * dhb:] = torch.tensor([0x00, 0x34], dtype=torch.uint8, device=dev).repeat((pb - Hkv * 64 * dhb) // 2)
        perm = torch.randperm(nblocks).reshape(B, mb)
        if os.environ.get('SAMEPAGE'): perm = perm * 0 + (perm % int(os.environ['SAMEPAGE']))
        t = torch.empty((B, 2, mb), dtype=torch.int64)
        t[:, 0] = kp.data_ptr() + perm * pb
        t[:, 1] = vp.data_ptr() + perm * pb
        pools.append((kp, vp)); tables.append(t.to(dev))
    qkv = torch.randn((B, (H + 2 * Hkv) * 128), dtype=torch.float16, device=dev)
    q, k, v = qkv.split([H * 128, Hkv * 128, Hkv * 128], dim=-1)
    q, k, v = q.reshape(B, H, 128), k.reshape(B, Hkv, 128), v.reshape(B, Hkv, 128)
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    bytes_ = B * (L - 1) * Hkv * (2 * dhb + 8)
    # every variant is measured ROUNDS times, round-robin (clock / thermal drift and the cold first measurement of a process
    # otherwise decide an A/B of a few per cent); the median is reported
    ROUNDS = int(os.environ.get("ROUNDS", "5"))
    res = {i: [] for i in range(len(VARS))}
    call = lambda i: fa.single_query_attention(q, k, v, tables[i % NL], lens, None, 8192, 64, Hkv * dhb, L, 128, 5e5, True, int4, True)
    if os.environ.get("FUSED"):       # FUSED=1: the attention + invoke_quant_fuse_sum launch the decode step uses
        from qserve_amd import fused as fz
        qq = torch.empty((B, H * 128), dtype=torch.int8, device=dev)
        qsc, qsm = torch.empty((B,), dtype=torch.float16, device=dev), torch.empty((B,), dtype=torch.float16, device=dev)
        call = lambda i: fz.single_query_attention_quant(q, k, v, tables[i % NL], lens, qq, qsc, 8192, 64, Hkv * dhb, L, 128, 5e5,
                                                         True, int4, True, quant_sum=qsm)
    timeit(call, reps=16)                                  # warm-up of clocks and caches, discarded
    for _ in range(ROUNDS):
        for i, var in enumerate(VARS):
            lib.qs_set_attention_variant(var)
            res[i].append(timeit(call, reps=16))
    lib.qs_set_attention_variant(0)
    row = []
    for i, var in enumerate(VARS):
        us = sorted(res[i])[len(res[i]) // 2]
        row.append(f"variant {var}: {us:7.2f} us {bytes_ / us / 1e3:7.0f} GB/s")
    print(f"KV{'4' if int4 else '8'} B={B} L={L}: " + "   ".join(row))
