#!/usr/bin/env python3
"""Which W4A8 GEMM problems of the reference's model zoo still dispatch to the two ROUND-1 decode kernels (family 1 = register-staged
split-K, gemm_w4a8.hip; family 2 = LDS-pair, gemm_w4a8_lds.hip) instead of the ring (3) / tiled (4) / wide (5) kernels?  Plan only
(qs_w4a8_gemm_plan runs the dispatcher without touching a device): works in the CPU container.  VERDICT r05 item 8.

    python scripts/list_dispatch_families.py            -> table on stdout (committed as profiles/round6_dispatch_families.txt)
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qserve_amd._lib import lib  # noqa: E402

MODELS = {   # hidden, q heads, kv heads, intermediate (the dense models of the reference's README list)
    "llama-3-8b": (4096, 32, 8, 14336), "llama-2-7b": (4096, 32, 32, 11008), "llama-2-13b": (5120, 40, 40, 13824),
    "llama-2-70b": (8192, 64, 8, 28672), "mistral-7b": (4096, 32, 8, 14336), "yi-34b": (7168, 56, 8, 20480),
    "qwen1.5-72b": (8192, 64, 64, 24576),
}
MS = [1, 4, 16, 32, 48, 64, 96, 128, 192, 256, 512, 1024, 2048, 4096, 65536]
FAM = {1: "split-K (round 1)", 2: "LDS-pair (round 1)", 3: "ring", 4: "tiled", 5: "wide"}


def shapes(tp):
    for name, (h, nh, nkv, inter) in MODELS.items():
        qkv = (nh + 2 * nkv) * 128
        if nkv % tp and tp > nkv:      # MHA-less sharding (kv heads replicated) is outside the loader's rules
            continue
        yield name, "qkv", qkv // tp, h
        yield name, "o", h, nh * 128 // tp
        yield name, "gate_up", 2 * inter // tp, h
        yield name, "down", h, inter // tp


def main():
    plan = (ctypes.c_int * 5)()
    legacy = {}
    total = 0
    for tp in (1, 2, 4, 8):
        for name, lin, N, K in shapes(tp):
            if N % 64 or K % 128:
                continue
            for pg in (0, 1):
                for M in MS:
                    if lib.qs_w4a8_gemm_plan(pg, M, N, K, plan) != 0:
                        continue
                    total += 1
                    if plan[0] in (1, 2):
                        legacy.setdefault((plan[0], name, lin, tp, N, K, pg), []).append(M)
    print(f"{total} (model, linear, tp, granularity, M) problems planned; {sum(len(v) for v in legacy.values())} of them on a round-1 kernel:")
    for (fam, name, lin, tp, N, K, pg), ms in sorted(legacy.items()):
        print(f"  {FAM[fam]:18s} {name:12s} {lin:8s} tp={tp} N={N:6d} K={K:6d} {'g128' if pg else 'per-channel'}: M = {ms}")


if __name__ == "__main__":
    main()
