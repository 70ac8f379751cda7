#!/usr/bin/env python3
"""Cycles per phase of the prefill attention's key loop (library built with QS_EXTRA_HIPCC_FLAGS=-DQS_FLASH_TRACE).  env: B, L."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flash_attn.flash_attn_interface import flash_attn_varlen_func
from qserve_amd._lib import LIB_PATH
lib = ctypes.CDLL(LIB_PATH)
dev = torch.device("cuda:0")
B, L, H, Hkv = int(os.environ.get("B", "4")), int(os.environ.get("L", "8192")), 32, 8
T = B * L
qkv = torch.randn((T, (H + 2 * Hkv) * 128), dtype=torch.float16, device=dev)
q, k, v = qkv.split([H * 128, Hkv * 128, Hkv * 128], dim=-1)
q, k, v = q.reshape(T, H, 128), k.reshape(T, Hkv, 128), v.reshape(T, Hkv, 128)
cu = torch.arange(0, B + 1, dtype=torch.int32, device=dev) * L
NW = int(os.environ.get("NW", "4"))
nq = (L + 32 * NW - 1) // (32 * NW)
buf = torch.zeros((B * H * nq * NW * 8,), dtype=torch.int64, device=dev)
lib.qs_debug_flash_trace.argtypes = [ctypes.c_void_p]
assert lib.qs_debug_flash_trace(buf.data_ptr()) == 0
for _ in range(2):
    flash_attn_varlen_func(q, k, v, cu, cu, L, L, causal=True)
torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
a = a[a[:, 5] > 0]
per = a[:, :5] / a[:, 5:6]
names = ["loop top", "DMA issue + Q.K^T", "mask + softmax + rescale", "P.V", "wait tile + barrier"]
print(f"B={B} L={L}: cycles per key tile and wave (mean over {len(a)} waves; s_memtime ticks)")
for i, n in enumerate(names):
    print(f"  {n:28s} {per[:, i].mean():8.0f}")
print(f"  {'total':28s} {per.sum(1).mean():8.0f}")
