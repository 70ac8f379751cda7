#!/usr/bin/env python3
"""Prefill KV writer (apply_bias_rope_update_kv_cache): time per call and GB/s for B x L prompt tokens of a Llama-3-8B
shaped qkv row (32 + 8 + 8 heads); variant 0 = vectorised form (RoPE table), 2 = per-lane form.  env: B, L, KV8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qserve_backend.fused_attention as fa
from qserve_amd._lib import lib
dev = torch.device("cuda:0")
B, L = int(os.environ.get("B", "64")), int(os.environ.get("L", "1024"))
int4 = not os.environ.get("KV8")
H, Hkv = 32, 8
T = B * L
mb = (L + 63) // 64
pb = Hkv * 64 * (64 if int4 else 128) + 64 * Hkv * 4
kp = torch.zeros((B * mb, pb), dtype=torch.uint8, device=dev)
vp = torch.zeros((B * mb, pb), dtype=torch.uint8, device=dev)
t = torch.empty((B, 2, mb), dtype=torch.int64)
idx = torch.arange(B * mb).reshape(B, mb)
t[:, 0] = kp.data_ptr() + idx * pb
t[:, 1] = vp.data_ptr() + idx * pb
t = t.to(dev)
qkv0 = torch.randn((T, (H + 2 * Hkv) * 128), dtype=torch.float16, device=dev)
seq = torch.full((B,), L, dtype=torch.int32, device=dev)
cu = torch.arange(0, B + 1, device=dev, dtype=torch.int32) * L
pad = fa.compute_padding_offsets(cu, L, T)
bytes_moved = T * ((H + 2 * Hkv) * 256 + (H + Hkv) * 256) + 2 * B * mb * pb     # row read, q/k written back, pages
for var in (0, 2, 0, 2):
    lib.qs_set_attention_variant(var)
    qkv = qkv0.clone()
    for _ in range(2):
        fa.apply_bias_rope_update_kv_cache(qkv, seq, pad, t, H, Hkv, L, 64, Hkv * (64 if int4 else 128), 128, 5e5, 8192, True, int4, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fa.apply_bias_rope_update_kv_cache(qkv, seq, pad, t, H, Hkv, L, 64, Hkv * (64 if int4 else 128), 128, 5e5, 8192, True, int4, True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    print(f"variant {var}: {T} tokens  {us:8.1f} us  {bytes_moved / us / 1e3:7.1f} GB/s")
lib.qs_set_attention_variant(0)
