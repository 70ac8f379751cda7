#!/usr/bin/env python3
"""Timeline of a row-op-head GEMM launch (library built with QS_EXTRA_HIPCC_FLAGS=-DQS_RING_TRACE --timing).  env: N (6144), SILU (0)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qserve_amd import fused as fz
from qserve_amd._lib import lib
dev = torch.device("cuda:0")
HID, M = 4096, 64
N, silu = int(os.environ.get("N", "6144")), os.environ.get("SILU", "0") == "1"
g = torch.Generator(device=dev).manual_seed(0)
NL = 8
W = [torch.randint(-128, 128, (N, HID // 2), dtype=torch.int8, device=dev, generator=g) for _ in range(NL)]
ws = (torch.rand((N,), device=dev, generator=g) * 0.01).half()
hidden = (torch.randn((M, HID), device=dev, generator=g) * 0.7).half()
gamma = (torch.rand((HID,), device=dev, generator=g) + 0.5).half()
delta = (torch.randn((M, HID), device=dev, generator=g) * 0.5).half()
qa = torch.empty((M, HID), dtype=torch.int8, device=dev)
sc = torch.zeros((M,), dtype=torch.float16, device=dev)
sm = torch.zeros((M,), dtype=torch.float16, device=dev)
out = torch.empty((M, N // 2 if silu else N), dtype=torch.float16, device=dev)
tmp = torch.empty((M, N), dtype=torch.float16, device=dev)
nwg = 4096
buf = torch.zeros((nwg * 8 * 16,), dtype=torch.int64, device=dev)
lib.qs_debug_ring_trace.argtypes = [ctypes.c_void_p]
lib.qs_debug_ring_trace.restype = ctypes.c_int
assert lib.qs_debug_ring_trace(buf.data_ptr()) == 0
n0 = lib.qs_debug_head_launch_count()
for i in range(6):
    buf.zero_()
    fz.add_norm_quant_gemm(qa, hidden, gamma, sc, 1e-5, W[i % NL], ws, out, delta=delta, input_sum=sm, w_szs=ws, silu_mul=silu, tmp=tmp)
torch.cuda.synchronize()
print("head launches taken:", lib.qs_debug_head_launch_count() - n0)
st = buf.cpu().numpy().reshape(nwg, 8, 16).astype(np.float64)
used = st[:, 0, 0] > 0
st = st[used]
nrow = (M + 1) // 2
rows, gem = st[:nrow], st[nrow:]
print(f"N={N} silu={silu}: {len(rows)} row workgroups, {len(gem)} GEMM workgroups; cycles since the WORKGROUP'S OWN entry (s_memtime bases differ by XCD)")
def line(name, blk, i):
    t0 = np.where(blk[:, :, 0] > 0, blk[:, :, 0], np.inf).min(axis=1, keepdims=True)
    x = np.where(blk[:, :, i] > 0, blk[:, :, i] - t0, np.nan)
    print(f"  {name:34s} mean {np.nanmean(x):8.0f}  min {np.nanmin(x):8.0f}  max {np.nanmax(x):8.0f}")
line("rows: row function done", rows, 1); line("rows: drained / census", rows, 2)
for i, nm in {15: "gemm: prologue start", 13: "gemm: flags seen", 14: "gemm: activations landed", 2: "gemm: stage 0 barrier", 3: "gemm: loop done", 4: "gemm: rings dead"}.items():
    line(nm, gem, i)
