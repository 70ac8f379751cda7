#!/usr/bin/env python3
"""Timeline of the decode ring GEMM (library built with QS_EXTRA_HIPCC_FLAGS=-DQS_RING_TRACE).  env: M, N, K, MODE."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import qserve_backend.qgemm_w4a8_per_chn as gc
from qserve_amd._lib import lib
dev = torch.device("cuda:0")
M, N, K = int(os.environ.get("M", "64")), int(os.environ.get("N", "28672")), int(os.environ.get("K", "4096"))
g = torch.Generator(device=dev).manual_seed(0)
NL = 8
A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
W = [torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=dev, generator=g) for _ in range(NL)]
ws = (torch.rand((N,), device=dev, generator=g) * 0.01).half()
sa = (torch.rand((M,), device=dev, generator=g) * 0.01).half()
out = torch.empty((M, N), dtype=torch.float16, device=dev)
nwg = 4096
buf = torch.zeros((nwg * 8 * 16,), dtype=torch.int64, device=dev)
lib.qs_debug_ring_trace.argtypes = [ctypes.c_void_p]
assert lib.qs_debug_ring_trace(buf.data_ptr()) == 0
for i in range(6):
    buf.zero_()
    gc.gemm_forward_cuda(A, W[i % NL], ws, sa, ws, sa, out)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(nwg, 8, 16).astype(np.float64)
used = st[:, 0, 0] > 0
st = st[used]
print(f"M={M} N={N} K={K}: {used.sum()} workgroups traced")
t0 = st[:, :, 0].min(axis=1, keepdims=True)
names = {0: "entry", 15: "first request", 1: "prologue issued", 2: "stage 0 landed", 3: "loop done", 4: "rings dead (sync)", 7: "partials written", 10: "sync", 11: "partials summed", 14: "seam done", 12: "epilogue math", 5: "tile staged (sync)", 6: "end"}
for i, nm in names.items():
    x = st[:, :, i] - t0
    x = np.where(st[:, :, i] > 0, x, np.nan)
    print(f"  {nm:16s} mean {np.nanmean(x):8.0f}  min {np.nanmin(x):8.0f}  max {np.nanmax(x):8.0f}")
if st[:, :, 13].max() > 0:
    fin = st[:, :, 13] > 0
    print(f"  K-slice seam: {int(fin.sum())} finishing waves, polls per batch-wave mean {st[:, :, 13][fin].mean():.2f} max {st[:, :, 13].max():.0f}; "
          f"loop done of the finishing waves {np.nanmean(np.where(fin, st[:, :, 3] - t0, np.nan)):.0f} vs the writers {np.nanmean(np.where((~fin) & (st[:, :, 3] > 0), st[:, :, 3] - t0, np.nan)):.0f}")
print(f"  in-loop: waiting (vmcnt + barrier) mean {st[:, :, 8].mean():8.0f} ticks, rounds (LDS reads + MFMA + DMA issue) mean {st[:, :, 9].mean():8.0f} ticks")
