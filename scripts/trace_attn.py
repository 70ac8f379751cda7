#!/usr/bin/env python3
"""Timeline of the KV4 decode attention kernel (EXP & 32 build: s_memtime stamps per wave).  env: B, L, VAR (232 / 241)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import qserve_backend.fused_attention as fa
from qserve_amd._lib import lib
dev = torch.device("cuda:0")
B, H, Hkv, L = int(os.environ.get("B", "64")), 32, 8, int(os.environ.get("L", "1033"))
VAR = int(os.environ.get("VAR", "232"))
mb = (L + 63) // 64 + 1
pb = Hkv * 64 * 64 + 64 * Hkv * 4
NL = 8
pools, tables = [], []
for _ in range(NL):
    kp = torch.randint(0, 255, (B * mb, pb), dtype=torch.uint8, device=dev)
    vp = torch.randint(0, 255, (B * mb, pb), dtype=torch.uint8, device=dev)
    kp[:, Hkv * 64 * 64:] = torch.tensor([0x00, 0x34], dtype=torch.uint8, device=dev).repeat((pb - Hkv * 64 * 64) // 2)
    vp[:, Hkv * 64 * 64:] = torch.tensor([0x00, 0x34], dtype=torch.uint8, device=dev).repeat((pb - Hkv * 64 * 64) // 2)
    perm = torch.randperm(B * mb).reshape(B, mb)
    t = torch.empty((B, 2, mb), dtype=torch.int64)
    t[:, 0] = kp.data_ptr() + perm * pb
    t[:, 1] = vp.data_ptr() + perm * pb
    pools.append((kp, vp)); tables.append(t.to(dev))
qkv = torch.randn((B, (H + 2 * Hkv) * 128), dtype=torch.float16, device=dev)
q, k, v = qkv.split([H * 128, Hkv * 128, Hkv * 128], dim=-1)
q, k, v = q.reshape(B, H, 128), k.reshape(B, Hkv, 128), v.reshape(B, Hkv, 128)
lens = torch.full((B,), L, dtype=torch.int32, device=dev)

lib.qs_set_attention_variant(VAR)
for i in range(6):   # the last launch's stamps survive; earlier launches make the caches / pools "cold" like the step
    fa.single_query_attention(q, k, v, tables[i % NL], lens, None, 8192, 64, Hkv * 64, L, 128, 5e5, True, True, True)
torch.cuda.synchronize()
n = B * Hkv * 8 * 16
out = torch.zeros((n,), dtype=torch.int64, device=dev)
assert lib.qs_debug_copy_split_workspace(out.data_ptr(), n * 8) == 0
st = out.cpu().numpy().reshape(B * Hkv, 8, 16).astype(np.float64)
st = np.where(st > 0, st, np.nan)
wg0 = np.nanmin(st[:, :, 0], axis=1)                  # every workgroup's own first entry
skew = wg0 - np.nanmin(wg0)
rel = st - wg0[:, None, None]
names = ["entry", "tl", "dma issued", "A0 issued", "flag seen", "A0 in", "u0 done", "A1 in", "u1 done", "qsum pass 1", "qsum pass 2", "qsum done", "loop done", "sync1", "merged", "end"]
print(f"L={L} VAR={VAR}: s_memtime ticks (shader cycles) since the workgroup's first wave entered; mean / min / max over workgroups x waves")
for i, nm in enumerate(names):
    if not nm: continue
    x = rel[:, :, i]
    if np.all(np.isnan(x)): continue
    print(f"  {i:2d} {nm:12s} mean {np.nanmean(x):7.0f}  min {np.nanmin(x):7.0f}  max {np.nanmax(x):7.0f}   wave0 {np.nanmean(rel[:, 0, i]):7.0f}  wave3 {np.nanmean(rel[:, 3, i]):7.0f}  wave7 {np.nanmean(rel[:, 7, i]):7.0f}")
print("  per wave (mean): " + "  ".join(f"w{w}: A0in {np.nanmean(rel[:, w, 5]):6.0f} done {np.nanmean(rel[:, w, 12]):6.0f}" for w in range(8)))
print("  slowest wave of a workgroup, loop done: mean %.0f;  earliest A0 in of a workgroup: mean %.0f" % (np.nanmean(np.nanmax(rel[:, :, 12], axis=1)), np.nanmean(np.nanmin(rel[:, :, 5], axis=1))))
end = np.nanmax(rel[:, :, 15], axis=1)
print(f"  workgroup lifetime: mean {end.mean():.0f}  min {end.min():.0f}  max {end.max():.0f};  start skew: mean {skew.mean():.0f}  max {skew.max():.0f};  "
      f"last end since first entry: {np.nanmax(st[:, :, 15]) - np.nanmin(wg0):.0f}")
lib.qs_set_attention_variant(0)
