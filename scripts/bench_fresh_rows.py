#!/usr/bin/env python3
"""Round 5: why do the row kernels cost 4.5-4.9 us inside the decode step (rocprofv3 timeline) and 3.1-3.2 us in their own chains?
Pairs GEMM -> row kernel in one hipGraph, the row kernel reading (a) the tile the GEMM has JUST written (all CUs, all XCDs: what the
step does) or (b) a buffer nobody wrote recently; per pair minus the GEMM's own chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from qserve_amd import fused as fz
import qserve_backend.fused_kernels as fk
import qserve_backend.qgemm_w4a8_per_chn as gc
from bench_rows_ab import timeit, dev, B, g, NB, hid, dl, gam, q8, sc, sm, act, q14, ws, sa   # noqa: E402  (prints its own line first)

A4 = torch.randint(-127, 128, (B, 4096), dtype=torch.int8, device=dev, generator=g)
Wo = [torch.randint(-128, 128, (4096, 2048), dtype=torch.int8, device=dev, generator=g) for _ in range(NB)]
Wgu = [torch.randint(-128, 128, (28672, 2048), dtype=torch.int8, device=dev, generator=g) for _ in range(NB)]
ws28 = (torch.rand((28672,), device=dev, generator=g) * 0.01).half()
og = torch.empty((B, 4096), dtype=torch.float16, device=dev)
gu = torch.empty((B, 14336), dtype=torch.float16, device=dev)
o_gemm = lambda i: gc.gemm_forward_cuda(A4, Wo[i % NB], ws, sa, ws, sa, og)
gu_gemm = lambda i: fz.gemm_silu_and_mul_per_chn(A4, Wgu[i % NB], ws28, sa, ws28, sa, gu)
norm_fresh = lambda i: fz.add_residual_rms_norm_general(q8, hid[i % NB], og, gam, sc, 1e-5, sm)
norm_stale = lambda i: fz.add_residual_rms_norm_general(q8, hid[i % NB], dl[i % NB], gam, sc, 1e-5, sm)
quant_fresh = lambda i: fk.invoke_quant_fuse_sum(q14, gu, sm, sc)
quant_stale = lambda i: fk.invoke_quant_fuse_sum(q14, act[i % NB], sm, sc)


def pair(a, b):
    def f(i):
        a(i)
        b(i)
    return f


for rnd in range(2):
    t_o, t_gu = timeit(o_gemm, 32), timeit(gu_gemm, 32)
    print(f"round {rnd}: o GEMM {t_o:5.2f}  gate_up+silu*mul {t_gu:5.2f} us")
    for name, a, ta, fresh, stale in (("o -> add+norm+quant", o_gemm, t_o, norm_fresh, norm_stale),
                                      ("gate_up -> quant[14336]", gu_gemm, t_gu, quant_fresh, quant_stale)):
        tf, ts = timeit(pair(a, fresh), 32), timeit(pair(a, stale), 32)
        print(f"  {name:26s}: row kernel behind the GEMM, reading its fresh output {tf - ta:5.2f} us | reading a stale buffer {ts - ta:5.2f} us")

# ---- does the number of DISTINCT kernels in the chain matter (instruction cache: the step alternates eight kernels, ~75 KB of code,
# over a 64 KB instruction cache per CU pair)?  chains of growing variety, each minus the sum of its members' own chains
NBW = 32      # weight sets per GEMM: every member streams from HBM in its own chain too (8 sets of the small GEMMs stay in the Infinity Cache)
Wo = [torch.randint(-128, 128, (4096, 2048), dtype=torch.int8, device=dev, generator=g) for _ in range(NBW)]
Wgu = [torch.randint(-128, 128, (28672, 2048), dtype=torch.int8, device=dev, generator=g) for _ in range(NBW)]
o_gemm = lambda i: gc.gemm_forward_cuda(A4, Wo[i % NBW], ws, sa, ws, sa, og)
gu_gemm = lambda i: fz.gemm_silu_and_mul_per_chn(A4, Wgu[i % NBW], ws28, sa, ws28, sa, gu)
A14 = torch.randint(-127, 128, (B, 14336), dtype=torch.int8, device=dev, generator=g)
Wqkv = [torch.randint(-128, 128, (6144, 2048), dtype=torch.int8, device=dev, generator=g) for _ in range(NBW)]
Wd = [torch.randint(-128, 128, (4096, 7168), dtype=torch.int8, device=dev, generator=g) for _ in range(NBW)]
ws6 = (torch.rand((6144,), device=dev, generator=g) * 0.01).half()
oq = torch.empty((B, 6144), dtype=torch.float16, device=dev)
od = torch.empty((B, 4096), dtype=torch.float16, device=dev)
qkv_gemm = lambda i: gc.gemm_forward_cuda(A4, Wqkv[i % NBW], ws6, sa, ws6, sa, oq)
down_gemm = lambda i: gc.gemm_forward_cuda(A14, Wd[i % NBW], ws, sa, ws, sa, od)
norm_down = lambda i: fz.add_residual_rms_norm_general(q8, hid[i % NB], od, gam, sc, 1e-5, sm)
members = dict(qkv=qkv_gemm, o=o_gemm, norm=norm_fresh, gate_up=gu_gemm, quant=quant_fresh, down=down_gemm, norm2=norm_down)
alone = {k: timeit(f, 64) for k, f in members.items()}
print("alone: " + "  ".join(f"{k} {v:.2f}" for k, v in alone.items()))


def chain(names):
    def f(i):
        for n in names:
            members[n](i)
    return f


for names in (["o", "norm"], ["gate_up", "quant"], ["o", "norm", "gate_up", "quant"], ["o", "norm", "gate_up", "quant", "down", "norm2"],
              ["qkv", "o", "norm", "gate_up", "quant", "down", "norm2"]):
    t = timeit(chain(names), 32)
    s = sum(alone[n] for n in names)
    print(f"chain {' -> '.join(names):60s}: {t:6.2f} us, members alone {s:6.2f}, excess {t - s:5.2f} us ({(t - s) / len(names):.2f} per launch)")
