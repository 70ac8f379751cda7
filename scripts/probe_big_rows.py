#!/usr/bin/env python3
"""Which op breaks at the reference's prompt-phase sizes (T = 65 536 token rows, gate_up output [T, 28 672] fp16 = 3.76 GB)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qserve_backend.qgemm_w4a8_per_chn as gemm
import qserve_backend.activation_ops as act
import qserve_backend.fused_kernels as fk
import qserve_backend.layernorm_ops as ln
dev = "cuda:0"
T, hid, inter = int(os.environ.get("T", "65536")), 4096, 14336
def step(name, fn):
    fn(); torch.cuda.synchronize(); print("ok", name, flush=True)
x8 = torch.randint(-127, 128, (T, hid), dtype=torch.int8, device=dev)
sc = torch.full((T,), 0.01, dtype=torch.float16, device=dev); sm = torch.zeros((T,), dtype=torch.float16, device=dev)
W = torch.randint(-128, 128, (2 * inter, hid // 2), dtype=torch.int8, device=dev)
ws = torch.full((2 * inter,), 0.005, dtype=torch.float16, device=dev); wz = torch.zeros((2 * inter,), dtype=torch.float16, device=dev)
gu = torch.empty((T, 2 * inter), dtype=torch.float16, device=dev)
step("gate_up gemm", lambda: gemm.gemm_forward_cuda(x8, W, ws, sc, wz, sm, gu))
ref_last = (x8[-1:].float() @ torch.zeros(1, 1, device=dev)) if False else None
print("last row finite:", bool(torch.isfinite(gu[-1]).all()), "first:", bool(torch.isfinite(gu[0]).all()))
a = torch.empty((T, inter), dtype=torch.float16, device=dev)
step("silu_and_mul", lambda: act.silu_and_mul(a, gu))
chk = (torch.nn.functional.silu(gu[-1, :inter].float()) * gu[-1, inter:].float())
print("silu last row max err", float((a[-1].float() - chk).abs().max()))
q = torch.empty((T, inter), dtype=torch.int8, device=dev)
step("invoke_quant_fuse_sum", lambda: fk.invoke_quant_fuse_sum(q, a, sm, sc))
step("invoke_quant", lambda: fk.invoke_quant(q, a, sc))
h = torch.randn((T, hid), dtype=torch.float16, device=dev); g = torch.ones((hid,), dtype=torch.float16, device=dev)
q2 = torch.empty((T, hid), dtype=torch.int8, device=dev)
step("rms_norm_general_fuse_sum", lambda: ln.rms_norm_general_fuse_sum(q2, h, g, sm, sc, 1e-5, True))
Wd = torch.randint(-128, 128, (hid, inter // 2), dtype=torch.int8, device=dev)
o = torch.empty((T, hid), dtype=torch.float16, device=dev)
step("down gemm", lambda: gemm.gemm_forward_cuda(q, Wd, ws[:hid], sc, wz[:hid], sm, o))
print("done")
