#!/usr/bin/env python3
"""The (add + norm + quant -> GEMM) edges of the decode step as two launches and as ONE head launch, hipGraph chains rotating over
NL weight sets, alternating; us per edge.  env V=5512 (timing library): activation requests through the caches (probe)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from qserve_amd import fused as fz
from qserve_amd._lib import lib
import qserve_backend.qgemm_w4a8_per_chn as opc

dev = torch.device("cuda:0")
HID, M, NL = 4096, 64, 12
g = torch.Generator(device=dev).manual_seed(1)


def weights(N, K):
    return dict(q=torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=dev, generator=g),
                ws=(torch.rand((N,), device=dev, generator=g) * 0.004 + 0.001).half(),
                wz=(torch.rand((N,), device=dev, generator=g) * 0.01).half())


hidden = (torch.randn((M, HID), device=dev, generator=g) * 0.7).half()
gamma = (torch.rand((HID,), device=dev, generator=g) + 0.5).half()
delta = (torch.randn((M, HID), device=dev, generator=g) * 0.5).half()
qa = torch.empty((M, HID), dtype=torch.int8, device=dev)
sc = torch.full((M,), 0.01, dtype=torch.float16, device=dev)
sm = torch.zeros((M,), dtype=torch.float16, device=dev)
down = weights(HID, 14336)
Aq = torch.randint(-127, 128, (M, 14336), dtype=torch.int8, device=dev, generator=g)
ks = fz.gemm_planes_plan(M, HID, 14336, False)
planes = torch.empty((ks, M, HID), dtype=torch.int32, device=dev)
fz.gemm_planes(Aq, down["q"], planes)
if os.environ.get("V"):
    lib.qs_set_gemm_variant(int(os.environ["V"]))
for name, N, silu, use_planes in (("anq(planes) -> qkv", 6144, False, True), ("anq(delta) -> gate_up+silu*mul", 28672, True, False)):
    Ws = [weights(N, HID) for _ in range(NL)]
    out = torch.empty((M, N // 2 if silu else N), dtype=torch.float16, device=dev)
    tmp = torch.empty((M, N), dtype=torch.float16, device=dev)

    def pair(i):
        w = Ws[i % NL]
        if use_planes:
            fz.add_residual_rms_norm_general_planes(qa, hidden, planes, down["ws"], sc, gamma, sc, 1e-5, w_szs=down["wz"], a_ssums=sm, input_sum=sm)
        else:
            fz.add_residual_rms_norm_general(qa, hidden, delta, gamma, sc, 1e-5, input_sum=sm)
        if silu:
            fz.gemm_silu_and_mul_per_chn(qa, w["q"], w["ws"], sc, w["wz"], sm, out, tmp)
        else:
            opc.gemm_forward_cuda(qa, w["q"], w["ws"], sc, w["wz"], sm, out)

    def one(i):
        w = Ws[i % NL]
        kw = dict(input_sum=sm, w_szs=w["wz"], silu_mul=silu, tmp=tmp)
        if use_planes:
            kw.update(planes=planes, p_wscales=down["ws"], p_w_szs=down["wz"], p_ascales=sc, p_a_ssums=sm)
        else:
            kw.update(delta=delta)
        fz.add_norm_quant_gemm(qa, hidden, gamma, sc, 1e-5, w["q"], w["ws"], out, **kw)

    def gemm_only(i):
        w = Ws[i % NL]
        if silu:
            fz.gemm_silu_and_mul_per_chn(qa, w["q"], w["ws"], sc, w["wz"], sm, out, tmp)
        else:
            opc.gemm_forward_cuda(qa, w["q"], w["ws"], sc, w["wz"], sm, out)

    t = {"pair": [], "one": [], "gemm": []}
    for _ in range(4):
        t["pair"].append(bench.time_kernel(pair, 2 * NL, torch))
        t["one"].append(bench.time_kernel(one, 2 * NL, torch))
        t["gemm"].append(bench.time_kernel(gemm_only, 2 * NL, torch))
    print(f"{name:34s} two launches {statistics.median(t['pair']):6.2f} us | one head launch {statistics.median(t['one']):6.2f} us | the GEMM alone "
          f"{statistics.median(t['gemm']):6.2f} us   {[round(x, 2) for x in t['pair']]} {[round(x, 2) for x in t['one']]}", flush=True)
