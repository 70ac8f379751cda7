#!/usr/bin/env python3
"""Workload for the PMC passes (scripts/gpu_pmc.sh): the step's dominant kernels at bench.py's shapes, eager launches,
weights / KV pools rotating over 32 layers exactly as in the decode step (nothing is served from the Infinity Cache)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qserve_amd import decode as D
import qserve_backend.fused_attention as fa

if os.environ.get("QS_GEMM_VARIANT"):       # A/B passes (e.g. 9096 = 5000 + 4096: K slices of a channel block on one XCD, rounds 3-5)
    from qserve_amd._lib import lib
    lib.qs_set_gemm_variant(int(os.environ["QS_GEMM_VARIANT"]))
eng = D.DecodeEngine(D.LLAMA3_8B, 64, 1024, 512, with_lm_head=False)
eng.prefill_cache(1024)
eng.lengths.fill_(1033)      # bench.py default: context_start = prompt 1024 + 1 + 8 warm-up steps
B, nl = eng.B, len(eng.layers)
q, k, v = eng.qkv_buf.split([eng.H * 128, eng.Hkv * 128, eng.Hkv * 128], dim=-1)
q, k, v = q.reshape(B, eng.H, 128), k.reshape(B, eng.Hkv, 128), v.reshape(B, eng.Hkv, 128)
eng.qkv_buf.normal_()
eng.q_act.random_(-127, 128)
eng.q_mlp.random_(-127, 128)
eng.q_attn.random_(-127, 128)
eng.q_scale.fill_(0.01)
eng.q_sum.fill_(1.0)
for rep in range(2):
    for i in range(nl):
        L = eng.layers[i]
        L["qkv"](eng.q_act, eng.q_scale, eng.q_sum, eng.qkv_buf)
        L["gate_up"].silu_mul(eng.q_act, eng.q_scale, eng.q_sum, eng.mlp_act, eng.gate_up_buf)   # what the step launches
        L["down"](eng.q_mlp, eng.q_scale, eng.q_sum, eng.proj_out)
        L["o"](eng.q_attn, eng.q_scale, eng.q_sum, eng.proj_out)
        if "down" in eng.planes and i + 1 < nl:   # the form the fused step launches: K-slice planes + the row kernel that finishes them
            L["down"].planes(eng.q_mlp, eng.planes["down"])
            L["down"].add_norm_quant_planes(eng.q_act, eng.hidden, eng.planes["down"], eng.q_scale, eng.q_sum,
                                            eng.layers[i + 1]["ln1"], eng.q_scale, eng.cfg["eps"], eng.q_sum)
            eng.q_scale.fill_(0.01)
            eng.q_sum.fill_(1.0)
        fa.single_query_attention(q, k, v, eng.tables[i], eng.lengths, None, 8192, 64, eng.size_per_token,
                                  eng.max_len, 128, eng.cfg["rope_theta"], True, eng.int4, True)
torch.cuda.synchronize()
print("pmc workload done")
