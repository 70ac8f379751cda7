"""Feasibility probe (timing only, results wrong by design): does a GEMM launched on a forked stream run CONCURRENTLY
with the row kernel that produces its activations, and what would that be worth per step?

The probe step issues every GEMM that follows a row kernel (gate_up after add+norm+quant, down after the quantiser, the
next layer's qkv after add+norm+quant) on a side stream forked BEFORE the row kernel, joins after both - the GEMM reads the
previous contents of the activation buffer.  Compared in one process against the ordinary serial step, both as hipGraphs.
The real thing (GEMM prefetches weights, waits for the row kernel's flag before its first activation load) lands between the two.

    python scripts/probe_overlap.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from qserve_amd import decode as D  # noqa: E402
from qserve_amd import fused as fusedmod  # noqa: E402
from qserve_amd.backend import fused_kernels, layernorm_ops  # noqa: E402


def probe_step(eng, side, mode):
    """mode 0: serial (the engine's own op sequence, per-channel, fused pairs); 1: GEMMs forked beside the row kernels."""
    cfg, B = eng.cfg, eng.B
    h, qa, qo, sums = eng.hidden, eng.q_act, eng.q_attn, eng.q_sum
    main = torch.cuda.current_stream()

    def beside(row_fn, gemm_fn):
        if mode == 0:
            row_fn()
            gemm_fn()
            return
        side.wait_stream(main)
        with torch.cuda.stream(side):
            gemm_fn()
        row_fn()
        main.wait_stream(side)

    nl = len(eng.layers)
    layernorm_ops.rms_norm_general_fuse_sum(qa, h, eng.layers[0]["ln1"], eng.q_sum, eng.q_scale, cfg["eps"], True)
    eng.layers[0]["qkv"](qa, eng.q_scale, eng.q_sum, eng.qkv_buf)
    for li, L in enumerate(eng.layers):
        q, k, v = eng.qkv_buf.split([eng.H * 128, eng.Hkv * 128, eng.Hkv * 128], dim=-1)
        fusedmod.single_query_attention_quant(
            q.reshape(B, eng.H, 128), k.reshape(B, eng.Hkv, 128), v.reshape(B, eng.Hkv, 128), eng.tables[li],
            eng.lengths, qo, eng.q_scale, 8192, 64, eng.size_per_token, eng.max_len, 128, cfg["rope_theta"],
            True, eng.int4, True, quant_sum=sums)
        L["o"](qo, eng.q_scale, eng.q_sum, eng.proj_out)
        beside(lambda: fusedmod.add_residual_rms_norm_general(qa, h, eng.proj_out, L["ln2"], eng.q_scale, cfg["eps"], sums),
               lambda: L["gate_up"].silu_mul(qa, eng.q_scale, eng.q_sum, eng.mlp_act, eng.gate_up_buf))
        beside(lambda: fused_kernels.invoke_quant_fuse_sum(eng.q_mlp, eng.mlp_act, eng.q_sum, eng.q_scale),
               lambda: L["down"](eng.q_mlp, eng.q_scale, eng.q_sum, eng.proj_out))
        if li + 1 < nl:
            nxt = eng.layers[li + 1]
            beside(lambda: fusedmod.add_residual_rms_norm_general(qa, h, eng.proj_out, nxt["ln1"], eng.q_scale, cfg["eps"],
                                                                  sums),
                   lambda: nxt["qkv"](qa, eng.q_scale, eng.q_sum, eng.qkv_buf))


def main():
    torch.cuda.set_device(0)
    eng = D.DecodeEngine(D.LLAMA3_8B, 64, 1024, 512, group_size=-1, int4_kv=True, device="cuda:0", with_lm_head=False)
    eng.prefill_cache(1024 + 16)
    eng.lengths.fill_(1030)
    side = torch.cuda.Stream()
    graphs = {}
    for mode in (0, 1):
        probe_step(eng, side, mode)
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                probe_step(eng, side, mode)
        torch.cuda.synchronize()
        graphs[mode] = (g, st)
    res = {0: [], 1: []}
    for rep in range(4):
        for mode in (0, 1):
            g, st = graphs[mode]
            with torch.cuda.stream(st):
                for _ in range(3):
                    g.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(20):
                    g.replay()
                e1.record(st)
            e1.synchronize()
            res[mode].append(e0.elapsed_time(e1) / 20)
    print("32 layers without lm_head, ms per step: serial " + " ".join(f"{x:.3f}" for x in res[0])
          + " | GEMMs beside their row kernels " + " ".join(f"{x:.3f}" for x in res[1]))


if __name__ == "__main__":
    main()
