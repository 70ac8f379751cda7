"""Drop-in proof: the reference's OWN, UNCHANGED Python - `LlamaForCausalLM` with its
`W4A8OF16LinearDynamicInputScale`, `RMSNormGeneral`, `SiluAndMulQuant`, `LlamaAttention`, `InputMetadata` /
`ActivationBuffer` (qserve/modeling/models/llama_w4a8_unpad.py, layers/*.py, utils/input_metadata.py) - is imported
over THIS repository's `qserve_backend` / `flash_attn` / `xformers` packages and run for a prefill and two decode steps.

The reference tree exists only in the authoring container (CPU) and the HIP kernels only run on the GPU box, so the C
entry points are replaced by tests/_fake_abi.py: a host-memory simulator that receives the exact C-ABI arguments
(addresses, sizes, strides, flags) and evaluates them with the oracle.  What is under test is everything between the
reference's call sites and the C ABI: module / function names, positional order, dtype / stride checks, in-place vs
returned outputs, M/N/K and stride lowering, the page-address tables.  The expected values come from a direct
composition of the oracle functions written from the model's mathematics (no shim involved); both routes use the same
arithmetic, so they must agree BIT FOR BIT (hidden states, logits, every byte of the KV pools).

Skipped where /root/reference is absent (the GPU box): tests/test_callsites.py then still binds every recorded call site
against the mirror's signatures, and tests/test_decode_gpu.py runs the same op sequence on the device.
"""
import os
import sys

import numpy as np
import pytest
import torch

import _fake_abi
from oracle import flash as oflash
from oracle import fused as ofused
from oracle import kvattn, w4a8

REF = os.environ.get("QSERVE_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "qserve", "modeling")),
                                reason="reference tree not present (authoring container only)")


@pytest.fixture(scope="module")
def ref(built_lib):
    """Import the reference's model code over this repo's backend packages (the import trick of
    tests/golden/make_golden.py: w4a8_linear.py:19 evaluates torch.cuda.current_device() at import)."""
    saved = torch.cuda.current_device
    torch.cuda.current_device = lambda: "cpu"
    sys.path.insert(0, REF)
    try:
        import qserve_backend                                                        # this repository's mirror
        assert os.path.dirname(os.path.abspath(qserve_backend.__file__)).startswith(
            os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import qserve.modeling.models.llama_w4a8_unpad as model_mod
        import qserve.utils.input_metadata as im_mod
        from qserve.sampling_params import SamplingParams
    finally:
        torch.cuda.current_device = saved
        sys.path.remove(REF)
    assert model_mod.__file__.startswith(REF), "must be the reference's own file"
    return model_mod, im_mod, SamplingParams


CFG = dict(hidden=256, heads=2, kv_heads=1, inter=512, layers=2, vocab=96, eps=1e-5, theta=10000.0)


def make_weights(group_size, bias, seed):
    """Random quantised checkpoint tensors in the reference's format (packed by the reference-pinned oracle packer)."""
    r = np.random.default_rng(seed)
    hid, H, Hkv, inter = CFG["hidden"], CFG["heads"], CFG["kv_heads"], CFG["inter"]

    def linear(n, k, with_bias):
        d = {}
        if group_size == -1:
            q = r.integers(0, 16, (n, k), dtype=np.uint8)
            z = r.integers(0, 16, (n,))
            s1 = r.uniform(0.002, 0.01, n).astype(np.float16)
            d["qweight"], d["s1_scales"], d["s1_szeros"] = w4a8.pack_per_channel(q, z, s1)
        else:
            from oracle import synth
            pr = synth.per_group_problem(1, n, k, seed=int(r.integers(1 << 30)))
            d["qweight"], d["s1_scales"] = pr["qweight"], (pr["wscales"].astype(np.float32) * 0.5).astype(np.float16)
            d["s2_scales"], d["s2_zeros"] = pr["s2_scales"], pr["s2_zeros"]
        if with_bias:
            d["bias"] = r.uniform(-0.5, 0.5, n).astype(np.float16)
        return d
    layers = []
    for _ in range(CFG["layers"]):
        layers.append(dict(
            ln1=r.uniform(0.5, 1.5, hid).astype(np.float16), ln2=r.uniform(0.5, 1.5, hid).astype(np.float16),
            qkv=linear((H + 2 * Hkv) * 128, hid, bias), o=linear(hid, H * 128, bias),
            gate_up=linear(2 * inter, hid, False), down=linear(hid, inter, False)))
    return dict(layers=layers, norm=r.uniform(0.5, 1.5, hid).astype(np.float16),
                embed=(r.standard_normal((CFG["vocab"], hid)) * 0.5).astype(np.float16),
                lm_head=(r.standard_normal((CFG["vocab"], hid)) * 0.05).astype(np.float16))


def build_reference_model(ref, W, group_size, bias, int4):
    model_mod, _, SamplingParams = ref
    from transformers import LlamaConfig
    cfg = LlamaConfig(hidden_size=CFG["hidden"], intermediate_size=CFG["inter"], num_hidden_layers=CFG["layers"],
                      num_attention_heads=CFG["heads"], num_key_value_heads=CFG["kv_heads"], vocab_size=CFG["vocab"],
                      rms_norm_eps=CFG["eps"], rope_theta=CFG["theta"], max_position_embeddings=8192)
    cfg.rope_theta = CFG["theta"]          # (transformers 5 moved it into rope_parameters; the reference reads the attribute)
    cfg.attention_bias = bias
    model = model_mod.LlamaForCausalLM(cfg, group_size, SamplingParams(),
                                       kv_cache_config={"INT4_ENABLED": int4, "ZEROS_ENABLED": True}).half()
    sd = model.state_dict()
    with torch.no_grad():
        for li, L in enumerate(W["layers"]):
            pre = f"model.layers.{li}."
            sd[pre + "input_layernorm.weight"].copy_(torch.from_numpy(L["ln1"]))
            sd[pre + "post_attention_layernorm.weight"].copy_(torch.from_numpy(L["ln2"]))
            for mod, key in (("self_attn.qkv_proj", "qkv"), ("self_attn.o_proj", "o"), ("mlp.gate_up_proj", "gate_up"),
                             ("mlp.down_proj", "down")):
                for name, val in L[key].items():
                    sd[pre + mod + "." + name].copy_(torch.from_numpy(val))
        sd["model.norm.weight"].copy_(torch.from_numpy(W["norm"]))
        sd["model.embed_tokens.weight"].copy_(torch.from_numpy(W["embed"]))
        sd["lm_head.weight"].copy_(torch.from_numpy(W["lm_head"]))
    return model


class OracleModel:
    """The same network written directly on the oracle functions (index-based page pools)."""

    def __init__(self, W, group_size, int4, nblocks):
        self.W, self.g, self.int4 = W, group_size, int4
        self.pools = [kvattn.PagePool(nblocks, CFG["kv_heads"], 128, int4) for _ in range(CFG["layers"])]

    def linear(self, L, x, scale, ssum):
        if self.g == -1:
            _, o = w4a8.gemm_per_chn(x, L["qweight"], L["s1_scales"], scale, L["s1_szeros"], ssum)
        else:
            _, o = w4a8.gemm_per_group(x, L["qweight"], L["s2_zeros"], L["s2_scales"], L["s1_scales"], scale)
        if "bias" in L:                                   # w4a8_linear.py:116-118: output_buffer += bias (fp16 add)
            o = (o.astype(np.float32) + L["bias"].astype(np.float32)[None, :]).astype(np.float16)
        return o

    def norm_quant(self, x, w):
        if self.g == -1:
            q, sc, sm, _ = ofused.rms_norm_general(x, w, CFG["eps"], with_sum=True)
            return q, sc, sm
        q, sc, _ = ofused.rms_norm_general(x, w, CFG["eps"])
        return q, sc, None

    def quant(self, x):
        if self.g == -1:
            q, sc, sm, _ = ofused.quant_per_token(x, with_sum=True)
            return q, sc, sm
        q, sc, _ = ofused.quant_per_token(x)
        return q, sc, None

    @staticmethod
    def add(a, b):
        return (a.astype(np.float32) + b.astype(np.float32)).astype(np.float16)

    def forward(self, tokens, is_prompt, lens, tables):
        H, Hkv = CFG["heads"], CFG["kv_heads"]
        h = self.W["embed"][tokens]
        B = len(lens)
        if is_prompt:
            cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            max_len = int(max(lens))
            pad = kvattn.compute_padding_offsets(cu, max_len, len(tokens))
        for li, L in enumerate(self.W["layers"]):
            xq, sc, sm = self.norm_quant(h, L["ln1"])
            qkv = self.linear(L["qkv"], xq, sc, sm)
            if is_prompt:
                kvattn.prefill_update_kv_cache(qkv, np.asarray(lens, np.int32), pad, tables, self.pools[li], H, Hkv,
                                               max_len, np.float32(CFG["theta"]))
                q = qkv[:, :H * 128].reshape(-1, H, 128)
                k = qkv[:, H * 128:(H + Hkv) * 128].reshape(-1, Hkv, 128)
                v = qkv[:, (H + Hkv) * 128:].reshape(-1, Hkv, 128)
                attn = oflash.attention_varlen(q, k, v, cu, cu, None, True).astype(np.float16).reshape(len(tokens), -1)
            else:
                q = qkv[:, :H * 128].reshape(B, H, 128)
                k = qkv[:, H * 128:(H + Hkv) * 128].reshape(B, Hkv, 128)
                v = qkv[:, (H + Hkv) * 128:].reshape(B, Hkv, 128)
                attn = kvattn.decode_attention(q, k, v, tables, np.asarray(lens, np.int32), self.pools[li],
                                               np.float32(CFG["theta"]), "kernel").reshape(B, -1)
            aq, sc, sm = self.quant(attn)
            h = self.add(h, self.linear(L["o"], aq, sc, sm))
            xq, sc, sm = self.norm_quant(h, L["ln2"])
            gu = self.linear(L["gate_up"], xq, sc, sm)
            mq, sc, sm = self.quant(ofused.silu_and_mul(gu))
            h = self.add(h, self.linear(L["down"], mq, sc, sm))
        return ofused.rms_norm(h, self.W["norm"], CFG["eps"])


@pytest.mark.parametrize("group_size,bias,int4", [(-1, False, True), (128, False, True), (-1, True, False)],
                         ids=["per_chn-kv4", "g128-kv4", "per_chn-bias-kv8"])
def test_unchanged_reference_model_runs_on_the_mirror(ref, monkeypatch, group_size, bias, int4):
    model_mod, im_mod, _ = ref
    calls = _fake_abi.install(monkeypatch)
    import qserve_backend.fused_attention as fused_attention
    W = make_weights(group_size, bias, seed=3)
    model = build_reference_model(ref, W, group_size, bias, int4)
    lens = [5, 70, 64]                                        # ragged prompts; one ends exactly on a page boundary
    B, nl, Hkv = len(lens), CFG["layers"], CFG["kv_heads"]
    mb = 3
    nblocks = B * mb + 2
    r = np.random.default_rng(0)
    tab_idx = np.stack([r.permutation(nblocks)[:B * mb].reshape(B, mb), r.permutation(nblocks)[:B * mb].reshape(B, mb)], 1)
    pb = kvattn.page_bytes(Hkv, 128, int4)
    # the engine's pools (cache_engine.py:100-114) and raw-address tables (model_runner.py:396-414, 494-520)
    gpu_cache = [(torch.zeros((nblocks, pb), dtype=torch.uint8), torch.zeros((nblocks, pb), dtype=torch.uint8))
                 for _ in range(nl)]
    offs = torch.from_numpy(tab_idx.astype(np.int64)) * pb
    layer_tables = []
    for l in range(nl):
        t = offs.clone()
        t[:, 0] += gpu_cache[l][0].data_ptr()
        t[:, 1] += gpu_cache[l][1].data_ptr()
        layer_tables.append(t)
    oracle_model = OracleModel(W, group_size, int4, nblocks)

    # ---- prefill (model_runner.py:333-451) -------------------------------------------------------------------------
    tokens = torch.from_numpy(r.integers(0, CFG["vocab"], sum(lens)))
    ctx = torch.tensor(lens, dtype=torch.int32)
    cu = torch.nn.functional.pad(torch.cumsum(ctx, 0).int(), (1, 0), value=0)
    pad = fused_attention.compute_padding_offsets(cu, max(lens), sum(lens))
    meta = im_mod.InputMetadata(is_prompt=True, context_lens=ctx, padding_offsets=pad, cu_seqlens=cu, max_seq_len=max(lens),
                                max_block_table_len=mb, block_tables=layer_tables, kv_cache_dtype="int8", kv_scales=None,
                                batched_seq_len=sum(lens), model=model)
    logits = model(tokens, meta)
    exp_h = oracle_model.forward(tokens.numpy(), True, lens, tab_idx)
    last = (cu[1:] - 1).long()
    exp_logits = torch.nn.functional.linear(torch.from_numpy(exp_h)[last], torch.from_numpy(W["lm_head"]))
    assert torch.equal(logits, exp_logits), "prefill logits differ"
    for l in range(nl):
        assert np.array_equal(gpu_cache[l][0].numpy(), oracle_model.pools[l].k), f"layer {l}: K pages differ after prefill"
        assert np.array_equal(gpu_cache[l][1].numpy(), oracle_model.pools[l].v), f"layer {l}: V pages differ after prefill"
    names = [c[0] for c in calls]
    per_layer = ["qs_w4a8_per_chn_gemm" if group_size == -1 else "qs_w4a8_per_group_gemm"]
    assert names.count(per_layer[0]) == 4 * nl and names.count("qs_flash_attn_varlen_fwd") == nl
    assert names.count("qs_apply_bias_rope_update_kv_cache") == nl and names.count("qs_single_query_attention") == 0

    # ---- two decode steps (model_runner.py:453-548: one [Layer, Seq, 2, Len] table tensor) -------------------------
    all_tables = torch.stack(layer_tables)
    for step in range(2):
        del calls[:]
        lens = [n + 1 for n in lens]                           # context INCLUDING the new token (seq_data.get_len())
        tok = torch.from_numpy(r.integers(0, CFG["vocab"], B))
        ctx = torch.tensor(lens, dtype=torch.int32)
        meta = im_mod.InputMetadata(is_prompt=False, cu_seqlens=None, padding_offsets=None, context_lens=ctx,
                                    max_seq_len=max(lens), max_block_table_len=mb, block_tables=all_tables, kv_scales=None,
                                    kv_cache_dtype="int8", batched_seq_len=B, model=model)
        logits = model(tok, meta)
        exp_h = oracle_model.forward(tok.numpy(), False, lens, tab_idx)
        exp_logits = torch.nn.functional.linear(torch.from_numpy(exp_h), torch.from_numpy(W["lm_head"]))
        assert torch.equal(logits, exp_logits), f"decode step {step}: logits differ"
        for l in range(nl):
            assert np.array_equal(gpu_cache[l][0].numpy(), oracle_model.pools[l].k)
            assert np.array_equal(gpu_cache[l][1].numpy(), oracle_model.pools[l].v)
        names = [c[0] for c in calls]
        assert names.count("qs_single_query_attention") == nl and names.count("qs_flash_attn_varlen_fwd") == 0
        sqa = [c for c in calls if c[0] == "qs_single_query_attention"][0]
        assert sqa[1:] == (B, CFG["heads"], Hkv, mb, max(lens), int(int4))


def test_mirror_takes_alibi_slopes_and_rotary_style_and_ignores_them_like_the_reference(monkeypatch):
    """fused_attention.cpp:91,109: the reference's set_params leaves `linear_bias_slopes` and `neox_rotary_style` unset (both
    lines are commented out), update_kv_cache.cu:57 hard-codes the NeoX pairing: the arguments are legal and inert.  The mirror
    forwards the same calls to the C ABI with and without them (host simulator), and rejects an ill-shaped alibi tensor as
    fused_attention.cpp:193-199 does."""
    from oracle import synth
    _fake_abi.install(monkeypatch)
    import qserve_backend.fused_attention as fa
    B, H, Hkv = 2, 4, 2
    pr = synth.attention_problem(B, H, Hkv, [9, 70], seed=4)
    pb = kvattn.page_bytes(Hkv, 128, True)
    seq = (pr["lengths"] - 1).astype(np.int32)
    hist = np.concatenate(pr["hist"])
    cu = np.concatenate([[0], np.cumsum(seq)]).astype(np.int32)
    outs = []
    for neox, slopes in ((True, None), (False, torch.linspace(0.5, 2.0, H, dtype=torch.float32))):
        kp = torch.zeros((pr["nblocks"], pb), dtype=torch.uint8)
        vp = torch.zeros((pr["nblocks"], pb), dtype=torch.uint8)
        t = torch.from_numpy(pr["tables"].astype(np.int64))
        ptrs = torch.empty_like(t)
        ptrs[:, 0] = kp.data_ptr() + t[:, 0] * pb
        ptrs[:, 1] = vp.data_ptr() + t[:, 1] * pb
        qkv = torch.from_numpy(hist.copy())
        pad = fa.compute_padding_offsets(torch.from_numpy(cu), int(seq.max()), hist.shape[0])
        fa.apply_bias_rope_update_kv_cache(qkv, torch.from_numpy(seq), pad, ptrs, H, Hkv, int(seq.max()), 64, Hkv * 64, 128,
                                           5e5, 8192, neox, True, True)
        buf = torch.from_numpy(np.concatenate([pr["q"].reshape(B, -1), pr["k"].reshape(B, -1), pr["v"].reshape(B, -1)], 1))
        q, k, v = buf.split([H * 128, Hkv * 128, Hkv * 128], dim=-1)
        o = fa.single_query_attention(q.reshape(B, H, 128), k.reshape(B, Hkv, 128), v.reshape(B, Hkv, 128), ptrs,
                                      torch.from_numpy(pr["lengths"]), slopes, 8192, 64, Hkv * 64, int(pr["lengths"].max()),
                                      128, 5e5, neox, True, True)
        outs.append((qkv, kp, vp, o))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    q0 = torch.zeros((B, H, 128), dtype=torch.float16)
    k0 = torch.zeros((B, Hkv, 128), dtype=torch.float16)
    for bad in (torch.zeros(H + 1, dtype=torch.float32), torch.zeros(H, dtype=torch.float64)):
        with pytest.raises(RuntimeError):
            fa.single_query_attention(q0, k0, k0, ptrs, torch.from_numpy(pr["lengths"]), bad, 8192, 64, Hkv * 64, 70, 128, 5e5,
                                      True, True, True)
