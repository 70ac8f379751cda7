#!/usr/bin/env python3
"""Run the reference's OWN, UNCHANGED serving engine (staged under oracle/_ref/ by scripts/stage_reference.sh) over this
repository's `qserve_backend` on the GPU:  EngineArgs -> LLMEngine.from_engine_args -> add_request -> engine.step() ...
(qserve_benchmark.py:40-67; llm_engine.py:525; worker/model_runner.py:333-548,645; llama_w4a8_unpad.py:185-291,330-361).

Test / measurement infrastructure - nothing in the product imports this.  Two modes:

  --mode ragged   in-flight batching (ifb_mode=True) over a SMALL Llama shape with a checkpoint in the reference's format:
                  prompts of different lengths, different generation lengths (sequences finish at different steps, the
                  scheduler keeps batching the rest), greedy sampling as ModelRunner configures it.  Prints one JSON line:
                  the generated token ids per request, whether every logit the lm_head produced was finite, the number of
                  engine steps.  tests/test_reference_engine_gpu.py runs it once over the compiled extension
                  (--backend ext = qserve_backend_ext.install()) and once over the ctypes mirror and compares the tokens.
  --mode protocol the reference's benchmark protocol itself (qserve_benchmark.py:process_requests, imported from the staged
                  file and called unchanged): Llama-3-8B shape, random-initialised quantised weights (no checkpoint: the
                  reference's own `quant_path=None` path), batch x 1024 prompt tokens -> 512 generated tokens, non-IFB
                  benchmarking mode as the reference's scripts run it.  Prints tokens/s as the reference computes it.

What is NOT the reference's and has to exist around it on this box (none of it touches the staged files):
  * a model directory: config.json, a tokenizer (the engine loads one unconditionally) and, for `ragged`, model.safetensors;
  * transformers 5 keeps `rope_theta` inside `rope_parameters`; the reference (written against 4.37) reads
    `config.rope_theta` - a read-only property is added to the CONFIG CLASS of the installed transformers when missing;
  * NUM_GPU_PAGE_BLOCKS (the reference's own environment knob, model_runner.py:306-308) bounds the page pool.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def write_model_dir(path, cfg, state_dict=None):
    """config.json + tokenizer (+ weights) of a Llama-architecture checkpoint directory."""
    os.makedirs(path, exist_ok=True)
    conf = dict(architectures=["LlamaForCausalLM"], model_type="llama", hidden_size=cfg["hidden"],
                intermediate_size=cfg["inter"], num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                num_key_value_heads=cfg["kv_heads"], vocab_size=cfg["vocab"], rms_norm_eps=cfg["eps"],
                rope_theta=cfg["rope_theta"], max_position_embeddings=cfg.get("max_pos", 8192), torch_dtype="float16",
                hidden_act="silu", tie_word_embeddings=False, bos_token_id=1, eos_token_id=2, attention_bias=False)
    json.dump(conf, open(os.path.join(path, "config.json"), "w"))
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import Whitespace
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    for i in range(3, min(cfg["vocab"], 512)):
        vocab[f"t{i}"] = i
    tok = Tokenizer(WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = Whitespace()
    tok.save(os.path.join(path, "tokenizer.json"))
    json.dump(dict(tokenizer_class="PreTrainedTokenizerFast", bos_token="<s>", eos_token="</s>", unk_token="<unk>"),
              open(os.path.join(path, "tokenizer_config.json"), "w"))
    if state_dict is not None:
        from safetensors.torch import save_file
        save_file(state_dict, os.path.join(path, "model.safetensors"))


def transformers_compat():
    """The reference reads config.rope_theta (transformers 4.37); transformers 5 moved it into rope_parameters."""
    from transformers import LlamaConfig
    probe = LlamaConfig(rope_theta=12345.0)
    if not hasattr(probe, "rope_theta"):
        def _get(self):
            rp = getattr(self, "rope_parameters", None) or {}
            return float(rp.get("rope_theta", 10000.0))
        LlamaConfig.rope_theta = property(_get)
        return "LlamaConfig.rope_theta property added (transformers keeps it in rope_parameters)"
    return "none needed"


def import_reference(backend):
    assert os.path.isdir(os.path.join(REF, "qserve")), "oracle/_ref/ is empty: run scripts/stage_reference.sh"
    import qserve_backend                                        # this repository's ctypes mirror (the default)
    assert os.path.dirname(os.path.abspath(qserve_backend.__file__)).startswith(ROOT)
    if backend == "ext":
        import qserve_backend_ext
        qserve_backend_ext.install()                             # `import qserve_backend.<module>` -> the compiled extension
    sys.path.insert(0, REF)
    import qserve
    assert os.path.abspath(qserve.__file__).startswith(REF), "must be the staged reference package"
    from qserve import EngineArgs, LLMEngine, SamplingParams
    import qserve.modeling.models.llama_w4a8_unpad as model_mod
    assert os.path.abspath(model_mod.__file__).startswith(REF)
    import qserve_backend.fused_attention as fa
    return EngineArgs, LLMEngine, SamplingParams, type(fa).__name__ + ":" + getattr(fa, "__file__", "compiled sub-module of qserve_backend_ext._C")


TINY = dict(name="tiny-ckpt", hidden=256, heads=2, kv_heads=2, inter=512, layers=2, vocab=96, rope_theta=1e4, eps=1e-5)
LLAMA3_8B = dict(name="llama3-8b", hidden=4096, heads=32, kv_heads=8, inter=14336, layers=32, vocab=128256, rope_theta=5e5,
                 eps=1e-5)


def run_ragged(args):
    import numpy as np
    import torch
    compat = transformers_compat()
    EngineArgs, LLMEngine, SamplingParams, backend_kind = import_reference(args.backend)
    from test_loader import make_checkpoint                       # reference-format checkpoint (pinned packer)
    d = tempfile.mkdtemp(prefix="qs_ref_model_")
    write_model_dir(d, TINY, make_checkpoint(args.group_size, False, seed=4, cfg=TINY))
    os.environ.setdefault("NUM_GPU_PAGE_BLOCKS", "64")
    ea = EngineArgs(model=d, quant_path=d, precision="w4a8kv4" if not args.kv8 else "w4a8kv8", ifb_mode=True,
                    benchmarking=False, group_size=args.group_size, max_num_seqs=8)
    engine = LLMEngine.from_engine_args(ea)
    model = engine.driver_worker.model_runner.model
    finite = [True]
    calls = [0]

    def hook(_m, _inp, out):                                     # every logit the unchanged model produced
        calls[0] += 1
        if not bool(torch.isfinite(out).all()):
            finite[0] = False
    model.lm_head.register_forward_hook(hook)
    sampled = []                                                 # what the unchanged sampler returned, step by step
    model.sampler.register_forward_hook(lambda _m, _i, out: sampled.append([int(x) for x in out.reshape(-1).tolist()]))
    r = np.random.default_rng(7)
    # ragged in-flight batch: different prompt lengths (one exactly a page, one a page + 1, one spanning three pages),
    # different generation lengths: request 1 finishes after 3 tokens, request 3 after 5, the others run on
    prompts = [5, 64, 65, 150, 31]
    gens = [9, 3, 12, 5, 12]
    for i, (pl, gl) in enumerate(zip(prompts, gens)):
        ids = r.integers(3, TINY["vocab"], pl).tolist()
        ok = engine.add_request(str(i), prompt=None, prompt_token_ids=ids,
                                sampling_params=SamplingParams(temperature=0.0, max_tokens=gl, ignore_eos=True))
        assert ok
    finished_at, batch_sizes, steps = {}, [], 0
    with torch.no_grad():
        while engine.has_unfinished_requests():
            outs = engine.step()
            steps += 1
            batch_sizes.append(len(outs))
            for o in outs:
                if o["finished"]:
                    finished_at[o["key"]] = steps
            assert steps < 100
    torch.cuda.synchronize()
    print(json.dumps(dict(mode="ragged", backend=args.backend, backend_module=backend_kind, transformers_compat=compat,
                          prompts=prompts, generation_lengths=gens, engine_steps=steps, finished_at_step=finished_at,
                          batch_size_per_step=batch_sizes, sampled_tokens_per_step=sampled, lm_head_calls=calls[0],
                          all_logits_finite=finite[0])))


def trace_ops():
    """Debugging aid (--trace-ops): print every backend call with its tensor shapes and synchronise after it."""
    import importlib

    import torch
    for m in ("qgemm_w4a8_per_chn", "qgemm_w4a8_per_group", "fused_attention", "fused_kernels", "layernorm_ops", "activation_ops"):
        mod = importlib.import_module("qserve_backend." + m)
        for name in dir(mod):
            fn = getattr(mod, name)
            if name.startswith("_") or not callable(fn) or isinstance(fn, type):
                continue

            def wrap(fn=fn, label=m + "." + name):
                def inner(*a, **k):
                    print("[op]", label, [tuple(x.shape) if isinstance(x, torch.Tensor) else x for x in a], flush=True)
                    r = fn(*a, **k)
                    torch.cuda.synchronize()
                    return r
                return inner
            setattr(mod, name, wrap())
    import flash_attn.flash_attn_interface as fai
    orig = fai.flash_attn_varlen_func

    def fa(*a, **k):
        print("[op] flash_attn_varlen_func", [tuple(x.shape) if isinstance(x, torch.Tensor) else x for x in a], k, flush=True)
        r = orig(*a, **k)
        torch.cuda.synchronize()
        return r
    fai.flash_attn_varlen_func = fa
    import qserve.modeling.models.llama_w4a8_unpad as mm
    if hasattr(mm, "flash_attn_varlen_func"):
        mm.flash_attn_varlen_func = fa


def run_protocol(args):
    import torch
    compat = transformers_compat()
    EngineArgs, LLMEngine, SamplingParams, backend_kind = import_reference(args.backend)
    if args.trace_ops:
        trace_ops()
    sys.path.insert(0, REF)
    import qserve_benchmark as qb                                 # the reference's benchmark driver, unchanged
    cfg = dict(LLAMA3_8B)
    if args.layers:
        cfg["layers"] = args.layers
    d = tempfile.mkdtemp(prefix="qs_ref_model_")
    write_model_dir(d, cfg)
    blocks = args.batch * ((args.prompt_len + args.gen_len + 63) // 64 + 1) + 16
    os.environ["NUM_GPU_PAGE_BLOCKS"] = str(blocks)
    ea = EngineArgs(model=d, quant_path=None, precision="w4a8kv4", ifb_mode=False, benchmarking=True, group_size=-1,
                    max_num_seqs=max(256, args.batch))
    res = []
    with torch.no_grad():
        for rnd in range(args.rounds):
            engine = LLMEngine.from_engine_args(ea)
            engine.profiling_mode = True
            # quant_path=None is the reference's own dummy-weight path: its tensors are torch.empty.  Fill them with synthetic
            # random-quantised values of the right scale (parameters only, not code) so that activations stay O(1) through
            # 32 layers and the sampler always sees finite logits - the same synthetic model as bench.py's own engine
            model = engine.driver_worker.model_runner.model
            g = torch.Generator(device="cuda").manual_seed(0)
            sd = model.state_dict()
            for n, t in sd.items():
                if n.endswith("qweight"):
                    t.copy_(torch.randint(-128, 128, t.shape, generator=g, device=t.device, dtype=torch.int8))
                elif n.endswith("s1_scales"):
                    t.copy_((torch.rand(t.shape, generator=g, device=t.device) * 0.001 + 0.0005).to(t.dtype))
                    sd[n[:-len("s1_scales")] + "s1_szeros"].copy_((t.float() * 7.5).to(t.dtype))
                elif n.endswith("s1_szeros"):
                    pass
                elif "layernorm" in n or n.endswith("norm.weight"):
                    t.fill_(1.0)
                elif n.endswith("embed_tokens.weight"):
                    t.copy_((torch.randn(t.shape, generator=g, device=t.device) * 0.5).to(t.dtype))
                elif n.endswith("lm_head.weight"):
                    t.copy_((torch.randn(t.shape, generator=g, device=t.device) * 0.02).to(t.dtype))
                elif t.dtype == torch.float16:
                    t.copy_((torch.rand(t.shape, generator=g, device=t.device) * 0.01).to(t.dtype))
            finite = [True]
            model.lm_head.register_forward_hook(lambda _m, _i, out: finite.__setitem__(0, finite[0] and bool(torch.isfinite(out).all())))
            t_lis, num_tokens = qb.process_requests(engine, batch_size=args.batch, prompt_len=args.prompt_len,
                                                    generation_len=args.gen_len)
            res.append(dict(round=rnd, tokens=num_tokens, seconds=round(sum(t_lis), 4),
                            tokens_per_s=round(num_tokens / sum(t_lis), 1), all_logits_finite=finite[0]))
            del engine, model
            torch.cuda.empty_cache()
    print(json.dumps(dict(mode="protocol", backend=args.backend, backend_module=backend_kind, transformers_compat=compat,
                          model=cfg["name"], layers=cfg["layers"], batch=args.batch, prompt_len=args.prompt_len,
                          generation_len=args.gen_len, rounds=res,
                          reference_engine_tokens_per_s=res[-1]["tokens_per_s"],
                          note="qserve_benchmark.py:process_requests called unchanged: tokens = batch x generation_len "
                               "request outputs, wall time incl. the prompt step and the engine's Python per step")))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="ragged", choices=["ragged", "protocol"])
    ap.add_argument("--backend", default="ext", choices=["ext", "ctypes"])
    ap.add_argument("--group-size", type=int, default=-1)
    ap.add_argument("--kv8", action="store_true")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--prompt-len", type=int, default=1024)
    ap.add_argument("--gen-len", type=int, default=512)
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--trace-ops", action="store_true", help="debugging: print and synchronise every backend call")
    a = ap.parse_args()
    (run_ragged if a.mode == "ragged" else run_protocol)(a)
