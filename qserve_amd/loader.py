"""Checkpoint path for the W4A8 models: reference-format state dicts -> the per-rank tensors the
kernels consume, with the q/k/v and gate/up fusion of the reference's `load_weights` and tile-aware tensor-parallel
sharding.

What the reference does (qserve/modeling/models/llama_w4a8_unpad.py:487-630, qserve/utils/weight_utils.py:88-260):
the checkpoint stores every projection separately packed - `...self_attn.{q,k,v,o}_proj.*`, `...mlp.{gate,up,down}_proj.*`
with `qweight` int8 [N, K/2], `s1_scales` f16 [N], and `s1_szeros` f16 [N] (per-channel) or `s2_scales` / `s2_zeros` int8
[K/128, N] (g128); `load_weights` copies q, k, v into row ranges of the fused `qkv_proj` buffers (dim 1 for the
[K/128, N] tensors) and gate, up into the halves of `gate_up_proj`; names containing "norm" are skipped (the layer-norm
weights stay at their initial ones - QoQ folds them into the quantised weights), `rotary_emb.inv_freq` is skipped,
embeddings and `lm_head` are loaded as fp16.  Its tensor-parallel scaffolding is inert (tp_size = 1) and its generic
row-parallel slice `[:, start:end]` would be wrong for the packed layout.

This module keeps the packed layout AS IS (no re-tiling: every kernel consumes the reference's byte order) and adds
the tensor-parallel rules of SURVEY 8e:
  * column-parallel (qkv, gate_up): each of q / k / v (gate / up) is sliced per rank on whole heads (128 rows, a
    multiple of the 32-row tile) BEFORE concatenation - a contiguous slice of the fused tensor would hand rank 0 only
    query heads; KV heads are replicated when there are fewer than ranks (llama_w4a8_unpad.py:118-127);
  * row-parallel (o, down): `tp.shard_row_parallel` slices the [N/32][K/32][512] tile view on its K axis; `s2_*` on
    dim 0; `s1_*` replicated; a bias is replicated too and added ONCE by every rank after the all-reduce
    (`W4A8Linear.defer_bias`), so all ranks keep identical hidden states.
Pure tensor indexing (any device); nothing here launches a kernel.
"""
import glob
import os

import torch

from . import tp

LINEAR_TENSORS = ("qweight", "s1_scales", "s1_szeros", "s2_scales", "s2_zeros", "bias")


def iterate_checkpoint(path):
    """(name, tensor) over a checkpoint directory or file: *.safetensors first, else *.bin / *.pt (the order of
    weight_utils.py:88-170 `hf_model_weights_iterator`)."""
    if os.path.isdir(path):
        files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if not files:
            files = [f for f in sorted(glob.glob(os.path.join(path, "*.bin")) + glob.glob(os.path.join(path, "*.pt")))
                     if not f.endswith(("training_args.bin", "optimizer.bin", "optimizer.pt", "scheduler.pt", "scaler.pt"))]
    else:
        files = [path]
    if not files:
        raise RuntimeError(f"Cannot find any model weights with `{path}`")
    for f in files:
        if f.endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(f, framework="pt") as sf:
                for name in sf.keys():
                    yield name, sf.get_tensor(name)
        else:
            state = torch.load(f, map_location="cpu")
            for name, t in state.items():
                yield name, t


def _linear(state, prefix):
    d = {k: state[f"{prefix}.{k}"] for k in LINEAR_TENSORS if f"{prefix}.{k}" in state}
    if "qweight" not in d:
        raise KeyError(f"checkpoint has no {prefix}.qweight")
    return d


def _rows(d, r0, r1):
    """Rows [r0, r1) of one separately packed projection (N split at multiples of 32)."""
    assert r0 % 32 == 0 and r1 % 32 == 0, "column-parallel split must fall on 32-row tiles"
    out = {}
    for k, t in d.items():
        out[k] = t[:, r0:r1] if k in ("s2_scales", "s2_zeros") else t[r0:r1]
    return out


def _concat(parts):
    out = {}
    for k in parts[0]:
        out[k] = torch.cat([p[k] for p in parts], dim=1 if k in ("s2_scales", "s2_zeros") else 0).contiguous()
    return out


def fuse_column_parallel(parts, sizes, rank=0, world=1, replicas=None):
    """[q, k, v] or [gate, up] (dicts of one projection's tensors) -> the fused per-rank tensors.
    sizes[i] = rows of part i; part i is cut into `world // replicas[i]` shards and rank r takes shard r // replicas[i]
    (replicas > 1 = KV heads shared by several ranks, llama_w4a8_unpad.py:536-571)."""
    replicas = replicas or [1] * len(parts)
    cut = []
    for d, n, rep in zip(parts, sizes, replicas):
        shards = world // rep
        assert n % shards == 0
        per = n // shards
        sid = rank // rep
        cut.append(_rows(d, sid * per, (sid + 1) * per))
    return _concat(cut)


def shard_row(d, rank, world, group_size):
    """Row-parallel shard of one projection: qweight tiles on K, s2_* on dim 0, s1_* and bias replicated (the bias is
    added after the reduce, not per partial)."""
    mats = [d[k] for k in ("s2_scales", "s2_zeros") if k in d]
    qw, mats = tp.shard_row_parallel(d["qweight"], mats, rank, world, group_size=group_size)
    out = {"qweight": qw, "s1_scales": d["s1_scales"]}
    if "s1_szeros" in d:
        out["s1_szeros"] = d["s1_szeros"]
    for k, m in zip([k for k in ("s2_scales", "s2_zeros") if k in d], mats):
        out[k] = m
    if "bias" in d:
        out["bias"] = d["bias"]
    return out


def load_llama_w4a8(state, cfg, group_size=-1, tp_rank=0, tp_world=1, load_norm_weights=False, device=None):
    """state: {name: tensor} in the reference checkpoint naming (or an iterable of pairs, e.g. iterate_checkpoint(dir)).
    cfg: dict with hidden, heads, kv_heads, inter, layers (qserve_amd.decode configs).
    Returns {"layers": [{"qkv","o","gate_up","down": {tensor dicts}, "ln1","ln2"}], "norm", "embed", "lm_head"} for this
    rank.  `load_norm_weights=False` mirrors the reference (`if "norm" in name: continue`): norms stay ones."""
    if not isinstance(state, dict):
        state = dict(state)
    H, Hkv, hid = cfg["heads"], cfg["kv_heads"], cfg["hidden"]
    hd = hid // H
    assert hd == 128, "the W4A8KV4 kernels are built for head_dim 128"
    assert H % tp_world == 0 and cfg["inter"] % (tp_world * 128) == 0
    kv_rep = max(1, tp_world // Hkv)
    assert (Hkv % tp_world == 0) if Hkv >= tp_world else (tp_world % Hkv == 0)

    def dev(d):
        return {k: (v.to(device) if device is not None else v).contiguous() for k, v in d.items()}

    def ones():
        return torch.ones((hid,), dtype=torch.float16)

    layers = []
    for li in range(cfg["layers"]):
        p = f"model.layers.{li}."
        q, k, v = (_linear(state, p + f"self_attn.{n}_proj") for n in ("q", "k", "v"))
        qkv = fuse_column_parallel([q, k, v], [H * hd, Hkv * hd, Hkv * hd], tp_rank, tp_world, [1, kv_rep, kv_rep])
        gate, up = (_linear(state, p + f"mlp.{n}_proj") for n in ("gate", "up"))
        gate_up = fuse_column_parallel([gate, up], [cfg["inter"]] * 2, tp_rank, tp_world)
        o = shard_row(_linear(state, p + "self_attn.o_proj"), tp_rank, tp_world, group_size)
        down = shard_row(_linear(state, p + "mlp.down_proj"), tp_rank, tp_world, group_size)
        ln1 = state[p + "input_layernorm.weight"].half() if load_norm_weights and p + "input_layernorm.weight" in state else ones()
        ln2 = (state[p + "post_attention_layernorm.weight"].half()
               if load_norm_weights and p + "post_attention_layernorm.weight" in state else ones())
        layers.append(dict(qkv=dev(qkv), o=dev(o), gate_up=dev(gate_up), down=dev(down),
                           ln1=ln1.to(device) if device is not None else ln1,
                           ln2=ln2.to(device) if device is not None else ln2))
    out = dict(layers=layers)
    norm = state["model.norm.weight"].half() if load_norm_weights and "model.norm.weight" in state else ones()
    for name, key, t in (("norm", None, norm), ("embed", "model.embed_tokens.weight", None), ("lm_head", "lm_head.weight", None)):
        if t is None:
            if key not in state:
                continue
            t = state[key].half()
        out[name] = t.to(device) if device is not None else t
    return out
