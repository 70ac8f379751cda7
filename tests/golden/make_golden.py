#!/usr/bin/env python3
"""Generate golden vectors for the W4A8 weight packer by RUNNING the reference's own Python.

Runs only in the authoring container (needs /root/reference). It imports the reference file
    qserve/modeling/layers/quantized_linear/w4a8_linear.py
by path, with the two CUDA-only extension modules it imports at the top (w4a8_linear.py:7-8)
replaced by empty stubs and `torch.cuda.current_device` / `Tensor.cuda` neutralised so that the
module-level default argument (w4a8_linear.py:19) and the `.cuda()` at w4a8_linear.py:281 work on
CPU.  Nothing of the reference is copied: the script calls
`W4A8OF16LinearDynamicInputScale.from_linear` (w4a8_linear.py:136-332) on seeded inputs and stores
inputs + outputs in `tests/golden/w4a8_pack_*.npz`.  The oracle (`oracle/w4a8.py`) is pinned
against these files by `tests/test_oracle_golden.py`.

Usage:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/qserve/modeling/layers/quantized_linear/w4a8_linear.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_module():
    pkg = types.ModuleType("qserve_backend")
    pkg.__path__ = []
    sys.modules["qserve_backend"] = pkg
    for name in ("qgemm_w4a8_per_chn", "qgemm_w4a8_per_group"):
        m = types.ModuleType("qserve_backend." + name)
        sys.modules["qserve_backend." + name] = m
        setattr(pkg, name, m)
    torch.cuda.current_device = lambda: "cpu"  # default arg evaluated at import
    torch.Tensor.cuda = lambda self, *a, **k: self  # w4a8_linear.py:281
    spec = importlib.util.spec_from_file_location("ref_w4a8_linear", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_per_channel(mod, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randint(0, 16, (N, K), generator=g)
    z = torch.randint(0, 16, (N,), generator=g)
    s1 = (torch.rand((N,), generator=g) * 0.018 + 0.002).to(torch.float16)
    W = (q - z[:, None]).float() * s1.float()[:, None]
    lin = torch.nn.Linear(K, N, bias=False)
    lin.weight.data = W.clone()
    ql = mod.W4A8OF16LinearDynamicInputScale.from_linear(
        lin, 4, -1, s1_scale=s1.clone(), zeros=z.clone().to(torch.int8)
    )
    return dict(
        q=q.numpy().astype(np.uint8),
        z=z.numpy().astype(np.uint8),
        s1=s1.numpy(),
        qweight=ql.qweight.numpy().copy(),
        s1_scales=ql.s1_scales.numpy().copy(),
        s1_szeros=ql.s1_szeros.numpy().copy(),
    )


def make_per_group(mod, N, K, seed, G=128):
    g = torch.Generator().manual_seed(seed)
    ng = K // G
    # QoQ-style protective range: w8 = (q - z) * s2 in [-128, 127], q*s2 <= 255
    s2 = torch.randint(1, 9, (N, ng), generator=g)
    z = torch.randint(0, 16, (N, ng), generator=g)
    q = torch.randint(0, 16, (N, ng, G), generator=g)
    w8 = (q - z[..., None]) * s2[..., None]
    # keep inside int8 by clamping q where needed (still integer, still exact)
    lo = torch.ceil((-128.0 / s2.float()) + z.float()).clamp(0, 15).long()
    hi = torch.floor((127.0 / s2.float()) + z.float()).clamp(0, 15).long()
    q = torch.maximum(torch.minimum(q, hi[..., None]), lo[..., None])
    w8 = (q - z[..., None]) * s2[..., None]
    assert w8.min() >= -128 and w8.max() <= 127
    s1 = (torch.rand((N,), generator=g) * 0.018 + 0.002).to(torch.float16)
    W = w8.reshape(N, K).float() * s1.float()[:, None]
    lin = torch.nn.Linear(K, N, bias=False)
    lin.weight.data = W.clone()
    ql = mod.W4A8OF16LinearDynamicInputScale.from_linear(
        lin, 4, G, s1_scale=s1.clone(), s2_scale=s2.clone().float(), zeros=z.clone().float()
    )
    return dict(
        q=q.reshape(N, K).numpy().astype(np.uint8),
        z=z.numpy().astype(np.uint8),
        s2=s2.numpy().astype(np.uint8),
        s1=s1.numpy(),
        qweight=ql.qweight.numpy().copy(),
        s1_scales=ql.s1_scales.numpy().copy(),
        s2_scales=ql.s2_scales.numpy().copy(),
        s2_zeros=ql.s2_zeros.numpy().copy(),
    )


def sha(t):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()


def full_size_inputs(kind, N, K, seed):
    """Seeded inputs of the full-size digests; shared with tests/test_oracle_golden.py (which re-creates them from the
    seed - torch's CPU generator is deterministic for a given torch build - instead of storing 8 MB of nibbles)."""
    g = torch.Generator().manual_seed(seed)
    if kind == "per_chn":
        q = torch.randint(0, 16, (N, K), generator=g)
        z = torch.randint(0, 16, (N,), generator=g)
        s1 = (torch.rand((N,), generator=g) * 0.018 + 0.002).to(torch.float16)
        return dict(q=q, z=z, s1=s1)
    ng = K // 128
    s2 = torch.randint(1, 9, (N, ng), generator=g)
    z = torch.randint(0, 16, (N, ng), generator=g)
    q = torch.randint(0, 16, (N, ng, 128), generator=g)
    lo = torch.ceil((-128.0 / s2.float()) + z.float()).clamp(0, 15).long()
    hi = torch.floor((127.0 / s2.float()) + z.float()).clamp(0, 15).long()
    q = torch.maximum(torch.minimum(q, hi[..., None]), lo[..., None])
    s1 = (torch.rand((N,), generator=g) * 0.018 + 0.002).to(torch.float16)
    return dict(q=q.reshape(N, K), z=z, s2=s2, s1=s1)


def make_full_size_digests(mod):
    """One Llama-3-8B GEMM shape (o_proj, 4096 x 4096) per mode through the reference's from_linear: only SHA-256
    digests of its outputs are stored (the inputs are re-created from the seed by the test)."""
    import json
    N = K = 4096
    out = {"shape": [N, K], "torch": torch.__version__, "entries": {}}
    i = full_size_inputs("per_chn", N, K, 11)
    lin = torch.nn.Linear(K, N, bias=False)
    lin.weight.data = (i["q"] - i["z"][:, None]).float() * i["s1"].float()[:, None]
    ql = mod.W4A8OF16LinearDynamicInputScale.from_linear(lin, 4, -1, s1_scale=i["s1"].clone(),
                                                         zeros=i["z"].clone().to(torch.int8))
    out["entries"]["per_chn"] = dict(seed=11, qweight=sha(ql.qweight.numpy()), s1_scales=sha(ql.s1_scales.numpy()),
                                     s1_szeros=sha(ql.s1_szeros.numpy()))
    i = full_size_inputs("per_group", N, K, 12)
    w8 = (i["q"].reshape(N, K // 128, 128) - i["z"][..., None]) * i["s2"][..., None]
    lin = torch.nn.Linear(K, N, bias=False)
    lin.weight.data = w8.reshape(N, K).float() * i["s1"].float()[:, None]
    ql = mod.W4A8OF16LinearDynamicInputScale.from_linear(lin, 4, 128, s1_scale=i["s1"].clone(),
                                                         s2_scale=i["s2"].clone().float(), zeros=i["z"].clone().float())
    out["entries"]["per_group"] = dict(seed=12, qweight=sha(ql.qweight.numpy()), s1_scales=sha(ql.s1_scales.numpy()),
                                       s2_scales=sha(ql.s2_scales.numpy()), s2_zeros=sha(ql.s2_zeros.numpy()))
    with open(os.path.join(OUT, "w4a8_pack_llama3_8b_o_proj_digests.json"), "w") as f:
        json.dump(out, f, indent=1)


def main():
    mod = load_reference_module()
    make_full_size_digests(mod)
    np.savez_compressed(os.path.join(OUT, "w4a8_pack_per_chn_128x512.npz"), **make_per_channel(mod, 128, 512, 5))
    np.savez_compressed(os.path.join(OUT, "w4a8_pack_per_chn_64x128.npz"), **make_per_channel(mod, 64, 128, 1))
    np.savez_compressed(os.path.join(OUT, "w4a8_pack_per_chn_96x256.npz"), **make_per_channel(mod, 96, 256, 2))
    np.savez_compressed(os.path.join(OUT, "w4a8_pack_per_group_64x256.npz"), **make_per_group(mod, 64, 256, 3))
    np.savez_compressed(os.path.join(OUT, "w4a8_pack_per_group_128x384.npz"), **make_per_group(mod, 128, 384, 4))
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
