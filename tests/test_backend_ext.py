"""The boundary in the form the reference itself binds: qserve_backend_ext = a compiled pybind11 torch extension
(torch::Tensor arguments, qserve_backend_ext/csrc/binding.cpp) over libqserve_amd.so, built with g++ against the installed
torch (python -m qserve_backend_ext.build).

CPU: it builds, exports the reference's seven module names and callables (kernels/setup.py:157-245), and every call site
recorded from the reference tree (tests/golden/callsites.json) binds against the pybind signatures.
GPU: every op through the extension == the same op through the ctypes mirror, bit for bit."""
import inspect
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SITES = json.load(open(os.path.join(ROOT, "tests", "golden", "callsites.json")))["sites"]
OUT_OF_SCOPE = {"invoke_dequant", "invoke_dequant_add_residual", "invoke_dequant_add_residual_rms_norm_quant", "gelu_new",
                "gelu_fast", "invoke_dequant_silu_and_mul_quant"}


@pytest.fixture(scope="module")
def ext(built_lib):
    from qserve_backend_ext import build
    build.build(verbose=False)
    import qserve_backend_ext
    qserve_backend_ext.load()
    return qserve_backend_ext


def pybind_signature(fn):
    """inspect.Signature from the first line of a pybind11 docstring: name(a: T, b: T = default, ...) -> R."""
    line = fn.__doc__.strip().splitlines()[0]
    inner = line[line.index("(") + 1: line.rindex(")")]
    params, depth, cur = [], 0, ""
    for ch in inner:
        if ch in "[(":
            depth += 1
        if ch in "])":
            depth -= 1
        if ch == "," and depth == 0:
            params.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        params.append(cur)
    out = []
    for p in params:
        p = p.strip()
        if p.startswith("*"):                       # *args / **kwargs of the out-of-scope stubs
            kind = inspect.Parameter.VAR_KEYWORD if p.startswith("**") else inspect.Parameter.VAR_POSITIONAL
            out.append(inspect.Parameter(p.lstrip("*").split(":")[0].strip() or "a", kind))
            continue
        name = p.split(":")[0].strip()
        default = inspect.Parameter.empty if "=" not in p.split(":", 1)[-1] else None
        out.append(inspect.Parameter(name, inspect.Parameter.POSITIONAL_OR_KEYWORD, default=default))
    return inspect.Signature(out)


def test_extension_exports_the_reference_modules(ext):
    assert ext.MODULES == ["qgemm_w4a8_per_chn", "qgemm_w4a8_per_group", "qgemm_w8a8", "fused_attention", "fused_kernels",
                           "layernorm_ops", "activation_ops"]
    import importlib
    for m in ext.MODULES:
        mod = importlib.import_module("qserve_backend_ext." + m)
        mirror = importlib.import_module("qserve_backend." + m)
        for name in dir(mirror):
            fn = getattr(mirror, name)
            if callable(fn) and not name.startswith("_") and getattr(fn, "__module__", "").startswith("qserve_amd.backend") \
                    and not name.endswith("_acc") and name not in ("check", "expect", "guard", "ptr", "stream"):
                assert hasattr(mod, name), f"qserve_backend_ext.{m} lacks {name}"


@pytest.mark.parametrize("site", [s for s in SITES if not s["module"].startswith("flash_attn")],
                         ids=[s["site"] for s in SITES if not s["module"].startswith("flash_attn")])
def test_call_site_binds_against_the_compiled_extension(ext, site):
    fn = getattr(getattr(ext, site["module"]), site["function"])
    if site["function"] in OUT_OF_SCOPE:
        with pytest.raises(RuntimeError):           # the name exists and says why it is not provided
            fn(*([None] * site["positional"]))
        return
    sig = pybind_signature(fn)
    sig.bind(*([object()] * site["positional"]), **{k: object() for k in site["keywords"]})


def test_extension_rejects_what_the_reference_rejects(ext):
    x = torch.zeros((2, 8), dtype=torch.float16)
    with pytest.raises(RuntimeError):               # CHECK_DEVICE
        ext.layernorm_ops.rms_norm(x, x, torch.zeros(8, dtype=torch.float16), 1e-5)
    with pytest.raises(TypeError):                  # wrong arity
        ext.qgemm_w4a8_per_chn.gemm_forward_cuda(x, x)


# ---- GPU: extension == ctypes mirror, bit for bit ------------------------------------------------------------------------
@pytest.mark.gpu
def test_extension_equals_ctypes_mirror_on_the_device(gpu, ext):
    import importlib
    from oracle import synth
    from _helpers import dev
    mir = {m: importlib.import_module("qserve_amd.backend." + m) for m in ext.MODULES}
    e = {m: getattr(ext, m) for m in ext.MODULES}
    # GEMMs (decode + prefill shapes)
    for M, N, K in ((64, 4096, 4096), (300, 512, 1024), (2048, 4096, 4096)):
        g = torch.Generator(device=gpu).manual_seed(M)
        A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu, generator=g)
        W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu, generator=g)
        ws = (torch.rand((N,), device=gpu, generator=g) * 0.004 + 0.001).half()
        sa = (torch.rand((M,), device=gpu, generator=g) * 0.02 + 0.005).half()
        wz = (torch.randint(0, 16, (N,), device=gpu, generator=g).half() * ws).half()
        ss = (sa.float() * A.float().sum(1)).half()
        o1, o2 = torch.empty((M, N), dtype=torch.float16, device=gpu), torch.empty((M, N), dtype=torch.float16, device=gpu)
        mir["qgemm_w4a8_per_chn"].gemm_forward_cuda(A, W, ws, sa, wz, ss, o1)
        e["qgemm_w4a8_per_chn"].gemm_forward_cuda(A, W, ws, sa, wz, ss, o2)
        assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))
        s2 = torch.randint(1, 9, (K // 128, N), dtype=torch.int8, device=gpu, generator=g)
        z2 = torch.randint(-100, 1, (K // 128, N), dtype=torch.int8, device=gpu, generator=g)
        mir["qgemm_w4a8_per_group"].gemm_forward_cuda(A, W, z2, s2, ws, sa, o1)
        e["qgemm_w4a8_per_group"].gemm_forward_cuda(A, W, z2, s2, ws, sa, o2)
        assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))
        Wd = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=gpu, generator=g)
        mir["qgemm_w8a8"].w8a8_gemm_forward_cuda(A, Wd, ws, sa, o1)
        e["qgemm_w8a8"].w8a8_gemm_forward_cuda(A, Wd, ws, sa, o2)
        assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))
    # activation-side ops
    T, H = 64, 4096
    x = (torch.randn((T, H), device=gpu) * 3).half()
    w = (torch.rand((H,), device=gpu) + 0.5).half()

    def bufs():
        return (torch.empty((T, H), dtype=torch.int8, device=gpu), torch.full((T,), -1, dtype=torch.float16, device=gpu),
                torch.full((T,), -1, dtype=torch.float16, device=gpu))
    for name, call in (("invoke_quant", lambda m, q, sc, sm: m["fused_kernels"].invoke_quant(q, x, sc)),
                       ("invoke_quant_fuse_sum", lambda m, q, sc, sm: m["fused_kernels"].invoke_quant_fuse_sum(q, x, sm, sc)),
                       ("rms_norm_general", lambda m, q, sc, sm: m["layernorm_ops"].rms_norm_general(q, x, w, sc, 1e-5, True)),
                       ("rms_norm_general_fuse_sum",
                        lambda m, q, sc, sm: m["layernorm_ops"].rms_norm_general_fuse_sum(q, x, w, sm, sc, 1e-5, True))):
        a, b = bufs(), bufs()
        call(mir, *a)
        call(e, *b)
        for u, v in zip(a, b):
            assert torch.equal(u.view(torch.int8) if u.dtype == torch.int8 else u.view(torch.int16),
                               v.view(torch.int8) if v.dtype == torch.int8 else v.view(torch.int16)), name
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    mir["layernorm_ops"].rms_norm(o1, x, w, 1e-5)
    e["layernorm_ops"].rms_norm(o2, x, w, 1e-5)
    assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))
    a1, a2 = torch.empty((T, H // 2), dtype=torch.float16, device=gpu), torch.empty((T, H // 2), dtype=torch.float16, device=gpu)
    mir["activation_ops"].silu_and_mul(a1, x)
    e["activation_ops"].silu_and_mul(a2, x)
    assert torch.equal(a1.view(torch.int16), a2.view(torch.int16))
    # prefill writer + decode attention (KV4 and KV8), strided q / k / v views, None / given lengths
    from test_attention_gpu import ROPE, DevPools
    for int4 in (True, False):
        B, Hh, Hkv, L = 4, 32, 8, 200
        pr = synth.attention_problem(B, Hh, Hkv, [L, 65, 130, 1], seed=3)
        spt = Hkv * (64 if int4 else 128)
        res = []
        for mods in (mir, e):
            pools = DevPools(pr["nblocks"], Hkv, int4, gpu, fill=0)
            ptrs = pools.pointers(pr["tables"])
            seq = (pr["lengths"] - 1).astype("int32")
            import numpy as np
            hist = np.concatenate(pr["hist"])
            cu = np.concatenate([[0], np.cumsum(seq)]).astype("int32")
            pad = mods["fused_attention"].compute_padding_offsets(dev(cu), int(seq.max()), hist.shape[0])
            qkv = dev(hist)
            mods["fused_attention"].apply_bias_rope_update_kv_cache(qkv, dev(seq), pad, ptrs, Hh, Hkv, int(seq.max()), 64, spt, 128,
                                                                    ROPE, 8192, True, int4, True)
            new = dev(np.concatenate([pr["q"].reshape(B, -1), pr["k"].reshape(B, -1), pr["v"].reshape(B, -1)], axis=1))
            q, k, v = new.split([Hh * 128, Hkv * 128, Hkv * 128], dim=-1)
            out = mods["fused_attention"].single_query_attention(q.reshape(B, Hh, 128), k.reshape(B, Hkv, 128), v.reshape(B, Hkv, 128),
                                                                 ptrs, dev(pr["lengths"]), None, 8192, 64, spt, L, 128, ROPE, True,
                                                                 int4, True)
            res.append((pad.clone(), qkv.clone(), out.clone(), pools.k.clone(), pools.v.clone()))
        for u, v_ in zip(*res):
            assert torch.equal(u.view(torch.uint8) if u.dtype == torch.uint8 else u.view(torch.int16) if u.dtype == torch.float16 else u,
                               v_.view(torch.uint8) if v_.dtype == torch.uint8 else v_.view(torch.int16) if v_.dtype == torch.float16 else v_)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_extension_runs_inside_a_hipgraph_on_the_capturing_stream(gpu, ext):
    """The binding launches on the CURRENT stream (c10::hip::getCurrentHIPStream), so an op is capturable."""
    M, N, K = 64, 4096, 4096
    A = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=gpu)
    W = torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=gpu)
    ws = torch.full((N,), 0.002, dtype=torch.float16, device=gpu)
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=gpu)
    out, ref = torch.zeros((M, N), dtype=torch.float16, device=gpu), torch.zeros((M, N), dtype=torch.float16, device=gpu)
    ext.qgemm_w4a8_per_chn.gemm_forward_cuda(A, W, ws, sa, ws, sa, ref)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ext.qgemm_w4a8_per_chn.gemm_forward_cuda(A, W, ws, sa, ws, sa, out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))


@pytest.mark.gpu
def test_extension_rejects_mismatched_shapes(gpu, ext):
    """M and N come from out_feats, K from in_feats (as in the reference): everything else must fit, or the kernels would
    read out of bounds - the compiled boundary refuses instead of launching."""
    i8, f16 = torch.int8, torch.float16
    M, N, K = 16, 128, 256
    A = torch.zeros((M, K), dtype=i8, device=gpu)
    W = torch.zeros((N, K // 2), dtype=i8, device=gpu)
    vN, vM = torch.ones((N,), dtype=f16, device=gpu), torch.ones((M,), dtype=f16, device=gpu)
    out = torch.empty((M, N), dtype=f16, device=gpu)
    ext.qgemm_w4a8_per_chn.gemm_forward_cuda(A, W, vN, vM, vN, vM, out)                       # well-formed: runs
    with pytest.raises(RuntimeError, match="kernel"):
        ext.qgemm_w4a8_per_chn.gemm_forward_cuda(A, W[:, :-8].contiguous(), vN, vM, vN, vM, out)   # K mismatch
    with pytest.raises(RuntimeError, match="kernel"):
        ext.qgemm_w4a8_per_chn.gemm_forward_cuda(A, W[:-32].contiguous(), vN, vM, vN, vM, out)     # N mismatch
    with pytest.raises(RuntimeError, match="in_feats"):
        ext.qgemm_w4a8_per_chn.gemm_forward_cuda(A[:-1].contiguous(), W, vN, vM, vN, vM, out)      # rows != M
    with pytest.raises(RuntimeError, match="per output channel"):
        ext.qgemm_w4a8_per_chn.gemm_forward_cuda(A, W, vN[:-1].contiguous(), vM, vN, vM, out)
    with pytest.raises(RuntimeError, match="per token"):
        ext.qgemm_w4a8_per_chn.gemm_forward_cuda(A, W, vN, vM[:-1].contiguous(), vN, vM, out)
    x = torch.zeros((M, K), dtype=f16, device=gpu)
    q = torch.empty((M, K), dtype=i8, device=gpu)
    with pytest.raises(RuntimeError, match="as many values"):
        ext.fused_kernels.invoke_quant(q[:-1].contiguous(), x, vM)
    with pytest.raises(RuntimeError, match="per token"):
        ext.fused_kernels.invoke_quant(q, x, vM[:-2].contiguous())
    with pytest.raises(RuntimeError, match="hidden element"):
        ext.layernorm_ops.rms_norm_general(q, x, torch.ones((K - 8,), dtype=f16, device=gpu), vM, 1e-5, True)
    with pytest.raises(RuntimeError, match="half as many"):
        ext.activation_ops.silu_and_mul(torch.empty((M, K), dtype=f16, device=gpu), x)
