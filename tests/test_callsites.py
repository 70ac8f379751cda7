"""Every call the reference's Python makes into `qserve_backend.*` and `flash_attn_varlen_func` (recorded from the
reference tree by scripts/record_callsites.py -> tests/golden/callsites.json: file:line, positional count, keyword names)
binds against the mirror's signatures.  Runs anywhere (no reference, no GPU); where the reference IS present the
fixture is re-derived and must be identical, so it cannot go stale."""
import importlib
import inspect
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "callsites.json")
SITES = json.load(open(FIX))["sites"]


def resolve(module, function):
    name = module if module.startswith("flash_attn") else "qserve_backend." + module
    return getattr(importlib.import_module(name), function)


@pytest.mark.parametrize("site", SITES, ids=[s["site"] for s in SITES])
def test_call_site_binds(built_lib, site):
    fn = resolve(site["module"], site["function"])
    sig = inspect.signature(fn)
    args = [object()] * site["positional"]
    kwargs = {k: object() for k in site["keywords"]}
    sig.bind(*args, **kwargs)          # TypeError = the reference's call would not reach the mirror


def test_hot_path_call_sites_are_covered():
    """The W4A8KV4 model's own call sites (a1, a8, a14, a25, a26 and the activation-side ops) are in the
    fixture, with the positional shapes the C ABI lowering relies on."""
    got = {(s["module"], s["function"]): s for s in SITES if "w4a8" in s["site"] or "layers/" in s["site"]
           or "model_runner" in s["site"]}
    assert got[("qgemm_w4a8_per_chn", "gemm_forward_cuda")]["positional"] == 7
    assert got[("qgemm_w4a8_per_group", "gemm_forward_cuda")]["positional"] == 7
    assert got[("fused_attention", "single_query_attention")]["positional"] == 15
    assert got[("fused_attention", "apply_bias_rope_update_kv_cache")]["positional"] == 15
    assert got[("fused_attention", "compute_padding_offsets")]["positional"] == 3
    assert got[("layernorm_ops", "rms_norm_general_fuse_sum")]["positional"] == 7
    assert got[("fused_kernels", "invoke_quant_fuse_sum")]["positional"] == 4
    fl = [s for s in SITES if s["function"] == "flash_attn_varlen_func" and "llama_w4a8" in s["site"]][0]
    assert fl["positional"] == 3 and set(fl["keywords"]) == {"cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q", "max_seqlen_k",
                                                             "dropout_p", "causal"}


@pytest.mark.skipif(not os.path.isdir("/root/reference/qserve"), reason="reference tree not present")
def test_fixture_is_current(tmp_path):
    before = open(FIX).read()
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "record_callsites.py")], check=True, capture_output=True)
    assert open(FIX).read() == before, "tests/golden/callsites.json is stale: re-run scripts/record_callsites.py"
