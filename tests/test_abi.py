"""The C-ABI library builds for gfx950, loads, and exports every symbol include/qserve_amd.h declares.
No kernel is launched here (argument validation happens before any HIP call)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "qserve_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qs_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    for s in ["qs_w4a8_per_chn_gemm", "qs_w4a8_per_group_gemm", "qs_single_query_attention",
              "qs_apply_bias_rope_update_kv_cache", "qs_compute_padding_offsets", "qs_w8a8_gemm"]:
        assert s in syms


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} declared in include/qserve_amd.h but not exported"


def test_ctypes_binding_covers_header(built_lib):
    from qserve_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()
    assert _lib.lib.qs_arch() == b"gfx950" and _lib.lib.qs_version() >= 1


def test_argument_validation_without_gpu(built_lib):
    from qserve_amd._lib import check, lib
    rc = lib.qs_w4a8_per_chn_gemm(1, 1, 1, 1, 1, 1, 1, 4, 100, 128, None)     # N % 64 != 0
    assert rc == -1 and b"multiple of 64" in lib.qs_last_error()
    rc = lib.qs_w4a8_per_group_gemm(1, 1, 1, 1, 1, 1, 1, 4, 64, 96, None)     # K % 128 != 0
    assert rc == -1
    rc = lib.qs_w4a8_per_chn_gemm(0, 1, 1, 1, 1, 1, 1, 4, 64, 128, None)      # null pointer
    assert rc == -1
    rc = lib.qs_single_query_attention(1, 1, 1, 1, 1, 1, 2, 32, 8, 64, 6144, 6144, 4, 8192, 64, 512, 10, 64, 1e4, 1, 1, 1, None)
    assert rc == -2                                                           # head_dim 64 never instantiated
    with pytest.raises(RuntimeError):
        check(rc, "x")


def test_backend_modules_mirror_reference_names(built_lib):
    import qserve_backend
    import qserve_backend.fused_attention as fa
    import qserve_backend.qgemm_w4a8_per_chn as pc
    import qserve_backend.qgemm_w4a8_per_group as pg
    assert callable(pc.gemm_forward_cuda) and callable(pg.gemm_forward_cuda)
    for name in ("single_query_attention", "apply_bias_rope_update_kv_cache", "compute_padding_offsets"):
        assert callable(getattr(fa, name))
    for mod, fn in (("fused_kernels", "invoke_quant_fuse_sum"), ("layernorm_ops", "rms_norm_general_fuse_sum"),
                    ("activation_ops", "silu_and_mul"), ("qgemm_w8a8", "w8a8_gemm_forward_cuda")):
        assert callable(getattr(getattr(qserve_backend, mod), fn))


def test_product_does_not_import_oracle():
    bad = []
    for base in ("qserve_amd", "qserve_backend"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".h")):
                    if re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(dp, f)).read(), flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, f"product code imports the oracle: {bad}"
