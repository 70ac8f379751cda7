"""ctypes binding of libqserve_amd.so -- the C ABI declared in include/qserve_amd.h.

The product path has NO fallback: if the shared library is missing or fails to load, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# QS_AMD_LIBRARY: measurement scripts point this at the -DQS_TIMING build (qserve_amd/libqserve_amd_timing.so, same ABI
# plus ablation switches); the product, the tests and bench.py never set it
LIB_PATH = os.environ.get("QS_AMD_LIBRARY") or os.path.join(_HERE, "libqserve_amd.so")

# name -> (restype, argtypes); must list every symbol of include/qserve_amd.h (tests/test_abi.py checks that)
_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
SIGNATURES = {
    "qs_version": (_i, []),
    "qs_arch": (C.c_char_p, []),
    "qs_last_error": (C.c_char_p, []),
    "qs_w4a8_per_chn_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "qs_w4a8_per_group_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "qs_w4a8_per_chn_gemm_acc": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "qs_w4a8_per_chn_gemm_silu_mul": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "qs_w4a8_per_group_gemm_silu_mul": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "qs_w4a8_per_group_gemm_acc": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "qs_w8a8_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "qs_set_gemm_variant": (None, [_i]),
    "qs_set_gemm_epilogue": (_i, [_i]),
    "qs_get_gemm_epilogue": (_i, []),
    "qs_debug_gemm_clock_probe": (_i, [_vp, _i]),
    "qs_set_row_sum_order": (_i, [_i]),
    "qs_get_row_sum_order": (_i, []),
    "qs_w4a8_gemm_plan": (_i, [_i, _i, _i, _i, _vp]),
    "qs_w4a8_gemm_planes_plan": (_i, [_i, _i, _i, _i, _vp]),
    "qs_w4a8_per_chn_gemm_planes": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "qs_w4a8_per_group_gemm_planes": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "qs_set_attention_variant": (None, [_i]),
    "qs_attention_plan": (_i, [_i, _i, _i, _i, _i, _i, _vp]),
    "qs_single_query_attention": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i, _i, _i, _i, _i,
                                       _i, _f, _i, _i, _i, _vp]),
    "qs_single_query_attention_quant": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i, _i,
                                             _i, _i, _i, _i, _f, _i, _i, _i, _vp]),
    "qs_apply_bias_rope_update_kv_cache": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i,
                                                _i, _i, _vp]),
    "qs_compute_padding_offsets": (_i, [_vp, _vp, _i, _i, _vp]),
    "qs_invoke_quant": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "qs_rms_norm_general": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "qs_rms_norm": (_i, [_vp, _vp, _vp, _f, _i, _i, _vp]),
    "qs_silu_and_mul": (_i, [_vp, _vp, _i, _i, _vp]),
    "qs_residual_add": (_i, [_vp, _vp, _i64, _vp]),
    "qs_argmax_rows": (_i, [_vp, _vp, _i, _i, _i64, _vp]),
    "qs_debug_argmax_split": (None, [_i]),
    "qs_add_residual_rms_norm_general": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "qs_add_residual_rms_norm_general_planes": (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "qs_silu_and_mul_quant": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "qs_debug_wave_reduce_selftest": (_i, [_vp, _vp, _i, _vp]),
    "qs_device_status": (_i, [C.POINTER(C.c_int)]),
    "qs_device_reset": (_i, []),
    "qs_debug_inject_fault": (_i, [_i]),
    "qs_stream_scratch_bind": (_i, [_vp]),
    "qs_stream_scratch_unbind": (_i, [_vp]),
    "qs_comm_create": (_i, [_i, _i, _i64, C.POINTER(C.c_void_p), _vp]),
    "qs_comm_connect": (_i, [_vp, _vp]),
    "qs_comm_connect_local": (_i, [_vp, C.POINTER(C.c_void_p)]),
    "qs_comm_input": (_vp, [_vp]),
    "qs_comm_output": (_vp, [_vp]),
    "qs_comm_all_reduce_f16": (_i, [_vp, _i64, _vp]),
    "qs_comm_all_reduce_f16_group": (_i, [C.POINTER(C.c_void_p), _i, _i64, _vp]),
    "qs_comm_error": (_i, [_vp]),
    "qs_comm_destroy": (_i, [_vp]),
    "qs_debug_copy_split_workspace": (_i, [_vp, C.c_size_t]),
    "qs_debug_flash_variant": (_i, [_i]),
    "qs_flash_attn_varlen_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _i, _f,
                                      _i, _vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m qserve_amd.build` (hipcc --offload-arch=gfx950). "
            "qserve_amd has no CPU / PyTorch fallback."
        )
    lib = C.CDLL(LIB_PATH)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)   # AttributeError if the symbol is missing: loud by design
        except AttributeError:
            # measurement scripts loading an OLDER build for an in-run A/B ask for it explicitly (QS_AMD_LIBRARY_AB=1): entry
            # points added since are simply absent there.  A mere QS_AMD_LIBRARY (a wrong or stale path) stays an error.
            if os.environ.get("QS_AMD_LIBRARY") and os.environ.get("QS_AMD_LIBRARY_AB") == "1":
                missing.append(name)
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    if missing:
        import sys
        print(f"[qserve_amd] QS_AMD_LIBRARY_AB=1: {LIB_PATH} lacks {len(missing)} entry point(s): {', '.join(missing)}",
              file=sys.stderr)
    return lib


lib = _load()


def device_status():
    """Error bits of the bounded in-launch waits on the current device (0 = healthy); blocking.  include/qserve_amd.h."""
    if not hasattr(lib, "qs_device_status"):       # (only possible under QS_AMD_LIBRARY_AB=1: an older build in an A/B)
        raise RuntimeError(f"{LIB_PATH} has no qs_device_status: the health of its bounded waits cannot be read")
    bits = C.c_int(0)
    rc = lib.qs_device_status(C.byref(bits))
    if rc != 0:
        msg = lib.qs_last_error()
        raise RuntimeError(f"qs_device_status: {msg.decode() if msg else 'error'} (code {rc})")
    return bits.value


def check(rc, what):
    """0 -> ok; otherwise raise RuntimeError like the reference's TORCH_CHECK does."""
    if rc != 0:
        msg = lib.qs_last_error()
        raise RuntimeError(f"{what}: {msg.decode() if msg else 'error'} (code {rc})")
