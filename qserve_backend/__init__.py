"""Drop-in `qserve_backend` package: the import names the reference's Python uses
(`import qserve_backend.qgemm_w4a8_per_chn` etc., kernels/setup.py:157-245), backed by libqserve_amd.so."""
import importlib
import sys

_MODULES = ["qgemm_w4a8_per_chn", "qgemm_w4a8_per_group", "qgemm_w8a8", "fused_attention", "fused_kernels",
            "layernorm_ops", "activation_ops"]
for _m in _MODULES:
    _mod = importlib.import_module("qserve_amd.backend." + _m)
    sys.modules[__name__ + "." + _m] = _mod
    globals()[_m] = _mod
