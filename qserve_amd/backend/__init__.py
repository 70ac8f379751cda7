"""Host-side mirror of the reference's `qserve_backend` extension modules (kernels/setup.py:157-245)."""
from . import (activation_ops, fused_attention, fused_kernels, layernorm_ops, qgemm_w4a8_per_chn,  # noqa: F401
               qgemm_w4a8_per_group, qgemm_w8a8)
