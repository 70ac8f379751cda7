"""See flash_attn/__init__.py: same callable, MI355X implementation."""
from qserve_amd.flash import flash_attn_varlen_func  # noqa: F401
