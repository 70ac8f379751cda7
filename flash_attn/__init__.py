"""Import shim: the reference imports `flash_attn.flash_attn_interface.flash_attn_varlen_func` (an un-vendored CUDA wheel,
README.md:77,104 of the reference).  With this repository on PYTHONPATH the name resolves to the MI355X provider in
qserve_amd/flash.py.  Only the varlen forward used by the reference's prefill path is provided."""
__version__ = "2.5.8+qserve_amd"
