"""qserve_amd -- MI355X (gfx950) implementation of QServe's W4A8KV4 hot path.

Layout:
  csrc/      hand-written HIP kernels + the C ABI (include/qserve_amd.h) -> libqserve_amd.so
  _lib.py    ctypes binding of that C ABI (no fallback: import fails loudly without the .so)
  backend/   host-side mirror of the reference's `qserve_backend.*` torch-extension modules
             (same callables / argument order / error behaviour); the top-level package
             `qserve_backend/` re-exports it under the reference's import names
  tp.py      tensor-parallel sharding of the packed weights + RCCL all-reduce
  decode.py  one Llama-style W4A8KV4 decode step expressed with the backend ops (bench / smoke driver)
"""
__version__ = "0.1.0"
